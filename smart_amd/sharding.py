"""Byte-offset sharding of a text across ranks.

Occurrences are independent per start offset, so a text shards with no exchange
step: rank g OWNS the start positions [a_g, a_{g+1}) and must be able to read
m-1 bytes past its last start.  The per-rank counts add up to the global count;
the only collective is the sum of the counts.
"""


def split_starts(n, m, rank, world):
    """Strong split of one text of n bytes: returns (byte_off, byte_len) of the
    slice rank `rank` has to hold and search in full (i.e. call
    search(off=byte_off, n=byte_len)); byte_len < m means "nothing to do"."""
    if m < 1 or n < m:
        return 0, 0
    starts = n - m + 1
    a = starts * rank // world
    b = starts * (rank + 1) // world
    if b <= a:
        return a, 0
    return a, (b - a) + m - 1


def weak_shard(shard_bytes, m, rank, world):
    """Weak scaling: the global text is world*shard_bytes long and rank g owns
    the starts [g*S, (g+1)*S) (the last rank: up to total-m).  Returns
    (global_off, local_len): the rank generates/holds global bytes
    [global_off, global_off+local_len) and searches all of it."""
    last = rank == world - 1
    return rank * shard_bytes, shard_bytes if last else shard_bytes + m - 1
