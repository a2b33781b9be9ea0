// tables.hpp — host-side preprocessing for the scan kernels (plain C++).
//
// Each builder produces the table of the cited SMART algorithm; the kernels
// stage these tables in LDS.  Layouts are the device layouts (narrow integer
// types, flags folded in) — `*_ref32` variants give the reference's int layout
// for the table-parity tests (smartgpu_build_table).
#pragma once
#include <cstdint>
#include <vector>

namespace sg {

constexpr uint32_t kSigma = 256;   // src/algos/include/define.h:27
constexpr uint32_t kXSize = 4200;  // src/algos/include/define.h:25

// Horspool / BM bad-character shifts (hor.c:26-30, bm.c:27-33): for every byte c
// the distance from its right-most occurrence in P[0..m-2] to P[m-1], m if none.
std::vector<int32_t> bad_char(const uint8_t* P, uint32_t m);

// BM good-suffix shifts (bm.c:36-66) via the suffix-length array.
std::vector<int32_t> good_suffix(const uint8_t* P, uint32_t m);

// KMP strong failure links next[0..m] (kmp.c:27-41), next[0] = -1.
std::vector<int32_t> kmp_next(const uint8_t* P, uint32_t m);

// KMP automaton (the failure function expanded into the transition table, Knuth-Morris-
// Pratt's delta): dfa[s*256 + c] = state after reading c in state s, states 0..m (m <= 255),
// state m = "an occurrence ends here".  (m+1)*256 bytes.
std::vector<uint8_t> kmp_dfa(const uint8_t* P, uint32_t m);

// The same table over the pattern's own alphabet only: colmap[c] = column of byte c
// (0 = "c does not occur in P": every state goes to 0), table[s*k1 + col], k1 columns.
// Returns colmap (256 bytes) followed by the table; *k1 = row stride.
std::vector<uint8_t> kmp_dfa_compressed(const uint8_t* P, uint32_t m, uint32_t* k1);
// kmp_runs' tables for the automaton of P[0..w), w <= 254, appended to `out` (layout: DESIGN.md §4, k_kmp.hip
// kmp_runs): the transitions — row id(s) XOR-swizzled by its id, every transition into the accept state leading to
// the absorbing row Z; kmp_runs_table_bytes(w) bytes: the w+1 rows of the states one after the other while their ids
// are 4s (w < 63; the kernel spreads them over (Z+1)*256 bytes of LDS and fills row Z), else all 256 rows —, then
// 272 bytes: Q[s] = P[s..s+4) for the states 0..K that have no border (64 dwords), thr = 4K, 12 bytes of padding.
// Built row by row from the border's row like kmp_dfa, in place.
// compact (w <= 62; kmp_runs<., false, COMPACT>, which keeps the rows as stored: row s at s * 256): the absorbing row's
// id is a multiple of 4 like every other, Z = 4 (w + 1) — the only difference, in the entries that lead INTO it.
void kmp_runs_tables(const uint8_t* P, uint32_t w, std::vector<uint8_t>& out, bool compact = false);
uint32_t kmp_runs_table_bytes(uint32_t w);
// Two-bit codes for a set of at most four byte values (bit c of the 256-bit set <=> value c is a member): the lowest
// shift < 7 such that (c >> shift) & 3 tells the members apart, and symtab = the member of each code in byte `code`
// (a code without a member: a byte value with that code — not a member).  false: more than four members, or no such shift.
bool four_symbol_codes(const uint32_t set[8], uint32_t* shift, uint32_t* symtab);

// Shift-Or: S[c] has bit i clear iff P[i]==c, over the first w=min(m,32) bytes
// (so.c:27-38,73-74).  The hit test "D < lim" (so.c:56) is "bit w-1 of D is 0".
std::vector<uint32_t> shift_or_masks(const uint8_t* P, uint32_t m);

// BNDM: B[c] has bit (w-1-i) set iff P[i]==c, w=min(m,32) (bndm.c:35-40,74-75).
std::vector<uint32_t> bndm_masks(const uint8_t* P, uint32_t m);

// Shift-And: S[c] has bit i SET iff P[i]==c, over the first w=min(m,32) bytes (sa.c:27-34,72).
std::vector<uint32_t> shift_and_masks(const uint8_t* P, uint32_t m);

// Quick Search shifts (qs.c:27-31): for the byte c that FOLLOWS the window, m+1 if c does not
// occur in P, else m minus its right-most position.
std::vector<int32_t> quick_search_shifts(const uint8_t* P, uint32_t m);

// HASHq (hash3.c:36-56, hash5.c, hash8.c): shift[h] for the 8-bit hash h = sum y[i-k]*2^k mod 256 of the
// window's last q bytes; the hash of the pattern's last q-gram gets 0 and `*after` the shift it
// replaced (at least 1).  m >= q.
std::vector<int32_t> qgram_hash_shifts(const uint8_t* P, uint32_t m, uint32_t q, int32_t* after);

}  // namespace sg
