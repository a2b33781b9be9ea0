"""Corpora as SMART's harness loads them (host side, no GPU).

`get_text(path, tsize)` restates getText (src/smart.c:95-138): the files named between '#' marks in
<path>/index.txt, in order, concatenated and cut at tsize bytes — what `smart -text <name>` searches.

`english_unit()` returns BASELINE config 4's unit, data/englishTexts loaded that way (bible.txt then
world192.txt, 6,520,792 bytes), from the committed data fixture tests/golden/english_bible_world192.txt.xz
(the reference tree does not travel to the GPU box; tests/golden/gen_golden.py wrote the fixture from it
and records its md5 in english_corpus_vectors.json).
"""
import hashlib
import json
import lzma
import os

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_GOLDEN = os.path.join(_ROOT, "tests", "golden")
ENGLISH_BYTES = 6520792
_cache = {}


def get_text(path, tsize=1 << 30):
    out = bytearray()
    with open(os.path.join(path, "index.txt"), "rb") as f:
        idx = f.read()
    i = 0
    while i < len(idx) and len(out) < tsize:
        if idx[i:i + 1] == b"#":
            j = idx.index(b"#", i + 1)
            with open(os.path.join(path, idx[i + 1:j].decode()), "rb") as g:
                out += g.read(tsize - len(out))
            i = j + 1
        else:
            i += 1
    return np.frombuffer(bytes(out), dtype=np.uint8)


def english_unit():
    """bible.txt || world192.txt as one uint8 array (checked against the recorded md5)."""
    if "english" not in _cache:
        with open(os.path.join(_GOLDEN, "english_bible_world192.txt.xz"), "rb") as f:
            data = lzma.decompress(f.read())
        with open(os.path.join(_GOLDEN, "english_corpus_vectors.json")) as f:
            meta = json.load(f)
        if len(data) != ENGLISH_BYTES or hashlib.md5(data).hexdigest() != meta["md5"]:
            raise RuntimeError("english_bible_world192.txt.xz does not decompress to the recorded corpus")
        _cache["english"] = np.frombuffer(data, dtype=np.uint8)
    return _cache["english"]
