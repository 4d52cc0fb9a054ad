"""fst::Set bytes -> word list.  Three layers:
  * the oracle (oracle/fst_oracle.py) against the five blobs milli itself wrote (tests/golden/index_blobs.json, from
    the reference's v1.12 upgrade-test index): decode == the keys of the databases they were built from, the
    builder reproduces every blob byte for byte, checksums match;
  * the product decoder (msi_fst_decode in libmsi.so — host code, callable without a GPU) against the same blobs
    and, key for key, against the oracle on large dictionaries that exercise what the blobs do not (256-byte
    transition index, uncommon inputs, multi-byte deltas, UTF-8, binary keys, the empty key);
  * malformed input: truncations, bit flips with and without the checksum — an error or a valid decode, never a
    crash or an unsorted list."""
import json
import os
import random

import numpy as np
import pytest

from meilisearch_amd import _lib, synth
from meilisearch_amd import typo as T
from oracle import fst_oracle as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FIX = json.load(open(os.path.join(ROOT, "tests", "golden", "index_blobs.json")))


def product_keys(blob, flags=0):
    concat, off = T.fst_decode(blob, flags)
    raw = concat.tobytes()
    return [raw[off[i]:off[i + 1]] for i in range(len(off) - 1)]


@pytest.mark.parametrize("entry", FIX["fst"], ids=[e["name"] for e in FIX["fst"]])
def test_blobs_written_by_milli(entry):
    blob = bytes.fromhex(entry["hex"])
    keys = F.fst_keys(blob)
    assert keys == sorted(set(keys))
    if entry["keys"] is not None:      # facet FST: exactly the normalised values of facet-id-normalized-string-strings
        assert [k.decode() for k in keys] == entry["keys"]
    if entry["contains"] is not None:  # words-fst: every key of word-docids
        assert set(entry["contains"]) <= {k.decode() for k in keys}
    assert F.fst_build(keys) == blob, "the builder restatement must reproduce milli's bytes"
    assert product_keys(blob) == keys


def test_known_key_sets_of_the_golden_index():
    by_name = {e["name"]: F.fst_keys(bytes.fromhex(e["hex"])) for e in FIX["fst"]}
    assert by_name["main[stop-words]"] == [b"le", b"un"]
    assert by_name["main[exact-words]"] == [b"kefir"]
    assert by_name["main[words-prefixes-fst]"] == []
    assert len(by_name["main[words-fst]"]) == 20


def dictionaries():
    rng = random.Random(7)
    words = synth.make_dictionary(60000, seed=11)
    yield "synthetic-60k", [w.encode() if isinstance(w, str) else bytes(w) for w in words]
    # every byte value as a first and second input: 256-transition nodes (n byte = 1), index tables, uncommon inputs
    yield "binary-2-bytes", sorted({bytes([a, b]) for a in range(256) for b in range(0, 256, 3)} | {bytes([a]) for a in range(256)})
    yield "empty-key-first", [b""] + sorted({bytes(rng.choices(b"abcxyz", k=rng.randint(1, 6))) for _ in range(500)})
    yield "only-empty-key", [b""]
    yield "one-long-key", [bytes(rng.randrange(256) for _ in range(5000))]
    yield "40-siblings", sorted({b"k" + bytes([65 + i]) + b"tail" for i in range(40)})   # 33..63 transitions: index, inline n
    yield "70-siblings", sorted({b"p" + bytes([40 + i]) for i in range(70)} | {b"p"})     # n in the extra byte, final node
    # shared suffixes far apart: multi-byte deltas
    yield "far-suffixes", sorted({("%05d" % i).encode() + b"-common-suffix" for i in range(0, 30000, 7)})
    yield "utf8", sorted({w.encode() for w in ["café", "cafe", "собака", "собак", "日本語", "日本", "naïve", "über", "z"]})


@pytest.mark.parametrize("name,keys", list(dictionaries()), ids=[n for n, _ in dictionaries()])
def test_product_decoder_matches_the_oracle_on_built_dictionaries(name, keys):
    blob = F.fst_build(keys)
    if len(keys) <= 5000:
        assert F.fst_keys(blob) == keys
    assert product_keys(blob) == keys


def test_sizing_call_and_small_buffers():
    import ctypes as C
    keys = [b"alpha", b"beta", b"gamma"]
    blob = np.frombuffer(F.fst_build(keys), dtype=np.uint8)
    L = _lib.lib()
    n, nb = C.c_uint32(0), C.c_uint64(0)
    ptr = blob.ctypes.data_as(C.c_void_p)
    assert L.msi_fst_decode(ptr, blob.size, 0, None, 0, None, 0, C.byref(n), C.byref(nb)) == 0
    assert (n.value, nb.value) == (3, 14)
    concat, off = np.zeros(14, np.uint8), np.zeros(4, np.uint32)
    for cap_b, cap_w in ((13, 3), (14, 2)):
        assert L.msi_fst_decode(ptr, blob.size, 0, concat.ctypes.data_as(C.c_void_p), cap_b, off.ctypes.data_as(C.c_void_p),
                                cap_w, C.byref(n), C.byref(nb)) == -1   # MSI_E_INVALID: buffers too small
    assert L.msi_fst_decode(ptr, blob.size, 0, concat.ctypes.data_as(C.c_void_p), 14, off.ctypes.data_as(C.c_void_p), 3,
                            C.byref(n), C.byref(nb)) == 0
    assert concat.tobytes() == b"alphabetagamma" and off.tolist() == [0, 5, 9, 14]
    # another format version is refused as unsupported, not as malformed
    v2 = blob.copy()
    v2[0] = 2
    assert L.msi_fst_decode(v2.ctypes.data_as(C.c_void_p), v2.size, 0, None, 0, None, 0, C.byref(n), C.byref(nb)) == -5


def test_malformed_input_is_refused_not_trusted():
    rng = random.Random(5)
    keys = sorted({bytes(rng.choices(b"abcdefgh\xc3\xa9", k=rng.randint(1, 9))) for _ in range(3000)} |
                  {b"q" + bytes([i]) for i in range(50)})
    blob = F.fst_build(keys)
    assert product_keys(blob) == keys
    # the checksum catches every flip
    for _ in range(200):
        bad = bytearray(blob)
        bad[rng.randrange(len(bad) - 4)] ^= 1 << rng.randrange(8)
        with pytest.raises(Exception):
            product_keys(bytes(bad))
    # without it the structural checks must hold: an error, or keys that are still strictly ascending and as many as
    # the footer says (a flipped input byte can yield another valid set); the oracle decoder agrees on which
    refused = 0
    for _ in range(3000):
        bad = bytearray(blob)
        for _ in range(rng.randint(1, 3)):
            bad[rng.randrange(len(bad))] ^= 1 << rng.randrange(8)
        bad = bytes(bad)
        try:
            want = F.fst_keys(bad, verify_checksum=False)
        except (F.FstError, IndexError):
            want = None
        try:
            got = product_keys(bad, T.FST_SKIP_CHECKSUM)
        except Exception:
            got = None
            refused += 1
        if got is not None:
            assert all(a < b for a, b in zip(got, got[1:]))
        assert (got is None) == (want is None), bad.hex()
        if got is not None:
            assert got == want
    assert refused > 100
    for cut in list(range(0, 40)) + [len(blob) - 1, len(blob) - 4, len(blob) - 20, len(blob) // 2]:
        with pytest.raises(Exception):
            product_keys(blob[:cut])


def test_posting_bytes_written_by_milli():
    """The same index's posting lists: every CboRoaringBitmap value there is the raw form (<= 7 docids, native-endian
    u32s, cbo_roaring_bitmap_codec.rs:33-51) and main["documents-ids"] is a RoaringBitmap of `roaring` 0.10 — the
    serialiser the synthetic index and the decode tests use must reproduce both byte for byte."""
    n_cbo = 0
    for b in FIX["bitmaps"]:
        raw = bytes.fromhex(b["hex"])
        if b["codec"] == "cbo":
            assert len(raw) % 4 == 0 and len(raw) <= 28
            ids = np.frombuffer(raw, dtype="<u4")
            assert ids.tolist() == sorted(set(ids.tolist())) and ids.size and int(ids.max()) < 2
            assert synth.cbo_serialize(ids) == raw
            n_cbo += 1
        else:
            assert synth.roaring_serialize(np.arange(b["n_documents"], dtype=np.uint32)) == raw
    assert n_cbo >= 70
