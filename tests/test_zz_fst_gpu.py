"""msi_dict_create_from_fst / msi_dict_create_values_from_fst: a dictionary staged from `fst::Set`
bytes answers exactly like the one staged from the flat word list, and like the oracle."""
import json
import os

import numpy as np
import pytest

import meilisearch_amd as ma
from meilisearch_amd import synth
from oracle import fst_oracle as F, oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_dictionary_from_fst_bytes_answers_like_the_word_list():
    ctx = ma.Context(0)
    words = synth.make_dictionary(20000, seed=21)
    concat, off = synth.flatten_words(words)
    flat = ma.GpuDictionary(ctx, concat=concat, offsets=off)
    raw = concat.tobytes()
    blob = F.fst_build([raw[off[i]:off[i + 1]] for i in range(len(off) - 1)])
    staged = ma.GpuDictionary.from_fst(ctx, blob)
    assert len(staged) == len(flat) == len(off) - 1
    queries = synth.make_typo_queries(words, 300, seed=22)
    odic = O.Dictionary.from_flat(concat, off)
    for (w, b, p), (a1, a2), (b1, b2) in zip(queries, flat.lookup(queries), staged.lookup(queries)):
        e1, e2 = O.typo_lookup(odic, w, b, p)
        assert a1.tolist() == b1.tolist() == e1.tolist() and a2.tolist() == b2.tolist() == e2.tolist()
    with pytest.raises(ma.MsiError):
        ma.GpuDictionary.from_fst(ctx, blob[:-1])


def test_the_index_milli_wrote():
    """main["words-fst"] and the facet FST of the reference's own index, staged from their bytes."""
    fix = json.load(open(os.path.join(ROOT, "tests", "golden", "index_blobs.json")))
    by_name = {e["name"]: bytes.fromhex(e["hex"]) for e in fix["fst"]}
    ctx = ma.Context(0)
    d = ma.GpuDictionary.from_fst(ctx, by_name["main[words-fst]"])
    words = F.fst_keys(by_name["main[words-fst]"])
    assert [d.word(i).encode() for i in range(len(d))] == words
    (one, two), = d.lookup([("kefr", 1, False)])
    assert [words[i] for i in one] == [b"kef", b"kefir"] and two.size == 0
    (one, two), = d.lookup([("migon", 2, False)])
    assert [words[i] for i in one] == [b"mignon"]
    facet = ma.GpuDictionary.from_fst(ctx, by_name["facet-id-string-fst[0002]"], facet_values=True)
    values = F.fst_keys(by_name["facet-id-string-fst[0002]"])
    idx, truncated = facet.search_values("kef", 0)
    assert [values[i] for i in idx] == [b"kef", b"kefirounet", b"kefkef"] and not truncated
