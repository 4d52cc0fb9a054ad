"""Pins oracle/ranking_oracle.py (the CPU restatement of milli's keyword ranking) against the reference's
own golden vectors: every search of search/new/tests/{proximity,attribute_fid,word_position,exactness,
words_tms,typo_proximity,proximity_typo,ngram_split_words,typo}.rs 
(tests/golden/ranking_snapshots.json, extracted by tests/golden/make_ranking_fixtures.py): expected docid
order and, where the reference snapshots them, the score details of every hit."""
import json
import os

import pytest

from oracle import oracle as O
from oracle import ranking_oracle as R
from tests.toy_milli import ToyMilli

FIX = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ranking_snapshots.json")))

# Cases that need a feature the toy index does not model (reason -> skipped, not failed).
UNSUPPORTED = {}


def make_ctx(index):
    dic = O.Dictionary(index.words)

    def lookup(word, max_typos, is_prefix):
        one, two = O.typo_lookup(dic, word, max_typos, is_prefix)
        return [index.words[i] for i in one], [index.words[i] for i in two]
    return R.Ctx(index, lookup)


def build_index(cfg):
    return ToyMilli(cfg["docs"], searchable=cfg.get("searchable"), exact_attributes=cfg.get("exact_attributes", ()),
                    exact_words=cfg.get("exact_words", ()), criteria=cfg.get("criteria"),
                    min_one=cfg.get("min_one", 5), min_two=cfg.get("min_two", 9),
                    authorize_typos=cfg.get("authorize_typos", True), synonyms=cfg.get("synonyms"),
                    stop_words=cfg.get("stop_words", ()), distinct=cfg.get("distinct"))


def debug_score(s):
    """Rust `{:#?}` of ScoreDetails with the whitespace removed."""
    k = s[0]
    if k == "Words":
        return f"Words(Words{{matching_words:{s[1]},max_matching_words:{s[2]},}},)"
    if k == "Typo":
        return f"Typo(Typo{{typo_count:{s[1]},max_typo_count:{s[2]},}},)"
    if k == "ExactWords":
        return f"ExactWords(ExactWords{{matching_words:{s[1]},max_matching_words:{s[2]},}},)"
    if k == "ExactAttribute":
        return f"ExactAttribute({s[1]},)"
    if k == "Skipped":
        return "Skipped"
    if k == "Sort":
        v = s[3]
        val = "Null" if v[0] == "Null" else (f"Number({float(v[1])!r})" if v[0] == "Number" else f'String("{v[1]}")')
        return (f'Sort(Sort{{field_name:"{s[1]}",ascending:{str(s[2]).lower()},redacted:false,value:{val},}},)')
    return f"{k}(Rank{{rank:{s[1]},max_rank:{s[2]},}},)"


def debug_scores(scores):
    return "[" + "".join("[" + "".join(debug_score(s) + "," for s in hit) + "]," for hit in scores) + "]"


def debug_ids_scores(ids, scores):
    return "[" + "".join(f"({i},[" + "".join(debug_score(s) + "," for s in hit) + "],)," for i, hit in zip(ids, scores)) + "]"


CASES = [c for c in FIX["cases"]]


@pytest.mark.parametrize("case", CASES, ids=[f'{c["src"].split("::")[1]}:{c["query"]}' for c in CASES])
def test_reference_snapshot(case):
    cfg = FIX["indexes"][case["index"]]
    if cfg.get("unsupported") or case["query"] in UNSUPPORTED:
        pytest.skip("needs " + str(cfg.get("unsupported") or UNSUPPORTED[case["query"]]))
    if case.get("needs"):
        pytest.skip("needs the " + case["needs"] + " ranking rule (not a keyword rule; facet databases)")
    index = build_index(cfg)
    ids, scores, _ = R.search(make_ctx(index), case["query"], tms=case["tms"], offset=case["offset"],
                              length=case["limit"], detailed=case["detailed"], stop_after=case.get("stop_after"),
                              distinct=case.get("distinct") or index.distinct_field, sort=case.get("sort"))
    if case["ids"] is not None:
        assert ids == case["ids"]
    if case.get("scores"):
        assert debug_scores(scores) == case["scores"]
    if case.get("global_scores"):
        assert [f"{R.global_score(sc):.4f}" for sc in scores] == case["global_scores"]
    if case.get("ids_scores"):
        assert debug_ids_scores(ids, scores) == case["ids_scores"]


@pytest.mark.parametrize("order", ["desc", 1, 2, 3])
def test_fid_and_position_edge_order_is_unobservable(order, monkeypatch):
    """The reference iterates a hash set / map when it pushes the Fid and Position edges of a node (fid/mod.rs:60-100,
    position/mod.rs:60-110) — an order nobody specified.  Every reference search with criteria that reach those rules,
    replayed with the edges reversed and shuffled: same docids, same score details.  (The product pushes them ascending.)"""
    monkeypatch.setattr(R, "EDGE_ORDER", order)
    n = 0
    for case in CASES:
        cfg = FIX["indexes"][case["index"]]
        if cfg.get("unsupported") or case.get("needs"):
            continue
        index = build_index(cfg)
        ids, scores, _ = R.search(make_ctx(index), case["query"], tms=case["tms"], offset=case["offset"],
                                  length=case["limit"], detailed=case["detailed"], stop_after=case.get("stop_after"),
                                  distinct=case.get("distinct") or index.distinct_field, sort=case.get("sort"))
        if case["ids"] is not None:
            assert ids == case["ids"], case["query"]
        if case.get("scores"):
            assert debug_scores(scores) == case["scores"], case["query"]
        if case.get("ids_scores"):
            assert debug_ids_scores(ids, scores) == case["ids_scores"], case["query"]
        n += 1
    assert n >= 100
