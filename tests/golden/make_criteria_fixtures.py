"""Extracts the reference's criteria tests (crates/milli/tests/search/query_criteria.rs: the `test_criterion!` cases and
`criteria_mixup`'s 120 orders over tests/assets/test_set.ndjson, query "hello world america", synonyms of
tests/search/mod.rs:40-44) into tests/golden/criteria_fixtures.json with the external ids the reference's own helper
expects — `expected_order` of crates/milli/tests/search/mod.rs:150-224 (stable sort + group by the dataset's rank
columns), restated below.  The documents are those of tests/golden/filter_fixtures.json (same dataset).
Run in the build container (reads /root/reference); the tests only read the JSON."""
import itertools
import json
import os
import re

REF = "/root/reference/crates/milli/tests"
OUT = os.path.join(os.path.dirname(__file__), "criteria_fixtures.json")
NAMES = {"Words": "words", "Typo": "typo", "Proximity": "proximity", "Attribute": "attribute", "Exactness": "exactness",
         "Sort": "sort"}
COLUMN = {"words": "word_rank", "typo": "typo_rank", "proximity": "proximity_rank", "attribute": "attribute_rank",
          "exactness": "exact_rank"}


def expected_order(docs, criteria, tms, sort):
    groups = [list(docs)]
    for c in criteria:
        new = []
        for g in groups:
            key, rev = None, False
            if c in COLUMN:
                key = COLUMN[c]
            elif c == "sort" and sort in ([["tag", "asc"]], [["tag", "desc"]]):
                key, rev = "sort_by_rank", sort[0][1] == "desc"
            elif c in ("asc:asc_desc_rank", "desc:asc_desc_rank"):
                key, rev = "asc_desc_rank", c.startswith("desc")
            if key is None:
                new.append(list(g))
                continue
            g = sorted(g, key=lambda d: d[key], reverse=rev)          # sort_by_key is stable, Reverse keeps equal keys in order
            for _, grp in itertools.groupby(g, key=lambda d: d[key]):
                new.append(list(grp))
        groups = new
    flat = [d for g in groups for d in g]
    return [d["id"] for d in flat if tms == "last" or d["word_rank"] == 0]


def criterion(tok):
    tok = tok.strip()
    m = re.fullmatch(r'(Asc|Desc)\(S\("([^"]+)"\)\)', tok)
    if m:
        return f"{m.group(1).lower()}:{m.group(2)}"
    return NAMES[tok]


def main():
    from make_filter_fixtures import load_docs
    docs = load_docs()
    src = open(f"{REF}/search/query_criteria.rs").read()
    cases = []
    for m in re.finditer(r"test_criterion!\(\s*(\w+),\s*(\w+),\s*vec!\[(.*?)\],\s*vec!\[(.*?)\]\s*\);", src, re.S):
        name, opt, crit, sort = m.groups()
        criteria = [criterion(t) for t in re.findall(r'(?:Asc|Desc)\(S\("[^"]+"\)\)|\w+', crit)] if crit.strip() else []
        s = [[f, d.lower()] for d, f in re.findall(r'AscDesc::(Asc|Desc)\(Member::Field\(S\("([^"]+)"\)\)\)', sort)]
        tms = "last" if opt == "ALLOW_OPTIONAL_WORDS" else "all"
        cases.append({"name": name, "criteria": criteria, "tms": tms, "sort": s, "ids": expected_order(docs, criteria, tms, s)})
    for perm in itertools.permutations(["attribute", "desc:asc_desc_rank", "exactness", "proximity", "typo"]):
        criteria = ["words"] + list(perm)
        cases.append({"name": "criteria_mixup", "criteria": criteria, "tms": "last", "sort": [],
                      "ids": expected_order(docs, criteria, "last", [])})
    # crates/milli/tests/search/distinct.rs: `test_distinct!(name, field, exhaustive, limit, offset, criteria, n_candidates)` —
    # the candidates count is a literal of the test; the ids are expected_order filtered to the first document of every
    # distinct value, then offset / limit (distinct.rs:60-74)
    dsrc = re.sub(r"//[^\n]*", "", open(f"{REF}/search/distinct.rs").read())
    dcases = []
    for m in re.finditer(r"test_distinct!\(\s*(\w+),\s*(\w+),\s*(true|false),\s*([\w.()]+),\s*(\d+),\s*vec!\[(.*?)\],\s*(\d+)\s*\);",
                         dsrc, re.S):
        name, field, exh, limit, offset, crit, n_res = m.groups()
        criteria = [criterion(t) for t in re.findall(r'(?:Asc|Desc)\(S\("[^"]+"\)\)|\w+', crit)] if crit.strip() else []
        limit = 17 if "EXTERNAL" in limit else int(limit)
        seen, ids = set(), []
        for did in expected_order(docs, criteria, "last", []):
            d = next(x for x in docs if x["id"] == did)
            if d[field] not in seen:
                seen.add(d[field])
                ids.append(did)
        dcases.append({"name": name, "distinct": field, "exhaustive": exh == "true", "limit": limit, "offset": int(offset),
                       "criteria": criteria, "candidates": int(n_res), "ids": ids[int(offset):][:limit]})
    assert len(dcases) == 19, len(dcases)
    out = {"distinct_cases": dcases, "query": "hello world america", "searchable": ["title", "description"],
           "synonyms": {"hello": ["good morning"], "world": ["earth"], "america": ["the united states"]},
           "cases": cases, "src": "crates/milli/tests/search/query_criteria.rs"}
    json.dump(out, open(OUT, "w"), indent=0, sort_keys=True)
    print(len(cases), "criteria cases,", len(dcases), "distinct cases ->", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    import sys
    sys.path.insert(0, os.path.dirname(__file__))
    main()
