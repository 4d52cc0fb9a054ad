"""Extracts the reference's filter tests (crates/milli/tests/search/filters.rs: `test_filter!` cases over
tests/assets/test_set.ndjson) into tests/golden/filter_fixtures.json: the documents, every case's filter groups
(Left = OR of its members, Right = one filter; the groups are ANDed) and the external ids the reference's own helper
expects — `execute_filter` / `expected_filtered_ids` of crates/milli/tests/search/mod.rs:226-380, restated below.
Run in the build container (reads /root/reference); the tests only read the JSON."""
import json
import os
import re

REF = "/root/reference/crates/milli/tests"
OUT = os.path.join(os.path.dirname(__file__), "filter_fixtures.json")


def load_docs():
    text = open(f"{REF}/assets/test_set.ndjson").read()
    dec, i, docs = json.JSONDecoder(), 0, []
    while i < len(text):
        while i < len(text) and text[i].isspace():
            i += 1
        if i >= len(text):
            break
        d, i = dec.raw_decode(text, i)
        docs.append(d)
    return docs


def normalize_facet(s):
    import unicodedata
    return unicodedata.normalize("NFKD", s.strip()).lower()


def is_empty_value(v):
    return v in ("", [], {})


def execute_filter(f, d):
    """mod.rs:226-310 — None = the helper does not cover this filter (nested opt1.opt2 cases are not extracted)."""
    if "opt1.opt2" in f:
        return None
    has1 = "opt1" in d
    if "!=" in f:
        field, v = f.split("!=", 1)
        return (field == "tag" and d["tag"] != v) or (field == "asc_desc_rank" and not (v.isdigit() and d["asc_desc_rank"] == int(v)))
    if "STARTS WITH" in f:
        field, prefix = f.split("STARTS WITH", 1)
        return normalize_facet(d[field.strip()]).startswith(normalize_facet(prefix.strip().strip("'")))
    if f.startswith("_geoRadius"):
        return d["geo_rank"] < 100000
    if f.startswith("NOT _geoRadius"):
        return d["geo_rank"] > 1000000
    if f in ("opt1 EXISTS", "NOT opt1 NOT EXISTS"):
        return has1
    if f in ("NOT opt1 EXISTS", "opt1 NOT EXISTS"):
        return not has1
    if f in ("opt1 IS NULL", "NOT opt1 IS NOT NULL"):
        return has1 and d["opt1"] is None
    if f in ("NOT opt1 IS NULL", "opt1 IS NOT NULL"):
        return not (has1 and d["opt1"] is None)
    if f in ("opt1 IS EMPTY", "NOT opt1 IS NOT EMPTY"):
        return has1 and is_empty_value(d["opt1"])
    if f in ("NOT opt1 IS EMPTY", "opt1 IS NOT EMPTY"):
        return not (has1 and is_empty_value(d["opt1"]))
    if f in ("tag_in IN[1, 2, 3, four, five]", "NOT tag_in NOT IN[1, 2, 3, four, five]"):
        return d["id"] in "ABCDE"
    if f == "tag_in NOT IN[1, 2, 3, four, five]":
        return d["id"] not in "ABCDE"
    if "=" in f:
        field, v = f.split("=", 1)
        return (field == "tag" and d["tag"] == v) or (field == "asc_desc_rank" and d["asc_desc_rank"] == int(v))
    if "<" in f:
        return d["asc_desc_rank"] < int(f.split("<", 1)[1])
    if ">" in f:
        return d["asc_desc_rank"] > int(f.split(">", 1)[1])
    raise ValueError(f)


def main():
    docs = load_docs()
    src = open(f"{REF}/search/filters.rs").read()
    cases = []
    for m in re.finditer(r"test_filter!\(\s*(\w+),\s*vec!\[(.*?)\]\s*\);", src, re.S):
        name, body = m.group(1), m.group(2)
        groups = []
        for g in re.finditer(r'Left\(vec!\[(.*?)\]\)|Right\("((?:[^"\\]|\\.)*)"\)', body, re.S):
            if g.group(2) is not None:
                groups.append([g.group(2)])
            else:
                groups.append(re.findall(r'"((?:[^"\\]|\\.)*)"', g.group(1)))
        ids, ok = {d["id"] for d in docs}, True
        for grp in groups:
            sel = set()
            for f in grp:
                r = [execute_filter(f, d) for d in docs]
                if any(x is None for x in r):
                    ok = False
                    break
                sel |= {d["id"] for d, x in zip(docs, r) if x}
            ids &= sel
        if ok:
            cases.append({"name": name, "groups": groups, "ids": sorted(ids)})
    json.dump({"docs": docs, "cases": cases, "src": "crates/milli/tests/search/filters.rs"}, open(OUT, "w"), indent=0,
              sort_keys=True, ensure_ascii=False)
    print(len(docs), "documents,", len(cases), "cases ->", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
