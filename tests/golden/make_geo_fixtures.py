"""Extracts the searches of crates/milli/src/search/new/tests/geo_sort.rs (documents, sort criteria, queries, the
inline docid snapshots and the score-detail snapshots, inline or under snapshots/) into tests/golden/geo_snapshots.json.
Run in the build container (reads /root/reference); the tests only read the JSON."""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(__file__))
from make_ranking_fixtures import balanced, functions, strip_comments   # noqa: E402

REF = "/root/reference/crates/milli/src/search/new/tests"
OUT = os.path.join(os.path.dirname(__file__), "geo_snapshots.json")


def documents(body):
    m = re.search(r"documents!\(\[", body)
    end = balanced(body, m.end() - 1, "[", "]")
    text = body[m.end() - 1:end].replace("RESERVED_GEO_FIELD_NAME", '"_geo"')
    text = re.sub(r",\s*([\]}])", r"\1", text)
    return json.loads(text)


def main():
    src = strip_comments(open(f"{REF}/geo_sort.rs").read())
    fns = functions(src)
    cases, indexes = [], {}
    for name, body in fns.items():
        if "index.search(" not in body or name.startswith(("create_", "execute_")):
            continue
        docs = documents(body)
        key = f"geo_sort::{name}"
        indexes[key] = {"docs": docs, "criteria": ["words", "sort"], "primary_key": "id"}
        asserts = [(m.start(), m.end()) for m in re.finditer(r"insta::assert_snapshot!\(", body)]
        events = []
        for m in re.finditer(r"s\.sort_criteria\(vec!\[", body):
            end = balanced(body, m.end() - 1, "[", "]")
            events.append((m.start(), "sort", body[m.end():end - 1]))
        for m in re.finditer(r's\.query\("([^"]*)"\)', body):
            events.append((m.start(), "query", m.group(1)))
        for m in re.finditer(r"s\.geo_max_bucket_size\((\d+)\)", body):
            events.append((m.start(), "cap", int(m.group(1))))
        for m in re.finditer(r"execute_iterative_and_rtree_returns_the_same\(", body):
            events.append((m.start(), "run", None))
        events.sort()
        sort, query, cap = None, "", None
        for k, (pos, kind, val) in enumerate(events):
            if kind == "sort":
                sort = []
                for d, member in re.findall(r"AscDesc::(Asc|Desc)\((Member::(?:Geo\(\[[^\]]*\]\)|Field\([^)]*\)\)))", val):
                    g = re.search(r"Geo\(\[\s*(-?[\d.]+),\s*(-?[\d.]+)\s*\]", member)
                    if g:
                        sort.append([["_geoPoint", float(g.group(1)), float(g.group(2))], d.lower()])
                    else:
                        sort.append([re.search(r'"([^"]+)"', member).group(1), d.lower()])
            elif kind == "query":
                query = val
            elif kind == "cap":
                cap = val
            elif kind == "run":
                nxt = next((p for p, kd, _ in events[k + 1:] if kd == "run"), len(body))
                case = {"src": f"crates/milli/src/search/new/tests/geo_sort.rs::{name}", "index": key, "query": query,
                        "sort": sort, "detailed": True, "ids": None, "scores": None}
                for n, (a, b) in enumerate(asserts, 1):
                    if not (pos <= a < nxt):
                        continue
                    e = balanced(body, b - 1, "(", ")")
                    text = body[b:e]
                    if "{ids:?}" in text:
                        case["ids"] = json.loads(re.search(r'@"(\[[^"]*\])"', text).group(1))
                    elif "{scores:#?}" in text:
                        if "@" in text:
                            snap = re.search(r'@r#*"(.*?)"#*\s*\)$', text, re.S).group(1)
                        else:
                            short = name[5:] if name.startswith("test_") else name
                            fn = f"{REF}/snapshots/milli__search__new__tests__geo_sort__{short}" + (f"-{n}" if n > 1 else "") + ".snap"
                            snap = open(fn).read().split("---", 2)[2]
                        case["scores"] = re.sub(r"\s+", "", snap)
                assert case["ids"] is not None and case["scores"] is not None, (name, pos)
                cases.append(case)
    json.dump({"indexes": indexes, "cases": cases}, open(OUT, "w"), indent=0, sort_keys=True)
    print(len(indexes), "indexes,", len(cases), "cases ->", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
