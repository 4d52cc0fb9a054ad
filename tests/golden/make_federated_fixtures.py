"""Extracts the merged hit orders of the reference's federated-search tests (crates/meilisearch/tests/search/multi/mod.rs:
every inline snapshot of a `"federation": {...}` request whose hits carry `_federation.queriesPosition` and
`_federation.weightedRankingScore`) into tests/golden/federated_fixtures.json: per case the sequence of
(queriesPosition, weightedRankingScore) in the order the reference returned them.  Cases whose order also depends on values
the snapshot does not hold (Sort / GeoSort members, vector similarities shown only as scores) are kept only when every
query is a plain keyword query — there the weighted global score IS the whole WeightedScoreValue sequence
(score_details.rs:177-199), so the interleaving of the per-query lists pins weighted_scores::compare and the tie rule
`left.query_index < right.query_index` (perform.rs:566,609).
Run in the build container (reads /root/reference); the tests only read the JSON."""
import json
import os
import re

SRC = "/root/reference/crates/meilisearch/tests/search/multi/mod.rs"
OUT = os.path.join(os.path.dirname(__file__), "federated_fixtures.json")


def main():
    src = open(SRC).read()
    cases = []
    for fn in re.finditer(r"async fn (\w+)\(\)(.*?)(?=\n#\[actix_rt::test\]|\Z)", src, re.S):
        name, text = fn.groups()
        if not name.startswith("federation") or any(w in name for w in ("sort", "vector", "facets", "distinct", "formatting", "error")):
            continue
        for snap in re.finditer(r'multi_search\(json!\((\{.*?\})\)\)\s*\.await;.*?snapshot!\(json_string!\(response.*?@r###"(.*?)"###\);', text, re.S):
            req, body = snap.groups()
            if '"sort"' in req or '"vector"' in req or '"hybrid"' in req:
                continue
            try:
                resp = json.loads(body)
            except json.JSONDecodeError:
                continue
            hits = resp.get("hits") or []
            seq = []
            for h in hits:
                f = h.get("_federation") or {}
                if "queriesPosition" not in f or "weightedRankingScore" not in f:
                    seq = None
                    break
                seq.append([int(f["queriesPosition"]), float(f["weightedRankingScore"])])
            if not seq or len({q for q, _ in seq}) < 2:
                continue
            cases.append({"src": f"multi/mod.rs::{name}", "offset": int(resp.get("offset", 0)), "hits": seq})
    assert len(cases) >= 3, len(cases)
    json.dump({"source": SRC.replace("/root/reference/", ""), "cases": cases}, open(OUT, "w"), indent=1)
    print(len(cases), "cases;", sum(len(c["hits"]) for c in cases), "hits;",
          sum(1 for c in cases for a, b in zip(c["hits"], c["hits"][1:]) if a[1] == b[1] and a[0] != b[0]), "ties between queries")


if __name__ == "__main__":
    main()
