"""Extracts the reference's matching-strategy tests (crates/meilisearch/tests/search/matching_strategy.rs: the seven
documents of SIMPLE_SEARCH_DOCUMENTS and every `index.search(json!({"q": ..., "matchingStrategy": ...}))` call with the
hit ids its inline snapshot holds) into tests/golden/matching_strategy_fixtures.json.  These are the only literals in
the reference that pin TermsMatchingStrategy::Frequency (query_graph.rs:303-344).
Run in the build container (reads /root/reference); the tests only read the JSON."""
import json
import os
import re

SRC = "/root/reference/crates/meilisearch/tests/search/matching_strategy.rs"
OUT = os.path.join(os.path.dirname(__file__), "matching_strategy_fixtures.json")


def main():
    src = open(SRC).read()
    m = re.search(r"static SIMPLE_SEARCH_DOCUMENTS.*?json!\(\[(.*?)\]\)\s*\}\);", src, re.S)
    body = re.sub(r",(\s*[}\]])", r"\1", "[" + m.group(1) + "]")       # trailing commas of the json! macro
    docs = json.loads(body)
    cases = []
    for fn in re.finditer(r"async fn (\w+)\(\)(.*?)(?=\n#\[actix_rt::test\]|\Z)", src, re.S):
        name, text = fn.groups()
        for s in re.finditer(r'json!\(\{"q":\s*"([^"]*)",\s*"matchingStrategy":\s*"(\w+)".*?\}\).*?'
                             r'snapshot!\(response\["hits"\],\s*@(?:r###")?(.*?)(?:"###)?\);', text, re.S):
            q, strategy, hits = s.groups()
            hits = hits.strip().strip('"')
            ids = [h["id"] for h in json.loads(hits)]
            cases.append({"src": f"matching_strategy.rs::{name}", "query": q, "strategy": strategy, "ids": ids})
    assert sum(1 for c in cases if c["strategy"] == "frequency") >= 3, cases
    json.dump({"source": SRC.replace("/root/reference/", ""), "documents": docs, "cases": cases}, open(OUT, "w"), indent=1)
    print(len(cases), "cases,", sum(1 for c in cases if c["strategy"] == "frequency"), "with the frequency strategy")


if __name__ == "__main__":
    main()
