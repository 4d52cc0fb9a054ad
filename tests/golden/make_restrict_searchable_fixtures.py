"""Extracts the flat-document tests of the reference's `attributesToSearchOn` suite
(crates/meilisearch/tests/search/restrict_searchable.rs) into tests/golden/restrict_searchable_fixtures.json: per test the
documents, the settings changes and searches IN ORDER, and what each search must return — the number of hits, or the hits'
`id` / `title` values in order.  The nested-field tests (`details.*`, `*.title`: field flattening at indexing time) are out of
scope and skipped.  Run where /root/reference exists; the JSON is committed.

    python tests/golden/make_restrict_searchable_fixtures.py
"""
import json
import os
import re

SRC = "/root/reference/crates/meilisearch/tests/search/restrict_searchable.rs"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "restrict_searchable_fixtures.json")
FLAT = ["simple_search_on_title", "search_no_searchable_attribute_set", "search_on_all_attributes",
        "search_on_all_attributes_restricted_set", "simple_prefix_search_on_title",
        "simple_search_on_title_matching_strategy_all", "simple_search_on_no_field", "word_ranking_rule_order",
        "word_ranking_rule_order_exact_words", "typo_ranking_rule_order", "attributes_ranking_rule_order",
        "exactness_ranking_rule_order", "search_on_exact_field", "phrase_search_on_title"]


def rust_json(text):
    """The body of a json!(...) invocation as Python data (JSON with trailing commas)."""
    return json.loads(re.sub(r",(\s*[}\]])", r"\1", text))


def balanced(text, start, open_ch, close_ch):
    """text[start] == open_ch -> index just after its matching close (string literals skipped)."""
    depth, i, in_str = 0, start, False
    while i < len(text):
        c = text[i]
        if in_str:
            if c == "\\":
                i += 1
            elif c == '"':
                in_str = False
        elif c == '"':
            in_str = True
        elif c == open_ch:
            depth += 1
        elif c == close_ch:
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1
    raise ValueError("unbalanced")


def main():
    src = open(SRC).read()
    m = re.search(r"static SIMPLE_SEARCH_DOCUMENTS[^=]*=\s*Lazy::new\(\|\| \{\s*json!\(", src)
    simple_docs = rust_json(src[m.end():balanced(src, m.end() - 1, "(", ")") - 1])
    cases = []
    for name in FLAT:
        m = re.search(r"async fn %s\(\) \{" % name, src)
        body = src[m.end():balanced(src, m.end() - 1, "{", "}")]
        docs = None
        if "&SIMPLE_SEARCH_DOCUMENTS" in body:
            docs = simple_docs
        events = []
        for jm in re.finditer(r"json!\(", body):
            end = balanced(body, jm.end() - 1, "(", ")")
            value = rust_json(body[jm.end():end - 1])
            before = body[max(0, jm.start() - 120):jm.start()]
            if isinstance(value, list) and value and isinstance(value[0], dict) and docs is None:
                docs = value
            elif re.search(r"update_settings_searchable_attributes\(\s*$", before):
                events.append({"settings": {"searchableAttributes": value}})
            elif re.search(r"update_settings_typo_tolerance\(\s*$", before):
                events.append({"settings": {"typoTolerance": value}})
            elif re.search(r"\.search\(\s*$", before):
                tail = body[end:end + 2500]
                tail = tail[:tail.index(".await")]
                want = {}
                lm = re.search(r'as_array\(\)\.unwrap\(\)\.len\(\), @"(\d+)"', tail)
                if lm:
                    want["n_hits"] = int(lm.group(1))
                hm = re.search(r'json_string!\(response\["hits"\]\),\s*@r###"(.*?)"###', tail, re.S)
                if hm:
                    want["hits"] = json.loads(hm.group(1))
                assert want, (name, tail[:200])
                events.append({"search": value, "want": want})
        assert docs is not None and any("search" in e for e in events), name
        cases.append({"src": f"restrict_searchable.rs::{name}", "documents": docs, "events": events})
    json.dump({"cases": cases}, open(OUT, "w"), indent=1, sort_keys=True)
    print(len(cases), "tests,", sum(1 for c in cases for e in c["events"] if "search" in e), "searches ->", OUT)


if __name__ == "__main__":
    main()
