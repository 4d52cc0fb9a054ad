#!/usr/bin/env python
"""Extracts the golden vectors of the reference's ranking-rule tests into tests/golden/ranking_snapshots.json:
for every `index.search(...)` of crates/milli/src/search/new/tests/{proximity,attribute_fid,word_position,
exactness,words_tms,typo_proximity,proximity_typo,ngram_split_words,typo}.rs the documents and settings of
the index, the query, and the expected docids (inline insta snapshots) and score details (snapshots/*.snap).
Only test DATA is extracted.  Runs where /root/reference exists (this container); the JSON travels.

    python tests/golden/make_ranking_fixtures.py
"""
import json
import os
import re

REF = "/root/reference/crates/milli/src/search/new/tests"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ranking_snapshots.json")
FILES = ["proximity", "attribute_fid", "word_position", "exactness", "words_tms", "typo_proximity",
         "proximity_typo", "ngram_split_words", "typo", "stop_words", "distinct", "sort"]
CRIT = {"Words": "words", "Typo": "typo", "Proximity": "proximity", "Attribute": "attribute",
        "AttributeRank": "attributeRank", "WordPosition": "wordPosition", "Exactness": "exactness", "Sort": "sort"}


RAW = re.compile(r'r(#*)"')


def literal_end(s, i):
    """If a string / raw string literal starts at s[i], the index one past its end; else None."""
    if s[i] == '"':
        j = i + 1
        while s[j] != '"':
            j += 2 if s[j] == "\\" else 1
        return j + 1
    if s[i] == "r" and (i == 0 or not (s[i - 1].isalnum() or s[i - 1] == "_")):
        m = RAW.match(s, i)
        if m:
            close = '"' + m.group(1)
            return s.index(close, m.end()) + len(close)
    return None


def balanced(s, start, open_c, close_c):
    """index one past the bracket that closes s[start] (s[start] == open_c), skipping string literals."""
    depth, i = 0, start
    while i < len(s):
        e = literal_end(s, i)
        if e is not None:
            i = e
            continue
        c = s[i]
        if c == open_c:
            depth += 1
        elif c == close_c:
            depth -= 1
            if depth == 0:
                return i + 1
        i += 1
    raise ValueError("unbalanced")


def strip_comments(s):
    out, i = [], 0
    while i < len(s):
        e = literal_end(s, i)
        if e is not None:
            out.append(s[i:e])
            i = e
        elif s.startswith("//", i):
            while i < len(s) and s[i] != "\n":
                i += 1
        elif s.startswith("/*", i):
            i = s.index("*/", i) + 2
        else:
            out.append(s[i])
            i += 1
    return "".join(out)


def parse_documents(body):
    docs = []
    for m in re.finditer(r"documents!\(", body):
        a = body.index("[", m.end() - 1)
        b = balanced(body, a, "[", "]")
        txt = strip_comments(body[a:b])
        txt = re.sub(r",(\s*[\]}])", r"\1", txt)
        docs += json.loads(txt, strict=False)
    return docs


def parse_settings(body, cfg):
    m = re.search(r"set_searchable_fields\(\s*(?:vec!)?\[(.*?)\]", body, re.S)
    if m:
        cfg["searchable"] = re.findall(r'"([^"]+)"', m.group(1))
    m = re.search(r"set_criteria\(vec!\[(.*?)\]\)", body, re.S)
    if m:
        cfg["criteria"] = [CRIT[c] if c in CRIT else f"{c.lower()}:{f}"
                           for c, f in re.findall(r'Criterion::(\w+)(?:\(S\("([^"]+)"\)\))?', m.group(1))]
    m = re.search(r"set_exact_attributes\(\[(.*?)\]", body, re.S)
    if m:
        cfg["exact_attributes"] = re.findall(r'"([^"]+)"', m.group(1))
    m = re.search(r"set_exact_words\(\s*\[(.*?)\]", body, re.S)
    if m:
        cfg["exact_words"] = re.findall(r'"([^"]+)"', m.group(1))
    m = re.search(r"set_authorize_typos\((\w+)\)", body)
    if m:
        cfg["authorize_typos"] = m.group(1) == "true"
    m = re.search(r"set_min_word_len_one_typo\((\d+)\)", body)
    if m:
        cfg["min_one"] = int(m.group(1))
    m = re.search(r"set_min_word_len_two_typos\((\d+)\)", body)
    if m:
        cfg["min_two"] = int(m.group(1))
    m = re.search(r"set_stop_words\(BTreeSet::from_iter\(\[(.*?)\]\)\)", body, re.S)
    if m:
        cfg["stop_words"] = re.findall(r'"([^"]+)"', m.group(1))
    m = re.search(r'set_distinct_field\("([^"]+)"', body)
    if m:
        cfg["distinct"] = m.group(1)
    if "reset_distinct_field()" in body:
        cfg.pop("distinct", None)
    syn = {}
    for m in re.finditer(r'\w+\.insert\("([^"]+)"\.to_owned\(\),\s*vec!\[(.*?)\]\)', body, re.S):
        syn[m.group(1)] = re.findall(r'"([^"]+)"', m.group(2))
    if syn and "set_synonyms" in body:
        cfg["synonyms"] = syn
    for feat in ("set_displayed_fields", "set_dictionary", "set_separator_tokens", "set_proximity_precision",
                 "set_searchable_fields(vec![])"):
        if feat in body:
            cfg.setdefault("unsupported", []).append(feat)


def functions(src):
    """{name: body} of the top-level fns."""
    out = {}
    for m in re.finditer(r"^(?:pub )?fn (\w+)\([^)]*\)[^{]*\{", src, re.M):
        a = m.end() - 1
        out[m.group(1)] = src[a:balanced(src, a, "{", "}")]
    return out


def unescape(s):
    return json.loads('"' + s + '"')


def cutoff_cases(indexes, cases):
    """cutoff.rs::degraded_search_and_score_details: the same search with Deadline::never().with_stop_after(n)."""
    src = strip_comments(open(f"{REF}/cutoff.rs").read())
    fns = functions(src)
    cfg = {"docs": parse_documents(fns["create_index"])}
    parse_settings(fns["create_index"], cfg)
    cfg.pop("unsupported", None)
    indexes["cutoff::create_index"] = cfg
    body = fns["degraded_search_and_score_details"]
    parts = re.split(r"search\.deadline\(", body)
    for part in parts[1:]:
        m = re.match(r"Deadline::never\(\)(?:\.with_stop_after\((\d+)\))?", part)
        snap = re.search(r'snapshot!\(.*?@r#*"(.*?)"#*\);', part, re.S)
        if not m or not snap:
            continue
        text = snap.group(1)
        ids = json.loads(re.search(r"IDs: (\[[^\]]*\])", text).group(1))
        scores = re.search(r"Scores: ([0-9. ]+)", text).group(1).split()
        details = text[text.index("Score Details:") + len("Score Details:"):]
        cases.append({"src": "crates/milli/src/search/new/tests/cutoff.rs::degraded_search_and_score_details",
                      "index": "cutoff::create_index", "query": "hello puppy kefir", "tms": "last", "detailed": True,
                      "limit": 4, "offset": 0, "ids": ids, "scores": re.sub(r"\s+", "", details),
                      "global_scores": scores, "stop_after": int(m.group(1)) if m.group(1) else None})


def main():
    indexes, cases = {}, []
    for mod in FILES:
        src = strip_comments(open(f"{REF}/{mod}.rs").read())
        fns = functions(src)
        for name, body in fns.items():
            if not name.startswith("test_"):
                continue
            mc = re.search(r"let index = (create_\w+)\(\);", body)
            if not mc:
                continue
            cfg = {"docs": parse_documents(fns[mc.group(1)])}
            parse_settings(fns[mc.group(1)], cfg)
            # events in textual order
            events = []
            for m in re.finditer(r"\.update_settings\(", body):
                events.append((m.start(), "settings", balanced(body, m.end() - 1, "(", ")")))
            for m in re.finditer(r"index\.search\(", body):
                events.append((m.start(), "search", None))
            for m in re.finditer(r"let index = (create_\w+)\(\);", body):
                if m.start() != mc.start():
                    events.append((m.start(), "index", m.group(1)))
            for m in re.finditer(r"index\s*\.add_documents\(", body):
                events.append((m.start(), "docs", balanced(body, m.end() - 1, "(", ")")))
            asserts = [(m.start(), m.end()) for m in
                       re.finditer(r"(?:insta::assert_snapshot!|insta::assert_debug_snapshot!|db_snap!)\(", body)]
            events.sort()
            version = 0
            for k, (pos, kind, end) in enumerate(events):
                if kind == "settings":
                    cfg = dict(cfg)
                    parse_settings(body[pos:end], cfg)
                    version += 1
                    continue
                if kind == "index":
                    cfg = {"docs": parse_documents(fns[end])}
                    parse_settings(fns[end], cfg)
                    version += 1
                    continue
                if kind == "docs":
                    cfg = dict(cfg)
                    cfg["docs"] = cfg["docs"] + parse_documents(body[pos:end])
                    version += 1
                    continue
                nxt = next((p for p, kd, _ in events[k + 1:] if kd == "search"), len(body))
                chunk = body[pos:nxt]
                q = re.search(r's\.query\("((?:[^"\\]|\\.)*)"\)', chunk)
                if not q and mod not in ("distinct", "sort"):
                    continue
                tms = re.search(r"TermsMatchingStrategy::(\w+)", chunk)
                case = {"src": f"crates/milli/src/search/new/tests/{mod}.rs::{name}", "index": f"{mod}::{name}::{version}",
                        "query": unescape(q.group(1)) if q else "", "tms": (tms.group(1).lower() if tms else "last"),
                        "detailed": "ScoringStrategy::Detailed" in chunk, "ids": None, "scores": None}
                lim = re.search(r"s\.limit\((\d+)\)", chunk)
                off = re.search(r"s\.offset\((\d+)\)", chunk)
                dm = re.search(r's\.distinct\(S\("([^"]+)"\)\)', chunk)
                if dm:
                    case["distinct"] = dm.group(1)
                sc = re.search(r"s\.sort_criteria\(vec!\[(.*?)\]\);", chunk, re.S)
                if sc:
                    case["sort"] = [[f, d.lower()] for d, f in
                                    re.findall(r'AscDesc::(Asc|Desc)\(Member::Field\(S\("([^"]+)"\)\)\)', sc.group(1))]
                case["limit"] = int(lim.group(1)) if lim else 20
                case["offset"] = int(off.group(1)) if off else 0
                for n, (a, b) in enumerate(asserts, 1):
                    if not (pos <= a < nxt):
                        continue
                    e = balanced(body, b - 1, "(", ")")
                    text = body[b:e]
                    if "documents_ids:?" in text and "document_ids_scores" not in text:
                        mi = re.search(r'@"(\[[^"]*\])"', text)
                        if mi:
                            case["ids"] = json.loads(mi.group(1))
                    elif "document_scores" in text or "document_ids_scores" in text:
                        if "@" in text:
                            mi = re.search(r'@r#*"(.*?)"#*\s*\)$', text, re.S)
                            snap = mi.group(1) if mi else None
                        else:
                            fn = f"{REF}/snapshots/milli__search__new__tests__{mod}__{name[5:]}" + \
                                 (f"-{n}" if n > 1 else "") + ".snap"
                            snap = open(fn).read().split("---", 2)[2] if os.path.exists(fn) else None
                        if snap is not None:
                            key = "ids_scores" if "document_ids_scores" in text else "scores"
                            case[key] = re.sub(r"\s+", "", snap)
                if case["ids"] is None and "ids_scores" not in case:
                    continue
                if case["index"] not in indexes:
                    indexes[case["index"]] = cfg
                cases.append(case)
    cutoff_cases(indexes, cases)
    json.dump({"indexes": indexes, "cases": cases}, open(OUT, "w"), indent=0, ensure_ascii=False, sort_keys=True)
    print(len(indexes), "indexes,", len(cases), "cases ->", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
