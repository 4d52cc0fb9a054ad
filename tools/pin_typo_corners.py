"""VERDICT r1 #9 — can any search the reference's own tests run tell apart the typo-matcher semantics libmsi restates
(OSA distance; prefix DFA = minimum over the prefixes of the dictionary word) from the ones it could be mistaken for
(unrestricted Damerau-Levenshtein; prefix DFA = distance of the first prefix within the budget)?

Runs HERE only (reads /root/reference): every query string of crates/milli/src/search/new/tests/*.rs and
crates/meilisearch/tests/search/*.rs against a dictionary that is a SUPERSET of every index those tests build (every
alphanumeric token of every string literal in those files and of the JSON datasets they load).  A (query word,
dictionary word) pair is reported when the variants disagree on its typo class {0, 1, 2, no match} under the word's
budget (number_of_typos_allowed: < 5 letters 0, < 9 letters 1, else 2; first-letter rule applied to all variants alike).
Writes tests/golden/typo_corner_scan.json."""
import ctypes as C
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
so = os.path.join(ROOT, "tools", "bin", "pin_typo_corners.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "pin_typo_corners.c")])
L = C.CDLL(so)

files = sorted(glob.glob(REF + "/crates/milli/src/search/new/tests/*.rs") + glob.glob(REF + "/crates/meilisearch/tests/search/*.rs")
               + glob.glob(REF + "/crates/meilisearch/tests/common/*.rs") + glob.glob(REF + "/crates/milli/tests/search/*.rs"))
datasets = sorted(glob.glob(REF + "/crates/meilisearch/tests/assets/*.json") + glob.glob(REF + "/crates/milli/tests/assets/*.jsonl")
                  + glob.glob(REF + "/crates/milli/tests/assets/*.json"))
word_re = re.compile(r"[a-z0-9]+")
queries, dictionary = set(), set()
for f in files:
    src = open(f, encoding="utf-8", errors="ignore").read()
    for m in re.finditer(r'(?:\.query\(\s*|"q"\s*:\s*)"((?:[^"\\]|\\.)*)"', src):
        queries.add(m.group(1).replace('\\"', '"').lower())
    for m in re.finditer(r'"((?:[^"\\]|\\.)*)"', src):
        dictionary.update(word_re.findall(m.group(1).lower()))
for f in datasets:
    if os.path.getsize(f) < 40 << 20:
        dictionary.update(word_re.findall(open(f, encoding="utf-8", errors="ignore").read().lower()))
dictionary = sorted(w for w in dictionary if len(w) <= 40)
qwords = set()
for q in queries:
    ws = word_re.findall(q)
    for i, w in enumerate(ws):
        last = i == len(ws) - 1 and not q.rstrip().endswith('"') and not q.endswith(" ")
        qwords.add((w, last))
        qwords.add((w, False))
        for n in (2, 3):                                   # ngrams are looked up like words
            if i + n <= len(ws):
                qwords.add(("".join(ws[i:i + n]), False))
qwords = sorted(w for w in qwords if 1 <= len(w[0]) <= 40)


def budget(w):
    return 0 if len(w) < 5 else (1 if len(w) < 9 else 2)


def cls(d, w, q, b):
    if d > 2:
        return None
    if d >= 1 and w[:1] != q[:1]:                          # a typo on the first letter counts double
        d = 2 if d == 1 else 3
    return d if d <= b else None


diffs = {"osa_vs_damerau": [], "prefix_min_vs_first": []}
pairs = 0
for q, is_prefix in qwords:
    b = budget(q)
    if b == 0:
        continue
    qb = q.encode()
    for w in dictionary:
        if not is_prefix and abs(len(w) - len(q)) > 2:
            continue
        if len(set(q[:6]) & set(w[:8])) < min(3, len(q) - 2):   # cheap reject: nothing in common up front
            continue
        wb = w.encode()
        pairs += 1
        if is_prefix:
            v1 = L.v_prefix(qb, len(qb), wb, len(wb), b, 0, 0)
            v2 = L.v_prefix(qb, len(qb), wb, len(wb), b, 0, 1)
            v3 = L.v_prefix(qb, len(qb), wb, len(wb), b, 1, 0)
        else:
            v1 = L.v_osa(qb, len(qb), wb, len(wb))
            v2 = L.v_damerau(qb, len(qb), wb, len(wb))
            v3 = v1
        c1, c2, c3 = cls(v1, w, q, b), cls(v2, w, q, b), cls(v3, w, q, b)
        if c1 != c2:
            diffs["osa_vs_damerau"].append({"query_word": q, "prefix": is_prefix, "dictionary_word": w, "osa": v1, "damerau": v2})
        if c1 != c3:
            diffs["prefix_min_vs_first"].append({"query_word": q, "dictionary_word": w, "min_over_prefixes": v1, "first_within_budget": v3})
out = {"generated_by": "tools/pin_typo_corners.py", "query_strings": len(queries), "query_words": len(qwords),
       "dictionary_superset_words": len(dictionary), "pairs_compared": pairs,
       "pairs_where_variants_disagree": {k: len(v) for k, v in diffs.items()},
       "osa_vs_damerau": diffs["osa_vs_damerau"][:50], "prefix_min_vs_first": diffs["prefix_min_vs_first"][:200]}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "typo_corner_scan.json"), "w"), indent=1)
print(json.dumps({k: out[k] for k in list(out)[:6]}, indent=1))
