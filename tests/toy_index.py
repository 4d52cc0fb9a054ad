"""TEST INFRASTRUCTURE: a toy inverted index built the way milli's write path builds
its databases and the host-side term derivation of
crates/milli/src/search/new/query_term/{parse_query.rs,compute_derivations.rs}, so
that the reference's snapshot tests (search/new/tests/typo.rs) can be replayed
against the device bucket sort.  Also a brute-force, per-document restatement of
the Words -> Typo order used as the oracle for random corpora."""
import re

import numpy as np

MAX_ONE, MAX_TWO, MAX_PREFIX = 150, 50, 1000   # search/new/limits.rs:5-9


def cbo_bytes(ids):
    """CboRoaringBitmapCodec::serialize_into_writer (cbo_roaring_bitmap_codec.rs:33-51): <= 7
    documents as raw native-endian u32s, else the portable Roaring serialisation."""
    import struct
    ids = np.unique(np.asarray(sorted(ids), dtype=np.uint32))
    if ids.size <= 7:
        return ids.astype("=u4").tobytes()
    keys = np.unique(ids >> 16)
    conts = [(int(k), (ids[(ids >> 16) == k] & 0xFFFF).astype(np.uint16)) for k in keys]
    out = bytearray(struct.pack("<II", 12346, len(conts)))
    for k, v in conts:
        out += struct.pack("<HH", k, len(v) - 1)
    out += b"\0" * (4 * len(conts))
    for k, v in conts:
        if len(v) <= 4096:
            out += v.astype("<u2").tobytes()
        else:
            words = np.zeros(1024, dtype=np.uint64)
            np.bitwise_or.at(words, (v >> 6).astype(np.int64), np.uint64(1) << (v & 63).astype(np.uint64))
            out += words.astype("<u8").tobytes()
    return bytes(out)


def tokenize(text):
    return re.findall(r"[0-9a-zà-ÿа-я]+", text.lower())


class ToyIndex:
    def __init__(self, docs):
        """docs: {docid: text}.  word_docids + word_pair_proximity_docids(prox = 1)."""
        self.docs = dict(docs)
        self.n_docs = max(self.docs) + 1 if self.docs else 0
        self.word_docids = {}
        self.pair1 = {}   # (left, right) adjacent -> docids
        for d, text in self.docs.items():
            toks = tokenize(text)
            for t in toks:
                self.word_docids.setdefault(t, set()).add(d)
            for a, b in zip(toks, toks[1:]):
                self.pair1.setdefault((a, b), set()).add(d)
        self.words = sorted(self.word_docids, key=lambda w: w.encode())   # words-fst order

    # -- the index side of msi_index_vtable (what the Rust shim answers from LMDB) --------
    exact_words = ()

    def word_docids_bytes(self, word, original):
        s = self.word_docids.get(word)
        return cbo_bytes(s) if s else None

    def pair_docids_bytes(self, prox, left, right):
        s = self.pair1.get((left, right)) if prox == 1 else None
        return cbo_bytes(s) if s else None

    def is_exact_word(self, word):
        return word in self.exact_words

    def budget(self, word, exact_words=(), authorize_typos=True):
        # number_of_typos_allowed, parse_query.rs:204-225 (5 / 9 chars)
        n = len(word)
        if not authorize_typos or n < 5 or word in exact_words:
            return 0
        return 1 if n < 9 else 2

    def split_best_frequency(self, word):
        # compute_derivations.rs:363-383
        best = None
        for i in range(1, len(word)):
            l, r = word[:i], word[i:]
            f = len(self.pair1.get((l, r), ()))
            if f and (best is None or f > best[0]):
                best = (f, l, r)
        return (best[1], best[2]) if best else None

    def graph_nodes(self, words, lookup, exact_words=(), authorize_typos=True):
        """Query graph of QueryGraph::from_query (query_graph.rs:96-180): every term, every
        2-gram and 3-gram of adjacent terms (make_ngram, parse_query.rs:227-300: the
        concatenation, typo budget = budget(concat) - (n-1) saturating, prefix flag of its
        last term).  -> [(first, last, zero, one, two, max_typo_cost)]."""
        n = len(words)
        nodes = []
        for last in range(n):
            for size in (1, 2, 3):
                first = last - size + 1
                if first < 0:
                    continue
                is_prefix = last == n - 1
                if size == 1:
                    z, o, t, mc = self.term_sets(words[last], is_prefix, lookup, exact_words, authorize_typos)
                else:
                    z, o, t, mc = self.term_sets("".join(words[first:last + 1]), is_prefix, lookup, exact_words,
                                                 authorize_typos, ngram_of=words[first:last + 1])
                nodes.append((first, last, z, o, t, mc))
        return nodes

    def term_sets(self, word, is_prefix, lookup, exact_words=(), authorize_typos=True, ngram_of=None):
        """-> (zero, one, two docid sets, max_typo_cost) of a single-word term:
        compute_query_term_subset_docids (resolve_query_graph.rs:33-130) over the zero /
        one / two typo subsets; max_typo_cost as query_term/mod.rs:340-370 (full subsets)."""
        b = self.budget(word, exact_words, authorize_typos)
        if ngram_of:
            b = max(0, b - (len(ngram_of) - 1))
        zero = set(self.word_docids.get(word, ()))
        if is_prefix:   # find_zero_typo_prefix_derivations (no prefix DB on a toy corpus)
            n = 0
            for w in self.words:
                if w.startswith(word) and w != word:
                    zero |= self.word_docids[w]
                    n += 1
                    if n >= MAX_PREFIX:
                        break
        one_words, two_words = [], []
        if b >= 1:
            o, t = lookup(word, b, is_prefix)
            one_words, two_words = [self.words[i] for i in o], [self.words[i] for i in t]
        one = set()
        for w in one_words:
            one |= self.word_docids[w]
        sp = self.split_best_frequency(word)      # split words sit in the one-typo subterm
        if sp and ngram_of and list(sp) == list(ngram_of):
            sp = None                              # compute_derivations.rs:296-309
        if sp:
            one |= self.pair1[sp]
        two = set()
        for w in two_words:
            two |= self.word_docids[w]
        max_cost = 1 if b <= 1 else 2             # budget 0: split words allowed -> 1
        return zero, one, two, max_cost


def brute_force_order(n_docs, terms, universe, strategy_all, use_typo):
    """terms: [(zero, one, two, max_cost)] as python sets.  Per-document sort key by
    definition: longest matched prefix of terms (first term mandatory), then the sum
    over the kept terms of the smallest typo level the document matches, then docid."""
    n = len(terms)
    out = []
    for d in sorted(universe):
        k, cost = 0, 0
        for z, o, t, mc in terms:
            lv = 0 if d in z else (1 if (d in o and mc >= 1) else (2 if (d in t and mc >= 2) else None))
            if lv is None:
                break
            k += 1
            cost += lv
        if k == 0 or (strategy_all and k < n):
            continue
        maxc = sum(mc for _, _, _, mc in terms[:k]) if use_typo else 0   # no Typo rule: no Typo score
        out.append((-k, cost if use_typo else 0, d, k, cost if use_typo else 0, maxc))
    out.sort()
    return [(d, k, c, m) for _, _, d, k, c, m in out]


def brute_force_graph_order(nodes, n_terms, universe, strategy_all, use_typo):
    """Per-document restatement over the n-gram DAG: positions reached, the smallest typo
    cost at the LARGEST reached position (n-gram base cost = its length), docid."""
    per_doc = []
    for d in sorted(universe):
        INF = 10 ** 9
        best = {0: 0}         # position -> min cost
        worst = {0: 0}        # position -> max structural cost of a matching path
        for p in range(1, n_terms + 1):
            for first, last, z, o, t, mc in nodes:
                if last != p - 1 or first not in best:
                    continue
                size = last - first + 1
                base = 0 if size == 1 else size
                lv = 0 if d in z else (1 if (d in o and mc >= 1) else (2 if (d in t and mc >= 2) else None))
                if lv is None:
                    continue
                c = best[first] + base + lv
                if c < best.get(p, INF):
                    best[p] = c
                w = worst[first] + base + mc
                if w > worst.get(p, -1):
                    worst[p] = w
        k = max(best)
        if k == 0 or (strategy_all and k < n_terms):
            continue
        per_doc.append((d, k, best[k] if use_typo else 0, worst[k]))
    maxc = {}
    for d, k, c, w in per_doc:
        maxc[k] = max(maxc.get(k, 0), w)
    out = sorted((-k, c, d, k, c, maxc[k] if use_typo else 0) for d, k, c, w in per_doc)
    return [(d, k, c, m) for _, _, d, k, c, m in out]
