"""TEST INFRASTRUCTURE — not part of the product, never imported by meilisearch_amd.

The checker's view of the synthetic inverted index of tools/ranked_bench.cpp (the 10 M-document index behind the
keyword leg of the headline step): the interface oracle/ranking_oracle.py reads an index through
(tests/toy_milli.ToyMilli's read methods = the reads of search/new/db_cache.rs), answered from the SAME stored
CboRoaringBitmap bytes the product's msi_index_vtable callbacks return (rb_read / rb_read_keys / rb_words of
tools/bin/libmsi_rankedbench.so), decoded here by oracle/docset.py — an independent decoder — into the oracle's
DocSet type.  The index lives in host memory; nothing here needs a device.

What the synthetic index is (tools/ranked_bench.cpp): a dictionary of n_words random words; word_docids with Zipf
document frequencies; word_fid_docids over fids 1..3 (weights 0..2), word_position_docids over 20 bucketed positions,
word_pair_proximity_docids at proximities 1..3 for ANY two words, field_id_word_count_docids; no exact attributes, no
word-prefix databases, no synonyms, no stop words.  The databases are not those of one coherent corpus — the ranking
rules only read them, and both sides read the same bytes.

Round 4: rb_create_corpus builds a COHERENT corpus instead (tools/ranked_bench.cpp, struct Corpus: documents of a title and
an overview, Zipf(1.07) words, every database derived from the same tokens the way tests/toy_milli.py derives them; two
searchable fields) behind the same reads; queries are taken out of the documents and misspelled."""
import bisect
import ctypes as C
import os

import numpy as np

from oracle import docset

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_CRITERIA = ["words", "typo", "proximity", "attributeRank", "sort", "wordPosition", "exactness"]


def runner_lib():
    # MSI_RUNNER_SO: the CPU tier's build of the same source against the emulated kernels (tests/emu/run_emulated.py)
    lib = C.CDLL(os.environ.get("MSI_RUNNER_SO") or os.path.join(ROOT, "tools", "bin", "libmsi_rankedbench.so"))
    lib.rb_create.restype = C.c_void_p
    lib.rb_create.argtypes = [C.c_uint64, C.c_uint32]
    lib.rb_create_corpus.restype = C.c_void_p
    lib.rb_create_corpus.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64]
    lib.rb_n_fields.restype = C.c_uint32
    lib.rb_n_fields.argtypes = [C.c_void_p]
    lib.rb_prepare_queries.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64]
    lib.rb_prepare_queries_ex.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32]
    lib.rb_enable_prefix_dbs.argtypes = [C.c_void_p, C.c_uint32]
    lib.rb_enable_synonyms.argtypes = [C.c_void_p]
    lib.rb_query_negatives.restype = C.c_uint32
    lib.rb_query_negatives.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32]
    lib.rb_synonyms.restype = C.c_uint32
    lib.rb_synonyms.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_uint32]
    lib.rb_has_prefix.restype = C.c_uint32
    lib.rb_has_prefix.argtypes = [C.c_void_p, C.c_char_p, C.c_uint32]
    lib.rb_read_multi.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32, C.c_void_p,
                                  C.c_uint64, C.c_void_p, C.c_uint32]
    lib.rb_destroy.argtypes = [C.c_void_p]
    lib.rb_doc_tokens.restype = C.c_uint32
    lib.rb_doc_tokens.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    lib.rb_n_words.restype = C.c_uint32
    lib.rb_n_words.argtypes = [C.c_void_p]
    lib.rb_words.restype = C.c_uint64
    lib.rb_words.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]
    lib.rb_query.restype = C.c_uint32
    lib.rb_query.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32]
    lib.rb_read.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32,
                            C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    lib.rb_read_keys.argtypes = [C.c_void_p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]
    lib.rb_run_detailed.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32] + [C.c_void_p] * 6
    lib.rb_run_universes.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32] + [C.c_void_p] * 6
    return lib


class SynthIndex:
    """The reads of search/new/db_cache.rs over the runner's index `h` (a Runner* of libmsi_rankedbench.so)."""

    def __init__(self, lib, h, n_docs, min_one=5, min_two=9, cache_bytes=6 << 30):
        self.lib, self.h, self.n_docs = lib, h, int(n_docs)
        self.DocSet = docset.docset_type(self.n_docs)
        self.min_one, self.min_two, self.authorize_typos = min_one, min_two, True
        self.criteria = list(DEFAULT_CRITERIA)
        nf = int(lib.rb_n_fields(h))          # 3 for the hashed index, 2 (title, overview) for the coherent corpus
        self.searchable_fids = list(range(1, nf + 1))
        self.weights = {f: f - 1 for f in self.searchable_fids}
        self.max_weight = nf - 1
        self.stop_words, self.exact_words, self.distinct_field = (), set(), None
        n = lib.rb_n_words(h)
        off = np.zeros(n + 1, np.uint32)
        size = lib.rb_words(h, None, 0, off.ctypes.data)
        buf = np.zeros(max(size, 1), np.uint8)
        lib.rb_words(h, buf.ctypes.data, size, off.ctypes.data)
        raw = buf.tobytes()
        self.concat, self.offsets = buf[:size], off
        self.words = [raw[off[i]:off[i + 1]].decode() for i in range(n)]      # dictionary (byte) order
        self._wordset = set(self.words)
        self._cache, self._cache_bytes, self._cache_cap = {}, 0, cache_bytes
        self.reads = 0

    # ---- stored bytes -> DocSet ---------------------------------------------------------------------------
    def _read(self, db, a=b"", b=b"", x=0, y=0):
        key = (db, a, b, x, y)
        if key in self._cache:
            hit = self._cache[key]
            return None if hit is None else self.DocSet(hit)
        self.reads += 1
        ptr, n = C.c_void_p(), C.c_size_t()
        st = self.lib.rb_read(self.h, db, a, len(a), b, len(b), x, y, C.byref(ptr), C.byref(n))
        assert st == 0
        if n.value == 0:
            val = None
        else:
            ids = docset.decode_cbo(C.string_at(ptr.value, n.value))
            val = self.DocSet.from_sorted(ids) if ids.size else None
        cost = 64 if val is None else (val.ids.nbytes if val.bits is None else val.bits.nbytes)
        if self._cache_bytes + cost > self._cache_cap:
            self._cache.clear()
            self._cache_bytes = 0
        self._cache[key] = val
        self._cache_bytes += cost
        return None if val is None else self.DocSet(val)

    def _keys(self, db, w):
        out = np.zeros(64, np.uint16)
        cnt = C.c_uint32()
        a = w.encode()
        assert self.lib.rb_read_keys(self.h, db, a, len(a), out.ctypes.data, 64, C.byref(cnt)) == 0
        return sorted(int(v) for v in out[:cnt.value])

    # ---- the interface of tests/toy_milli.ToyMilli that oracle/ranking_oracle.py uses --------------------------
    def all_docids(self):
        return self.DocSet.full()

    def contains_word(self, w):
        return w in self._wordset

    def get_word_docids(self, w, original):
        return self._read(0, w.encode())

    def get_pair(self, prox, w1, w2):
        return self._read(1, w1.encode(), w2.encode(), prox)

    def get_word_fid_docids(self, w, fid):
        return self._read(2, w.encode(), b"", fid)

    def get_word_position_docids(self, w, pos):
        return self._read(3, w.encode(), b"", pos)

    def get_word_fids(self, w):
        return self._keys(0, w)

    def get_word_positions(self, w):
        return self._keys(1, w)

    def get_fid_word_count_docids(self, fid, count):
        return self._read(4, b"", b"", fid, count)

    # the word-prefix databases (rb_enable_prefix_dbs; without them every read below finds nothing): the values the
    # index hands to the engine's sink, decoded by this side's own decoder and united
    def _read_multi(self, db, a=b"", b=b"", x=0):
        key = ("m", db, a, b, x)
        if key in self._cache:
            hit = self._cache[key]
            return None if hit is None else self.DocSet(hit)
        self.reads += 1
        cap = 64 << 20
        if not hasattr(self, "_multi_buf"):
            self._multi_buf = np.zeros(cap, np.uint8)
            self._multi_lens = np.zeros(1 << 16, np.uint32)
        n = self.lib.rb_read_multi(self.h, db, a, len(a), b, len(b), x, self._multi_buf.ctypes.data, cap,
                                   self._multi_lens.ctypes.data, self._multi_lens.size)
        assert n >= 0, "rb_read_multi: more values than the probe buffer holds"
        val, at = None, 0
        for i in range(n):
            ln = int(self._multi_lens[i])
            ids = docset.decode_cbo(self._multi_buf[at:at + ln].tobytes())
            at += ln
            if ids.size:
                one = self.DocSet.from_sorted(ids)
                val = one if val is None else val | one
        self._cache[key] = val
        return None if val is None else self.DocSet(val)

    def has_prefix(self, pfx, include_exact):
        a = pfx.encode()
        return bool(self.lib.rb_has_prefix(self.h, a, len(a)))

    def get_word_prefix_docids(self, pfx, original):
        return self._read_multi(5, pfx.encode())

    def get_word_prefix_fid_docids(self, pfx, fid):
        return self._read_multi(6, pfx.encode(), b"", fid)

    def get_word_prefix_position_docids(self, pfx, pos):
        return self._read_multi(7, pfx.encode(), b"", pos)

    def get_word_prefix_fids(self, pfx):
        return self._keys(2, pfx) if self.has_prefix(pfx, False) else []

    def get_word_prefix_positions(self, pfx):
        return self._keys(3, pfx) if self.has_prefix(pfx, False) else []

    def get_word_prefix_pair(self, prox, w1, pfx2):
        got = self._read_multi(8, w1.encode(), pfx2.encode(), prox)
        return self.DocSet() if got is None else got

    def prefix_words(self, prefix):
        lo = bisect.bisect_left(self.words, prefix)       # ASCII words: str order = byte order
        out = []
        while lo < len(self.words) and self.words[lo].startswith(prefix):
            out.append(self.words[lo])
            lo += 1
        return out

    def get_synonyms(self, words):
        buf = C.create_string_buffer(1024)
        n = self.lib.rb_synonyms(self.h, " ".join(words).encode(), buf, 1024)
        return [line.split(" ") for line in buf.value.decode().split("\n") if line] if n else []

    def budget(self, word):
        n = len(word)
        if n < self.min_one:
            return 0
        return 1 if n < self.min_two else 2

    def query(self, i):
        buf = C.create_string_buffer(512)
        self.lib.rb_query(self.h, i, buf, 512)
        return buf.value.decode()

    def negatives(self, i):
        """The negative terms of query i (rb_prepare_queries_ex, flag 8): [word | (phrase words...)]."""
        if not hasattr(self.lib, "rb_query_negatives"):
            return []
        buf = C.create_string_buffer(512)
        self.lib.rb_query_negatives(self.h, i, buf, 512)
        out = []
        for line in buf.value.decode().split("\n"):
            if line:
                out.append(tuple(line.strip('"').split(" ")) if line.startswith('"') else line)
        return out


class KeywordOracle:
    """oracle/ranking_oracle.py over a SynthIndex: search(query) -> (docids, score details per hit, n candidates)."""

    def __init__(self, index):
        from oracle import oracle as O
        self.index = index
        self.dic = O.Dictionary.from_flat(index.concat, index.offsets)
        self.O = O

    def lookup(self, word, max_typos, is_prefix):
        one, two = self.O.typo_lookup(self.dic, word, max_typos, is_prefix)
        w = self.index.words
        return [w[i] for i in one], [w[i] for i in two]

    def search(self, query, limit=20, detailed=True, tms="last", universe=None, negatives=()):
        from oracle import ranking_oracle as RO
        with RO.use_docset(self.index.DocSet):
            uni = None if universe is None else self.index.DocSet.from_sorted(np.unique(np.asarray(universe, dtype=np.uint32)))
            ids, scores, cand = RO.search(RO.Ctx(self.index, self.lookup), query, tms=tms, criteria=self.index.criteria,
                                          length=limit, detailed=detailed, universe=uni, negatives=negatives)
        return ids, scores, len(cand)


SCORE_KINDS = None


def product_details(ids, n, details, n_details, limit, max_details):
    """The runner's flat outputs of one query -> [(docid, [(kind name, a, b)])] in the oracle's vocabulary."""
    global SCORE_KINDS
    if SCORE_KINDS is None:
        from meilisearch_amd.ranking import SCORE_KINDS as K
        SCORE_KINDS = K
    out = []
    for i in range(int(n)):
        det = [(SCORE_KINDS[int(details[i, k, 0])], int(details[i, k, 1]), int(details[i, k, 2])) for k in range(int(n_details[i]))]
        out.append((int(ids[i]), det))
    return out


def oracle_detail(s):
    if s[0] == "ExactAttribute":
        return ("ExactAttribute", {"ExactMatch": 3, "MatchesStart": 2, "NoExactMatch": 1}[s[1]], 3)
    return tuple(s)
