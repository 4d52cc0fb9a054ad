"""TEST INFRASTRUCTURE: a toy index with every database the ranking rules read, built the
way milli's write path builds them, so that the reference's snapshot
tests (crates/milli/src/search/new/tests/*.rs) can be replayed.  Not part of the product.

  word_docids / exact_word_docids          extract_word_docids.rs:76-81
  word_fid_docids, word_position_docids    extract_word_docids.rs:91-99 (bucketed_position, lib.rs:248-262)
  field_id_word_count_docids (<= 30 words) extract_word_docids.rs:169-190
  word_pair_proximity_docids               extract_word_pair_proximity_docids.rs:504-515 (forward pairs, prox 1..3,
                                           the minimum proximity of a pair per document: :232-233)
  positions                                tokenize_document.rs:131-157 (+1 per word, +8 over a hard separator)
  words fst                                keys of word_docids only (post_processing/mod.rs:192-208)
"""
import re

from oracle.ranking_oracle import bucketed_position
from tests.toy_index import cbo_bytes  # noqa: F401  (re-exported for the device tests)

HARD_RE = re.compile(r"[.,]\s|[!;?]")       # charabia CONTEXT_SEPARATORS (Latin subset): ". " ", " "!" ";" "?"
WORD_RE = re.compile(r"[0-9a-zà-öø-ÿ]+")
MAX_COUNTED_WORDS = 30
PREFIX_MAX_LEN = 4
MAX_POSITION_PER_ATTRIBUTE = 1 << 16      # lib.rs (u16 relative positions)
INDEX_MAX_DISTANCE = 8      # tokenize_document.rs:13
MAX_DISTANCE = 4            # proximity.rs:7


def tokenize_with_positions(text, start=0, stop_words=()):
    """[(word, position)]: process_tokens, tokenize_document.rs:131-157.  Stop words (matched on the token as
    written: "The" is not "the", stop_words.rs:5) keep their position but are dropped."""
    raw, text = text, text.lower()
    out, pos, prev_end, first = [], start, 0, True
    for m in WORD_RE.finditer(text):
        sep = text[prev_end:m.start()]
        if first:
            first = False
        else:
            hard = bool(HARD_RE.search(sep))
            pos += INDEX_MAX_DISTANCE if hard else 1
        if raw[m.start():m.end()] not in stop_words:
            out.append((m.group(0), pos))
        prev_end = m.end()
    return out


def normalize_facet(s):
    """lib.rs:442-444: CompatibilityDecompositionNormalizer (NFKD) of the trimmed value, lowercased."""
    import unicodedata
    return unicodedata.normalize("NFKD", s.strip()).lower()


class ToyMilli:
    def __init__(self, docs, searchable=None, exact_attributes=(), exact_words=(), criteria=None,
                 min_one=5, min_two=9, authorize_typos=True, primary_key="id", prefix_threshold=100, synonyms=None,
                 stop_words=(), distinct=None):
        self.min_one, self.min_two, self.authorize_typos = min_one, min_two, authorize_typos
        self.exact_words = set(exact_words)
        self.distinct_field = distinct
        self.stop_words = set(stop_words)      # case sensitive, compared with the token as written
        # index.synonyms: normalised key words -> synonym phrases as word lists (settings: "a b" -> ["c d", ...])
        self.synonyms = {tuple(WORD_RE.findall(k.lower())): [WORD_RE.findall(v.lower()) for v in vs]
                         for k, vs in (synonyms or {}).items()}
        self.criteria = criteria or ["words", "typo", "proximity", "attributeRank", "sort", "wordPosition", "exactness"]
        # internal docids in order of first appearance of the external id; a later document with
        # the same id replaces the earlier one
        ext, merged = {}, []
        for d in docs:
            k = d[primary_key]
            if k in ext:
                merged[ext[k]] = d
            else:
                ext[k] = len(merged)
                merged.append(d)
        self.docs = merged
        self.n_docs = len(merged)
        # geo_faceted_documents_ids + the (lat, lng) geo_value reads (documents/geo_sort.rs:247-276): numbers, or
        # strings that parse as f64
        self.geo_points = {}
        for docid, d in enumerate(merged):
            g = d.get("_geo")
            if isinstance(g, dict) and "lat" in g and "lng" in g:
                self.geo_points[docid] = (float(g["lat"]), float(g["lng"]))
        self.fields = {}
        for d in merged:
            for name in d:
                self.fields.setdefault(name, len(self.fields))
        if searchable is None:
            self.searchable = [n for n in self.fields]
            self.weights = {self.fields[n]: 0 for n in self.searchable}
            self.max_weight = None
        else:
            self.searchable = [n for n in searchable if n in self.fields]
            self.weights = {self.fields[n]: i for i, n in enumerate(searchable) if n in self.fields}
            self.max_weight = max(len(searchable) - 1, 0)
        self.searchable_fids = [self.fields[n] for n in self.searchable]
        exact_attr = set(exact_attributes)
        self.exact_attribute_names = exact_attr
        self.word_docids, self.exact_word_docids = {}, {}
        self.word_fid_docids, self.word_position_docids = {}, {}
        self.fid_word_count, self.pair = {}, {}
        for docid, d in enumerate(merged):
            pairs = {}
            for name in self.searchable:
                if name not in d or isinstance(d[name], bool):
                    continue
                fid = self.fields[name]
                if isinstance(d[name], list):
                    # every further value of the same field starts INDEX_MAX_DISTANCE after the last position of
                    # the previous one (tokenize_document.rs:77-83) — pinned by the `surname` arrays of the index
                    # milli wrote (tests/golden/index_blobs.json: kef 0, kefkef 8, kefirounet 16, boubou 24)
                    toks, start = [], 0
                    for v in d[name]:
                        if isinstance(v, bool) or not isinstance(v, (str, int, float)):
                            continue
                        part = tokenize_with_positions(str(v), start=start, stop_words=self.stop_words)
                        if part:
                            toks += part
                            start = part[-1][1] + INDEX_MAX_DISTANCE
                elif isinstance(d[name], (str, int, float)):
                    toks = tokenize_with_positions(str(d[name]), stop_words=self.stop_words)
                else:
                    continue
                toks = [(w, p) for w, p in toks if p < MAX_POSITION_PER_ATTRIBUTE]
                target = self.exact_word_docids if name in exact_attr else self.word_docids
                for w, p in toks:
                    target.setdefault(w, set()).add(docid)
                    self.word_fid_docids.setdefault((w, fid), set()).add(docid)
                    self.word_position_docids.setdefault((w, bucketed_position(p)), set()).add(docid)
                if 0 < len(toks) <= MAX_COUNTED_WORDS:
                    self.fid_word_count.setdefault((fid, len(toks)), set()).add(docid)
                for i, (w1, p1) in enumerate(toks):
                    for w2, p2 in toks[i + 1:]:
                        prox = min(p2 - p1, MAX_DISTANCE)
                        if 0 < prox < MAX_DISTANCE:
                            key = (w1, w2)
                            if key not in pairs or prox < pairs[key]:
                                pairs[key] = prox
            for (w1, w2), prox in pairs.items():
                self.pair.setdefault((prox, w1, w2), set()).add(docid)
        # facet databases of scalar values (facet_id_string_docids / facet_id_f64_docids level 0 and the per-document
        # field_id_docid_facet_* entries): what `distinct` reads (search/new/distinct.rs:38-62)
        self.facet_docids, self.doc_facets = {}, {}
        for docid, d in enumerate(merged):
            for name, v in d.items():
                vals = v if isinstance(v, list) else [v]
                for x in vals:
                    if isinstance(x, bool):
                        key = ("s", str(x).lower())
                    elif isinstance(x, (int, float)):
                        key = ("n", float(x))
                    elif isinstance(x, str) and x:
                        key = ("s", normalize_facet(x))
                    else:
                        continue
                    self.facet_docids.setdefault((name, key), set()).add(docid)
                    self.doc_facets.setdefault((name, docid), []).append(key)
        self.words = sorted(self.word_docids, key=lambda w: w.encode())      # words fst
        self.all_words = sorted(set(self.word_docids) | set(self.exact_word_docids), key=lambda w: w.encode())
        # word-prefix databases: prefixes of 1..4 bytes (at a char boundary) shared by >= 100 words of the words
        # fst (word_fst_builder.rs:100-140, index.rs:1884-1887), each with the union of its words' postings
        # (update/new/words_prefix_docids.rs)
        counts = {}
        for w in self.words:
            b = w.encode()
            for n in range(1, PREFIX_MAX_LEN + 1):
                try:
                    pfx = b[:n].decode()
                except UnicodeDecodeError:
                    continue
                if len(b) >= n:
                    counts[pfx] = counts.get(pfx, 0) + 1
        self.prefixes = {p_ for p_, c in counts.items() if c >= prefix_threshold}
        self.word_prefix_docids, self.exact_word_prefix_docids = {}, {}
        self.word_prefix_fid_docids, self.word_prefix_position_docids = {}, {}
        for pfx in self.prefixes:
            for w, s_ in self.word_docids.items():
                if w.startswith(pfx):
                    self.word_prefix_docids.setdefault(pfx, set()).update(s_)
            for w, s_ in self.exact_word_docids.items():
                if w.startswith(pfx):
                    self.exact_word_prefix_docids.setdefault(pfx, set()).update(s_)
            for (w, fid), s_ in self.word_fid_docids.items():
                if w.startswith(pfx):
                    self.word_prefix_fid_docids.setdefault((pfx, fid), set()).update(s_)
            for (w, pos), s_ in self.word_position_docids.items():
                if w.startswith(pfx):
                    self.word_prefix_position_docids.setdefault((pfx, pos), set()).update(s_)
        self._fids_of, self._pos_of = {}, {}
        for (w, fid) in self.word_fid_docids:
            self._fids_of.setdefault(w, []).append(fid)
        for (w, p) in self.word_position_docids:
            self._pos_of.setdefault(w, []).append(p)

    # ---- the reads of search/new/db_cache.rs ----------------------------------------
    def all_docids(self):
        return set(range(self.n_docs))

    def restricted(self, attributes_to_search_on):
        """The view of this index a request with `attributesToSearchOn` searches (search/new/mod.rs:140-222): the same
        databases read through other functions (db_cache.rs:208-345,540-575).  `index_view` names the view for the engine's
        caches (msi_search_params::index_view).  "*" = no restriction (the index itself)."""
        if "*" in attributes_to_search_on:
            return self
        import copy
        import zlib
        v = copy.copy(self)
        exact_attr = getattr(self, "exact_attribute_names", set())
        fids = [self.fields[n] for n in attributes_to_search_on if n in self.fields and n in self.searchable]
        v.tolerant_fids = [f for f in fids if self.searchable[self.searchable_fids.index(f)] not in exact_attr]
        v.exact_fids = [f for f in fids if f not in v.tolerant_fids]
        v.index_view = 1 + zlib.crc32(",".join(sorted(attributes_to_search_on)).encode())
        base = self

        def union(keys, db):
            found = [db[k] for k in keys if k in db]
            if not found:
                return None
            out = set()
            for s_ in found:
                out |= s_
            return out

        def get_word_docids(w, original):   # db_cache.rs:208-269 + :183-205
            t = union([(w, f) for f in v.tolerant_fids], base.word_fid_docids)
            if not original:
                return t
            e = union([(w, f) for f in v.exact_fids], base.word_fid_docids)
            if t is None and e is None:
                return None
            return (t or set()) | (e or set())

        def get_word_prefix_docids(pfx, original):   # db_cache.rs:297-358 + :272-294
            t = union([(pfx, f) for f in v.tolerant_fids], base.word_prefix_fid_docids)
            if not original:
                return t
            e = union([(pfx, f) for f in v.exact_fids], base.word_prefix_fid_docids)
            if t is None and e is None:
                return None
            return (t or set()) | (e or set())

        v.get_word_docids = get_word_docids
        v.get_word_prefix_docids = get_word_prefix_docids
        v.contains_word = base.contains_word            # Index::contains_word reads the plain databases
        v.get_word_fid_docids = lambda w, fid: base.word_fid_docids.get((w, fid)) if fid in fids else None   # :540-543
        v.get_word_prefix_fid_docids = lambda pfx, fid: base.word_prefix_fid_docids.get((pfx, fid)) if fid in fids else None
        # the vtable side (what the shim's callbacks hand over under this view)
        v.word_docids_bytes = lambda word, original: (lambda s_: cbo_bytes(s_) if s_ else None)(get_word_docids(word, original))
        v.word_fid_docids_bytes = lambda word, fid: (lambda s_: cbo_bytes(s_) if s_ else None)(v.get_word_fid_docids(word, fid))
        v.word_prefix_docids_values = lambda pfx, original: (lambda s_: [cbo_bytes(s_)] if s_ else [])(get_word_prefix_docids(pfx, original))
        v.word_prefix_fid_docids_values = lambda pfx, fid: (lambda s_: [cbo_bytes(s_)] if s_ else [])(v.get_word_prefix_fid_docids(pfx, fid))
        return v

    def contains_word(self, w):
        return w in self.word_docids or w in self.exact_word_docids

    def get_word_docids(self, w, original):
        """SearchContext::word_docids (db_cache.rs:183-205): Original = exact | tolerant, Derived = tolerant."""
        t = self.word_docids.get(w)
        if not original:
            return t
        e = self.exact_word_docids.get(w)
        if t is None and e is None:
            return None
        return (t or set()) | (e or set())

    def get_pair(self, prox, w1, w2):
        return self.pair.get((prox, w1, w2))

    def get_word_fid_docids(self, w, fid):
        return self.word_fid_docids.get((w, fid))

    def get_word_position_docids(self, w, pos):
        return self.word_position_docids.get((w, pos))

    def get_word_fids(self, w):
        return sorted(self._fids_of.get(w, ()))

    def get_word_positions(self, w):
        return sorted(self._pos_of.get(w, ()))

    def get_fid_word_count_docids(self, fid, count):
        return self.fid_word_count.get((fid, count))

    # -- word-prefix databases ----------------------------------------------------------------------
    def get_word_prefix_docids(self, pfx, original):
        """SearchContext::word_prefix_docids (db_cache.rs:272-294)."""
        t = self.word_prefix_docids.get(pfx)
        if not original:
            return t
        e = self.exact_word_prefix_docids.get(pfx)
        if t is None and e is None:
            return None
        return (t or set()) | (e or set())

    def has_prefix(self, pfx, include_exact):
        return pfx in self.word_prefix_docids or (include_exact and pfx in self.exact_word_prefix_docids)

    def get_word_prefix_fid_docids(self, pfx, fid):
        return self.word_prefix_fid_docids.get((pfx, fid))

    def get_word_prefix_position_docids(self, pfx, pos):
        return self.word_prefix_position_docids.get((pfx, pos))

    def get_word_prefix_fids(self, pfx):
        return sorted({f for (p_, f) in self.word_prefix_fid_docids if p_ == pfx})

    def get_word_prefix_positions(self, pfx):
        return sorted({q for (p_, q) in self.word_prefix_position_docids if p_ == pfx})

    def get_word_prefix_pair(self, prox, w1, pfx2):
        """get_db_word_prefix_pair_proximity_docids (db_cache.rs:451-520): prefix_iter over word_pair_proximity."""
        out = set()
        for (p_, a, b), s_ in self.pair.items():
            if p_ == prox and a == w1 and b.startswith(pfx2):
                out |= s_
        return out

    def prefix_words(self, prefix):
        """word_docids and exact_word_docids keys with the prefix, merged in key order
        (find_zero_typo_prefix_derivations, compute_derivations.rs:40-73)."""
        return [w for w in self.all_words if w.startswith(prefix)]

    def budget(self, word):
        n = len(word)
        if not self.authorize_typos or n < self.min_one or word in self.exact_words:
            return 0
        return 1 if n < self.min_two else 2

    # ---- the index side of msi_index_vtable (what the Rust shim answers from LMDB) -------------
    def word_docids_bytes(self, word, original):
        s = self.get_word_docids(word, original)
        return cbo_bytes(s) if s else None

    def pair_docids_bytes(self, prox, left, right):
        s = self.pair.get((prox, left, right))
        return cbo_bytes(s) if s else None

    def is_exact_word(self, word):
        return word in self.exact_words

    def word_fid_docids_bytes(self, word, fid):
        s = self.word_fid_docids.get((word, fid))
        return cbo_bytes(s) if s else None

    def word_position_docids_bytes(self, word, pos):
        s = self.word_position_docids.get((word, pos))
        return cbo_bytes(s) if s else None

    def word_fids(self, word):
        return self.get_word_fids(word)

    def word_positions(self, word):
        return self.get_word_positions(word)

    def fid_word_count_docids_bytes(self, fid, count):
        s = self.fid_word_count.get((fid, count))
        return cbo_bytes(s) if s else None


    def order_keys(self, field, ascending):
        """What the shim stages for a Sort / Asc / Desc rule (include/msi.h, msi_doc_keys): per document the rank of
        the first facet value of `field` that ascending_facet_sort / descending_facet_sort meets (numbers, then
        strings, each in the rule's direction; sort.rs:95-233), 0xFFFFFFFF without a value.
        -> (keys[n_docs] as a list, values[rank] = ("n", float) | ("s", str))."""
        keys = [k for (f, k) in self.facet_docids if f == field]
        nums = sorted((k for k in keys if k[0] == "n"), key=lambda k: k[1], reverse=not ascending)
        strs = sorted((k for k in keys if k[0] == "s"), key=lambda k: k[1].encode(), reverse=not ascending)
        values = nums + strs
        out = [0xFFFFFFFF] * self.n_docs
        for rank, k in enumerate(values):
            for d in self.facet_docids[(field, k)]:
                out[d] = min(out[d], rank)
        return out, values

    # ---- what a filter reads (search/facet/filter/index_filter.rs) --------------------------------------------------
    def facet_numbers(self, field):
        """per document: the numbers of facet_id_f64_docids for `field`; `_geo.lat` / `_geo.lng` come from the point."""
        if field in ("_geo.lat", "_geo.lng"):
            i = 0 if field.endswith("lat") else 1
            return [[self.geo_points[d][i]] if d in self.geo_points else [] for d in range(self.n_docs)]
        return [[k[1] for k in self.doc_facets.get((field, d), ()) if k[0] == "n"] for d in range(self.n_docs)]

    def facet_strings(self, field):
        """-> (per document: ranks of its normalised strings, the field's distinct normalised strings in byte order —
        what facet_id_string_fst enumerates)"""
        values = sorted({k[1] for (f, k) in self.facet_docids if f == field and k[0] == "s"}, key=lambda v: v.encode())
        rank = {v: i for i, v in enumerate(values)}
        return [[rank[k[1]] for k in self.doc_facets.get((field, d), ()) if k[0] == "s"] for d in range(self.n_docs)], values

    def exists_docids(self, field):
        return {d for d, doc in enumerate(self.docs) if field in doc}

    def null_docids(self, field):
        return {d for d, doc in enumerate(self.docs) if field in doc and doc[field] is None}

    def empty_docids(self, field):
        return {d for d, doc in enumerate(self.docs) if field in doc and doc[field] in ("", [], {})}

    def distinct_values(self, field):
        """What the shim stages for msi_doc_values_create: per document the ids of its facet values of `field`
        (numbers and strings, field_id_docid_facet_f64s / _strings), and the number of distinct values."""
        ids, per_doc = {}, []
        for d in range(self.n_docs):
            per_doc.append([ids.setdefault(k, len(ids)) for k in self.doc_facets.get((field, d), ())])
        return per_doc, max(len(ids), 1)

    def distinct_excluded(self, field, docid):
        """distinct_single_docid (search/new/distinct.rs:38-62): the documents that share a facet value with docid."""
        out = set()
        for key in self.doc_facets.get((field, docid), ()):
            out |= self.facet_docids[(field, key)]
        return out

    def exact_words_with_prefix(self, prefix):
        return sorted((w for w in self.exact_word_docids if w.startswith(prefix)), key=lambda w: w.encode())

    def get_synonyms(self, words):
        return self.synonyms.get(tuple(words), [])

    def word_prefix_docids_values(self, pfx, original):
        vals = [self.word_prefix_docids.get(pfx)] + ([self.exact_word_prefix_docids.get(pfx)] if original else [])
        return [cbo_bytes(v) for v in vals if v]

    def word_prefix_fid_docids_values(self, pfx, fid):
        v = self.word_prefix_fid_docids.get((pfx, fid))
        return [cbo_bytes(v)] if v else []

    def word_prefix_position_docids_values(self, pfx, pos):
        v = self.word_prefix_position_docids.get((pfx, pos))
        return [cbo_bytes(v)] if v else []

    def word_prefix_pair_values(self, prox, w1, pfx2):
        return [cbo_bytes(s_) for (p_, a, b), s_ in sorted(self.pair.items())
                if p_ == prox and a == w1 and b.startswith(pfx2) and s_]


TOKEN_RE_CS = re.compile(r"[0-9a-zA-Zà-öø-ÿÀ-ÖØ-ß]+|[^0-9a-zA-Zà-öø-ÿÀ-ÖØ-ß]+")
TOKEN_RE = re.compile(r"[0-9a-zà-öø-ÿ]+|[^0-9a-zà-öø-ÿ]+")


def query_terms(query, words_limit=10, stop_words=()):
    """The located terms of located_query_terms_from_tokens (parse_query.rs:28-202) for the Latin subset of
    charabia: [(words, is_phrase, position_start, position_end, is_prefix)] — what the Rust shim hands to
    msi_keyword_search_ranked.  Negative operators are not modelled."""
    toks = TOKEN_RE_CS.findall(query)
    terms, phrase, position = [], None, -1

    def close(ph):
        if ph and any(w is not None for w, _ in ph):
            start = next(p_ for w, p_ in ph if w is not None)     # PhraseBuilder::push_word, parse_query.rs:318-335
            terms.append(([w for w, _ in ph], True, start, ph[-1][1], False))

    for k, tok in enumerate(toks):
        if len(terms) >= words_limit:
            break
        if WORD_RE.match(tok.lower()):
            position += 1
            stop = tok in stop_words
            tok = tok.lower()
            if phrase is not None:
                phrase.append((None if stop else tok, position))
            elif k == len(toks) - 1:
                terms.append(([tok], False, position, position, True))      # the last word is kept even if a stop word
            elif not stop:
                terms.append(([tok], False, position, position, False))
        else:
            if HARD_RE.search(tok):
                position += 7
                if phrase is not None:
                    close(phrase)
                    phrase = []
            q = tok.count('"')
            if q == 0:
                continue
            if phrase is not None:
                q -= 1
                close(phrase)
                phrase = None
            if q % 2 == 1:
                phrase = []
    if phrase is not None:
        close(phrase)
    return terms
