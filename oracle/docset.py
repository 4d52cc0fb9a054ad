"""TEST INFRASTRUCTURE — not part of the product, never imported by meilisearch_amd.

A docid-set type with the interface of the built-in `set` that oracle/ranking_oracle.py uses (`&`, `|`, `-`, their
in-place forms, `len`, truth, ascending iteration, `in`, `<=`, `==`, `add`, `discard`, `update`, `difference_update`),
so that the SAME oracle code that is pinned to the reference's snapshots on toy corpora also runs over the 10 M-document
index of the headline configuration, where Python sets of millions of ints would take minutes per query.

Representation (a small Roaring in numpy): either a sorted unique uint32 array (`ids`) or a dense array of 64-bit words
(`bits`, bit d of word d >> 6 = document d).  A set turns dense when it grows past `DENSE_ABOVE` members and turns
sparse again when an intersection / difference leaves few.  Arrays are shared copy-on-write between copies.

`ranking_oracle.DocSet` is `set` by default; `docset_type(n_docs)` makes the class to put there.
tests/test_docset_cpu.py holds it equal to `set` operation by operation and replays the reference's snapshot searches
through the oracle in this mode."""
import numpy as np

_ONE = np.uint64(1)


def docset_type(n_docs, dense_above=4096, sparse_below=2048):
    words = (int(n_docs) + 63) // 64

    class DocSet:
        __slots__ = ("ids", "bits", "own")
        N_DOCS, WORDS, DENSE_ABOVE, SPARSE_BELOW = int(n_docs), words, dense_above, sparse_below

        # ---- construction ---------------------------------------------------------------------------
        def __init__(self, it=None):
            self.bits, self.own = None, True
            if it is None:
                self.ids = np.empty(0, np.uint32)
            elif isinstance(it, DocSet):
                self.ids, self.bits = it.ids, it.bits
                self.own = it.own = False                 # shared until one of the two writes
            elif isinstance(it, np.ndarray):
                self.ids = np.unique(it.astype(np.uint32))
                self._maybe_dense()
            else:
                self.ids = np.unique(np.fromiter(it, dtype=np.uint32))
                self._maybe_dense()

        @classmethod
        def from_sorted(cls, ids):
            """ids: sorted unique uint32 array (not copied)."""
            s = cls()
            s.ids = ids
            s._maybe_dense()
            return s

        @classmethod
        def from_words(cls, bits):
            s = cls()
            s.ids, s.bits = None, bits
            return s

        @classmethod
        def full(cls):
            b = np.full(words, ~np.uint64(0), np.uint64)
            if cls.N_DOCS & 63:
                b[-1] = (_ONE << np.uint64(cls.N_DOCS & 63)) - _ONE
            return cls.from_words(b)

        # ---- representation -------------------------------------------------------------------------
        @staticmethod
        def _to_bits(ids):
            b = np.zeros(words, np.uint64)
            if ids.size:
                np.bitwise_or.at(b, ids >> 6, _ONE << (ids & 63).astype(np.uint64))
            return b

        @staticmethod
        def _to_ids(bits):
            return np.flatnonzero(np.unpackbits(bits.view(np.uint8), bitorder="little")).astype(np.uint32)

        def _maybe_dense(self):
            if self.ids is not None and self.ids.size > self.DENSE_ABOVE:
                self.bits, self.ids, self.own = self._to_bits(self.ids), None, True

        def _maybe_sparse(self):
            if self.bits is not None and int(np.bitwise_count(self.bits).sum()) < self.SPARSE_BELOW:
                self.ids, self.bits, self.own = self._to_ids(self.bits), None, True

        @staticmethod
        def _test(bits, ids):
            """mask of the ids present in bits"""
            return ((bits[ids >> 6] >> (ids & 63).astype(np.uint64)) & _ONE).astype(bool)

        @classmethod
        def _coerce(cls, o):
            return o if isinstance(o, DocSet) else cls(o)

        # ---- queries --------------------------------------------------------------------------------
        def __len__(self):
            return int(self.ids.size) if self.bits is None else int(np.bitwise_count(self.bits).sum())

        def __bool__(self):
            return bool(self.ids.size) if self.bits is None else bool(self.bits.any())

        def to_array(self):
            return self.ids if self.bits is None else self._to_ids(self.bits)

        def __iter__(self):
            return iter(self.to_array().tolist())

        def __contains__(self, d):
            d = int(d)
            if d < 0 or d >= self.N_DOCS:
                return False
            if self.bits is None:
                i = int(np.searchsorted(self.ids, d))
                return i < self.ids.size and int(self.ids[i]) == d
            return bool((int(self.bits[d >> 6]) >> (d & 63)) & 1)

        def __eq__(self, o):
            if not isinstance(o, DocSet):
                o = self._coerce(o)
            return np.array_equal(self.to_array(), o.to_array())

        __hash__ = None

        def __le__(self, o):
            return not (self - self._coerce(o))

        def __ge__(self, o):
            return not (self._coerce(o) - self)

        def __repr__(self):
            a = self.to_array()
            return "DocSet(%d: %s%s)" % (a.size, a[:8].tolist(), "…" if a.size > 8 else "")

        # ---- algebra (new objects) ------------------------------------------------------------------
        def __and__(self, o):
            o = self._coerce(o)
            if self.bits is None and o.bits is None:
                return DocSet.from_sorted(np.intersect1d(self.ids, o.ids, assume_unique=True))
            if self.bits is None:
                return DocSet.from_sorted(self.ids[self._test(o.bits, self.ids)])
            if o.bits is None:
                return DocSet.from_sorted(o.ids[self._test(self.bits, o.ids)])
            r = DocSet.from_words(self.bits & o.bits)
            r._maybe_sparse()
            return r

        __rand__ = __and__

        def __or__(self, o):
            o = self._coerce(o)
            if self.bits is None and o.bits is None:
                return DocSet.from_sorted(np.union1d(self.ids, o.ids))
            if self.bits is None:
                self, o = o, self
            if o.bits is None:
                b = self.bits.copy()
                if o.ids.size:
                    np.bitwise_or.at(b, o.ids >> 6, _ONE << (o.ids & 63).astype(np.uint64))
                return DocSet.from_words(b)
            return DocSet.from_words(self.bits | o.bits)

        __ror__ = __or__

        def __sub__(self, o):
            o = self._coerce(o)
            if self.bits is None:
                if o.bits is None:
                    return DocSet.from_sorted(np.setdiff1d(self.ids, o.ids, assume_unique=True))
                return DocSet.from_sorted(self.ids[~self._test(o.bits, self.ids)])
            if o.bits is None:
                b = self.bits.copy()
                if o.ids.size:
                    np.bitwise_and.at(b, o.ids >> 6, ~(_ONE << (o.ids & 63).astype(np.uint64)))
                r = DocSet.from_words(b)
            else:
                r = DocSet.from_words(self.bits & ~o.bits)
            r._maybe_sparse()
            return r

        def __rsub__(self, o):
            return self._coerce(o) - self

        # ---- in-place forms (same aliasing as set: the object is modified, copies are not) -----------
        def _take(self, r):
            self.ids, self.bits, self.own = r.ids, r.bits, True
            return self

        def __iand__(self, o):
            return self._take(self & o)

        def __ior__(self, o):
            o = self._coerce(o)
            if self.bits is not None and o.bits is None and self.own:      # the common case: scatter a few ids
                if o.ids.size:
                    np.bitwise_or.at(self.bits, o.ids >> 6, _ONE << (o.ids & 63).astype(np.uint64))
                return self
            if self.bits is not None and o.bits is not None and self.own:
                np.bitwise_or(self.bits, o.bits, out=self.bits)
                return self
            return self._take(self | o)

        def __isub__(self, o):
            return self._take(self - o)

        def update(self, o):
            self.__ior__(o)

        def difference_update(self, o):
            self.__isub__(o)

        def intersection_update(self, o):
            self.__iand__(o)

        def add(self, d):
            self.__ior__(DocSet.from_sorted(np.array([d], np.uint32)))

        def discard(self, d):
            self.__isub__(DocSet.from_sorted(np.array([d], np.uint32)))

    DocSet.__qualname__ = DocSet.__name__ = "DocSet"
    return DocSet


def decode_cbo(data):
    """CboRoaringBitmapCodec::deserialize_from (heed_codec/roaring_bitmap/cbo_roaring_bitmap_codec.rs:53-69): at most
    7 ids -> raw native-endian u32s; else the portable Roaring format (cookie 12346: array / bitmap containers with an
    offset header; 12347: run flags, offsets only from 4 containers on).  -> sorted uint32 array."""
    if len(data) <= 7 * 4:
        return np.frombuffer(data, dtype="<u4").astype(np.uint32)
    return decode_roaring(data)


def decode_roaring(data):
    """RoaringBitmap::deserialize_from of `roaring` 0.10 (third party; the portable format of the RoaringFormatSpec)."""
    buf = np.frombuffer(data, dtype=np.uint8)
    cookie = int(np.frombuffer(data[:4], "<u4")[0])
    pos = 4
    if cookie & 0xFFFF == 12347:
        n = (cookie >> 16) + 1
        run_flags = buf[pos:pos + (n + 7) // 8]
        pos += (n + 7) // 8
        has_offsets = n >= 4
    elif cookie == 12346:
        n = int(np.frombuffer(data[4:8], "<u4")[0])
        pos = 8
        run_flags = None
        has_offsets = True
    else:
        raise ValueError("not a Roaring bitmap")
    desc = np.frombuffer(data[pos:pos + 4 * n], "<u2").reshape(n, 2)
    pos += 4 * n
    if has_offsets:
        pos += 4 * n
    out = []
    for i in range(n):
        key, card = int(desc[i, 0]) << 16, int(desc[i, 1]) + 1
        is_run = run_flags is not None and (int(run_flags[i >> 3]) >> (i & 7)) & 1
        if is_run:
            n_runs = int(np.frombuffer(data[pos:pos + 2], "<u2")[0])
            runs = np.frombuffer(data[pos + 2:pos + 2 + 4 * n_runs], "<u2").reshape(n_runs, 2).astype(np.uint32)
            pos += 2 + 4 * n_runs
            out.append(np.concatenate([np.arange(s, s + l + 1, dtype=np.uint32) for s, l in runs]) + np.uint32(key))
        elif card > 4096:
            bm = np.frombuffer(data[pos:pos + 8192], np.uint8)
            pos += 8192
            out.append(np.flatnonzero(np.unpackbits(bm, bitorder="little")).astype(np.uint32) + np.uint32(key))
        else:
            out.append(np.frombuffer(data[pos:pos + 2 * card], "<u2").astype(np.uint32) + np.uint32(key))
            pos += 2 * card
    return np.concatenate(out) if out else np.empty(0, np.uint32)
