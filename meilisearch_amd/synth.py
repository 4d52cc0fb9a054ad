"""Seeded synthetic workloads (the reference's datasets are
remote URLs, so every config is restated synthetically with fixed seeds)."""
import numpy as np

# English unigram letter frequencies (a..z), BASELINE.md C3
_LETTER_P = np.array([8.17, 1.49, 2.78, 4.25, 12.70, 2.23, 2.02, 6.09, 6.97, 0.15, 0.77, 4.03, 2.41,
                      6.75, 7.51, 1.93, 0.10, 5.99, 6.33, 9.06, 2.76, 0.98, 2.36, 0.15, 1.97, 0.07])
_TWO_BYTE = [chr(c) for c in list(range(0xE0, 0xF7)) + list(range(0x430, 0x450))]  # à.. / а..я


def make_dictionary(n_words, seed=99, digits=0.08, two_byte=0.02, mean_len=8.0, sd_len=3.0,
                    max_len=40):
    """C3: unique sorted lowercase words, len ~ clamp(round(N(8,3)),1,40), letters by
    English unigram frequency + 8 % digits + 2 % 2-byte UTF-8.  Returns list[str]
    sorted by UTF-8 bytes."""
    rng = np.random.default_rng(seed)
    alphabet = [chr(ord("a") + i) for i in range(26)] + [str(i) for i in range(10)] + _TWO_BYTE
    p = np.concatenate([_LETTER_P / _LETTER_P.sum() * (1 - digits - two_byte),
                        np.full(10, digits / 10), np.full(len(_TWO_BYTE), two_byte / len(_TWO_BYTE))])
    words = set()
    while len(words) < n_words:
        need = int((n_words - len(words)) * 1.2) + 16
        lens = np.clip(np.rint(rng.normal(mean_len, sd_len, need)), 1, max_len).astype(np.int64)
        chars = rng.choice(len(alphabet), size=int(lens.sum()), p=p)
        pos = 0
        for L in lens:
            words.add("".join(alphabet[c] for c in chars[pos:pos + L]))
            pos += L
            if len(words) >= n_words:
                break
    return sorted(words, key=lambda w: w.encode("utf-8"))


def flatten_words(words):
    bs = [w.encode("utf-8") for w in words]
    concat = np.frombuffer(b"".join(bs), dtype=np.uint8).copy()
    off = np.zeros(len(bs) + 1, dtype=np.uint32)
    np.cumsum([len(b) for b in bs], out=off[1:])
    return concat, off


def edit_word(word, n_edits, rng, alphabet="abcdefghijklmnopqrstuvwxyz0123456789"):
    """Apply n random edits (insert / delete / substitute / transpose)."""
    w = list(word)
    for _ in range(n_edits):
        op = rng.integers(0, 4)
        if op == 0 or len(w) == 0:
            w.insert(int(rng.integers(0, len(w) + 1)), alphabet[int(rng.integers(0, len(alphabet)))])
        elif op == 1 and len(w) > 1:
            del w[int(rng.integers(0, len(w)))]
        elif op == 2:
            w[int(rng.integers(0, len(w)))] = alphabet[int(rng.integers(0, len(alphabet)))]
        elif len(w) > 1:
            i = int(rng.integers(0, len(w) - 1))
            w[i], w[i + 1] = w[i + 1], w[i]
    return "".join(w) or "a"


def make_typo_queries(words, n_queries, seed=7, prefix_frac=0.3, min_one=5, min_two=9):
    """C3: words sampled from the dictionary with e in {0,1,2} random edits, 30 %
    is_prefix, budgets from the char count (5 / 9).  Returns
    [(word, max_typos, is_prefix)] with budget-0 words dropped (they never reach
    the dictionary scan, compute_derivations.rs:21-37)."""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n_queries:
        w = words[int(rng.integers(0, len(words)))]
        q = edit_word(w, int(rng.integers(0, 3)), rng)
        n = len(q)
        budget = 0 if n < min_one else (1 if n < min_two else 2)
        if budget == 0 or len(q.encode("utf-8")) > 250:
            continue
        out.append((q, budget, bool(rng.random() < prefix_frac)))
    return out


def make_embeddings(n, d, seed=1234):
    """C2/C4: rows ~ N(0,1)^d, not normalised (numpy, host)."""
    rng = np.random.default_rng(seed)
    return rng.standard_normal((n, d), dtype=np.float32)


def round_to_bf16(x):
    """f32 array -> f32 array holding the nearest-even bf16 values (what a
    storage="bf16" store keeps; v_cvt_pk_bf16_f32 semantics for finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    bias = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
    return ((u + bias) & np.uint32(0xFFFF0000)).view(np.float32)


def roaring_serialize(ids):
    """RoaringBitmap::serialize_into of a sorted unique u32 array (`roaring` 0.10, the portable RoaringFormatSpec
    without run containers): cookie 12346, container count, (key, cardinality - 1) pairs, byte offset of every
    container, then array containers (<= 4096 values) or 8 KiB bitmap containers.  Pinned byte for byte by
    main["documents-ids"] of the reference's own index (tests/golden/index_blobs.json)."""
    import struct
    ids = np.asarray(ids, dtype=np.uint32)
    hi = (ids >> 16).astype(np.uint32)
    keys, starts = np.unique(hi, return_index=True)
    ends = np.append(starts[1:], ids.size)
    out = bytearray(struct.pack("<II", 12346, len(keys)))
    for k, a, b in zip(keys, starts, ends):
        out += struct.pack("<HH", int(k), int(b - a) - 1)
    at = 8 + 8 * len(keys)
    for a, b in zip(starts, ends):
        out += struct.pack("<I", at)
        at += 2 * int(b - a) if b - a <= 4096 else 8192
    for a, b in zip(starts, ends):
        v = (ids[a:b] & 0xFFFF).astype(np.uint16)
        if v.size <= 4096:
            out += v.astype("<u2").tobytes()
        else:
            words = np.zeros(1024, dtype=np.uint64)
            np.bitwise_or.at(words, (v >> 6).astype(np.int64), np.uint64(1) << (v & 63).astype(np.uint64))
            out += words.astype("<u8").tobytes()
    return bytes(out)


def cbo_serialize(ids):
    """CboRoaringBitmapCodec::serialize_into_writer (cbo_roaring_bitmap_codec.rs:33-51) of a sorted unique
    u32 array: <= 7 documents as raw native-endian u32s, else the portable Roaring serialisation."""
    ids = np.asarray(ids, dtype=np.uint32)
    if ids.size <= 7:
        return ids.astype("=u4").tobytes()
    return roaring_serialize(ids)


class SynthIndex:
    """A synthetic inverted index at scale behind msi_index_vtable (what the Rust shim would answer from
    LMDB): Zipf word frequencies over `n_docs` documents, 3 searchable fields, bucketed positions, word
    pairs at proximities 1..3.  Postings are generated lazily from a per-key seed and cached as the
    CboRoaringBitmap bytes milli stores."""
    FIDS = (1, 2, 3)
    POSITIONS = tuple(range(16)) + (24, 32, 64, 128)

    def __init__(self, n_docs, words, seed=4242):
        self.n_docs, self.seed = n_docs, seed
        self.words = sorted(set(words), key=lambda w: w.encode())
        self.rank = {w: i for i, w in enumerate(np.random.default_rng(seed).permutation(self.words))}
        self.exact_words = ()
        self._ids, self._bytes = {}, {}
        self.searchable_fids = list(self.FIDS)
        self.weights = {1: 0, 2: 1, 3: 2}
        self.max_weight = 2

    def _rng(self, *key):
        import zlib
        return np.random.default_rng(zlib.crc32(repr((self.seed,) + key).encode()))

    def ids(self, word):
        if word not in self._ids:
            if word not in self.rank:
                self._ids[word] = None
            else:
                p = min(0.4, 0.6 / (1 + self.rank[word]) ** 0.9)       # Zipf-like document frequency
                k = max(1, int(self.n_docs * p))
                self._ids[word] = np.unique(self._rng("w", word).integers(0, self.n_docs, k, dtype=np.uint32))
        return self._ids[word]

    def _cached(self, key, make):
        if key not in self._bytes:
            ids = make()
            self._bytes[key] = cbo_serialize(ids) if ids is not None and ids.size else None
        return self._bytes[key]

    def _part(self, word, salt, n_parts):
        ids = self.ids(word)
        h = (ids.astype(np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(salt * 0x632BE5AB + 1)) >> np.uint64(40)
        return ids, (h % np.uint64(n_parts)).astype(np.int64)

    def word_docids_bytes(self, word, original):
        return self._cached(("w", word), lambda: self.ids(word))

    def is_exact_word(self, word):
        return False

    def word_fid_docids_bytes(self, word, fid):
        def make():
            if self.ids(word) is None:
                return None
            ids, part = self._part(word, 1, 4)          # a document holds the word in 1-2 fields
            f = self.FIDS.index(fid)
            return ids[(part == f) | (part == 3) & (f < 2)]
        return self._cached(("f", word, fid), make) if fid in self.FIDS else None

    def word_position_docids_bytes(self, word, pos):
        def make():
            if self.ids(word) is None:
                return None
            ids, part = self._part(word, 2, len(self.POSITIONS))
            return ids[part == self.POSITIONS.index(pos)]
        return self._cached(("p", word, pos), make) if pos in self.POSITIONS else None

    def word_fids(self, word):
        return list(self.FIDS) if self.ids(word) is not None else []

    def word_positions(self, word):
        return list(self.POSITIONS) if self.ids(word) is not None else []

    def pair_docids_bytes(self, prox, left, right):
        def make():
            a, b = self.ids(left), self.ids(right)
            if a is None or b is None or not 1 <= prox <= 3:
                return None
            both = np.intersect1d(a, b, assume_unique=True)
            h = (both.astype(np.uint64) * np.uint64(0xD6E8FEB86659FD93) >> np.uint64(37)) % np.uint64(6)
            return both[h == np.uint64(prox - 1)]          # half of the co-occurrences are close
        return self._cached(("pp", prox, left, right), make)

    def fid_word_count_docids_bytes(self, fid, count):
        def make():
            k = max(1, self.n_docs // 200)
            return np.unique(self._rng("c", fid, count).integers(0, self.n_docs, k, dtype=np.uint32))
        return self._cached(("c", fid, count), make) if count <= 30 else None


def device_rows(n, d, dev, seed=1234, chunk=1_000_000):
    """C2 / C4 / C5: the N x d f32 row matrix ~ N(0,1), generated IN HBM (torch generator, fixed seed, 1 M-row
    chunks so that the stream of random numbers does not depend on how much is drawn per call).  bench.py and
    tests/test_configs_gpu.py build their stores from this one function."""
    import torch
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    rows_t = torch.empty((n, d), dtype=torch.float32, device=dev)
    for c0 in range(0, n, chunk):
        rows_t[c0:min(n, c0 + chunk)].normal_(generator=gen)
    return rows_t


def device_rows_chunks(n, d, dev, seed=1234, chunk=1_000_000):
    """The rows of device_rows(n, d, dev, seed) AGAIN, one chunk at a time: (c0, c1, rows[c0:c1]) from the same generator, the
    same draws in the same order — for a full-size parity check that does not keep the 30 GB of C4's rows in HBM beside the
    store built from them (bench.py frees them after the upload)."""
    import torch
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        yield c0, c1, torch.empty((c1 - c0, d), dtype=torch.float32, device=dev).normal_(generator=gen)


def device_queries(nq, d, dev, seed=5678):
    import torch
    gq = torch.Generator(device=dev)
    gq.manual_seed(seed)
    return torch.empty((nq, d), dtype=torch.float32, device=dev).normal_(generator=gq)


def device_rows_clustered(n, d, dev, seed=4321, n_centres=10_000, spread=(1e-3, 1e-2), dup_frac=0.01, chunk=1_000_000):
    """Rows shaped like an embedding corpus instead of i.i.d. noise: `n_centres` centres ~ N(0,1); every row is
    its centre plus noise whose size gives the row a cosine distance to the centre of 1 - cos in `spread` (log-uniform: tight
    and loose members in every cluster), and `dup_frac` of the rows are exact copies of another row.  A cluster holds
    n / n_centres rows within ~2 x spread of each other — thousands inside the bf16x2 proof's margin at 10 M rows.
    Generated in HBM, 1 M-row chunks, fixed seed.  Returns (rows [n, d] f32, centre index of every row [n] int64)."""
    import torch
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    centres = torch.empty((n_centres, d), dtype=torch.float32, device=dev).normal_(generator=gen)
    rows_t = torch.empty((n, d), dtype=torch.float32, device=dev)
    which = torch.empty(n, dtype=torch.int64, device=dev)
    lo, hi = float(np.log(spread[0])), float(np.log(spread[1]))
    for c0 in range(0, n, chunk):
        c1 = min(n, c0 + chunk)
        m = c1 - c0
        idx = torch.randint(0, n_centres, (m,), generator=gen, device=dev)
        which[c0:c1] = idx
        one_minus_cos = torch.exp(torch.empty(m, device=dev).uniform_(lo, hi, generator=gen))
        # |c| ~ sqrt(d); c + s * eps with eps ~ N(0, I): 1 - cos ~ s^2 / 2  (for s << 1, relative to |c|^2 / d = 1)
        sigma = torch.sqrt(2.0 * one_minus_cos)
        noise = torch.empty((m, d), dtype=torch.float32, device=dev).normal_(generator=gen)
        rows_t[c0:c1] = centres[idx] + sigma[:, None] * noise
        del noise
    n_dup = int(n * dup_frac)
    if n_dup:
        dst = torch.randint(0, n, (n_dup,), generator=gen, device=dev)
        src = torch.randint(0, n, (n_dup,), generator=gen, device=dev)
        rows_t[dst] = rows_t[src]
        which[dst] = which[src]
    return rows_t, which


def device_queries_near_rows(rows_t, nq, seed=8765, one_minus_cos=5e-3):
    """Queries that look like the corpus: a stored row each, moved by noise of about `one_minus_cos` — their neighbours are the
    row's cluster, whose members' scores differ in the fourth decimal."""
    import torch
    dev = rows_t.device
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    idx = torch.randint(0, rows_t.shape[0], (nq,), generator=gen, device=dev)
    base = rows_t[idx]
    noise = torch.empty_like(base).normal_(generator=gen)
    scale = base.norm(dim=1, keepdim=True) / float(np.sqrt(rows_t.shape[1]))
    return base + float(np.sqrt(2.0 * one_minus_cos)) * scale * noise


def random_bitset_words(n_bits, density, seed):
    """Dense candidate bitset (u64 words, LSB first) with Bernoulli(density) bits below n_bits (C5 filters, seed 31)."""
    rng = np.random.default_rng(seed)
    words = (n_bits + 63) // 64
    bits = rng.random(words * 64) < density
    bits[n_bits:] = False
    return np.packbits(bits, bitorder="little").view(np.uint64)
