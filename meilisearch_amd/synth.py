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
