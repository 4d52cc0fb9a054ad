// fst::Set bytes -> the flat, byte-lexicographic word list the device dictionary is built from (host code; no
// kernel here — the file is a .hip only so that the one Makefile rule builds it).
//
// milli keeps main["words-fst"] (crates/milli/src/index.rs:1225-1243), the exact-word / stop-word sets and one FST
// per faceted field (facet-id-string-fst, search/facet/search.rs:122-190) as `fst` 0.4.7 blobs.  SURVEY §8 f2: with
// this decoder the shim hands over `index.words_fst(rtxn)?.as_fst().as_bytes()` (a borrow of the LMDB page) and no
// Rust-side `stream()` + copy of two million strings is needed.  The format (version 3) as published by that
// crate (raw/node.rs, raw/build.rs):
//   header  u64 version, u64 type
//   nodes   children first; a node's address is the position of its LAST byte (the state byte), it is read
//           backwards; address 0 = the final node without transitions (never written)
//             11cccccc one transition, not final, to the node written just before
//             10cccccc one transition, not final: [output][delta][sizes][input?][state]
//             0fnnnnnn any: [final output][outputs][deltas][inputs][256-byte index if n > 32][sizes][n?][state]
//           cccccc = 1 + index into the common-input table (0: the input byte precedes the state byte);
//           sizes = bytes per delta << 4 | bytes per output; delta = own first byte - target (0 = address 0)
//   footer  u64 number of keys, u64 root address, u32 masked CRC-32C of everything before it
// Every read is bounds-checked against the header, a transition must point below the node that holds it (so the
// walk terminates on any input), inputs must ascend, and the key count must match the footer: a blob that fails
// any of these is refused (MSI_E_INVALID) and the shim falls back to streaming the FST itself.
#include <string.h>

#include <vector>

#include "msi_common.h"

namespace {

constexpr uint64_t FST_VERSION = 3, HEADER = 16, FOOTER = 20;
constexpr uint32_t INDEX_THRESHOLD = 32;
// common_inputs.rs: bytes by descending frequency; a 6-bit field holds index + 1 of the first 63
const uint8_t COMMON_INPUTS_INV[] = "te/oasripcnw.hlm-du012g=:bf3y5&_4v9678k%?xCDASFIBEjPTzRNM+LOqHG";
static_assert(sizeof(COMMON_INPUTS_INV) - 1 == 63, "6-bit common-input field");

uint32_t crc32c_masked(const uint8_t *p, size_t n) {  // raw/crc32.rs: CRC-32C, then rotate_right(15) + 0xA282EAD8
  static uint32_t table[8][256];
  static const bool init = [] {
    for (uint32_t i = 0; i < 256; ++i) {
      uint32_t c = i;
      for (int k = 0; k < 8; ++k) c = (c >> 1) ^ ((c & 1) ? 0x82F63B78u : 0u);
      table[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
      for (int t = 1; t < 8; ++t) table[t][i] = (table[t - 1][i] >> 8) ^ table[0][table[t - 1][i] & 0xFF];
    return true;
  }();
  (void)init;
  uint32_t c = 0xFFFFFFFFu;
  while (n >= 8) {  // slicing-by-8
    uint64_t v;
    memcpy(&v, p, 8);
    const uint32_t lo = (uint32_t)v ^ c, hi = (uint32_t)(v >> 32);
    c = table[7][lo & 0xFF] ^ table[6][(lo >> 8) & 0xFF] ^ table[5][(lo >> 16) & 0xFF] ^ table[4][lo >> 24] ^
        table[3][hi & 0xFF] ^ table[2][(hi >> 8) & 0xFF] ^ table[1][(hi >> 16) & 0xFF] ^ table[0][hi >> 24];
    p += 8;
    n -= 8;
  }
  while (n--) c = table[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
  c ^= 0xFFFFFFFFu;
  return ((c >> 15) | (c << 17)) + 0xA282EAD8u;
}

struct Node {
  bool final_ = false;
  uint32_t n = 0;        // transitions
  uint64_t first = 0;    // lowest byte of the node
  uint64_t inputs = 0;   // any: position of input 0 (input i at inputs - i); one: unused
  uint64_t deltas = 0;   // position of delta 0 (delta i at deltas - i * tsize)
  uint32_t tsize = 0;
  uint8_t one_input = 0;
  uint64_t one_target = 0;
  bool one = false;
};

struct Reader {
  const uint8_t *d;
  uint64_t end;  // one past the last node byte (start of the footer)

  static uint64_t le(const uint8_t *p, uint32_t n) {
    uint64_t v = 0;
    for (uint32_t i = 0; i < n; ++i) v |= (uint64_t)p[i] << (8 * i);
    return v;
  }
  // false: malformed
  bool node(uint64_t addr, Node &o) const {
    o = Node();
    if (addr == 0) {
      o.final_ = true;
      return true;
    }
    if (addr < HEADER || addr >= end) return false;
    const uint8_t st = d[addr];
    if ((st >> 6) >= 2) {
      const uint32_t idx = st & 0x3F;
      const uint64_t in_len = idx ? 0 : 1;
      if (addr < HEADER + in_len) return false;
      o.one = true;
      o.n = 1;
      o.one_input = idx ? COMMON_INPUTS_INV[idx - 1] : d[addr - 1];
      if ((st >> 6) == 3) {
        o.first = addr - in_len;
        if (o.first < HEADER + 1) return false;
        o.one_target = o.first - 1;
        return o.one_target >= HEADER;
      }
      if (addr < HEADER + in_len + 1) return false;
      const uint8_t sizes = d[addr - in_len - 1];
      const uint32_t tsize = sizes >> 4, osize = sizes & 15;
      if (tsize < 1 || tsize > 8 || osize > 8) return false;
      if (addr < HEADER + in_len + 1 + tsize + osize) return false;
      const uint64_t at = addr - in_len - 1 - tsize;
      o.first = at - osize;
      const uint64_t delta = le(d + at, tsize);
      if (delta > o.first) return false;
      o.one_target = delta ? o.first - delta : 0;
      return true;
    }
    o.final_ = (st & 0x40) != 0;
    uint32_t n = st & 0x3F;
    uint64_t n_len = 0;
    if (n == 0) {
      n_len = 1;
      if (addr < HEADER + 1) return false;
      n = d[addr - 1];
      if (n == 1) n = 256;
    }
    if (addr < HEADER + n_len + 1) return false;
    const uint64_t base = addr - n_len - 1;
    const uint8_t sizes = d[base];
    const uint32_t tsize = sizes >> 4, osize = sizes & 15;
    if (tsize > 8 || osize > 8 || (n && !tsize)) return false;
    const uint64_t index = n > INDEX_THRESHOLD ? 256 : 0;
    const uint64_t body = index + n + (uint64_t)n * tsize + (uint64_t)n * osize + (o.final_ ? osize : 0);
    if (base < HEADER + body) return false;
    o.n = n;
    o.tsize = tsize;
    o.first = base - body;
    o.inputs = base - index - 1;
    o.deltas = base - index - n - tsize;
    return true;
  }
  uint8_t input(const Node &o, uint32_t i) const { return o.one ? o.one_input : d[o.inputs - i]; }
  // false: malformed (a transition that does not point below its node)
  bool target(const Node &o, uint32_t i, uint64_t &t) const {
    if (o.one) {
      t = o.one_target;
    } else {
      const uint64_t delta = le(d + o.deltas - (uint64_t)i * o.tsize, o.tsize);
      if (delta > o.first) return false;
      t = delta ? o.first - delta : 0;
    }
    return t == 0 || (t >= HEADER && t < o.first);
  }
};

struct Frame {
  Node node;
  uint32_t next;
};

// The walk itself; `emit(key, len)` returns a status.  Iterative DFS with an explicit stack (keys may be long).
template <class Emit>
int32_t walk_fst(const uint8_t *fst, size_t len, uint32_t flags, Emit &&emit_key, uint64_t *out_n_keys) {
  auto bad = [](const char *what) {
    msi_set_error("msi_fst_decode: malformed fst: %s", what);
    return (int32_t)MSI_E_INVALID;
  };
  if (len < HEADER + FOOTER) return bad("shorter than header + footer");
  uint64_t version, n_keys, root;
  memcpy(&version, fst, 8);
  if (version != FST_VERSION) {
    msi_set_error("msi_fst_decode: fst format version %llu is not supported (fst 0.4 writes version 3)",
                  (unsigned long long)version);
    return MSI_E_UNSUPPORTED;
  }
  memcpy(&n_keys, fst + len - 20, 8);
  memcpy(&root, fst + len - 12, 8);
  if (!(flags & MSI_FST_SKIP_CHECKSUM)) {
    uint32_t sum;
    memcpy(&sum, fst + len - 4, 4);
    if (sum != crc32c_masked(fst, len - 4)) return bad("checksum mismatch");
  }
  if (n_keys > 0xFFFFFFFEull) {
    msi_set_error("msi_fst_decode: %llu keys exceed the 32-bit dictionary index", (unsigned long long)n_keys);
    return MSI_E_UNSUPPORTED;
  }
  const Reader r{fst, (uint64_t)len - FOOTER};
  std::vector<Frame> stack;
  std::vector<uint8_t> key;
  uint64_t n_words = 0, n_bytes = 0;
  auto emit = [&]() -> int32_t {
    if (n_words >= n_keys) return bad("more keys than the footer declares");
    MSI_TRY(emit_key(key.data(), key.size(), n_words, n_bytes));
    ++n_words;
    n_bytes += key.size();
    if (n_bytes > 0xFFFFFFFFull) {
      msi_set_error("msi_fst_decode: more than 4 GiB of key bytes");
      return MSI_E_UNSUPPORTED;
    }
    return MSI_OK;
  };
  Frame f;
  if (!r.node(root, f.node)) return bad("root node");
  f.next = 0;
  if (f.node.final_) MSI_TRY(emit());
  stack.push_back(f);
  while (!stack.empty()) {
    Frame &top = stack.back();
    if (top.next == top.node.n) {
      stack.pop_back();
      if (!key.empty()) key.pop_back();
      continue;
    }
    const uint32_t i = top.next++;
    const uint8_t in = r.input(top.node, i);
    if (i && in <= r.input(top.node, i - 1)) return bad("transition inputs do not ascend");
    uint64_t t;
    if (!r.target(top.node, i, t)) return bad("a transition does not point below its node");
    Frame child;
    if (!r.node(t, child.node)) return bad("node out of bounds");
    child.next = 0;
    key.push_back(in);
    if (child.node.final_) MSI_TRY(emit());
    stack.push_back(child);  // invalidates `top`
  }
  if (n_words != n_keys) return bad("key count differs from the footer");
  *out_n_keys = n_words;
  return MSI_OK;
}

}  // namespace

extern "C" int32_t msi_fst_decode(const uint8_t *fst, size_t len, uint32_t flags, uint8_t *out_concat, uint64_t cap_bytes,
                                  uint32_t *out_offsets, uint32_t cap_words, uint32_t *out_n_words,
                                  uint64_t *out_n_bytes) {
  if (!fst || !out_n_words || !out_n_bytes || (out_concat && !out_offsets) || (flags & ~(uint32_t)MSI_FST_SKIP_CHECKSUM)) {
    msi_set_error("msi_fst_decode: invalid argument");
    return MSI_E_INVALID;
  }
  *out_n_words = 0;
  *out_n_bytes = 0;
  const bool write = out_concat != nullptr;
  uint64_t total = 0, n = 0;
  MSI_TRY(walk_fst(fst, len, flags, [&](const uint8_t *key, size_t klen, uint64_t idx, uint64_t at) -> int32_t {
    if (write) {
      if (idx >= cap_words || at + klen > cap_bytes) {
        msi_set_error("msi_fst_decode: output buffers too small (call with out_concat = NULL for the sizes)");
        return MSI_E_INVALID;
      }
      out_offsets[idx] = (uint32_t)at;
      if (klen) memcpy(out_concat + at, key, klen);
    }
    total = at + klen;
    return MSI_OK;
  }, &n));
  if (write) out_offsets[n] = (uint32_t)total;
  *out_n_words = (uint32_t)n;
  *out_n_bytes = total;
  return MSI_OK;
}

namespace {
int32_t dict_from_fst(msi_ctx *ctx, const uint8_t *fst, size_t len, msi_dict **out, bool values) {
  if (!ctx || !out || !fst) {
    msi_set_error("msi_dict_create_from_fst: invalid argument");
    return MSI_E_INVALID;
  }
  std::vector<uint8_t> concat;
  std::vector<uint32_t> offsets;
  concat.reserve(len * 3);  // shared prefixes and suffixes expand
  uint64_t n = 0;
  MSI_TRY(walk_fst(fst, len, 0, [&](const uint8_t *key, size_t klen, uint64_t, uint64_t at) -> int32_t {
    offsets.push_back((uint32_t)at);
    concat.insert(concat.end(), key, key + klen);
    return MSI_OK;
  }, &n));
  offsets.push_back((uint32_t)concat.size());
  if (concat.empty()) concat.push_back(0);
  return values ? msi_dict_create_values(ctx, concat.data(), offsets.data(), (uint32_t)n, out)
                : msi_dict_create(ctx, concat.data(), offsets.data(), (uint32_t)n, out);
}
}  // namespace

extern "C" int32_t msi_dict_create_from_fst(msi_ctx *ctx, const uint8_t *fst, size_t len, msi_dict **out) {
  return dict_from_fst(ctx, fst, len, out, false);
}

extern "C" int32_t msi_dict_create_values_from_fst(msi_ctx *ctx, const uint8_t *fst, size_t len, msi_dict **out) {
  return dict_from_fst(ctx, fst, len, out, true);
}
