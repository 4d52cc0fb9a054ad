// msi_arena.h — the host memory of ONE msi_keyword_search_ranked call.
//
// A detailed three-term search makes ~6 700 small allocations (paths, edge lists, set handles, per-bucket graphs): with
// glibc's allocator that was a quarter of the search's host CPU (profiles/r3_ranked_cpu_profile_after.txt), and the host
// CPU is what bounds the keyword leg.  Everything a search allocates dies with it, on the thread that called it (its
// tasks are fibers of that thread), so: a per-thread region handed out by bumping a pointer, size-class free lists for
// what is released meanwhile, everything forgotten at once when the call returns.
//
//   ArenaScope scope;                 // outermost scope on this thread owns the reset
//   Vec<int> v;  Map<K, V> m;         // containers whose allocator is the thread's arena
//
// Rules: a container of these types must not outlive the scope it allocated in (msi_search.hip: all of them live inside
// `Ctx` or below); memory obtained outside any scope comes from malloc and is recognised on release.  MSI_ARENA_POISON=1
// fills released and forgotten memory with 0xDD (tests).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <set>
#include <vector>

namespace msi_arena {

struct Arena {
  static constexpr size_t CHUNK = 2u << 20, BIG = 128u << 10, N_SMALL = 64, N_POW = 8;
  struct Chunk { char *base; size_t size; };
  std::vector<Chunk> chunks;
  size_t ci = 0;
  char *cur = nullptr, *end = nullptr;
  int depth = 0;
  bool poison = false;
  long live = 0;      // blocks handed out and not yet given back (checked at reset under MSI_ARENA_POISON)
  // free lists: 16-byte classes up to 1 KiB, then powers of two up to BIG
  void *small_free[N_SMALL] = {};
  void *pow_free[N_POW] = {};
  Arena() { poison = getenv("MSI_ARENA_POISON") != nullptr; }
  ~Arena() { for (auto &c : chunks) free(c.base); }
  static size_t pow_class(size_t n) {          // n in (1 KiB, BIG] -> 0..N_POW-1 (2 KiB, 4 KiB, ... 256 KiB)
    size_t k = 0, cap = 2048;
    while (cap < n) { cap <<= 1; ++k; }
    return k;
  }
  bool owns(const void *p) const {
    for (const auto &c : chunks)
      if ((const char *)p >= c.base && (const char *)p < c.base + c.size) return true;
    return false;
  }
  void next_chunk(size_t need) {
    while (ci + 1 < chunks.size()) {
      ++ci;
      if (chunks[ci].size >= need) { cur = chunks[ci].base; end = cur + chunks[ci].size; return; }
    }
    const size_t size = need > CHUNK ? need : CHUNK;
    char *b = (char *)malloc(size);
    if (!b) throw std::bad_alloc();
    chunks.push_back(Chunk{b, size});
    ci = chunks.size() - 1;
    cur = b;
    end = b + size;
  }
  void *bump(size_t n) {
    if (!cur || (size_t)(end - cur) < n) next_chunk(n);
    void *p = cur;
    cur += n;
    return p;
  }
  void *take(size_t n) {
    ++live;
    if (n <= 1024) {
      const size_t cls = n ? (n - 1) >> 4 : 0;
      if (void *p = small_free[cls]) { small_free[cls] = *(void **)p; return p; }
      return bump((cls + 1) << 4);
    }
    const size_t k = pow_class(n);
    if (void *p = pow_free[k]) { pow_free[k] = *(void **)p; return p; }
    return bump((size_t)2048 << k);
  }
  void give(void *p, size_t n) {
    --live;
    if (poison) memset(p, 0xDD, n);
    if (n <= 1024) {
      const size_t cls = n ? (n - 1) >> 4 : 0;
      *(void **)p = small_free[cls];
      small_free[cls] = p;
    } else {
      const size_t k = pow_class(n);
      *(void **)p = pow_free[k];
      pow_free[k] = p;
    }
  }
  void reset() {
    if (poison) {
      if (live != 0) {   // a container of the arena's types outlived its search (or was released twice)
        fprintf(stderr, "msi_arena: %ld blocks still live at the end of the scope\n", live);
        abort();
      }
      for (auto &c : chunks) memset(c.base, 0xDD, c.size);
    }
    live = 0;
    // a search that needed more than two chunks was an outlier: give the rest back
    while (chunks.size() > 2) { free(chunks.back().base); chunks.pop_back(); }
    ci = 0;
    cur = chunks.empty() ? nullptr : chunks[0].base;
    end = chunks.empty() ? nullptr : cur + chunks[0].size;
    memset(small_free, 0, sizeof small_free);
    memset(pow_free, 0, sizeof pow_free);
  }
};

inline Arena &arena() {
  thread_local Arena a;
  return a;
}

struct ArenaScope {
  Arena &a;
  ArenaScope() : a(arena()) { ++a.depth; }
  ~ArenaScope() { if (--a.depth == 0) a.reset(); }
  ArenaScope(const ArenaScope &) = delete;
  ArenaScope &operator=(const ArenaScope &) = delete;
};

template <class T>
struct Alloc {
  using value_type = T;
  Alloc() noexcept = default;
  template <class U> Alloc(const Alloc<U> &) noexcept {}
  T *allocate(size_t n) {
    static_assert(alignof(T) <= 16, "the arena hands out 16-byte aligned blocks");
    const size_t bytes = n * sizeof(T);
    Arena &a = arena();
    if (a.depth && bytes <= Arena::BIG) return (T *)a.take(bytes);
    void *p = malloc(bytes ? bytes : 1);
    if (!p) throw std::bad_alloc();
    return (T *)p;
  }
  void deallocate(T *p, size_t n) noexcept {
    const size_t bytes = n * sizeof(T);
    Arena &a = arena();
    if (bytes <= Arena::BIG && a.owns(p)) { a.give(p, bytes); return; }
    free(p);
  }
  template <class U> bool operator==(const Alloc<U> &) const noexcept { return true; }
  template <class U> bool operator!=(const Alloc<U> &) const noexcept { return false; }
};

template <class T> using Vec = std::vector<T, Alloc<T>>;
template <class K, class V, class C = std::less<K>> using Map = std::map<K, V, C, Alloc<std::pair<const K, V>>>;
template <class K, class C = std::less<K>> using OrdSet = std::set<K, C, Alloc<K>>;
template <class T, class... A> std::shared_ptr<T> make_shared(A &&...args) {
  return std::allocate_shared<T>(Alloc<T>(), std::forward<A>(args)...);
}

}  // namespace msi_arena
