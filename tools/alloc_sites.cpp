// tools/alloc_sites.cpp — LD_PRELOAD shim that counts C++ allocations and samples their call stacks (every
// ALLOC_SITES_EVERY-th `operator new`, default 53), written in the dump format of tools/r3_symbolize.py ("M" = a line of
// /proc/self/maps, "S" = return addresses).  Used to find where the keyword search's host logic allocates:
//   g++ -O2 -fPIC -shared -std=c++17 tools/alloc_sites.cpp -o /tmp/alloc_sites.so
//   ALLOC_SITES_OUT=/tmp/sites.txt LD_PRELOAD=/tmp/alloc_sites.so <command>; python tools/r3_symbolize.py /tmp/sites.txt
// ALLOC_SITES_ON is read through alloc_sites_enable(): samples are only taken between enable(1) and enable(0).
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <execinfo.h>
#include <new>

namespace {
constexpr int DEPTH = 14, MAX_SAMPLES = 200000;
void *g_pc[MAX_SAMPLES][DEPTH];
int g_n[MAX_SAMPLES];
unsigned g_size[MAX_SAMPLES];
std::atomic<int> g_count{0};
std::atomic<unsigned long> g_calls{0}, g_bytes{0};
std::atomic<int> g_on{0};
int g_every = 53;
thread_local unsigned tl_tick = 0;
thread_local bool tl_inside = false;

void dump() {
  const char *path = getenv("ALLOC_SITES_OUT");
  if (!path) path = "/tmp/alloc_sites.txt";
  FILE *f = fopen(path, "w");
  if (!f) return;
  FILE *m = fopen("/proc/self/maps", "r");
  char line[1024];
  while (m && fgets(line, sizeof line, m))
    if (strstr(line, " r-xp ") || strstr(line, " r--p ")) fprintf(f, "M %s", line);
  if (m) fclose(m);
  int n = g_count.load();
  if (n > MAX_SAMPLES) n = MAX_SAMPLES;
  for (int i = 0; i < n; ++i) {
    fprintf(f, "S");
    for (int k = 1; k < g_n[i]; ++k) fprintf(f, " %lx", (unsigned long) g_pc[i][k]);   // [0] is this shim
    fprintf(f, "\nZ %u\n", g_size[i]);
  }
  fprintf(f, "# calls %lu bytes %lu sampled every %d\n", g_calls.load(), g_bytes.load(), g_every);
  fclose(f);
}

struct Init {
  Init() {
    if (const char *e = getenv("ALLOC_SITES_EVERY")) g_every = atoi(e);
    void *warm[4];
    backtrace(warm, 4);
    atexit(dump);
  }
} g_init;

inline void note(size_t n) {
  if (!g_on.load(std::memory_order_relaxed) || tl_inside) return;
  g_calls.fetch_add(1, std::memory_order_relaxed);
  g_bytes.fetch_add(n, std::memory_order_relaxed);
  if (++tl_tick % (unsigned) g_every) return;
  tl_inside = true;
  int i = g_count.fetch_add(1);
  if (i < MAX_SAMPLES) {
    g_n[i] = backtrace(g_pc[i], DEPTH);
    g_size[i] = (unsigned) n;
  }
  tl_inside = false;
}
}  // namespace

extern "C" void alloc_sites_enable(int on) { g_on.store(on); }
extern "C" unsigned long alloc_sites_calls() { return g_calls.load(); }
extern "C" unsigned long alloc_sites_bytes() { return g_bytes.load(); }

void *operator new(size_t n) {
  note(n);
  void *p = malloc(n ? n : 1);
  if (!p) throw std::bad_alloc();
  return p;
}
void *operator new[](size_t n) { return operator new(n); }
void operator delete(void *p) noexcept { free(p); }
void operator delete[](void *p) noexcept { free(p); }
void operator delete(void *p, size_t) noexcept { free(p); }
void operator delete[](void *p, size_t) noexcept { free(p); }
