// TEST INFRASTRUCTURE — the few RCCL entry points msi_group.hip resolves with dlsym, for the CPU-emulated build
// (tests/emu/run_emulated.py points MSI_RCCL_LIBRARY at the library built from this file).  "Devices" of the emulation
// share the host's memory, so a collective is memcpy between the ranks' buffers; what this checks is the HOST side of
// msi_group / msi_vs_group: one communicator per device, group start / end around the per-device calls, buffer sizes and
// offsets of the packed exchange, the per-rank form joining through a unique id from several threads.
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>

#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <vector>

namespace {

struct World {
  int n = 0;
  std::mutex mu;
  std::condition_variable cv;
  // one all-gather at a time per world: what each rank handed in
  std::vector<const void *> send;
  std::vector<void *> recv;
  std::vector<size_t> bytes;
  int arrived = 0;
  uint64_t generation = 0;
  int joined = 0;   // per-rank form: ranks that have called ncclCommInitRank
};
struct Comm {
  std::shared_ptr<World> w;
  int rank = 0;
};
std::mutex g_mu;
std::map<uint64_t, std::shared_ptr<World>> g_by_id;   // per-rank form: unique id -> world
uint64_t g_next_id = 1;
thread_local int t_group_depth = 0;
thread_local std::vector<Comm *> t_pending;   // all-gathers recorded between ncclGroupStart and ncclGroupEnd

size_t elem_size(int dtype) { return dtype == 0 || dtype == 1 ? 1 : (dtype == 2 || dtype == 3 || dtype == 7 ? 4 : 8); }

// every rank of the world has handed in its buffers: recv[r] := send[0] | send[1] | ... for every r
void exchange(World &w) {
  for (int r = 0; r < w.n; ++r)
    for (int j = 0; j < w.n; ++j) memcpy((char *)w.recv[r] + (size_t)j * w.bytes[j], w.send[j], w.bytes[j]);
}
int arrive(Comm *c, const void *send, void *recv, size_t bytes) {
  World &w = *c->w;
  std::unique_lock<std::mutex> lk(w.mu);
  w.send[c->rank] = send;
  w.recv[c->rank] = recv;
  w.bytes[c->rank] = bytes;
  const uint64_t gen = w.generation;
  if (++w.arrived == w.n) {
    for (int r = 1; r < w.n; ++r)
      if (w.bytes[r] != w.bytes[0]) return 5;   // ncclInvalidArgument
    exchange(w);
    w.arrived = 0;
    ++w.generation;
    w.cv.notify_all();
    return 0;
  }
  if (t_group_depth) return 0;   // inside a group the other ranks' calls follow on this same thread
  w.cv.wait(lk, [&] { return w.generation != gen; });
  return 0;
}

// ---- ranks in DIFFERENT processes (bench.py --gpus 2 under torch.distributed.run in the CPU tier: tests/
// test_bench_two_ranks_cpu.py) -----------------------------------------------------------------------------------------
// The unique id names a POSIX shared-memory segment (its creator's pid + counter); a rank whose process did not create the
// id joins through the segment: per rank a slot its all-gather input is copied into, a sense-reversing barrier on atomics
// in the segment before the slots are read and after (so that nobody rewrites its slot while a peer still reads it).
constexpr int SHM_MAX_RANKS = 8;
constexpr size_t SHM_SLOT = 8u << 20;
struct ShmWorld {
  std::atomic<uint32_t> magic, joined, arrived, generation, left;
  uint32_t n;
  uint64_t bytes[SHM_MAX_RANKS];
  alignas(64) uint8_t data[SHM_MAX_RANKS][SHM_SLOT];
};
struct ShmComm {
  ShmWorld *w = nullptr;
  int rank = 0;
  char name[96] = "";
};
void shm_name(char (&out)[96], const char *id) {
  uint64_t v, pid;
  memcpy(&v, id, 8);
  memcpy(&pid, id + 16, 8);
  snprintf(out, sizeof out, "/msi_rccl_emu_%llu_%llu", (unsigned long long)pid, (unsigned long long)v);
}
void shm_barrier(ShmWorld *w) {
  const uint32_t g = w->generation.load(std::memory_order_acquire);
  if (w->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == w->n) {
    w->arrived.store(0, std::memory_order_relaxed);
    w->generation.store(g + 1, std::memory_order_release);
  } else {
    while (w->generation.load(std::memory_order_acquire) == g) usleep(50);
  }
}
ShmComm *shm_join(const char *id, int world, int rank) {
  if (world > SHM_MAX_RANKS) return nullptr;
  ShmComm *c = new ShmComm();
  shm_name(c->name, id);
  bool creator = true;
  int fd = shm_open(c->name, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) {
    creator = false;
    fd = shm_open(c->name, O_RDWR, 0600);
  }
  if (fd < 0 || (creator && ftruncate(fd, sizeof(ShmWorld)) != 0)) {
    delete c;
    return nullptr;
  }
  if (!creator) {   // the creator may not have sized the segment yet
    struct stat st;
    for (int i = 0; i < 20000 && (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(ShmWorld)); ++i) usleep(100);
  }
  void *m = mmap(nullptr, sizeof(ShmWorld), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) {
    delete c;
    return nullptr;
  }
  c->w = (ShmWorld *)m;
  c->rank = rank;
  if (creator) {   // (a fresh segment is zero-filled)
    c->w->n = (uint32_t)world;
    c->w->magic.store(0x6D736931u, std::memory_order_release);
  } else {
    while (c->w->magic.load(std::memory_order_acquire) != 0x6D736931u) usleep(50);
  }
  c->w->joined.fetch_add(1, std::memory_order_acq_rel);
  while (c->w->joined.load(std::memory_order_acquire) < (uint32_t)world) usleep(50);   // as the real call: everybody has joined
  return c;
}
// a Comm handle is either an in-process Comm or (tagged by its first word being null) a ShmComm
struct AnyComm {
  Comm *inproc = nullptr;
  ShmComm *shm = nullptr;
};

}  // namespace

extern "C" {

struct NcclId { char internal[128]; };

int ncclCommInitAll(AnyComm **comms, int n, const int *devices) {
  if (!comms || n < 1) return 5;
  (void)devices;
  auto w = std::make_shared<World>();
  w->n = n;
  w->send.assign(n, nullptr);
  w->recv.assign(n, nullptr);
  w->bytes.assign(n, 0);
  for (int i = 0; i < n; ++i) {
    comms[i] = new AnyComm();
    comms[i]->inproc = new Comm();
    comms[i]->inproc->w = w;
    comms[i]->inproc->rank = i;
  }
  return 0;
}
int ncclGetUniqueId(NcclId *id) {
  std::lock_guard<std::mutex> lk(g_mu);
  memset(id->internal, 0, sizeof id->internal);
  const uint64_t v = g_next_id++, pid = (uint64_t)getpid();
  memcpy(id->internal, &v, sizeof v);
  memcpy(id->internal + 8, "rccl-emu", 8);
  memcpy(id->internal + 16, &pid, sizeof pid);   // (+ the creating process: ranks of other processes join through shared memory)
  return 0;
}
int ncclCommInitRank(AnyComm **comm, int world, NcclId id, int rank) {
  if (!comm || world < 1 || rank < 0 || rank >= world || memcmp(id.internal + 8, "rccl-emu", 8)) return 5;
  uint64_t v, pid;
  memcpy(&v, id.internal, sizeof v);
  memcpy(&pid, id.internal + 16, sizeof pid);
  // MSI_RCCL_EMU_SHM=1 (one process per rank: every rank, the id's creator too, joins through the shared segment)
  const char *shm = getenv("MSI_RCCL_EMU_SHM");
  if ((shm && shm[0] == '1') || pid != (uint64_t)getpid()) {
    ShmComm *sc = shm_join(id.internal, world, rank);
    if (!sc) return 2;
    *comm = new AnyComm();
    (*comm)->shm = sc;
    return 0;
  }
  std::shared_ptr<World> w;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto &slot = g_by_id[v];
    if (!slot) {
      slot = std::make_shared<World>();
      slot->n = world;
      slot->send.assign(world, nullptr);
      slot->recv.assign(world, nullptr);
      slot->bytes.assign(world, 0);
    }
    w = slot;
  }
  if (w->n != world) return 5;
  {   // as the real call: returns once every rank of the world has joined
    std::unique_lock<std::mutex> lk(w->mu);
    ++w->joined;
    w->cv.notify_all();
    w->cv.wait(lk, [&] { return w->joined >= w->n; });
  }
  *comm = new AnyComm();
  (*comm)->inproc = new Comm();
  (*comm)->inproc->w = w;
  (*comm)->inproc->rank = rank;
  return 0;
}
int ncclCommDestroy(AnyComm *c) {
  if (!c) return 0;
  if (c->shm) {
    // the last rank to leave removes the name (a crashed run leaves /dev/shm/msi_rccl_emu_*: the test sweeps them)
    if (c->shm->w->left.fetch_add(1, std::memory_order_acq_rel) + 1 == c->shm->w->n) shm_unlink(c->shm->name);
    munmap(c->shm->w, sizeof(ShmWorld));
    delete c->shm;
  }
  delete c->inproc;
  delete c;
  return 0;
}
int ncclGroupStart() {
  ++t_group_depth;
  return 0;
}
int ncclGroupEnd() {
  if (t_group_depth <= 0) return 5;
  --t_group_depth;
  return 0;
}
int ncclAllGather(const void *send, void *recv, size_t count, int dtype, AnyComm *c, void *stream) {
  (void)stream;   // the emulation's launches and copies have completed when they return
  if (!c || !send || !recv) return 5;
  const size_t bytes = count * elem_size(dtype);
  if (c->shm) {
    ShmWorld *w = c->shm->w;
    if (bytes > SHM_SLOT) return 5;
    memcpy(w->data[c->shm->rank], send, bytes);
    w->bytes[c->shm->rank] = bytes;
    shm_barrier(w);
    int rc = 0;
    for (uint32_t j = 0; j < w->n; ++j) {
      if (w->bytes[j] != bytes) rc = 5;
      else memcpy((char *)recv + (size_t)j * bytes, w->data[j], bytes);
    }
    shm_barrier(w);
    return rc;
  }
  return arrive(c->inproc, send, recv, bytes);
}
const char *ncclGetErrorString(int r) { return r == 0 ? "no error" : r == 5 ? "invalid argument (rccl emulation)" : "error (rccl emulation)"; }

}  // extern "C"
