// TEST INFRASTRUCTURE — the few RCCL entry points msi_group.hip resolves with dlsym, for the CPU-emulated build
// (tests/emu/run_emulated.py points MSI_RCCL_LIBRARY at the library built from this file).  "Devices" of the emulation
// share the host's memory, so a collective is memcpy between the ranks' buffers; what this checks is the HOST side of
// msi_group / msi_vs_group: one communicator per device, group start / end around the per-device calls, buffer sizes and
// offsets of the packed exchange, the per-rank form joining through a unique id from several threads.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

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

}  // namespace

extern "C" {

struct NcclId { char internal[128]; };

int ncclCommInitAll(Comm **comms, int n, const int *devices) {
  if (!comms || n < 1) return 5;
  (void)devices;
  auto w = std::make_shared<World>();
  w->n = n;
  w->send.assign(n, nullptr);
  w->recv.assign(n, nullptr);
  w->bytes.assign(n, 0);
  for (int i = 0; i < n; ++i) {
    comms[i] = new Comm();
    comms[i]->w = w;
    comms[i]->rank = i;
  }
  return 0;
}
int ncclGetUniqueId(NcclId *id) {
  std::lock_guard<std::mutex> lk(g_mu);
  memset(id->internal, 0, sizeof id->internal);
  const uint64_t v = g_next_id++;
  memcpy(id->internal, &v, sizeof v);
  memcpy(id->internal + 8, "rccl-emu", 8);
  return 0;
}
int ncclCommInitRank(Comm **comm, int world, NcclId id, int rank) {
  if (!comm || world < 1 || rank < 0 || rank >= world || memcmp(id.internal + 8, "rccl-emu", 8)) return 5;
  uint64_t v;
  memcpy(&v, id.internal, sizeof v);
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
  *comm = new Comm();
  (*comm)->w = w;
  (*comm)->rank = rank;
  return 0;
}
int ncclCommDestroy(Comm *c) {
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
int ncclAllGather(const void *send, void *recv, size_t count, int dtype, Comm *c, void *stream) {
  (void)stream;   // the emulation's launches and copies have completed when they return
  if (!c || !send || !recv) return 5;
  return arrive(c, send, recv, count * elem_size(dtype));
}
const char *ncclGetErrorString(int r) { return r == 0 ? "no error" : r == 5 ? "invalid argument (rccl emulation)" : "error (rccl emulation)"; }

}  // extern "C"
