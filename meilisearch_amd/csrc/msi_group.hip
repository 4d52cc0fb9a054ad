// msi_group.hip — multi-GPU inside libmsi (SURVEY §8 e): the reference is ONE server process that fans searches out
// from spawn_blocking threads (crates/meilisearch/src/search/federated/perform.rs:224), so the multi-device entry
// points live behind the C ABI, not in a Python harness.
//
//   msi_group           one context per device + one RCCL communicator per device.  Two ways to form it:
//                         in-process  msi_group_create(devices, n)            (ncclCommInitAll; the server owns all GPUs)
//                         per-rank    msi_group_create_rank(ctx, rank, world, id)   (one process per GPU, as bench.py is
//                                     launched; `id` = msi_group_unique_id() of rank 0, handed round by the launcher)
//   msi_vs_group        a vector store over the group (in-process):
//                         MSI_GROUP_REPLICATE   every device holds all rows; a batch of queries is split across the
//                                               devices, no exchange step at all (the north_star's query sharding);
//                         MSI_GROUP_SHARD_ROWS  contiguous row ranges; every device scans its rows for the whole batch,
//                                               the per-device top-k travel as ONE packed buffer per device
//                                               ({distance bits, docid}[B][k] + counts[B]) in one ncclAllGather over
//                                               xGMI, then the k-way merge kernel (msi_merge_topk_device) — the
//                                               concatenate + sort_unstable_by_key tail of store.rs:1059,1090.
//   msi_group_allgather the same exchange for the per-rank form: device buffers in, RCCL inside the library.
//
// RCCL is loaded with dlopen at group creation (librccl.so is not needed by anything else in the library; a box
// without it still loads libmsi.so and runs single-device).  Payloads are B*k*8 bytes: latency-bound, far from the
// per-link xGMI bound, so one collective per batch is the whole communication design.
#include <dlfcn.h>
#include <string.h>

#include <algorithm>
#include <thread>
#include <vector>

#include "msi_common.h"

namespace {

// the few RCCL entry points used, resolved at run time
typedef struct ncclComm *ncclComm_t;
struct NcclId {
  char internal[128];
};
struct Rccl {
  void *h = nullptr;
  int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  int (*GetUniqueId)(NcclId *) = nullptr;
  int (*CommInitRank)(ncclComm_t *, int, NcclId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int /*ncclDataType_t*/, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
};
Rccl g_rccl;
std::mutex g_rccl_mu;
constexpr int NCCL_INT8 = 0;   // ncclInt8 / ncclChar

int32_t rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.h) return MSI_OK;
  // MSI_RCCL_LIBRARY: the library to load instead of the loader's librccl.so (a side-by-side ROCm; the test tier's stand-in)
  const char *named = getenv("MSI_RCCL_LIBRARY");
  void *h = named && named[0] ? dlopen(named, RTLD_NOW | RTLD_GLOBAL) : nullptr;
  if (named && named[0] && !h) {
    msi_set_error("msi_group: MSI_RCCL_LIBRARY=%s could not be loaded (%s)", named, dlerror());
    return MSI_E_UNSUPPORTED;
  }
  if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!h) {
    msi_set_error("msi_group: librccl.so could not be loaded (%s)", dlerror());
    return MSI_E_UNSUPPORTED;
  }
  Rccl r;
  r.h = h;
  r.CommInitAll = (decltype(r.CommInitAll))dlsym(h, "ncclCommInitAll");
  r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
  r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
  r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
  r.AllGather = (decltype(r.AllGather))dlsym(h, "ncclAllGather");
  r.GroupStart = (decltype(r.GroupStart))dlsym(h, "ncclGroupStart");
  r.GroupEnd = (decltype(r.GroupEnd))dlsym(h, "ncclGroupEnd");
  r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
  if (!r.CommInitAll || !r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.AllGather || !r.GroupStart || !r.GroupEnd) {
    msi_set_error("msi_group: librccl.so lacks an expected symbol");
    return MSI_E_UNSUPPORTED;
  }
  g_rccl = r;
  return MSI_OK;
}

#define MSI_NCCL_TRY(expr)                                                                              \
  do {                                                                                                  \
    const int _r = (expr);                                                                              \
    if (_r != 0) {                                                                                      \
      msi_set_error("%s failed: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?");   \
      return MSI_E_HIP;                                                                                 \
    }                                                                                                   \
  } while (0)

}  // namespace

struct msi_group {
  std::vector<msi_ctx *> ctx;       // in-process: one per device (owned); per-rank: the caller's context (borrowed)
  std::vector<ncclComm_t> comm;
  uint32_t rank = 0, world = 1;     // per-rank form
  bool in_process = true;
};

struct msi_vs_group {
  msi_group *g = nullptr;
  uint32_t dim = 0;
  int32_t mode = 0;
  std::vector<msi_vs *> st;
  // row-shard exchange buffers, per device: send {dist bits, docid}[B][k] + counts[B] as int32, recv world x that
  std::vector<DevBuf> send, recv, q, o_ids, o_dist, o_cnt, m_ids, m_dist, m_cnt, g_ids, g_dist, g_cnt, inex;
};

__global__ void group_pack_kernel(const float *__restrict__ dist, const uint32_t *__restrict__ ids,
                                  const uint32_t *__restrict__ cnt, uint32_t nq, uint32_t k, uint32_t *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = nq * k;
  if (i < n) {
    out[i] = __float_as_uint(dist[i]);
    out[n + i] = ids[i];
  }
  if (i < nq) out[2 * n + i] = cnt[i];
}
// gathered [world][2*nq*k + nq] -> the [world][nq][k] / [world][nq] arrays msi_merge_topk_device reads
__global__ void group_unpack_kernel(const uint32_t *__restrict__ in, uint32_t world, uint32_t nq, uint32_t k,
                                    float *__restrict__ dist, uint32_t *__restrict__ ids, uint32_t *__restrict__ cnt) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t n = nq * k, per = 2 * n + nq;
  if (i < world * n) {
    const uint32_t w = i / n, j = i % n;
    dist[i] = __uint_as_float(in[(size_t)w * per + j]);
    ids[i] = in[(size_t)w * per + n + j];
  }
  if (i < world * nq) cnt[i] = in[(size_t)(i / nq) * per + 2 * n + (i % nq)];
}

extern "C" {

int32_t msi_group_create(const int32_t *devices, uint32_t n, msi_group **out) {
  if (!devices || !n || !out || n > 64) {
    msi_set_error("msi_group_create: invalid argument");
    return MSI_E_INVALID;
  }
  *out = nullptr;
  MSI_TRY(rccl_load());
  msi_group *g = new msi_group();
  g->world = n;
  for (uint32_t i = 0; i < n; ++i) {
    msi_ctx *c = nullptr;
    const int32_t st = msi_ctx_create(devices[i], &c);
    if (st != MSI_OK) {
      for (msi_ctx *x : g->ctx) msi_ctx_destroy(x);
      delete g;
      return st;
    }
    g->ctx.push_back(c);
  }
  g->comm.assign(n, nullptr);
  std::vector<int> devs(devices, devices + n);
  const int r = g_rccl.CommInitAll(g->comm.data(), (int)n, devs.data());
  if (r != 0) {
    msi_set_error("ncclCommInitAll over %u devices failed: %s", n, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
    for (msi_ctx *x : g->ctx) msi_ctx_destroy(x);
    delete g;
    return MSI_E_HIP;
  }
  *out = g;
  return MSI_OK;
}

int32_t msi_group_unique_id(uint8_t out_id[128]) {
  if (!out_id) return MSI_E_INVALID;
  MSI_TRY(rccl_load());
  NcclId id;
  MSI_NCCL_TRY(g_rccl.GetUniqueId(&id));
  memcpy(out_id, id.internal, 128);
  return MSI_OK;
}

int32_t msi_group_create_rank(msi_ctx *ctx, uint32_t rank, uint32_t world, const uint8_t id[128], msi_group **out) {
  if (!ctx || !out || !id || world == 0 || rank >= world) {
    msi_set_error("msi_group_create_rank: invalid argument");
    return MSI_E_INVALID;
  }
  *out = nullptr;
  MSI_TRY(rccl_load());
  DeviceGuard dg(ctx->device);
  NcclId nid;
  memcpy(nid.internal, id, 128);
  ncclComm_t comm = nullptr;
  MSI_NCCL_TRY(g_rccl.CommInitRank(&comm, (int)world, nid, (int)rank));
  msi_group *g = new msi_group();
  g->in_process = false;
  g->rank = rank;
  g->world = world;
  g->ctx.push_back(ctx);
  g->comm.push_back(comm);
  msi_ctx_retain(ctx);
  *out = g;
  return MSI_OK;
}

void msi_group_destroy(msi_group *g) {
  if (!g) return;
  for (ncclComm_t c : g->comm)
    if (c && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c);
  if (g->in_process) {
    for (msi_ctx *c : g->ctx) msi_ctx_destroy(c);
  } else {
    for (msi_ctx *c : g->ctx) msi_ctx_release(c);
  }
  delete g;
}

uint32_t msi_group_size(const msi_group *g) { return g ? g->world : 0; }
msi_ctx *msi_group_ctx(msi_group *g, uint32_t i) { return g && i < g->ctx.size() ? g->ctx[i] : nullptr; }

// Per-rank form: every rank contributes `bytes` from d_send; d_recv receives world x bytes in rank order.  Enqueued on
// the context's stream (in order with the searches that produced d_send), not synchronised.
int32_t msi_group_allgather(msi_group *g, const void *d_send, size_t bytes, void *d_recv) {
  if (!g || g->in_process || !d_send || !d_recv || !bytes) {
    msi_set_error("msi_group_allgather: needs a per-rank group (msi_group_create_rank) and device buffers");
    return MSI_E_INVALID;
  }
  msi_ctx *ctx = g->ctx[0];
  std::lock_guard<std::mutex> lk(ctx->mu);
  DeviceGuard dg(ctx->device);
  MSI_NCCL_TRY(g_rccl.AllGather(d_send, d_recv, bytes, NCCL_INT8, g->comm[0], ctx->stream));
  return MSI_OK;
}

int32_t msi_vs_group_create(msi_group *g, uint32_t dim, int32_t storage, int32_t mode, msi_vs_group **out) {
  if (!g || !g->in_process || !out || (mode != MSI_GROUP_REPLICATE && mode != MSI_GROUP_SHARD_ROWS)) {
    msi_set_error("msi_vs_group_create: needs an in-process group and a valid mode");
    return MSI_E_INVALID;
  }
  msi_vs_group *v = new msi_vs_group();
  v->g = g;
  v->dim = dim;
  v->mode = mode;
  const size_t n = g->ctx.size();
  for (size_t i = 0; i < n; ++i) {
    msi_vs *s = nullptr;
    const int32_t st = msi_vs_create_typed(g->ctx[i], dim, storage, &s);
    if (st != MSI_OK) {
      for (msi_vs *x : v->st) msi_vs_destroy(x);
      delete v;
      return st;
    }
    v->st.push_back(s);
  }
  for (auto *b : {&v->send, &v->recv, &v->q, &v->o_ids, &v->o_dist, &v->o_cnt, &v->m_ids, &v->m_dist, &v->m_cnt, &v->g_ids,
                  &v->g_dist, &v->g_cnt, &v->inex})
    b->resize(n);
  *out = v;
  return MSI_OK;
}

void msi_vs_group_destroy(msi_vs_group *v) {
  if (!v) return;
  for (size_t i = 0; i < v->st.size(); ++i) {
    DeviceGuard dg(v->g->ctx[i]->device);
    (void)hipStreamSynchronize(v->g->ctx[i]->stream);
    for (auto *b : {&v->send, &v->recv, &v->q, &v->o_ids, &v->o_dist, &v->o_cnt, &v->m_ids, &v->m_dist, &v->m_cnt, &v->g_ids,
                    &v->g_dist, &v->g_cnt, &v->inex})
      (*b)[i].release();
    msi_vs_destroy(v->st[i]);
  }
  delete v;
}

int32_t msi_vs_group_upload(msi_vs_group *v, const uint32_t *docids, const float *rows, uint64_t n_rows) {
  if (!v || (n_rows && (!docids || !rows))) return MSI_E_INVALID;
  const uint64_t n = v->st.size();
  for (uint64_t i = 0; i < n; ++i) {
    if (v->mode == MSI_GROUP_REPLICATE) {
      MSI_TRY(msi_vs_upload(v->st[i], docids, rows, n_rows));
    } else {
      const uint64_t r0 = n_rows * i / n, r1 = n_rows * (i + 1) / n;   // contiguous ranges: docids stay global
      MSI_TRY(msi_vs_upload(v->st[i], docids + r0, rows + r0 * v->dim, r1 - r0));
    }
  }
  return MSI_OK;
}

int32_t msi_vs_group_search(msi_vs_group *v, const float *queries, uint32_t nq, uint32_t k, uint32_t *out_docids,
                            float *out_dist, uint32_t *out_counts) {
  if (!v || (nq && (!queries || !out_docids || !out_dist || !out_counts)) || !k) return MSI_E_INVALID;
  if (!nq) return MSI_OK;
  const uint32_t n = (uint32_t)v->st.size();
  if (v->mode == MSI_GROUP_REPLICATE) {
    // queries sharded over the replicas: no exchange step; one caller thread per device (the searches block)
    std::vector<int32_t> st(n, MSI_OK);
    std::vector<std::string> err(n);
    std::vector<std::thread> th;
    for (uint32_t i = 0; i < n; ++i) {
      const uint32_t q0 = (uint32_t)((uint64_t)nq * i / n), q1 = (uint32_t)((uint64_t)nq * (i + 1) / n);
      if (q1 == q0) continue;
      th.emplace_back([=, &st, &err] {
        st[i] = msi_vs_search(v->st[i], queries + (size_t)q0 * v->dim, q1 - q0, k, nullptr, 0, nullptr,
                              out_docids + (size_t)q0 * k, out_dist + (size_t)q0 * k, out_counts + q0);
        if (st[i] != MSI_OK) err[i] = msi_last_error();
      });
    }
    for (auto &t : th) t.join();
    for (uint32_t i = 0; i < n; ++i)
      if (st[i] != MSI_OK) {
        msi_set_error("%s", err[i].c_str());
        return st[i];
      }
    return MSI_OK;
  }
  // rows sharded: every device scans its rows for the whole batch, ONE packed all-gather, merge on device 0
  const size_t per = (size_t)2 * nq * k + nq;   // u32 words per device
  if ((size_t)n * k > 2048) {
    msi_set_error("msi_vs_group_search: devices x k = %zu exceeds the merge kernel's 2048", (size_t)n * k);
    return MSI_E_UNSUPPORTED;
  }
  for (uint32_t i = 0; i < n; ++i) {
    msi_ctx *c = v->g->ctx[i];
    DeviceGuard dg(c->device);
    MSI_TRY(v->q[i].ensure((size_t)nq * v->dim * 4));
    MSI_TRY(v->o_ids[i].ensure((size_t)nq * k * 4));
    MSI_TRY(v->o_dist[i].ensure((size_t)nq * k * 4));
    MSI_TRY(v->o_cnt[i].ensure((size_t)nq * 4));
    MSI_TRY(v->send[i].ensure(per * 4));
    MSI_TRY(v->recv[i].ensure(per * 4 * n));
    MSI_HIP_TRY(hipMemcpyAsync(v->q[i].p, queries, (size_t)nq * v->dim * 4, hipMemcpyHostToDevice, c->stream));
    MSI_HIP_TRY(hipMemsetAsync(v->o_cnt[i].p, 0, (size_t)nq * 4, c->stream));
    MSI_TRY(v->inex[i].ensure((size_t)nq * 4));
    MSI_HIP_TRY(hipMemsetAsync(v->inex[i].p, 0, (size_t)nq * 4, c->stream));
    // the device path keeps the lists on the device; a query whose exactness proof fails on a shard is flagged and the
    // whole batch is then answered through the host entry points below (exhaustive rerun included)
    MSI_TRY(msi_vs_search_device(v->st[i], v->q[i].as<float>(), nq, k, nullptr, 0, v->o_ids[i].as<uint32_t>(),
                                 v->o_dist[i].as<float>(), v->o_cnt[i].as<uint32_t>(), v->inex[i].as<uint32_t>()));
    hipLaunchKernelGGL(group_pack_kernel, dim3((uint32_t)((nq * k + 255) / 256)), dim3(256), 0, c->stream,
                       v->o_dist[i].as<float>(), v->o_ids[i].as<uint32_t>(), v->o_cnt[i].as<uint32_t>(), nq, k,
                       v->send[i].as<uint32_t>());
    MSI_HIP_TRY(hipGetLastError());
  }
  MSI_NCCL_TRY(g_rccl.GroupStart());
  for (uint32_t i = 0; i < n; ++i) {
    DeviceGuard dg(v->g->ctx[i]->device);
    const int r = g_rccl.AllGather(v->send[i].p, v->recv[i].p, per * 4, NCCL_INT8, v->g->comm[i], v->g->ctx[i]->stream);
    if (r != 0) {
      (void)g_rccl.GroupEnd();
      msi_set_error("ncclAllGather failed: %s", g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "?");
      return MSI_E_HIP;
    }
  }
  MSI_NCCL_TRY(g_rccl.GroupEnd());
  {
    msi_ctx *c = v->g->ctx[0];
    DeviceGuard dg(c->device);
    MSI_TRY(v->g_ids[0].ensure((size_t)n * nq * k * 4));
    MSI_TRY(v->g_dist[0].ensure((size_t)n * nq * k * 4));
    MSI_TRY(v->g_cnt[0].ensure((size_t)n * nq * 4));
    MSI_TRY(v->m_ids[0].ensure((size_t)nq * k * 4));
    MSI_TRY(v->m_dist[0].ensure((size_t)nq * k * 4));
    MSI_TRY(v->m_cnt[0].ensure((size_t)nq * 4));
    hipLaunchKernelGGL(group_unpack_kernel, dim3((uint32_t)(((size_t)n * nq * k + 255) / 256)), dim3(256), 0, c->stream,
                       v->recv[0].as<uint32_t>(), n, nq, k, v->g_dist[0].as<float>(), v->g_ids[0].as<uint32_t>(),
                       v->g_cnt[0].as<uint32_t>());
    MSI_HIP_TRY(hipGetLastError());
    MSI_TRY(msi_merge_topk_device(c, v->g_ids[0].as<uint32_t>(), v->g_dist[0].as<float>(), v->g_cnt[0].as<uint32_t>(), n, nq, k,
                                  v->m_ids[0].as<uint32_t>(), v->m_dist[0].as<float>(), v->m_cnt[0].as<uint32_t>()));
    MSI_HIP_TRY(hipMemcpyAsync(out_docids, v->m_ids[0].p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, c->stream));
    MSI_HIP_TRY(hipMemcpyAsync(out_dist, v->m_dist[0].p, (size_t)nq * k * 4, hipMemcpyDeviceToHost, c->stream));
    MSI_HIP_TRY(hipMemcpyAsync(out_counts, v->m_cnt[0].p, (size_t)nq * 4, hipMemcpyDeviceToHost, c->stream));
  }
  bool inexact = false;
  std::vector<uint32_t> flags(nq);
  for (uint32_t i = 0; i < n; ++i) {
    DeviceGuard dg(v->g->ctx[i]->device);
    MSI_HIP_TRY(hipStreamSynchronize(v->g->ctx[i]->stream));
    MSI_HIP_TRY(hipMemcpy(flags.data(), v->inex[i].p, (size_t)nq * 4, hipMemcpyDeviceToHost));
    for (uint32_t f : flags) inexact |= f != 0;
  }
  if (inexact) {
    // rare (duplicates / adversarial rows): every shard through msi_vs_search, lists merged on the host exactly as
    // nns_by_vector concatenates and sorts the per-store lists (store.rs:1059,1090)
    std::vector<uint32_t> ids((size_t)n * nq * k), cnt((size_t)n * nq);
    std::vector<float> dist((size_t)n * nq * k);
    for (uint32_t i = 0; i < n; ++i)
      MSI_TRY(msi_vs_search(v->st[i], queries, nq, k, nullptr, 0, nullptr, ids.data() + (size_t)i * nq * k,
                            dist.data() + (size_t)i * nq * k, cnt.data() + (size_t)i * nq));
    std::vector<uint32_t> qi((size_t)n * k), qc(n);
    std::vector<float> qd((size_t)n * k);
    for (uint32_t q = 0; q < nq; ++q) {
      for (uint32_t i = 0; i < n; ++i) {
        memcpy(qi.data() + (size_t)i * k, ids.data() + ((size_t)i * nq + q) * k, (size_t)k * 4);
        memcpy(qd.data() + (size_t)i * k, dist.data() + ((size_t)i * nq + q) * k, (size_t)k * 4);
        qc[i] = cnt[(size_t)i * nq + q];
      }
      out_counts[q] = msi_merge_topk(qi.data(), qd.data(), qc.data(), n, k, k, out_docids + (size_t)q * k, out_dist + (size_t)q * k);
    }
  }
  return MSI_OK;
}

}  // extern "C"
