// Microbenchmark (diagnostics): how fast does gfx950 start workgroups?  Empty kernels of N workgroups x 256 threads,
// with and without 20 KiB of LDS, one stream and four streams.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <vector>
__global__ void k_empty(int *p) { if (p && threadIdx.x == 0 && blockIdx.x == 0xFFFFFFF) *p = 1; }
__global__ void k_lds(int *p) {
  __shared__ int s[5120];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  if (p && s[(threadIdx.x + 1) & 255] == -1) *p = 1;
}
__global__ void k_work(int *p, int iters) {   // ~iters dependent loads: a latency-bound workgroup
  __shared__ int s[5120];
  s[threadIdx.x] = threadIdx.x;
  __syncthreads();
  int v = threadIdx.x;
  for (int i = 0; i < iters; ++i) v = p[(v * 64 + blockIdx.x * 1024) & 0xFFFFF];
  if (v == -1) p[0] = s[0];
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
  int *d;
  hipMalloc(&d, 4 << 20);
  hipMemset(d, 0, 4 << 20);
  hipStream_t st[8];
  for (auto &s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  const int Ns[] = {153, 612, 2448, 9792, 39168, 156672};
  for (int mode = 0; mode < 3; ++mode)
    for (int N : Ns) {
      const int reps = N > 10000 ? 20 : 200;
      for (int warm = 0; warm < 2; ++warm) {
        hipDeviceSynchronize();
        const double t0 = now();
        for (int r = 0; r < reps; ++r) {
          if (mode == 0) hipLaunchKernelGGL(k_empty, dim3(N), dim3(256), 0, st[0], d);
          else if (mode == 1) hipLaunchKernelGGL(k_lds, dim3(N), dim3(256), 0, st[0], d);
          else hipLaunchKernelGGL(k_work, dim3(N), dim3(256), 0, st[0], d, 20);
        }
        hipDeviceSynchronize();
        const double t = (now() - t0) / reps;
        if (warm) printf("mode %d (%s) 1 stream  N=%6d: %8.1f us per kernel, %.3f us per workgroup\n", mode, mode == 0 ? "empty" : mode == 1 ? "20KB LDS" : "20 dependent loads", N, t * 1e6, t * 1e6 / N);
      }
    }
  // concurrent small kernels on 8 streams: aggregate workgroup start rate
  for (int mode = 1; mode < 3; ++mode)
    for (int N : {153, 612, 2448}) {
      const int reps = 200;
      for (int warm = 0; warm < 2; ++warm) {
        hipDeviceSynchronize();
        const double t0 = now();
        for (int r = 0; r < reps; ++r)
          for (int s = 0; s < 8; ++s) {
            if (mode == 1) hipLaunchKernelGGL(k_lds, dim3(N), dim3(256), 0, st[s], d);
            else hipLaunchKernelGGL(k_work, dim3(N), dim3(256), 0, st[s], d, 20);
          }
        hipDeviceSynchronize();
        const double t = (now() - t0) / reps / 8;
        if (warm) printf("mode %d 8 streams N=%6d: %8.1f us per kernel (amortised), %.3f us per workgroup\n", mode, N, t * 1e6, t * 1e6 / N);
      }
    }
  return 0;
}
