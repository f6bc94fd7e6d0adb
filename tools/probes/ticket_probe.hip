// Probe: can the per-tile InstanceNorm partial sums be reduced by the LAST-ARRIVING workgroup of the producing
// kernel (no k_stats_finalize launch) without the device-scope release fence that cost 7.6x (lowc_probe)?
// Protocol under test (gfx942 / gfx950 memory model: agent-scope monotonic atomics carry sc1 = coherent across the
// 8 XCD L2s; a release FENCE would add buffer_wbl2 = write back everything dirty in this XCD's L2):
//   every workgroup: y tile with ordinary stores; its partials with agent-scope RELAXED atomic stores;
//                    s_waitcnt vmcnt(0); ticket = agent-scope relaxed fetch_add
//   last workgroup : partials of all tiles with agent-scope relaxed atomic loads, fixed-order fp64 sum
// Checks the result every iteration (a stale read shows as a wrong sum) and times: no stats / ticket / fence+ticket.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/ticket_probe.hip -o tools/probes/bin/ticket_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>   // 0: partials only (finalize would be another launch)  1: ticket, relaxed  2: __threadfence + ticket
__global__ __launch_bounds__(256) void k_producer(float* y, float* part, unsigned* ticket, double* out, int C, int iter) {
  const int tile = blockIdx.x, T = gridDim.x, tid = threadIdx.x;
  // the "convolution output": 128 pixels x 32 channels per workgroup, ordinary stores
  float4 v = make_float4(tile + iter, tid, 1.f, 2.f);
  for (int i = 0; i < 4; ++i) reinterpret_cast<float4*>(y)[((size_t)tile * 4 + i) * 256 + tid] = v;
  __shared__ int last;
  if (tid < C) {
    const float p = (float)((tile * 7 + tid * 3 + iter) % 101);   // this tile's partial sum of channel tid
    if (MODE == 0) part[(size_t)tile * C + tid] = p;
    else __hip_atomic_store(&part[(size_t)tile * C + tid], p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  if (MODE == 0) return;
  if (MODE == 2) __threadfence();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    last = (t == (unsigned)(T - 1));
  }
  __syncthreads();
  if (!last) return;
  // last arriver: 256 threads = C channels x (256 / C) slices, fixed-order fp64 sums, then across slices in LDS
  __shared__ double red[256];
  const int c = tid % C, sl = tid / C, S = 256 / C;
  double a = 0.0;
  for (int t = sl; t < T; t += S) a += (double)__hip_atomic_load(&part[(size_t)t * C + c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  red[tid] = a;
  __syncthreads();
  if (tid < C) {
    double s = 0.0;
    for (int k = 0; k < S; ++k) s += red[k * C + tid];
    out[tid] = s;
  }
  if (tid == 0) *ticket = 0;   // ready for the next launch (kernel boundary orders it)
}

__global__ void k_finalize(const float* part, double* out, int T, int C) {   // the separate launch of MODE 0
  __shared__ double red[256];
  const int tid = threadIdx.x, c = tid % C, sl = tid / C, S = 256 / C;
  double a = 0.0;
  for (int t = sl; t < T; t += S) a += (double)part[(size_t)t * C + c];
  red[tid] = a;
  __syncthreads();
  if (tid < C) { double s = 0.0; for (int k = 0; k < S; ++k) s += red[k * C + tid]; out[tid] = s; }
}

int main() {
  const int C = 32;
  for (int T : {2048, 512, 128, 32}) {
    float *y, *part; unsigned* ticket; double* out;
    hipMalloc(&y, (size_t)T * 16384); hipMalloc(&part, (size_t)T * C * 4); hipMalloc(&ticket, 4); hipMalloc(&out, C * 8);
    hipMemset(ticket, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[3];
    long bad[3] = {0, 0, 0};
    std::vector<double> h(C);
    for (int mode = 0; mode < 3; ++mode) {
      const int iters = 300;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        for (int it = 0; it < iters; ++it) {
          if (mode == 0) { hipLaunchKernelGGL(k_producer<0>, dim3(T), dim3(256), 0, 0, y, part, ticket, out, C, it); hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, 0, part, out, T, C); }
          else if (mode == 1) hipLaunchKernelGGL(k_producer<1>, dim3(T), dim3(256), 0, 0, y, part, ticket, out, C, it);
          else hipLaunchKernelGGL(k_producer<2>, dim3(T), dim3(256), 0, 0, y, part, ticket, out, C, it);
        }
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms[mode], e0, e1);
      }
      // correctness: every iteration checked
      for (int it = 0; it < 2000; ++it) {
        if (mode == 0) { hipLaunchKernelGGL(k_producer<0>, dim3(T), dim3(256), 0, 0, y, part, ticket, out, C, it); hipLaunchKernelGGL(k_finalize, dim3(1), dim3(256), 0, 0, part, out, T, C); }
        else if (mode == 1) hipLaunchKernelGGL(k_producer<1>, dim3(T), dim3(256), 0, 0, y, part, ticket, out, C, it);
        else hipLaunchKernelGGL(k_producer<2>, dim3(T), dim3(256), 0, 0, y, part, ticket, out, C, it);
        hipMemcpy(h.data(), out, C * 8, hipMemcpyDeviceToHost);
        for (int c = 0; c < C; ++c) {
          double want = 0.0;
          for (int t = 0; t < T; ++t) want += (double)((t * 7 + c * 3 + it) % 101);
          if (h[c] != want) ++bad[mode];
        }
      }
      ms[mode] /= 300;
    }
    printf("T=%4d tiles: producer + separate finalize %6.2f us | relaxed ticket %6.2f us (wrong sums %ld / 64000) | fence + ticket %6.2f us (wrong %ld)\n",
           T, ms[0] * 1e3, ms[1] * 1e3, bad[1], ms[2] * 1e3, bad[2]);
    hipFree(y); hipFree(part); hipFree(ticket); hipFree(out);
  }
  return 0;
}
