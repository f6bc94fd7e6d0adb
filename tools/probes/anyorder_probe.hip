// anyorder_probe.hip - does hipExtAnyOrderLaunch (AQL barrier bit cleared) let two independent kernels of ONE
// stream run concurrently on gfx950?  hip_ext.h says the flag "is not supported on AMD GFX9xx boards".
// Two kernels that each occupy a fraction of the chip for ~100 us: back to back = ~2x, overlapped = ~1x.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/anyorder_probe.hip -o tools/probes/bin/anyorder_probe
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <vector>

__global__ void spin(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; ++i) a = a * b + 1e-7f;
  if (a == 123.f) out[blockIdx.x] = a;
}

static double run(int nk, int blocks, int iters, unsigned flags, hipStream_t st, float* buf) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0, st);
    for (int k = 0; k < nk; ++k)
      hipExtLaunchKernelGGL(spin, dim3(blocks), dim3(256), 0, st, nullptr, nullptr, k == 0 ? 0u : flags, buf, iters);
    hipEventRecord(e1, st);
    hipEventSynchronize(e1);
  }
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e3;
}

int main() {
  hipStream_t st; hipStreamCreate(&st);
  float* buf; hipMalloc(&buf, 1 << 20);
  for (int blocks : {8, 64, 256, 1024}) {
    const int iters = 40000;
    const double one = run(1, blocks, iters, 0, st, buf);
    const double seq = run(4, blocks, iters, 0, st, buf);
    const double any = run(4, blocks, iters, hipExtAnyOrderLaunch, st, buf);
    printf("blocks %5d: 1 kernel %8.1f us | 4 in order %8.1f us | 4 any-order %8.1f us\n", blocks, one, seq, any);
  }
  // launch-latency bound: 64 tiny kernels
  const double seq = run(64, 1, 10, 0, st, buf), any = run(64, 1, 10, hipExtAnyOrderLaunch, st, buf);
  printf("64 tiny kernels: in order %8.1f us | any-order %8.1f us\n", seq, any);
  return 0;
}
