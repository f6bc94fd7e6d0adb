// Micro-probe: what limits the fp32 MFMA issue rate in an LDS-fed, barrier-synchronised loop?
// hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_probe.hip -o gpurun_out/mfma_probe && ./gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE bit0: LDS reads feed the MFMAs; bit1: barrier per stage; bit2: global load + ds_write per stage
// MPS = MFMAs per stage per wave (multiple of 4); NACC accumulators
template <int MODE, int MPS, int NACC>
__global__ __launch_bounds__(256) void probe(const float* g, float* out, int stages) {
  __shared__ __attribute__((aligned(16))) float sA[180 * 36];
  __shared__ __attribute__((aligned(16))) float sB[2][64 * 36];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5, wave = tid >> 6;
  for (int i = tid; i < 180 * 36; i += 256) sA[i] = g[i];
  for (int i = tid; i < 2 * 64 * 36; i += 256) (&sB[0][0])[i] = g[i + 7000];
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float4 breg = make_float4(0, 0, 0, 0);
  if (MODE & 4) breg = *reinterpret_cast<const float4*>(g + (size_t)(blockIdx.x % 64) * 1024 + tid * 4);
  const float* pa = sA + ((wave * 2 + li / 16) * 18 + (li % 16)) * 36 + lh * 4;
  float4 ra = *reinterpret_cast<const float4*>(pa), rb = *reinterpret_cast<const float4*>(sB[0] + li * 36 + lh * 4);
  for (int s = 0; s < stages; ++s) {
    const int buf = s & 1;
    if (MODE & 4) {
      *reinterpret_cast<float4*>(sB[buf] + (tid / 8) * 36 + (tid % 8) * 4) = breg;
      breg = *reinterpret_cast<const float4*>(g + (size_t)((blockIdx.x + s) % 64) * 1024 + tid * 4);
    }
    if (MODE & 2) __syncthreads();
    const float* pb = sB[buf] + li * 36 + lh * 4;
    const int tapoff = ((s % 9) / 3 * 18 + (s % 3)) * 36;
#pragma unroll
    for (int k = 0; k < MPS / 4; ++k) {
      if (MODE & 1) {
        ra = *reinterpret_cast<const float4*>(pa + tapoff + (k % 4) * 8);
        rb = *reinterpret_cast<const float4*>(pb + (k % 4) * 8);
      }
      acc[(4 * k + 0) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra.x, rb.x, acc[(4 * k + 0) % NACC], 0, 0, 0);
      acc[(4 * k + 1) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra.y, rb.y, acc[(4 * k + 1) % NACC], 0, 0, 0);
      acc[(4 * k + 2) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra.z, rb.z, acc[(4 * k + 2) % NACC], 0, 0, 0);
      acc[(4 * k + 3) % NACC] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra.w, rb.w, acc[(4 * k + 3) % NACC], 0, 0, 0);
    }
  }
  float t = 0.f;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) t += acc[a][r];
  out[(size_t)blockIdx.x * 256 + tid] = t;
}

template <int MODE, int MPS, int NACC>
void run(const char* name, const float* g, float* out, int wgs) {
  const int total_mfma = 4608;           // per wave
  const int stages = total_mfma / MPS;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((probe<MODE, MPS, NACC>), dim3(wgs), dim3(256), 0, 0, g, out, stages);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  const double flops = (double)wgs * 4 * total_mfma * 4096.0;
  printf("%-34s mode %d MPS %3d NACC %d wgs %5d : %7.1f us  %6.1f TFLOP/s\n", name, MODE, MPS, NACC, wgs, ms * 1e3, flops / ms / 1e9);
}

int main() {
  float *g, *out;
  hipMalloc(&g, 1 << 22); hipMalloc(&out, 8192 * 256 * 4);
  std::vector<float> h(1 << 20);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) / 1000.f - 0.5f;
  hipMemcpy(g, h.data(), 1 << 22, hipMemcpyHostToDevice);
  for (int wgs : {256, 512, 1024, 2048}) {
    run<0, 16, 1>("regs only, 1 acc", g, out, wgs);
    run<0, 16, 2>("regs only, 2 acc", g, out, wgs);
    run<0, 16, 4>("regs only, 4 acc", g, out, wgs);
    run<1, 16, 1>("lds reads, no barrier, 1 acc", g, out, wgs);
    run<1, 16, 2>("lds reads, no barrier, 2 acc", g, out, wgs);
    run<3, 16, 2>("lds + barrier/16", g, out, wgs);
    run<3, 32, 2>("lds + barrier/32", g, out, wgs);
    run<3, 64, 2>("lds + barrier/64", g, out, wgs);
    run<3, 144, 2>("lds + barrier/144", g, out, wgs);
    run<7, 16, 2>("lds + barrier/16 + stage", g, out, wgs);
    run<7, 32, 2>("lds + barrier/32 + stage", g, out, wgs);
    run<7, 64, 2>("lds + barrier/64 + stage", g, out, wgs);
    run<7, 32, 4>("lds + barrier/32 + stage, 4 acc", g, out, wgs);
  }
  return 0;
}
