// Probe 3: filter fragments straight from global memory (L1/L2) into registers, one tap ahead:
// no LDS round trip and no workgroup barrier per tap; A halo tile staged in LDS once per chunk.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: LDS-staged B (reference structure), 1: B fragments from global
__global__ __launch_bounds__(256) void conv_like(const float* x, const float* w, float* y, int H, int W, int Cin, int tilesX) {
  constexpr int BK = 32, CK = 36, IH = 10, IW = 18;
  __shared__ __attribute__((aligned(16))) float sA[IH * IW * CK];
  __shared__ __attribute__((aligned(16))) float sB[MODE == 0 ? 2 * 32 * CK : 4];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5, wave = tid >> 6;
  const int tile = blockIdx.x, ty0 = (tile / tilesX) * 8, tx0 = (tile % tilesX) * 16;
  const int fy = wave * 2 + li / 16, fx = li % 16;
  f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int ac4 = tid % 8;
  const int wrow = 9 * Cin;
  float4 areg[6], breg;
  auto prefetchA = [&](int kc) {
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int idx = tid + i * 256, pix = idx / 8;
      int iy = ty0 - 1 + pix / IW, ix = tx0 - 1 + pix % IW;
      iy = min(max(iy, 0), H - 1); ix = min(max(ix, 0), W - 1);
      areg[i] = *reinterpret_cast<const float4*>(x + ((size_t)iy * W + ix) * Cin + kc + ac4 * 4);
    }
  };
  auto writeA = [&]() {
#pragma unroll
    for (int i = 0; i < 6; ++i) { const int idx = tid + i * 256, pix = idx / 8; if (idx < IH * IW * 8) *reinterpret_cast<float4*>(sA + pix * CK + ac4 * 4) = areg[i]; }
  };
  const float* wlane = w + (size_t)li * wrow + lh * 4;      // this lane's filter row (column li), k offset 4*lh
  float4 bnext[4], bcur[4];
  auto loadFrag = [&](int kc, int tap) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) bnext[kb] = *reinterpret_cast<const float4*>(wlane + tap * Cin + kc + kb * 8);
  };
  prefetchA(0);
  if (MODE == 1) loadFrag(0, 0);
  else breg = *reinterpret_cast<const float4*>(w + (size_t)(tid / 8) * wrow + ac4 * 4);
  for (int kc = 0; kc < Cin; kc += BK) {
    __syncthreads();
    writeA();
    if (kc + BK < Cin) prefetchA(kc + BK);
    if (MODE == 1) __syncthreads();
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      if (MODE == 0) {
        const int buf = tap & 1;
        *reinterpret_cast<float4*>(sB + buf * 32 * CK + (tid / 8) * CK + ac4 * 4) = breg;
        { int ntap = tap + 1, nkc = kc; if (ntap == 9) { ntap = 0; nkc = kc + BK; } if (nkc < Cin) breg = *reinterpret_cast<const float4*>(w + (size_t)(tid / 8) * wrow + ntap * Cin + nkc + ac4 * 4); }
        __syncthreads();
        const int aoff = ((fy + tap / 3) * IW + fx + tap % 3) * CK + lh * 4;
        const float* sBb = sB + buf * 32 * CK + li * CK + lh * 4;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const float4 a = *reinterpret_cast<const float4*>(sA + aoff + kb * 8);
          const float4 b = *reinterpret_cast<const float4*>(sBb + kb * 8);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) bcur[kb] = bnext[kb];
        { int ntap = tap + 1, nkc = kc; if (ntap == 9) { ntap = 0; nkc = kc + BK < Cin ? kc + BK : kc; } loadFrag(nkc, ntap); }
        const int aoff = ((fy + tap / 3) * IW + fx + tap % 3) * CK + lh * 4;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const float4 a = *reinterpret_cast<const float4*>(sA + aoff + kb * 8);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bcur[kb].x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bcur[kb].y, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bcur[kb].z, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bcur[kb].w, acc, 0, 0, 0);
        }
      }
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
    const int oy = ty0 + wave * 2 + row / 16, ox = tx0 + row % 16;
    y[((size_t)oy * W + ox) * 32 + li] = acc[r];
  }
}

template <int MODE>
void run(const char* name, const float* x, const float* w, float* y, int H, int W, int Cin) {
  const int tilesX = W / 16, tiles = (H / 8) * tilesX;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((conv_like<MODE>), dim3(tiles), dim3(256), 0, 0, x, w, y, H, W, Cin, tilesX);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  ms /= 10;
  printf("H=%4d Cin=%3d %-40s: %7.1f us %6.1f TFLOP/s (%d WGs)\n", H, Cin, name, ms * 1e3, 2.0 * Cin * 9 * 32 * (double)H * W / ms / 1e9, tiles);
}

int main() {
  float *x, *w, *y;
  hipMalloc(&x, (size_t)512 * 512 * 64 * 4); hipMalloc(&w, (size_t)64 * 9 * 512 * 4 + (1 << 20)); hipMalloc(&y, (size_t)512 * 512 * 32 * 4);
  hipMemset(x, 0x3c, (size_t)512 * 512 * 64 * 4); hipMemset(w, 0x3c, (size_t)64 * 9 * 512 * 4 + (1 << 20));
  struct { int H, C; } cases[] = {{512, 64}, {256, 64}, {128, 256}, {64, 512}};
  for (auto c : cases) {
    run<0>("LDS-staged filters, barrier per tap", x, w, y, c.H, c.H, c.C);
    run<1>("filter fragments from global, no tap barrier", x, w, y, c.H, c.H, c.C);
  }
  return 0;
}
