// Probe 2: start from the real per-tap conv loop and strip features to find what costs MFMA rate.
// FEAT bits: 1 = per-chunk A staging from global (prefetched regs), 2 = real strided weight loads,
//            4 = epilogue stores, 8 = per-tap aoff VALU math
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int FEAT>
__global__ __launch_bounds__(256) void conv_like(const float* x, const float* w, float* y, int H, int W, int Cin, int tilesX) {
  constexpr int BK = 32, CK = 36, IH = 10, IW = 18, BN = 32;
  __shared__ __attribute__((aligned(16))) float sA[IH * IW * CK];
  __shared__ __attribute__((aligned(16))) float sB[2][BN * CK];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5, wave = tid >> 6;
  const int fy = wave * 2 + li / 16, fx = li % 16;
  const int ntiles_total = (H / 8) * tilesX;
  for (int tile = blockIdx.x; tile < ntiles_total; tile += gridDim.x) {
  const int ty0 = (tile / tilesX) * 8, tx0 = (tile % tilesX) * 16;
  f32x16 acc, acc2;
  for (int r = 0; r < 16; ++r) { acc[r] = 0.f; acc2[r] = 0.f; }
  const int ac4 = tid % 8;
  float4 areg[6], breg;
  const int wrow = 9 * Cin;
  auto loadB = [&](int kc, int tap) {
    const int row = tid / 8;
    if (FEAT & 2) breg = *reinterpret_cast<const float4*>(w + (size_t)(row % 32) * wrow + tap * Cin + kc + ac4 * 4);
    else if (FEAT & 32) breg = *reinterpret_cast<const float4*>(w + ((tap * 7 + kc + blockIdx.x) % 64) * 1024 + tid * 4);
    else breg = *reinterpret_cast<const float4*>(w + ((tap * 7 + kc) % 64) * 1024 + tid * 4);
  };
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
    for (int i = 0; i < 6; ++i) {
      const int idx = tid + i * 256, pix = idx / 8;
      if (idx < IH * IW * 8) *reinterpret_cast<float4*>(sA + pix * CK + ac4 * 4) = areg[i];
    }
  };
  loadB(0, 0);
  prefetchA(0);
  if (!(FEAT & 1)) { writeA(); }
  for (int kc = 0; kc < Cin; kc += BK) {
    __syncthreads();
    if (FEAT & 1) writeA();
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const int buf = tap & 1;
      if (tid < 256) *reinterpret_cast<float4*>(sB[buf] + (tid / 8) * CK + ac4 * 4) = breg;
      { int ntap = tap + 1, nkc = kc; if (ntap == 9) { ntap = 0; nkc = kc + BK; } if (nkc < Cin) loadB(nkc, ntap); }
      if ((FEAT & 1) && tap == 0 && kc + BK < Cin) prefetchA(kc + BK);
      __syncthreads();
      int aoff;
      if (FEAT & 8) { const int dy = tap / 3, dx = tap % 3; aoff = ((fy + dy) * IW + fx + dx) * CK + lh * 4; }
      else aoff = (fy * IW + fx) * CK + lh * 4;
      const float* sBb = sB[buf] + li * CK + lh * 4;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const float4 a = *reinterpret_cast<const float4*>(sA + aoff + kb * 8);
        const float4 b = *reinterpret_cast<const float4*>(sBb + kb * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc2, 0, 0, 0);
      }
    }
  }
  if (FEAT & 4) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int oy = ty0 + wave * 2 + row / 16, ox = tx0 + row % 16;
      y[((size_t)oy * W + ox) * 32 + li] = acc[r] + acc2[r];
    }
  } else {
    float t = 0.f;
    for (int r = 0; r < 16; ++r) t += acc[r] + acc2[r];
    y[(size_t)tile * 256 + tid] = t;
  }
  __syncthreads();
  }
}

template <int FEAT>
void run(const char* name, const float* x, const float* w, float* y, int H, int W, int Cin, int grid = 0) {
  const int tilesX = W / 16, tiles_all = (H / 8) * tilesX;
  const int tiles = grid > 0 && grid < tiles_all ? grid : tiles_all;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((conv_like<FEAT>), dim3(tiles), dim3(256), 0, 0, x, w, y, H, W, Cin, tilesX);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  ms /= 10;
  const double flops = 2.0 * Cin * 9 * 32 * (double)H * W;
  printf("H=%4d Cin=%3d feat %2d %-44s: %7.1f us %6.1f TFLOP/s (%d WGs)\n", H, Cin, FEAT, name, ms * 1e3, flops / ms / 1e9, tiles);
}

int main() {
  float *x, *w, *y;
  hipMalloc(&x, (size_t)512 * 512 * 64 * 4); hipMalloc(&w, (size_t)64 * 9 * 512 * 4 + (1 << 20)); hipMalloc(&y, (size_t)512 * 512 * 32 * 4);
  hipMemset(x, 0x3c, (size_t)512 * 512 * 64 * 4); hipMemset(w, 0x3c, (size_t)64 * 9 * 512 * 4 + (1 << 20));
  struct { int H, C; } cases[] = {{512, 64}, {256, 64}, {128, 256}, {64, 512}};
  for (auto c : cases) {
    run<0>("bare loop (tiny L2 weights, no A staging)", x, w, y, c.H, c.H, c.C);
    run<8>("+ per-tap window offsets", x, w, y, c.H, c.H, c.C);
    run<32>("bare loop, per-WG staggered weight blocks", x, w, y, c.H, c.H, c.C);
    run<32>("bare staggered, persistent grid 1024", x, w, y, c.H, c.H, c.C, 1024);
    run<10>("+ real strided weight loads", x, w, y, c.H, c.H, c.C);
    run<11>("+ per-chunk A staging (reg prefetch)", x, w, y, c.H, c.H, c.C);
    run<15>("+ epilogue stores (= real kernel)", x, w, y, c.H, c.H, c.C);
    run<5>("A staging + epilogue, tiny weights", x, w, y, c.H, c.H, c.C);
    run<15>("real kernel, persistent grid 256", x, w, y, c.H, c.H, c.C, 256);
    run<15>("real kernel, persistent grid 512", x, w, y, c.H, c.H, c.C, 512);
    run<15>("real kernel, persistent grid 1024", x, w, y, c.H, c.H, c.C, 1024);
    run<0>("bare loop, persistent grid 512", x, w, y, c.H, c.H, c.C, 512);
    run<0>("bare loop, persistent grid 1024", x, w, y, c.H, c.H, c.C, 1024);
  }
  return 0;
}
