// Harness: time the REAL k_igemm on one layer shape, with elimination switches compiled into a copy of
// kernels.hip.h (-DRIB_EXP=<bits>):
//   bit0  filter loads all hit one 4 KB block (no L2 / MALL misses, no address math)
//   bit1  no per-chunk input staging after the first chunk (prefetchA / writeA skipped)
//   bit2  no epilogue stores
//   2048  s_memtime stamps: where a wave's time goes in the main loop (chunk head | filter store or fill issue | waiting at
//         the slice barrier | MFMA section), averaged over the waves of the launch
// hipcc -O3 --offload-arch=gfx950 -std=c++17 -DRIB_EXP=2048 -I render-in-between_amd/csrc tools/probes/igemm_harness.hip -o tools/probes/bin/igemm_e2048
#include "kernels.hip.h"
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
#include <cmath>
using namespace rib;

static std::vector<float> g_prev;
// cold-start experiment: what a launch pays in the frame (its filters and code come from HBM: the frame cycles ~310 MB of
// filters through a 256 MB memory-side cache) and what touching the filters beforehand, from ANOTHER kernel, gives back
__global__ void k_flush(float4* p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void k_touch(const float4* p, size_t n4, float* sink) {      // one 16-byte load per 128-byte line
  float acc = 0.f;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n4; i += (size_t)gridDim.x * blockDim.x * 8) acc += p[i].x;
  if (acc == 123.456f) *sink = acc;
}
static float4* g_flush = nullptr;
static const size_t FLUSH4 = (size_t)640 << 16;      // 640 MB
template <int FRW, int WM, int WN, int MF, int NF, int BK, int STRIDE, int KW, int TB, int DMA>
void run(const char* name, int Hout, int Wout, int Cin, int Cout, int ksplit, float* x, float* w, float* bias, float* y, float* slab, float* zeros) {
  typedef IgemmGeom<FRW, WM, WN, MF, NF, BK, STRIDE, 3, false, KW, TB, 0, DMA> G;
  IgemmParams p;
  memset(&p, 0, sizeof p);
  const int Hin = Hout * STRIDE, Win = Wout * STRIDE;
  p.x = x; p.Hin = Hin; p.Win = Win; p.xC = Cin; p.Cin = Cin;
  p.w = w; p.bias = bias; p.CoutPad = Cout; p.Hout = Hout; p.Wout = Wout;
  p.tilesX = (Wout + G::TW - 1) / G::TW; p.tilesY = (Hout + G::TH - 1) / G::TH; p.xcd_chunk = 0;
  p.y = y; p.yC = Cout; p.yoff = 0; p.Cout = Cout; p.act = 0; p.ksplit = ksplit; p.slab = ksplit > 1 ? slab : nullptr;
  p.w_mod = 0; p.zeros = zeros;
  dim3 grid(p.tilesX * p.tilesY, Cout / G::BN, ksplit);
  auto fn = k_igemm<FRW, WM, WN, MF, NF, BK, STRIDE, 3, false, false, 0, false, false, KW, TB, DMA>;
  constexpr int THREADS = 256 * KW * ((DMA & 4) ? 2 : 1);
  {   // bit-compare with the previous run of the same problem (the unspecialised kernel is run first)
    std::vector<float>& prev = g_prev;
    std::vector<float> cur((size_t)Hout * Wout * Cout);
    (void)hipMemset(y, 0, cur.size() * 4);
    hipLaunchKernelGGL(fn, grid, dim3(THREADS), 0, 0, p);
    (void)hipMemcpy(cur.data(), y, cur.size() * 4, hipMemcpyDeviceToHost);
    if (DMA & 4) {
      size_t bad = prev.size() == cur.size() ? 0 : (size_t)-1;
      if (!bad) for (size_t i = 0; i < cur.size(); ++i) bad += memcmp(&cur[i], &prev[i], 4) != 0;
      double sum = 0; for (float v : cur) sum += fabs((double)v);
      printf("      warp-specialised vs the run before: %zu mismatching outputs of %zu (sum |y| %.4g)\n", bad, cur.size(), sum);
    }
    prev.swap(cur);
  }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fn, grid, dim3(THREADS), 0, 0, p);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1);
  }
  ms /= 20;
  if (!(DMA & 4) && g_flush) {
    float cold = 0, pref = 0, prefx = 0;
    const size_t wbytes = (size_t)Cout * 9 * Cin * 4, xbytes = (size_t)Hin * Win * Cin * 4;
    for (int rep = 0; rep < 5; ++rep) {
      float t;
      k_flush<<<2048, 256>>>(g_flush, FLUSH4);
      (void)hipEventRecord(e0); hipLaunchKernelGGL(fn, grid, dim3(THREADS), 0, 0, p); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&t, e0, e1); cold += t / 5;
      k_flush<<<2048, 256>>>(g_flush, FLUSH4);
      k_touch<<<256, 256>>>((const float4*)w, wbytes / 16, slab);
      (void)hipEventRecord(e0); hipLaunchKernelGGL(fn, grid, dim3(THREADS), 0, 0, p); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&t, e0, e1); pref += t / 5;
      k_flush<<<2048, 256>>>(g_flush, FLUSH4);
      k_touch<<<256, 256>>>((const float4*)w, wbytes / 16, slab);
      k_touch<<<256, 256>>>((const float4*)x, xbytes / 16, slab);
      (void)hipEventRecord(e0); hipLaunchKernelGGL(fn, grid, dim3(THREADS), 0, 0, p); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      (void)hipEventElapsedTime(&t, e0, e1); prefx += t / 5;
    }
    printf("      single launch, events around it: after a 640 MB flush %.1f us | filters (%.1f MB) touched by another kernel first %.1f us | filters and input (%.1f MB) touched %.1f us | warm, back to back %.1f us\n",
           cold * 1e3, wbytes / 1e6, pref * 1e3, xbytes / 1e6, prefx * 1e3, ms * 1e3);
  }
  const double flops = 2.0 * Cin * 9 * Cout * (double)Hout * Wout;
  printf("exp %d %-40s out %dx%d %d->%d s%d ksplit %d grid %4d x %d waves: %7.1f us %6.1f TFLOP/s\n", RIB_EXP, name, Hout, Wout, Cin, Cout, STRIDE, ksplit,
         grid.x * grid.y * grid.z, 4 * KW, ms * 1e3, flops / ms / 1e9);
#if RIB_EXP & 2048
  {
    const size_t nw = (size_t)grid.x * grid.y * grid.z * 4 * KW;
    std::vector<long long> hs2(2 * nw * 8);
    long long* d; (void)hipMemcpyFromSymbol(&d, HIP_SYMBOL(g_igemm_stamps), sizeof d);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(hs2.data(), d, hs2.size() * 8, hipMemcpyDeviceToHost);
    std::vector<long long> hs(hs2.begin(), hs2.begin() + nw * 8);
    double a[5] = {0, 0, 0, 0, 0};
    long long tmin = hs[5], tmax = 0;
    for (size_t i = 0; i < nw; ++i) {
      for (int k = 0; k < 5; ++k) a[k] += (double)hs[i * 8 + k] / nw;
      tmin = std::min(tmin, hs[i * 8 + 5]); tmax = std::max(tmax, hs[i * 8 + 5] + hs[i * 8 + 4]);
    }
    const double chunks = (double)Cin / BK / ksplit;
    const double mfma_cycles = chunks * 9 * (BK / 2) * MF * NF * 64.0 / KW;      // the wave's MFMAs at 64 cycles each
    printf("      per wave (cycles): main loop + prologue %.0f (launch span %lld) | head %.0f | store / fill issue %.0f | barrier wait %.0f | MFMA section %.0f, of which bare MFMA time %.0f\n",
           a[4], tmax - tmin, a[0], a[1], a[2], a[3], mfma_cycles);
    if ((DMA & 4) && TB == 3) {
      double b[5] = {0, 0, 0, 0, 0};
      for (size_t i = nw; i < 2 * nw; ++i) for (int k = 0; k < 5; ++k) b[k] += (double)hs2[i * 8 + k] / nw;
      printf("      loader waves: lifetime %.0f | filter store (with the wait for its loads) %.0f | tile commit + prefetch %.0f | load issue %.0f | barrier wait %.0f\n", b[4], b[0], b[1], b[3], b[2]);
    }
  }
#endif
}

int main() {
  float *x, *w, *bias, *y, *slab, *zeros;
  (void)hipMalloc(&x, (size_t)512 * 512 * 64 * 4); (void)hipMalloc(&w, (size_t)512 * 9 * 512 * 4); (void)hipMalloc(&bias, 4096);
  (void)hipMalloc(&y, (size_t)512 * 512 * 64 * 4); (void)hipMalloc(&slab, (size_t)64 << 20); (void)hipMalloc(&zeros, 4096);
  (void)hipMemset(zeros, 0, 4096); (void)hipMemset(bias, 0, 4096);
  {   // operands: small-magnitude pseudo-random values (all-equal or uniform +-1 operands run at other clocks than real tensors)
    std::vector<float> h((size_t)512 * 512 * 64);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; const float u = (float)(s >> 8) / 16777216.f - 0.5f; v = u * u * u * 2.f; }
    (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < 9; ++i) (void)hipMemcpy(w + (size_t)i * 512 * 512, h.data() + (size_t)i * 1000003 % (h.size() - 512 * 512), (size_t)512 * 512 * 4, hipMemcpyHostToDevice);
  }
  (void)hipMalloc(&g_flush, FLUSH4 * 16);
#if RIB_EXP & 2048
  long long* st; (void)hipMalloc(&st, (size_t)64 << 20); (void)hipMemset(st, 0, (size_t)64 << 20);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_igemm_stamps), &st, sizeof st);
#endif
  // the tuned picks of the 512x512 frame (profiles/r03_prof_ops_512.txt)
  run<16, 4, 1, 1, 2, 16, 2, 1, 1, 3>("ref_embedding.down_1 (8x16 BN64 BK16 DMA)", 128, 128, 128, 256, 1, x, w, bias, y, slab, zeros);
  run<16, 4, 1, 1, 2, 16, 2, 1, 1, 7>("   the same, warp-specialised", 128, 128, 128, 256, 1, x, w, bias, y, slab, zeros);
  run<16, 4, 1, 1, 2, 16, 2, 1, 3, 0>("ref_embedding.down_1 (8x16 BN64 BK16 tb3)", 128, 128, 128, 256, 1, x, w, bias, y, slab, zeros);
  run<16, 4, 1, 1, 2, 16, 2, 1, 3, 4>("   the same, warp-specialised", 128, 128, 128, 256, 1, x, w, bias, y, slab, zeros);
  run<8, 2, 2, 1, 2, 16, 2, 1, 1, 3>("ref_embedding.down_0 (8x8 BN128 BK16 DMA)", 256, 256, 64, 128, 1, x, w, bias, y, slab, zeros);
  run<8, 2, 2, 1, 2, 16, 2, 1, 1, 7>("   the same, warp-specialised", 256, 256, 64, 128, 1, x, w, bias, y, slab, zeros);
  run<16, 4, 1, 1, 2, 8, 2, 1, 9, 3>("ref_embedding.down_2 (8x16 BN64 BK8 DMA9)", 64, 64, 256, 512, 1, x, w, bias, y, slab, zeros);
  run<16, 4, 1, 1, 2, 16, 2, 1, 3, 0>("ref_embedding.down_2 (8x16 BN64 BK16 tb3)", 64, 64, 256, 512, 1, x, w, bias, y, slab, zeros);
  run<16, 4, 1, 1, 2, 16, 2, 1, 3, 4>("   the same, warp-specialised", 64, 64, 256, 512, 1, x, w, bias, y, slab, zeros);
  run<16, 4, 1, 1, 2, 16, 1, 1, 3, 0>("down_1.conv_block_1 (8x16 BN64 BK16 tb3)", 256, 256, 64, 64, 1, x, w, bias, y, slab, zeros);
  run<16, 4, 1, 1, 2, 16, 1, 1, 3, 4>("   the same, warp-specialised", 256, 256, 64, 64, 1, x, w, bias, y, slab, zeros);
  run<16, 4, 1, 1, 1, 32, 1, 2, 3, 0>("down_2.conv_block_1 (8x16 BN32 BK32 kw2 tb3)", 128, 128, 128, 128, 1, x, w, bias, y, slab, zeros);
  run<16, 4, 1, 1, 1, 32, 1, 1, 3, 0>("down_2.conv_block_1 (8x16 BN32 BK32 tb3)", 128, 128, 128, 128, 1, x, w, bias, y, slab, zeros);
  run<16, 4, 1, 1, 1, 32, 1, 1, 3, 4>("   the same, warp-specialised", 128, 128, 128, 128, 1, x, w, bias, y, slab, zeros);
  run<16, 4, 1, 1, 2, 32, 1, 1, 3, 0>("down_2.conv_block_1 (8x16 BN64 BK32 tb3)", 128, 128, 128, 128, 1, x, w, bias, y, slab, zeros);
  run<16, 4, 1, 1, 2, 32, 1, 1, 3, 4>("   the same, warp-specialised", 128, 128, 128, 128, 1, x, w, bias, y, slab, zeros);
  run<16, 4, 1, 1, 1, 32, 1, 1, 1, 0>("up_1.conv_block_1 (8x16 BN32 BK32 tb1)", 256, 256, 64, 32, 1, x, w, bias, y, slab, zeros);
  run<16, 4, 1, 1, 1, 32, 1, 1, 3, 0>("up_1.conv_block_1 (8x16 BN32 BK32 tb3)", 256, 256, 64, 32, 1, x, w, bias, y, slab, zeros);
  run<16, 4, 1, 1, 1, 32, 1, 1, 3, 4>("   the same, warp-specialised", 256, 256, 64, 32, 1, x, w, bias, y, slab, zeros);
  run<16, 4, 1, 1, 1, 16, 1, 1, 3, 0>("up_0.conv_block_1 (8x16 BN32 BK16 tb3)", 512, 512, 32, 32, 1, x, w, bias, y, slab, zeros);
  run<16, 4, 1, 1, 1, 16, 1, 1, 3, 4>("   the same, warp-specialised", 512, 512, 32, 32, 1, x, w, bias, y, slab, zeros);
  return 0;
}
