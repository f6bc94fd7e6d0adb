// Harness (round 2): the production 1x1 k_igemm variants (the SPADE / Winograd-domain GEMM core) on LONG launches, to
// separate the steady-state rate of the chunk loop from ramp / tail effects, with the elimination switches of
// igemm_harness.hip compiled in (-DRIB_EXP: bit0 filter loads hit one 4 KB block, bit1 no input staging after the first
// chunk).   hipcc -O3 --offload-arch=gfx950 -DRIB_EXP=0 tools/probes/gemm1x1_harness.hip -o tools/probes/bin/gemm1x1_e0
#include "../../render-in-between_amd/csrc/kernels.hip.h"
#include <cstdio>
using namespace rib;

template <int FRW, int WM, int WN, int MF, int NF, int BK, int KW = 1, int TB = 1>
void run(const char* name, int H, int W, int Cin, int Cout, int G, float* x, float* w, float* bias, float* y) {
  typedef IgemmGeom<FRW, WM, WN, MF, NF, BK, 1, 1, false, KW, TB> Geo;
  IgemmParams p{};
  p.x = x; p.Hin = H; p.Win = W; p.xC = Cin; p.Cin = Cin;
  p.w = w; p.bias = bias; p.CoutPad = Cout; p.Hout = H; p.Wout = W;
  p.tilesX = (W + Geo::TW - 1) / Geo::TW; p.tilesY = (H + Geo::TH - 1) / Geo::TH; p.xcd_chunk = 0;
  p.y = y; p.yC = Cout; p.yoff = 0; p.Cout = Cout; p.act = 0; p.ksplit = 1;
  p.w_mod = G > 1 ? G : 0; p.w_stride = (unsigned)((size_t)Cout * Cin);
  dim3 grid(p.tilesX * p.tilesY, Cout / Geo::BN, G);
  auto fn = k_igemm<FRW, WM, WN, MF, NF, BK, 1, 1, false, false, 0, false, false, KW, TB>;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  const int it = 10;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < it; ++i) hipLaunchKernelGGL(fn, grid, dim3(256 * KW), 0, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  ms /= it;
  const double flops = 2.0 * Cin * Cout * (double)H * W * G;
  printf("exp %d %-26s M %7d K %3d N %4d G %2d grid %6d: %8.1f us %6.1f TFLOP/s\n", RIB_EXP, name, H * W, Cin, Cout, G,
         grid.x * grid.y * grid.z, ms * 1e3, flops / ms / 1e9);
}

int main() {
  float *x, *w, *bias, *y;
  const size_t n = (size_t)1 << 28;   // 1 GiB each
  hipMalloc(&x, n); hipMalloc(&y, n); hipMalloc(&w, 64 << 20); hipMalloc(&bias, 1 << 16);
  hipMemset(x, 0x3c, n); hipMemset(w, 0x3c, 64 << 20); hipMemset(bias, 0, 1 << 16);
  // long launches: 512x512 "pixels", K = N = 256 (34 GFLOP) and K = 512
  run<8, 2, 2, 1, 1, 64>("8x8 BN64 BK64", 512, 512, 256, 256, 1, x, w, bias, y);
  run<8, 2, 2, 1, 1, 64, 2>("8x8 BN64 BK64 kw2", 512, 512, 256, 256, 1, x, w, bias, y);
  run<8, 2, 2, 1, 1, 64, 1, 2>("8x8 BN64 BK64 tb2", 512, 512, 256, 256, 1, x, w, bias, y);
  run<16, 4, 1, 1, 2, 32>("8x16 BN64 BK32", 512, 512, 256, 256, 1, x, w, bias, y);
  run<16, 4, 1, 1, 2, 64>("8x16 BN64 BK64", 512, 512, 256, 256, 1, x, w, bias, y);
  run<16, 4, 1, 2, 2, 32>("16x16 BN64 BK32", 512, 512, 256, 256, 1, x, w, bias, y);
  run<16, 4, 1, 1, 1, 32>("8x16 BN32 BK32", 512, 512, 256, 256, 1, x, w, bias, y);
  run<8, 1, 4, 1, 1, 32>("4x8 BN128 BK32", 512, 512, 256, 256, 1, x, w, bias, y);
  run<8, 2, 2, 1, 1, 64>("8x8 BN64 BK64", 256, 256, 512, 512, 1, x, w, bias, y);
  // the frame's shapes
  run<8, 2, 2, 1, 1, 64>("8x8 BN64 BK64", 16, 16, 256, 256, 36, x, w, bias, y);      // res_flow wino4
  run<8, 1, 4, 1, 1, 32, 1, 2>("4x8 BN128 BK32 tb2", 16, 16, 256, 256, 36, x, w, bias, y);
  run<8, 2, 2, 1, 1, 64, 2>("8x8 BN64 BK64 kw2", 32, 32, 256, 256, 16, x, w, bias, y);   // res_flow wino2
  return 0;
}
