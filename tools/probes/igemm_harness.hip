// Harness: time the REAL k_igemm on one layer shape, with elimination switches compiled into a copy of
// kernels.hip.h (-DRIB_EXP=<bits>, see tools/probes/igemm_harness.md in DESIGN notes):
//   bit0  filter loads all hit one 4 KB block (no L2 / MALL misses, no address math)
//   bit1  no per-chunk input staging after the first chunk (prefetchA / writeA skipped)
//   bit2  no epilogue stores
// hipcc -O3 --offload-arch=gfx950 -DRIB_EXP=0 tools/probes/igemm_harness.hip -o /tmp/igemm_harness
#include "../../render-in-between_amd/csrc/kernels.hip.h"
#include <cstdio>
#include <vector>
using namespace rib;

template <int FRW, int WM, int WN, int MF, int NF, int BK, bool AUX, bool PRO, int KW = 1>
void run(const char* name, int H, int W, int Cin, int Cout, int ksplit, float* x, float* w, float* bias, float* y, float* slab) {
  typedef IgemmGeom<FRW, WM, WN, MF, NF, BK, 1, 3, false, KW> G;
  IgemmParams p{};
  p.x = x; p.Hin = H; p.Win = W; p.xC = Cin; p.Cin = Cin;
  p.w = w; p.bias = bias; p.CoutPad = Cout; p.Hout = H; p.Wout = W;
  p.tilesX = (W + G::TW - 1) / G::TW; p.tilesY = (H + G::TH - 1) / G::TH; p.xcd_chunk = 0;
  p.y = y; p.yC = Cout; p.yoff = 0; p.Cout = Cout; p.act = 0; p.ksplit = ksplit; p.slab = ksplit > 1 ? slab : nullptr;
  dim3 grid(p.tilesX * p.tilesY, Cout / G::BN, ksplit);
  auto fn = k_igemm<FRW, WM, WN, MF, NF, BK, 1, 3, false, false, false, AUX, PRO, KW>;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fn, grid, dim3(256 * KW), 0, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  ms /= 20;
  const double flops = 2.0 * Cin * 9 * Cout * (double)H * W;
  printf("exp %d %-34s %dx%d %d->%d ksplit %d grid %4d: %7.1f us %6.1f TFLOP/s\n", RIB_EXP, name, H, W, Cin, Cout, ksplit,
         grid.x * grid.y * grid.z, ms * 1e3, flops / ms / 1e9);
}

int main() {
  float *x, *w, *bias, *y, *slab;
  hipMalloc(&x, (size_t)512 * 512 * 64 * 4); hipMalloc(&w, (size_t)512 * 9 * 512 * 4); hipMalloc(&bias, 4096);
  hipMalloc(&y, (size_t)512 * 512 * 64 * 4); hipMalloc(&slab, (size_t)64 << 20);
  hipMemset(x, 0x3c, (size_t)512 * 512 * 64 * 4); hipMemset(w, 0x3c, (size_t)512 * 9 * 512 * 4); hipMemset(bias, 0, 4096);
  run<16, 4, 1, 1, 1, 32, false, false>("8x16 BN32 BK32 lean", 64, 64, 256, 256, 2, x, w, bias, y, slab);
  run<16, 4, 1, 1, 1, 32, false, false>("8x16 BN32 BK32 lean", 64, 64, 256, 256, 1, x, w, bias, y, slab);
  run<16, 4, 1, 1, 1, 32, false, false>("8x16 BN32 BK32 lean", 64, 64, 256, 256, 4, x, w, bias, y, slab);
  run<16, 4, 1, 1, 1, 32, false, false>("8x16 BN32 BK32 lean", 32, 32, 512, 512, 4, x, w, bias, y, slab);
  run<16, 4, 1, 1, 1, 32, false, false>("8x16 BN32 BK32 lean", 128, 128, 128, 128, 1, x, w, bias, y, slab);
  run<16, 4, 1, 1, 1, 32, false, false>("8x16 BN32 BK32 lean", 256, 256, 64, 64, 1, x, w, bias, y, slab);
  run<16, 4, 1, 1, 1, 32, false, false, 2>("8x16 BN32 BK32 lean KW2", 64, 64, 256, 256, 1, x, w, bias, y, slab);
  run<16, 4, 1, 1, 1, 32, false, false, 2>("8x16 BN32 BK32 lean KW2", 64, 64, 256, 256, 2, x, w, bias, y, slab);
  run<16, 4, 1, 1, 1, 32, false, false, 4>("8x16 BN32 BK32 lean KW4", 64, 64, 256, 256, 1, x, w, bias, y, slab);
  run<16, 4, 1, 1, 1, 32, false, false, 2>("8x16 BN32 BK32 lean KW2", 32, 32, 512, 512, 2, x, w, bias, y, slab);
  run<16, 4, 1, 1, 1, 32, false, false, 4>("8x16 BN32 BK32 lean KW4", 32, 32, 512, 512, 1, x, w, bias, y, slab);
  run<16, 4, 1, 1, 2, 32, false, false>("8x16 BN64 BK32 lean", 64, 64, 256, 256, 2, x, w, bias, y, slab);
  run<16, 4, 1, 2, 1, 32, false, false>("16x16 BN32 BK32 lean", 64, 64, 256, 256, 2, x, w, bias, y, slab);
  return 0;
}
