// Harness: the phase-decomposed upsample convolution (k_igemm, UPS) on the three up_flow shapes of the
// mask network at 512x512, per tile variant and split-K factor; -DRIB_UPS_ROLL=0/1 picks the loop structure.
// hipcc -O3 --offload-arch=gfx950 -DRIB_UPS_ROLL=0 tools/probes/ups_harness.hip -o /tmp/ups_harness
#include "../../render-in-between_amd/csrc/kernels.hip.h"
#include <cstdio>
using namespace rib;

template <int FRW, int WM, int WN, int MF, int NF, int BK, bool PRO>
void run(const char* name, int Hs, int Ws, int Cin, int Cout, int ksplit, float* x, float* w, float* bias, float* y, float* slab, float* sc) {
  typedef IgemmGeom<FRW, WM, WN, MF, NF, BK, 1, 3, true> G;
  IgemmParams p{};
  p.x = x; p.Hin = Hs; p.Win = Ws; p.xC = Cin; p.Cin = Cin;
  p.w = w; p.bias = bias; p.CoutPad = Cout; p.Hout = 2 * Hs; p.Wout = 2 * Ws;
  p.tilesX = (Ws + G::TW - 1) / G::TW; p.tilesY = (Hs + G::TH - 1) / G::TH;
  p.xcd_chunk = (p.tilesX * p.tilesY) % 8 == 0 ? p.tilesX * p.tilesY / 8 : 0;
  p.y = y; p.yC = Cout; p.yoff = 0; p.Cout = Cout; p.act = 0; p.ksplit = ksplit; p.slab = ksplit > 1 ? slab : nullptr;
  if (PRO) { p.pro_scale = sc; p.pro_shift = sc + 1024; p.pro_ld = 1024; p.pro_lrelu = 1; }
  dim3 grid(p.tilesX * p.tilesY, (Cout + G::BN - 1) / G::BN, ksplit);
  auto fn = k_igemm<FRW, WM, WN, MF, NF, BK, 1, 3, true, false, false, false, PRO>;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fn, grid, dim3(256), 0, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  ms /= 20;
  const double flops = 2.0 * Cin * 9 * Cout * (double)Hs * Ws * 4;   // algorithmic (nine-tap) count
  printf("roll %d %-24s src %3dx%-3d %3d->%-3d ksplit %d grid %4d: %7.1f us %6.1f TFLOP/s (algorithmic)\n", RIB_UPS_ROLL, name, Hs, Ws,
         Cin, Cout, ksplit, grid.x * grid.y * grid.z, ms * 1e3, flops / ms / 1e9);
}

int main() {
  float *x, *w, *bias, *y, *slab, *sc;
  hipMalloc(&x, (size_t)256 * 256 * 64 * 4); hipMalloc(&w, (size_t)128 * 16 * 256 * 4); hipMalloc(&bias, 4096); hipMalloc(&sc, 8192);
  hipMalloc(&y, (size_t)512 * 512 * 32 * 4); hipMalloc(&slab, (size_t)256 << 20);
  hipMemset(x, 0x3c, (size_t)256 * 256 * 64 * 4); hipMemset(w, 0x3c, (size_t)128 * 16 * 256 * 4); hipMemset(bias, 0, 4096); hipMemset(sc, 0x3c, 8192);
#define SHAPES(FRW, WM, WN, MF, NF, BK, NAME)                                                               \
  run<FRW, WM, WN, MF, NF, BK, false>(NAME " raw", 64, 64, 256, 128, 1, x, w, bias, y, slab, sc);           \
  run<FRW, WM, WN, MF, NF, BK, false>(NAME " raw", 64, 64, 256, 128, 2, x, w, bias, y, slab, sc);           \
  run<FRW, WM, WN, MF, NF, BK, true>(NAME " pro", 128, 128, 128, 64, 1, x, w, bias, y, slab, sc);           \
  run<FRW, WM, WN, MF, NF, BK, true>(NAME " pro", 128, 128, 128, 64, 2, x, w, bias, y, slab, sc);           \
  run<FRW, WM, WN, MF, NF, BK, true>(NAME " pro", 256, 256, 64, 32, 1, x, w, bias, y, slab, sc);
  SHAPES(16, 4, 1, 1, 1, 32, "8x16 BN32 BK32")
  SHAPES(16, 4, 1, 1, 1, 16, "8x16 BN32 BK16")
  SHAPES(16, 4, 1, 1, 2, 32, "8x16 BN64 BK32")
  SHAPES(16, 4, 1, 2, 1, 32, "16x16 BN32 BK32")
  SHAPES(8, 2, 2, 1, 1, 32, "8x8 BN64 BK32")
  SHAPES(8, 2, 2, 1, 1, 16, "8x8 BN64 BK16")
  return 0;
}
