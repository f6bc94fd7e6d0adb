// Harness: the production fused-SPADE kernel (k_igemm<..., SPADE=true>) on the 512x512 launches,
// timed against its own HBM byte count.  -DRIB_EXP bits as in igemm_harness.hip.
#include "../../render-in-between_amd/csrc/kernels.hip.h"
#include <cstdio>
using namespace rib;

template <int FRW, int WM, int WN, int MF, int NF, int BK>
void run(const char* name, int H, int W, int Ccond, int C, int nsets, float* cond, float* w, float* bias, float* xm, float* sc, float* y0, float* y1) {
  typedef IgemmGeom<FRW, WM, WN, MF, NF, BK, 1, 1, false> G;
  IgemmParams p{};
  const int npad = ((nsets * C + 31) / 32) * 64;
  p.x = cond; p.Hin = H; p.Win = W; p.xC = Ccond; p.Cin = Ccond;
  p.w = w; p.bias = bias; p.CoutPad = npad; p.Hout = H; p.Wout = W;
  p.tilesX = (W + G::TW - 1) / G::TW; p.tilesY = (H + G::TH - 1) / G::TH; p.xcd_chunk = 0; p.ksplit = 1;
  p.xm = xm; p.xmC = C; p.xm_ups = 0; p.m_scale = sc; p.m_shift = sc + 1024; p.m_ld = C; p.C = C; p.nsets = nsets;
  p.ys0 = y0; p.ys1 = y1; p.act0 = 1; p.act1 = 0;
  dim3 grid(p.tilesX * p.tilesY, (npad + G::BN - 1) / G::BN, 1);
  auto fn = k_igemm<FRW, WM, WN, MF, NF, BK, 1, 1, false, true, false>;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(fn, grid, dim3(256), 0, 0, p);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  ms /= 20;
  const double bytes = (double)H * W * 4.0 * (Ccond * grid.y + C + nsets * C);
  printf("exp %d %-28s %dx%d cond %d C %d sets %d grid %u,%u: %7.1f us  %6.2f TB/s (alg. bytes %.0f MB)\n", RIB_EXP, name, H, W, Ccond, C, nsets,
         grid.x, grid.y, ms * 1e3, bytes / ms / 1e9, bytes / 1e6);
}

int main() {
  float *cond, *w, *bias, *xm, *sc, *y0, *y1;
  hipMalloc(&cond, (size_t)512 * 512 * 128 * 4); hipMalloc(&w, 1 << 22); hipMalloc(&bias, 1 << 16); hipMalloc(&xm, (size_t)512 * 512 * 64 * 4);
  hipMalloc(&sc, 1 << 16); hipMalloc(&y0, (size_t)512 * 512 * 64 * 4); hipMalloc(&y1, (size_t)512 * 512 * 64 * 4);
  hipMemset(cond, 0x3c, (size_t)512 * 512 * 128 * 4); hipMemset(w, 0x3c, 1 << 22); hipMemset(bias, 0, 1 << 16);
  hipMemset(xm, 0x3c, (size_t)512 * 512 * 64 * 4); hipMemset(sc, 0x3c, 1 << 16);
  run<16, 4, 1, 1, 2, 32>("8x16 BN64 BK32", 512, 512, 64, 16, 1, cond, w, bias, xm, sc, y0, y1);   // down_0.1 / up_0.1
  run<16, 4, 1, 1, 2, 32>("8x16 BN64 BK32", 512, 512, 64, 16, 2, cond, w, bias, xm, sc, y0, y1);   // down_0.0
  run<16, 4, 1, 1, 2, 32>("8x16 BN64 BK32", 512, 512, 64, 32, 2, cond, w, bias, xm, sc, y0, y1);   // up_0.0
  run<16, 4, 1, 1, 2, 64>("8x16 BN64 BK64", 512, 512, 64, 16, 1, cond, w, bias, xm, sc, y0, y1);
  run<16, 4, 1, 2, 2, 32>("16x16 BN64 BK32", 512, 512, 64, 16, 1, cond, w, bias, xm, sc, y0, y1);
  run<16, 4, 1, 2, 2, 64>("16x16 BN64 BK64", 512, 512, 64, 16, 1, cond, w, bias, xm, sc, y0, y1);
  run<16, 4, 1, 1, 2, 32>("8x16 BN64 BK32", 256, 256, 128, 32, 1, cond, w, bias, xm, sc, y0, y1);  // down_1.1 / up_1.1
  return 0;
}
