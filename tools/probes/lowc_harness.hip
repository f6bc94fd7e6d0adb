// k_conv_lowc (the four first-layer convolutions on the caller's NCHW tensors) with parts switched off, to see which phase
// its time is: build once per RIB_EXP value (0 full | 64 no MFMA | 128 no stores | 256 no gather loads | 512 no fp64 statistics
// | 1024 s_memtime stamps of the phases), and a persistent double-buffered LDS-DMA version of it (k_conv_lowc_dma, below:
// bit-identical, not faster - DESIGN "Round 3" - so it lives here and not in the library)
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -DRIB_EXP=64 -I render-in-between_amd/csrc tools/probes/lowc_harness.hip -o tools/probes/bin/lowc_e64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include "kernels.hip.h"
namespace rib {
__device__ const float* g_lowc_zeros;      // >= 4 bytes of zeros: the source of every padding element of k_conv_lowc_dma
// ---------------------------------------------------------------------------------------------
// k_conv_lowc_dma (round 3): k_conv_lowc as a persistent, double-buffered kernel.
// k_conv_lowc is bulk-synchronous: at 512x512 all 1024 workgroups are resident at once, so the chip gathers (HBM busy, matrix
// pipes idle: 14 us of down_lbl.0's 48), then computes (21 us of MFMA at best, HBM idle), then stores (8 us) - phase stamps
// in tools/probes/lowc_harness.hip.  Here a workgroup walks tiles blockIdx.x, + gridDim.x, ... of its sample and the halo tile
// of the NEXT tile is gathered by LDS-DMA (global_load_lds_dword: 4 bytes per lane from per-lane addresses into 64 consecutive
// LDS words; no staging registers) into the other LDS buffer while the matrix cores work on this one; the stores of a tile
// drain under the next tile's MFMAs.  What changes with it:
//   * LDS layout: channel planes [c][10][TW + 2] (x contiguous: the order of the NCHW source), a plane padded to whole DMA
//     instructions (340 -> 384 words), so that an instruction lies inside ONE plane: wave w gathers planes w, w + 4, ... and
//     the source tensor / channel / "a padding channel" of an instruction is wave-uniform (scalar); per lane only the
//     offset of its (row, x) in a plane, the same for every plane, computed once per tile.  Padding elements (outside the
//     image, channels beyond the real ones) read a page of zeros.  A fragment's 32 (16) pixels are consecutive words of a
//     plane row: conflict-free without an odd pitch; lane half lh (k parity) is the next plane.
//   * all filter fragments stay in registers for the whole kernel (no staging registers compete: 99 + 32 accumulators on the
//     22-channel layer); 2 workgroups per CU (2 x 33 KB of LDS each).
// Same reduction order as k_conv_lowc: bit-identical results.  grid (workgroups per sample, B).
// ---------------------------------------------------------------------------------------------
template <int CE, int NCOL, int ST, int TW = 32>
__global__ __launch_bounds__(256, 2) void k_conv_lowc_dma(const LowcParams p) {      // (2 waves per SIMD: 256 registers)
  constexpr bool N16 = NCOL == 16;
  static_assert(NCOL == 16 || NCOL == 32 || NCOL == 64, "16-, 32- or 64-column layers");
  static_assert(CE % (N16 ? 4 : 2) == 0, "channel count rounded up to the k-group of one MFMA");
  static_assert(TW == 32 || TW == 16, "8x32 or 8x16 pixel tiles");
  constexpr int TH = 8, IH = TH + 2, IW = TW + 2;
  constexpr int NPI = (IH * IW + 63) / 64, PL = NPI * 64;      // a plane = NPI whole DMA instructions (340 -> 384, 180 -> 192 words)
  constexpr int MF = TW / 16;
  constexpr int KG = N16 ? 4 : 2;
  constexpr int S = 9 * CE / KG;
  constexpr int NF = N16 ? 1 : NCOL / 32;
  constexpr int BUF = CE * PL;
  __shared__ __attribute__((aligned(16))) float sA[2 * BUF];
  __shared__ __attribute__((aligned(16))) double red[4][NCOL][2];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n = blockIdx.y;
  const int ntiles = p.tilesX * p.tilesY;
  const int ctot = p.c0 + p.c1 + p.c2;
  const size_t HW = (size_t)p.H * p.W;
  typedef __attribute__((address_space(3))) void lds_void;
  const uint32_t lds0 = (uint32_t)(size_t)(lds_void*)sA;
  // Wave w gathers planes w, w + 4, ...: the plane (source tensor, channel, in range or padding) is wave-uniform, and word
  // k * 64 + lane of a plane is the same (row, x) of the halo tile for every plane and tile
  int yx[NPI];
#pragma unroll
  for (int k = 0; k < NPI; ++k) {
    const int r = k * 64 + lane;
    yx[k] = r < IH * IW ? (r / IW) << 8 | (r % IW) : -1;      // (words beyond the 10 x (TW + 2) elements: never read)
  }
  auto gather = [&](int tile, int buf) {
    const int ty0 = (tile / p.tilesX) * TH - 1, tx0 = (tile % p.tilesX) * TW - 1;
    int rel[NPI];      // element offset inside a channel plane of the source, or -1: a padding element
#pragma unroll
    for (int k = 0; k < NPI; ++k) {
      const int gy = ty0 + (yx[k] >> 8), gx = tx0 + (yx[k] & 255);
      rel[k] = (yx[k] >= 0 && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) ? gy * p.W + gx : -1;
    }
#pragma unroll
    for (int j = 0; j < (CE + 3) / 4; ++j) {
      const int c = j * 4 + wave;
      if (c < CE) {
        const float* base = nullptr;      // wave-uniform
        if (c < p.c0) base = p.s0 + ((size_t)n * p.c0 + c) * HW;
        else if (c < p.c0 + p.c1) base = p.s1 + ((size_t)n * p.c1 + (c - p.c0)) * HW;
        else if (c < ctot) base = p.s2 + ((size_t)n * p.c2 + (c - p.c0 - p.c1)) * HW;
#pragma unroll
        for (int k = 0; k < NPI; ++k) {
          const float* src = (base && rel[k] >= 0) ? base + rel[k] : g_lowc_zeros;
          const uint32_t dst = lds0 + (uint32_t)(buf * BUF + c * PL + k * 64) * 4u;      // wave-uniform; the lane's word follows from its id
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" ::"s"(dst), "v"(src) : "memory");
        }
      }
    }
  };
  int tile = blockIdx.x;
  RIB_STAMP(0);
  if (tile < ntiles) gather(tile, 0);
  // this lane's filter fragments, for every tile of the walk
  float bw[S][NF];
  {
    const float* pw = N16 ? p.w + (lane >> 4) * 16 + (lane & 15) : p.w + (lane >> 5) * NCOL + (lane & 31);
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) bw[s][nf] = pw[s * KG * NCOL + nf * 32];
  }
  for (int buf = 0; tile < ntiles; tile += gridDim.x, buf ^= 1) {
    const int ty0 = (tile / p.tilesX) * TH, tx0 = (tile % p.tilesX) * TW;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's part of the tile has landed (and its stores of the previous tile are out)
    __syncthreads();                                        // ... everybody's; everybody is done with the other buffer and with `red`
    if (tile == (int)blockIdx.x) RIB_STAMP(1);
    if (tile + (int)gridDim.x < ntiles) gather(tile + gridDim.x, buf ^ 1);
    if (tile == (int)blockIdx.x) RIB_STAMP(2);
    const float* sT = sA + buf * BUF;
    double s1 = 0.0, s2 = 0.0;
    if constexpr (!N16) {
      const int li = lane & 31, lh = lane >> 5;
      f32x16 acc[MF][NF];
#pragma unroll
      for (int mf = 0; mf < MF; ++mf)
#pragma unroll
        for (int nf = 0; nf < NF; ++nf)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;
      // window origin of this lane's pixel: (row 2*wave [+ mf], x = li) or, 16 wide, (row 2*wave + li/16, x = li%16); k parity lh = the next plane
      const float* pa = sT + (TW == 32 ? (wave * 2) * IW + li : (wave * 2 + (li >> 4)) * IW + (li & 15)) + lh * PL;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int tap = (2 * s) / CE, c = (2 * s) % CE;
        const int off = c * PL + (tap / 3) * IW + (tap % 3);
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
          const float a = pa[off + mf * IW];
#pragma unroll
          for (int nf = 0; nf < NF; ++nf) acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[s][nf], acc[mf][nf], 0, 0, 0);
        }
      }
#if RIB_EXP & 1024
      if (acc[0][0][0] == 123.456f) g_lowc_stamps[1 << 20] = 1;      // (the stamp waits for the accumulators)
#endif
      if (tile == (int)blockIdx.x) RIB_STAMP(3);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const int col = nf * 32 + li;
        const float bv = p.bias[col];
        const bool cok = col < p.Cout;
        double c1 = 0.0, c2 = 0.0;
#pragma unroll
        for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;           // pixel of the fragment
            const int oy = ty0 + wave * 2 + (TW == 32 ? mf : (m >> 4));
            const int ox = tx0 + (TW == 32 ? m : (m & 15));
            float v = apply_act(acc[mf][nf][r] + bv, p.act);
            if (ST != ST_F32) v = round16<ST>(v);
            const bool ok = cok && oy < p.H && ox < p.W;
            if (ok) st_act<ST>(p.y, ((size_t)n * HW + (size_t)oy * p.W + ox) * p.yC + p.yoff + col, v);
            v = ok ? v : 0.f;
            c1 += (double)v; c2 += (double)v * (double)v;
          }
        }
        if (p.stat_part) {
          c1 += __shfl_xor(c1, 32); c2 += __shfl_xor(c2, 32);
          if (lh == 0) { red[wave][col][0] = c1; red[wave][col][1] = c2; }
        }
      }
    } else {
      const int l15 = lane & 15, lq = lane >> 4;
      constexpr int NFR = TW / 8;                    // 16-pixel fragments per wave: (row 2*wave + f/2, x half f%2) or (row 2*wave + f)
      f32x4 acc[NFR];
#pragma unroll
      for (int f = 0; f < NFR; ++f)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[f][r] = 0.f;
      const float* pa = sT + (wave * 2) * IW + l15 + lq * PL;
#pragma unroll
      for (int s = 0; s < S; ++s) {
        const int tap = (4 * s) / CE, c = (4 * s) % CE;
        const int off = c * PL + (tap / 3) * IW + (tap % 3);
#pragma unroll
        for (int f = 0; f < NFR; ++f)
          acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(pa[off + (TW == 32 ? ((f >> 1) * IW + (f & 1) * 16) : f * IW)], bw[s][0], acc[f], 0, 0, 0);
      }
      const int col = l15;
      const float bv = p.bias[col];
      const bool cok = col < p.Cout;
#pragma unroll
      for (int f = 0; f < NFR; ++f) {
        const int oy = ty0 + wave * 2 + (TW == 32 ? (f >> 1) : f);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ox = tx0 + (TW == 32 ? (f & 1) * 16 : 0) + lq * 4 + r;
          float v = apply_act(acc[f][r] + bv, p.act);
          if (ST != ST_F32) v = round16<ST>(v);
          const bool ok = cok && oy < p.H && ox < p.W;
          if (ok) st_act<ST>(p.y, ((size_t)n * HW + (size_t)oy * p.W + ox) * p.yC + p.yoff + col, v);
          v = ok ? v : 0.f;
          s1 += (double)v; s2 += (double)v * (double)v;
        }
      }
      if (p.stat_part) {
        s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
        s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
        if (lq == 0) { red[wave][col][0] = s1; red[wave][col][1] = s2; }
      }
    }
    if (p.stat_part) {
      __syncthreads();
      for (int c = tid; c < p.CoutPad; c += 256) {
        double a1 = 0.0, a2 = 0.0;
        if (c < NCOL) {
#pragma unroll
          for (int w = 0; w < 4; ++w) { a1 += red[w][c][0]; a2 += red[w][c][1]; }
        }
        double* dst = p.stat_part + (((size_t)n * ntiles + tile) * 2) * p.CoutPad;
        dst[c] = a1;
        dst[p.CoutPad + c] = a2;
      }
    }
    if (tile == (int)blockIdx.x) RIB_STAMP(4);
    if (tile + (int)gridDim.x >= ntiles) RIB_STAMP(5);
  }
}

}  // namespace rib
using namespace rib;

template <typename F> float time_us(F launch) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) launch();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    (void)hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms * 1000.f / 20);
  }
  return best;
}

template <int CE, int NCOL, int TW> void run(const char* name, int cin, int cout, bool stats) {
  const int H = 512, W = 512;
  float *x, *w, *b, *y; double* part;
  (void)hipMalloc(&x, (size_t)cin * H * W * 4); (void)hipMalloc(&w, (size_t)9 * CE * NCOL * 4 + 4096); (void)hipMalloc(&b, NCOL * 4);
  (void)hipMalloc(&y, (size_t)H * W * NCOL * 4);
  const int tilesX = (W + TW - 1) / TW, tilesY = (H + 7) / 8;
  (void)hipMalloc(&part, (size_t)tilesX * tilesY * 2 * NCOL * 8);
  std::vector<float> h((size_t)cin * H * W);
  for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
  (void)hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(w, h.data(), (size_t)9 * CE * NCOL * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(b, h.data(), NCOL * 4, hipMemcpyHostToDevice);
  LowcParams p;
  memset(&p, 0, sizeof p);
  p.s0 = x; p.c0 = cin; p.H = H; p.W = W; p.w = w; p.bias = b; p.y = y; p.yC = NCOL; p.yoff = 0; p.Cout = cout; p.act = 0;
  p.stat_part = stats ? part : nullptr; p.CoutPad = NCOL; p.tilesX = tilesX; p.tilesY = tilesY;
  dim3 g(tilesX * tilesY, 1);
#if RIB_EXP & 1024
  long long* stamps; (void)hipMalloc(&stamps, ((1 << 20) + 8) * 8); (void)hipMemset(stamps, 0, ((1 << 20) + 8) * 8);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lowc_stamps), &stamps, sizeof stamps);
#endif
  const float t = time_us([&] { k_conv_lowc<CE, NCOL, ST_F32, TW><<<g, 256>>>(p); });
#if RIB_EXP & 1024
  {
    std::vector<long long> hs((size_t)g.x * 8);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost);
    long long t0min = hs[0], t5max = 0; double ph[5] = {0, 0, 0, 0, 0};
    for (unsigned b = 0; b < g.x; ++b) {
      t0min = std::min(t0min, hs[b * 8]); t5max = std::max(t5max, hs[b * 8 + 5]);
      for (int i = 0; i < 5; ++i) ph[i] += (double)(hs[b * 8 + i + 1] - hs[b * 8 + i]) / g.x;
    }
    long long first_end = hs[5], last_start = 0;
    for (unsigned b = 0; b < g.x; ++b) { first_end = std::min(first_end, hs[b * 8 + 5]); last_start = std::max(last_start, hs[b * 8]); }
    printf("   stamps (ticks; kernel span %lld = first start .. last end; last start at +%lld, first end at +%lld): gather+LDS %.0f | filters+barrier %.0f | MFMA loop %.0f | epilogue %.0f | stats tail %.0f\n",
           t5max - t0min, last_start - t0min, first_end - t0min, ph[0], ph[1], ph[2], ph[3], ph[4]);
  }
#endif
  printf("RIB_EXP %3d  %-28s CE %2d NCOL %2d TW %2d  grid %5d  %6.1f us\n", RIB_EXP, name, CE, NCOL, TW, g.x, t);
  {   // the persistent LDS-DMA version against it: same bits, time per workgroup count
    std::vector<float> y0((size_t)H * W * NCOL), y1(y0.size());
    std::vector<double> p0((size_t)tilesX * tilesY * 2 * NCOL), p1(p0.size());
    (void)hipMemset(y, 0, y0.size() * 4);
    k_conv_lowc<CE, NCOL, ST_F32, TW><<<g, 256>>>(p);
    (void)hipMemcpy(y0.data(), y, y0.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(p0.data(), part, p0.size() * 8, hipMemcpyDeviceToHost);
    float* z; (void)hipMalloc(&z, 256); (void)hipMemset(z, 0, 256);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_lowc_zeros), &z, sizeof z);
    for (int wgs : {256, 384, 512, 768, 1024}) {
      if (wgs > (int)g.x) continue;
      (void)hipMemset(y, 0, y0.size() * 4); (void)hipMemset(part, 0, p0.size() * 8);
      dim3 g2(wgs, 1);
      k_conv_lowc_dma<CE, NCOL, ST_F32, TW><<<g2, 256>>>(p);
      (void)hipMemcpy(y1.data(), y, y1.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(p1.data(), part, p1.size() * 8, hipMemcpyDeviceToHost);
      size_t bad = 0, badp = 0;
      for (size_t i = 0; i < y0.size(); ++i) bad += memcmp(&y0[i], &y1[i], 4) != 0;
      if (stats) for (size_t i = 0; i < p0.size(); ++i) badp += memcmp(&p0[i], &p1[i], 8) != 0;
      const float t2 = time_us([&] { k_conv_lowc_dma<CE, NCOL, ST_F32, TW><<<g2, 256>>>(p); });
#if RIB_EXP & 1024
      if (NCOL != 16) {
        std::vector<long long> hs((size_t)wgs * 8);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(hs.data(), stamps, hs.size() * 8, hipMemcpyDeviceToHost);
        double ph[5] = {0, 0, 0, 0, 0};
        for (int b = 0; b < wgs; ++b) for (int i = 0; i < 5; ++i) ph[i] += (double)(hs[b * 8 + i + 1] - hs[b * 8 + i]) / wgs;
        printf("                 first tile (ticks): first gather lands %.0f | issue next gather %.0f | MFMA loop %.0f | epilogue %.0f | rest of the walk %.0f\n", ph[0], ph[1], ph[2], ph[3], ph[4]);
      }
#endif
      printf("             persistent DMA gather, %4d workgroups: %6.1f us   mismatching outputs %zu, partials %zu\n", wgs, t2, bad, badp);
    }
    (void)hipFree(z);
  }
  (void)hipFree(x); (void)hipFree(w); (void)hipFree(b); (void)hipFree(y); (void)hipFree(part);
}

int main() {
  run<6, 64, 16>("ref_embedding.conv_first", 6, 64, false);
  run<24, 16, 32>("down_first", 22, 16, true);
  run<22, 32, 32>("down_lbl.0", 22, 32, true);
  run<10, 32, 32>("down_img.0", 9, 32, true);
  return 0;
}
