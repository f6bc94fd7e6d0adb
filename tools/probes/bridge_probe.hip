// Probe (round 5, VERDICT r04 item 1, premise tests ii / iii): can the launches between two Winograd-domain GEMMs of
// consecutive convolutions - output transform of convolution k, InstanceNorm statistics of its output, prologue and input
// transform of convolution k+1 - become ONE launch without a device-wide barrier (grid_barrier_probe.hip: 4.0-7.7 us, more
// than the boundary it would replace)?  InstanceNorm needs the statistics of the WHOLE map before the first normalised value;
// a workgroup that owns a few channels of ALL pixels has that dependency inside itself ("bridge").  Built and measured here:
//   * the Winograd-domain operands in a channel-group layout X[z][C/4][tiles][4] between the two GEMMs, so that such a
//     workgroup reads and writes contiguous memory; k_gemm_cg = the production k_gemm_dma with (a) an A operand in that
//     layout (the 8 rows of an LDS-DMA fill make one 128-byte run per k-group) and (b) a C store in it: the MFMA operands
//     trade places, a lane then holds one tile and 4 consecutive channels per accumulator quad = one 16-byte store;
//   * k_bridge4: workgroup = one group of 4 channels (float4 per pixel), 256 threads, thread = tile;
//     k_bridge1: workgroup = ONE channel, thread = tile (4x the workgroups; 4-byte accesses at a 16-byte stride).
//   Phases: A  M -> A^T M A + bias -> the map into an LDS image (pixel columns de-interleaved: conflict-free) + fp64 sums;
//           statistics (shuffles + one LDS step, fixed order); B1 normalise + LeakyReLU in place; B2 patches -> B^T d B -> V.
// Result (MI355X, profiles/r05_bridge_probe.txt; kernel durations from rocprofv3 --kernel-trace):
//   F(4x4) 64x64 C=256 B=1 (the mask network's res_flow blocks): out 5.9 + in 10.2 us -> bridge4 16.2 us, bridge1 13.8 us;
//   F(2x2) 32x32 C=512 B=1 (res_0 / res_1):                      out 4.9 + in 5.7 us  -> bridge4  9.1 us, bridge1 10.0 us;
//   the sequence GEMM - bridge - GEMM against GEMM - out - in - GEMM at batch 1: 51.2 vs 52.3 us, 52.1 vs 54.5 us (-2 .. -4 %);
//   B=4: out 14.1 + in 22.6 -> bridge4 22.6 us (F(4x4)), 8.8 + 13.8 -> 15.4 us (F(2x2)); sequences 118.7 vs 134.7 us (-12 %),
//   166.6 vs 172.5 us (-3 %); 40x60 (320x480 frames) B=8: 163.7 vs 184.1 us (-11 %).
//   The channel-group C store alone makes the GEMM 1.0-3.2 us (2-6 %) shorter.
// Why it does not pay at batch 1: the transforms are ~2800 vector instructions per tile (SQ_INSTS_VALU 706 k per launch), and
// a workgroup per channel GROUP runs them on C/4 = 64 CUs at one wavefront per SIMD (SQ_ACTIVE_INST_VALU = 11 k cycles per
// wavefront of a 43 k-cycle launch, the rest waits on memory with nothing else to issue), where the two production kernels
// spread 2.5x as many instructions over all 256 CUs and still finish sooner.  One channel per workgroup fills the chip but moves
// 4 bytes of every 16 (phase A 6 us, B2 9 us).  The statistics dependency pins a channel's pixels to one workgroup, the
// vector work of a layer is too large for C/4 workgroups, and nothing cheaper than a launch boundary joins more of them.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/probes/bridge_probe.hip -o tools/probes/bin/bridge_probe
//   bridge_probe [shape index] [1 = bridge1 instead of bridge4]      (WB_XCD=1: bridge1 takes channels in an XCD-aware order)
#include <hip/hip_runtime.h>
__device__ int g_wb_stop = 0;          // the bridge returns after phase A (1), after the statistics (2), after B1 (3)
#define RIB_WB_STOP g_wb_stop
#include "../../render-in-between_amd/csrc/kernels.hip.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
using namespace rib;

namespace rib {
// ---- k_gemm_dma (kernels.hip.h) with the channel-group layout options ----
struct GemmCgParams {
  const float* A; const float* B; float* C;
  int M, N, K, lda, ldc;
  size_t sA, sB, sC;      // element strides of the batch index z (B: of z % modB)
  int modB;               // 0: one B for every z
  // Channel-group layout of the Winograd-domain operands between two bridged convolutions (k_wino_bridge, round 5):
  // X[z][K / 4][M][4] instead of X[z][M][K] - a 4-channel group of ALL rows (tiles) is contiguous, which is what a
  // transform workgroup that owns one channel group reads and writes.  A in that layout: a_cg != 0 (a lane's 16-byte DMA
  // source is then row * 4 + kgroup * M * 4: the 8 rows of a fill make one 128-byte run per k-group instead of one per row);
  // C in that layout: the CGC instantiation (below).
  int a_cg;
};

// ST = ST_BF16 / ST_F16: A and B hold 16-bit elements (lda, K, sA, sB in elements), a 128-byte row chunk is 64 of them, a
// lane's 16-byte slot feeds ONE v_mfma_f32_32x32x16 (k = 8 per lane half) where it feeds four fp32 MFMAs; C stays fp32 (the
// level slab k_spade_modulate reads).  Same tile, same swizzle, same two stages.
// CGC: C is written in the channel-group layout C[z][N / 4][M][4].  The MFMA operands trade places (D^T = B A^T), so a lane holds
// ONE row (tile) and 4 consecutive columns (channels) per accumulator quad: one 16-byte store per quad, 32 lanes = 512 contiguous bytes.
template <int WM, int WN, int NF, int ST = ST_F32, bool CGC = false>
__global__ __launch_bounds__(256) void k_gemm_cg(const GemmCgParams p) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int BM = 32 * WM, BN = 32 * NF * WN, BK = 32;      // BK: 4-byte words of a row chunk (128 bytes)
  constexpr int EPW = ST == ST_F32 ? 1 : 2;                    // elements per 4-byte word
  constexpr int STAGE = (BM + BN) * BK;              // floats
  constexpr int NFILL = (BM + BN) / 32;              // DMA instructions per wave and chunk (8 rows each)
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int z = blockIdx.z;
  // (4-byte word pointers; element strides are even in the 16-bit modes: padded channel counts)
  const float* A = p.A + (size_t)z * p.sA / EPW;
  const float* B = p.B + (size_t)(p.modB ? z % p.modB : 0) * p.sB / EPW;
  f32x16 acc[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nf][r] = 0.f;
  const int nch = p.K / (BK * EPW);
  typedef __attribute__((address_space(3))) void lds_void;
  const uint32_t lds0 = (uint32_t)(size_t)(lds_void*)smem;
  // this lane's source rows: fill j of a wave covers tile rows [wave * 8 + 32 j, + 8) of the stacked (A rows | B rows) tile;
  // lane -> row lane / 8, physical slot lane % 8 = logical slot ^ ((row >> 1) & 7)
  const float* src[NFILL];
#pragma unroll
  for (int j = 0; j < NFILL; ++j) {
    const int trow = wave * 8 + 32 * j + (lane >> 3);                 // row of the stacked tile
    const bool isA = 32 * j < BM;                                     // (BM is a multiple of 32: a fill never straddles A | B)
    const int row = isA ? trow : trow - BM;
    const int ls = (lane & 7) ^ ((row >> 1) & 7);
    if (isA && p.a_cg) src[j] = A + ((size_t)ls * p.M + min(m0 + row, p.M - 1)) * 4;      // (fp32 only) k-group ls of this chunk, row `row`
    else src[j] = isA ? A + (size_t)min(m0 + row, p.M - 1) * (p.lda / EPW) + ls * 4 : B + (size_t)min(n0 + row, p.N - 1) * (p.K / EPW) + ls * 4;
  }
  const size_t a_kstep = p.a_cg ? (size_t)p.M : 1;      // floats the A source advances per k element
  auto fill = [&](int st, int kc) {
#pragma unroll
    for (int j = 0; j < NFILL; ++j) {
      const uint32_t dst = lds0 + (uint32_t)(st * STAGE + (wave * 8 + 32 * j) * BK) * 4u;      // wave-uniform byte address
      const float* s_ = src[j] + (32 * j < BM ? (size_t)kc * a_kstep : (size_t)kc);
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(s_) : "memory");
    }
  };
  fill(0, 0);
  for (int c = 0; c < nch; ++c) {
    const int st = c & 1;
    // chunk c has landed (this wave's part; the barrier collects the others') and everybody is done with chunk c - 1
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (c + 1 < nch) fill(st ^ 1, (c + 1) * BK);
    const float* sA = smem + st * STAGE;
    const float* sB = sA + BM * BK;
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      const int slot = kb * 2 + lh;
      const int ra = wm * 32 + li;
      const float4 a = *reinterpret_cast<const float4*>(sA + ra * BK + (slot ^ ((ra >> 1) & 7)) * 4);
      float4 b[NF];
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const int rb = (wn * NF + nf) * 32 + li;
        b[nf] = *reinterpret_cast<const float4*>(sB + rb * BK + (slot ^ ((rb >> 1) & 7)) * 4);
      }
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        if constexpr (ST == ST_F32 && CGC) {
          acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[nf].x, a.x, acc[nf], 0, 0, 0);
          acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[nf].y, a.y, acc[nf], 0, 0, 0);
          acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[nf].z, a.z, acc[nf], 0, 0, 0);
          acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(b[nf].w, a.w, acc[nf], 0, 0, 0);
        } else if constexpr (ST == ST_F32) {
          acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[nf].x, acc[nf], 0, 0, 0);
          acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[nf].y, acc[nf], 0, 0, 0);
          acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[nf].z, acc[nf], 0, 0, 0);
          acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[nf].w, acc[nf], 0, 0, 0);
        } else if constexpr (ST == ST_F16) {
          acc[nf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(&a), *reinterpret_cast<const f16x8*>(&b[nf]), acc[nf], 0, 0, 0);
        } else {
          acc[nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b[nf]), acc[nf], 0, 0, 0);
        }
      }
    }
  }
  float* C = p.C + (size_t)z * p.sC;
  if constexpr (CGC) {
    static_assert(ST == ST_F32, "channel-group output: fp32 (the Winograd path)");
    // operands traded: accumulator element r of lane l is row (tile) l & 31, column (channel) (r & 3) + 8 * (r >> 2) + 4 * (l >> 5)
    const int row = m0 + wm * 32 + li;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = n0 + (wn * NF + nf) * 32 + 8 * q + 4 * lh;
        if (row < p.M && col < p.N)
          *reinterpret_cast<float4*>(C + ((size_t)(col >> 2) * p.M + row) * 4) = make_float4(acc[nf][4 * q], acc[nf][4 * q + 1], acc[nf][4 * q + 2], acc[nf][4 * q + 3]);
      }
    return;
  }
  // accumulator element r of lane l: row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), column l & 31
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int col = n0 + (wn * NF + nf) * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (row < p.M && col < p.N) C[(size_t)row * p.ldc + col] = acc[nf][r];
    }
  }
}


// ---- variant "1": a workgroup owns ONE channel ----
enum { WB_PLAIN = 0, WB_SPADE = 1, WB_JOIN = 2 };
struct WinoBridgeParams {
  const float* m; int tilesY, tilesX, C;          // M [B*NP][C/4][tiles][4] (channel-group layout, GemmDmaParams): the producer's CoutPad = the consumer's Cin
  const float* bias; const float* res; int resC;  // residual added before the statistics (the res blocks' identity shortcut)
  float* y; int yC;                               // optional: the producer's raw output, [B][H][W][yC]
  int H, W;
  int norm; float inv_count; const float* gamma; const float* beta;   // InstanceNorm of the producer's output (+ its affine) or none
  int lrelu;
  const float* slab; int slab_ld, col0; const float* sbias;           // WB_SPADE: the level's gamma/beta slab, this group's bias
  const float* xres; float* o; int oC;                                // WB_JOIN: o = IN(y) + xres, stored
  float* v;                                       // V [B*NP][C/4][tiles][4] of the consumer (same layout), or nullptr (terminal)
  float* act_out; int aC;                         // terminal: the prologue's result, [B][H][W][aC]
  int xcd_groups;                                 // C / 8 when C % 8 == 0: workgroup b takes channel (b % 8) * (C / 8) + b / 8, else 0 (channel b)
};

// the 1-D transforms on one channel
template <int WM> struct WinoT1;
template <> struct WinoT1<2> {
  // A^T = [1 1 1 0; 0 1 -1 -1], B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
  static __device__ __forceinline__ void out1d(const float (&u)[4], float (&y)[2]) { y[0] = u[0] + u[1] + u[2]; y[1] = u[1] - u[2] - u[3]; }
  static __device__ __forceinline__ void in1d(const float (&t)[4], float (&o)[4]) { o[0] = t[0] - t[2]; o[1] = t[1] + t[2]; o[2] = t[2] - t[1]; o[3] = t[1] - t[3]; }
};
template <> struct WinoT1<4> {     // the dyadic points of kWino4AT / kWino4BT
  static __device__ __forceinline__ void out1d(const float (&u)[6], float (&y)[4]) {
    const float s12 = u[1] + u[2], d12 = u[1] - u[2], s34 = u[3] + u[4], d34 = u[3] - u[4];
    y[0] = u[0] + s12 + s34;
    y[1] = fmaf(3.f / 4, d12, (3.f / 2) * d34);
    y[2] = fmaf(9.f / 16, s12, (9.f / 4) * s34);
    y[3] = fmaf(27.f / 64, d12, fmaf(27.f / 8, d34, u[5]));
  }
  static __device__ __forceinline__ void in1d(const float (&t)[6], float (&o)[6]) {
    const float e24 = fmaf(-45.f / 16, t[2], t[4]), o13 = fmaf(-45.f / 16, t[3], t[5]);
    const float ev1 = fmaf(-9.f / 4, t[2], t[4]), od1 = fmaf(-27.f / 16, t[1], (3.f / 4) * t[3]);
    const float ev2 = fmaf(-9.f / 16, t[2], t[4]), od2 = fmaf(-27.f / 32, t[1], (3.f / 2) * t[3]);
    o[0] = fmaf(81.f / 64, t[0], e24); o[1] = ev1 + od1; o[2] = ev1 - od1; o[3] = ev2 + od2; o[4] = ev2 - od2; o[5] = fmaf(81.f / 64, t[1], o13);
  }
};

template <int WM, int MODE>
__global__ __launch_bounds__(256) void k_bridge1(const WinoBridgeParams p) {
  constexpr int T = WM + 2, NP = T * T, LW = WM == 4 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) float wb_img[];      // [H][WM * tilesX], pixel columns de-interleaved
  __shared__ double wb_red[4][2];
  __shared__ float wb_ss[2];
  const int b = blockIdx.x, n = blockIdx.y;
  const int ch = p.xcd_groups ? (b & 7) * p.xcd_groups + (b >> 3) : b;      // this workgroup's channel
  const int c4 = ch >> 2, e = ch & 3;
  const int ntiles = p.tilesY * p.tilesX;
  const int pitch = WM * p.tilesX;
  const size_t plane = (size_t)ntiles * p.C;
  const float bv = p.bias[ch];
  // ---- phase A: output transform of the producer, its statistics, the raw pixels into the LDS image ----
  // (every load and every LDS read of a phase is issued unconditionally from clamped coordinates and masked afterwards: a
  // per-element branch makes the compiler wait for each access in turn)
  double s1 = 0.0, s2 = 0.0;
  for (int tile = threadIdx.x; tile < ntiles; tile += 256) {
    const int ty = tile / p.tilesX, tx = tile - ty * p.tilesX;
    const float* mb = p.m + (size_t)n * NP * plane + ((size_t)c4 * ntiles + tile) * 4 + e;
    float mm[T][T];
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
      for (int q = 0; q < T; ++q) mm[a][q] = mb[(size_t)(a * T + q) * plane];
    float rr[WM][WM];
    if (p.res) {
#pragma unroll
      for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WM; ++j)
          rr[i][j] = p.res[(((size_t)n * p.H + min(WM * ty + i, p.H - 1)) * p.W + min(WM * tx + j, p.W - 1)) * p.resC + ch];
    }
    float uu[WM][T];      // columns first: uu[i][q] = sum_a A^T[i][a] M[a][q], then rows
#pragma unroll
    for (int q = 0; q < T; ++q) {
      float col[T], o[WM];
#pragma unroll
      for (int a = 0; a < T; ++a) col[a] = mm[a][q];
      WinoT1<WM>::out1d(col, o);
#pragma unroll
      for (int i = 0; i < WM; ++i) uu[i][q] = o[i];
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      float yv[WM];
      WinoT1<WM>::out1d(uu[i], yv);
      const int oy = WM * ty + i;
#pragma unroll
      for (int j = 0; j < WM; ++j) {
        const int ox = WM * tx + j;
        float t = yv[j] + bv;
        if (p.res) t += rr[i][j];
        const bool ok = oy < p.H && ox < p.W;
        if (ok) wb_img[oy * pitch + j * p.tilesX + tx] = t;
        if (ok && p.y) p.y[(((size_t)n * p.H + oy) * p.W + ox) * p.yC + ch] = t;
        if (!ok) t = 0.f;
        s1 += (double)t; s2 += (double)t * (double)t;
      }
    }
  }
  if (RIB_WB_STOP == 1) { if (s1 == 12345.0) p.v[0] = 1.f; return; }
  // ---- the consumer prologue's own operands: issued before the reduction so that their latency hides behind it ----
  // (one tile per thread is the common case - 256 tiles at 64x64 F(4x4) / 32x32 F(2x2); further tiles load inside B1)
  float pg[WM][WM], pb[WM][WM];
  const int tile0 = threadIdx.x;
  const int ty0 = tile0 / p.tilesX, tx0 = tile0 - ty0 * p.tilesX;
  const int colg = (ch / 32) * 64 + (ch % 32);
  auto load_pro = [&](int ty, int tx, float (&g)[WM][WM], float (&bb)[WM][WM]) {
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WM; ++j) {
        const size_t pix = ((size_t)n * p.H + min(WM * ty + i, p.H - 1)) * p.W + min(WM * tx + j, p.W - 1);
        if constexpr (MODE == WB_SPADE) {
          const float* sl = p.slab + pix * p.slab_ld + p.col0 + colg;
          g[i][j] = sl[0];
          bb[i][j] = sl[32];
        } else if constexpr (MODE == WB_JOIN) g[i][j] = p.xres[pix * p.oC + ch];
      }
  };
  if (MODE != WB_PLAIN && tile0 < ntiles) load_pro(ty0, tx0, pg, pb);
  // ---- statistics: fixed-order fp64 reduction over the workgroup ----
  float sc = 1.f, sh = 0.f;
  if (p.norm) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { s1 += __shfl_xor(s1, off); s2 += __shfl_xor(s2, off); }
    if ((threadIdx.x & 63) == 0) { wb_red[threadIdx.x >> 6][0] = s1; wb_red[threadIdx.x >> 6][1] = s2; }
    __syncthreads();
    if (threadIdx.x == 0) {
      const double t1 = ((wb_red[0][0] + wb_red[1][0]) + wb_red[2][0]) + wb_red[3][0];
      const double t2 = ((wb_red[0][1] + wb_red[1][1]) + wb_red[2][1]) + wb_red[3][1];
      float a, c;
      scale_shift_of(t1, t2, p.inv_count, p.gamma ? p.gamma[ch] : 1.f, p.beta ? p.beta[ch] : 0.f, a, c);
      wb_ss[0] = a; wb_ss[1] = c;
    }
    __syncthreads();
    sc = wb_ss[0]; sh = wb_ss[1];
  }
  if (RIB_WB_STOP == 2) { if (sc == 12345.f) p.v[0] = 1.f; return; }
  float bg = 0.f, bbias = 0.f;
  if constexpr (MODE == WB_SPADE) { bg = p.sbias[colg]; bbias = p.sbias[colg + 32]; }
  // ---- phase B1: the consumer's prologue on this thread's own pixels, in place ----
  for (int tile = threadIdx.x; tile < ntiles; tile += 256) {
    const int ty = tile / p.tilesX, tx = tile - ty * p.tilesX;
    if (MODE != WB_PLAIN && tile != tile0) load_pro(ty, tx, pg, pb);
    float r[WM][WM];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WM; ++j) r[i][j] = wb_img[min(WM * ty + i, p.H - 1) * pitch + j * p.tilesX + tx];     // (columns beyond W: unwritten slots of the pitch, unused)
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WM; ++j) {
        const int oy = WM * ty + i, ox = WM * tx + j;
        const bool ok = oy < p.H && ox < p.W;
        float w;
        if constexpr (MODE == WB_SPADE) w = spade_mod1(r[i][j], sc, sh, pg[i][j], bg, pb[i][j], bbias);
        else {
          w = r[i][j] * sc + sh;
          if constexpr (MODE == WB_JOIN) { w += pg[i][j]; if (ok) p.o[(((size_t)n * p.H + oy) * p.W + ox) * p.oC + ch] = w; }
        }
        if (p.lrelu) w = lrelu(w);
        if (p.v) { if (ok) wb_img[oy * pitch + j * p.tilesX + tx] = w; }
        else if (ok) p.act_out[(((size_t)n * p.H + oy) * p.W + ox) * p.aC + ch] = w;
      }
  }
  if (!p.v) return;
  __syncthreads();
  if (RIB_WB_STOP == 3) return;
  // ---- phase B2: input transform of the consumer out of the LDS image ----
  for (int tile = threadIdx.x; tile < ntiles; tile += 256) {
    const int ty = tile / p.tilesX, tx = tile - ty * p.tilesX;
    float d[T][T];
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
      for (int q = 0; q < T; ++q) {
        const int cy = min(max(WM * ty - 1 + a, 0), p.H - 1), cx = min(max(WM * tx - 1 + q, 0), p.W - 1);
        d[a][q] = wb_img[cy * pitch + (cx & (WM - 1)) * p.tilesX + (cx >> LW)];
      }
    float t[T][T];      // zero padding comes after the prologue: mask now; columns t[r][q] = sum_a B^T[r][a] d[a][q], then rows
#pragma unroll
    for (int q = 0; q < T; ++q) {
      float col[T], o[T];
#pragma unroll
      for (int a = 0; a < T; ++a) {
        const int iy = WM * ty - 1 + a, ix = WM * tx - 1 + q;
        col[a] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? d[a][q] : 0.f;
      }
      WinoT1<WM>::in1d(col, o);
#pragma unroll
      for (int r = 0; r < T; ++r) t[r][q] = o[r];
    }
    float* vb = p.v + (size_t)n * NP * plane + ((size_t)c4 * ntiles + tile) * 4 + e;
#pragma unroll
    for (int r = 0; r < T; ++r) {
      float o[T];
      WinoT1<WM>::in1d(t[r], o);
#pragma unroll
      for (int q = 0; q < T; ++q) vb[(size_t)(r * T + q) * plane] = o[q];
    }
  }
}

// ---- variant "4": a workgroup owns a GROUP of 4 channels (one float4 per pixel), 256 threads, thread = tile ----
#define P_F4_SUB(a, b) make_float4((a).x - (b).x, (a).y - (b).y, (a).z - (b).z, (a).w - (b).w)
#define P_F4_ADD(a, b) make_float4((a).x + (b).x, (a).y + (b).y, (a).z + (b).z, (a).w + (b).w)
template <int WM> struct WinoT4;
template <> struct WinoT4<2> {
  static __device__ __forceinline__ void out1d(const float4 (&u)[4], float4 (&y)[2]) { y[0] = P_F4_ADD(P_F4_ADD(u[0], u[1]), u[2]); y[1] = P_F4_SUB(P_F4_SUB(u[1], u[2]), u[3]); }
  static __device__ __forceinline__ void in1d(const float4 (&t)[4], float4 (&o)[4]) { o[0] = P_F4_SUB(t[0], t[2]); o[1] = P_F4_ADD(t[1], t[2]); o[2] = P_F4_SUB(t[2], t[1]); o[3] = P_F4_SUB(t[1], t[3]); }
};
template <> struct WinoT4<4> {
  static __device__ __forceinline__ void out1d(const float4 (&u)[6], float4 (&y)[4]) {
    const float4 s12 = P_F4_ADD(u[1], u[2]), d12 = P_F4_SUB(u[1], u[2]), s34 = P_F4_ADD(u[3], u[4]), d34 = P_F4_SUB(u[3], u[4]);
    y[0] = P_F4_ADD(P_F4_ADD(u[0], s12), s34);
    y[1] = f4_fma(3.f / 4, d12, f4_scale(3.f / 2, d34));
    y[2] = f4_fma(9.f / 16, s12, f4_scale(9.f / 4, s34));
    y[3] = f4_fma(27.f / 64, d12, f4_fma(27.f / 8, d34, u[5]));
  }
  static __device__ __forceinline__ void in1d(const float4 (&t)[6], float4 (&o)[6]) {
    const float4 e24 = f4_fma(-45.f / 16, t[2], t[4]), o13 = f4_fma(-45.f / 16, t[3], t[5]);
    const float4 ev1 = f4_fma(-9.f / 4, t[2], t[4]), od1 = f4_fma(-27.f / 16, t[1], f4_scale(3.f / 4, t[3]));
    const float4 ev2 = f4_fma(-9.f / 16, t[2], t[4]), od2 = f4_fma(-27.f / 32, t[1], f4_scale(3.f / 2, t[3]));
    o[0] = f4_fma(81.f / 64, t[0], e24); o[1] = P_F4_ADD(ev1, od1); o[2] = P_F4_SUB(ev1, od1);
    o[3] = P_F4_ADD(ev2, od2); o[4] = P_F4_SUB(ev2, od2); o[5] = f4_fma(81.f / 64, t[1], o13);
  }
};

template <int WM>      // prologue: InstanceNorm affine + LeakyReLU (the mask network's conv_block_0 -> conv_block_1)
__global__ __launch_bounds__(256) void k_bridge4(const WinoBridgeParams p) {
  constexpr int T = WM + 2, NP = T * T, LW = WM == 4 ? 2 : 1;
  extern __shared__ __attribute__((aligned(16))) float4 img4[];      // [H][WM * tilesX] float4, columns de-interleaved
  __shared__ double red4[4][8];
  __shared__ float ss4[8];
  const int c4 = blockIdx.x, n = blockIdx.y;
  const int ntiles = p.tilesY * p.tilesX, pitch = WM * p.tilesX;
  const size_t plane = (size_t)ntiles * p.C;
  const float4 bv = *reinterpret_cast<const float4*>(p.bias + c4 * 4);
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  for (int tile = threadIdx.x; tile < ntiles; tile += 256) {
    const int ty = tile / p.tilesX, tx = tile - ty * p.tilesX;
    const float* mb = p.m + (size_t)n * NP * plane + ((size_t)c4 * ntiles + tile) * 4;
    float4 mm[T][T];
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
      for (int q = 0; q < T; ++q) mm[a][q] = *reinterpret_cast<const float4*>(mb + (size_t)(a * T + q) * plane);
    float4 uu[WM][T];
#pragma unroll
    for (int q = 0; q < T; ++q) {
      float4 col[T], o[WM];
#pragma unroll
      for (int a = 0; a < T; ++a) col[a] = mm[a][q];
      WinoT4<WM>::out1d(col, o);
#pragma unroll
      for (int i = 0; i < WM; ++i) uu[i][q] = o[i];
    }
#pragma unroll
    for (int i = 0; i < WM; ++i) {
      float4 yv[WM];
      WinoT4<WM>::out1d(uu[i], yv);
      const int oy = WM * ty + i;
#pragma unroll
      for (int j = 0; j < WM; ++j) {
        const int ox = WM * tx + j;
        float4 t = P_F4_ADD(yv[j], bv);
        const bool ok = oy < p.H && ox < p.W;
        if (ok) img4[oy * pitch + j * p.tilesX + tx] = t;
        if (!ok) t = make_float4(0.f, 0.f, 0.f, 0.f);
        s1[0] += (double)t.x; s2[0] += (double)t.x * (double)t.x; s1[1] += (double)t.y; s2[1] += (double)t.y * (double)t.y;
        s1[2] += (double)t.z; s2[2] += (double)t.z * (double)t.z; s1[3] += (double)t.w; s2[3] += (double)t.w * (double)t.w;
      }
    }
  }
  if (RIB_WB_STOP == 1) { if (s1[0] == 12345.0) p.v[0] = 1.f; return; }
#pragma unroll
  for (int e = 0; e < 4; ++e)
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { s1[e] += __shfl_xor(s1[e], off); s2[e] += __shfl_xor(s2[e], off); }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { red4[threadIdx.x >> 6][e] = s1[e]; red4[threadIdx.x >> 6][4 + e] = s2[e]; }
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int e = threadIdx.x;
    const double t1 = ((red4[0][e] + red4[1][e]) + red4[2][e]) + red4[3][e];
    const double t2 = ((red4[0][4 + e] + red4[1][4 + e]) + red4[2][4 + e]) + red4[3][4 + e];
    float a, c;
    scale_shift_of(t1, t2, p.inv_count, p.gamma[c4 * 4 + e], p.beta[c4 * 4 + e], a, c);
    ss4[e] = a; ss4[4 + e] = c;
  }
  __syncthreads();
  const float4 sc = make_float4(ss4[0], ss4[1], ss4[2], ss4[3]), sh = make_float4(ss4[4], ss4[5], ss4[6], ss4[7]);
  if (RIB_WB_STOP == 2) { if (sc.x == 12345.f) p.v[0] = 1.f; return; }
  for (int tile = threadIdx.x; tile < ntiles; tile += 256) {
    const int ty = tile / p.tilesX, tx = tile - ty * p.tilesX;
    float4 r[WM][WM];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WM; ++j) r[i][j] = img4[min(WM * ty + i, p.H - 1) * pitch + j * p.tilesX + tx];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
      for (int j = 0; j < WM; ++j) {
        const float4 w = lrelu4(make_float4(r[i][j].x * sc.x + sh.x, r[i][j].y * sc.y + sh.y, r[i][j].z * sc.z + sh.z, r[i][j].w * sc.w + sh.w));
        if (WM * ty + i < p.H && WM * tx + j < p.W) img4[(WM * ty + i) * pitch + j * p.tilesX + tx] = w;
      }
  }
  __syncthreads();
  if (RIB_WB_STOP == 3) return;
  for (int tile = threadIdx.x; tile < ntiles; tile += 256) {
    const int ty = tile / p.tilesX, tx = tile - ty * p.tilesX;
    float4 d[T][T];
#pragma unroll
    for (int a = 0; a < T; ++a)
#pragma unroll
      for (int q = 0; q < T; ++q) {
        const int cy = min(max(WM * ty - 1 + a, 0), p.H - 1), cx = min(max(WM * tx - 1 + q, 0), p.W - 1);
        d[a][q] = img4[cy * pitch + (cx & (WM - 1)) * p.tilesX + (cx >> LW)];
      }
    float4 t[T][T];
#pragma unroll
    for (int q = 0; q < T; ++q) {
      float4 col[T], o[T];
#pragma unroll
      for (int a = 0; a < T; ++a) {
        const int iy = WM * ty - 1 + a, ix = WM * tx - 1 + q;
        col[a] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? d[a][q] : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      WinoT4<WM>::in1d(col, o);
#pragma unroll
      for (int r = 0; r < T; ++r) t[r][q] = o[r];
    }
    float* vb = p.v + (size_t)n * NP * plane + ((size_t)c4 * ntiles + tile) * 4;
#pragma unroll
    for (int r = 0; r < T; ++r) {
      float4 o[T];
      WinoT4<WM>::in1d(t[r], o);
#pragma unroll
      for (int q = 0; q < T; ++q) *reinterpret_cast<float4*>(vb + (size_t)(r * T + q) * plane) = o[q];
    }
  }
}

}  // namespace rib

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_fill(float* p, size_t n, unsigned seed, float scale) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = scale * ((float)(h & 0xffffff) / 8388608.f - 1.f);
  }
}

static int g_variant = 4;

template <int WM, int TWM, int TWN, int TNF>
void run(int H, int W, int C, int B) {
  constexpr int T = WM + 2, NP = T * T;
  constexpr int BM = 32 * TWM, BN = 32 * TNF * TWN;
  const int tilesY = (H + WM - 1) / WM, tilesX = (W + WM - 1) / WM, ntiles = tilesY * tilesX;
  float *vin, *u, *m, *mc, *v0, *v1, *m2a, *m2b, *y, *bias, *gamma, *beta;
  double* part;
  const size_t nm = (size_t)B * NP * ntiles * C;
  for (float** q : {&vin, &m, &mc, &v0, &v1, &m2a, &m2b}) CHECK(hipMalloc(q, nm * 4));
  CHECK(hipMalloc(&u, (size_t)NP * C * C * 4));
  CHECK(hipMalloc(&y, (size_t)B * H * W * C * 4)); CHECK(hipMalloc(&bias, C * 4)); CHECK(hipMalloc(&gamma, C * 4)); CHECK(hipMalloc(&beta, C * 4));
  const int units = ntiles * WM, blocks = std::max(1, std::min((units + 15) / 16, (int)STATS_MAX_PARTIALS));
  CHECK(hipMalloc(&part, (size_t)B * blocks * 2 * C * 8));
  k_fill<<<1024, 256>>>(vin, nm, 1u, 1.f); k_fill<<<1024, 256>>>(u, (size_t)NP * C * C, 9u, 0.06f);
  k_fill<<<4, 256>>>(bias, C, 2u, 0.5f); k_fill<<<4, 256>>>(gamma, C, 3u, 1.f); k_fill<<<4, 256>>>(beta, C, 4u, 0.3f);
  const int nsl = (C + 63) / 64;
  GemmCgParams g; memset(&g, 0, sizeof g);
  g.M = ntiles; g.N = C; g.K = C; g.lda = C; g.ldc = C; g.sA = (size_t)ntiles * C; g.sB = (size_t)C * C; g.sC = (size_t)ntiles * C; g.modB = NP; g.B = u;
  const dim3 ggrid((ntiles + BM - 1) / BM, (C + BN - 1) / BN, B * NP);
  WinoOutParams wo; memset(&wo, 0, sizeof wo);
  wo.m = m; wo.tilesY = tilesY; wo.tilesX = tilesX; wo.CoutPad = C; wo.bias = bias; wo.y = y; wo.yC = C; wo.Cout = C; wo.Hout = H; wo.Wout = W;
  wo.act = ACT_NONE; wo.stat_part = part; wo.nslices = nsl; wo.ublocks = blocks;
  WinoInParams wi; memset(&wi, 0, sizeof wi);
  wi.x = y; wi.H = H; wi.W = W; wi.xC = C; wi.Cin = C; wi.pro_lrelu = 1; wi.v = v0; wi.tilesY = tilesY; wi.tilesX = tilesX;
  wi.st.part = part; wi.st.tiles = blocks; wi.st.Cs = C; wi.st.inv_count = 1.f / ((float)H * W); wi.st.gamma = gamma; wi.st.beta = beta;
  wi.nslices = nsl; wi.ublocks = std::max(1, std::min((ntiles * T + 15) / 16, 512 / nsl));
  WinoBridgeParams wb; memset(&wb, 0, sizeof wb);
  wb.m = mc; wb.tilesY = tilesY; wb.tilesX = tilesX; wb.C = C; wb.bias = bias; wb.H = H; wb.W = W; wb.norm = 1; wb.inv_count = 1.f / ((float)H * W);
  wb.gamma = gamma; wb.beta = beta; wb.lrelu = 1; wb.v = v1; wb.xcd_groups = getenv("WB_XCD") ? C / 8 : 0;
  const size_t lds = (size_t)H * WM * tilesX * (g_variant == 4 ? 16 : 4);
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bridge4<WM>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)((size_t)H * WM * tilesX * 16)));
  auto gemm = [&](const float* a, float* c, int a_cg, bool c_cg) {
    GemmCgParams q = g; q.A = a; q.C = c; q.a_cg = a_cg;
    if (c_cg) k_gemm_cg<TWM, TWN, TNF, ST_F32, true><<<ggrid, 256>>>(q); else k_gemm_cg<TWM, TWN, TNF, ST_F32, false><<<ggrid, 256>>>(q);
  };
  auto path_a = [&]() {
    gemm(vin, m, 0, false);
    if (WM == 4) { k_wino4_out<<<dim3(blocks * nsl, B), 256>>>(wo); k_wino4_in<WSRC_PLAIN><<<dim3(wi.ublocks * nsl, B), 256>>>(wi); }
    else { k_wino_out<<<dim3(blocks * nsl, B), 256>>>(wo); k_wino_in<WSRC_PLAIN><<<dim3(wi.ublocks * nsl, B), 256>>>(wi); }
    gemm(v0, m2a, 0, false);
  };
  auto path_b = [&]() {
    gemm(vin, mc, 0, true);
    if (g_variant == 4) k_bridge4<WM><<<dim3(C / 4, B), 256, lds>>>(wb); else k_bridge1<WM, WB_PLAIN><<<dim3(C, B), 256, lds>>>(wb);
    gemm(v1, m2b, 1, false);
  };
  auto gemms_only = [&]() { gemm(vin, m, 0, false); gemm(v0, m2a, 0, false); };
  path_a(); path_b();
  CHECK(hipDeviceSynchronize());
  std::vector<float> a(nm), b(nm);
  CHECK(hipMemcpy(a.data(), m2a, nm * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(b.data(), m2b, nm * 4, hipMemcpyDeviceToHost));
  double md = 0, mx = 0;
  for (size_t i = 0; i < nm; ++i) { md = std::max(md, (double)std::fabs(a[i] - b[i])); mx = std::max(mx, (double)std::fabs(a[i])); }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int N = 200;
  auto timeit = [&](int mode) {
    for (int i = 0; i < 10; ++i) { if (mode == 0) gemms_only(); if (mode == 1) path_a(); if (mode == 2) path_b(); }
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < N; ++i) { if (mode == 0) gemms_only(); if (mode == 1) path_a(); if (mode == 2) path_b(); }
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); return ms * 1e3f / N;
  };
  const float t_g = timeit(0), t_a = timeit(1), t_b = timeit(2);
  float t_stage[3];
  for (int st = 1; st <= 3; ++st) { CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_wb_stop), &st, sizeof st)); t_stage[st - 1] = timeit(2); }
  { const int z = 0; CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_wb_stop), &z, sizeof z)); }
  printf("F(%dx%d) %3dx%-3d C=%3d B=%d tile %dx%d bridge%d: result max|diff| %.2e (max %.1f)   2 GEMMs %6.2f us | GEMM, out, in, GEMM %6.2f us | GEMM, bridge, GEMM %6.2f us   (bridge cut after A / statistics / B1: %.2f / %.2f / %.2f)\n",
         WM, WM, H, W, C, B, BM, BN, g_variant, md, mx, t_g, t_a, t_b, t_stage[0], t_stage[1], t_stage[2]);
  for (void* q : {(void*)vin, (void*)u, (void*)m, (void*)mc, (void*)v0, (void*)v1, (void*)m2a, (void*)m2b, (void*)y, (void*)bias, (void*)gamma, (void*)beta, (void*)part}) CHECK(hipFree(q));
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;      // one shape (for a rocprofv3 --kernel-trace run) or all
  g_variant = argc > 2 && atoi(argv[2]) == 1 ? 1 : 4;
  if (only < 0 || only == 0) run<4, 2, 2, 1>(64, 64, 256, 1);
  if (only < 0 || only == 1) run<2, 2, 2, 2>(32, 32, 512, 1);
  if (only < 0 || only == 2) run<2, 2, 2, 1>(32, 32, 512, 1);
  if (only < 0 || only == 3) run<4, 2, 2, 1>(64, 64, 256, 4);
  if (only < 0 || only == 4) run<2, 2, 2, 2>(32, 32, 512, 4);
  if (only < 0 || only == 5) run<2, 2, 2, 1>(40, 60, 256, 1);     // 320x480 frames: the mask network's level (F(2x2): 600 tiles, three per thread)
  if (only < 0 || only == 6) run<2, 2, 2, 1>(20, 30, 512, 1);
  if (only < 0 || only == 7) run<4, 2, 2, 1>(40, 60, 256, 8);
  return 0;
}
