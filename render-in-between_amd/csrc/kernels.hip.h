// kernels.hip.h — the kernels of the generator forward other than k_igemm (igemm.hip.h): the LDS-DMA GEMM, InstanceNorm
// finalize, split-K sum, Winograd transforms, SPADE modulate, pooling, the residual join, the heads, the first-layer convolution
// over the caller's NCHW tensors, pack / blend / quantise / warp.  Compiled into rib.o only.
#pragma once
#include "igemm.hip.h"

namespace rib {

// ---------------------------------------------------------------------------------------------
// k_gemm_dma (round 3): the plain GEMMs of the frame - the 16 / 36 batched Winograd-domain GEMMs of a deep 3x3 layer and
// the gamma/beta GEMM of a condition level - with their operand tiles staged by LDS-DMA.
//   C[z][M][N] = sum_k A[z][M][K] * B[z % modB][N][K]      fp32, v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 sums)
// A rows are pixels / Winograd tiles with K = channels contiguous (row stride lda), B rows are output channels with K
// contiguous (the filter layouts of the blob and of the Winograd sets as they are).
// k_igemm stages its tiles global -> registers -> LDS (ds_write_b128) and double-buffers through registers; here every
// wave issues global_load_lds_dwordx4 (gfx950: 16 bytes per lane straight into LDS, no staging registers, no ds_write):
// one instruction lands 64 consecutive 16-byte slots = 8 rows of a 32-float K chunk, so rows cannot be padded against bank
// conflicts; the slot of a row is XOR-swizzled instead (slot' = slot ^ ((row >> 1) & 7): conflict-free for the lane groups
// of ds_read_b128).  Two LDS stages; per chunk: s_waitcnt vmcnt(0), one barrier, the fills of the next chunk, then the MFMAs
// of this one run over them.  tools/probes/ldsdma_probe.hip, same process, same problems: 74.9 vs 90.8 us on the level GEMM
// (115 vs 95 TFLOP/s), 22.4 vs 36.7 us on the 16-position 512-channel GEMM, 279 vs 306 us on a 34 GFLOP launch; a
// four-stage ring was slower than two stages (LDS: fewer workgroups per CU).
// The DMA instruction is inline assembly on purpose: the compiler's wait-count pass cannot tell LDS stages apart and puts
// s_waitcnt vmcnt(0) in front of every ds_read that follows a builtin fill - the prefetch would never overlap anything.
// Rows beyond M / N are clamped on the load side (valid addresses, values unused) and skipped on the store side; K is a
// multiple of 32.  Workgroup = 4 waves as WM x WN; a wave owns 32 rows x (32 NF) columns; tile (32 WM) x (32 NF WN).
// grid (ceil(M / BM), ceil(N / BN), Z).
// ---------------------------------------------------------------------------------------------
struct GemmDmaParams {
  const float* A; const float* B; float* C;
  int M, N, K, lda, ldc;
  size_t sA, sB, sC;      // element strides of the batch index z (B: of z % modB)
  int modB;               // 0: one B for every z
};

// ST = ST_BF16 / ST_F16: A and B hold 16-bit elements (lda, K, sA, sB in elements), a 128-byte row chunk is 64 of them, a
// lane's 16-byte slot feeds ONE v_mfma_f32_32x32x16 (k = 8 per lane half) where it feeds four fp32 MFMAs; C stays fp32 (the
// level slab k_spade_modulate reads).  Same tile, same swizzle, same two stages.
//
// X3 (round 6, rib_set_products(RIB_PRODUCTS_BF16X3), OPT-IN, fp32 storage only): the same fp32 tiles, but every operand value
// is split after its ds_read into hi + mid + lo - three bf16 numbers by truncation, the two subtractions are exact - and a
// product becomes six v_mfma_f32_32x32x16_bf16 (hh, hm, mh, hl, lh, mm; the dropped terms ml, lm, ll are below 2^-24 |a b|)
// accumulated in fp32: 3/8 of the matrix-pipe time of the exact-fp32 MFMAs, paid for with 11 vector instructions per pair of
// values.  X3 = 2 keeps the hh products in an accumulator of their own, so the five small terms never round against the large
// sum.  Against an fp64 GEMM the rms error is 0.87 (X3 = 1) / 0.38 (X3 = 2) of the exact-fp32 MFMA chain's (one rounding
// per MFMA instead of one per product); tools/probes/gemm_x3_probe.hip, profiles/r06_gemm_x3_probe.txt.  It is NOT the
// reference's arithmetic, hence never the default and always named in bench.py's workload string.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split3_bf16(const float4 p, const float4 q, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  const float x[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
  u32x4 h, m, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t u0 = __float_as_uint(x[2 * i]), u1 = __float_as_uint(x[2 * i + 1]);
    h[i] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);      // the high halves of two words = two truncated bf16 values
    const float r0 = x[2 * i] - __uint_as_float(u0 & 0xffff0000u), r1 = x[2 * i + 1] - __uint_as_float(u1 & 0xffff0000u);
    const uint32_t v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m[i] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    l[i] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
  }
  hi = __builtin_bit_cast(bf16x8, h); mid = __builtin_bit_cast(bf16x8, m); lo = __builtin_bit_cast(bf16x8, l);
}

template <int WM, int WN, int NF, int ST = ST_F32, int X3 = 0>
__global__ __launch_bounds__(256) void k_gemm_dma(const GemmDmaParams p) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(X3 == 0 || ST == ST_F32, "split products: fp32 storage");
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
  f32x16 acc_s[X3 == 2 ? NF : 1];      // X3 = 2: the five small product terms
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[nf][r] = 0.f; if constexpr (X3 == 2) acc_s[nf][r] = 0.f; }
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
    src[j] = isA ? A + (size_t)min(m0 + row, p.M - 1) * (p.lda / EPW) + ls * 4 : B + (size_t)min(n0 + row, p.N - 1) * (p.K / EPW) + ls * 4;
  }
  auto fill = [&](int st, int kc) {
#pragma unroll
    for (int j = 0; j < NFILL; ++j) {
      const uint32_t dst = lds0 + (uint32_t)(st * STAGE + (wave * 8 + 32 * j) * BK) * 4u;      // wave-uniform byte address
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src[j] + kc) : "memory");
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
    if constexpr (X3 != 0) {
#pragma unroll
      for (int ks = 0; ks < BK / 16; ++ks) {
        // this lane's 8 k values of the 16-channel step: slots 4 ks + lh and 4 ks + 2 + lh (which 8 does not matter: A and B agree)
        const int s0 = ks * 4 + lh, s1 = s0 + 2;
        const int ra = wm * 32 + li;
        bf16x8 ah, am, al;
        split3_bf16(*reinterpret_cast<const float4*>(sA + ra * BK + (s0 ^ ((ra >> 1) & 7)) * 4),
                    *reinterpret_cast<const float4*>(sA + ra * BK + (s1 ^ ((ra >> 1) & 7)) * 4), ah, am, al);
#pragma unroll
        for (int nf = 0; nf < NF; ++nf) {
          const int rb = (wn * NF + nf) * 32 + li;
          bf16x8 bh, bm, bl;
          split3_bf16(*reinterpret_cast<const float4*>(sB + rb * BK + (s0 ^ ((rb >> 1) & 7)) * 4),
                      *reinterpret_cast<const float4*>(sB + rb * BK + (s1 ^ ((rb >> 1) & 7)) * 4), bh, bm, bl);
          f32x16& sm = X3 == 2 ? acc_s[nf] : acc[nf];      // small terms first
          sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, sm, 0, 0, 0);
          sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, sm, 0, 0, 0);
          sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, sm, 0, 0, 0);
          sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, sm, 0, 0, 0);
          sm = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, sm, 0, 0, 0);
          acc[nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[nf], 0, 0, 0);
        }
      }
      continue;
    }
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
        if constexpr (ST == ST_F32) {
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
  // accumulator element r of lane l: row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5), column l & 31
  float* C = p.C + (size_t)z * p.sC;
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int col = n0 + (wn * NF + nf) * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float v = acc[nf][r];
      if constexpr (X3 == 2) v += acc_s[nf][r];
      if (row < p.M && col < p.N) C[(size_t)row * p.ldc + col] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_stats_finalize: per-tile partial sums -> (scale, shift) of the InstanceNorm that follows.
//   mean = S1/N, var = S2/N - mean^2 (biased), rstd = 1/sqrt(var+eps)
//   scale = rstd*gamma, shift = beta - mean*scale   (gamma=1, beta=0 when affine is absent)
// The partials are fp64 sums made by the producers' epilogues (fp64 from the first add) and are summed here in a fixed
// order in fp64: deterministic, and E[x^2] - mean^2 keeps the variance for |mean| / std up to ~1e6.
// grid (Cs/16, B), block 1024 = 16 channels x 64 tile slices.
// ---------------------------------------------------------------------------------------------
struct FinalizeParams {
  const double* part;  // [B][tiles][2][Cs]
  int tiles, Cs, C;    // C valid channels
  const float* gamma;  // [C] or nullptr
  const float* beta;
  float* scale;        // [B][ld] at channel offset off
  float* shift;
  int ld, off;
  float inv_count;     // 1 / (H*W)
  float eps;
  // paired producer (IgemmParams::pair): sample n belongs to convolution n & 1, whose IN affine lies g_stride floats behind
  // the first one's; pair_merge: the two rows of a pair go into ONE row n >> 1 of the arrays, at channel offsets off and
  // off + pair_off (the concatenated tensor's statistics)
  int g_stride, pair_merge, pair_off;
};

// CH = 16: block = 16 channels x 64 tile slices, grid (Cs / 16, B).  CH = 4 (launches with more than 512 partials per channel: the
// full-resolution layers): 4 channels x 256 slices, grid (Cs / 4, B) - with 32 channels the 16-channel shape runs on TWO CUs and
// every thread walks 16-32 rows (9.6 us for 2048 partials in the bf16 plan); four times the workgroups and a quarter of the walk.
// The sums are fixed-order either way; the two shapes associate them differently (fp64: not visible in the fp32 scale / shift).
template <int CH>
__global__ __launch_bounds__(1024) void k_stats_finalize(const FinalizeParams p) {
  static_assert(CH == 16 || CH == 4, "16 channels x 64 slices or 4 channels x 256 slices");
  constexpr int SL = 1024 / CH;
  __shared__ double red[2][16][CH];
  const int cl = threadIdx.x % CH, sl = threadIdx.x / CH;
  const int c = blockIdx.x * CH + cl;
  const int n = blockIdx.y;
  double a1 = 0.0, a2 = 0.0;
  if (c < p.Cs) stats_from_partials(p.part + (size_t)n * p.tiles * 2 * p.Cs, p.tiles, p.Cs, c, sl, SL, a1, a2);
  // fixed-shape reduction over the slices: the slices of a wavefront with shuffles, then the 16
  // wavefront sums in order by one thread per channel (one barrier; an LDS tree needed seven)
#pragma unroll
  for (int off = CH; off < 64; off <<= 1) { a1 += __shfl_xor(a1, off); a2 += __shfl_xor(a2, off); }
  if ((threadIdx.x & 63) < CH) { red[0][threadIdx.x >> 6][cl] = a1; red[1][threadIdx.x >> 6][cl] = a2; }
  __syncthreads();
  if (sl == 0) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { t1 += red[0][w][cl]; t2 += red[1][w][cl]; }
    red[0][0][cl] = t1; red[1][0][cl] = t2;
  }
  if (sl == 0 && c < p.C) {
    const double s1 = red[0][0][cl], s2 = red[1][0][cl];
    const double mean = s1 * (double)p.inv_count;
    double var = s2 * (double)p.inv_count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float rstd = (float)(1.0 / sqrt(var + (double)p.eps));
    const int gs = (n & 1) * p.g_stride;
    const float g = p.gamma ? p.gamma[gs + c] : 1.f;
    const float b = p.beta ? p.beta[gs + c] : 0.f;
    const float sc = rstd * g;
    const size_t at = p.pair_merge ? (size_t)(n >> 1) * p.ld + p.off + (n & 1) * p.pair_off + c : (size_t)n * p.ld + p.off + c;
    p.scale[at] = sc;
    p.shift[at] = b - (float)mean * sc;
  }
}

// ---------------------------------------------------------------------------------------------
// k_splitk_epilogue: finishes a split-K convolution: sums the partial slabs in a fixed order, then
// the same epilogue as k_igemm (bias, residual, activation, store, per-block statistics partials).
// grid (blocks, B); thread = (pixel slot, 4-channel group); block covers slots*4 pixels.
// ---------------------------------------------------------------------------------------------
struct SplitEpiParams {
  const float* slab; int ksplit; int B;
  const float* bias; int CoutPad;
  float* y; int yC, yoff, Cout;
  int act;
  const float* res; int resC, res_ups;
  double* stat_part; int blocks;
  int Hout, Wout;
};

template <int ST>
__global__ __launch_bounds__(256) void k_splitk_epilogue(const SplitEpiParams p) {
  __shared__ __attribute__((aligned(16))) double red[2][256][4];
  const int c4n = p.CoutPad / 4;
  const int slots = 256 / c4n;
  const int c4 = threadIdx.x % c4n, slot = threadIdx.x / c4n;
  const int n = blockIdx.y;
  const int npix = p.Hout * p.Wout;
  const int ppb = slots * 4;
  const size_t sstride = (size_t)p.B * npix * p.CoutPad;
  const float4 bv = *reinterpret_cast<const float4*>(p.bias + c4 * 4);
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  // all slab reads of the block's four pixel rounds are issued before the first add: one memory round trip
  // for split factors <= 4 (a load -> add loop per pixel paid one per pixel and per four slabs); the slabs are
  // still added in the fixed order 0, 1, 2, ...
  float4 acc4[4];
  {
    float4 v[4][4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pix = blockIdx.x * ppb + k * slots + slot;
      const float* src = p.slab + ((size_t)n * npix + (pix < npix ? pix : 0)) * p.CoutPad + c4 * 4;
#pragma unroll
      for (int s = 0; s < 4; ++s) v[k][s] = s < p.ksplit ? *reinterpret_cast<const float4*>(src + s * sstride) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float4 a = v[k][0];
#pragma unroll
      for (int s = 1; s < 4; ++s)
        if (s < p.ksplit) { a.x += v[k][s].x; a.y += v[k][s].y; a.z += v[k][s].z; a.w += v[k][s].w; }
      acc4[k] = a;
    }
    for (int s0 = 4; s0 < p.ksplit; s0 += 4) {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int pix = blockIdx.x * ppb + k * slots + slot;
        const float* src = p.slab + ((size_t)n * npix + (pix < npix ? pix : 0)) * p.CoutPad + c4 * 4;
#pragma unroll
        for (int s = 0; s < 4; ++s) v[k][s] = s0 + s < p.ksplit ? *reinterpret_cast<const float4*>(src + (s0 + s) * sstride) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int s = 0; s < 4; ++s)
          if (s0 + s < p.ksplit) { acc4[k].x += v[k][s].x; acc4[k].y += v[k][s].y; acc4[k].z += v[k][s].z; acc4[k].w += v[k][s].w; }
    }
  }
  // residual reads of the four pixel rounds as one batch (clamped addresses), then values, then the stores: a
  // read inside the store loop waits for the previous store's acknowledgement (see k_igemm's epilogue)
  float rres[4][4];
  if (p.res) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pix = min(blockIdx.x * ppb + k * slots + slot, npix - 1);
      const int oy = pix / p.Wout, ox = pix % p.Wout;
      const size_t rpix = p.res_ups ? ((size_t)n * (p.Hout >> 1) + (oy >> 1)) * (p.Wout >> 1) + (ox >> 1) : (size_t)n * npix + pix;
#pragma unroll
      for (int e = 0; e < 4; ++e) rres[k][e] = ld_act<ST>(p.res, rpix * p.resC + min(c4 * 4 + e, p.resC - 1));
    }
  }
  float vals[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int pix = blockIdx.x * ppb + k * slots + slot;
    const float4 a = acc4[k];
    const float v[4] = {a.x + bv.x, a.y + bv.y, a.z + bv.z, a.w + bv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float t = v[e];
      if (p.res) t += rres[k][e];
      t = apply_act(t, p.act);
      if constexpr (ST != ST_F32) t = round16<ST>(t);      // statistics of the tensor as it is stored
      vals[k][e] = (pix < npix && c4 * 4 + e < p.Cout) ? t : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) { s1[e] += (double)vals[k][e]; s2[e] += (double)vals[k][e] * (double)vals[k][e]; }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int pix = blockIdx.x * ppb + k * slots + slot;
    if (pix < npix) {
      const size_t opix = (size_t)n * npix + pix;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (c4 * 4 + e < p.Cout) st_act<ST>(p.y, opix * p.yC + p.yoff + c4 * 4 + e, vals[k][e]);
    }
  }
  if (p.stat_part) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { red[0][threadIdx.x][e] = s1[e]; red[1][threadIdx.x][e] = s2[e]; }
    __syncthreads();
    for (int c = threadIdx.x; c < p.CoutPad; c += 256) {
      const int g = c / 4, e = c % 4;
      double a1 = 0.0, a2 = 0.0;
      for (int s = 0; s < slots; ++s) { a1 += red[0][s * c4n + g][e]; a2 += red[1][s * c4n + g][e]; }
      double* dst = p.stat_part + (((size_t)n * p.blocks + blockIdx.x) * 2) * p.CoutPad;
      dst[c] = a1;
      dst[p.CoutPad + c] = a2;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// Consumer-side InstanceNorm finalize for the light elementwise kernels (round 3).  k_wino_in / k_wino4_in,
// k_spade_modulate and k_in_add work on CHANNEL SLICES of at most 64 channels per workgroup, so a workgroup can reduce the
// producer's per-tile partial sums of its own channels (<= STATS_MAX_PARTIALS tiles x 64 channels x 2 doubles, L2-resident)
// at kernel start instead of reading (scale, shift) arrays written by a k_stats_finalize launch: one dependent launch
// less per normalised tensor.  Same arithmetic as k_stats_finalize (fp64, fixed order, biased variance, eps 1e-5).
// 256 threads: thread = (channel j = tid % 64, partial slice q = tid / 64); slice q sums tiles q, q + 4, ...; the four
// slice sums are added in the order 0..3.  Results in s_sc / s_sh[64]; ends with a barrier.
// ---------------------------------------------------------------------------------------------
struct StatSrc {
  const double* part;      // [B][tiles][2][Cs] or nullptr (then the consumer reads its (scale, shift) arrays)
  int tiles, Cs;
  float inv_count;
  const float* gamma;      // IN affine of the producing layer (indexed by channel) or nullptr
  const float* beta;
};

__device__ __forceinline__ void block_stats_64(const StatSrc& s, int n, int c0, int nch, double* red /* [4][64][2] */,
                                               float* s_sc, float* s_sh) {
  const int j = threadIdx.x & 63, q = threadIdx.x >> 6;
  double a1 = 0.0, a2 = 0.0;
  if (j < nch) stats_from_partials(s.part + (size_t)n * s.tiles * 2 * s.Cs, s.tiles, s.Cs, c0 + j, q, 4, a1, a2);
  red[(q * 64 + j) * 2] = a1; red[(q * 64 + j) * 2 + 1] = a2;
  __syncthreads();
  if (q == 0 && j < nch) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { t1 += red[(k * 64 + j) * 2]; t2 += red[(k * 64 + j) * 2 + 1]; }
    float sc, sh;
    scale_shift_of(t1, t2, s.inv_count, s.gamma ? s.gamma[c0 + j] : 1.f, s.beta ? s.beta[c0 + j] : 0.f, sc, sh);
    s_sc[j] = sc; s_sh[j] = sh;
  }
  __syncthreads();
}

// ---------------------------------------------------------------------------------------------
// Winograd F(2x2, 3x3) for the 3x3 stride-1 convolutions on the <= 64x64 maps (round 2).  At batch 1 those layers
// cannot fill 256 CUs (a 32x32 map is 8 spatial tiles) and run as split-K launches of short workgroups; they are half
// of the frame.  In the Winograd domain the convolution is 16 independent GEMMs [2x2-tiles x Cin] . [Cin x Cout] - one
// per position of the 4x4 transformed tile - with 4/9 of the multiplications:
//     V = B^T d B  (k_wino_in: 4x4 input tile d at offset (2ty-1, 2tx-1), prologue + zero padding applied first)
//     M[xi] = V[xi] . U[xi],  U = G g G^T folded on the host (rib_finalize_weights)     (ONE k_igemm launch, KS = 1,
//             16*B "samples" of a tilesY x tilesX "image", filter set per sample: IgemmParams::w_mod)
//     Y = A^T M A  (k_wino_out: 2x2 outputs per tile, + bias, residual, activation, statistics partials)
// The GEMM launch has 16x the independent rows of the direct convolution, so it fills the chip without split-K, and V
// and M (4x the activation each) stay in L2 / the memory-side cache at these sizes.  fp32 throughout: the transforms
// only add and halve, max |diff| to the direct kernel ~3e-6 on O(1) outputs (round 1's probe).
// Layouts: V [B*16][tilesY][tilesX][Cin], M [B*16][tilesY][tilesX][CoutPad], sample index = n*16 + xi, xi = 4*row + col.
// Work split of the transforms (round 3): a workgroup owns a slice of 64 channels (16 float4 lanes) and 16 (tile, row)
// units per pass; blockIdx.x = unit block * nslices + slice.  The output transforms emit ONE statistics partial per unit
// block (<= 64 of them), so that the consumer of the normalised tensor can reduce them itself (block_stats_64).
// ---------------------------------------------------------------------------------------------
enum { WSRC_PLAIN = 0, WSRC_SPADE = 1, WSRC_JOIN = 2 };
struct WinoInParams {
  const float* x; int H, W, xC, Cin;        // input activation [B][H][W][xC], Cin channels used (multiple of 4)
  const float* pro_scale; const float* pro_shift; int pro_ld, pro_lrelu;   // optional prologue (as k_igemm's)
  StatSrc st;                                // consumer-side finalize: replaces pro_scale / pro_shift when st.part != nullptr
  float* v; int tilesY, tilesX;
  int nslices, ublocks;                      // grid.x = ublocks * nslices (channel slices of 64)
  // WSRC_SPADE: the input is the output of an unfused SPADE that is never stored (round 3): x is the tensor being
  // normalised (at half resolution when x_ups), (scale, shift) its InstanceNorm, gamma/beta come from the condition level's
  // slab (k_spade_modulate's arithmetic, spade_mod1), then LeakyReLU when pro_lrelu
  int x_ups; const float* slab; int slab_ld, col0; const float* sbias;
  // WSRC_JOIN: the input is the mask network's residual join (k_in_add's arithmetic), out = IN(x) + (x2 ? IN2(x2) : xres),
  // which the unit that owns a pixel also stores to o (the next join's residual); no activation
  const float* x2; const float* xres; float* o; const float* pro2_scale; const float* pro2_shift; StatSrc st2;
};

// out = IN(x) * (1 + gamma) + beta with the roundings pinned (two fused multiply-adds), shared by k_spade_modulate and the
// Winograd input transforms that apply the modulation on the fly: bit-identical either way
__device__ __forceinline__ float spade_mod1(float x, float sc, float sh, float g, float bg, float b, float bb) {
  return fmaf(fmaf(x, sc, sh), 1.f + (g + bg), b + bb);
}

template <int MODE> struct WinoConsts { float4 sc, sh, bg, bb, sc2, sh2; bool aff; };
template <int MODE> struct WinoRaw { float4 a, b, c; };

// per-thread constants of a Winograd input transform: (scale, shift) of this thread's four channels from the workgroup's
// own reduction, from the arrays, or identity; SPADE: the gamma/beta bias; JOIN: the second tensor's (scale, shift)
template <int MODE>
__device__ __forceinline__ void wino_in_consts(const WinoInParams& p, int n, int slice, int c4, bool cok, double* red, float* s_sc,
                                               float* s_sh, WinoConsts<MODE>& k) {
  k.sc = make_float4(1.f, 1.f, 1.f, 1.f); k.sh = make_float4(0.f, 0.f, 0.f, 0.f);
  k.aff = false;
  const int l = (threadIdx.x & 15) * 4;
  if (p.st.part) {
    block_stats_64(p.st, n, slice * 64, min(64, p.Cin - slice * 64), red, s_sc, s_sh);
    k.sc = make_float4(s_sc[l], s_sc[l + 1], s_sc[l + 2], s_sc[l + 3]);
    k.sh = make_float4(s_sh[l], s_sh[l + 1], s_sh[l + 2], s_sh[l + 3]);
    k.aff = true;
  } else if (p.pro_scale && cok) {
    k.sc = *reinterpret_cast<const float4*>(p.pro_scale + (size_t)n * p.pro_ld + c4 * 4);
    k.sh = *reinterpret_cast<const float4*>(p.pro_shift + (size_t)n * p.pro_ld + c4 * 4);
    k.aff = true;
  }
  if constexpr (MODE == WSRC_SPADE) {
    const int v = c4 * 4, colg = (v / 32) * 64 + (v % 32);
    k.bg = cok ? *reinterpret_cast<const float4*>(p.sbias + colg) : k.sh;
    k.bb = cok ? *reinterpret_cast<const float4*>(p.sbias + colg + 32) : k.sh;
  }
  if constexpr (MODE == WSRC_JOIN) {
    k.sc2 = make_float4(1.f, 1.f, 1.f, 1.f); k.sh2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.x2) {
      if (p.st2.part) {
        __syncthreads();                       // s_sc / s_sh of the first tensor have been read by every thread
        block_stats_64(p.st2, n, slice * 64, min(64, p.Cin - slice * 64), red, s_sc, s_sh);
        k.sc2 = make_float4(s_sc[l], s_sc[l + 1], s_sc[l + 2], s_sc[l + 3]);
        k.sh2 = make_float4(s_sh[l], s_sh[l + 1], s_sh[l + 2], s_sh[l + 3]);
      } else if (cok) {
        k.sc2 = *reinterpret_cast<const float4*>(p.pro2_scale + (size_t)n * p.pro_ld + c4 * 4);
        k.sh2 = *reinterpret_cast<const float4*>(p.pro2_shift + (size_t)n * p.pro_ld + c4 * 4);
      }
    }
  }
}

// the loads of one input element (clamped coordinates cy, cx: always valid addresses)
template <int MODE>
__device__ __forceinline__ void wino_fetch(const WinoInParams& p, int n, int cy, int cx, int c4, WinoRaw<MODE>& r) {
  if constexpr (MODE == WSRC_SPADE) {
    const int Hs = p.x_ups ? p.H >> 1 : p.H, Ws = p.x_ups ? p.W >> 1 : p.W;
    const int sy = p.x_ups ? cy >> 1 : cy, sx = p.x_ups ? cx >> 1 : cx;
    r.a = *reinterpret_cast<const float4*>(p.x + (((size_t)n * Hs + sy) * Ws + sx) * p.xC + c4 * 4);
    const int v = c4 * 4, colg = (v / 32) * 64 + (v % 32);
    const float* sl = p.slab + (((size_t)n * p.H + cy) * p.W + cx) * p.slab_ld + p.col0 + colg;
    r.b = *reinterpret_cast<const float4*>(sl);
    r.c = *reinterpret_cast<const float4*>(sl + 32);
  } else {
    const size_t e = (((size_t)n * p.H + cy) * p.W + cx) * p.xC + c4 * 4;
    r.a = *reinterpret_cast<const float4*>(p.x + e);
    if constexpr (MODE == WSRC_JOIN) r.b = *reinterpret_cast<const float4*>((p.x2 ? p.x2 : p.xres) + e);
  }
}

// the value the convolution sees at an in-image element (zero padding is the caller's)
template <int MODE>
__device__ __forceinline__ float4 wino_value(const WinoInParams& p, const WinoConsts<MODE>& k, const WinoRaw<MODE>& r) {
  float4 w = r.a;
  if constexpr (MODE == WSRC_SPADE) {
    w = make_float4(spade_mod1(r.a.x, k.sc.x, k.sh.x, r.b.x, k.bg.x, r.c.x, k.bb.x), spade_mod1(r.a.y, k.sc.y, k.sh.y, r.b.y, k.bg.y, r.c.y, k.bb.y),
                    spade_mod1(r.a.z, k.sc.z, k.sh.z, r.b.z, k.bg.z, r.c.z, k.bb.z), spade_mod1(r.a.w, k.sc.w, k.sh.w, r.b.w, k.bg.w, r.c.w, k.bb.w));
    if (p.pro_lrelu) w = lrelu4(w);
  } else if constexpr (MODE == WSRC_JOIN) {
    w = make_float4(w.x * k.sc.x + k.sh.x, w.y * k.sc.y + k.sh.y, w.z * k.sc.z + k.sh.z, w.w * k.sc.w + k.sh.w);
    if (p.x2) { w.x += r.b.x * k.sc2.x + k.sh2.x; w.y += r.b.y * k.sc2.y + k.sh2.y; w.z += r.b.z * k.sc2.z + k.sh2.z; w.w += r.b.w * k.sc2.w + k.sh2.w; }
    else { w.x += r.b.x; w.y += r.b.y; w.z += r.b.z; w.w += r.b.w; }
  } else {
    if (k.aff) w = make_float4(w.x * k.sc.x + k.sh.x, w.y * k.sc.y + k.sh.y, w.z * k.sc.z + k.sh.z, w.w * k.sc.w + k.sh.w);
    if (p.pro_lrelu) w = lrelu4(w);
  }
  return w;
}

#define RIB_F4_SUB(a, b) make_float4((a).x - (b).x, (a).y - (b).y, (a).z - (b).z, (a).w - (b).w)
#define RIB_F4_ADD(a, b) make_float4((a).x + (b).x, (a).y + (b).y, (a).z + (b).z, (a).w + (b).w)

template <int MODE>
__global__ __launch_bounds__(256) void k_wino_in(const WinoInParams p) {
  // thread = ((tile, row r of the transformed tile), 4 channels of the slice): a row needs two input rows (8 elements)
  __shared__ double red[4 * 64 * 2];
  __shared__ float s_sc[64], s_sh[64];
  const int c4n = p.Cin / 4;
  const int n = blockIdx.y;
  const int slice = blockIdx.x % p.nslices, ub = blockIdx.x / p.nslices;
  const int c4 = slice * 16 + (threadIdx.x & 15);
  const bool cok = c4 < c4n;
  const int ntiles = p.tilesY * p.tilesX;
  WinoConsts<MODE> kc;
  wino_in_consts<MODE>(p, n, slice, c4, cok, red, s_sc, s_sh, kc);
  if (!cok) return;
  for (int u = ub * 16 + (threadIdx.x >> 4); u < ntiles * 4; u += p.ublocks * 16) {
    const int r = u & 3, tile = u >> 2;
    const int ty = tile / p.tilesX, tx = tile % p.tilesX;
    // B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: row r of B^T d is d[ia] + sg * d[ib]
    const int ia = r == 0 ? 0 : (r == 2 ? 2 : 1), ib = r == 0 ? 2 : (r == 1 ? 2 : (r == 2 ? 1 : 3));
    const float sg = r == 1 ? 1.f : -1.f;
    WinoRaw<MODE> d[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int iy = 2 * ty - 1 + (a ? ib : ia), ix = 2 * tx - 1 + q;
        wino_fetch<MODE>(p, n, min(max(iy, 0), p.H - 1), min(max(ix, 0), p.W - 1), c4, d[a][q]);
      }
    float4 t[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 v[2];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int iy = 2 * ty - 1 + (a ? ib : ia), ix = 2 * tx - 1 + q;
        const bool inb = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        float4 w = wino_value<MODE>(p, kc, d[a][q]);
        if constexpr (MODE == WSRC_JOIN) {      // row r = 1 reads the tile's own 2x2 pixels (rows 1, 2; columns 1, 2): it stores them
          if (r == 1 && (q == 1 || q == 2) && inb) *reinterpret_cast<float4*>(p.o + (((size_t)n * p.H + iy) * p.W + ix) * p.xC + c4 * 4) = w;
        }
        if (!inb) w = make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding after the prologue
        v[a] = w;
      }
      t[q] = make_float4(v[0].x + sg * v[1].x, v[0].y + sg * v[1].y, v[0].z + sg * v[1].z, v[0].w + sg * v[1].w);
    }
    const size_t plane = (size_t)ntiles * p.Cin;              // one position's [tiles][Cin] matrix
    float* vb = p.v + (size_t)n * 16 * plane + (size_t)tile * p.Cin + c4 * 4;
    *reinterpret_cast<float4*>(vb + (size_t)(r * 4 + 0) * plane) = RIB_F4_SUB(t[0], t[2]);
    *reinterpret_cast<float4*>(vb + (size_t)(r * 4 + 1) * plane) = RIB_F4_ADD(t[1], t[2]);
    *reinterpret_cast<float4*>(vb + (size_t)(r * 4 + 2) * plane) = RIB_F4_SUB(t[2], t[1]);
    *reinterpret_cast<float4*>(vb + (size_t)(r * 4 + 3) * plane) = RIB_F4_SUB(t[1], t[3]);
  }
}

struct WinoOutParams {
  const float* m; int tilesY, tilesX, CoutPad;   // M [B*16][tiles][CoutPad]
  const float* bias;
  float* y; int yC, yoff, Cout, Hout, Wout;
  int act;
  const float* res; int resC;
  double* stat_part;                              // [B][ublocks][2][CoutPad]: one partial per unit block
  int nslices, ublocks;                           // grid.x = ublocks * nslices (channel slices of 64)
};

// the statistics partial of one (unit block, channel slice) workgroup: thread (unit slot us = tid >> 4, float4 lane cl = tid & 15)
// holds s1 / s2 of its four channels; the 16 unit slots are summed in a fixed order by one thread per channel
__device__ __forceinline__ void wino_out_partials(double (*red)[256][4], const double s1[4], const double s2[4], double* stat_part,
                                                  int n, int ublocks, int ub, int slice, int CoutPad) {
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[0][threadIdx.x][e] = s1[e]; red[1][threadIdx.x][e] = s2[e]; }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int c = slice * 64 + threadIdx.x;
    if (c < CoutPad) {
      const int g = threadIdx.x >> 2, e = threadIdx.x & 3;
      double a1 = 0.0, a2 = 0.0;
#pragma unroll
      for (int s = 0; s < 16; ++s) { a1 += red[0][s * 16 + g][e]; a2 += red[1][s * 16 + g][e]; }
      double* dst = stat_part + (((size_t)n * ublocks + ub) * 2) * CoutPad;
      dst[c] = a1;
      dst[CoutPad + c] = a2;
    }
  }
}

// grid (ublocks * nslices, B); thread = ((tile, output row r of the 2x2 tile), 4 channels of the slice)
__global__ __launch_bounds__(256) void k_wino_out(const WinoOutParams p) {
  __shared__ __attribute__((aligned(16))) double red[2][256][4];
  const int c4n = p.CoutPad / 4;
  const int slice = blockIdx.x % p.nslices, ub = blockIdx.x / p.nslices;
  const int c4 = slice * 16 + (threadIdx.x & 15);
  const int n = blockIdx.y;
  const int ntiles = p.tilesY * p.tilesX;
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  if (c4 < c4n) {
    const float4 bv = *reinterpret_cast<const float4*>(p.bias + c4 * 4);
    const size_t plane = (size_t)ntiles * p.CoutPad;
    for (int u = ub * 16 + (threadIdx.x >> 4); u < ntiles * 2; u += p.ublocks * 16) {
      const int r = u & 1, tile = u >> 1;
      const float* mb = p.m + (size_t)n * 16 * plane + (size_t)tile * p.CoutPad + c4 * 4;
      // A^T = [1 1 1 0; 0 1 -1 -1]: output row r combines the rows r, r+1, r+2 of M: (+, +, +) for r = 0, (+, -, -) for r = 1
      float4 mm[3][4];
#pragma unroll
      for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) mm[a][q] = *reinterpret_cast<const float4*>(mb + (size_t)((r + a) * 4 + q) * plane);
      const float sg = r ? -1.f : 1.f;
      float4 uu[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        uu[q] = make_float4(mm[0][q].x + sg * (mm[1][q].x + mm[2][q].x), mm[0][q].y + sg * (mm[1][q].y + mm[2][q].y),
                            mm[0][q].z + sg * (mm[1][q].z + mm[2][q].z), mm[0][q].w + sg * (mm[1][q].w + mm[2][q].w));
      float4 yv[2];
      yv[0] = RIB_F4_ADD(RIB_F4_ADD(uu[0], uu[1]), uu[2]);
      yv[1] = RIB_F4_SUB(RIB_F4_SUB(uu[1], uu[2]), uu[3]);
      const int ty = tile / p.tilesX, tx = tile % p.tilesX;
      const int oy = 2 * ty + r;
      float rr[2][4];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int cy = min(oy, p.Hout - 1), cx = min(2 * tx + q, p.Wout - 1);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          rr[q][e] = p.res ? p.res[(((size_t)n * p.Hout + cy) * p.Wout + cx) * p.resC + min(c4 * 4 + e, p.resC - 1)] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int ox = 2 * tx + q;
        const bool inb = oy < p.Hout && ox < p.Wout;
        const float v4[4] = {yv[q].x + bv.x, yv[q].y + bv.y, yv[q].z + bv.z, yv[q].w + bv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = apply_act(v4[e] + rr[q][e], p.act);
          const bool ok = inb && c4 * 4 + e < p.Cout;
          if (ok) p.y[(((size_t)n * p.Hout + oy) * p.Wout + ox) * p.yC + p.yoff + c4 * 4 + e] = t;
          t = ok ? t : 0.f;
          s1[e] += (double)t; s2[e] += (double)t * (double)t;
        }
      }
    }
  }
  if (p.stat_part) wino_out_partials(red, s1, s2, p.stat_part, n, p.ublocks, ub, slice, p.CoutPad);
}

// ---------------------------------------------------------------------------------------------
// Winograd F(4x4, 3x3): 36 positions, 4x4 outputs per tile, 1/4 of the nine-tap multiplications (F(2x2): 4/9) and 2.25
// transformed values per activation instead of 4.  Interpolation points {0, +-3/4, +-3/2, inf} instead of the textbook
// {0, +-1, +-2}: every entry of B^T and A^T stays dyadic (exact in fp32) and the fp32 error of a 256-channel layer is
// 9e-6 max / 1.4e-6 rms on O(4) outputs against 4.9e-5 / 2.9e-6 for the textbook points, 2.7e-6 / 4.6e-7 for F(2x2) and
// 1.4e-6 for the direct fp32 convolution (numpy model of the three stages, tools/probes/wino_points.py).  G (thirds) is
// folded into U on the host in fp64.  Same three launches and layouts as F(2x2) with 36 in place of 16:
// V [B*36][tilesY][tilesX][Cin], M [B*36][tilesY][tilesX][CoutPad], sample = n*36 + 6*row + col, tile origin (4ty-1, 4tx-1).
// ---------------------------------------------------------------------------------------------
__device__ __constant__ float kWino4BT[6][6] = {
    {81.f / 64, 0.f, -45.f / 16, 0.f, 1.f, 0.f},        {0.f, -27.f / 16, -9.f / 4, 3.f / 4, 1.f, 0.f},
    {0.f, 27.f / 16, -9.f / 4, -3.f / 4, 1.f, 0.f},     {0.f, -27.f / 32, -9.f / 16, 3.f / 2, 1.f, 0.f},
    {0.f, 27.f / 32, -9.f / 16, -3.f / 2, 1.f, 0.f},    {0.f, 81.f / 64, 0.f, -45.f / 16, 0.f, 1.f}};
__device__ __constant__ float kWino4AT[4][6] = {
    {1.f, 1.f, 1.f, 1.f, 1.f, 0.f},                      {0.f, 3.f / 4, -3.f / 4, 3.f / 2, -3.f / 2, 0.f},
    {0.f, 9.f / 16, 9.f / 16, 9.f / 4, 9.f / 4, 0.f},    {0.f, 27.f / 64, -27.f / 64, 27.f / 8, -27.f / 8, 1.f}};

__device__ __forceinline__ float4 f4_fma(float a, float4 x, float4 acc) {
  return make_float4(fmaf(a, x.x, acc.x), fmaf(a, x.y, acc.y), fmaf(a, x.z, acc.z), fmaf(a, x.w, acc.w));
}
__device__ __forceinline__ float4 f4_scale(float a, float4 x) { return make_float4(a * x.x, a * x.y, a * x.z, a * x.w); }

// thread = ((tile, transformed row r), 4 channels of the slice); row r of B^T d needs the 3 or 4 input rows with a non-zero coefficient
template <int MODE>
__global__ __launch_bounds__(256) void k_wino4_in(const WinoInParams p) {
  __shared__ double red[4 * 64 * 2];
  __shared__ float s_sc[64], s_sh[64];
  const int c4n = p.Cin / 4;
  const int n = blockIdx.y;
  const int slice = blockIdx.x % p.nslices, ub = blockIdx.x / p.nslices;
  const int c4 = slice * 16 + (threadIdx.x & 15);
  const bool cok = c4 < c4n;
  const int ntiles = p.tilesY * p.tilesX;
  WinoConsts<MODE> kc;
  wino_in_consts<MODE>(p, n, slice, c4, cok, red, s_sc, s_sh, kc);
  if (!cok) return;
  for (int u = ub * 16 + (threadIdx.x >> 4); u < ntiles * 6; u += p.ublocks * 16) {
    const int r = u % 6, tile = u / 6;
    const int ty = tile / p.tilesX, tx = tile % p.tilesX;
    // rows with a non-zero coefficient: r = 0 -> {0, 2, 4}; r = 1..4 -> {1, 2, 3, 4}; r = 5 -> {1, 3, 5}
    const int a0 = r == 0 ? 0 : 1, da = (r == 0 || r == 5) ? 2 : 1, na = (r == 0 || r == 5) ? 3 : 4;
    float4 t[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) t[q] = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (MODE == WSRC_PLAIN) {
      // all 24 loads of the unit first (one memory round trip), then the arithmetic
      WinoRaw<MODE> d[4][6];
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const int iy = 4 * ty - 1 + a0 + min(k, na - 1) * da, ix = 4 * tx - 1 + q;
          wino_fetch<MODE>(p, n, min(max(iy, 0), p.H - 1), min(max(ix, 0), p.W - 1), c4, d[k][q]);
        }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int a = a0 + min(k, na - 1) * da;
        const float coef = k < na ? kWino4BT[r][a] : 0.f;
        const int iy = 4 * ty - 1 + a;
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const int ix = 4 * tx - 1 + q;
          float4 w = wino_value<MODE>(p, kc, d[k][q]);
          if (!(iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)) w = make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding after the prologue
          t[q] = f4_fma(coef, w, t[q]);
        }
      }
    } else {
      // two or three loads per element: one input row at a time (6 elements in flight) to stay within the registers
#pragma unroll 1
      for (int k = 0; k < na; ++k) {
        const int a = a0 + k * da;
        const float coef = kWino4BT[r][a];
        const int iy = 4 * ty - 1 + a;
        WinoRaw<MODE> d[6];
#pragma unroll
        for (int q = 0; q < 6; ++q) wino_fetch<MODE>(p, n, min(max(iy, 0), p.H - 1), min(max(4 * tx - 1 + q, 0), p.W - 1), c4, d[q]);
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          const int ix = 4 * tx - 1 + q;
          const bool inb = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
          float4 w = wino_value<MODE>(p, kc, d[q]);
          if constexpr (MODE == WSRC_JOIN) {    // row r = 1 reads the tile's own 4x4 pixels (rows 1..4; columns 1..4): it stores them
            if (r == 1 && q >= 1 && q <= 4 && inb) *reinterpret_cast<float4*>(p.o + (((size_t)n * p.H + iy) * p.W + ix) * p.xC + c4 * 4) = w;
          }
          if (!inb) w = make_float4(0.f, 0.f, 0.f, 0.f);
          t[q] = f4_fma(coef, w, t[q]);
        }
      }
    }
    const size_t plane = (size_t)ntiles * p.Cin;
    float* vb = p.v + (size_t)n * 36 * plane + (size_t)tile * p.Cin + c4 * 4;
    // columns: V[r][j] = sum_q B^T[j][q] t[q] (compile-time coefficients)
    const float4 e24 = f4_fma(-45.f / 16, t[2], t[4]);                 // -45/16 t2 + t4
    const float4 o13 = f4_fma(-45.f / 16, t[3], t[5]);                 // -45/16 t3 + t5
    const float4 ev1 = f4_fma(-9.f / 4, t[2], t[4]), od1 = f4_fma(-27.f / 16, t[1], f4_scale(3.f / 4, t[3]));
    const float4 ev2 = f4_fma(-9.f / 16, t[2], t[4]), od2 = f4_fma(-27.f / 32, t[1], f4_scale(3.f / 2, t[3]));
    *reinterpret_cast<float4*>(vb + (size_t)(r * 6 + 0) * plane) = f4_fma(81.f / 64, t[0], e24);
    *reinterpret_cast<float4*>(vb + (size_t)(r * 6 + 1) * plane) = RIB_F4_ADD(ev1, od1);
    *reinterpret_cast<float4*>(vb + (size_t)(r * 6 + 2) * plane) = RIB_F4_SUB(ev1, od1);
    *reinterpret_cast<float4*>(vb + (size_t)(r * 6 + 3) * plane) = RIB_F4_ADD(ev2, od2);
    *reinterpret_cast<float4*>(vb + (size_t)(r * 6 + 4) * plane) = RIB_F4_SUB(ev2, od2);
    *reinterpret_cast<float4*>(vb + (size_t)(r * 6 + 5) * plane) = f4_fma(81.f / 64, t[1], o13);
  }
}

// grid (ublocks * nslices, B); thread = ((tile, output row r of the 4x4 tile), 4 channels of the slice)
__global__ __launch_bounds__(256) void k_wino4_out(const WinoOutParams p) {
  __shared__ __attribute__((aligned(16))) double red[2][256][4];
  const int c4n = p.CoutPad / 4;
  const int slice = blockIdx.x % p.nslices, ub = blockIdx.x / p.nslices;
  const int c4 = slice * 16 + (threadIdx.x & 15);
  const int n = blockIdx.y;
  const int ntiles = p.tilesY * p.tilesX;
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  if (c4 < c4n) {
    const float4 bv = *reinterpret_cast<const float4*>(p.bias + c4 * 4);
    const size_t plane = (size_t)ntiles * p.CoutPad;
    for (int unit = ub * 16 + (threadIdx.x >> 4); unit < ntiles * 4; unit += p.ublocks * 16) {
      const int tile = unit >> 2, r = unit & 3;
      const float* mb = p.m + (size_t)n * 36 * plane + (size_t)tile * p.CoutPad + c4 * 4;
      // u[q] = sum_a A^T[r][a] M[a][q]: rows 1..4 always, row 0 for r = 0, row 5 for r = 3
      float4 u[6];
#pragma unroll
      for (int q = 0; q < 6; ++q) u[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      const int ax = r == 0 ? 0 : 5;               // the extra row (coefficient 1 for r = 0 / 3, unused otherwise)
      const float cx_ = (r == 0 || r == 3) ? 1.f : 0.f;
      float4 mm[5][6];
#pragma unroll
      for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int q = 0; q < 6; ++q) mm[k][q] = *reinterpret_cast<const float4*>(mb + (size_t)((k < 4 ? k + 1 : ax) * 6 + q) * plane);
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const float coef = k < 4 ? kWino4AT[r][k + 1] : cx_;
#pragma unroll
        for (int q = 0; q < 6; ++q) u[q] = f4_fma(coef, mm[k][q], u[q]);
      }
      // y[j] = sum_q A^T[j][q] u[q]
      const float4 s12 = RIB_F4_ADD(u[1], u[2]);
      const float4 d12 = RIB_F4_SUB(u[1], u[2]);
      const float4 s34 = RIB_F4_ADD(u[3], u[4]);
      const float4 d34 = RIB_F4_SUB(u[3], u[4]);
      float4 yv[4];
      yv[0] = make_float4(u[0].x + s12.x + s34.x, u[0].y + s12.y + s34.y, u[0].z + s12.z + s34.z, u[0].w + s12.w + s34.w);
      yv[1] = f4_fma(3.f / 4, d12, f4_scale(3.f / 2, d34));
      yv[2] = f4_fma(9.f / 16, s12, f4_scale(9.f / 4, s34));
      yv[3] = f4_fma(27.f / 64, d12, f4_fma(27.f / 8, d34, u[5]));
      const int ty = tile / p.tilesX, tx = tile % p.tilesX;
      const int oy = 4 * ty + r;
      float rr[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int cy = min(oy, p.Hout - 1), cx = min(4 * tx + q, p.Wout - 1);
#pragma unroll
        for (int e = 0; e < 4; ++e)
          rr[q][e] = p.res ? p.res[(((size_t)n * p.Hout + cy) * p.Wout + cx) * p.resC + min(c4 * 4 + e, p.resC - 1)] : 0.f;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int ox = 4 * tx + q;
        const bool inb = oy < p.Hout && ox < p.Wout;
        const float v4[4] = {yv[q].x + bv.x, yv[q].y + bv.y, yv[q].z + bv.z, yv[q].w + bv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float t = apply_act(v4[e] + rr[q][e], p.act);
          const bool ok = inb && c4 * 4 + e < p.Cout;
          if (ok) p.y[(((size_t)n * p.Hout + oy) * p.Wout + ox) * p.yC + p.yoff + c4 * 4 + e] = t;
          t = ok ? t : 0.f;
          s1[e] += (double)t; s2[e] += (double)t * (double)t;
        }
      }
    }
  }
  if (p.stat_part) wino_out_partials(red, s1, s2, p.stat_part, n, p.ublocks, ub, slice, p.CoutPad);
}
#undef RIB_F4_SUB
#undef RIB_F4_ADD

// ---------------------------------------------------------------------------------------------
// k_spade_modulate: second half of an UNFUSED SPADE (used where the map is small and the fused
// kernel cannot fill the chip).  The gamma/beta 1x1 GEMM ran as a convolution into a slab
// [S][B][HW][slab_ld] - either its own (split-K) launch or, since round 3, ONE launch per condition level that
// computes the gamma/beta of every SPADE of that level (Builder::cond_level_gemm: the filters of the level's SPADE
// groups are concatenated along N; this group's columns start at col0) - in the fused kernel's column layout
// ([gamma(32) | beta(32)] per 32 virtual channels, virtual channel v = set*C + c); this kernel sums the slices and applies
//   out_set = act_set( (x*scale + shift) * (1 + gamma) + beta )
// grid (pblocks * nslices, B): a workgroup owns a slice of 64 virtual channels (16 float4 lanes) and 16 pixels per pass,
// so that it can take (scale, shift) of its 64 channels from the producer's partials itself (block_stats_64).
// ---------------------------------------------------------------------------------------------
struct ModulateParams {
  const float* slab; int ksplit, B, slab_ld, col0;
  const float* bias;       // this group's [npad] bias vector
  const float* xm; int xmC, xm_ups;
  const float* m_scale; const float* m_shift; int m_ld;
  StatSrc st;              // replaces m_scale / m_shift when st.part != nullptr (needs C % 64 == 0: a slice stays inside one set)
  int C, nsets;
  float* ys0; float* ys1; int act0, act1;
  int Hout, Wout;
  int nslices, pblocks;
};

template <int ST>
__global__ __launch_bounds__(256) void k_spade_modulate(const ModulateParams p) {
  __shared__ double red[4 * 64 * 2];
  __shared__ float s_sc[64], s_sh[64];
  const int v4n = p.nsets * p.C / 4;
  const int n = blockIdx.y;
  const int slice = blockIdx.x % p.nslices, pb = blockIdx.x / p.nslices;
  const int v4 = slice * 16 + (threadIdx.x & 15);
  const int npix = p.Hout * p.Wout;
  const size_t sstride = (size_t)p.B * npix * p.slab_ld;
  const int Hm = p.xm_ups ? p.Hout / 2 : p.Hout, Wm = p.xm_ups ? p.Wout / 2 : p.Wout;
  const int v = v4 * 4;
  const int set = v >= p.C ? 1 : 0;
  const int c = v - set * p.C;
  float4 sc, sh;
  if (p.st.part) {
    const int v0 = slice * 64, c0 = v0 >= p.C ? v0 - p.C : v0;
    block_stats_64(p.st, n, c0, min(64, p.C - c0), red, s_sc, s_sh);
    const int l = (threadIdx.x & 15) * 4;
    sc = make_float4(s_sc[l], s_sc[l + 1], s_sc[l + 2], s_sc[l + 3]);
    sh = make_float4(s_sh[l], s_sh[l + 1], s_sh[l + 2], s_sh[l + 3]);
  }
  if (v4 >= v4n) return;
  if (!p.st.part) {
    sc = *reinterpret_cast<const float4*>(p.m_scale + (size_t)n * p.m_ld + c);
    sh = *reinterpret_cast<const float4*>(p.m_shift + (size_t)n * p.m_ld + c);
  }
  const int colg = (v / 32) * 64 + (v % 32);
  const float4 bg = *reinterpret_cast<const float4*>(p.bias + colg);
  const float4 bb = *reinterpret_cast<const float4*>(p.bias + colg + 32);
  const int act = set ? p.act1 : p.act0;
  float* yout = set ? p.ys1 : p.ys0;
  // four pixels per iteration: all loads (clamped addresses) first, then the arithmetic and the stores - a load behind a
  // store of the previous pixel would wait for that store ("reading the ISA")
  const int pstep = p.pblocks * 16;
  for (int pix0 = pb * 16 + (threadIdx.x >> 4); pix0 < npix; pix0 += 4 * pstep) {
    float4 g[4], b[4], x[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pix = min(pix0 + k * pstep, npix - 1);
      const float* src = p.slab + ((size_t)n * npix + pix) * p.slab_ld + p.col0 + colg;
      g[k] = *reinterpret_cast<const float4*>(src);
      b[k] = *reinterpret_cast<const float4*>(src + 32);
      for (int s = 1; s < p.ksplit; ++s) {
        const float4 g2 = *reinterpret_cast<const float4*>(src + s * sstride);
        const float4 b2 = *reinterpret_cast<const float4*>(src + s * sstride + 32);
        g[k].x += g2.x; g[k].y += g2.y; g[k].z += g2.z; g[k].w += g2.w;
        b[k].x += b2.x; b[k].y += b2.y; b[k].z += b2.z; b[k].w += b2.w;
      }
      const int oy = pix / p.Wout, ox = pix % p.Wout;
      const int sy = p.xm_ups ? (oy >> 1) : oy, sx = p.xm_ups ? (ox >> 1) : ox;
      x[k] = ld_act4<ST>(p.xm, (((size_t)n * Hm + sy) * Wm + sx) * p.xmC + c);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pix = pix0 + k * pstep;
      if (pix >= npix) break;
      float4 o;
      o.x = apply_act(spade_mod1(x[k].x, sc.x, sh.x, g[k].x, bg.x, b[k].x, bb.x), act);
      o.y = apply_act(spade_mod1(x[k].y, sc.y, sh.y, g[k].y, bg.y, b[k].y, bb.y), act);
      o.z = apply_act(spade_mod1(x[k].z, sc.z, sh.z, g[k].z, bg.z, b[k].z, bb.z), act);
      o.w = apply_act(spade_mod1(x[k].w, sc.w, sh.w, g[k].w, bg.w, b[k].w, bb.w), act);
      st_act4<ST>(yout, ((size_t)n * npix + pix) * p.C + c, o);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_avgpool: AvgPool2d(3, stride 2, pad 1), divisor always 9, NHWC; also emits the per-block
// partial sums of its OUTPUT (the next block's SPADE normalises the pooled tensor).
// grid (blocks, B); each block covers PPB = 256/(C/4) * 4 output pixels, thread = (pixel slot, c4).
// ---------------------------------------------------------------------------------------------
struct PoolParams {
  const float* x; float* y;
  int H, W, C;          // input size; output H/2 x W/2
  double* stat_part;    // [B][blocks][2][C]
  int blocks;
};

template <int ST>
__global__ __launch_bounds__(256) void k_avgpool(const PoolParams p) {
  __shared__ __attribute__((aligned(16))) double red[2][256][4];
  const int c4n = p.C / 4;
  const int slots = 256 / c4n;
  const int c4 = threadIdx.x % c4n, slot = threadIdx.x / c4n;
  const int Ho = p.H / 2, Wo = p.W / 2;
  const int n = blockIdx.y;
  const int npix = Ho * Wo;
  const int ppb = slots * 4;
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  const size_t xn = (size_t)n * p.H * p.W * p.C;
  for (int k = 0; k < 4; ++k) {
    const int pix = blockIdx.x * ppb + k * slots + slot;
    if (pix < npix && slot < slots) {
      const int oy = pix / Wo, ox = pix % Wo;
      // the nine taps are loaded unconditionally from clamped coordinates (one batch of independent loads)
      // and masked when added: same sum, same order as skipping the out-of-range taps
      float4 v[9]; float m[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = oy * 2 - 1 + t / 3, ix = ox * 2 - 1 + t % 3;
        m[t] = (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) ? 1.f : 0.f;
        const int cy = min(max(iy, 0), p.H - 1), cx = min(max(ix, 0), p.W - 1);
        v[t] = ld_act4<ST>(p.x, xn + ((size_t)cy * p.W + cx) * p.C + c4 * 4);
      }
      float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int t = 0; t < 9; ++t)
        if (m[t] != 0.f) { a.x += v[t].x; a.y += v[t].y; a.z += v[t].z; a.w += v[t].w; }
      const float inv9 = 1.f / 9.f;
      a.x *= inv9; a.y *= inv9; a.z *= inv9; a.w *= inv9;
      if constexpr (ST != ST_F32) a = make_float4(round16<ST>(a.x), round16<ST>(a.y), round16<ST>(a.z), round16<ST>(a.w));
      st_act4<ST>(p.y, ((size_t)n * npix + pix) * p.C + c4 * 4, a);
      const float av[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { s1[e] += (double)av[e]; s2[e] += (double)av[e] * (double)av[e]; }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[0][threadIdx.x][e] = s1[e]; red[1][threadIdx.x][e] = s2[e]; }
  __syncthreads();
  // thread t < C sums channel t over the pixel slots (fixed order)
  for (int c = threadIdx.x; c < p.C; c += 256) {
    const int g = c / 4, e = c % 4;
    double a1 = 0.0, a2 = 0.0;
    for (int s = 0; s < slots; ++s) { a1 += red[0][s * c4n + g][e]; a2 += red[1][s * c4n + g][e]; }
    double* dst = p.stat_part + (((size_t)n * p.blocks + blockIdx.x) * 2) * p.C;
    dst[c] = a1;
    dst[p.C + c] = a2;
  }
}

// ---------------------------------------------------------------------------------------------
// k_in_add: mask-network residual join (Res2dBlock 'CNACN', PGNR/models/generator.py:465-476):
//   out = IN_affine(t1) + (ts ? IN_affine(ts) : xres)      float4 over [B][HW][C]
// grid (pblocks * nslices, B): channel slices of 64 as k_spade_modulate, for the same reason (st1 / sts: the partials
// of the two producers instead of their (scale, shift) arrays).
// ---------------------------------------------------------------------------------------------
struct InAddParams {
  const float* t1; const float* sc1; const float* sh1;
  const float* ts; const float* scs; const float* shs;   // ts may be nullptr
  const float* xres;                                      // used when ts == nullptr
  float* out;
  int C, HW;
  int ld;   // leading dim of the scale/shift arrays
  StatSrc st1, sts;
  int nslices, pblocks;
};

template <int ST>
__global__ __launch_bounds__(256) void k_in_add(const InAddParams p) {
  __shared__ double red[4 * 64 * 2];
  __shared__ float s_sc[2][64], s_sh[2][64];
  const int c4n = p.C / 4;
  const int n = blockIdx.y;
  const int slice = blockIdx.x % p.nslices, pb = blockIdx.x / p.nslices;
  const int c4 = slice * 16 + (threadIdx.x & 15);
  const int nch = min(64, p.C - slice * 64);
  if (p.st1.part) block_stats_64(p.st1, n, slice * 64, nch, red, s_sc[0], s_sh[0]);
  if (p.ts && p.sts.part) block_stats_64(p.sts, n, slice * 64, nch, red, s_sc[1], s_sh[1]);
  if (c4 >= c4n) return;
  const int l = (threadIdx.x & 15) * 4;
  float4 s, t, s2 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = s2;
  if (p.st1.part) { s = make_float4(s_sc[0][l], s_sc[0][l + 1], s_sc[0][l + 2], s_sc[0][l + 3]); t = make_float4(s_sh[0][l], s_sh[0][l + 1], s_sh[0][l + 2], s_sh[0][l + 3]); }
  else { s = *reinterpret_cast<const float4*>(p.sc1 + (size_t)n * p.ld + c4 * 4); t = *reinterpret_cast<const float4*>(p.sh1 + (size_t)n * p.ld + c4 * 4); }
  if (p.ts) {
    if (p.sts.part) { s2 = make_float4(s_sc[1][l], s_sc[1][l + 1], s_sc[1][l + 2], s_sc[1][l + 3]); t2 = make_float4(s_sh[1][l], s_sh[1][l + 1], s_sh[1][l + 2], s_sh[1][l + 3]); }
    else { s2 = *reinterpret_cast<const float4*>(p.scs + (size_t)n * p.ld + c4 * 4); t2 = *reinterpret_cast<const float4*>(p.shs + (size_t)n * p.ld + c4 * 4); }
  }
  const float* second = p.ts ? p.ts : p.xres;
  if (!p.ts) { s2 = make_float4(1.f, 1.f, 1.f, 1.f); t2 = make_float4(0.f, 0.f, 0.f, 0.f); }     // + xres
  const int pstep = p.pblocks * 16;
  for (int pix0 = pb * 16 + (threadIdx.x >> 4); pix0 < p.HW; pix0 += 4 * pstep) {   // four pixels per iteration, loads first
    float4 a[4], b[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t e = ((size_t)n * p.HW + min(pix0 + k * pstep, p.HW - 1)) * p.C + c4 * 4;
      a[k] = ld_act4<ST>(p.t1, e);
      b[k] = ld_act4<ST>(second, e);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int pix = pix0 + k * pstep;
      if (pix >= p.HW) break;
      float4 o = make_float4(a[k].x * s.x + t.x, a[k].y * s.y + t.y, a[k].z * s.z + t.z, a[k].w * s.w + t.w);
      if (p.ts) { o.x += b[k].x * s2.x + t2.x; o.y += b[k].y * s2.y + t2.y; o.z += b[k].z * s2.z + t2.z; o.w += b[k].w * s2.w + t2.w; }
      else { o.x += b[k].x; o.y += b[k].y; o.z += b[k].z; o.w += b[k].w; }
      st_act4<ST>(p.out, ((size_t)n * p.HW + pix) * p.C + c4 * 4, o);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_conv_small: 3x3 stride-1 convolution with 1..4 output channels (the RGB head conv_img 16 -> 3 and
// the mask head conv_mask.0 32 -> 1, PGNR/models/generator.py:114-116,482-485), on the vector ALUs.
// On the matrix cores these layers pad N to 16 columns for 1 or 3 real ones (25.5 / 32.6 us at 512x512,
// 5 .. 9 TFLOP/s "achieved"); they are really HBM-bound reads of a 16 / 32-channel map.  A workgroup
// stages the 18 x 18 halo of a 16 x 16 pixel tile (all input channels, prologue applied once per
// element) and the CO x 9 x Cin filter in LDS; thread = pixel; filter reads are LDS broadcasts.
// Uses IgemmParams (x, prologue, w [CoutPad][9][Cin], bias, y / y_nchw, act); grid (tiles, 1, B).
// ---------------------------------------------------------------------------------------------
template <int CO, int ST = 0>
__global__ __launch_bounds__(256) void k_conv_small(const IgemmParams p) {
  extern __shared__ __attribute__((aligned(16))) float smem_dyn[];
  const int Cin = p.Cin, CK = Cin + 4, C4 = Cin / 4;
  float* sA = smem_dyn;                       // [18 * 18][CK]
  float* sW = smem_dyn + 18 * 18 * CK;        // [CO][9][Cin]
  const int tid = threadIdx.x;
  const int n = blockIdx.z;
  const int tile = p.xcd_chunk ? (int)(blockIdx.x & 7) * p.xcd_chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int ty0 = (tile / p.tilesX) * 16, tx0 = (tile % p.tilesX) * 16;
  const size_t xn = (size_t)n * p.Hin * p.Win * p.xC;
  for (int i = tid; i < CO * 9 * C4; i += 256)
    *reinterpret_cast<float4*>(sW + i * 4) = *reinterpret_cast<const float4*>(p.w + i * 4);   // rows 0..CO-1 are contiguous
  const int total4 = 18 * 18 * C4;
  // a thread keeps one channel group across its staging slots (256 % C4 == 0): prologue constants loaded once
  const int c4 = tid % C4;
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.pro_scale) {
    sc = *reinterpret_cast<const float4*>(p.pro_scale + (size_t)n * p.pro_ld + c4 * 4);
    sh = *reinterpret_cast<const float4*>(p.pro_shift + (size_t)n * p.pro_ld + c4 * 4);
  }
  float bias[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) bias[co] = p.bias[co];
  for (int base = 0; base < total4; base += 256 * 8) {
    // one batch of UNCONDITIONAL loads from clamped coordinates (a per-element `if (in range) load` is control flow,
    // after which the compiler waits for vmcnt(0) per element: eight serialised round trips), then the prologue
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = min(base + u * 256 + tid, total4 - 1);
      const int pix = idx / C4;
      const int iy = min(max(ty0 - 1 + pix / 18, 0), p.Hin - 1), ix = min(max(tx0 - 1 + pix % 18, 0), p.Win - 1);
      v[u] = ld_act4<ST>(p.x, xn + (unsigned)((iy * p.Win + ix) * p.xC + c4 * 4));
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * 256 + tid;
      const int pix = idx / C4;
      const int iy = ty0 - 1 + pix / 18, ix = tx0 - 1 + pix % 18;
      float4 t = v[u];
      if (p.pro_scale) t = make_float4(t.x * sc.x + sh.x, t.y * sc.y + sh.y, t.z * sc.z + sh.z, t.w * sc.w + sh.w);
      if (p.pro_lrelu) t = lrelu4(t);
      if (!(iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win)) t = make_float4(0.f, 0.f, 0.f, 0.f);   // zero padding after the prologue
      if (idx < total4) *reinterpret_cast<float4*>(sA + pix * CK + c4 * 4) = t;
    }
  }
  __syncthreads();
  const int ly = tid >> 4, lx = tid & 15;
  float acc[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) acc[co] = 0.f;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const float* a = sA + ((ly + tap / 3) * 18 + lx + tap % 3) * CK;
    const float* w = sW + tap * Cin;
    for (int c = 0; c < Cin; c += 4) {
      const float4 av = *reinterpret_cast<const float4*>(a + c);
#pragma unroll
      for (int co = 0; co < CO; ++co) {
        const float4 wv = *reinterpret_cast<const float4*>(w + co * 9 * Cin + c);
        acc[co] = fmaf(av.x, wv.x, acc[co]); acc[co] = fmaf(av.y, wv.y, acc[co]);
        acc[co] = fmaf(av.z, wv.z, acc[co]); acc[co] = fmaf(av.w, wv.w, acc[co]);
      }
    }
  }
  const int oy = ty0 + ly, ox = tx0 + lx;
  if (oy >= p.Hout || ox >= p.Wout) return;
  const size_t pix = ((size_t)n * p.Hout + oy) * p.Wout + ox;
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    const float v = apply_act(acc[co] + bias[co], p.act);
    if (p.y) { if (ST != ST_F32 && !p.y_f32) st_act<ST>(p.y, pix * p.yC + p.yoff + co, v); else p.y[pix * p.yC + p.yoff + co] = v; }
    if (p.y_nchw) p.y_nchw[(((size_t)n * p.Cout + co) * p.Hout + oy) * p.Wout + ox] = v;
  }
  for (int co = CO; co < p.Cout && p.y; ++co) {   // channel padding, as k_igemm stores it
    if (ST != ST_F32 && !p.y_f32) st_act<ST>(p.y, pix * p.yC + p.yoff + co, apply_act(0.f, p.act)); else p.y[pix * p.yC + p.yoff + co] = apply_act(0.f, p.act);
  }
}

// ---------------------------------------------------------------------------------------------
// k_conv_head: the two heads again (conv_img 16 -> 3 + tanh, conv_mask.0 32 -> 1 + sigmoid), on the matrix cores with the
// TAPS as GEMM columns.  k_conv_small above is LDS-read-bound (nine shifted reads of every staged input value); here
// every input pixel is read ONCE, straight from global memory in MFMA operand layout, and multiplied by all
// 9 x CO (tap, output channel) filter columns at once (v_mfma_f32_16x16x4_f32, exact fp32):
//     P[pixel][tap*CO + co] = sum_c x[pixel][c] * w[co][tap][c]        for the 18 x 18 halo of a 16 x 16 output tile
//     out[y][x][co]         = bias[co] + sum_tap P[(y + dy, x + dx)][tap*CO + co]
// P goes through LDS (one plane per column); the nine-term sum is the only LDS traffic.  The filters live in registers
// (8 values per lane).  Zero padding falls out of zeroing the operand of out-of-image halo pixels (after the prologue).
// For the mask head the driver's blend is fused: the thread that has the pixel's mask also writes the fused frame.
// CIN = padded input channels (16 or 32); grid (tiles, 1, B), block 256; IgemmParams as for k_conv_small.
// ---------------------------------------------------------------------------------------------
template <int CO, int CIN, int ST = 0>
__global__ __launch_bounds__(256) void k_conv_head(const IgemmParams p) {
  constexpr int NCOL = 9 * CO, NB = (NCOL + 15) / 16, NG = CIN / 16;
  constexpr int HW_ = 18, HPX = HW_ * HW_, NBLK = (HPX + 15) / 16, PP = HPX + 1;
  __shared__ float sP[NCOL * PP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, q = lane >> 4;
  const int n = blockIdx.z;
  const int tile = p.xcd_chunk ? (int)(blockIdx.x & 7) * p.xcd_chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int ty0 = (tile / p.tilesX) * 16, tx0 = (tile % p.tilesX) * 16;
  // B operand: lane (k group q, column l15) holds w[co][tap][16 g + 4 q + t] of its column for every (g, t) MFMA step
  float bw[NG][4][NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    const int col = nb * 16 + l15;
    const int tap = col / CO, co = col % CO;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float4 w4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (col < NCOL) w4 = *reinterpret_cast<const float4*>(p.w + ((size_t)co * 9 + tap) * CIN + g * 16 + q * 4);
      bw[g][0][nb] = w4.x; bw[g][1][nb] = w4.y; bw[g][2][nb] = w4.z; bw[g][3][nb] = w4.w;
    }
  }
  float4 psc[NG], psh[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    psc[g] = make_float4(1.f, 1.f, 1.f, 1.f); psh[g] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.pro_scale) {
      psc[g] = *reinterpret_cast<const float4*>(p.pro_scale + (size_t)n * p.pro_ld + g * 16 + q * 4);
      psh[g] = *reinterpret_cast<const float4*>(p.pro_shift + (size_t)n * p.pro_ld + g * 16 + q * 4);
    }
  }
  const size_t xn = (size_t)n * p.Hin * p.Win * p.xC;
  // every wave owns NBW 16-pixel blocks of the halo; ALL their operand loads are issued before the first use (clamped,
  // always valid addresses): one memory round trip per workgroup instead of one per block
  constexpr int NBW = (NBLK + 3) / 4;
  float4 areg[NBW][NG];
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    const int px = min((wave + 4 * i) * 16 + l15, HPX - 1);
    const int cy = min(max(ty0 - 1 + px / HW_, 0), p.Hin - 1), cx = min(max(tx0 - 1 + px % HW_, 0), p.Win - 1);
#pragma unroll
    for (int g = 0; g < NG; ++g) areg[i][g] = ld_act4<ST>(p.x, xn + (size_t)(unsigned)((cy * p.Win + cx) * p.xC + g * 16 + q * 4));
  }
#pragma unroll
  for (int i = 0; i < NBW; ++i) {
    const int blk = wave + 4 * i;
    if (blk >= NBLK) break;
    const int px = blk * 16 + l15;                       // halo pixel of this lane's A row
    const int iy = ty0 - 1 + px / HW_, ix = tx0 - 1 + px % HW_;
    const bool inb = px < HPX && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
    float4 a[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      float4 v = areg[i][g];
      if (p.pro_scale) v = make_float4(v.x * psc[g].x + psh[g].x, v.y * psc[g].y + psh[g].y, v.z * psc[g].z + psh[g].z, v.w * psc[g].w + psh[g].w);
      if (p.pro_lrelu) v = lrelu4(v);
      if (!inb) v = make_float4(0.f, 0.f, 0.f, 0.f);     // zero padding after the prologue
      a[g] = v;
    }
    f32x4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const float av = t == 0 ? a[g].x : t == 1 ? a[g].y : t == 2 ? a[g].z : a[g].w;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw[g][t][nb], acc[nb], 0, 0, 0);
      }
    // accumulator element r of lane (q, l15): pixel blk*16 + 4 q + r, column nb*16 + l15
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const int col = nb * 16 + l15;
      if (col < NCOL) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int ppx = blk * 16 + q * 4 + r;
          if (ppx < HPX) sP[col * PP + ppx] = acc[nb][r];
        }
      }
    }
  }
  __syncthreads();
  const int ly = tid >> 4, lx = tid & 15;
  const int oy = ty0 + ly, ox = tx0 + lx;
  if (oy >= p.Hout || ox >= p.Wout) return;
  float o[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) o[co] = 0.f;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int hp = (ly + tap / 3) * HW_ + lx + tap % 3;
#pragma unroll
    for (int co = 0; co < CO; ++co) o[co] += sP[(tap * CO + co) * PP + hp];
  }
  const size_t pix = ((size_t)n * p.Hout + oy) * p.Wout + ox;
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    const float v = apply_act(o[co] + p.bias[co], p.act);
    o[co] = v;
    if (p.y) { if (ST != ST_F32 && !p.y_f32) st_act<ST>(p.y, pix * p.yC + p.yoff + co, v); else p.y[pix * p.yC + p.yoff + co] = v; }
    if (p.y_nchw) p.y_nchw[(((size_t)n * p.Cout + co) * p.Hout + oy) * p.Wout + ox] = v;
  }
  for (int co = CO; co < p.Cout && p.y; ++co) {   // channel padding, as k_igemm stores it
    if (ST != ST_F32 && !p.y_f32) st_act<ST>(p.y, pix * p.yC + p.yoff + co, apply_act(0.f, p.act)); else p.y[pix * p.yC + p.yoff + co] = apply_act(0.f, p.act);
  }
  if constexpr (CO == 1) {
    if (p.bl_fuse) {                        // evaluator.py:256-258, same operation order as k_blend
      const float m = o[0];
      const size_t hw = (size_t)p.Hout * p.Wout, at = (size_t)oy * p.Wout + ox;
      for (int c = 0; c < p.bl_C; ++c) {
        const size_t i = ((size_t)n * p.bl_C + c) * hw + at;
        p.bl_fuse[i] = blend1(p.bl_img[i], m, p.bl_dain[i]);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_gather6: up to six device-to-device copies in one launch (rib_chain: a frame's slices of the batched
// label-only results into the frame plan's slots).  Sizes are multiples of 16 bytes; grid (blocks, 6).
// ---------------------------------------------------------------------------------------------
struct Gather6Params { const float4* src[6]; float4* dst[6]; unsigned n4[6]; };

__global__ __launch_bounds__(256) void k_gather6(const Gather6Params p) {
  const int r = blockIdx.y;
  const float4* s = p.src[r];
  float4* d = p.dst[r];
  const unsigned n = p.n4[r];
  for (unsigned i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) d[i] = s[i];
}

// ---------------------------------------------------------------------------------------------
// k_conv_lowc: the four 3x3 stride-1 convolutions that read the caller's tensors (Cin = 6, 9, 22, 22 at full resolution:
// ref_embedding.conv_first, flow_network_temp.down_img.0 / down_lbl.0, down_first), round 2 (VERDICT #4b).
// k_igemm runs them as nine taps x a channel chunk padded to 8 / 16 / 32 after a k_pack launch made the zero-padded NHWC
// copy.  Here the reduction runs over the REAL channels, k = tap * CE + c with CE = Cin rounded up to 2 (4 on the
// 16-column path), and the input is read where the caller left it:
//   * the halo tile (10 x 34 pixels x CE channels) is gathered straight from up to three NCHW fp32 tensors (coalesced
//     along x; torch.cat at PGNR/models/generator.py:197,232 happens in this gather) into LDS at an odd channel pitch;
//   * the filter fragments [K/2][2][NCOL] (K-pair-major: a fragment is 32 consecutive words) live in REGISTERS, K/2 x NF
//     of them per lane, loaded once from global memory: fp32 32x32x2 MFMAs consume 512 operand bytes per 16 issue
//     cycles, i.e. the LDS's whole 128 B/clk per CU when both operands come from it (the first version did that and ran
//     at 45 TFLOP/s, slower than k_igemm); with B in registers LDS delivers one A word per lane and MFMA;
//   * ONE barrier per workgroup, then K/2 steps of {one 4-byte LDS read per M fragment at a compile-time offset from the
//     lane's pixel, MFMAs}: consecutive k of a pair are consecutive channels of the same tap, so lane half lh adds
//     4 bytes and everything else is an immediate;
//   * epilogue as k_igemm's: bias, activation, NHWC store in the storage type, fp64 statistics partials per tile.
// Workgroup = 8 x 32 (or 8 x 16) output pixels, 4 waves x 2 rows; NCOL = 32 / 64: v_mfma_f32_32x32x2_f32 (fragment = 32
// pixels of a row, or 2 rows x 16); NCOL = 16: v_mfma_f32_16x16x4_f32 (fragment = 16 pixels).  fp32 arithmetic (the inputs
// are the caller's fp32 tensors); used by the fp32-storage modes.  grid (tilesX * tilesY, B).
// ---------------------------------------------------------------------------------------------
struct LowcParams {
  const float* s0; const float* s1; const float* s2; int c0, c1, c2;   // NCHW sources [B][ci][H][W], concatenated along channels
  int H, W;
  const float* w;        // [K/2][2][NCOL] (NCOL >= 32) or [K/4][4][16]
  const float* bias;     // [NCOL]
  float* y; int yC, yoff, Cout;     // NHWC destination (storage type), first Cout columns stored
  int act;
  double* stat_part; int CoutPad;   // [B][tiles][2][CoutPad] or nullptr
  int tilesX, tilesY;
};

template <int CE, int NCOL, int ST, int TW = 32>
__global__ __launch_bounds__(256) void k_conv_lowc(const LowcParams p) {
  constexpr bool N16 = NCOL == 16;
  static_assert(NCOL == 16 || NCOL == 32 || NCOL == 64, "16-, 32- or 64-column layers");
  static_assert(CE % (N16 ? 4 : 2) == 0, "channel count rounded up to the k-group of one MFMA");
  static_assert(TW == 32 || TW == 16, "8x32 or 8x16 pixel tiles");
  constexpr int TH = 8, IH = TH + 2, IW = TW + 2;
  constexpr int MF = TW / 16;                   // 32-pixel fragments per wave: one row each (TW = 32) or one fragment of 2 rows x 16
  constexpr int CP = CE + 1;                     // odd LDS pitch: the 32 (16) pixels of a fragment read 32 (16) different banks
  constexpr int KG = N16 ? 4 : 2;                // k per MFMA
  constexpr int S = 9 * CE / KG;                 // MFMA steps
  constexpr int NF = N16 ? 1 : NCOL / 32;
  __shared__ __attribute__((aligned(16))) float sA[IH * IW * CP];
  __shared__ __attribute__((aligned(16))) double red[4][NCOL][2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.y;
  const int tile = blockIdx.x;
  const int ty0 = (tile / p.tilesX) * TH, tx0 = (tile % p.tilesX) * TW;
  const int ctot = p.c0 + p.c1 + p.c2;
  const size_t HW = (size_t)p.H * p.W;
  // ---- stage the halo tile: all of a thread's loads are issued before the first LDS store (one memory round trip) ----
  {
    constexpr int TOT = CE * IH * IW, NIT = (TOT + 255) / 256;
    float v[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 256;
      const int c = i / (IH * IW), rem = i - c * (IH * IW);
      const int y = rem / IW, x = rem - y * IW;
      const int gy = ty0 - 1 + y, gx = tx0 - 1 + x;
      v[it] = 0.f;
      if (i < TOT && c < ctot && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W) {
        const float* src; int cc, cn;
        if (c < p.c0) { src = p.s0; cc = c; cn = p.c0; }
        else if (c < p.c0 + p.c1) { src = p.s1; cc = c - p.c0; cn = p.c1; }
        else { src = p.s2; cc = c - p.c0 - p.c1; cn = p.c2; }
        v[it] = src[((size_t)n * cn + cc) * HW + (size_t)gy * p.W + gx];
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + it * 256;
      const int c = i / (IH * IW), rem = i - c * (IH * IW);
      const int y = rem / IW, x = rem - y * IW;
      if (i < TOT) sA[(y * IW + x) * CP + c] = v[it];
    }
  }
  // ---- this lane's filter fragments (registers; after the staging so that its 30 values in flight are dead: 116 instead
  // of 198 VGPRs on the 22-channel layers) ----
  // More than 64 of them (the 22-channel layers: 99) are loaded in two halves: 99 + 32 accumulators is 3 waves per SIMD,
  // i.e. 768 workgroup slots for the 1024 tiles of a 512x512 frame - a second round a third full; ~50 + 32 fits 4.
  constexpr int HV = (!N16 && S * NF > 64) ? 2 : 1;
  constexpr int SH = (S + HV - 1) / HV;
  float bw[SH][NF];
  const float* pw = N16 ? p.w + (lane >> 4) * 16 + (lane & 15) : p.w + (lane >> 5) * NCOL + (lane & 31);
  auto load_b = [&](int h) {
#pragma unroll
    for (int s = 0; s < SH; ++s)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
        if (h * SH + s < S) bw[s][nf] = pw[(h * SH + s) * KG * NCOL + nf * 32];
  };
  load_b(0);
  __syncthreads();
  double s1 = 0.0, s2 = 0.0;                      // this lane's column: sum and sum of squares over its valid pixels
  if constexpr (!N16) {
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc[MF][NF];
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NF; ++nf)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;
    // window origin of this lane's pixel: (row 2*wave [+ mf], x = li) or, 16 wide, (row 2*wave + li/16, x = li%16); k parity lh
    const float* pa = sA + (TW == 32 ? ((wave * 2) * IW + li) : ((wave * 2 + (li >> 4)) * IW + (li & 15))) * CP + lh;
#pragma unroll
    for (int h = 0; h < HV; ++h) {
      if (h > 0) load_b(h);
#pragma unroll
      for (int sl = 0; sl < SH; ++sl) {
        const int s = h * SH + sl;
        if (s < S) {
          const int tap = (2 * s) / CE, c = (2 * s) % CE;
          const int off = ((tap / 3) * IW + (tap % 3)) * CP + c;
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) {
            const float a = pa[off + mf * IW * CP];
#pragma unroll
            for (int nf = 0; nf < NF; ++nf) {
              acc[mf][nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bw[sl][nf], acc[mf][nf], 0, 0, 0);
            }
          }
        }
      }
    }
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
    const float* pa = sA + ((wave * 2) * IW + l15) * CP + lq;
#pragma unroll
    for (int s = 0; s < S; ++s) {
      const int tap = (4 * s) / CE, c = (4 * s) % CE;
      const int off = ((tap / 3) * IW + (tap % 3)) * CP + c;
#pragma unroll
      for (int f = 0; f < NFR; ++f) {
        const float a = pa[off + (TW == 32 ? ((f >> 1) * IW + (f & 1) * 16) : f * IW) * CP];
        acc[f] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, bw[s][0], acc[f], 0, 0, 0);
      }
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
      {
        double* dst = p.stat_part + (((size_t)n * (p.tilesX * p.tilesY) + tile) * 2) * p.CoutPad;
        dst[c] = a1;
        dst[p.CoutPad + c] = a2;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_pack: concatenate up to 3 NCHW sources along channels into one zero-padded NHWC tensor
// (torch.cat at PGNR/models/generator.py:197,232 and the NCHW->NHWC boundary conversion).
// 64 pixels per 256-thread block; grid (ceil(HW/64), B); dC <= 32.
// ---------------------------------------------------------------------------------------------
struct PackParams {
  const float* s0; const float* s1; const float* s2;
  int c0, c1, c2;
  float* dst; int dC;   // dC multiple of 4
  int HW;
};

template <int ST>
__global__ __launch_bounds__(256) void k_pack(const PackParams p) {
  // 64 pixels per block.  Phase 1: thread = (pixel, channel slice) reads NCHW coalesced along the
  // pixels; phase 2, after an LDS transpose: consecutive lanes write consecutive 16 bytes of NHWC.
  __shared__ float tile[64][33];
  const int n = blockIdx.y;
  const int pix0 = blockIdx.x * 64;
  const int lp = threadIdx.x & 63, slice = threadIdx.x >> 6;        // 4 channel slices
  const int ctot = p.c0 + p.c1 + p.c2;
  const int pix = pix0 + lp;
  for (int j = slice; j < p.dC; j += 4) {
    float t = 0.f;
    if (pix < p.HW && j < ctot) {
      if (j < p.c0) t = p.s0[((size_t)n * p.c0 + j) * p.HW + pix];
      else if (j < p.c0 + p.c1) t = p.s1[((size_t)n * p.c1 + (j - p.c0)) * p.HW + pix];
      else t = p.s2[((size_t)n * p.c2 + (j - p.c0 - p.c1)) * p.HW + pix];
    }
    tile[lp][j] = t;
  }
  __syncthreads();
  const int c4n = p.dC / 4;
  for (int idx = threadIdx.x; idx < 64 * c4n; idx += 256) {
    const int q = idx / c4n, g = idx % c4n;
    if (pix0 + q < p.HW)
      st_act4<ST>(p.dst, ((size_t)n * p.HW + pix0 + q) * p.dC + g * 4,
                    make_float4(tile[q][g * 4], tile[q][g * 4 + 1], tile[q][g * 4 + 2], tile[q][g * 4 + 3]));
  }
}

// NHWC (channel stride sC, first C channels) -> NCHW, for taps / debugging
template <int ST>
__global__ __launch_bounds__(256) void k_unpack(const float* src, int sC, int C, int HW, int ups, int H, int W, float* dst) {
  const int pix = blockIdx.x * 256 + threadIdx.x;
  const int n = blockIdx.y;
  if (pix >= HW) return;
  (void)ups; (void)H; (void)W;
  for (int c = 0; c < C; ++c) dst[((size_t)n * C + c) * HW + pix] = ld_act<ST>(src, ((size_t)n * HW + pix) * sC + c);
}

// ---------------------------------------------------------------------------------------------
// driver-side elementwise kernels (NCHW, as the reference driver holds its tensors)
// ---------------------------------------------------------------------------------------------
// fuse = img*mask + dain*(1-mask), mask broadcast over C (PGNR/models/evaluator.py:256-258)
__global__ __launch_bounds__(256) void k_blend(const float* img, const float* mask, const float* dain,
                                               float* fuse, int C, int HW, size_t total) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t pix = i % HW;
    const size_t n = i / ((size_t)C * HW);
    const float m = mask[n * HW + pix];
    fuse[i] = blend1(img[i], m, dain[i]);
  }
}

// uint8 HWC = uint8(clip(x*0.5+0.5, 0, 1)*255)  (truncation; PGNR/utils/utils.py:129-142; the
// reference evaluates this in float64, so do we)
__global__ __launch_bounds__(256) void k_quantise(const float* img, uint8_t* out, int C, int HW, size_t total) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t c = i % C;
    const size_t pix = (i / C) % HW;
    const size_t n = i / ((size_t)C * HW);
    double v = (double)img[(n * C + c) * HW + pix] * 0.5 + 0.5;
    v = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
    out[i] = (uint8_t)(v * 255.0);
  }
}

// Bilinear flow warp == grid_sample(img, base + flow*2/(size-1), bilinear, border, align_corners=True)
// i.e. sample img at (x + fx, y + fy) in pixel units with border clamping.  Extension op (the north star
// names it, the reference has no such call), pinned to torch's grid_sample.
// Round 5 (the round-4 kernel: 16 x 16 tiles under a 32 x 32 window = 4x the source bytes through the caches, 2.3x the
// algorithmic HBM traffic, 0.22-0.27 of the HBM roof): a workgroup owns a 64 x 32 tile of OUTPUT pixels and stages the
// (64 + 2R) x (32 + 2R) window of the source around it (R = WARP_R pixels of flow reach, all channels, border-clamped
// coordinates: 1.9x the tile instead of 4x) in LDS as 16-byte row segments (the frames are NCHW, x-contiguous planes; rows
// and the window origin are 16-byte aligned when W % 4 == 0, else element by element); workgroups take tiles in an
// XCD-aware order (block b -> tile (b % 8) * tiles / 8 + b / 8), so that the halo rows two vertically adjacent tiles share
// are fetched by ONE L2.  A thread renders two rows of 4 consecutive pixels: the flow is read once per pixel as two
// float4 (shared by all channels), the result stored as one float4 per channel and row.  A pixel whose four taps fall
// inside the window reads them from LDS; a pixel whose flow reaches further than R takes global loads, so any flow field is
// handled; same arithmetic on both paths.  The sampling position follows torch's own fp32 steps - base grid as
// torch.linspace(-1, 1, size) builds it (start + step i below the middle, end - step (size - 1 - i) above), offset
// (flow * 2) / (size - 1), un-normalisation ((g + 1) / 2) (size - 1), weights (1 - w) as grid_sample's CPU kernel forms them
// - so that a 1024-wide frame agrees with grid_sample to ~1e-5 (a base grid computed as 2 x / (W - 1) - 1 is off by an ulp
// of 1, i.e. 1e-4 pixels at that width).  grid (tilesX * tilesY, B), block 256.
// Measured (profiles/r05_warp*.json): bit-identical to grid_sample on 512x512 / 1024x1024 frames; HBM-side traffic 1.05-1.09x
// the algorithmic bytes (round 4: 2.3x); 1024x1024 batch 4: 48 us = 2.8 TB/s = 0.35 of the 8 TB/s roof (0.27).  What is left:
// ~180 vector instructions per pixel (two IEEE divisions among them, kept for the exact sampling position) keep the vector
// ALUs busy 40 % of the launch, and a workgroup stages, then computes - a 16-row tile (5 workgroups per CU) is no faster; the
// next step would be a persistent workgroup that stages tile t + 1 under the arithmetic of tile t.
enum { WARP_R = 8, WARP_TW = 64, WARP_TH = 32, WARP_WW = WARP_TW + 2 * WARP_R, WARP_WH = WARP_TH + 2 * WARP_R, WARP_PITCH = WARP_WW + 4 };

__device__ __forceinline__ float warp_linspace(int i, int size) {      // torch.linspace(-1, 1, size)[i] in fp32
  if (size <= 1) return -1.f;
  const float step = 2.f / (float)(size - 1);
  return i < size / 2 ? -1.f + step * (float)i : 1.f - step * (float)(size - 1 - i);
}

__global__ __launch_bounds__(256) void k_warp(const float* img, const float* flow, float* out,
                                              int C, int H, int W, int tilesX, int xcd_chunk) {
  extern __shared__ __attribute__((aligned(16))) float s_win[];     // [C][WARP_WH][WARP_PITCH]
  const int n = blockIdx.y;
  const int tile = xcd_chunk > 0 ? (blockIdx.x & 7) * xcd_chunk + (blockIdx.x >> 3) : blockIdx.x;
  const int ty0 = (tile / tilesX) * WARP_TH, tx0 = (tile % tilesX) * WARP_TW;
  const int wy0 = ty0 - WARP_R, wx0 = tx0 - WARP_R;
  const int HW = H * W;
  const float* src = img + (size_t)n * C * HW;
  const bool vec = (W & 3) == 0;      // rows start 16-byte aligned (hipMalloc'd tensors; wx0 is a multiple of 4)
  constexpr int SEG = WARP_WW / 4;
  for (int i = threadIdx.x; i < C * WARP_WH * SEG; i += 256) {
    const int c = i / (WARP_WH * SEG), r = i - c * (WARP_WH * SEG);
    const int wy = r / SEG, sg = r - wy * SEG;
    const int sy = min(max(wy0 + wy, 0), H - 1), xs = wx0 + 4 * sg;   // border padding = clamped coordinates
    const float* row = src + (size_t)c * HW + (size_t)sy * W;
    float4 v;
    if (vec && xs >= 0 && xs + 3 < W) v = *reinterpret_cast<const float4*>(row + xs);
    else v = make_float4(row[min(max(xs, 0), W - 1)], row[min(max(xs + 1, 0), W - 1)], row[min(max(xs + 2, 0), W - 1)], row[min(max(xs + 3, 0), W - 1)]);
    *reinterpret_cast<float4*>(s_win + (c * WARP_WH + wy) * WARP_PITCH + 4 * sg) = v;
  }
  __syncthreads();
  const int x = tx0 + 4 * (threadIdx.x & 15);
  const float fw = (float)(W - 1), fh = (float)(H - 1);
#pragma unroll
  for (int k = 0; k < WARP_TH / 16; ++k) {
    const int y = ty0 + (threadIdx.x >> 4) + 16 * k;
    if (y >= H || x >= W) continue;
    const size_t pix = (size_t)y * W + x;
    const bool full = vec && x + 3 < W;
    float fx[4], fy[4];
    if (full) {
      const float4 a = *reinterpret_cast<const float4*>(flow + ((size_t)n * 2 + 0) * HW + pix);
      const float4 b = *reinterpret_cast<const float4*>(flow + ((size_t)n * 2 + 1) * HW + pix);
      fx[0] = a.x; fx[1] = a.y; fx[2] = a.z; fx[3] = a.w; fy[0] = b.x; fy[1] = b.y; fy[2] = b.z; fy[3] = b.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const size_t q = (size_t)y * W + min(x + j, W - 1);
        fx[j] = flow[((size_t)n * 2 + 0) * HW + q]; fy[j] = flow[((size_t)n * 2 + 1) * HW + q];
      }
    }
    const float by = warp_linspace(y, H);
    float w00[4], w01[4], w10[4], w11[4];
    int o00[4], o01[4], o10[4], o11[4];      // tap offsets: into the window (in_win) or into a channel plane
    bool in_win[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      // grid_sample un-normalises in fp32: ((g + 1) / 2) * (size - 1) with g = base + flow * 2 / (size - 1)
      const float gx = warp_linspace(min(x + j, W - 1), W) + (W > 1 ? (fx[j] * 2.f) / fw : 0.f);
      const float gy = by + (H > 1 ? (fy[j] * 2.f) / fh : 0.f);
      float sx = ((gx + 1.f) / 2.f) * fw, sy = ((gy + 1.f) / 2.f) * fh;
      sx = fminf(fmaxf(sx, 0.f), fw);
      sy = fminf(fmaxf(sy, 0.f), fh);
      const float xf = floorf(sx), yf = floorf(sy);
      const int x0 = (int)xf, y0 = (int)yf;
      const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
      const float ax = sx - xf, ay = sy - yf, ex = 1.f - ax, ey = 1.f - ay;
      w00[j] = ex * ey; w01[j] = ax * ey; w10[j] = ex * ay; w11[j] = ax * ay;
      // window slot of source column / row s: the window holds clamp(w0 + k) at slot k, so an in-image s sits at s - w0
      const int lx0 = x0 - wx0, lx1 = x1 - wx0, ly0 = y0 - wy0, ly1 = y1 - wy0;
      in_win[j] = lx0 >= 0 && lx1 < WARP_WW && ly0 >= 0 && ly1 < WARP_WH;
      if (in_win[j]) { o00[j] = ly0 * WARP_PITCH + lx0; o01[j] = ly0 * WARP_PITCH + lx1; o10[j] = ly1 * WARP_PITCH + lx0; o11[j] = ly1 * WARP_PITCH + lx1; }
      else { o00[j] = y0 * W + x0; o01[j] = y0 * W + x1; o10[j] = y1 * W + x0; o11[j] = y1 * W + x1; }
    }
    // (the two tap sources are kept apart: a pointer selected between LDS and global memory is a generic pointer, and its loads
    // go down the flat path - 145 vector-memory instructions per wavefront instead of LDS reads, round 5's first version)
    const bool all_win = in_win[0] && in_win[1] && in_win[2] && in_win[3];
    for (int c = 0; c < C; ++c) {
      const float* wsrc = s_win + c * WARP_WH * WARP_PITCH;
      const float* gsrc = src + (size_t)c * HW;
      float r[4];
      if (__all(all_win)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = wsrc[o00[j]] * w00[j] + wsrc[o01[j]] * w01[j] + wsrc[o10[j]] * w10[j] + wsrc[o11[j]] * w11[j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (in_win[j]) r[j] = wsrc[o00[j]] * w00[j] + wsrc[o01[j]] * w01[j] + wsrc[o10[j]] * w10[j] + wsrc[o11[j]] * w11[j];
          else r[j] = gsrc[o00[j]] * w00[j] + gsrc[o01[j]] * w01[j] + gsrc[o10[j]] * w10[j] + gsrc[o11[j]] * w11[j];
        }
      }
      float* dst = out + ((size_t)n * C + c) * HW + pix;
      if (full) *reinterpret_cast<float4*>(dst) = make_float4(r[0], r[1], r[2], r[3]);
      else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (x + j < W) dst[j] = r[j];
      }
    }
  }
}

}  // namespace rib
