// igemm.hip.h — k_igemm, the implicit-GEMM convolution / fused-SPADE kernel of the generator forward, and the helpers every
// kernel shares.  This is the header the igemm_shard_<s>.o objects are compiled from (csrc/build.py hashes it into their
// stamps); the other kernels live in kernels.hip.h, which includes this file and is compiled into rib.o only - editing
// them does not rebuild the ~320 k_igemm instantiations.
//
// Hand-written CDNA4 (gfx950) kernels of the generator forward.
//
// Data layout: every activation is NHWC fp32 in HBM (channel-contiguous: 16-byte coalesced
// loads along C, concatenation = channel-offset writes, K of the implicit GEMM contiguous).
// The dominant kernel, k_igemm, is an implicit-GEMM convolution on the exact-fp32 matrix
// cores (v_mfma_f32_32x32x2_f32): M = output pixels of one spatial tile, N = output channels,
// K = taps x input channels.  The input halo tile is staged ONCE per channel chunk into LDS
// (with the fused InstanceNorm-affine + LeakyReLU prologue applied on the way in), and the
// taps read shifted windows of it; filter slices are double-buffered in LDS.
//
// Reference semantics restated by these kernels (PGNR = Pose_Guided_Neural_Rendering):
//   conv / zero padding / stride 2      PGNR/models/layers/conv.py:93-104, generator.py:55,344-348
//   InstanceNorm (biased var, eps 1e-5) PGNR/models/layers/activation_norm.py:399-402
//   SPADE  IN(x)*(1+gamma)+beta         PGNR/models/layers/activation_norm.py:211-234
//   LeakyReLU(0.2), sigmoid, tanh       PGNR/models/layers/nonlinearity.py:21-28, generator.py:228
//   nearest x2 upsample (src = dst>>1)  PGNR/models/generator.py:128,248-249,480
//   AvgPool2d(3,2,1) divisor 9          PGNR/models/generator.py:127
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

namespace rib {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// 8 consecutive fp32 values -> one bf16 MFMA operand (round to nearest even: v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16x8 to_bf16x8(const float4 lo, const float4 hi) {
  bf16x8 r;
  r[0] = (__bf16)lo.x; r[1] = (__bf16)lo.y; r[2] = (__bf16)lo.z; r[3] = (__bf16)lo.w;
  r[4] = (__bf16)hi.x; r[5] = (__bf16)hi.y; r[6] = (__bf16)hi.z; r[7] = (__bf16)hi.w;
  return r;
}

enum { ACT_NONE = 0, ACT_LRELU = 1, ACT_TANH = 2, ACT_SIGMOID = 3 };

// fuse = img*m + dain*(1-m) (PGNR/models/evaluator.py:256-258) with torch's roundings: two products, one difference,
// one sum, nothing contracted into an fma - the stand-alone k_blend and the blend fused into the mask head agree bit
// for bit with each other and with the reference's expression
__device__ __forceinline__ float blend1(float img, float m, float dain) {
  return __fadd_rn(__fmul_rn(img, m), __fmul_rn(dain, __fsub_rn(1.f, m)));
}

// ---- storage type ST of the activations / filters: ST_F32 (default, the reference's arithmetic), ST_BF16 (BASELINE
// configs[2]: bf16 NHWC tensors in HBM and bf16 tiles in LDS - half the bytes everywhere - bf16 matrix-core
// operands, fp32 accumulation, fp32 InstanceNorm statistics of the ROUNDED values, fp32 SPADE arithmetic) or ST_F16
// (round 3: the same 16-bit layouts and kernels with IEEE half elements and v_mfma_f32_32x32x16_f16 - 11 significant bits
// instead of 8: the bf16 mode's error is the format's, DESIGN 6, and this network's activations and filters sit well
// inside half's range).  Pointers stay `float*` in the parameter structs; a 16-bit kernel indexes them as 2-byte elements.
enum { ST_F32 = 0, ST_BF16 = 1, ST_F16 = 2 };
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int ST> __device__ __forceinline__ float h16_to_f32(uint16_t b) {
  if constexpr (ST == ST_F16) return (float)__builtin_bit_cast(_Float16, b);
  else return __uint_as_float((uint32_t)b << 16);
}
template <int ST> __device__ __forceinline__ uint16_t f32_to_h16(float v) {      // round to nearest even (v_cvt_pk_bf16_f32 / v_cvt_f16_f32)
  if constexpr (ST == ST_F16) { const _Float16 hv = (_Float16)v; return __builtin_bit_cast(uint16_t, hv); }
  else { const __bf16 b = (__bf16)v; return __builtin_bit_cast(uint16_t, b); }
}
template <int ST> __device__ __forceinline__ float round16(float v) { return h16_to_f32<ST>(f32_to_h16<ST>(v)); }
template <int ST> __device__ __forceinline__ float ld_act(const float* base, size_t i) {
  if constexpr (ST != ST_F32) return h16_to_f32<ST>(reinterpret_cast<const uint16_t*>(base)[i]);
  else return base[i];
}
template <int ST> __device__ __forceinline__ void st_act(float* base, size_t i, float v) {
  if constexpr (ST != ST_F32) reinterpret_cast<uint16_t*>(base)[i] = f32_to_h16<ST>(v);
  else base[i] = v;
}
// four consecutive elements (16 B of fp32 / 8 B of a 16-bit type)
template <int ST> __device__ __forceinline__ float4 ld_act4(const float* base, size_t i) {
  if constexpr (ST != ST_F32) {
    const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const uint16_t*>(base) + i);
    return make_float4(h16_to_f32<ST>((uint16_t)(r.x & 0xffffu)), h16_to_f32<ST>((uint16_t)(r.x >> 16)),
                       h16_to_f32<ST>((uint16_t)(r.y & 0xffffu)), h16_to_f32<ST>((uint16_t)(r.y >> 16)));
  } else return *reinterpret_cast<const float4*>(base + i);
}
template <int ST> __device__ __forceinline__ void st_act4(float* base, size_t i, float4 v) {
  if constexpr (ST != ST_F32) {
    uint2 r;
    r.x = (uint32_t)f32_to_h16<ST>(v.x) | ((uint32_t)f32_to_h16<ST>(v.y) << 16);
    r.y = (uint32_t)f32_to_h16<ST>(v.z) | ((uint32_t)f32_to_h16<ST>(v.w) << 16);
    *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(base) + i) = r;
  } else *reinterpret_cast<float4*>(base + i) = v;
}

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : 0.2f * v; }
__device__ __forceinline__ float4 lrelu4(float4 v) {
  return make_float4(lrelu(v.x), lrelu(v.y), lrelu(v.z), lrelu(v.w));
}
__device__ __forceinline__ float apply_act(float v, int act) {
  if (act == ACT_LRELU) return lrelu(v);
  if (act == ACT_TANH) return tanhf(v);
  if (act == ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
  return v;
}

// ---------------------------------------------------------------------------------------------
// k_igemm parameters.  All pointers are device pointers; tensors NHWC with an explicit channel
// stride (xC, yC, ...) so that channel slices of wider buffers can be read / written in place.
// ---------------------------------------------------------------------------------------------
struct IgemmParams {
  // A operand: input activation (conv) or SPADE condition map
  const float* x;
  int Hin, Win, xC;      // stored spatial size and channel stride of x
  int Cin;               // channels consumed (multiple of BK, <= xC)
  // prologue on x: v = lrelu?(v*scale[n][c] + shift[n][c]); zero padding applied AFTER it
  const float* pro_scale;  // [B][pro_ld] or nullptr
  const float* pro_shift;
  int pro_ld;
  int pro_lrelu;
  // optional second operand of a fused 1x1 convolution that accumulates into the same output
  // (the learned shortcut of a residual block: out = conv3x3(y1) + conv1x1(ys) + bias):
  // x2 [B][Hout][Wout][x2C], w2 [CoutPad][Cin2]; 3x3 stride-1 variants only
  const float* x2; const float* w2;
  int x2C, Cin2;
  // B operand: filters [CoutPad][taps][Cin] (Cin contiguous), bias [CoutPad]
  const float* w;
  const float* bias;
  int CoutPad;           // rows present in w / bias (multiple of 32)
  // output
  int Hout, Wout;
  int tilesX, tilesY;
  int xcd_chunk;         // > 0: tiles/8; block b works on tile (b%8)*xcd_chunk + b/8, so that each XCD's L2
                         //      (workgroups go round-robin over the 8 XCDs) holds one contiguous band of tiles
  // --- conv epilogue ---
  float* y;              // [B][Hout][Wout][yC], written at channel offset yoff
  int yC, yoff, Cout;    // Cout = valid output channels
  int y_f32;             // bf16-storage kernels: y is a caller's fp32 tensor (the mask head), not a workspace activation
  int act;
  const float* res;      // residual added before act/store, [B][Hout][Wout][resC] or nullptr
  int resC, res_ups;     // res_ups: residual stored at half resolution (nearest x2 upsample on read)
  float* y_nchw;         // optional second copy of the output as [B][Cout][Hout][Wout]
  double* stat_part;     // optional per-tile partial sums [B][tiles][2][CoutPad], fp64 from the first add on
  // (the bf16 matrix-core mode, v_mfma_f32_32x32x16_bf16 on the same fp32 LDS tiles with fp32
  // accumulate / statistics / storage, is the BF16 template flag: a runtime switch inside the tap
  // loop cost the fp32 kernels 8 % through register pressure)
  int ksplit;            // >= 1: K is split over blockIdx.z
  float* slab;           // when set: raw partial sums go to [ksplit][B][Hout][Wout][CoutPad] instead of y
  // --- SPADE epilogue (template SPADE): out_s = act_s( (xm*scale+shift)*(1+gamma)+beta ) ---
  const float* xm;       // tensor being normalised, [B][Hm][Wm][xmC]
  int xmC, xm_ups;       // xm_ups: xm is stored at half resolution (nearest x2 upsample on read)
  const float* m_scale;  // [B][m_ld] = rstd          (IN affine=False)
  const float* m_shift;  // [B][m_ld] = -mean*rstd
  int m_ld;
  int C;                 // channels of xm
  int nsets;             // 1 or 2 modulations of the same xm (conv_block_0 + conv_block_s)
  float* ys0; float* ys1;  // outputs [B][Hout][Wout][C]
  int act0, act1;
  // --- consumer-side InstanceNorm finalize ---
  // When the producer of a normalised tensor left few per-tile partial sums (<= STATS_MAX_PARTIALS), the consumer
  // reduces them itself instead of reading (scale, shift) arrays written by a k_stats_finalize launch: one dependent
  // launch less per normalised tensor.  part = [B][tiles][2][Cs] fp64: the producers' epilogues accumulate sum(x)
  // and sum(x^2) in fp64 from the element level, so E[x^2] - mean^2 does not cancel when |mean| >> std (an fp32
  // partial of x^2 loses the variance once mean^2 / var reaches ~1e6; torch's instance_norm does not).
  const double* m_part; int m_tiles, m_Cs; float m_inv;          // SPADE epilogue: replaces m_scale / m_shift (no affine)
  // --- driver blend fused into the mask head (k_conv_head<1>): fuse = img*m + dain*(1-m), NCHW fp32 caller tensors
  // (PGNR/models/evaluator.py:256-258); all three null when the caller did not ask for the fused frame ---
  const float* bl_img; const float* bl_dain; float* bl_fuse; int bl_C;
  // --- batched GEMM with one filter set per "sample" (the 16 Winograd positions: k_wino_in / k_wino_out): sample n
  // reads the filters at w + (n % w_mod) * w_stride elements; w_mod = 0: one filter set for all samples ---
  int w_mod; unsigned w_stride;
  // --- two convolutions of identical shape in ONE launch (round 3: level i of the mask network's label encoder and of its
  // image encoder, PGNR/models/generator.py:449-459): samples come in pairs, sample 2b + j is image b of convolution j;
  // w_mod = 2 picks the filters, b_stride (floats) the bias.  pair != 0: the two results of a pair land in ONE output
  // image, y[b] at channel offsets yoff and yoff + pair_yoff (the torch.cat of generator.py:505); the statistics partials
  // stay per sample ---
  unsigned b_stride; int pair, pair_yoff;
  // --- DMA instantiations (operand tiles staged by global_load_lds_dwordx4): 64 bytes of zeros in device memory, the
  // source of the input tile's out-of-image (zero padding) pixels ---
  const float* zeros;
};

enum { STATS_MAX_PARTIALS = 128 };

// (scale, shift) of ONE channel from the per-tile partial sums of its producer; `base` points at the sample's
// [tiles][2][Cs] block.  Same arithmetic as k_stats_finalize (fp64 sums in a fixed tile order, biased variance,
// eps 1e-5), so the two paths agree to the last bit whenever the tile order of the sum is the same.
__device__ __forceinline__ void stats_from_partials(const double* base, int tiles, int Cs, int c, int t0, int tstep,
                                                    double& a1, double& a2) {
  a1 = 0.0; a2 = 0.0;
  int t = t0;
  for (; t + 3 * tstep < tiles; t += 4 * tstep) {   // 8 independent loads in flight
    const double x0 = base[(size_t)t * 2 * Cs + c], y0 = base[(size_t)t * 2 * Cs + Cs + c];
    const double x1 = base[(size_t)(t + tstep) * 2 * Cs + c], y1 = base[(size_t)(t + tstep) * 2 * Cs + Cs + c];
    const double x2 = base[(size_t)(t + 2 * tstep) * 2 * Cs + c], y2 = base[(size_t)(t + 2 * tstep) * 2 * Cs + Cs + c];
    const double x3 = base[(size_t)(t + 3 * tstep) * 2 * Cs + c], y3 = base[(size_t)(t + 3 * tstep) * 2 * Cs + Cs + c];
    a1 += (x0 + x1) + (x2 + x3);
    a2 += (y0 + y1) + (y2 + y3);
  }
  for (; t < tiles; t += tstep) {
    a1 += base[(size_t)t * 2 * Cs + c];
    a2 += base[(size_t)t * 2 * Cs + Cs + c];
  }
}
__device__ __forceinline__ void scale_shift_of(double s1, double s2, float inv_count, float g, float b, float& sc, float& sh) {
  const double mean = s1 * (double)inv_count;
  double var = s2 * (double)inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + 1e-5));
  sc = rstd * g;
  sh = b - (float)mean * sc;
}

// Tile geometry: a 32-row MFMA fragment covers FRH x FRW pixels (FRH = 32 / FRW); a wave owns
// MF fragments stacked vertically and NF 32-channel column fragments; the workgroup is
// WM x WN waves (WM*WN == 4).  Spatial tile = (FRH*MF*WM) x FRW pixels, BN = 32*NF*WN channels.
// PREC: 0 = fp32 (exact-fp32 matrix cores), 1 = bf16 storage + bf16 matrix cores.  (Round 2's exploratory "f32x3" mode -
// fp32 storage, operands split into three bf16 terms, six bf16 MFMAs per step - never beat fp32 once the deep layers ran
// in the Winograd domain and was retired in round 3.)
enum { PREC_F32 = 0, PREC_BF16 = 1, PREC_F16 = 2 };      // = ST_F32 / ST_BF16 / ST_F16
// DMA (round 3): bit 0 = the filter slices, bit 1 = the input tile are staged by LDS-DMA (global_load_lds_dwordx4: no staging
// registers, no ds_write) instead of global -> registers -> LDS.  A DMA instruction lands 64 consecutive 16-byte slots, so the
// rows of a DMA-staged tile are NOT padded; the 16-byte slot of a row is XOR-swizzled instead (swz below), which is
// conflict-free for the filter reads and nearly so for the shifted-window reads of the input tile.  The DMA-staged input
// tile is double-buffered (the fill of chunk k + 1 lands while chunk k is being read).
template <int FRW, int WM, int WN, int MF, int NF, int BK, int STRIDE, int KS, bool UPS, int KW = 1, int TB = 1, int PREC = 0, int DMA = 0>
struct IgemmGeom {
  static constexpr bool BF16 = PREC != PREC_F32;      // 16-bit storage (bf16 or half): the layouts only depend on the element size
  static constexpr int NT = 256 * KW;          // threads: KW groups of 4 waves share the tile and split each tap's K
  static constexpr int FRH = 32 / FRW;
  static constexpr int TH = FRH * MF * WM;
  static constexpr int TW = FRW;
  // UPS (3x3 on a nearest-x2-upsampled input) runs as four 2x2 "phase" convolutions of the stored
  // half-resolution input (see k_igemm): the tile is TH x TW SOURCE pixels = 2TH x 2TW output pixels,
  // its halo is the 3x3 halo of the source tile, and there are 4 phases x 4 taps = 16 filter slices
  static constexpr int PH = UPS ? 4 : 1;
  static constexpr int IH = UPS ? (TH + 2) : ((TH - 1) * STRIDE + KS);
  static constexpr int IW = UPS ? (TW + 2) : ((TW - 1) * STRIDE + KS);
  static constexpr int EPS = BF16 ? 8 : 4;     // activation elements per 16-byte staging slot
  static constexpr int KF = BK * 4 / EPS;     // floats of LDS one pixel's / filter row's K chunk takes (bf16: BK / 2)
  static constexpr int GPR = BK / EPS;         // 16-byte global loads per input pixel and chunk
  static constexpr int GPRB = KF / 4;          // 16-byte global loads (= LDS slots) per filter row and slice
  static constexpr int CK = KF + 4;            // padded LDS row: conflict-free ds_read_b128
  // Stride 2: the halo tile is stored with even and odd columns de-interleaved (column x -> (x & 1) * IWH + x / 2),
  // so that the 16 lanes of a ds_read_b128 phase, which step 2 pixels in x, read consecutive LDS pixels as in the
  // stride-1 case (stepping 2 * CK floats they hit only half of the banks: 35-49 % of the LDS cycles of the
  // stride-2 layers were bank conflicts, PMC SQ_LDS_BANK_CONFLICT).  8-wide fragments put two tile rows into one
  // phase: their LDS row pitch is padded to 4 (mod 8) pixels, which moves the second row onto the other 32 banks.
  static constexpr int IWH = (IW + 1) / 2;
  static constexpr int IWP = (STRIDE == 2 && FRW == 8) ? ((IW + 3) / 8 * 8 + 4) : IW;   // LDS row pitch in pixels
  // NF == 0 selects the 16-column path (v_mfma_f32_16x16x4_f32) for layers with <= 16 output
  // channels: no half-empty 32-column fragments
  static constexpr int BN = NF == 0 ? 16 * WN : 32 * NF * WN;
  static constexpr int AP = (DMA & 2) ? KF : CK;     // floats per pixel row of the input tile in LDS
  static constexpr int BP = (DMA & 1) ? KF : CK;     // floats per filter row in LDS
  static constexpr int SLOTS = KF / 4;               // 16-byte slots per row of a K chunk
  static constexpr int SWZ_SHIFT = SLOTS >= 16 ? 0 : (SLOTS == 8 ? 1 : (SLOTS == 4 ? 2 : 3));
  // physical slot of logical slot s in row r of a DMA-staged tile: the 16 lanes of a ds_read_b128 group (rows 0-3, 12-15,
  // 20-27 / 4-11, 16-19, 28-31 of a fragment at one slot) then hit 16 different 4-bank groups
  __device__ static constexpr int swz(int r, int s) { return s ^ ((r >> SWZ_SHIFT) & (SLOTS - 1)); }
  static constexpr int NQA = (DMA & 2) ? (IH * IWP * SLOTS + 255) / 256 : 0;   // DMA instructions per wave for the input tile of a chunk
  static constexpr int NQB = (DMA & 1) ? (BN * SLOTS + 255) / 256 : 0;         //                            for one filter slice
  static constexpr int SA = (DMA & 2) ? NQA * 256 * 4 : IH * IWP * CK;         // floats (DMA: whole instructions)
  static constexpr int SB = (DMA & 1) ? NQB * 256 * 4 : BN * CK;               // floats, one of two buffers
  static constexpr int NB4 = (BN * GPRB + NT - 1) / NT;    // 16-byte filter loads per thread per tap
  static constexpr int SRED = WM * BN * 4;     // floats: [WM][BN][2] fp64 statistics partials
  static constexpr int SKW = KW > 1 ? (NF == 0 ? 1 : NF) * MF * 16 * 256 : 0;   // one wave group's accumulators
  // TB = 9 (3x3) and TB = 2 (1x1: "chunk pairs") also double-buffer the input tile: one barrier per chunk
  static constexpr int NA = (TB == 9 || (KS == 1 && TB == 2) || (DMA & 2)) ? 2 : 1;
  static constexpr int TBB = (KS == 1 && TB == 2) ? 1 : TB;   // filter slices per buffer
  static constexpr int SMEM0 = NA * SA + 2 * TBB * SB > SRED ? NA * SA + 2 * TBB * SB : SRED;
  static constexpr int SMEM = SMEM0 > SKW ? SMEM0 : SKW;
  static constexpr int TAPS = UPS ? 16 : KS * KS;   // filter slices per channel chunk (row stride of w)
};

// AUX / PRO: whether the fused-1x1-shortcut loop and the input prologue are compiled in.  They are
// run-time options of the generic kernel, but merely carrying their code costs 24 + 15 VGPRs in the
// main loop (135 -> 89 for the 8x16 / 32-column / 32-channel variant: 3 -> 4 waves per SIMD), so the
// launcher picks the leanest instantiation that covers a launch (pick_igemm_fn in rib.hip).
// TB = 3: the three filter slices of one filter row are staged per barrier (3 barriers per chunk instead
// of 9, +4 slices of LDS): pays on launches that leave LDS to spare (<= 2 workgroups per CU).
// KW > 1: in-workgroup split-K.  KW groups of 4 waves (256*KW threads) work on the SAME tile: they share
// the staged input tile and filter slices, each group runs 1/KW of every tap's channel steps, and the
// partial accumulators are summed through LDS before the epilogue.  Gives an under-filled launch KW x
// the wavefronts without the slab round trip and the second launch of grid-level split-K.
// UPS: a 3x3 convolution of a nearest-x2-upsampled tensor (nn.Upsample(2) -> Conv2dBlock, PGNR/models/
// generator.py:480).  Output pixel (2Y+py, 2X+px) reads upsampled rows 2Y+py-1+dy, i.e. source rows
// Y-1+((py+dy)>>1): only TWO distinct source rows (and columns) per phase (py, px), so the nine taps
// collapse to a 2x2 filter per phase whose entries are sums of the original taps
//   py = 0: rows {Y-1: w[0], Y: w[1]+w[2]}     py = 1: rows {Y: w[0]+w[1], Y+1: w[2]}   (same along x)
// (summed on the host, rib_finalize_weights): 4/9 of the matrix work and the same zero padding (source
// row -1 / H is exactly where the upsampled row -1 / 2H falls).  A workgroup owns a TH x TW tile of
// SOURCE pixels and keeps four accumulator sets, one per phase; filter slice t = phase*4 + a*2 + b
// multiplies the shifted window (py+a, px+b) of the ordinary 3x3 halo tile of the source.
template <int FRW, int WM, int WN, int MF, int NF, int BK, int STRIDE, int KS, bool UPS, bool SPADE, int PREC = 0,
          bool AUX = true, bool PRO = true, int KW = 1, int TB = 1, int DMA = 0>
// Second launch bound = minimum waves per SIMD the register allocator must leave room for.  The fp32 variants with
// two column fragments per wave (NF = 2: 32 accumulator registers) sit exactly on the 128-register boundary of 4 waves
// per SIMD; one more live value in an epilogue made the allocator give up and settle at 3 (97 + 32 registers), which
// cost the launches using them 5-20 % (tools/occupancy_diff.py).  With the bound it keeps the accumulators in VGPRs
// and fits 99-104 registers without spilling.  (Likewise the 16-bit 8x16 BN32 BK16 generic variant: 81 + 16 registers is one
// over the boundary of 5 waves.)
__global__ __launch_bounds__(256 * KW, ((PREC == 0 || SPADE) && NF == 2 && MF == 1 && !UPS && KW == 1) ? 4 : ((PREC == 0 && UPS && MF * NF == 1 && KW == 1) ? 2 : ((PREC != 0 && NF == 1 && MF == 1 && BK == 16 && KS == 3 && !UPS && KW == 1 && TB == 1) ? 5 : 1))) void k_igemm(const IgemmParams p) {
  typedef IgemmGeom<FRW, WM, WN, MF, NF, BK, STRIDE, KS, UPS, KW, TB, PREC, DMA> G;
  constexpr int DM = DMA & 3;                   // which operands are staged by LDS-DMA
  static_assert(DM == 0 || (PREC == PREC_F32 && (TB == 1 || (TB == 9 && DM == 3 && KS == 3)) && KW == 1 && !AUX && NF > 0 && !UPS && (DMA & 1)),
                "DMA staging: fp32, 32-column path, no fused shortcut; one slice per barrier, or all nine of a chunk with a DMA-staged input tile");
  static_assert(!(DMA & 2) || !PRO, "the input tile can only be staged by DMA when no prologue transforms it on the way into LDS");
  constexpr bool BF16 = G::BF16;                // 16-bit storage, bf16 or half (ST says which)
  constexpr int ST = PREC;
  constexpr int ESZ = BF16 ? 2 : 4;             // bytes per stored activation element
  constexpr int WSZ = BF16 ? 2 : 4;     // bytes per stored filter element
  constexpr int EPS = G::EPS, GPR = G::GPR, GPRB = G::GPRB;
  constexpr int EPB = 16 / WSZ;                 // filter elements per 16-byte slot
  static_assert(TB == 1 || ((TB == 3 || TB == 9) && KS == 3 && !UPS) || (TB == 4 && UPS) || (TB == 2 && KS == 1 && !UPS && STRIDE == 1),
                "filter slices per barrier: one tap, one row of a 3x3 filter, all nine; phase convolutions: the four taps of a phase");
  constexpr int NT = G::NT;
  static_assert(KW == 1 || (NF > 0 && (BK / (BF16 ? 16 : 8)) % KW == 0), "in-workgroup split-K: 32-column path (conv or SPADE), K steps of a chunk divisible by KW");
  constexpr bool N16 = (NF == 0);
  constexpr int NFE = N16 ? 1 : NF;
  static_assert(WM * WN == 4, "4 waves per workgroup");
  static_assert(FRW >= 8 && 32 % FRW == 0, "fragment = 32 / FRW rows of FRW pixels; the epilogue's element -> pixel map needs FRW >= 8");
  static_assert(!SPADE || (NF % 2 == 0 && NF > 0) || (NF == 1 && PREC == PREC_F32 && KW == 1), "SPADE needs gamma/beta fragment pairs, or ONE fragment [gamma(16) | beta(16)] (fp32)");
  static_assert(!N16 || (FRW == 16 && WN == 1 && STRIDE == 1 && !UPS && !SPADE && BK % 16 == 0 && PREC == PREC_F32), "16-column path: 8x16-style tiles only, fp32");
  static_assert(!UPS || (STRIDE == 1 && KS == 3 && KW == 1 && (TB == 1 || TB == 4) && NF > 0 && !SPADE), "phase-decomposed upsample conv: 3x3 stride 1, 32-column path");
  constexpr int PH = G::PH;
  static_assert(NT % GPR == 0, "a thread keeps one channel group across its staging slots");
  __shared__ __attribute__((aligned(16))) float smem[G::SMEM];
  // consumer-side InstanceNorm finalize: (scale, shift) of the input channels (prologue) / of this workgroup's
  // modulated channels (SPADE epilogue), reduced from the producer's partial sums at kernel start
  __shared__ __attribute__((aligned(16))) float s_stat[SPADE ? 2 * (G::BN / 2) : 4];
  float* sA = smem;
  float* sB = smem + G::NA * G::SA;

  const int tid = (int)threadIdx.x;
  const int lane = tid & 63;
  const int kw = KW == 1 ? 0 : (tid >> 8);      // wave group (in-workgroup K slice)
  const int wave = (tid >> 6) & 3;
  const int wm = wave / WN, wn = wave % WN;
  const int li = lane & 31, lh = lane >> 5;

  // blockIdx.z = n * ksplit + split: split-K slices share the tile and write partial slabs
  const int n = blockIdx.z / p.ksplit;
  const int split = blockIdx.z - n * p.ksplit;
  const int tile = p.xcd_chunk ? (int)(blockIdx.x & 7) * p.xcd_chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
  const int ty0 = (tile / p.tilesX) * G::TH;
  const int tx0 = (tile % p.tilesX) * G::TW;
  const int n0 = blockIdx.y * G::BN;
  const int nchunks = p.Cin / BK;
  const int kc_begin = (split * nchunks / p.ksplit) * BK;
  const int kc_end = ((split + 1) * nchunks / p.ksplit) * BK;

  // input-tile origin in stored-input coordinates
  int iy0, ix0;
  if (UPS) { iy0 = ty0 - 1; ix0 = tx0 - 1; }   // ty0 / tx0 are SOURCE coordinates in phase mode
  else { iy0 = ty0 * STRIDE - (KS / 2); ix0 = tx0 * STRIDE - (KS / 2); }

  // per-lane tile pixel of each M fragment
  int fy[MF], fx;
  fx = li % FRW;
#pragma unroll
  for (int mf = 0; mf < MF; ++mf) fy[mf] = (wm * MF + mf) * G::FRH + li / FRW;

  f32x16 acc[PH * MF][NFE];   // UPS: accumulator set ph*MF + mf belongs to phase ph = py*2 + px
  f32x4 acc16[MF][2];     // 16-column path: two 16-pixel sub-fragments (tile rows) per 32-pixel block
  if (!N16) {
#pragma unroll
    for (int mf = 0; mf < PH * MF; ++mf)
#pragma unroll
      for (int nf = 0; nf < NFE; ++nf)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mf][nf][r] = 0.f;
  } else {
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc16[mf][sub][r] = 0.f;
  }

  const char* xn = reinterpret_cast<const char*>(p.x) + (size_t)n * p.Hin * p.Win * p.xC * ESZ;
  const char* wb = reinterpret_cast<const char*>(p.w) + (p.w_mod ? (size_t)(n % p.w_mod) * p.w_stride * WSZ : 0);
  const int wrow = G::TAPS * p.Cin;   // elements per filter row
  // byte offset of the 16-byte slot g of filter row `row`, slice `tap`, chunk kc: [row][tap][Cin]
  auto w_off = [&](int row, int tap, int kc, int g) -> size_t {
    return ((size_t)row * wrow + tap * p.Cin + kc + g * EPB) * WSZ;
  };

  // ---- operand staging, software-pipelined through registers: the global loads of the NEXT
  // filter slice / input chunk are in flight while the current one feeds the matrix cores ----
  float4 breg[G::NB4 * G::TBB];
  auto loadB = [&](int kc, int tap) {
#pragma unroll
    for (int i = 0; i < G::NB4; ++i) {
      const int idx = tid + i * NT;
      const int row = idx / GPRB, c4 = idx % GPRB;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < G::BN && n0 + row < p.CoutPad)
        v = *reinterpret_cast<const float4*>(wb + w_off(n0 + row, tap, kc, c4));
      breg[i] = v;
    }
  };
  auto storeB = [&](int buf) {
#pragma unroll
    for (int i = 0; i < G::NB4; ++i) {
      const int idx = tid + i * NT;
      const int row = idx / GPRB, c4 = idx % GPRB;
      if (row < G::BN)
        *reinterpret_cast<float4*>(sB + buf * G::SB + row * G::CK + c4 * 4) = breg[i];
    }
  };

  auto loadB3 = [&](int kc, int dy) {     // TB == 3: the three slices of filter row dy; TB == 9: all nine (dy = 0)
#pragma unroll
    for (int t = 0; t < (TB > 1 ? TB : 3); ++t)
#pragma unroll
      for (int i = 0; i < G::NB4; ++i) {
        const int idx = tid + i * NT;
        const int row = idx / GPRB, c4 = idx % GPRB;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < G::BN && n0 + row < p.CoutPad)
          v = *reinterpret_cast<const float4*>(wb + w_off(n0 + row, dy * (TB == 4 ? 4 : 3) + t, kc, c4));
        breg[(TB > 1 ? t : 0) * G::NB4 + i] = v;
      }
  };
  auto storeB3 = [&](int buf) {
#pragma unroll
    for (int t = 0; t < (TB > 1 ? TB : 3); ++t)
#pragma unroll
      for (int i = 0; i < G::NB4; ++i) {
        const int idx = tid + i * NT;
        const int row = idx / GPRB, c4 = idx % GPRB;
        if (row < G::BN)
          *reinterpret_cast<float4*>(sB + (buf * (TB > 1 ? TB : 3) + t) * G::SB + row * G::CK + c4 * 4) = breg[(TB > 1 ? t : 0) * G::NB4 + i];
      }
  };

  constexpr int total4 = G::IH * G::IW * GPR;
  constexpr int NA4 = (total4 + NT - 1) / NT;
  const int ac4 = tid % GPR;                // this thread's channel group (EPS channels), the same in every slot
  float4 areg[NA4];                         // 16 raw bytes per slot: 4 fp32 or 8 bf16 channels
  constexpr int PV = EPS / 4;               // float4s of prologue constants per slot
  float4 psc[PV], psh[PV];
#pragma unroll
  for (int q = 0; q < PV; ++q) { psc[q] = make_float4(1.f, 1.f, 1.f, 1.f); psh[q] = make_float4(0.f, 0.f, 0.f, 0.f); }
  auto slot_inb = [&](int i, int& pix, int& iy, int& ix) -> bool {
    const int idx = tid + i * NT;
    pix = idx / GPR;
    const int ly = pix / G::IW, lx = pix % G::IW;
    iy = iy0 + ly; ix = ix0 + lx;
    return idx < total4 && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
  };
  // element offset of slot i's 16 bytes at chunk 0, or -1 for a slot outside the image (zero padding) or beyond the tile: the
  // pixel decomposition, the bounds tests and the multiplies are done once, not per chunk in prefetchA AND in writeA
  int aslot[NA4];
#pragma unroll
  for (int i = 0; i < NA4; ++i) {
    int pix, iy, ix;
    aslot[i] = slot_inb(i, pix, iy, ix) ? (iy * p.Win + ix) * p.xC + ac4 * EPS : -1;      // < 2^31 elements per sample (checked by the host)
  }
  auto prefetchA = [&](int kc) {
#pragma unroll
    for (int i = 0; i < NA4; ++i) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (aslot[i] >= 0) v = *reinterpret_cast<const float4*>(xn + (size_t)(unsigned)(aslot[i] + kc) * ESZ);
      areg[i] = v;
    }
    if constexpr (PRO) {
      if (p.pro_scale) {
#pragma unroll
        for (int q = 0; q < PV; ++q) {
          psc[q] = *reinterpret_cast<const float4*>(p.pro_scale + (size_t)n * p.pro_ld + kc + ac4 * EPS + q * 4);
          psh[q] = *reinterpret_cast<const float4*>(p.pro_shift + (size_t)n * p.pro_ld + kc + ac4 * EPS + q * 4);
        }
      }
    }
  };
  // fused prologue on the way into LDS: InstanceNorm affine + LeakyReLU; conv zero padding is
  // applied AFTER the transform (the reference pads the activated tensor)
  auto writeA = [&](bool raw) {
#pragma unroll
    for (int i = 0; i < NA4; ++i) {
      const int pix = (tid + i * NT) / GPR;
      const bool inb = aslot[i] >= 0;
      float4 v = areg[i];
      if constexpr (PRO) {
        const bool aff = !raw && p.pro_scale, lr = !raw && p.pro_lrelu;
        if constexpr (BF16) {
          if (aff || lr) {        // 8 packed bf16 channels: unpack, fp32 prologue, round back
            const uint32_t w[4] = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
            float e[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { e[2 * k] = h16_to_f32<ST>((uint16_t)(w[k] & 0xffffu)); e[2 * k + 1] = h16_to_f32<ST>((uint16_t)(w[k] >> 16)); }
            if (aff) {
              const float sc[8] = {psc[0].x, psc[0].y, psc[0].z, psc[0].w, psc[PV - 1].x, psc[PV - 1].y, psc[PV - 1].z, psc[PV - 1].w};
              const float sh[8] = {psh[0].x, psh[0].y, psh[0].z, psh[0].w, psh[PV - 1].x, psh[PV - 1].y, psh[PV - 1].z, psh[PV - 1].w};
#pragma unroll
              for (int k = 0; k < 8; ++k) e[k] = e[k] * sc[k] + sh[k];
            }
            if (lr) {
#pragma unroll
              for (int k = 0; k < 8; ++k) e[k] = lrelu(e[k]);
            }
            uint32_t o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (uint32_t)f32_to_h16<ST>(e[2 * k]) | ((uint32_t)f32_to_h16<ST>(e[2 * k + 1]) << 16);
            v = make_float4(__uint_as_float(o[0]), __uint_as_float(o[1]), __uint_as_float(o[2]), __uint_as_float(o[3]));
          }
        } else {
          if (aff) v = make_float4(v.x * psc[0].x + psh[0].x, v.y * psc[0].y + psh[0].y, v.z * psc[0].z + psh[0].z, v.w * psc[0].w + psh[0].w);
          if (lr) v = lrelu4(v);
        }
      }
      if (!inb) v = make_float4(0.f, 0.f, 0.f, 0.f);
      int lpix = pix;
      if constexpr (STRIDE == 2) { const int ly = pix / G::IW, lx = pix % G::IW; lpix = ly * G::IWP + (lx & 1) * G::IWH + (lx >> 1); }
      if (tid + i * NT < total4) *reinterpret_cast<float4*>(sA + lpix * G::CK + ac4 * 4) = v;
    }
  };

  // ---- consumer-side InstanceNorm finalize (see IgemmParams): called once, right after the first operand loads
  // have been issued, so that the partial-sum loads overlap them ----
  auto consumer_stats = [&]() {
    if constexpr (SPADE) {
      if (p.m_part) {
        // this workgroup modulates the virtual channels n0/2 .. n0/2 + BN/2 - 1; thread = (channel j, tile slice)
        constexpr int NCH = G::BN / 2, S = NT / NCH;
        const int j = tid % NCH, sl = tid / NCH;
        const int v = n0 / 2 + j;
        const bool vv = v < p.nsets * p.C;
        const int c = vv ? (v >= p.C ? v - p.C : v) : 0;
        double a1, a2;
        stats_from_partials(p.m_part + (size_t)n * p.m_tiles * 2 * p.m_Cs, p.m_tiles, p.m_Cs, c, sl, S, a1, a2);
        double* red = reinterpret_cast<double*>(smem);     // [S][NCH][2]; the main loop has not touched smem yet
        red[(sl * NCH + j) * 2] = a1; red[(sl * NCH + j) * 2 + 1] = a2;
        __syncthreads();
        if (sl == 0) {
          double t1 = 0.0, t2 = 0.0;
#pragma unroll
          for (int k = 0; k < S; ++k) { t1 += red[(k * NCH + j) * 2]; t2 += red[(k * NCH + j) * 2 + 1]; }
          float sc, sh;
          scale_shift_of(t1, t2, p.m_inv, 1.f, 0.f, sc, sh);
          s_stat[j] = sc; s_stat[NCH + j] = sh;
        }
        __syncthreads();
      }
    }
  };

  // one tap of one chunk: (BK/8) x {fragment reads, MF*NF*4 MFMAs} on the shifted LDS window
  auto compute_tap = [&](int dy, int dx, int buf, int ph = 0) {
    if constexpr (N16) {
      // v_mfma_f32_16x16x4_f32: lane l holds A[pixel l&15][k = l>>4] and B[k = l>>4][column l&15];
      // one float4 per lane (channels 4*(l>>4) .. +3 of a 16-channel step) feeds 4 MFMAs
      const int l15 = lane & 15, lq = lane >> 4;
      const float* sBb = sB + buf * G::SB + l15 * G::CK + lq * 4;
#pragma unroll
      for (int kb = 0; kb < BK / 16; ++kb) {
        float4 a[MF][2];
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
          for (int sub = 0; sub < 2; ++sub) {
            const int r = (wm * MF + mf) * 2 + sub + dy, c = l15 + dx;
            a[mf][sub] = *reinterpret_cast<const float4*>(sA + (r * G::IW + c) * G::CK + lq * 4 + kb * 16);
          }
        const float4 b = *reinterpret_cast<const float4*>(sBb + kb * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) {
              const float av = t == 0 ? a[mf][sub].x : t == 1 ? a[mf][sub].y : t == 2 ? a[mf][sub].z : a[mf][sub].w;
              const float bw = t == 0 ? b.x : t == 1 ? b.y : t == 2 ? b.z : b.w;
              acc16[mf][sub] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bw, acc16[mf][sub], 0, 0, 0);
            }
      }
      return;
    }
    int aoff[MF];   // LDS float offset of this lane's pixel in the (dy, dx) window
    int apix[MF];   // ... and its pixel index in the tile (the row of the XOR swizzle of a DMA-staged tile)
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      const int r = fy[mf] * STRIDE + dy;
      const int c = STRIDE == 2 ? (dx & 1) * G::IWH + fx + (dx >> 1) : fx * STRIDE + dx;   // stride 2: de-interleaved columns
      apix[mf] = r * G::IWP + c;
      aoff[mf] = apix[mf] * G::AP;
    }
    const int rb0 = wn * NFE * 32 + li;      // this lane's filter row of column fragment 0
    const float* sBrow = sB + buf * G::SB + rb0 * G::BP;
    static_assert(!BF16 || (BK % 16 == 0 && NF > 0), "bf16 matrix-core path: 16-channel steps, 32-column fragments");
    if constexpr (BF16) {
      {
        // v_mfma_f32_32x32x16_bf16: lane (row/col = l&31, half h = l>>5) holds k = 8h .. 8h+7 of a
        // 16-channel step: 16 contiguous bytes of the bf16 LDS row, one ds_read_b128 and no conversion
        constexpr int KBW16 = BK / 16 / KW;
#pragma unroll
        for (int kj = 0; kj < KBW16; ++kj) {
          const int kb = kw * KBW16 + kj;
          typedef typename std::conditional<ST == ST_F16, f16x8, bf16x8>::type op8;      // the same 16 bytes either way
          op8 a[MF], b[NFE];
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) a[mf] = *reinterpret_cast<const op8*>(sA + aoff[mf] + kb * 8 + lh * 4);
#pragma unroll
          for (int nf = 0; nf < NFE; ++nf) b[nf] = *reinterpret_cast<const op8*>(sBrow + nf * 32 * G::CK + kb * 8 + lh * 4);
#pragma unroll
          for (int mf = 0; mf < MF; ++mf)
#pragma unroll
            for (int nf = 0; nf < NFE; ++nf) {
              if constexpr (ST == ST_F16) acc[ph * MF + mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mf], b[nf], acc[ph * MF + mf][nf], 0, 0, 0);
              else acc[ph * MF + mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mf], b[nf], acc[ph * MF + mf][nf], 0, 0, 0);
            }
        }
        return;
      }
    }
    constexpr int KBW = BK / 8 / KW;          // 8-channel steps of one wave group
#pragma unroll
    for (int kj = 0; kj < KBW; ++kj) {
      const int kb = kw * KBW + kj;
      float4 a[MF], b[NFE];
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        if constexpr (DMA & 2) a[mf] = *reinterpret_cast<const float4*>(sA + aoff[mf] + G::swz(apix[mf], kb * 2 + lh) * 4);
        else a[mf] = *reinterpret_cast<const float4*>(sA + aoff[mf] + lh * 4 + kb * 8);
      }
#pragma unroll
      for (int nf = 0; nf < NFE; ++nf) {
        if constexpr (DMA & 1) b[nf] = *reinterpret_cast<const float4*>(sBrow + nf * 32 * G::BP + G::swz(rb0 + nf * 32, kb * 2 + lh) * 4);
        else b[nf] = *reinterpret_cast<const float4*>(sBrow + lh * 4 + nf * 32 * G::CK + kb * 8);
      }
      {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
          for (int nf = 0; nf < NFE; ++nf) {
            f32x16& d = acc[ph * MF + mf][nf];
            d = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mf].x, b[nf].x, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mf].y, b[nf].y, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mf].z, b[nf].z, d, 0, 0, 0);
            d = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mf].w, b[nf].w, d, 0, 0, 0);
          }
      }
    }
  };

  if constexpr (UPS && TB == 4) {
    // the four taps of a phase per barrier (4 barriers per chunk instead of 16)
    loadB3(kc_begin, 0);
    prefetchA(kc_begin);
    consumer_stats();
    int stage = 0;
    for (int kc = kc_begin; kc < kc_end; kc += BK) {
      __syncthreads();
      writeA(false);
#pragma unroll
      for (int ph = 0; ph < 4; ++ph, ++stage) {
        const int buf = stage & 1;
        storeB3(buf);
        if (ph < 3) loadB3(kc, ph + 1);
        else if (kc + BK < kc_end) loadB3(kc + BK, 0);
        if (ph == 0 && kc + BK < kc_end) prefetchA(kc + BK);
        __syncthreads();
#pragma unroll
        for (int t = 0; t < 4; ++t) compute_tap((ph >> 1) + (t >> 1), (ph & 1) + (t & 1), buf * 4 + t, ph);
      }
    }
  } else if constexpr (UPS) {
    // 16 (phase, tap) steps per chunk, fully unrolled so that the accumulator set is a compile-time choice
    loadB(kc_begin, 0);
    prefetchA(kc_begin);
    consumer_stats();
    for (int kc = kc_begin; kc < kc_end; kc += BK) {
      __syncthreads();
      writeA(false);
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int buf = t & 1;
        storeB(buf);
        if (t < 15) loadB(kc, t + 1);
        else if (kc + BK < kc_end) loadB(kc + BK, 0);
        if (t == 0 && kc + BK < kc_end) prefetchA(kc + BK);
        __syncthreads();
        const int ph = t >> 2;
        compute_tap((ph >> 1) + ((t >> 1) & 1), (ph & 1) + (t & 1), buf, ph);
      }
    }
  } else if constexpr (KS == 1 && TB == 2) {
    // 1x1 with input tile AND filter slice double-buffered: ONE barrier per channel chunk instead of two (a 1x1 chunk is
    // a single "tap" of 16-32 MFMAs per wave, so the barriers weigh far more than in a nine-tap 3x3 chunk): chunk k+1
    // is written to the other buffers right after the MFMAs of chunk k - its global loads were in flight during them
    loadB(kc_begin, 0);
    prefetchA(kc_begin);
    consumer_stats();
    writeA(false);
    storeB(0);
    int stage = 0;
    for (int kc = kc_begin; kc < kc_end; kc += BK, ++stage) {
      const int buf = stage & 1;
      const bool more = kc + BK < kc_end;
      if (more) { loadB(kc + BK, 0); prefetchA(kc + BK); }
      __syncthreads();
      sA = smem + buf * G::SA;
      compute_tap(0, 0, buf);
      if (more) { sA = smem + (buf ^ 1) * G::SA; writeA(false); storeB(buf ^ 1); }
    }
    sA = smem;
  } else if constexpr (TB == 9 && DM == 0) {
    // all nine filter slices of a chunk staged at once; input tile AND filters double-buffered across chunks, so a
    // chunk costs ONE barrier: chunk k+1 is written to the other buffers right after the MFMAs of chunk k (its
    // global loads were in flight during them); whoever is past the barrier of chunk k has finished chunk k-1
    loadB3(kc_begin, 0);
    prefetchA(kc_begin);
    consumer_stats();
    writeA(false);
    storeB3(0);
    int stage = 0;
    for (int kc = kc_begin; kc < kc_end; kc += BK, ++stage) {
      const int buf = stage & 1;
      const bool more = kc + BK < kc_end;
      if (more) { loadB3(kc + BK, 0); prefetchA(kc + BK); }
      __syncthreads();
      sA = smem + buf * G::SA;
#pragma unroll
      for (int t = 0; t < 9; ++t) compute_tap(t / 3, t % 3, buf * 9 + t);
      if (more) { sA = smem + (buf ^ 1) * G::SA; writeA(false); storeB3(buf ^ 1); }
    }
    sA = smem;
  } else if constexpr (TB == 3) {
    loadB3(kc_begin, 0);
    prefetchA(kc_begin);
    consumer_stats();
    int stage = 0;
    for (int kc = kc_begin; kc < kc_end; kc += BK) {
      __syncthreads();
      writeA(false);
#pragma unroll 1
      for (int dy = 0; dy < 3; ++dy, ++stage) {
        const int buf = stage & 1;
        storeB3(buf);
        {
          int ndy = dy + 1, nkc = kc;
          if (ndy == 3) { ndy = 0; nkc = kc + BK; }
          if (nkc < kc_end) loadB3(nkc, ndy);
        }
        if (dy == 0 && kc + BK < kc_end) prefetchA(kc + BK);
        __syncthreads();
        compute_tap(dy, 0, buf * 3 + 0);
        compute_tap(dy, 1, buf * 3 + 1);
        compute_tap(dy, 2, buf * 3 + 2);
      }
    }
  } else if constexpr (DM != 0) {
    // ---- operand tiles staged by LDS-DMA (see IgemmGeom).  ONE barrier per filter slice and no store phase: behind the
    // barrier of slice t every wave issues the fill of slice t + 1 into the other filter buffer (and, at slice 0, the fill of
    // the next chunk's input tile into the other tile buffer), then runs the MFMAs of slice t over them.  The DMA is inline
    // assembly: the compiler's wait-count pass cannot tell LDS buffers apart and would wait for every fill before every
    // ds_read.  Its own vmcnt waits (register-staged input tile of the prologue variants) stay correct: younger
    // operations in flight only make an in-order vmcnt(k) wait stricter.
    typedef __attribute__((address_space(3))) void lds_void;
    const uint32_t lds0 = (uint32_t)(size_t)(lds_void*)smem;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    constexpr int S = G::SLOTS;
    // filter slice: instruction qi = q * 4 + wave covers slots [qi * 64, + 64) of the [BN][S] slice; lane -> (row, physical slot)
    const char* bsrc[G::NQB > 0 ? G::NQB : 1];
#pragma unroll
    for (int q = 0; q < G::NQB; ++q) {
      const int L = (q * 4 + wave) * 64 + lane;
      const int row = L / S, ls = G::swz(row, L % S);
      bsrc[q] = wb + ((size_t)min(n0 + min(row, G::BN - 1), p.CoutPad - 1) * wrow + ls * 4) * 4;
    }
    // (m0 cannot be named in the clobber list: the AMDGPU backend treats it as a reserved register and rejects the clobber with
    // -Winline-asm "may not be preserved".  It writes m0 itself immediately in front of each of ITS m0 consumers - movrel, LDS-DMA
    // builtins, s_sendmsg - and these kernels contain none of those, so nothing of the compiler's is live in m0 across the asm.)
    auto fillB = [&](int buf, int kc, int tap) {
#pragma unroll
      for (int q = 0; q < G::NQB; ++q) {
        if ((q * 4 + wv) * 64 < G::BN * S) {         // (wave-uniform: whole instructions beyond the slice are not issued)
          const uint32_t dst = lds0 + (uint32_t)(G::NA * G::SA + buf * G::SB + (q * 4 + wv) * 256) * 4u;
          const char* src = bsrc[q] + (size_t)(tap * p.Cin + kc) * 4;
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory");
        }
      }
    };
    // input tile: the same split; a slot outside the tile, outside the image (zero padding) or on a pad column reads zeros
    const char* asrc[G::NQA > 0 ? G::NQA : 1];
    unsigned amask = 0;                     // bit q: slot q of this lane is an in-image element (its address advances with the chunk)
    if constexpr (DMA & 2) {
#pragma unroll
      for (int q = 0; q < G::NQA; ++q) {
        const int L = (q * 4 + wave) * 64 + lane;
        const int lp = L / S, ls = G::swz(lp, L % S);
        const int ly = lp / G::IWP, c = lp % G::IWP;
        int lx = c;
        bool ok = lp < G::IH * G::IWP;
        if constexpr (STRIDE == 2) {          // de-interleaved columns: [even columns | odd columns | pad]
          if (c < G::IWH) lx = 2 * c;
          else { lx = 2 * (c - G::IWH) + 1; ok = ok && lx < G::IW; }
        }
        const int iy = iy0 + ly, ix = ix0 + lx;
        ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        asrc[q] = ok ? xn + ((size_t)(unsigned)((iy * p.Win + ix) * p.xC) + ls * 4) * 4 : reinterpret_cast<const char*>(p.zeros);
        amask |= ok ? (1u << q) : 0u;
      }
    }
    auto fillA = [&](int abuf, int kc) {
#pragma unroll
      for (int q = 0; q < G::NQA; ++q) {
        const uint32_t dst = lds0 + (uint32_t)(abuf * G::SA + (q * 4 + wv) * 256) * 4u;
        const char* src = asrc[q] + ((amask >> q) & 1u ? (size_t)kc * 4 : 0);
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory");
      }
    };
    consumer_stats();                       // (uses smem as scratch: before the first fill)
    if constexpr (TB == 9) {
      // k_gemm_dma's cadence for a 3x3 convolution: a stage = the input tile + all nine filter slices of a channel chunk, two
      // stages, ONE barrier per chunk; behind it the fills of the next chunk are issued and the 9 x (BK / 8) MFMA steps of this
      // one run over them.  The register-staged TB = 9 variants paid 9 x NB4 staging registers for this cadence; these pay none.
      fillA(0, kc_begin);
#pragma unroll
      for (int t = 0; t < 9; ++t) fillB(t, kc_begin, t);
      int st = 0;
      for (int kc = kc_begin; kc < kc_end; kc += BK) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                      // the chunk has landed; everybody is done with the previous one (its stage is free)
        if (kc + BK < kc_end) {
          fillA(st ^ 1, kc + BK);
#pragma unroll
          for (int t = 0; t < 9; ++t) fillB((st ^ 1) * 9 + t, kc + BK, t);
        }
        sA = smem + st * G::SA;
#pragma unroll
        for (int t = 0; t < 9; ++t) compute_tap(t / 3, t % 3, st * 9 + t);
        st ^= 1;
      }
      sA = smem;
    } else {
    if constexpr (DMA & 2) fillA(0, kc_begin); else prefetchA(kc_begin);
    fillB(0, kc_begin, 0);
    int abuf = 0, stage = 0;               // stage: running slice count (the slice count of a chunk may be odd: 9, 1)
    for (int kc = kc_begin; kc < kc_end; kc += BK) {
      if constexpr (!(DMA & 2)) {
        __syncthreads();   // every wave is done reading sA of the previous chunk
        writeA(false);
      }
#pragma unroll 1
      for (int tap = 0; tap < G::TAPS; ++tap, ++stage) {
        const int buf = stage & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's part of slice `tap` (and, at slice 0, of the chunk's input tile) has landed
        __syncthreads();                                      // ... and everybody's; everybody is done with slice tap - 1
        {
          int ntap = tap + 1, nkc = kc;
          if (ntap == G::TAPS) { ntap = 0; nkc = kc + BK; }
          if (nkc < kc_end) fillB(buf ^ 1, nkc, ntap);
        }
        if (tap == 0 && kc + BK < kc_end) {
          if constexpr (DMA & 2) fillA(abuf ^ 1, kc + BK); else prefetchA(kc + BK);
        }
        if constexpr (DMA & 2) sA = smem + abuf * G::SA;
        compute_tap(tap / KS, tap % KS, buf);
      }
      abuf ^= 1;
    }
    sA = smem;
    }
  } else {
  loadB(kc_begin, 0);
  prefetchA(kc_begin);
  consumer_stats();
  for (int kc = kc_begin; kc < kc_end; kc += BK) {
    __syncthreads();   // every wave is done reading sA / sB of the previous chunk
    writeA(false);
#pragma unroll 1
    for (int tap = 0; tap < G::TAPS; ++tap) {
      const int buf = tap & 1;
      storeB(buf);
      {  // prefetch the next filter slice while this tap computes
        int ntap = tap + 1, nkc = kc;
        if (ntap == G::TAPS) { ntap = 0; nkc = kc + BK; }
        if (nkc < kc_end) loadB(nkc, ntap);
      }
      // next input chunk: issued AFTER the filter load so that the in-order vmcnt wait at the
      // next storeB does not have to cover it
      if (tap == 0 && kc + BK < kc_end) prefetchA(kc + BK);
      __syncthreads();
      compute_tap(tap / KS, tap % KS, buf);
    }
  }
  }

  // ---- fused 1x1 operand (learned shortcut): extra K chunks on the centre tap, last K slice only ----
  if constexpr (AUX && KS == 3 && STRIDE == 1 && !UPS && !SPADE) if (p.x2 != nullptr && split == p.ksplit - 1) {
    const char* x2n = reinterpret_cast<const char*>(p.x2) + (size_t)n * p.Hin * p.Win * p.x2C * ESZ;
    const char* w2b = reinterpret_cast<const char*>(p.w2);
    auto loadB2 = [&](int kc) {
#pragma unroll
      for (int i = 0; i < G::NB4; ++i) {
        const int idx = tid + i * NT;
        const int row = idx / GPRB, c4 = idx % GPRB;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < G::BN && n0 + row < p.CoutPad) {
          const size_t off = ((size_t)(n0 + row) * p.Cin2 + kc + c4 * EPB) * WSZ;      // w2 [CoutPad][Cin2]
          v = *reinterpret_cast<const float4*>(w2b + off);
        }
        breg[i] = v;
      }
    };
    auto prefetchA2 = [&](int kc) {
#pragma unroll
      for (int i = 0; i < NA4; ++i) {
        int pix, iy, ix;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (slot_inb(i, pix, iy, ix))
          v = *reinterpret_cast<const float4*>(x2n + (size_t)(unsigned)((iy * p.Win + ix) * p.x2C + kc + ac4 * EPS) * ESZ);
        areg[i] = v;
      }
    };
    loadB2(0);
    prefetchA2(0);
    for (int kc = 0; kc < p.Cin2; kc += BK) {
      __syncthreads();
      writeA(true);
      storeB(0);
      if (kc + BK < p.Cin2) { loadB2(kc + BK); prefetchA2(kc + BK); }
      __syncthreads();
      compute_tap(1, 1, 0);
    }
  }

  if constexpr (KW > 1) {
    // sum the wave groups' partial accumulators through LDS, one group per round (16 KB per fragment);
    // groups 1.. are done afterwards (finished waves do not take part in later barriers)
    float* rb = smem + (size_t)wave * (MF * NFE * 16 * 64) + lane;
#pragma unroll 1
    for (int g = 1; g < KW; ++g) {
      __syncthreads();
      if (kw == g) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
          for (int nf = 0; nf < NFE; ++nf)
#pragma unroll
            for (int r = 0; r < 16; ++r) rb[((mf * NFE + nf) * 16 + r) * 64] = acc[mf][nf][r];
      }
      __syncthreads();
      if (kw == 0) {
#pragma unroll
        for (int mf = 0; mf < MF; ++mf)
#pragma unroll
          for (int nf = 0; nf < NFE; ++nf)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mf][nf][r] += rb[((mf * NFE + nf) * 16 + r) * 64];
      }
    }
    if (kw != 0) return;
  }

  // ------------------------------------ epilogue ------------------------------------
  // accumulator element r of lane l: row = (r&3) + 8*(r>>2) + 4*(l>>5)  (pixel), col = l&31 (channel)
  if constexpr (N16) {
    // accumulator element r of lane l: pixel x = (l>>4)*4 + r of tile row (block, sub), column l&15
    const int l15 = lane & 15, lq = lane >> 4;
    const int col = n0 + l15;
    const bool cvalid = col < p.Cout;
    const float bv = p.bias[min(col, p.CoutPad - 1)];
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf)
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int oy = ty0 + (wm * MF + mf) * 2 + sub;
          const int ox = tx0 + lq * 4 + r;
          const float v = apply_act(acc16[mf][sub][r] + bv, p.act);
          if (cvalid && oy < p.Hout && ox < p.Wout) {
            const size_t pix = ((size_t)n * p.Hout + oy) * p.Wout + ox;
            p.y[pix * p.yC + p.yoff + col] = v;
            if (p.y_nchw) p.y_nchw[(((size_t)n * p.Cout + col) * p.Hout + oy) * p.Wout + ox] = v;
            s1 += (double)v;
            s2 += (double)v * (double)v;
          }
        }
    if (p.stat_part) {
      __syncthreads();
      double* red = reinterpret_cast<double*>(smem);   // [WM][16][2]
      double a1 = s1 + __shfl_xor(s1, 16); a1 += __shfl_xor(a1, 32);
      double a2 = s2 + __shfl_xor(s2, 16); a2 += __shfl_xor(a2, 32);
      if (lane < 16) { red[(wm * 16 + l15) * 2] = a1; red[(wm * 16 + l15) * 2 + 1] = a2; }
      __syncthreads();
      if (tid < 16) {
        double b1 = 0.0, b2 = 0.0;
#pragma unroll
        for (int m = 0; m < WM; ++m) { b1 += red[(m * 16 + tid) * 2]; b2 += red[(m * 16 + tid) * 2 + 1]; }
        double* dst = p.stat_part + (((size_t)n * (p.tilesX * p.tilesY) + tile) * 2) * p.CoutPad;
        dst[n0 + tid] = b1;
        dst[p.CoutPad + n0 + tid] = b2;
      }
    }
  } else if (!SPADE && p.slab != nullptr) {
    // split-K: raw partial sums to slab [split][B][Hout][Wout][CoutPad]; k_splitk_epilogue finishes
    float* slab = p.slab + ((size_t)split * gridDim.z / p.ksplit + n) * p.Hout * p.Wout * p.CoutPad;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      const int col = n0 + (wn * NF + nf) * 32 + li;
#pragma unroll
      for (int ph = 0; ph < PH; ++ph)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
          int oy = ty0 + (wm * MF + mf) * G::FRH + row / FRW;
          int ox = tx0 + row % FRW;
          if (UPS) { oy = 2 * oy + (ph >> 1); ox = 2 * ox + (ph & 1); }
          if (col < p.CoutPad && oy < p.Hout && ox < p.Wout)
            slab[((size_t)oy * p.Wout + ox) * p.CoutPad + col] = acc[ph * MF + mf][nf][r];
        }
      }
    }
  } else if (!SPADE) {
    // statistics partials: fp64 from the fragment level on (see IgemmParams: no fp32 cancellation in E[x^2] - mean^2).
    // ONE accumulator pair is live at a time - a column fragment's sums go to LDS before the next one starts: a pair
    // per fragment cost 4 more registers in the NF = 2 variants, i.e. 4 -> 3 waves per SIMD (96 + 32 -> 99 + 32)
    double* red = reinterpret_cast<double*>(smem);   // [WM][BN][2]
    if (p.stat_part) __syncthreads();                // all waves finished the main loop: smem can be reused
    const float* pbias = p.bias + (p.w_mod ? (size_t)(n % p.w_mod) * p.b_stride : 0);
    const int ny = p.pair ? (n >> 1) : n;            // output image and channel offset of this sample
    const int yoff = p.pair ? p.yoff + (n & 1) * p.pair_yoff : p.yoff;
#pragma unroll
    for (int nf = 0; nf < NF; ++nf) {
      // sum(x), sum(x^2) of this lane's valid elements without cancellation at fp32 cost: deviations from a pivot (the
      // lane's first value) are summed in fp32 - they are of the size of the spread, not of the mean - and the pivot is
      // put back in fp64 once per column fragment:  sum x = k*p + sum d,  sum x^2 = k*p^2 + 2p*sum d + sum d^2
      // (three live values per column fragment - round 1's fp32 sums had four across NF = 2 fragments and the NF = 2
      // variants sit exactly on the 96 + 32 register boundary of 4 waves per SIMD; the element count is recomputed)
      float pv = 0.f, d1 = 0.f, d2 = 0.f;
      const int col = n0 + (wn * NF + nf) * 32 + li;
      const bool cvalid = col < p.Cout;
      const float bv = (col < p.CoutPad) ? pbias[col] : 0.f;
#pragma unroll
      for (int ph = 0; ph < PH; ++ph)
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        // the 16 residual reads of a fragment are issued as one batch from clamped (always valid) addresses: inside
        // the bounds-checked store loop below each read sat behind the previous element's store (the compiler
        // cannot reorder a load over a possibly aliasing store): 16 serialised memory round trips per fragment
        // Element r of the fragment is pixel (oy0 + RY(r), ox0 + RX(r)) with compile-time RY / RX (FRW >= 8: the lane's 4 * lh
        // never carries into the row), oy0 wave-uniform and ox0 = tx0 + 4 * lh.  Where the fragment lies inside the image
        // (`full`: every launch of the frame except ragged sizes) every address below is ONE per-lane base plus a scalar offset
        // per element - the generic form spent ~10 vector instructions per element, quarter-rate 32-bit multiplies among them,
        // on ((n * H + oy) * W + ox) * C: a third of the kernel on the small-K layers (ISA count, DESIGN round 3).
        const int oy0 = ty0 + (__builtin_amdgcn_readfirstlane(wm) * MF + mf) * G::FRH, ox0 = tx0 + 4 * lh;
        const bool full = !UPS && oy0 + G::FRH <= p.Hout && tx0 + FRW <= p.Wout;      // (uniform)
        float rv[16];
        if (p.res && full) {
          const int sh = p.res_ups ? 1 : 0;
          const size_t rrow = (size_t)(p.Wout >> sh) * p.resC;
          const size_t rbase = (((size_t)n * (p.Hout >> sh) + (oy0 >> sh)) * (p.Wout >> sh) + (ox0 >> sh)) * p.resC + min(col, p.resC - 1);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int RY = (8 * (r >> 2)) / FRW, RX = (8 * (r >> 2)) % FRW + (r & 3);
            rv[r] = ld_act<ST>(p.res, rbase + (size_t)(RY >> sh) * rrow + (size_t)(RX >> sh) * p.resC);
          }
        } else if (p.res) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            int oy = ty0 + (wm * MF + mf) * G::FRH + row / FRW;
            int ox = tx0 + row % FRW;
            if (UPS) { oy = 2 * oy + (ph >> 1); ox = 2 * ox + (ph & 1); }
            oy = min(oy, p.Hout - 1); ox = min(ox, p.Wout - 1);
            const size_t rpix = p.res_ups ? ((size_t)n * (p.Hout >> 1) + (oy >> 1)) * (p.Wout >> 1) + (ox >> 1)
                                          : ((size_t)n * p.Hout + oy) * p.Wout + ox;
            rv[r] = ld_act<ST>(p.res, rpix * p.resC + min(col, p.resC - 1));
          }
        }
        // values first (no memory operations, the activation chosen once per fragment), then the stores: with
        // loads, uniform branches and stores interleaved per element the compiler waited for vmcnt(0) - i.e. for
        // the previous element's STORE to be acknowledged - before every element (16 serialised round trips)
        float vv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) vv[r] = acc[ph * MF + mf][nf][r] + bv;
        if (p.res) {
#pragma unroll
          for (int r = 0; r < 16; ++r) vv[r] += rv[r];
        }
        if (p.act == ACT_LRELU) {
#pragma unroll
          for (int r = 0; r < 16; ++r) vv[r] = lrelu(vv[r]);
        } else if (p.act == ACT_TANH) {
#pragma unroll
          for (int r = 0; r < 16; ++r) vv[r] = tanhf(vv[r]);
        } else if (p.act == ACT_SIGMOID) {
#pragma unroll
          for (int r = 0; r < 16; ++r) vv[r] = 1.f / (1.f + __expf(-vv[r]));
        }
        float vf[16];       // unrounded values for an fp32 side copy (the image head's NCHW output)
        if constexpr (BF16) {
          if (!p.y_f32) {   // the stored tensor is bf16: the statistics describe what the consumer will read
#pragma unroll
            for (int r = 0; r < 16; ++r) { vf[r] = vv[r]; vv[r] = round16<ST>(vv[r]); }
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) vf[r] = vv[r];
          }
        }
        unsigned okm = 0;
        if (full) okm = cvalid ? 0xffffu : 0u;
        else {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            int oy = ty0 + (wm * MF + mf) * G::FRH + row / FRW;
            int ox = tx0 + row % FRW;
            if (UPS) { oy = 2 * oy + (ph >> 1); ox = 2 * ox + (ph & 1); }   // phase ph of source pixel (oy, ox)
            const bool ok = cvalid && oy < p.Hout && ox < p.Wout;
            okm |= ok ? (1u << r) : 0u;
          }
        }
        if (ph == 0 && mf == 0) pv = vv[0];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = (okm & (1u << r)) ? vv[r] - pv : 0.f;
          d1 += d;
          d2 += d * d;
        }
        if (full) {
          if (cvalid) {
            const size_t yrow = (size_t)p.Wout * p.yC;
            const size_t ybase = (((size_t)ny * p.Hout + oy0) * p.Wout + ox0) * p.yC + yoff + col;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int RY = (8 * (r >> 2)) / FRW, RX = (8 * (r >> 2)) % FRW + (r & 3);
              const size_t yi = ybase + (size_t)RY * yrow + (size_t)RX * p.yC;
              if constexpr (BF16) { if (p.y_f32) p.y[yi] = vv[r]; else st_act<ST>(p.y, yi, vv[r]); }
              else p.y[yi] = vv[r];
            }
            if (p.y_nchw) {
              const size_t nbase = (((size_t)n * p.Cout + col) * p.Hout + oy0) * p.Wout + ox0;
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int RY = (8 * (r >> 2)) / FRW, RX = (8 * (r >> 2)) % FRW + (r & 3);
                p.y_nchw[nbase + (size_t)RY * p.Wout + RX] = BF16 ? vf[r] : vv[r];
              }
            }
          }
        } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
          int oy = ty0 + (wm * MF + mf) * G::FRH + row / FRW;
          int ox = tx0 + row % FRW;
          if (UPS) { oy = 2 * oy + (ph >> 1); ox = 2 * ox + (ph & 1); }
          if (okm & (1u << r)) {
            const size_t yi = (((size_t)ny * p.Hout + oy) * p.Wout + ox) * p.yC + yoff + col;
            if constexpr (BF16) { if (p.y_f32) p.y[yi] = vv[r]; else st_act<ST>(p.y, yi, vv[r]); }
            else p.y[yi] = vv[r];
          }
        }
        if (p.y_nchw) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
            int oy = ty0 + (wm * MF + mf) * G::FRH + row / FRW;
            int ox = tx0 + row % FRW;
            if (UPS) { oy = 2 * oy + (ph >> 1); ox = 2 * ox + (ph & 1); }
            if (okm & (1u << r)) p.y_nchw[(((size_t)n * p.Cout + col) * p.Hout + oy) * p.Wout + ox] = BF16 ? vf[r] : vv[r];
          }
        }
        }
      }
      if (p.stat_part) {
        int cnt = 0;      // valid elements of this lane in this column fragment (coordinates only: nothing kept live for it)
#pragma unroll
        for (int ph = 0; ph < PH; ++ph)
#pragma unroll
          for (int mf = 0; mf < MF; ++mf) {
            const int oy0 = ty0 + (__builtin_amdgcn_readfirstlane(wm) * MF + mf) * G::FRH;
            if (!UPS && oy0 + G::FRH <= p.Hout && tx0 + FRW <= p.Wout) { cnt += cvalid ? 16 : 0; continue; }      // (a full fragment)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
              int oy = ty0 + (wm * MF + mf) * G::FRH + row / FRW;
              int ox = tx0 + row % FRW;
              if (UPS) { oy = 2 * oy + (ph >> 1); ox = 2 * ox + (ph & 1); }
              cnt += (cvalid && oy < p.Hout && ox < p.Wout) ? 1 : 0;
            }
          }
        const double k = (double)cnt, pd = (double)pv;
        const double s1 = k * pd + (double)d1;
        const double s2 = (k * pd + 2.0 * (double)d1) * pd + (double)d2;
        const double a1 = s1 + __shfl_xor(s1, 32);
        const double a2 = s2 + __shfl_xor(s2, 32);
        if (lh == 0) {
          const int c = (wn * NF + nf) * 32 + li;
          red[(wm * G::BN + c) * 2 + 0] = a1;
          red[(wm * G::BN + c) * 2 + 1] = a2;
        }
      }
    }
    if (p.stat_part) {   // deterministic per-tile partial sums for the following InstanceNorm
      __syncthreads();
      for (int c = tid; c < G::BN; c += 256) {
        double a1 = 0.0, a2 = 0.0;
#pragma unroll
        for (int m = 0; m < WM; ++m) { a1 += red[(m * G::BN + c) * 2]; a2 += red[(m * G::BN + c) * 2 + 1]; }
        const int col = n0 + c;
        if (col < p.CoutPad) {
          double* dst = p.stat_part + (((size_t)n * (p.tilesX * p.tilesY) + tile) * 2) * p.CoutPad;
          dst[col] = a1;
          dst[p.CoutPad + col] = a2;
        }
      }
    }
  } else if constexpr (NF == 1) {
    // ONE fragment per wave: columns [gamma(16) | beta(16)] of 16 consecutive virtual channels (layers that modulate 16
    // channels in all: the pair layout would multiply two half-empty fragments).  A lane holds gamma (li < 16) or beta
    // (li >= 16) of channel li % 16 for the fragment's 16 rows; the halves exchange them with one shuffle per row and then
    // each finishes 8 rows: gamma lanes rows 0-7, beta lanes rows 8-15.
    const int hb = li >> 4, c16 = li & 15;
    const int colg = n0 + wn * 32 + c16;                        // gamma column in w / bias; beta = + 16
    const int v = (n0 / 2) + wn * 16 + c16;                     // virtual channel
    const bool vvalid = v < p.nsets * p.C;
    const int set = (vvalid && v >= p.C) ? 1 : 0;
    const int c = v - set * p.C;
    float bg = 0.f, bb = 0.f, sc = 0.f, sh = 0.f;
    if (vvalid) {
      bg = p.bias[colg]; bb = p.bias[colg + 16];
      if (p.m_part) { sc = s_stat[v - n0 / 2]; sh = s_stat[G::BN / 2 + v - n0 / 2]; }
      else { sc = p.m_scale[(size_t)n * p.m_ld + c]; sh = p.m_shift[(size_t)n * p.m_ld + c]; }
    }
    float* yout = set ? p.ys1 : p.ys0;
    const int act = set ? p.act1 : p.act0;
    const int Hm = p.xm_ups ? p.Hout / 2 : p.Hout, Wm = p.xm_ups ? p.Wout / 2 : p.Wout;
#pragma unroll
    for (int mf = 0; mf < MF; ++mf) {
      // (a fragment inside the image: one per-lane base + a scalar offset per element, as in the convolution epilogue; this
      // lane's elements are r = 8 hb + k: pixel row oy0 + (16 hb) / FRW + RY(k), column ox0 + RX(k))
      const int oy0 = ty0 + (__builtin_amdgcn_readfirstlane(wm) * MF + mf) * G::FRH;
      const bool full = oy0 + G::FRH <= p.Hout && tx0 + FRW <= p.Wout;      // (uniform)
      const int oyl = oy0 + (16 * hb) / FRW, oxl = tx0 + 4 * lh;
      const int s = p.xm_ups ? 1 : 0;
      const size_t xrow = (size_t)Wm * p.xmC, yrow = (size_t)p.Wout * p.C;
      const size_t xbase = (((size_t)n * Hm + (oyl >> s)) * Wm + (oxl >> s)) * p.xmC + (vvalid ? c : 0);
      const size_t ybase = (((size_t)n * p.Hout + oyl) * p.Wout + oxl) * p.C + c;
      float xr[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        if (full) {
          const int RY = (8 * (k >> 2)) / FRW, RX = (8 * (k >> 2)) % FRW + (k & 3);
          xr[k] = ld_act<ST>(p.xm, xbase + (size_t)(RY >> s) * xrow + (size_t)(RX >> s) * p.xmC);
        } else {
          const int r = hb * 8 + k;
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int oy = min(ty0 + (wm * MF + mf) * G::FRH + row / FRW, p.Hout - 1);
          const int ox = min(tx0 + row % FRW, p.Wout - 1);
          const int sy = p.xm_ups ? (oy >> 1) : oy, sx = p.xm_ups ? (ox >> 1) : ox;
          xr[k] = ld_act<ST>(p.xm, (((size_t)n * Hm + sy) * Wm + sx) * p.xmC + (vvalid ? c : 0));
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float lo = acc[mf][0][k], hi = acc[mf][0][8 + k];
        const float olo = __shfl_xor(lo, 16), ohi = __shfl_xor(hi, 16);
        const float gamma = (hb ? ohi : lo) + bg;               // gamma lanes own row k, beta lanes fetch row 8 + k's gamma
        const float beta = (hb ? hi : olo) + bb;
        float o = (xr[k] * sc + sh) * (1.f + gamma) + beta;
        o = apply_act(o, act);
        if (full) {
          const int RY = (8 * (k >> 2)) / FRW, RX = (8 * (k >> 2)) % FRW + (k & 3);
          if (vvalid) st_act<ST>(yout, ybase + (size_t)RY * yrow + (size_t)RX * p.C, o);
        } else {
          const int r = hb * 8 + k;
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int oy = ty0 + (wm * MF + mf) * G::FRH + row / FRW;
          const int ox = tx0 + row % FRW;
          if (vvalid && oy < p.Hout && ox < p.Wout) st_act<ST>(yout, (((size_t)n * p.Hout + oy) * p.Wout + ox) * p.C + c, o);
        }
      }
    }
  } else {
    // fragment pair (2q, 2q+1) = (gamma, beta) of 32 consecutive virtual channels
#pragma unroll
    for (int q = 0; q < NF / 2; ++q) {
      const int colg = n0 + (wn * NF + 2 * q) * 32 + li;       // gamma column in w / bias
      const int v = (n0 / 2) + (wn * NF / 2 + q) * 32 + li;    // virtual channel
      const bool vvalid = v < p.nsets * p.C;
      const int set = (vvalid && v >= p.C) ? 1 : 0;
      const int c = v - set * p.C;
      float bg = 0.f, bb = 0.f, sc = 0.f, sh = 0.f;
      if (vvalid) {
        bg = p.bias[colg]; bb = p.bias[colg + 32];
        if (p.m_part) { sc = s_stat[v - n0 / 2]; sh = s_stat[G::BN / 2 + v - n0 / 2]; }   // consumer-side finalize
        else { sc = p.m_scale[(size_t)n * p.m_ld + c]; sh = p.m_shift[(size_t)n * p.m_ld + c]; }
      }
      float* yout = set ? p.ys1 : p.ys0;
      const int act = set ? p.act1 : p.act0;
      const int Hm = p.xm_ups ? p.Hout / 2 : p.Hout, Wm = p.xm_ups ? p.Wout / 2 : p.Wout;
#pragma unroll
      for (int mf = 0; mf < MF; ++mf) {
        // all 16 reads of the normalised tensor first (clamped, always valid addresses), then the modulation and
        // the stores: one memory round trip per fragment instead of 16 serialised load -> store pairs
        // (a fragment inside the image: one per-lane base + a scalar offset per element, as in the convolution epilogue)
        const int oy0 = ty0 + (__builtin_amdgcn_readfirstlane(wm) * MF + mf) * G::FRH, ox0 = tx0 + 4 * lh;
        const bool full = oy0 + G::FRH <= p.Hout && tx0 + FRW <= p.Wout;      // (uniform)
        float xr[16];
        if (full) {
          const int s = p.xm_ups ? 1 : 0;
          const size_t xrow = (size_t)Wm * p.xmC;
          const size_t xbase = (((size_t)n * Hm + (oy0 >> s)) * Wm + (ox0 >> s)) * p.xmC + (vvalid ? c : 0);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int RY = (8 * (r >> 2)) / FRW, RX = (8 * (r >> 2)) % FRW + (r & 3);
            xr[r] = ld_act<ST>(p.xm, xbase + (size_t)(RY >> s) * xrow + (size_t)(RX >> s) * p.xmC);
          }
          if (vvalid) {
            const size_t yrow = (size_t)p.Wout * p.C;
            const size_t ybase = (((size_t)n * p.Hout + oy0) * p.Wout + ox0) * p.C + c;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int RY = (8 * (r >> 2)) / FRW, RX = (8 * (r >> 2)) % FRW + (r & 3);
              const float gamma = acc[mf][2 * q][r] + bg;
              const float beta = acc[mf][2 * q + 1][r] + bb;
              float o = (xr[r] * sc + sh) * (1.f + gamma) + beta;
              o = apply_act(o, act);
              st_act<ST>(yout, ybase + (size_t)RY * yrow + (size_t)RX * p.C, o);
            }
          }
        } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int oy = min(ty0 + (wm * MF + mf) * G::FRH + row / FRW, p.Hout - 1);
          const int ox = min(tx0 + row % FRW, p.Wout - 1);
          const int sy = p.xm_ups ? (oy >> 1) : oy, sx = p.xm_ups ? (ox >> 1) : ox;
          xr[r] = ld_act<ST>(p.xm, (((size_t)n * Hm + sy) * Wm + sx) * p.xmC + (vvalid ? c : 0));
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int oy = ty0 + (wm * MF + mf) * G::FRH + row / FRW;
          const int ox = tx0 + row % FRW;
          if (vvalid && oy < p.Hout && ox < p.Wout) {
            const float xv = xr[r];
            const float gamma = acc[mf][2 * q][r] + bg;
            const float beta = acc[mf][2 * q + 1][r] + bb;
            float o = (xv * sc + sh) * (1.f + gamma) + beta;
            o = apply_act(o, act);
            st_act<ST>(yout, (((size_t)n * p.Hout + oy) * p.Wout + ox) * p.C + c, o);
          }
        }
        }
      }
    }
  }
}

}  // namespace rib
