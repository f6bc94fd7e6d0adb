// rib.hip — runtime and C ABI (include/rib.h) of the MI355X-native generator.
//
// Host-side responsibilities (all native C++; Python only moves pointers):
//   * the layer inventory of the generator derived from rib_config
//     (reference: PGNR/models/generator.py:43-178,315-358,423-491);
//   * checkpoint ingestion in the reference's state-dict vocabulary, eval-mode spectral-norm
//     fold (PGNR/models/layers/weight_norm.py:84-85 hook), filter re-layout, one device blob;
//   * a static launch plan per (B,H,W): workspace layout + the ordered kernel launches that
//     restate Generator.forward (PGNR/models/generator.py:181-234) and MaskGenerator.forward
//     (:493-510);
//   * the autoregressive segment driver (PGNR/models/evaluator.py:238-262).
#include "kernels.hip.h"
#include "raster.hip.h"
#include <hip/hip_ext.h>
#include "../../include/rib.h"

#include <algorithm>
#include <array>
#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <vector>


// ---- the k_igemm instantiations live in the igemm_shard_<s>.o objects (variants.def, build.py) ----
#include "variants.hip.h"
#define RIB_V(sec, ...) RIB_I_V(RIB_F_EXTERN, __VA_ARGS__)
#define RIB_VK(sec, ...) RIB_I_VK(RIB_F_EXTERN, __VA_ARGS__)
#define RIB_VT(sec, ...) RIB_I_VT(RIB_F_EXTERN, __VA_ARGS__)
#define RIB_VTK(sec, ...) RIB_I_VTK(RIB_F_EXTERN, __VA_ARGS__)
#define RIB_V9(sec, ...) RIB_I_V9(RIB_F_EXTERN, __VA_ARGS__)
#define RIB_VU4(sec, ...) RIB_I_VU4(RIB_F_EXTERN, __VA_ARGS__)
#define RIB_VS(sec, ...) RIB_I_VS(RIB_F_EXTERN, __VA_ARGS__)
#define RIB_VSK(sec, ...) RIB_I_VSK(RIB_F_EXTERN, __VA_ARGS__)
#define RIB_VB(sec, ...) RIB_I_VB(RIB_F_EXTERN, __VA_ARGS__)
#define RIB_VBX(sec, ...) RIB_I_VBX(RIB_F_EXTERN, __VA_ARGS__)
#define RIB_V1D(sec, ...) RIB_I_V1D(RIB_F_EXTERN, __VA_ARGS__)
#define RIB_VS1D(sec, ...) RIB_I_VS1D(RIB_F_EXTERN, __VA_ARGS__)
#define RIB_VD(sec, ...) RIB_I_VD(RIB_F_EXTERN, __VA_ARGS__)
#define RIB_VSD(sec, ...) RIB_I_VSD(RIB_F_EXTERN, __VA_ARGS__)
#define RIB_VD9(sec, ...) RIB_I_VD9(RIB_F_EXTERN, __VA_ARGS__)
#include "variants.def"
#undef RIB_V
#undef RIB_VK
#undef RIB_VT
#undef RIB_VTK
#undef RIB_V9
#undef RIB_VU4
#undef RIB_VS
#undef RIB_VSK
#undef RIB_VB
#undef RIB_VBX
#undef RIB_V1D
#undef RIB_VS1D
#undef RIB_VD
#undef RIB_VSD
#undef RIB_VD9

using namespace rib;

// Every kernel launch of this file goes through RIB_KLAUNCH.  Outside a kernel-time profiling pass it is hipLaunchKernelGGL; inside
// one (rib_profile_begin_kernels) the launch carries a start and a stop event that the runtime binds to the dispatch itself
// (hipExtLaunchKernelGGL): their difference is the kernel's own execution time from the queue's timestamps - what rocprofv3's
// kernel trace reports - with no event packet between two launches.
namespace { struct ProfPair { hipEvent_t start = nullptr, stop = nullptr; }; thread_local ProfPair g_prof_pair; }
#define RIB_KLAUNCH(KERNEL, grid, block, lds, st, ...)                                                                       \
  do {                                                                                                                       \
    if (g_prof_pair.start) hipExtLaunchKernelGGL(KERNEL, grid, block, lds, st, g_prof_pair.start, g_prof_pair.stop, 0, __VA_ARGS__); \
    else hipLaunchKernelGGL(KERNEL, grid, block, lds, st, __VA_ARGS__);                                                      \
  } while (0)

#if !defined(RIB_BUILD_STAMP) || !defined(RIB_SHARD_STAMP)
#error "compile through csrc/build.py (-DRIB_BUILD_STAMP / -DRIB_SHARD_STAMP: content hashes of the sources, see build.py)"
#endif
// the stamp strings of the RIB_NSECTIONS k_igemm shard objects (igemm_shard.hip) and this object's own
extern "C" {
#define RIB_X(s) extern const char rib_stamp_section_##s[];
RIB_FOR_SECTIONS(RIB_X)
#undef RIB_X
}
extern "C" __attribute__((used, visibility("hidden"))) const char kLibStamp[] = "rib-stamp lib " RIB_BUILD_STAMP;

namespace {

thread_local std::string g_create_error;

std::string fmt(const char* f, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, f);
  vsnprintf(buf, sizeof buf, f, ap);
  va_end(ap);
  return buf;
}

// channel padding of an activation: 8, 16, or a multiple of 32 (so that every tensor admits the
// 16- or 32-channel K chunks of the fast kernel variants; the 22-channel label map becomes 32)
inline int pad8(int c) { return c <= 8 ? 8 : (c <= 16 ? 16 : (c + 31) / 32 * 32); }
// bf16 storage: 16-channel K steps (v_mfma_f32_32x32x16_bf16), so the smallest activation is 16 channels wide
inline int pad16(int c) { return c <= 16 ? 16 : (c + 31) / 32 * 32; }
// float -> bf16, round to nearest even (what v_cvt_pk_bf16_f32 does on the device)
inline uint16_t host_bf16(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
// float -> IEEE half, round to nearest even (what v_cvt_f16_f32 does on the device), subnormals and overflow to inf included
inline uint16_t host_f16(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  const uint32_t a = u & 0x7fffffffu;
  if (a > 0x7f800000u) return (uint16_t)(sign | 0x7e00u);                      // NaN
  if (a >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);                     // >= 65520 rounds to inf
  if (a < 0x33000001u) return (uint16_t)sign;                                  // <= 2^-25 rounds to zero
  int e = (int)(a >> 23) - 127;
  uint32_t m = (a & 0x7fffffu) | 0x800000u;                                    // 24-bit significand
  int shift = e < -14 ? 13 + (-14 - e) : 13;                                   // bits dropped (subnormal: more)
  uint32_t half_m = m >> shift;
  const uint32_t rem = m & ((1u << shift) - 1), halfway = 1u << (shift - 1);
  if (rem > halfway || (rem == halfway && (half_m & 1u))) half_m += 1;
  uint32_t h16;
  if (e < -14) h16 = half_m;                                                   // subnormal (a carry into 0x400 is the smallest normal: correct)
  else h16 = ((uint32_t)(e + 15) << 10) + (half_m - 0x400u);                   // a mantissa carry bumps the exponent: correct
  return (uint16_t)(sign | h16);
}
inline int pad32(int c) { return (c + 31) / 32 * 32; }
inline int pick_bk(int cp) { return cp % 32 == 0 ? 32 : (cp % 16 == 0 ? 16 : 8); }
inline size_t align256(size_t b) { return (b + 255) / 256 * 256; }

// ------------------------------------------------------------------------------------------
// layer inventory (mirrors render_in_between_amd/spec.py; pinned by the state-dict key test)
// ------------------------------------------------------------------------------------------
struct ConvDef {
  std::string name;
  int cin = 0, cout = 0, ks = 3, stride = 1;
  bool spectral = true;
  int spade_cond = 0;
  bool in_affine = false;
  bool used = true;
  // device blob offsets (floats), filled by finalize
  size_t w_off = 0, b_off = 0, g_off = 0, be_off = 0;
  size_t fb_off = 0;   // conv_block_1 of a learned-shortcut SPADE block: bias + shortcut bias (fused launch)
  // convolution of a nearest-x2-upsampled tensor (mask network, generator.py:480): the launch runs on the
  // four 2x2 phase filters [CoutPad][16 = phase*4 + a*2 + b][CinPad] at wp_off (sums of the taps at w_off)
  bool ups_in = false;
  size_t wp_off = 0;
  int cinp = 0, coutp = 0;
  // bf16 storage mode: bf16 copies of the filters (same [CoutPad][tap][CinPad] layout), offsets in floats of the blob
  size_t w16_off = 0, wp16_off = 0;
  // Winograd F(2x2, 3x3) filters U = G g G^T, [16 positions][CoutPad][CinPad] fp32, and a zero bias vector for the
  // batched GEMM (the real bias is added by the output transform); 0: the layer never runs that way
  // Winograd: the layer may run in the Winograd domain (wino_ok); the transformed filter sets U = G g G^T live OUTSIDE the
  // blob, made on the device from the folded filters when a plan first needs one (WinoSet, ensure_wino_set); zero_off: a
  // zero bias vector for the batched GEMM (the real bias is added by the output transform)
  bool wino_ok = false; size_t zero_off = 0;
  size_t wl_off = 0; int lowc_ce = 0, lowc_ncol = 0;   // k_conv_lowc filter [9*CE/KG][KG][NCOL] (real-channel K order), CE / NCOL of its instantiation
};

struct TensorDef {
  std::string name;
  std::vector<int64_t> dims;
  bool used;
  std::vector<float> data;
  bool set = false;
};

struct SpadeGroup {   // one SPADE launch: 1 or 2 modulations sharing the normalised tensor
  std::string key;    // "<block>.0s" / "<block>.0" / "<block>.1"
  int C = 0, Cp = 0, cond = 0, nsets = 1;
  size_t w_off = 0, b_off = 0;
  size_t w16_off = 0; // bf16 copy of the gamma/beta filters (bf16 storage mode)
  int npad = 0;       // virtual columns (multiple of 64)
  // 16 modulated channels in all (C = 16, one set): a second copy in [gamma(16) | beta(16)] order, one 32-column
  // fragment instead of two half-empty ones (fp32 SPADE variants with NF = 1)
  size_t w1_off = 0, b1_off = 0;
  // condition level (index of the ref_embedding map this SPADE reads) and first column of this group inside the level's
  // concatenated gamma/beta filter matrix [sum of npad][cond] (the groups of a level are laid out back to back in the blob,
  // in execution order, so that ONE 1x1 GEMM on the level's map can produce the gamma/beta of all of them: cond_level_gemm)
  int level = 0, col0 = 0;
};

struct Cfg {
  rib_config c;
  int nf(int i) const { return std::min(c.max_num_filters, c.num_filters << i); }
  int mask_nf(int i) const { return std::min(c.mask_max_filters, c.mask_filters << i); }
  int emb_ch(int i) const { return std::min(c.emb_max_filters, c.emb_filters << i); }
  int cond_ch(int i) const { return std::min(c.max_num_filters, c.emb_filters << std::min(i, c.emb_down)); }
  int num_res_blocks() const { return (int)std::ceil((c.num_layers - c.num_down_img) / 2.0) * 2; }
};

void add_embedder(std::vector<ConvDef>& L, const Cfg& g, const std::string& prefix, int cin, bool used) {
  ConvDef c; c.name = prefix + ".conv_first"; c.cin = cin; c.cout = g.emb_ch(0); c.used = used; L.push_back(c);
  for (int i = 0; i < g.c.emb_down; ++i) {
    ConvDef d; d.name = prefix + ".down_" + std::to_string(i); d.cin = g.emb_ch(i); d.cout = g.emb_ch(i + 1);
    d.stride = 2; d.used = used; L.push_back(d);
  }
}
void add_spade_block(std::vector<ConvDef>& L, const std::string& name, int cin, int cout, int cond) {
  const int hidden = std::min(cin, cout);
  ConvDef a; a.name = name + ".conv_block_0"; a.cin = cin; a.cout = hidden; a.spade_cond = cond; L.push_back(a);
  ConvDef b; b.name = name + ".conv_block_1"; b.cin = hidden; b.cout = cout; b.spade_cond = cond; L.push_back(b);
  if (cin != cout) {
    ConvDef s; s.name = name + ".conv_block_s"; s.cin = cin; s.cout = cout; s.ks = 1; s.spade_cond = cond; L.push_back(s);
  }
}
void add_mask_block(std::vector<ConvDef>& L, const std::string& name, int cin, int cout) {
  const int hidden = std::min(cin, cout);
  ConvDef a; a.name = name + ".conv_block_0"; a.cin = cin; a.cout = hidden; a.in_affine = true; L.push_back(a);
  ConvDef b; b.name = name + ".conv_block_1"; b.cin = hidden; b.cout = cout; b.in_affine = true; L.push_back(b);
  if (cin != cout) {
    ConvDef s; s.name = name + ".conv_block_s"; s.cin = cin; s.cout = cout; s.ks = 1; s.in_affine = true; L.push_back(s);
  }
}

std::vector<ConvDef> build_inventory(const Cfg& g) {
  std::vector<ConvDef> L;
  const rib_config& c = g.c;
  add_embedder(L, g, "ref_embedding", c.image_nc * 2, true);
  add_embedder(L, g, "label_embedding", c.label_nc, false);
  for (int i = c.num_down_img; i >= 0; --i) add_spade_block(L, "up_" + std::to_string(i), g.nf(i + 1), g.nf(i), g.cond_ch(i));
  { ConvDef d; d.name = "conv_img"; d.cin = c.num_filters; d.cout = c.image_nc; d.spectral = false; L.push_back(d); }
  { ConvDef d; d.name = "conv_mask"; d.cin = c.num_filters; d.cout = 1; d.spectral = false; d.used = false; L.push_back(d); }
  { ConvDef d; d.name = "down_first"; d.cin = c.label_nc; d.cout = c.num_filters; d.spectral = false; L.push_back(d); }
  for (int i = 0; i <= c.num_down_img; ++i) add_spade_block(L, "down_" + std::to_string(i), g.nf(i), g.nf(i + 1), g.cond_ch(i));
  const int res_ch = g.nf(c.num_down_img + 1);
  for (int i = 0; i < g.num_res_blocks(); ++i) add_spade_block(L, "res_" + std::to_string(i), res_ch, res_ch, g.cond_ch(c.num_down_img + 1));
  const std::string m = "flow_network_temp";
  const char* branches[2] = {"down_lbl", "down_img"};
  const int bcin[2] = {c.label_nc, c.image_nc * 3};
  for (int b = 0; b < 2; ++b) {
    ConvDef d; d.name = m + "." + branches[b] + ".0"; d.cin = bcin[b]; d.cout = c.mask_filters; d.in_affine = true; L.push_back(d);
    for (int i = 0; i < c.mask_down; ++i) {
      ConvDef e; e.name = m + "." + branches[b] + "." + std::to_string(i + 1); e.cin = g.mask_nf(i); e.cout = g.mask_nf(i + 1);
      e.stride = 2; e.in_affine = true; L.push_back(e);
    }
  }
  const int ch = g.mask_nf(c.mask_down);
  for (int i = 0; i < c.mask_res_blocks; ++i) add_mask_block(L, m + ".res_flow." + std::to_string(i), i == 0 ? ch * 2 : ch, ch);
  for (int j = 0; j < c.mask_down; ++j) {
    const int i = c.mask_down - 1 - j;
    ConvDef d; d.name = m + ".up_flow." + std::to_string(2 * j + 1); d.cin = g.mask_nf(i + 1); d.cout = g.mask_nf(i); d.in_affine = true; d.ups_in = true; L.push_back(d);
  }
  { ConvDef d; d.name = m + ".conv_mask.0"; d.cin = c.mask_filters; d.cout = 1; d.spectral = false; L.push_back(d); }
  return L;
}

// ------------------------------------------------------------------------------------------
// kernel variant table
// ------------------------------------------------------------------------------------------
typedef void (*IgemmFn)(const IgemmParams);
struct Variant {
  int FRW, WM, WN, MF, NF, BK, STRIDE, KS; bool UPS, SPADE;
  IgemmFn fn;          // generic instantiation (fused-shortcut loop and input prologue compiled in)
  int BF16 = 0;        // precision of this instantiation: PREC_F32 / PREC_BF16 / PREC_F16 (16-bit storage)
  IgemmFn fn_pro = nullptr;    // without the fused-shortcut loop
  IgemmFn fn_lean = nullptr;   // without the fused-shortcut loop and without the prologue
  int KW = 1;                  // in-workgroup split-K: KW groups of 4 waves (256*KW threads) per tile
  int TB = 1;                  // filter slices staged per barrier: 1 tap, or 3 = one row of a 3x3 filter
  // FRW == 0: not a k_igemm instantiation but a tile of k_gemm_dma (plain GEMM, operands staged by LDS-DMA): (32 WM) x (32 NF WN),
  // fp32 only; serves the batched Winograd-domain GEMMs and the condition-level gamma/beta GEMM
  int DMAK = 0;                // k_igemm instantiations whose operand tiles are staged by LDS-DMA (filters always, the input tile in
                               // the lean one); no fused-shortcut instantiation; reported as TB = 100 in the exported geometry
  typedef void (*GemmDmaFn)(const GemmDmaParams);
  GemmDmaFn gfn = nullptr;
  GemmDmaFn gfn_x3 = nullptr;  // the same tile with split-bf16 products (rib_set_products(RIB_PRODUCTS_BF16X3); fp32 storage only)
  bool dma() const { return FRW == 0; }
  int BM() const { return 32 * WM; }
  int TH() const { return (32 / FRW) * MF * WM; }
  int TW() const { return FRW; }
  int BN() const { return NF == 0 ? 16 * WN : 32 * NF * WN; }   // NF == 0: 16-column MFMA path
  int lds_bytes() const {
    if (dma()) return 2 * (BM() + BN()) * 32 * 4;
    if (DMAK) {      // IgemmGeom with DMA: unpadded rows, whole DMA instructions, two input-tile buffers (lean) - the larger of lean / prologue
      const int ih_ = (TH() - 1) * STRIDE + KS, iw_ = (TW() - 1) * STRIDE + KS;
      const int iwp_ = (STRIDE == 2 && FRW == 8) ? ((iw_ + 3) / 8 * 8 + 4) : iw_;
      const int sl = BK / 4;
      const int sb = (BN() * sl + 255) / 256 * 256 * 4;
      const int lean = 2 * ((ih_ * iwp_ * sl + 255) / 256 * 256 * 4) + 2 * TB * sb, pro = ih_ * iwp_ * (BK + 4) + 2 * sb;
      return (TB == 9 || lean > pro ? lean : pro) * 4;
    }
    const int ih = UPS ? TH() + 2 : (TH() - 1) * STRIDE + KS, iw = UPS ? TW() + 2 : (TW() - 1) * STRIDE + KS;
    const int iwp = (STRIDE == 2 && FRW == 8) ? ((iw + 3) / 8 * 8 + 4) : iw;   // IgemmGeom::IWP
    const int ck = (BF16 != PREC_F32 ? BK / 2 : BK) + 4;   // IgemmGeom::CK
    const bool db1 = KS == 1 && TB == 2;                                                        // 1x1, everything double-buffered
    const int main_loop = (((TB == 9 || db1) ? 2 : 1) * ih * iwp * ck + 2 * (db1 ? 1 : TB) * BN() * ck) * 4;   // IgemmGeom::NA, TBB
    const int kw_reduce = KW > 1 ? (NF == 0 ? 1 : NF) * MF * 16 * 256 * 4 : 0;
    return main_loop > kw_reduce ? main_loop : kw_reduce;
  }
};
// RIB_V: convolution geometries, three instantiations (generic / no shortcut loop / lean); the
// shortcut loop only exists for 3x3 stride-1 gathers, elsewhere "generic" already is "pro".
// RIB_VS: SPADE geometries (one instantiation).  RIB_VB: bf16 twins (generic only).
#define RIB_V(sec, FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP)                                                         \
  Variant{FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP,                                                             \
          &k_igemm<FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, false, (KS == 3 && S == 1 && !UPS), true>, false,   \
          &k_igemm<FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, false, false, true>,                               \
          &k_igemm<FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, false, false, false>},
// RIB_VK: in-workgroup split-K twin of a convolution geometry (fp32, 32-column path)
#define RIB_VK(sec, FRW, WM, WN, MF, NF, BK, S, KS, UPS, KW)                                                                  \
  Variant{FRW, WM, WN, MF, NF, BK, S, KS, UPS, false,                                                                    \
          &k_igemm<FRW, WM, WN, MF, NF, BK, S, KS, UPS, false, false, (KS == 3 && S == 1 && !UPS), true, KW>, false,      \
          &k_igemm<FRW, WM, WN, MF, NF, BK, S, KS, UPS, false, false, false, true, KW>,                                  \
          &k_igemm<FRW, WM, WN, MF, NF, BK, S, KS, UPS, false, false, false, false, KW>, KW},
// RIB_VT: three-taps-per-barrier twin of a 3x3 convolution geometry
#define RIB_VT(sec, FRW, WM, WN, MF, NF, BK, S, UPS)                                                                         \
  Variant{FRW, WM, WN, MF, NF, BK, S, 3, UPS, false,                                                                    \
          &k_igemm<FRW, WM, WN, MF, NF, BK, S, 3, UPS, false, false, (S == 1 && !UPS), true, 1, 3>, false,               \
          &k_igemm<FRW, WM, WN, MF, NF, BK, S, 3, UPS, false, false, false, true, 1, 3>,                                \
          &k_igemm<FRW, WM, WN, MF, NF, BK, S, 3, UPS, false, false, false, false, 1, 3>, 1, 3},
// RIB_VTK: three taps per barrier AND KW wave groups per tile (8 / 16 waves share the 3-slice filter buffers)
#define RIB_VTK(sec, FRW, WM, WN, MF, NF, BK, S, KW)                                                                         \
  Variant{FRW, WM, WN, MF, NF, BK, S, 3, false, false,                                                                  \
          &k_igemm<FRW, WM, WN, MF, NF, BK, S, 3, false, false, false, (S == 1), true, KW, 3>, false,                    \
          &k_igemm<FRW, WM, WN, MF, NF, BK, S, 3, false, false, false, false, true, KW, 3>,                             \
          &k_igemm<FRW, WM, WN, MF, NF, BK, S, 3, false, false, false, false, false, KW, 3>, KW, 3},
// RIB_V9: all nine filter slices of a chunk per barrier pair (TB = 9), optionally with KW wave groups
#define RIB_V9(sec, FRW, WM, WN, MF, NF, BK, S, KW)                                                                          \
  Variant{FRW, WM, WN, MF, NF, BK, S, 3, false, false,                                                                  \
          &k_igemm<FRW, WM, WN, MF, NF, BK, S, 3, false, false, false, (S == 1), true, KW, 9>, false,                    \
          &k_igemm<FRW, WM, WN, MF, NF, BK, S, 3, false, false, false, false, true, KW, 9>,                             \
          &k_igemm<FRW, WM, WN, MF, NF, BK, S, 3, false, false, false, false, false, KW, 9>, KW, 9},
// RIB_VU4: phase-decomposed upsample convolution with the four taps of a phase per barrier (TB = 4)
#define RIB_VU4(sec, FRW, WM, WN, MF, NF, BK)                                                                                \
  Variant{FRW, WM, WN, MF, NF, BK, 1, 3, true, false,                                                                   \
          &k_igemm<FRW, WM, WN, MF, NF, BK, 1, 3, true, false, false, false, true, 1, 4>, false,                         \
          &k_igemm<FRW, WM, WN, MF, NF, BK, 1, 3, true, false, false, false, true, 1, 4>,                               \
          &k_igemm<FRW, WM, WN, MF, NF, BK, 1, 3, true, false, false, false, false, 1, 4>, 1, 4},
#define RIB_VS(sec, FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP) \
  Variant{FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, &k_igemm<FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, false>, false},
// RIB_VSK: in-workgroup split-K twin of a SPADE geometry: the fused kernel on the small deep maps with 8 / 16 waves
// per tile instead of the unfused pair (split-K GEMM into slabs + k_spade_modulate)
#define RIB_VSK(sec, FRW, WM, WN, MF, NF, BK, KW) \
  Variant{FRW, WM, WN, MF, NF, BK, 1, 1, false, true, &k_igemm<FRW, WM, WN, MF, NF, BK, 1, 1, false, true, false, true, true, KW>, false, nullptr, nullptr, KW},
// (every 16-bit geometry exists twice: bf16 and half elements, PREC_BF16 / PREC_F16)
#define RIB_VB(sec, FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP) \
  Variant{FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, &k_igemm<FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, 1>, 1}, \
  Variant{FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, &k_igemm<FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, 2>, 2},
// 1x1, everything double-buffered (TB = 2): convolution (pro / lean) and SPADE
#define RIB_V1D(sec, FRW, WM, WN, MF, NF, BK, KW)                                                                    \
  Variant{FRW, WM, WN, MF, NF, BK, 1, 1, false, false,                                                               \
          &k_igemm<FRW, WM, WN, MF, NF, BK, 1, 1, false, false, 0, false, true, KW, 2>, 0,                            \
          &k_igemm<FRW, WM, WN, MF, NF, BK, 1, 1, false, false, 0, false, true, KW, 2>,                               \
          &k_igemm<FRW, WM, WN, MF, NF, BK, 1, 1, false, false, 0, false, false, KW, 2>, KW, 2},
#define RIB_VS1D(sec, FRW, WM, WN, MF, NF, BK, KW) \
  Variant{FRW, WM, WN, MF, NF, BK, 1, 1, false, true, &k_igemm<FRW, WM, WN, MF, NF, BK, 1, 1, false, true, 0, true, true, KW, 2>, 0, nullptr, nullptr, KW, 2},
#define RIB_VBX(sec, FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, KW, TB)                                                            \
  Variant{FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP,                                                                               \
          &k_igemm<FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, 1, (KS == 3 && S == 1 && !UPS && !SP), true, KW, TB>, 1, nullptr, nullptr, KW, TB}, \
  Variant{FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP,                                                                               \
          &k_igemm<FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, 2, (KS == 3 && S == 1 && !UPS && !SP), true, KW, TB>, 2, nullptr, nullptr, KW, TB},

// RIB_VD / RIB_VSD: LDS-DMA-staged twins (variants.def)
#define RIB_VD(sec, FRW, WM, WN, MF, NF, BK, S, KS) \
  dmak_variant(Variant{FRW, WM, WN, MF, NF, BK, S, KS, false, false, nullptr, 0,                                          \
                       &k_igemm<FRW, WM, WN, MF, NF, BK, S, KS, false, false, 0, false, true, 1, 1, 1>,                   \
                       &k_igemm<FRW, WM, WN, MF, NF, BK, S, KS, false, false, 0, false, false, 1, 1, 3>}),
#define RIB_VSD(sec, FRW, WM, WN, MF, NF, BK) \
  dmak_variant(Variant{FRW, WM, WN, MF, NF, BK, 1, 1, false, true, &k_igemm<FRW, WM, WN, MF, NF, BK, 1, 1, false, true, 0, false, false, 1, 1, 3>, 0}),

#define RIB_VD9(sec, FRW, WM, WN, MF, NF, BK, S) \
  dmak_variant(Variant{FRW, WM, WN, MF, NF, BK, S, 3, false, false, nullptr, 0, nullptr,                                   \
                       &k_igemm<FRW, WM, WN, MF, NF, BK, S, 3, false, false, 0, false, false, 1, 9, 3>, 1, 9}),

// the leanest instantiation that covers a launch
inline IgemmFn pick_igemm_fn(const Variant* v, const IgemmParams& p) {
  if (p.x2 != nullptr || (v->fn && !v->fn_pro)) return v->fn;
  if (p.pro_scale == nullptr && !p.pro_lrelu && v->fn_lean) return v->fn_lean;
  return v->fn_pro ? v->fn_pro : v->fn;
}

inline Variant dmak_variant(Variant v) { v.DMAK = 1; return v; }
inline Variant dma_variant(int WM, int WN, int NF, Variant::GemmDmaFn fn, int prec = PREC_F32, Variant::GemmDmaFn fn_x3 = nullptr) {
  Variant v{0, WM, WN, 1, NF, prec == PREC_F32 ? 32 : 64, 1, 1, false, false, nullptr};      // BK: elements of a 128-byte row chunk
  v.gfn = fn;
  v.gfn_x3 = fn_x3;
  v.BF16 = prec;
  return v;
}
// split-bf16 twin of an fp32 tile: the hh products in their own accumulator where the registers allow it (NF <= 2)
#define RIB_DMA_X3(WM, WN, NF) PREC_F32, &k_gemm_dma<WM, WN, NF, ST_F32, (NF <= 2 ? 2 : 1)>
const Variant kVariants[] = {
#include "variants.def"
    // k_gemm_dma tiles (instantiated in this translation unit): 128x64, 64x64, 64x128, 128x128, 128x32
    dma_variant(4, 1, 2, &k_gemm_dma<4, 1, 2>, RIB_DMA_X3(4, 1, 2)), dma_variant(2, 2, 1, &k_gemm_dma<2, 2, 1>, RIB_DMA_X3(2, 2, 1)),
    dma_variant(2, 2, 2, &k_gemm_dma<2, 2, 2>, RIB_DMA_X3(2, 2, 2)), dma_variant(4, 1, 4, &k_gemm_dma<4, 1, 4>, RIB_DMA_X3(4, 1, 4)),
    dma_variant(4, 1, 1, &k_gemm_dma<4, 1, 1>, RIB_DMA_X3(4, 1, 1)),
    // ... and their 16-bit storage twins (the condition-level GEMMs of the bf16 / half modes)
    dma_variant(4, 1, 2, &k_gemm_dma<4, 1, 2, ST_BF16>, PREC_BF16), dma_variant(2, 2, 1, &k_gemm_dma<2, 2, 1, ST_BF16>, PREC_BF16),
    dma_variant(2, 2, 2, &k_gemm_dma<2, 2, 2, ST_BF16>, PREC_BF16), dma_variant(4, 1, 4, &k_gemm_dma<4, 1, 4, ST_BF16>, PREC_BF16),
    dma_variant(4, 1, 2, &k_gemm_dma<4, 1, 2, ST_F16>, PREC_F16), dma_variant(2, 2, 1, &k_gemm_dma<2, 2, 1, ST_F16>, PREC_F16),
    dma_variant(2, 2, 2, &k_gemm_dma<2, 2, 2, ST_F16>, PREC_F16), dma_variant(4, 1, 4, &k_gemm_dma<4, 1, 4, ST_F16>, PREC_F16),
};
const int kNumVariants = sizeof(kVariants) / sizeof(kVariants[0]);

// Choose (variant, split-K factor) with a small analytic cost model, in shader cycles:
//   per wave and tap   (BK/8)*MF*NF*4 MFMAs of 64 cycles + a fixed barrier/LDS overhead
//   latency bound      rounds of resident workgroups x the dependent chain of one workgroup
//   matrix-core bound  each CU runs the MFMAs of its workgroups serially on its 4 SIMDs
// A split-K launch pays a second (slab-summing) kernel.
struct Choice { const Variant* v = nullptr; int ksplit = 1; double cycles = 0; };

Choice choose_variant_dt(int bf16, int stride, int ks, bool ups, bool spade, int ncols, int B, int Hout, int Wout,
                         int Cin, bool allow_split, int Cin2, bool allow_n16);

Choice choose_variant(int bf16, int stride, int ks, bool ups, bool spade, int ncols, int B, int Hout, int Wout,
                      int Cin, bool allow_split, int Cin2 = 0, bool allow_n16 = false) {
  // a bf16 handle stores bf16 activations: only the bf16 kernels can read them
  return choose_variant_dt(bf16, stride, ks, ups, spade, ncols, B, Hout, Wout, Cin, allow_split, Cin2, allow_n16 && bf16 == PREC_F32);
}

Choice choose_variant_dt(int bf16, int stride, int ks, bool ups, bool spade, int ncols, int B, int Hout, int Wout,
                         int Cin, bool allow_split, int Cin2, bool allow_n16) {
  Choice best;
  best.cycles = 1e300;
  static const int kSplits[] = {1, 2, 3, 4, 6, 8, 12, 16};
  for (int i = 0; i < kNumVariants; ++i) {
    const Variant& v = kVariants[i];
    if (v.dma() || v.DMAK) continue;        // (k_gemm_dma tiles: pick_gemm_dma; DMA-staged k_igemm twins: tuned choices only)
    if (v.STRIDE != stride || v.KS != ks || v.UPS != ups || v.SPADE != spade || Cin % v.BK != 0 || Cin2 % v.BK != 0) continue;
    if (v.NF == 0 && !allow_n16) continue;
    if (v.SPADE && v.NF == 1) continue;      // 16-channel SPADE layout: chosen explicitly (Builder::spade)
    if (v.BF16 != bf16) continue;   // (bf16 carries the handle's precision mode)
    const int BK = v.BK;
    // in-workgroup split-K only where the tile grid alone cannot fill the chip
    if ((v.KW > 1 || v.TB > 1) && (long)((Hout + v.TH() - 1) / v.TH()) * ((Wout + v.TW() - 1) / v.TW()) * ((ncols + v.BN() - 1) / v.BN()) * B >= 512) continue;
    const int nchunks = Cin / BK;
    // phase-decomposed upsample conv: tiles of SOURCE pixels, 16 (phase, tap) steps per chunk
    const int Ht = ups ? Hout / 2 : Hout, Wt = ups ? Wout / 2 : Wout;
    const long tiles = (long)((Ht + v.TH() - 1) / v.TH()) * ((Wt + v.TW() - 1) / v.TW());
    const long ntiles = v.NF == 0 ? 1 : (ncols + v.BN() - 1) / v.BN();
    const int occ = std::max(1, std::min(v.MF * v.NF * (ups ? 4 : 1) >= 4 ? 4 : 6, 160 * 1024 / v.lds_bytes()));
    for (int S : kSplits) {
      if (S > nchunks || (S > 1 && (!allow_split || v.NF == 0))) break;
      const long wgs = tiles * ntiles * B * S;
      const int chunks = (nchunks + S - 1) / S;
      const int taps = ups ? 16 : ks * ks;
      const double mfma_tap = v.NF == 0 ? (BK / 16) * v.MF * 8 * 32.0
                                        : (v.BF16 != PREC_F32 ? (BK / 16) * v.MF * v.NF * 32.0 : (BK / 8) * v.MF * v.NF * 4 * 64.0);
      // per tap: barrier + LDS write/read latency; per chunk: halo-tile commit; per workgroup:
      // first global loads (HBM latency) + epilogue.  Overheads of one workgroup hide behind the
      // matrix work of the other `occ` resident ones (measured: occupancy is the dominant lever).
      // in-workgroup split-K: KW wave groups interleave on the same SIMDs, which hides 1/KW more of the
      // overheads; the accumulator hand-over costs two barriers and 16 KB of LDS traffic per extra group
      const double ovh_wg = (chunks * (std::max(1, taps / v.TB) * 350.0 + (ks == 1 && v.TB == 2 ? 300.0 : 600.0)) + 7000.0) / v.KW + (v.KW - 1) * 1500.0;
      const double mfma_wg = (double)chunks * taps * mfma_tap;
      const double t_lat = std::ceil((double)wgs / (256.0 * occ)) * (mfma_wg + ovh_wg);
      const double t_cu = std::ceil((double)wgs / 256.0) * (mfma_wg + ovh_wg / occ);
      double t = std::max(t_lat, t_cu);
      if (S > 1) t += 6000.0 + (double)(S + 1) * B * Hout * Wout * pad32(ncols) * 4.0 / 1667.0;
      if (t < best.cycles) { best.cycles = t; best.v = &v; best.ksplit = S; }
    }
  }
  return best;
}

// default k_gemm_dma tile of a Z x [M x N x K] problem: the largest tile that still gives two workgroups per CU, else the
// tile with the most workgroups (RIB_NO_DMA=1: none - the callers fall back to k_igemm's 1x1 variants)
const Variant* pick_gemm_dma(int prec, long Z, int M, int N, int K) {
  static const bool off = getenv("RIB_NO_DMA") != nullptr;
  if (off || K % (prec == PREC_F32 ? 32 : 64) != 0) return nullptr;
  const Variant* best = nullptr; long best_area = 0, best_wgs = 0;
  for (int i = 0; i < kNumVariants; ++i) {
    const Variant& v = kVariants[i];
    if (!v.dma() || v.BF16 != prec) continue;
    const long wgs = Z * ((M + v.BM() - 1) / v.BM()) * ((N + v.BN() - 1) / v.BN());
    const long area = (long)v.BM() * v.BN();
    const bool full = wgs >= 512, bfull = best_wgs >= 512;
    if (!best || (full && (!bfull || area > best_area)) || (!full && !bfull && wgs > best_wgs)) { best = &v; best_area = area; best_wgs = wgs; }
  }
  return best;
}

// ------------------------------------------------------------------------------------------
// launch plan
// ------------------------------------------------------------------------------------------
enum OpKind { OP_PACK, OP_IGEMM, OP_FINALIZE, OP_POOL, OP_INADD, OP_SPLITEPI, OP_MODULATE, OP_WINO_IN, OP_WINO_OUT, OP_LOWC, OP_GEMM };

// pointer encoding inside a plan: workspace-relative offsets (bytes) or weight-blob offsets
// (floats); resolved at launch time.
enum PtrSpace { PS_NULL = 0, PS_WS, PS_WEIGHT, PS_USER, PS_WINO };   // PS_WINO: off = index of a Winograd filter set of the handle
enum UserSlot { U_LABEL = 0, U_FAKE, U_PREV, U_IMG, U_MASK, U_FUSE, U_COUNT };   // U_FUSE: optional fused frame (null: no blend)
struct PRef {
  PtrSpace sp = PS_NULL;
  size_t off = 0;   // bytes for WS, floats for WEIGHT, slot id for USER
};

// A plan runs its launches in order on the caller's stream.  (Rounds 1-2 carried an option to run the condition encoder
// and the label branch on side streams: measured slower three times - two queues cost more than the overlap returns on
// this runtime - and incompatible with the lifetime-shared workspace; retired in round 3.)

struct Op {
  OpKind kind;
  int kclass;
  std::string name;
  // igemm
  const Variant* var = nullptr;
  std::string for_op;        // a finalize launch emitted on behalf of this consumer (rib_time_op times them together)
  bool label_only = false;   // depends on the label map only (pack.label, down_first, the mask network's label branch)
  int small_co = 0;   // > 0: a head convolution with small_co output channels: k_conv_head (matrix cores, taps as GEMM
                      //      columns) when `head`, else k_conv_small (direct, vector ALUs), instead of k_igemm
  bool head = false;
  bool fuse_blend = false;   // the mask head also writes the driver's blend into user slot U_FUSE when the caller gave one
  IgemmParams ip;   // scalar fields pre-filled; pointers resolved from the PRefs below
  PRef x, pro_scale, pro_shift, w, bias, y, res, y_nchw, stat, xm, m_scale, m_shift, ys0, ys1, slab, x2, w2;
  PRef m_part;   // consumer-side InstanceNorm finalize in the SPADE epilogue (IgemmParams)
  // consumer-side finalize in k_wino_in / k_spade_modulate (slot 0) and k_in_add (slots 0, 1): the StatSrc pointers
  PRef st_part[2], st_gamma[2], st_beta[2];
  // split-K epilogue
  SplitEpiParams sp; PRef s_slab, s_bias, s_y, s_res, s_stat;
  // unfused SPADE modulate
  ModulateParams mp; PRef m_slab, m_bias, m_xm, m_sc, m_sh, m_ys0, m_ys1;
  dim3 grid;
  double flops = 0;
  // finalize
  FinalizeParams fp; PRef f_part, f_gamma, f_beta, f_scale, f_shift;
  // pool
  PoolParams pp; PRef p_x, p_y, p_stat;
  // in_add
  InAddParams ap; PRef a_t1, a_sc1, a_sh1, a_ts, a_scs, a_shs, a_x, a_out;
  // pack
  PackParams kp; PRef k_s0, k_s1, k_s2, k_dst;
  LowcParams lc; PRef lc_s0, lc_s1, lc_s2, lc_w, lc_bias, lc_y, lc_stat; int lowc_ce = 0, lowc_ncol = 0, lowc_tw = 32;
  // Winograd transforms
  WinoInParams wi; PRef wi_x, wi_sc, wi_sh, wi_v, wi_slab, wi_sbias, wi_x2, wi_xres, wi_o, wi_sc2, wi_sh2; int wi_mode = 0;
  WinoOutParams wo; PRef wo_m, wo_bias, wo_y, wo_res, wo_stat;
  GemmDmaParams gp; PRef g_a, g_b, g_c;      // OP_GEMM: k_gemm_dma (tile = var)
  bool wino = false;   // this k_igemm launch is the 16- / 36-way batched Winograd-domain GEMM (executes 4/9 or 1/4 of its nine-tap FLOP count)
  int wino_m = 0;      // 2 or 4 on the three launches of a Winograd convolution
};

struct PendingStats {
  Op fin;
  bool pushed = false;
  size_t part_off = 0; int tiles = 0, Cs = 0; float inv_count = 0.f;
  bool affine = false; size_t g_off = 0, be_off = 0;
};

struct Tap { std::string name; size_t off; int Cp, C, H, W; };

struct LazySrc;
struct Act {       // NHWC activation in the workspace
  size_t off = 0;  // bytes
  int Cp = 0, C = 0, H = 0, W = 0;
  // virt: never materialised - the channel concatenation of up to three of the caller's NCHW fp32 tensors, read in place
  // by k_conv_lowc (usrc: user slots, uc: their channel counts)
  bool virt = false; int usrc[3] = {0, 0, 0}, uc[3] = {0, 0, 0};
  // lazy: never stored - the output of an unfused SPADE (WSRC_SPADE) or of a mask-network join (WSRC_JOIN) that the
  // Winograd input transform of its only consumer computes on the fly (Builder::conv_wino)
  std::shared_ptr<LazySrc> lazy;
};
struct PendingStats;
struct Norm {      // (scale, shift) arrays [B][ld]
  size_t sc = 0, sh = 0; int ld = 0;
  bool valid = false;
  // the k_stats_finalize launch that would fill the arrays, not emitted yet: a consumer that can reduce the
  // producer's partial sums itself (k_igemm prologue / SPADE epilogue, few partials) takes them from here and the
  // launch never happens; any other consumer emits it first (Builder::materialize)
  std::shared_ptr<PendingStats> pend;
};

struct LazySrc {
  int mode = 0;                 // WSRC_SPADE / WSRC_JOIN
  std::string name;             // the launch this replaces (for error messages)
  Act x; Norm nx;               // SPADE: tensor being normalised + its statistics; JOIN: t1 + n1
  bool x_ups = false, lrelu = false;
  size_t slab_off = 0; int slab_ld = 0, col0 = 0; size_t sbias_off = 0;      // SPADE: gamma/beta slab of the level, this group's bias
  bool has2 = false; Act x2; Norm n2; Act xres; Act o;                         // JOIN: ts + ns (learned shortcut) or xres; o = the stored join
};

// results of the label-only launches, as (offset, bytes) into the plan's workspace: in an autoregressive chain the
// label maps of all T frames are known up front, so rib_chain runs these launches ONCE at batch T*B (a
// labels-only plan) and copies each frame's slices into the frame plan's slots
struct LabelSlots { size_t x0 = 0, x0_b = 0, nx_sc = 0, nx_sh = 0, nx_b = 0, cat = 0, cat_b = 0, ncat_sc = 0, ncat_sh = 0, ncat_b = 0; };

struct Plan {
  bool labels_only = false;
  LabelSlots ls;
  int B, H, W;
  size_t ws_bytes = 0;
  size_t ws_virtual = 0;   // bytes before lifetime-based reuse (one range per buffer)
  std::vector<Op> ops;
  std::vector<Tap> taps;
  double flops[RIB_KC_COUNT] = {0};
};

}  // namespace

struct rib_handle {
  Cfg g;
  int device = 0;
  std::string err;
  std::vector<ConvDef> convs;
  std::map<std::string, int> conv_index;
  std::vector<TensorDef> tensors;
  std::map<std::string, int> tensor_index;
  std::vector<SpadeGroup> spades;
  std::map<std::string, int> spade_index;
  std::map<int, std::vector<int>> level_groups;   // condition level -> its SPADE groups (indices into spades), execution order
  float* d_blob = nullptr;
  size_t d_blob_floats = 0;      // allocated size (the bf16 storage mode carries bf16 filter copies: a larger blob)
  std::vector<float> host_blob;   // host-only handles (device < 0) keep the folded blob here
  size_t blob_floats = 0;
  size_t zero_off = 0;            // 64 floats of zeros in the blob: the source of zero-padding pixels for DMA-staged input tiles
  uint64_t layout_hash = 0;       // of the blob's offsets (assign_weight_layout); part of the blob header
  // Winograd-domain filter sets U = G g G^T, [positions][CoutPad][CinPad] fp32 each, made on the device from the folded
  // filters of the blob when a plan first asks for one (round 3: they were 391 of the blob's 514 MB, both sets of every layer,
  // folded on the host, uploaded and broadcast; a 512x512 frame uses 17 of the 34)
  struct WinoSet { int conv = 0, wm = 0; float* d = nullptr; size_t floats = 0; };
  std::vector<WinoSet> wino_sets;
  std::map<std::pair<int, int>, int> wino_index;
  bool weights_ready = false;
  int prec_mode = PREC_F32;    // rib_set_compute_dtype: PREC_BF16 / PREC_F16 = 16-bit storage + 16-bit matrix-core operands
  int prec() const { return prec_mode; }
  int products = 0;            // rib_set_products: RIB_PRODUCTS_BF16X3 = the k_gemm_dma launches of the fp32 mode run their split-bf16 twins
  bool mc16() const { return prec_mode != PREC_F32; }                    // the matrix-core kernels read 16-bit filter copies, 16-channel steps
  int padc(int c) const { return mc16() ? pad16(c) : pad8(c); }          // channel padding of an activation
  int esz() const { return mc16() ? 2 : 4; }                       // bytes per stored activation element
  bool warp_lds_ready = false; // rib_warp has raised k_warp's dynamic-LDS limit on this handle's device
  bool keep_taps = false;      // rib_set_debug_taps: intermediate activations stay intact until the end of a forward
  // rib_set_plan_batch: > 0 = every launch plan, whatever its batch, follows the kernel choices (tuned table / cost model, split-K,
  // Winograd tile, fused or level-wise SPADE) of THIS batch size, so that a sample's arithmetic does not depend on how many other
  // samples ran beside it (the folder driver: frames independent of segment grouping and of the world size); 0 = each batch its own
  int plan_batch = 0;
  std::map<std::array<int, 5>, std::unique_ptr<Plan>> plans;      // key {flags, tuneB, B, H, W}: see get_plan
  // tuned (variant, split-K) per "B,H,W|op name"; consulted before the analytic cost model
  std::map<std::string, std::pair<int, int>> choices;
  // profiling
  // rib_chain graph replay (rib_set_graph_replay / RIB_GRAPH=1): instantiated graphs of whole segments, keyed by everything a
  // launch parameter depends on - the shape and every pointer of the call; least recently used first out
  struct ChainGraph { std::array<uintptr_t, 12> key; hipGraphExec_t exec; uint64_t used; hipStream_t stream; };   // stream: of its last launch
  std::vector<ChainGraph> chain_graphs;
  bool graph_replay = false;
  uint64_t graph_clock = 0, graph_captures = 0, graph_replays = 0;
  // rib_rasterise: page-locked staging for the host tables (two slots; a slot is reused once the copy out of it has completed,
  // which its event says), so that the call only enqueues - the caller's pageable arrays are free again on return
  struct RasterStage { char* host = nullptr; size_t bytes = 0; hipEvent_t done = nullptr; };
  RasterStage raster_stage[2];
  int raster_next = 0;
  bool profiling = false;
  bool prof_kernels = false;   // rib_profile_begin_kernels: (start, stop) pairs bound to the dispatches instead of interval events
  std::vector<std::pair<int, hipEvent_t>> prof_events;   // (class of the launch behind the event, -1: end of a plan run)
  struct KernelEv { int kclass; hipEvent_t start, stop; };
  std::vector<KernelEv> prof_kernel_events;
  int64_t prof_launches[RIB_KC_COUNT] = {0};
  double prof_ms[RIB_KC_COUNT] = {0};
};

namespace {

#define HIP_TRY(h, expr)                                                                   \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess) {                                                                \
      (h)->err = fmt("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
      return RIB_ERR_HIP;                                                                  \
    }                                                                                      \
  } while (0)

// instantiated chain graphs hold the plans' kernel parameters: whatever rebuilds plans or moves the blob drops them
// an instantiated graph may still be queued or running on the stream of its last launch (the launch only enqueues): that stream
// is drained before the executable goes away (evictions and plan changes are rare; a replay never comes here)
void destroy_chain_graph(rib_handle::ChainGraph& g) {
  // The graph may have been replayed on several streams (the key does not hold the stream) and a stream of an earlier replay may
  // be gone by now: wait for the whole device instead of for the last stream alone.  Evictions and plan changes are rare; a
  // replay never comes here.  (rib_set_choice / rib_set_plan_batch / rib_set_products therefore BLOCK the host when the handle
  // holds captured segments: include/rib.h says so.)
  (void)hipDeviceSynchronize();
  (void)hipGraphExecDestroy(g.exec);
}
void drop_chain_graphs(rib_handle* h) {
  if (h->chain_graphs.empty()) return;
  if (h->device >= 0) (void)hipSetDevice(h->device);      // (the device-wide wait below must be this handle's device)
  for (auto& g : h->chain_graphs) destroy_chain_graph(g);
  h->chain_graphs.clear();
}

int fail(rib_handle* h, int code, const std::string& msg) {
  h->err = msg;
  return code;
}

const ConvDef& conv_of(const rib_handle* h, const std::string& name) {
  return h->convs[h->conv_index.at(name)];
}

// ------------------------------------------------------------------------------------------
// Winograd-domain filter sets, made on the device from the folded 3x3 filters
// ------------------------------------------------------------------------------------------
// U[xi = T r + q][o][i] = (G g G^T)[r][q], T = m + 2, g[dy][dx] = w[o][dy*3+dx][i] (the blob's [CoutPad][9][CinPad] layout; padded
// rows / columns are zeros and stay zeros).  fp64 with the roundings of the plain expression (no contraction), stored fp32.
// F(2x2): G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1].  F(4x4): Cook-Toom with the points {0, +-3/4, +-3/2, inf}
// (kernels.hip.h, k_wino4_in): G[j] = (1, a_j, a_j^2) / prod_{k != j} (a_j - a_k), last row (0, 0, 1).
__global__ __launch_bounds__(256) void k_wino_filters(const float* w, float* u, int coutp, int cinp, int wm) {
#pragma clang fp contract(off)
  const double G2[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  const double G4[6][3] = {{64.0 / 81, 0, 0},
                           {-128.0 / 243, -32.0 / 81, -8.0 / 27}, {-128.0 / 243, 32.0 / 81, -8.0 / 27},
                           {32.0 / 243, 16.0 / 81, 8.0 / 27},     {32.0 / 243, -16.0 / 81, 8.0 / 27},
                           {0, 0, 1}};
  const int T = wm + 2;
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (size_t)coutp * cinp) return;
  const int o = (int)(idx / cinp), i = (int)(idx % cinp);
  double g[3][3], t[6][3];
  for (int k = 0; k < 9; ++k) g[k / 3][k % 3] = (double)w[((size_t)o * 9 + k) * cinp + i];
  for (int r = 0; r < T; ++r)
    for (int dx = 0; dx < 3; ++dx) {
      const double* Gr = wm == 2 ? G2[r] : G4[r];
      t[r][dx] = Gr[0] * g[0][dx] + Gr[1] * g[1][dx] + Gr[2] * g[2][dx];
    }
  for (int r = 0; r < T; ++r)
    for (int q = 0; q < T; ++q) {
      const double* Gq = wm == 2 ? G2[q] : G4[q];
      u[((size_t)(r * T + q) * coutp + o) * cinp + i] = (float)(t[r][0] * Gq[0] + t[r][1] * Gq[1] + t[r][2] * Gq[2]);
    }
}

// (re)compute set `si` from the blob on `st`; the caller orders it against whatever reads the set
int fill_wino_set(rib_handle* h, int si, hipStream_t st) {
  rib_handle::WinoSet& ws = h->wino_sets[si];
  const ConvDef& c = h->convs[ws.conv];
  if (!ws.d || !h->d_blob) return RIB_OK;
  const size_t n = (size_t)c.coutp * c.cinp;
  RIB_KLAUNCH(k_wino_filters, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, h->d_blob + c.w_off, ws.d, c.coutp, c.cinp, ws.wm);
  HIP_TRY(h, hipGetLastError());
  return RIB_OK;
}

// index of the F(wm x wm) filter set of conv `ci`, created on first use (plan build): device memory + the transform when the
// weights are already there.  Synchronises the device (before and after the transform) once per new set: plan builds are
// one-time work, and include/rib.h says so for every entry point that can build a plan.  < 0: error (h->err).
int ensure_wino_set(rib_handle* h, int ci, int wm) {
  auto it = h->wino_index.find({ci, wm});
  if (it != h->wino_index.end()) return it->second;
  const ConvDef& c = h->convs[ci];
  rib_handle::WinoSet ws; ws.conv = ci; ws.wm = wm; ws.floats = (size_t)(wm + 2) * (wm + 2) * c.coutp * c.cinp;
  if (h->device >= 0) {
    hipError_t e = hipSetDevice(h->device);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&ws.d), ws.floats * sizeof(float));
    if (e != hipSuccess) { h->err = fmt("Winograd filter set of %s: %s", c.name.c_str(), hipGetErrorString(e)); return -1; }
  }
  const int si = (int)h->wino_sets.size();
  h->wino_sets.push_back(ws);
  h->wino_index[{ci, wm}] = si;
  if (h->device >= 0 && h->weights_ready) {
    // the blob may still be landing on a (non-blocking) stream of the caller's - rib_import_weights returns behind an
    // asynchronous copy - and the NULL stream does not order against such a stream: drain the device first, then transform
    if (hipDeviceSynchronize() != hipSuccess) { h->err = "Winograd filter transform: device not idle-able"; return -1; }
    if (fill_wino_set(h, si, nullptr) != RIB_OK) return -1;
    if (hipDeviceSynchronize() != hipSuccess) { h->err = "Winograd filter transform failed"; return -1; }
  }
  return si;
}

// after the blob changed (finalize / import): every existing set again, on `st`
int refresh_wino_sets(rib_handle* h, hipStream_t st) {
  for (int si = 0; si < (int)h->wino_sets.size(); ++si) {
    const int rc = fill_wino_set(h, si, st);
    if (rc) return rc;
  }
  return RIB_OK;
}
void free_wino_sets(rib_handle* h) {
  for (auto& ws : h->wino_sets) if (ws.d) (void)hipFree(ws.d);
  h->wino_sets.clear(); h->wino_index.clear();
}

// ------------------------------------------------------------------------------------------
// weight layout: sizes are fixed by the config, so offsets are assigned at create time
// ------------------------------------------------------------------------------------------
// k_conv_lowc instantiations: (channels rounded up to the MFMA's k-group, output columns)
static bool lowc_instantiated(int ce, int ncol) {
  return (ce == 6 && ncol == 64) || (ce == 10 && ncol == 32) || (ce == 22 && ncol == 32) || (ce == 24 && ncol == 16);
}
// (fp32 storage only: the 16-bit modes pack their inputs and run the first layers on the 16-bit matrix cores)
template <int CE, int NCOL> static void launch_lowc_t(int tw, dim3 grid, hipStream_t st, const LowcParams& p) {
  if (tw == 16) RIB_KLAUNCH((k_conv_lowc<CE, NCOL, ST_F32, 16>), grid, dim3(256), 0, st, p);
  else RIB_KLAUNCH((k_conv_lowc<CE, NCOL, ST_F32, 32>), grid, dim3(256), 0, st, p);
}
static void launch_lowc(int ce, int ncol, int tw, dim3 grid, hipStream_t st, const LowcParams& p) {
  if (ce == 6 && ncol == 64) launch_lowc_t<6, 64>(tw, grid, st, p);
  else if (ce == 10 && ncol == 32) launch_lowc_t<10, 32>(tw, grid, st, p);
  else if (ce == 22 && ncol == 32) launch_lowc_t<22, 32>(tw, grid, st, p);
  else if (ce == 24 && ncol == 16) launch_lowc_t<24, 16>(tw, grid, st, p);
}

// The blob starts with a header the importer checks (rib_import_weights): a blob is only meaningful to a handle with
// the same precision mode and the same layout (which also depends on RIB_NO_WINO / RIB_NO_LOWC / RIB_NO_SPADE16 as read
// when the layout was assigned).  64 floats are reserved; 8 words are used.
enum { BLOB_HEADER_FLOATS = 64, BLOB_MAGIC = 0x57424952 /* "RIBW" */, BLOB_VERSION = 3 };
struct BlobHeader { uint32_t magic, version, prec, reserved; uint64_t floats, layout_hash; };

void assign_weight_layout(rib_handle* h) {
  size_t off = BLOB_HEADER_FLOATS;
  h->spades.clear(); h->spade_index.clear();
  auto take = [&](size_t nfloats) { size_t o = off; off += (nfloats + 63) / 64 * 64; return o; };
  h->zero_off = take(64);
  for (auto& c : h->convs) {
    if (!c.used) continue;
    c.cinp = h->padc(c.cin);
    c.coutp = pad32(c.cout);
    c.w16_off = c.wp16_off = 0; c.fb_off = 0;
    c.w_off = take((size_t)c.coutp * c.ks * c.ks * c.cinp);
    if (c.ups_in) c.wp_off = take((size_t)c.coutp * 16 * c.cinp);
    c.b_off = take(c.coutp);
    if (c.in_affine) { c.g_off = take(c.coutp); c.be_off = take(c.coutp); }
    if (c.spade_cond && c.name.size() > 13 && c.name.compare(c.name.size() - 13, 13, ".conv_block_1") == 0 &&
        h->conv_index.count(c.name.substr(0, c.name.size() - 1) + "s"))
      c.fb_off = take(c.coutp);
  }
  // SPADE groups: block_0 (+ block_s when the shortcut is learned) share one launch.  Created level by level, in the
  // order the forward runs them (down_i, res_*, up_i), and their filters taken back to back: the filter matrices of a
  // level's groups form ONE [sum npad][cond] matrix in the blob (SpadeGroup::col0)
  {
    const rib_config& gc = h->g.c;
    const int D = gc.num_down_img;
    std::vector<std::pair<std::string, int>> blocks;      // (block name, condition level) in execution order
    for (int i = 0; i <= D; ++i) blocks.push_back({"down_" + std::to_string(i), std::min(gc.emb_down, i)});
    for (int i = 0; i < h->g.num_res_blocks(); ++i) blocks.push_back({"res_" + std::to_string(i), std::min(gc.emb_down, D + 1)});
    for (int i = D; i >= 0; --i) blocks.push_back({"up_" + std::to_string(i), std::min(i, gc.emb_down)});
    h->level_groups.clear();
    for (int level = 0; level <= gc.emb_down; ++level) {
      int col = 0;
      for (auto& bl : blocks) {
        if (bl.second != level) continue;
        for (const char* which : {"0", "1"}) {
          auto it = h->conv_index.find(bl.first + ".conv_block_" + which);
          if (it == h->conv_index.end()) continue;
          const ConvDef& c = h->convs[it->second];
          if (!c.used || !c.spade_cond) continue;
          SpadeGroup sg;
          sg.C = c.cin; sg.Cp = h->padc(c.cin); sg.cond = c.spade_cond;
          sg.nsets = (which[0] == '0' && h->conv_index.count(bl.first + ".conv_block_s")) ? 2 : 1;
          sg.key = bl.first + "." + which;
          sg.npad = (sg.nsets * sg.Cp + 31) / 32 * 64;
          sg.level = level; sg.col0 = col; col += sg.npad;
          sg.w_off = take((size_t)sg.npad * h->padc(sg.cond));      // (a multiple of 64 floats: no gap between two groups)
          h->level_groups[level].push_back((int)h->spades.size());
          h->spade_index[sg.key] = (int)h->spades.size();
          h->spades.push_back(sg);
        }
      }
    }
    for (auto& sg : h->spades) sg.b_off = take(sg.npad);
  }
  for (auto& c : h->convs) { c.wino_ok = false; c.zero_off = 0; }
  if (h->prec() == PREC_F32 && !getenv("RIB_NO_WINO")) {
    // 3x3 stride-1 convolutions with >= 128 input channels that own their launch (no fused 1x1 shortcut, no upsampled
    // input): the deep residual blocks of the generator and of the mask network
    for (auto& c : h->convs) {
      // (measured per layer at 512x512, transforms included: 512->512 at 32x32 58 -> 38 us, 256->256 at 64x64 47-52 -> 42 us,
      // 512->256 at 64x64 95 -> 63 us; 128->128 at 64x64 gains nothing: the two transforms cost ~13 us per layer)
      if (!c.used || c.ks != 3 || c.stride != 1 || c.ups_in || c.fb_off != 0 || c.cin < 256 || c.cout < 64 || c.cinp % 32 || 128 % (c.coutp / 4)) continue;
      // the plan picks F(4x4) or F(2x2) per layer from the map size (conv_wino) and asks for that filter set then
      c.wino_ok = true;
      c.zero_off = take(c.coutp);
    }
  }
  for (auto& sg : h->spades) {
    sg.w1_off = sg.b1_off = 0;
    if (h->prec() == PREC_F32 && sg.nsets * sg.Cp == 16 && !getenv("RIB_NO_SPADE16")) {
      sg.w1_off = take((size_t)32 * h->padc(sg.cond));
      sg.b1_off = take(32);
    }
  }
  // first-layer convolutions over the caller's tensors (k_conv_lowc, fp32 arithmetic): filter in real-channel K order.  Not
  // with bf16 storage: there the packed bf16 copy + bf16 matrix cores are faster (679 vs 672 frames/s, A/B)
  for (auto& c : h->convs) {
    c.wl_off = 0; c.lowc_ce = c.lowc_ncol = 0;
    if (!c.used || c.ks != 3 || c.stride != 1 || c.ups_in || c.fb_off != 0 || c.cin > 24 || h->mc16() || getenv("RIB_NO_LOWC")) continue;
    const int ncol = c.cout <= 16 ? 16 : c.coutp;      // (coutp is a multiple of 32; the 16-column instantiation serves Cout <= 16)
    const int ce = ncol == 16 ? (c.cin + 3) / 4 * 4 : (c.cin + 1) / 2 * 2;
    if (!lowc_instantiated(ce, ncol)) continue;
    c.lowc_ce = ce; c.lowc_ncol = ncol;
    c.wl_off = take((size_t)9 * ce * ncol);
  }
  if (h->mc16()) {   // bf16 copies of every filter tensor the matrix-core kernels read (two elements per float)
    const size_t pl = 1;
    for (auto& c : h->convs) {
      if (!c.used) continue;
      c.w16_off = take((pl * c.coutp * c.ks * c.ks * c.cinp + 1) / 2);
      if (c.ups_in) c.wp16_off = take((pl * c.coutp * 16 * c.cinp + 1) / 2);
    }
    for (auto& sg : h->spades) sg.w16_off = take((pl * sg.npad * h->padc(sg.cond) + 1) / 2);
  }
  h->blob_floats = off;
  // FNV-1a over every offset the kernels will be handed
  uint64_t hs = 1469598103934665603ull;
  auto mix = [&](uint64_t v) { for (int i = 0; i < 8; ++i) { hs ^= (v >> (8 * i)) & 0xff; hs *= 1099511628211ull; } };
  for (const auto& c : h->convs)
    for (uint64_t v : {(uint64_t)c.w_off, (uint64_t)c.b_off, (uint64_t)c.g_off, (uint64_t)c.be_off, (uint64_t)c.fb_off, (uint64_t)c.wp_off, (uint64_t)c.w16_off,
                       (uint64_t)c.wp16_off, (uint64_t)c.zero_off, (uint64_t)c.wl_off}) mix(v);
  for (const auto& sg : h->spades)
    for (uint64_t v : {(uint64_t)sg.w_off, (uint64_t)sg.b_off, (uint64_t)sg.w16_off, (uint64_t)sg.w1_off, (uint64_t)sg.b1_off}) mix(v);
  mix(off);
  h->layout_hash = hs;
}
BlobHeader blob_header_of(const rib_handle* h) {
  BlobHeader bh; bh.magic = BLOB_MAGIC; bh.version = BLOB_VERSION; bh.prec = (uint32_t)h->prec(); bh.reserved = 0;
  bh.floats = h->blob_floats; bh.layout_hash = h->layout_hash;
  return bh;
}

void register_tensors(rib_handle* h) {
  auto add = [&](const std::string& n, std::vector<int64_t> d, bool used) {
    TensorDef t; t.name = n; t.dims = d; t.used = used;
    h->tensor_index[n] = (int)h->tensors.size();
    h->tensors.push_back(t);
  };
  for (auto& c : h->convs) {
    if (c.spade_cond) {
      add(c.name + ".layers.norm.mlps.0.0.layers.conv.weight", {2 * c.cin, c.spade_cond, 1, 1}, c.used);
      add(c.name + ".layers.norm.mlps.0.0.layers.conv.bias", {2 * c.cin}, c.used);
    }
    const std::string p = c.name + ".layers.conv";
    if (c.spectral) {
      add(p + ".bias", {c.cout}, c.used);
      add(p + ".weight_orig", {c.cout, c.cin, c.ks, c.ks}, c.used);
      add(p + ".weight_u", {c.cout}, c.used);
      add(p + ".weight_v", {(int64_t)c.cin * c.ks * c.ks}, c.used);
    } else {
      add(p + ".weight", {c.cout, c.cin, c.ks, c.ks}, c.used);
      add(p + ".bias", {c.cout}, c.used);
    }
    if (c.in_affine) {
      add(c.name + ".layers.norm.weight", {c.cout}, c.used);
      add(c.name + ".layers.norm.bias", {c.cout}, c.used);
    }
  }
}

const std::vector<float>* tensor_data(rib_handle* h, const std::string& name) {
  auto it = h->tensor_index.find(name);
  if (it == h->tensor_index.end()) return nullptr;
  TensorDef& t = h->tensors[it->second];
  return t.set ? &t.data : nullptr;
}

// ------------------------------------------------------------------------------------------
// plan builder
// ------------------------------------------------------------------------------------------
// XCD-aware tile order (RIB_NO_XCD=1 disables): needs the tile count to be a multiple of 8
static int xcd_chunk_of(int tiles) {
  static const bool off = getenv("RIB_NO_XCD") != nullptr;
  return (!off && tiles >= 64 && tiles % 8 == 0) ? tiles / 8 : 0;
}

// ------------------------------------------------------------------------------------------
// FusionPolicy: every decision of the launch plan that is a POLICY rather than arithmetic - which launches are fused, which
// kernel family serves a layer, where a tensor is never materialised - by name, with the switch that turns it off for an
// A/B or a test.  One instance per plan build, read from the environment at that moment (tests flip the switches between
// two Generator constructions).  Three more switches act where the weight LAYOUT is assigned, because they decide which
// filter copies exist at all (assign_weight_layout: RIB_NO_WINO, RIB_NO_LOWC, RIB_NO_SPADE16), one where the k_gemm_dma tile
// of a GEMM is picked (pick_gemm_dma: RIB_NO_DMA), one in the tile order of a grid (xcd_chunk_of: RIB_NO_XCD).
// The tuned tables (rib_set_choice) choose WITHIN what the policy allows: tile variant, split-K factor, Winograd tile.
// ------------------------------------------------------------------------------------------
struct FusionPolicy {
  // InstanceNorm statistics: a consumer that can reduce <= STATS_MAX_PARTIALS producer partials itself does so and the
  // k_stats_finalize launch never exists (RIB_NO_CONSUMER_STATS=1: one finalize launch per normalised tensor)
  bool consumer_stats = !getenv("RIB_NO_CONSUMER_STATS");
  // heads with <= 3 output channels: k_conv_head (taps as MFMA columns; RIB_NO_HEADCONV=1 -> k_conv_small, direct on the
  // vector ALUs; RIB_NO_SMALLCONV=1 -> k_igemm with the columns padded to 16)
  bool head_conv = !getenv("RIB_NO_HEADCONV");
  bool small_conv = !getenv("RIB_NO_SMALLCONV");
  // 3x3 stride-1 layers on maps of at most this many pixels run in the Winograd domain (k_wino_in / batched GEMM / k_wino_out)
  long wino_max_px = getenv("RIB_WINO_MAX_PX") ? atol(getenv("RIB_WINO_MAX_PX")) : 16384;
  int wino_force = getenv("RIB_WINO_M") ? atoi(getenv("RIB_WINO_M")) : 0;      // 2 / 4: that Winograd tile wherever one is eligible
  // grid-level split-K (+ a k_splitk_epilogue launch) for layers whose tile grid cannot fill the chip
  bool split_k = !getenv("RIB_NO_SPLITK");
  // layers with <= 16 output channels on the 16-column MFMA path (v_mfma_f32_16x16x4_f32; fp32 storage only)
  bool n16 = !getenv("RIB_NO_N16");
  // k_conv_lowc tile width: 0 = per layer (16 for the store-bound 64-column layer without statistics, else 32)
  int lowc_tw = getenv("RIB_LOWC_TW") ? (atoi(getenv("RIB_LOWC_TW")) == 16 ? 16 : 32) : 0;
  // condition levels of at most this many pixels compute gamma/beta of ALL their SPADEs in one GEMM into a slab (0: off)
  long cond_gemm_max_px = getenv("RIB_COND_GEMM_MAX_PX") ? atol(getenv("RIB_COND_GEMM_MAX_PX")) : 4096;
  // SPADE on maps of <= 4096 pixels (x batch): unfused (slab GEMM + k_spade_modulate) instead of k_igemm<SPADE>
  bool unfused_spade = !getenv("RIB_NO_UNFUSED_SPADE");
  // an unfused SPADE's output / a mask-network join whose only consumer is a Winograd input transform is never stored
  bool lazy_sources = !getenv("RIB_NO_LAZY");
  // the learned 1x1 shortcut of a res block as extra K chunks of conv_block_1 (one launch less, no shortcut tensor)
  bool fuse_shortcut = !getenv("RIB_NO_FUSE_SHORTCUT");
  // buffers with disjoint lifetimes share workspace bytes (assign_physical)
  bool ws_reuse = !getenv("RIB_NO_WS_REUSE");
  // level i of the mask network's label encoder and of its image encoder in ONE paired launch (batch-1 frame plans)
  bool pair_mask_encoders = !getenv("RIB_NO_PAIR");
};

struct Builder {
  rib_handle* h;
  Plan* P;
  int B;
  const FusionPolicy pol;
  size_t ws = 0;
  std::string error;
  bool mark_label = false;
  bool pair_mask = false;    // the two encoders of the mask network level by level in paired launches (mask_branches_paired)
  bool defer_stats = true;   // false: every k_stats_finalize launch is emitted where its producer is (labels-only plans)
  // a consumer needs the (scale, shift) ARRAYS of n: emit the finalize launch now if it is still pending
  void materialize(const Norm& n, const std::string& consumer = std::string()) {
    if (n.pend && !n.pend->pushed) { n.pend->pushed = true; n.pend->fin.for_op = consumer; push(n.pend->fin); }
  }
  // the producer's partial sums are still there for a consumer-side finalize
  static bool has_partials(const Norm& n) { return n.pend && !n.pend->pushed; }
  // emit or defer the finalize launch f of a producer with `tiles` partials per sample
  void finalize_or_defer(Op& f, Norm* out, bool now, size_t part_off, int tiles, int Cs, float inv_count, bool affine, size_t g_off, size_t be_off) {
    if (now || !pol.consumer_stats || !defer_stats || mark_label || tiles > STATS_MAX_PARTIALS) { push(f); return; }
    auto ps = std::make_shared<PendingStats>();
    ps->fin = f; ps->part_off = part_off; ps->tiles = tiles; ps->Cs = Cs; ps->inv_count = inv_count;
    ps->affine = affine; ps->g_off = g_off; ps->be_off = be_off;
    out->pend = ps;
  }
  // A light consumer (k_wino_in, k_spade_modulate, k_in_add: channel slices of 64) reduces the producer's partials itself:
  // fills the StatSrc scalars and the op's PRefs from the pending finalize of n, which then never launches unless another
  // consumer needs the arrays.  false: the partials are gone / too many / the launch is forced -> the caller materializes.
  bool take_partials(const Norm& n, StatSrc& st, Op& op, int slot) {
    if (!has_partials(n) || n.pend->tiles > STATS_MAX_PARTIALS) return false;
    const PendingStats& ps = *n.pend;
    memset(&st, 0, sizeof st);
    st.tiles = ps.tiles; st.Cs = ps.Cs; st.inv_count = ps.inv_count;
    op.st_part[slot] = WS(ps.part_off);
    if (ps.affine) { op.st_gamma[slot] = WT(ps.g_off); op.st_beta[slot] = WT(ps.be_off); }
    return true;
  }
  // gamma/beta slab of a condition level (cond_level_gemm): [B][H*W][ld] fp32, group columns at SpadeGroup::col0
  struct LevelSlab { size_t off = 0; int ld = 0; };
  std::map<int, LevelSlab> level_slab;
  int tuneB = 0;      // batch whose tuned choices / cost-model decisions this plan follows (0: its own).  The labels-only
                      // plan of a chain follows the frame plan's, so that its results are bit-identical to per-frame launches
  int TB_() const { return tuneB > 0 ? tuneB : B; }
  void push(Op op) {
    op.label_only = mark_label;
    P->ops.push_back(op);
  }

  // Workspace layout.  During the build every buffer gets its own range of a virtual address space (bump
  // allocation, as the plan used to run: 1.1 GB at 512x512); assign_physical() then gives buffers whose lifetimes
  // (first .. last launch that touches them) do not overlap the same bytes, so that the working set of a frame stays
  // inside the 256 MB memory-side cache instead of streaming 1.1 GB of distinct addresses through it.
  struct AllocRec { size_t voff, bytes, phys; int first, last; };
  std::vector<AllocRec> allocs;
  size_t alloc(size_t bytes) {
    size_t o = ws; ws += align256(bytes);
    allocs.push_back(AllocRec{o, align256(bytes), o, INT_MAX, -1});
    return o;
  }
  Act act(int C, int H, int W) {
    Act a; a.C = C; a.Cp = h->padc(C); a.H = H; a.W = W;
    a.off = alloc((size_t)B * H * W * a.Cp * h->esz());
    return a;
  }
  Norm norm(int Cp, int images = 0) {
    Norm n; n.ld = Cp; n.valid = true;
    n.sc = alloc((size_t)(images ? images : B) * Cp * sizeof(float));
    n.sh = alloc((size_t)(images ? images : B) * Cp * sizeof(float));
    return n;
  }
  Act act_n(int images, int C, int H, int W) {     // an activation of `images` images instead of B
    Act a; a.C = C; a.Cp = h->padc(C); a.H = H; a.W = W;
    a.off = alloc((size_t)images * H * W * a.Cp * h->esz());
    return a;
  }
  // an activation that is never materialised: up to three of the caller's NCHW tensors, concatenated along channels
  Act virt(int C, int H, int W, int s0, int c0, int s1 = 0, int c1 = 0, int s2 = 0, int c2 = 0) {
    Act a; a.virt = true; a.C = C; a.Cp = 0; a.H = H; a.W = W;
    a.usrc[0] = s0; a.uc[0] = c0; a.usrc[1] = s1; a.uc[1] = c1; a.usrc[2] = s2; a.uc[2] = c2;
    return a;
  }
  // every consumer of a packed input tensor has a k_conv_lowc instantiation
  bool lowc_serves(std::initializer_list<const char*> names) {
    for (const char* nm : names) {
      auto it = h->conv_index.find(nm);
      if (it == h->conv_index.end() || !h->convs[it->second].wl_off) return false;
    }
    return true;
  }
  // conv_img can skip its NHWC copy only on the head kernel (conv(): `small` + `head`)
  bool head_without_nhwc(const ConvDef& c) {
    return c.cout <= 3 && c.ks == 3 && c.stride == 1 && (c.cinp == 16 || c.cinp == 32) && pol.small_conv && pol.head_conv;
  }
  static PRef WS(size_t off) { PRef r; r.sp = PS_WS; r.off = off; return r; }
  static PRef WT(size_t off) { PRef r; r.sp = PS_WEIGHT; r.off = off; return r; }
  static PRef US(int slot) { PRef r; r.sp = PS_USER; r.off = (size_t)slot; return r; }
  void tap(const std::string& name, const Act& a) { P->taps.push_back(Tap{name, a.off, a.Cp, a.C, a.H, a.W}); }

  // ---- generic convolution launch --------------------------------------------------------
  struct ConvArgs {
    const ConvDef* cd = nullptr;
    Act in;                 // input activation (stored size; half-res when ups)
    bool ups = false;
    const Norm* pro = nullptr; size_t pro_choff = 0;   // prologue affine (+channel offset into the arrays)
    bool pro_lrelu = false;
    Act out; int yoff = 0;  // destination buffer and channel offset
    int cout_store = -1;    // columns stored (default: out slice padded width)
    int act = ACT_NONE;
    const Act* res = nullptr; bool res_ups = false;
    const ConvDef* aux = nullptr; Act aux_in;   // fused 1x1 shortcut operand (accumulated into the same output)
    PRef y_nchw;            // optional NCHW copy
    PRef y_user;            // when set, y is this user tensor (yC = cout exactly)
    bool y_none = false;    // no NHWC destination at all (a head that only writes its NCHW copy)
    bool want_stats = false;
    Norm* stats_out = nullptr; size_t stats_choff = 0;  // finalize target (+channel offset)
    bool affine = false;    // finalize with the conv's IN gamma/beta
    bool stats_now = false; // emit the finalize launch at the producer (the arrays are shared / copied between plans)
    // Two convolutions of identical shape in one launch (IgemmParams::pair): `pair` is the second one; `in` / `out` hold
    // 2B images (image 2b + j belongs to convolution j).  pair_merge: the outputs of a pair land in ONE image of `out`
    // (B images) at channel offsets yoff and yoff + pair_yoff, and their statistics in one row of stats_out.
    const ConvDef* pair = nullptr; bool pair_merge = false; int pair_yoff = 0;
    bool no_split = false;              // never grid-level split-K (the paired launch cannot, and its unpaired twins must match it)
    std::vector<std::string> choice_names;   // tuned-choice keys tried before the op's own name
  };

  // Winograd F(2x2, 3x3) / F(4x4, 3x3) on the small maps (k_wino_in / batched 1x1 k_igemm / k_wino_out; kernels.hip.h):
  // maps up to 128x128 (1024x1024 frames: +3.5 % at batch 1 and 4); beyond that V and M (4x the activation each)
  // leave the caches and the direct kernel, which fills the chip there, was not beaten
  bool runs_wino(const ConvArgs& a, int Hout, int Wout) const {
    const ConvDef& c = *a.cd;
    return h->prec() == PREC_F32 && c.wino_ok && !a.ups && !a.aux && !a.res_ups && !a.pair && !a.in.virt && a.y_nchw.sp == PS_NULL &&
           a.y_user.sp == PS_NULL && (long)Hout * Wout <= pol.wino_max_px && Hout >= 2 && Wout >= 2;
  }
  // would conv `cd` on an HxW map (stride 1, plain arguments) run in the Winograd domain?  (callers that want to hand it a lazy input)
  bool would_wino(const ConvDef& cd, int H, int W) const { ConvArgs t; t.cd = &cd; return runs_wino(t, H, W); }

  bool conv(const ConvArgs& a, const std::string& opname) {
    const ConvDef& c = *a.cd;
    const int Hout = a.ups ? a.in.H * 2 : (c.stride == 2 ? a.in.H / 2 : a.in.H);
    const int Wout = a.ups ? a.in.W * 2 : (c.stride == 2 ? a.in.W / 2 : a.in.W);
    if (a.in.virt) return conv_lowc(a, opname);
    const int nB = a.pair ? 2 * B : B;       // images in `in`
    if (a.pair) {
      const ConvDef& d = *a.pair;
      if (d.cinp != c.cinp || d.coutp != c.coutp || d.cout != c.cout || d.ks != c.ks || d.stride != c.stride || d.ups_in || c.ups_in || a.ups || a.res || a.aux ||
          a.y_nchw.sp != PS_NULL || a.y_user.sp != PS_NULL || a.y_none || !a.want_stats || c.in_affine != d.in_affine || d.w_off <= c.w_off || d.b_off <= c.b_off ||
          (c.in_affine && (d.g_off <= c.g_off || d.g_off - c.g_off != d.be_off - c.be_off)) || (h->mc16() && d.w16_off <= c.w16_off)) {
        error = opname + ": the two convolutions of a paired launch must have the same shape"; return false;
      }
    }
    if (a.in.Cp != c.cinp) { error = fmt("%s: input channels %d != expected %d", opname.c_str(), a.in.Cp, c.cinp); return false; }
    if (runs_wino(a, Hout, Wout)) return conv_wino(a, opname, Hout, Wout);
    if (a.in.lazy) { error = opname + ": a lazy input (" + a.in.lazy->name + ") needs the Winograd path"; return false; }
    // split-K needs the slab-summing epilogue: float4 channel groups that tile a 256-thread block,
    // and no NCHW side copy
    const bool can_split = (256 % (c.coutp / 4) == 0) && a.y_nchw.sp == PS_NULL && !a.pair && !a.no_split && pol.split_k;
    // the 16-column path serves layers with <= 16 output channels and no residual read
    const bool can_n16 = c.cout <= 16 && !a.res && !h->mc16() && pol.n16;
    Choice ch = choose_variant(h->prec(), c.stride, c.ks, a.ups, false, c.coutp, TB_(), Hout, Wout, c.cinp, can_split, a.aux ? a.aux->cinp : 0, can_n16);
    {
      auto it = h->choices.end();
      for (const std::string& nm : a.choice_names) {
        it = h->choices.find(fmt("%d,%d,%d|%s", TB_(), P->H, P->W, nm.c_str()));
        if (it != h->choices.end()) break;
      }
      if (it == h->choices.end()) it = h->choices.find(fmt("%d,%d,%d|%s", TB_(), P->H, P->W, opname.c_str()));
      if (it != h->choices.end()) {
        const Variant& tv = kVariants[it->second.first];
        const int ts = it->second.second;
        const bool ok = !tv.dma() && !(tv.DMAK && (a.aux || ts != 1)) && !(tv.DMAK && !tv.fn_pro && !tv.fn && (a.pro || a.pro_lrelu)) && tv.BF16 == h->prec() && tv.STRIDE == c.stride && tv.KS == c.ks && tv.UPS == a.ups && !tv.SPADE && c.cinp % tv.BK == 0 &&
                        (!a.aux || a.aux->cinp % tv.BK == 0) && (tv.NF != 0 || (can_n16 && ts == 1)) &&
                        ts >= 1 && ts <= c.cinp / tv.BK && (ts == 1 || can_split);
        if (!ok) { error = fmt("%s: tuned choice (variant %d, ksplit %d) does not fit this layer", opname.c_str(), it->second.first, ts); return false; }
        ch.v = &tv; ch.ksplit = ts;
      }
    }
    const Variant* v = ch.v;
    if (!v) { error = fmt("%s: no kernel variant for Cin=%d stride=%d ks=%d ups=%d", opname.c_str(), c.cinp, c.stride, c.ks, (int)a.ups); return false; }
    const int S = ch.ksplit;
    Op op; op.kind = OP_IGEMM; op.kclass = RIB_KC_IGEMM; op.name = opname; op.var = v;
    IgemmParams& p = op.ip;
    memset(&p, 0, sizeof p);
    p.Hin = a.in.H; p.Win = a.in.W; p.xC = a.in.Cp; p.Cin = c.cinp;
    p.pro_ld = a.pro ? a.pro->ld : 0; p.pro_lrelu = a.pro_lrelu ? 1 : 0;
    p.CoutPad = c.coutp;
    p.Hout = Hout; p.Wout = Wout;
    // phase-decomposed upsample conv: tiles of SOURCE pixels (each yields a 2TH x 2TW output patch)
    p.tilesX = ((a.ups ? a.in.W : Wout) + v->TW() - 1) / v->TW(); p.tilesY = ((a.ups ? a.in.H : Hout) + v->TH() - 1) / v->TH();
    p.xcd_chunk = xcd_chunk_of(p.tilesX * p.tilesY);
    if (a.ups && (!c.ups_in || c.ks != 3 || c.stride != 1)) { error = opname + ": no phase filters for this upsample convolution"; return false; }
    p.act = a.act; p.ksplit = S;
    op.x = WS(a.in.off);
    // heads with 1..4 output channels (conv_img, conv_mask.0): direct convolution on the vector ALUs; the
    // matrix-core kernels would pad N to 16 columns.  y_nchw's channel count is Cout of the conv itself.
    const bool small = c.cout <= 4 && c.ks == 3 && c.stride == 1 && !a.ups && !a.res && !a.aux && !a.want_stats &&
                       c.cinp <= 32 && 256 % (c.cinp / 4) == 0 &&   // halo tile + filter within the default 64 KB of dynamic LDS
                       pol.small_conv;
    if (a.pro) {
      // (a consumer-side finalize in the PROLOGUE was tried and removed: its (scale, shift) table cost every prologue
      // variant 4 KB of LDS - an occupancy step for several of them - to save four launches; the SPADE epilogue keeps its own)
      materialize(*a.pro, opname);
      op.pro_scale = WS(a.pro->sc + a.pro_choff * sizeof(float)); op.pro_shift = WS(a.pro->sh + a.pro_choff * sizeof(float));
    }
    // matrix-core kernels of a bf16 handle read the bf16 filter copies; the direct head convolutions keep fp32 filters
    const bool w16 = h->mc16() && !small;
    op.w = WT(a.ups ? (w16 ? c.wp16_off : c.wp_off) : (w16 ? c.w16_off : c.w_off)); op.bias = WT(c.b_off);
    double aux_flops = 0.0;
    if (a.aux) {
      if (c.ks != 3 || c.stride != 1 || a.ups || !c.fb_off || a.aux->ks != 1 || a.aux->coutp != c.coutp || a.aux_in.Cp != a.aux->cinp ||
          a.aux_in.H != Hout || a.aux_in.W != Wout) { error = opname + ": fused shortcut operand does not fit"; return false; }
      op.x2 = WS(a.aux_in.off); op.w2 = WT(h->mc16() ? a.aux->w16_off : a.aux->w_off); op.bias = WT(c.fb_off);
      p.x2C = a.aux_in.Cp; p.Cin2 = a.aux->cinp;
      aux_flops = 2.0 * a.aux->cin * a.aux->cout * (double)Hout * Wout * B;
    }
    if (a.y_user.sp != PS_NULL) {
      op.y = a.y_user; p.yC = c.cout; p.yoff = 0; p.Cout = c.cout; p.y_f32 = 1;   // a caller's fp32 tensor
    } else if (a.y_none) {
      op.y = PRef(); p.yC = 0; p.yoff = 0; p.Cout = a.cout_store >= 0 ? a.cout_store : c.cout;
    } else {
      op.y = WS(a.out.off); p.yC = a.out.Cp; p.yoff = a.yoff;
      p.Cout = a.cout_store >= 0 ? a.cout_store : h->padc(c.cout);
      if (a.out.H != Hout || a.out.W != Wout) { error = fmt("%s: output size mismatch", opname.c_str()); return false; }
    }
    if (a.pair) {
      const bool w16p = h->mc16();
      p.w_mod = 2;
      p.w_stride = (unsigned)(w16p ? 2 * (a.pair->w16_off - c.w16_off) : a.pair->w_off - c.w_off);      // elements of the filter's storage type
      p.b_stride = (unsigned)(a.pair->b_off - c.b_off);
      p.pair = a.pair_merge ? 1 : 0; p.pair_yoff = a.pair_yoff;
    }
    if (a.res) { op.res = WS(a.res->off); p.resC = a.res->Cp; p.res_ups = a.res_ups ? 1 : 0; }
    op.y_nchw = a.y_nchw;
    if (a.y_none && !(small && c.cout <= 3 && (c.cinp == 16 || c.cinp == 32) && pol.head_conv && a.y_nchw.sp != PS_NULL)) {
      error = opname + ": only the head kernel can run without an NHWC destination"; return false;
    }
    if (small) {
      op.small_co = c.cout;
      op.head = c.cout <= 3 && (c.cinp == 16 || c.cinp == 32) && pol.head_conv;
      // the mask head has every pixel's mask in a register: the driver's blend (evaluator.py:256-258) rides along
      op.fuse_blend = op.head && c.cout == 1 && a.y_user.sp == PS_USER && a.y_user.off == (size_t)U_MASK;
      p.bl_C = h->g.c.image_nc;
      p.ksplit = 1;
      p.tilesX = (Wout + 15) / 16; p.tilesY = (Hout + 15) / 16; p.xcd_chunk = xcd_chunk_of(p.tilesX * p.tilesY);
      op.grid = dim3(p.tilesX * p.tilesY, 1, B);
      op.flops = 2.0 * c.cin * c.ks * c.ks * c.cout * (double)Hout * Wout * B;
      P->flops[RIB_KC_IGEMM] += op.flops;
      push(op);
      return true;
    }
    int tiles = p.tilesX * p.tilesY;
    size_t part_off = 0;
    op.grid = dim3(tiles, v->NF == 0 ? 1 : (c.coutp + v->BN() - 1) / v->BN(), nB * S);
    op.flops = 2.0 * c.cin * c.ks * c.ks * c.cout * (double)Hout * Wout * nB + aux_flops;
    P->flops[RIB_KC_IGEMM] += op.flops;
    if (a.pair && (S != 1 || v->NF == 0)) { error = opname + ": a paired launch cannot split K or use the 16-column path"; return false; }
    if (S == 1) {
      if (a.want_stats) { part_off = alloc((size_t)nB * tiles * 2 * c.coutp * sizeof(double)); op.stat = WS(part_off); }
      push(op);
    } else {
      // split-K: the conv writes raw partial slabs; a second kernel sums them and runs the epilogue
      const size_t slab_off = alloc((size_t)S * B * Hout * Wout * c.coutp * sizeof(float));
      op.slab = WS(slab_off);
      Op e; e.kind = OP_SPLITEPI; e.kclass = RIB_KC_CONVAUX; e.name = opname + ".splitk_sum";
      memset(&e.sp, 0, sizeof e.sp);
      e.sp.ksplit = S; e.sp.B = B; e.sp.CoutPad = c.coutp; e.sp.yC = p.yC; e.sp.yoff = p.yoff; e.sp.Cout = p.Cout;
      e.sp.act = p.act; e.sp.resC = p.resC; e.sp.res_ups = p.res_ups; e.sp.Hout = Hout; e.sp.Wout = Wout;
      const int slots = 256 / (c.coutp / 4), ppb = slots * 4;
      const int blocks = (Hout * Wout + ppb - 1) / ppb;
      e.sp.blocks = blocks;
      e.s_slab = WS(slab_off); e.s_bias = op.bias; e.s_y = op.y; e.s_res = op.res;
      if (a.want_stats) { part_off = alloc((size_t)B * blocks * 2 * c.coutp * sizeof(double)); e.s_stat = WS(part_off); }
      e.grid = dim3(blocks, B, 1);
      tiles = blocks;   // the statistics partials now come from the epilogue kernel's blocks
      op.y = PRef(); op.res = PRef();
      push(op);
      push(e);
    }
    if (a.want_stats) {
      Op f; f.kind = OP_FINALIZE; f.kclass = RIB_KC_STATS; f.name = opname + ".stats";
      memset(&f.fp, 0, sizeof f.fp);
      f.fp.tiles = tiles; f.fp.Cs = c.coutp; f.fp.C = h->padc(c.cout);
      f.fp.ld = a.stats_out->ld; f.fp.off = (int)a.stats_choff;
      f.fp.inv_count = 1.0f / ((float)Hout * (float)Wout); f.fp.eps = 1e-5f;
      f.f_part = WS(part_off);
      if (a.affine) { f.f_gamma = WT(c.g_off); f.f_beta = WT(c.be_off); }
      f.f_scale = WS(a.stats_out->sc); f.f_shift = WS(a.stats_out->sh);
      f.grid = dim3(c.coutp / 16, nB, 1);
      if (a.pair) { f.fp.g_stride = a.affine ? (int)(a.pair->g_off - c.g_off) : 0; f.fp.pair_merge = a.pair_merge ? 1 : 0; f.fp.pair_off = a.pair_yoff; }
      // (a channel offset means two producers share the arrays - the concatenated encoders of the mask network -
      // and consumers would need two partial sources: those keep their launch; so does a paired launch)
      finalize_or_defer(f, a.stats_out, a.stats_choff != 0 || a.stats_now || a.pair != nullptr, part_off, tiles, c.coutp, f.fp.inv_count, a.affine, c.g_off, c.be_off);
    }
    return true;
  }

  // first-layer 3x3 convolution over the caller's NCHW tensors (k_conv_lowc): no packed copy, reduction over the real channels
  bool conv_lowc(const ConvArgs& a, const std::string& opname) {
    const ConvDef& c = *a.cd;
    const int H = a.in.H, W = a.in.W;
    if (!c.wl_off || c.ks != 3 || c.stride != 1 || a.ups || a.pro || a.res || a.aux || a.y_nchw.sp != PS_NULL || a.y_user.sp != PS_NULL ||
        a.in.uc[0] + a.in.uc[1] + a.in.uc[2] != c.cin) { error = opname + ": not a layer k_conv_lowc can run"; return false; }
    if (a.out.H != H || a.out.W != W) { error = fmt("%s: output size mismatch", opname.c_str()); return false; }
    Op op; op.kind = OP_LOWC; op.kclass = RIB_KC_IGEMM; op.name = opname; op.lowc_ce = c.lowc_ce; op.lowc_ncol = c.lowc_ncol;
    memset(&op.lc, 0, sizeof op.lc);
    LowcParams& p = op.lc;
    p.c0 = a.in.uc[0]; p.c1 = a.in.uc[1]; p.c2 = a.in.uc[2];
    p.H = H; p.W = W; p.yC = a.out.Cp; p.yoff = a.yoff; p.Cout = a.cout_store >= 0 ? a.cout_store : h->padc(c.cout);
    p.act = a.act; p.CoutPad = c.coutp;
    // 8x16 tiles (2048 workgroups at 512x512: two rounds, whose gather / MFMA / store phases overlap) win where the layer is
    // store-bound and has no statistics (conv_first 44.6 -> 38.4 us); with statistics the finalize of twice the partials
    // eats the gain (down_lbl.0 -3 +3 us) and the 16-column layer loses (29.5 -> 32.7)
    const int tw = pol.lowc_tw ? pol.lowc_tw : ((c.lowc_ncol == 64 && !a.want_stats) ? 16 : 32);
    op.lowc_tw = tw;
    p.tilesX = (W + tw - 1) / tw; p.tilesY = (H + 7) / 8;
    if (p.Cout > c.lowc_ncol || c.cout > c.lowc_ncol) { error = opname + ": more output columns than the k_conv_lowc instantiation has"; return false; }
    op.lc_s0 = US(a.in.usrc[0]); if (p.c1) op.lc_s1 = US(a.in.usrc[1]); if (p.c2) op.lc_s2 = US(a.in.usrc[2]);
    op.lc_w = WT(c.wl_off); op.lc_bias = WT(c.b_off); op.lc_y = WS(a.out.off);
    const int tiles = p.tilesX * p.tilesY;
    size_t part_off = 0;
    if (a.want_stats) { part_off = alloc((size_t)B * tiles * 2 * c.coutp * sizeof(double)); op.lc_stat = WS(part_off); }
    op.grid = dim3(tiles, B, 1);
    op.flops = 2.0 * c.cin * 9.0 * c.cout * (double)H * W * B;
    P->flops[RIB_KC_IGEMM] += op.flops;
    push(op);
    if (a.want_stats) {
      Op f; f.kind = OP_FINALIZE; f.kclass = RIB_KC_STATS; f.name = opname + ".stats";
      memset(&f.fp, 0, sizeof f.fp);
      f.fp.tiles = tiles; f.fp.Cs = c.coutp; f.fp.C = h->padc(c.cout);
      f.fp.ld = a.stats_out->ld; f.fp.off = (int)a.stats_choff;
      f.fp.inv_count = 1.0f / ((float)H * (float)W); f.fp.eps = 1e-5f;
      f.f_part = WS(part_off);
      if (a.affine) { f.f_gamma = WT(c.g_off); f.f_beta = WT(c.be_off); }
      f.f_scale = WS(a.stats_out->sc); f.f_shift = WS(a.stats_out->sh);
      f.grid = dim3(c.coutp / 16, B, 1);
      finalize_or_defer(f, a.stats_out, a.stats_choff != 0 || a.stats_now, part_off, tiles, c.coutp, f.fp.inv_count, a.affine, c.g_off, c.be_off);
    }
    return true;
  }

  bool conv_wino(const ConvArgs& a, const std::string& opname, int Hout, int Wout) {
    const ConvDef& c = *a.cd;
    // F(4x4, 3x3) (36 positions, 1/4 of the multiplications) where its per-position GEMM still has >= 256 rows (maps of
    // 64x64 and more at batch 1); below that the 36 GEMMs are filter-streaming-bound (512x512 filters x 36 = 38 MB for 64
    // rows) and F(2x2, 3x3) (16 positions, 4/9) is faster.  Measured per layer at 512x512, transforms included:
    // 256->256 at 64x64 42.4 -> 37.3 us, 512->256 at 64x64 67.4 -> 56.2; 512->512 at 32x32 38.3 -> 41.0 (kept on F(2x2)).
    // RIB_WINO_M = 2 / 4 forces one of them.
    const int wino_force = pol.wino_force;
    const int wm = wino_force == 2 || wino_force == 4 ? wino_force : ((long)TB_() * ((Hout + 3) / 4) * ((Wout + 3) / 4) >= 256 ? 4 : 2);
    const int NP = (wm + 2) * (wm + 2);      // output tile edge, Winograd positions
    const int tilesY = (Hout + wm - 1) / wm, tilesX = (Wout + wm - 1) / wm, ntiles = tilesY * tilesX;
    const std::string gname = opname + (wm == 2 ? ".wino" : ".wino4");
    const size_t v_off = alloc((size_t)B * NP * ntiles * c.cinp * sizeof(float));
    const size_t m_off = alloc((size_t)B * NP * ntiles * c.coutp * sizeof(float));
    {   // input transform (with the convolution's prologue)
      Op op; op.kind = OP_WINO_IN; op.kclass = RIB_KC_CONVAUX; op.name = opname + ".wino_in"; op.for_op = gname; op.wino_m = wm;
      memset(&op.wi, 0, sizeof op.wi);
      op.wi.H = a.in.H; op.wi.W = a.in.W; op.wi.xC = a.in.Cp; op.wi.Cin = c.cinp; op.wi.tilesY = tilesY; op.wi.tilesX = tilesX;
      op.wi.pro_lrelu = a.pro_lrelu ? 1 : 0;
      op.wi_x = WS(a.in.off); op.wi_v = WS(v_off);
      if (a.in.lazy) {
        const LazySrc& L = *a.in.lazy;
        if (a.pro || a.pro_lrelu) { error = gname + ": a lazy input carries its own prologue"; return false; }
        op.wi_mode = L.mode;
        op.wi_x = WS(L.x.off); op.wi.xC = L.x.Cp;
        op.wi.pro_ld = L.nx.ld;
        if (!take_partials(L.nx, op.wi.st, op, 0)) { materialize(L.nx, gname); op.wi_sc = WS(L.nx.sc); op.wi_sh = WS(L.nx.sh); }
        if (L.mode == WSRC_SPADE) {
          op.wi.x_ups = L.x_ups ? 1 : 0; op.wi.pro_lrelu = L.lrelu ? 1 : 0;
          op.wi_slab = WS(L.slab_off); op.wi.slab_ld = L.slab_ld; op.wi.col0 = L.col0; op.wi_sbias = WT(L.sbias_off);
        } else {
          if (L.has2) {
            op.wi_x2 = WS(L.x2.off);
            if (L.n2.ld != L.nx.ld) { error = gname + ": join operands with different statistics rows"; return false; }
            if (!take_partials(L.n2, op.wi.st2, op, 1)) { materialize(L.n2, gname); op.wi_sc2 = WS(L.n2.sc); op.wi_sh2 = WS(L.n2.sh); }
          } else op.wi_xres = WS(L.xres.off);
          op.wi_o = WS(L.o.off);
        }
      } else if (a.pro && !(a.pro_choff == 0 && take_partials(*a.pro, op.wi.st, op, 0))) {
        materialize(*a.pro, gname);
        op.wi.pro_ld = a.pro->ld;
        op.wi_sc = WS(a.pro->sc + a.pro_choff * sizeof(float)); op.wi_sh = WS(a.pro->sh + a.pro_choff * sizeof(float));
      }
      // workgroup = (16 (tile, transformed row) units per pass) x (slice of 64 channels)
      const int nsl = (c.cinp + 63) / 64, units = ntiles * (wm + 2);
      // (every workgroup that reduces partials re-reads tiles x 64 channels x 16 bytes: few, fatter workgroups then)
      op.wi.nslices = nsl; op.wi.ublocks = std::max(1, std::min((units + 15) / 16, (op.wi.st.tiles > 0 ? 512 : 4096) / nsl));
      op.grid = dim3(op.wi.ublocks * nsl, B, 1);
      push(op);
    }
    {   // the 16 / 36 GEMMs as one 1x1 "convolution" of NP*B samples of a tilesY x tilesX image, one filter set per position
      // k_gemm_dma (operands staged by LDS-DMA) unless a tuned choice names a k_igemm 1x1 variant; TB_() decides as for every choice
      Choice ch;
      ch.v = pick_gemm_dma(h->prec(), (long)TB_() * NP, ntiles, c.coutp, c.cinp);
      if (!ch.v) ch = choose_variant(h->prec(), 1, 1, false, false, c.coutp, TB_() * NP, tilesY, tilesX, c.cinp, false, 0, false);
      auto it = h->choices.find(fmt("%d,%d,%d|%s", TB_(), P->H, P->W, gname.c_str()));
      if (it != h->choices.end()) {
        const Variant& tv = kVariants[it->second.first];
        if (tv.BF16 != h->prec() || tv.KS != 1 || tv.STRIDE != 1 || tv.UPS || tv.SPADE || tv.NF == 0 || c.cinp % tv.BK != 0 || it->second.second != 1) {
          error = fmt("%s: tuned choice (variant %d, ksplit %d) does not fit this layer", gname.c_str(), it->second.first, it->second.second); return false;
        }
        ch.v = &tv; ch.ksplit = 1;
      }
      const Variant* v = ch.v;
      if (!v) { error = gname + ": no 1x1 kernel variant"; return false; }
      const int set = ensure_wino_set(h, (int)(&c - h->convs.data()), wm);
      if (set < 0) { error = gname + ": " + h->err; return false; }
      Op op; op.kind = OP_IGEMM; op.kclass = RIB_KC_IGEMM; op.name = gname; op.var = v; op.wino = true; op.wino_m = wm;
      op.flops = 2.0 * c.cin * 9.0 * c.cout * (double)Hout * Wout * B;      // the convolution's algorithmic count (executed: 4/9 or 1/4 of it)
      if (v->dma()) {
        // M[n*NP + xi] = V[n*NP + xi] . U[xi]^T: Z = B*NP problems of ntiles x coutp x cinp
        op.kind = OP_GEMM;
        GemmDmaParams& g = op.gp;
        memset(&g, 0, sizeof g);
        g.M = ntiles; g.N = c.coutp; g.K = c.cinp; g.lda = c.cinp; g.ldc = c.coutp;
        g.sA = (size_t)ntiles * c.cinp; g.sB = (size_t)c.coutp * c.cinp; g.sC = (size_t)ntiles * c.coutp; g.modB = NP;
        op.g_a = WS(v_off); op.g_b = PRef(); op.g_b.sp = PS_WINO; op.g_b.off = (size_t)set; op.g_c = WS(m_off);
        op.grid = dim3((ntiles + v->BM() - 1) / v->BM(), (c.coutp + v->BN() - 1) / v->BN(), B * NP);
        P->flops[RIB_KC_IGEMM] += op.flops;
        push(op);
      } else {
      IgemmParams& p = op.ip;
      memset(&p, 0, sizeof p);
      p.Hin = tilesY; p.Win = tilesX; p.xC = c.cinp; p.Cin = c.cinp; p.CoutPad = c.coutp; p.Hout = tilesY; p.Wout = tilesX;
      p.tilesX = (tilesX + v->TW() - 1) / v->TW(); p.tilesY = (tilesY + v->TH() - 1) / v->TH(); p.xcd_chunk = xcd_chunk_of(p.tilesX * p.tilesY);
      p.act = ACT_NONE; p.ksplit = 1; p.yC = c.coutp; p.yoff = 0; p.Cout = c.coutp;
      p.w_mod = NP; p.w_stride = (unsigned)((size_t)c.coutp * c.cinp);
      op.x = WS(v_off); op.w = PRef(); op.w.sp = PS_WINO; op.w.off = (size_t)set; op.bias = WT(c.zero_off); op.y = WS(m_off);
      op.grid = dim3(p.tilesX * p.tilesY, (c.coutp + v->BN() - 1) / v->BN(), B * NP);
      P->flops[RIB_KC_IGEMM] += op.flops;
      push(op);
      }
    }
    {   // output transform + the convolution's epilogue
      Op op; op.kind = OP_WINO_OUT; op.kclass = RIB_KC_CONVAUX; op.name = opname + ".wino_out"; op.for_op = gname; op.wino_m = wm;
      memset(&op.wo, 0, sizeof op.wo);
      // workgroup = (16 (tile, output row) units per pass) x (slice of 64 channels); ONE statistics partial per unit block,
      // at most STATS_MAX_PARTIALS of them, so the consumer of the normalised tensor can reduce them itself
      const int nsl = (c.coutp + 63) / 64, units = ntiles * wm;
      const int blocks = std::max(1, std::min((units + 15) / 16, (int)STATS_MAX_PARTIALS));
      op.wo.tilesY = tilesY; op.wo.tilesX = tilesX; op.wo.CoutPad = c.coutp; op.wo.ublocks = blocks; op.wo.nslices = nsl;
      op.wo.yC = a.out.Cp; op.wo.yoff = a.yoff; op.wo.Cout = a.cout_store >= 0 ? a.cout_store : h->padc(c.cout);
      op.wo.Hout = Hout; op.wo.Wout = Wout; op.wo.act = a.act;
      if (a.out.H != Hout || a.out.W != Wout) { error = fmt("%s: output size mismatch", opname.c_str()); return false; }
      op.wo_m = WS(m_off); op.wo_bias = WT(c.b_off); op.wo_y = WS(a.out.off);
      if (a.res) { op.wo_res = WS(a.res->off); op.wo.resC = a.res->Cp; }
      size_t part_off = 0;
      if (a.want_stats) { part_off = alloc((size_t)B * blocks * 2 * c.coutp * sizeof(double)); op.wo_stat = WS(part_off); }
      op.grid = dim3(blocks * nsl, B, 1);
      push(op);
      if (a.want_stats) {
        Op f; f.kind = OP_FINALIZE; f.kclass = RIB_KC_STATS; f.name = opname + ".stats";
        memset(&f.fp, 0, sizeof f.fp);
        f.fp.tiles = blocks; f.fp.Cs = c.coutp; f.fp.C = h->padc(c.cout);
        f.fp.ld = a.stats_out->ld; f.fp.off = (int)a.stats_choff;
        f.fp.inv_count = 1.0f / ((float)Hout * (float)Wout); f.fp.eps = 1e-5f;
        f.f_part = WS(part_off);
        if (a.affine) { f.f_gamma = WT(c.g_off); f.f_beta = WT(c.be_off); }
        f.f_scale = WS(a.stats_out->sc); f.f_shift = WS(a.stats_out->sh);
        f.grid = dim3(c.coutp / 16, B, 1);
        finalize_or_defer(f, a.stats_out, a.stats_choff != 0 || a.stats_now, part_off, blocks, c.coutp, f.fp.inv_count, a.affine, c.g_off, c.be_off);
      }
    }
    return true;
  }

  // ---- gamma/beta of EVERY SPADE of a condition level in one 1x1 GEMM (round 3) -----------------------------------------
  // The gamma/beta of a SPADE depend only on the condition map.  On the small deep maps (<= 64x64 at 512x512, batch 1) each
  // SPADE's own GEMM is a 10-30 us launch of which 10-13 us are ramp and drain, and the ten of them
  // that ran unfused were ten such launches.  The filter matrices of a level's SPADE groups lie back to back in the blob,
  // so ONE k_igemm 1x1 launch on cond[level] with N = sum of their columns (8192 at level 4, 2048 at level 3 of HSM.yaml:
  // 8.6 GFLOP each) writes a slab [B][H*W][N] and every SPADE of the level is then only its k_spade_modulate on its own
  // column range.  Levels whose map is larger stay fused (gamma/beta never touch memory there).
  bool cond_level_gemm(int level, const Act& cond) {
    const long max_px = pol.cond_gemm_max_px;   // 0: off
    const auto it = h->level_groups.find(level);
    if (it == h->level_groups.end() || it->second.size() < 2 || (long)TB_() * cond.H * cond.W > max_px) return true;
    int N = 0; double fl = 0.0;
    for (int gi : it->second) {
      const SpadeGroup& sg = h->spades[gi];
      if (sg.col0 != N || h->padc(sg.cond) != cond.Cp) { error = fmt("cond level %d: SPADE group %s does not continue the level's filter matrix", level, sg.key.c_str()); return false; }
      N += sg.npad;
      fl += 2.0 * sg.cond * 2.0 * sg.nsets * sg.C * (double)cond.H * cond.W * B;
    }
    const SpadeGroup& first = h->spades[it->second[0]];
    const std::string name = fmt("cond_%d.gammabeta", level);
    Choice ch;
    ch.v = pick_gemm_dma(h->prec(), TB_(), cond.H * cond.W, N, cond.Cp);
    if (!ch.v) ch = choose_variant(h->prec(), 1, 1, false, false, N, TB_(), cond.H, cond.W, cond.Cp, false);
    {
      auto ct = h->choices.find(fmt("%d,%d,%d|%s", TB_(), P->H, P->W, name.c_str()));
      if (ct != h->choices.end()) {
        const Variant& tv = kVariants[ct->second.first];
        if (tv.BF16 != h->prec() || tv.KS != 1 || tv.STRIDE != 1 || tv.UPS || tv.SPADE || tv.NF == 0 || cond.Cp % tv.BK != 0 || ct->second.second != 1) {
          error = fmt("%s: tuned choice (variant %d, ksplit %d) does not fit this launch", name.c_str(), ct->second.first, ct->second.second); return false;
        }
        ch.v = &tv; ch.ksplit = 1;
      }
    }
    const Variant* v = ch.v;
    if (!v) { error = name + ": no 1x1 kernel variant"; return false; }
    const size_t slab_off = alloc((size_t)B * cond.H * cond.W * N * sizeof(float));
    Op op; op.kind = OP_IGEMM; op.kclass = RIB_KC_SPADE; op.name = name; op.var = v;
    op.flops = fl;
    if (v->dma()) {
      // slab[b] = cond[b] . W_level^T: B problems of (H*W) x N x Cp
      op.kind = OP_GEMM;
      GemmDmaParams& g = op.gp;
      memset(&g, 0, sizeof g);
      g.M = cond.H * cond.W; g.N = N; g.K = cond.Cp; g.lda = cond.Cp; g.ldc = N;
      g.sA = (size_t)g.M * cond.Cp; g.sB = 0; g.sC = (size_t)g.M * N; g.modB = 0;
      op.g_a = WS(cond.off); op.g_b = WT(h->mc16() ? first.w16_off : first.w_off); op.g_c = WS(slab_off);
      op.grid = dim3((g.M + v->BM() - 1) / v->BM(), (N + v->BN() - 1) / v->BN(), B);
      P->flops[RIB_KC_SPADE] += fl;
      push(op);
      LevelSlab ls; ls.off = slab_off; ls.ld = N;
      level_slab[level] = ls;
      return true;
    }
    IgemmParams& p = op.ip;
    memset(&p, 0, sizeof p);
    p.Hin = cond.H; p.Win = cond.W; p.xC = cond.Cp; p.Cin = cond.Cp;
    p.CoutPad = N; p.Hout = cond.H; p.Wout = cond.W; p.ksplit = 1;
    p.tilesX = (cond.W + v->TW() - 1) / v->TW(); p.tilesY = (cond.H + v->TH() - 1) / v->TH(); p.xcd_chunk = xcd_chunk_of(p.tilesX * p.tilesY);
    op.x = WS(cond.off); op.w = WT(h->mc16() ? first.w16_off : first.w_off); op.bias = WT(first.b_off); op.slab = WS(slab_off);
    op.grid = dim3(p.tilesX * p.tilesY, (N + v->BN() - 1) / v->BN(), B);
    op.flops = fl;
    P->flops[RIB_KC_SPADE] += fl;
    push(op);
    LevelSlab ls; ls.off = slab_off; ls.ld = N;
    level_slab[level] = ls;
    return true;
  }

  // ---- SPADE launch: ys0 (= lrelu(mod0(x))), optional ys1 (= mods(x), no activation) -------
  // lazy_ok: the only consumer of ys0 is a convolution that runs in the Winograd domain - when this SPADE is just a modulate
  // of the level's gamma/beta slab (one set), ys0 is not stored: the convolution's input transform computes it (LazySrc)
  bool spade(const std::string& key, const Act& cond, const Act& x, bool x_ups, const Norm& nx,
             Act* ys0, Act* ys1, bool act0, bool lazy_ok = false) {
    const SpadeGroup& sg = h->spades[h->spade_index.at(key)];
    const int Hout = x_ups ? x.H * 2 : x.H, Wout = x_ups ? x.W * 2 : x.W;
    if (cond.H != Hout || cond.W != Wout) { error = fmt("%s: cond map %dx%d != %dx%d (SPADE resize must be the identity)", key.c_str(), cond.H, cond.W, Hout, Wout); return false; }
    if (cond.Cp != h->padc(sg.cond) || cond.Cp % 32 != 0) { error = fmt("%s: cond channels %d unsupported (need a multiple of 32)", key.c_str(), cond.Cp); return false; }
    if (x.Cp != sg.Cp) { error = fmt("%s: x channels %d != %d", key.c_str(), x.Cp, sg.Cp); return false; }
    // Fused (one kernel: gamma/beta GEMM + modulate epilogue) where the map is large; UNFUSED on the
    // small deep maps, where the fused kernel is one long K chain on a handful of workgroups: the
    // GEMM runs as a split-K 1x1 convolution into partial slabs and k_spade_modulate finishes.
    const Variant* v = choose_variant(h->prec(), 1, 1, false, true, sg.npad, TB_(), Hout, Wout, cond.Cp, false).v;
    if (sg.w1_off)      // 16 modulated channels: the one-fragment layout halves the matrix work (first fitting NF = 1 variant)
      for (int i = 0; i < kNumVariants; ++i) {
        const Variant& t = kVariants[i];
        if (t.SPADE && t.NF == 1 && t.BF16 == h->prec() && t.KW == 1 && t.TB == 1 && cond.Cp % t.BK == 0) { v = &t; break; }
      }
    Choice uf;   // unfused candidate
    const bool small_map = (long)Hout * Wout * TB_() <= 4096 && pol.unfused_spade;
    if (small_map) uf = choose_variant(h->prec(), 1, 1, false, false, sg.npad, TB_(), Hout, Wout, cond.Cp, true);
    bool unfused = small_map && uf.v != nullptr;
    {
      auto it = h->choices.find(fmt("%d,%d,%d|%s", TB_(), P->H, P->W, (key + ".spade").c_str()));
      if (it != h->choices.end()) {
        const Variant& tv = kVariants[it->second.first];
        const int ts = it->second.second;
        if (tv.dma() || tv.BF16 != h->prec() || tv.KS != 1 || tv.STRIDE != 1 || tv.UPS || cond.Cp % tv.BK != 0 || ts < 1 || ts > cond.Cp / tv.BK || (tv.SPADE && ts != 1) ||
            (tv.SPADE && tv.NF == 1 && !sg.w1_off)) {
          error = key + ": tuned SPADE choice does not fit"; return false;
        }
        if (tv.SPADE) { v = &tv; unfused = false; }
        else { uf.v = &tv; uf.ksplit = ts; unfused = true; }
      }
    }
    // the level's gamma/beta GEMM already ran (cond_level_gemm): this SPADE is only its modulate
    const auto lvl = level_slab.find(sg.level);
    const bool from_level = lvl != level_slab.end();
    if (from_level) unfused = true;
    if (!unfused && !v) { error = "no SPADE variant"; return false; }
    const bool no_lazy = !pol.lazy_sources;
    if (from_level && lazy_ok && sg.nsets == 1 && !h->keep_taps && !no_lazy && h->prec() == PREC_F32) {
      auto L = std::make_shared<LazySrc>();
      L->mode = WSRC_SPADE; L->name = key + ".spade.modulate"; L->x = x; L->nx = nx; L->x_ups = x_ups; L->lrelu = act0;
      L->slab_off = lvl->second.off; L->slab_ld = lvl->second.ld; L->col0 = sg.col0; L->sbias_off = sg.b_off;
      Act y; y.C = sg.C; y.Cp = sg.Cp; y.H = Hout; y.W = Wout; y.lazy = L;
      *ys0 = y;
      return true;
    }
    if (unfused) {
      *ys0 = act(sg.C, Hout, Wout);
      if (sg.nsets == 2) *ys1 = act(sg.C, Hout, Wout);
      int S = 1, slab_ld = 0, col0 = 0;
      size_t slab_off = 0;
      if (from_level) {
        slab_off = lvl->second.off; slab_ld = lvl->second.ld; col0 = sg.col0;
      } else {
        const Variant* cv = uf.v; S = uf.ksplit;
        slab_off = alloc((size_t)S * B * Hout * Wout * sg.npad * sizeof(float)); slab_ld = sg.npad;
        Op op; op.kind = OP_IGEMM; op.kclass = RIB_KC_SPADE; op.name = key + ".spade"; op.var = cv;
        IgemmParams& p = op.ip;
        memset(&p, 0, sizeof p);
        p.Hin = cond.H; p.Win = cond.W; p.xC = cond.Cp; p.Cin = cond.Cp;
        p.CoutPad = sg.npad; p.Hout = Hout; p.Wout = Wout; p.ksplit = S;
        p.tilesX = (Wout + cv->TW() - 1) / cv->TW(); p.tilesY = (Hout + cv->TH() - 1) / cv->TH(); p.xcd_chunk = xcd_chunk_of(p.tilesX * p.tilesY);
        op.x = WS(cond.off); op.w = WT(h->mc16() ? sg.w16_off : sg.w_off); op.bias = WT(sg.b_off); op.slab = WS(slab_off);
        op.grid = dim3(p.tilesX * p.tilesY, (sg.npad + cv->BN() - 1) / cv->BN(), B * S);
        op.flops = 2.0 * sg.cond * 2.0 * sg.nsets * sg.C * (double)Hout * Wout * B;
        P->flops[RIB_KC_SPADE] += op.flops;
        push(op);
      }
      Op mo; mo.kind = OP_MODULATE; mo.kclass = RIB_KC_ELTWISE; mo.name = key + ".spade.modulate";
      memset(&mo.mp, 0, sizeof mo.mp);
      mo.mp.ksplit = S; mo.mp.B = B; mo.mp.slab_ld = slab_ld; mo.mp.col0 = col0; mo.mp.xmC = x.Cp; mo.mp.xm_ups = x_ups ? 1 : 0;
      mo.mp.m_ld = nx.ld; mo.mp.C = sg.Cp; mo.mp.nsets = sg.nsets; mo.mp.act0 = act0 ? ACT_LRELU : ACT_NONE; mo.mp.act1 = ACT_NONE;
      mo.mp.Hout = Hout; mo.mp.Wout = Wout;
      mo.m_slab = WS(slab_off); mo.m_bias = WT(sg.b_off); mo.m_xm = WS(x.off);
      // a slice of 64 virtual channels lies inside one set when C is a multiple of 64: the modulate can then reduce the
      // producer's partials of its own channels
      if (!(sg.Cp % 64 == 0 && nx.pend && !nx.pend->affine && sg.Cp <= nx.pend->Cs && take_partials(nx, mo.mp.st, mo, 0))) {
        materialize(nx, key + ".spade");
        mo.m_sc = WS(nx.sc); mo.m_sh = WS(nx.sh);
      }
      mo.m_ys0 = WS(ys0->off); if (sg.nsets == 2) mo.m_ys1 = WS(ys1->off);
      const int nsl = (sg.nsets * sg.Cp + 63) / 64;
      mo.mp.nslices = nsl; mo.mp.pblocks = std::max(1, std::min((Hout * Wout + 15) / 16, (mo.mp.st.tiles > 0 ? 384 : 2048) / nsl));
      mo.grid = dim3(mo.mp.pblocks * nsl, B, 1);
      push(mo);
      return true;
    }
    *ys0 = act(sg.C, Hout, Wout);
    if (sg.nsets == 2) *ys1 = act(sg.C, Hout, Wout);
    Op op; op.kind = OP_IGEMM; op.kclass = RIB_KC_SPADE; op.name = key + ".spade"; op.var = v;
    IgemmParams& p = op.ip;
    memset(&p, 0, sizeof p);
    p.Hin = cond.H; p.Win = cond.W; p.xC = cond.Cp; p.Cin = cond.Cp;
    const bool one_frag = v->NF == 1;      // [gamma(16) | beta(16)] layout
    const int ncols = one_frag ? 32 : sg.npad;
    p.CoutPad = ncols; p.Hout = Hout; p.Wout = Wout;
    p.tilesX = (Wout + v->TW() - 1) / v->TW(); p.tilesY = (Hout + v->TH() - 1) / v->TH(); p.xcd_chunk = xcd_chunk_of(p.tilesX * p.tilesY);
    p.ksplit = 1;
    p.xmC = x.Cp; p.xm_ups = x_ups ? 1 : 0; p.m_ld = nx.ld; p.C = sg.Cp; p.nsets = sg.nsets;
    p.act0 = act0 ? ACT_LRELU : ACT_NONE; p.act1 = ACT_NONE;
    op.x = WS(cond.off); op.w = WT(h->mc16() ? sg.w16_off : sg.w_off); op.bias = WT(sg.b_off);
    if (one_frag) { op.w = WT(sg.w1_off); op.bias = WT(sg.b1_off); }
    op.xm = WS(x.off);
    if (has_partials(nx) && !nx.pend->affine && sg.Cp <= nx.pend->Cs) {   // consumer-side finalize in the SPADE epilogue
      const PendingStats& ps = *nx.pend;
      op.m_part = WS(ps.part_off); p.m_tiles = ps.tiles; p.m_Cs = ps.Cs; p.m_inv = ps.inv_count;
    } else {
      materialize(nx, key + ".spade");
      op.m_scale = WS(nx.sc); op.m_shift = WS(nx.sh);
    }
    op.ys0 = WS(ys0->off); if (sg.nsets == 2) op.ys1 = WS(ys1->off);
    op.grid = dim3(p.tilesX * p.tilesY, (ncols + v->BN() - 1) / v->BN(), B);
    op.flops = 2.0 * sg.cond * 2.0 * sg.nsets * sg.C * (double)Hout * Wout * B;
    P->flops[RIB_KC_SPADE] += op.flops;
    push(op);
    return true;
  }

  // ---- Res2dBlock 'NACNAC' (PGNR/models/layers/residual.py:129-151) -------------------------
  bool spade_block(const std::string& name, const Act& x, bool x_ups, const Norm& nx, const Act& cond,
                   Act* out, Norm* nout) {
    const ConvDef& c0 = conv_of(h, name + ".conv_block_0");
    const ConvDef& c1 = conv_of(h, name + ".conv_block_1");
    const bool learned = h->conv_index.count(name + ".conv_block_s") > 0;
    const int Hout = x_ups ? x.H * 2 : x.H, Wout = x_ups ? x.W * 2 : x.W;
    Act ys0, ys1;
    Act hbuf = act(c0.cout, Hout, Wout);
    Norm nh = norm(hbuf.Cp);
    { ConvArgs a; a.cd = &c0; a.out = hbuf; a.want_stats = true; a.stats_out = &nh;
      if (!spade(name + ".0", cond, x, x_ups, nx, &ys0, &ys1, true, runs_wino(a, Hout, Wout))) return false;
      if (!ys0.lazy) tap(name + ".ys0", ys0);
      a.in = ys0;
      if (!conv(a, name + ".conv_block_0")) return false; }
    tap(name + ".h", hbuf);
    Act y1, dummy;
    {
      // conv_block_1 runs in the Winograd domain only with an identity shortcut at the same resolution (the res blocks)
      ConvArgs t; t.cd = &c1; t.res = &x; t.res_ups = !learned && x_ups;
      const bool lazy1 = !learned && runs_wino(t, Hout, Wout);
      if (!spade(name + ".1", cond, hbuf, false, nh, &y1, &dummy, true, lazy1)) return false;
    }
    if (!y1.lazy) tap(name + ".y1", y1);
    // learned shortcut (residual.py:98-108): its 1x1 convolution on SPADE_s(x) is fused into
    // conv_block_1's launch as extra K chunks accumulating into the same output tile
    const bool fuse_s = learned && pol.fuse_shortcut;
    Act outs;
    if (learned && !fuse_s) {
      const ConvDef& cs = conv_of(h, name + ".conv_block_s");
      outs = act(cs.cout, Hout, Wout);
      ConvArgs a; a.cd = &cs; a.in = ys1; a.out = outs;
      if (!conv(a, name + ".conv_block_s")) return false;
    }
    *out = act(c1.cout, Hout, Wout);
    { ConvArgs a; a.cd = &c1; a.in = y1; a.out = *out;
      if (fuse_s) { a.aux = &conv_of(h, name + ".conv_block_s"); a.aux_in = ys1; }
      else a.res = learned ? &outs : &x;
      a.res_ups = !learned && x_ups;   // identity shortcut of an upsampled input: x_up[y][x] = x[y>>1][x>>1]
      if (nout) { *nout = norm(out->Cp); a.want_stats = true; a.stats_out = nout; }
      if (!conv(a, name + ".conv_block_1")) return false; }
    tap(name, *out);
    return true;
  }

  // one of the two stride-2 encoders of the mask network (generator.py:449-459); its last level
  // lands in half of the concatenated tensor and of the concatenated (scale, shift) arrays
  bool mask_branch(int b, const Act& input, const Act& CAT, const Norm& ncat, int chm) {
    const rib_config& c = h->g.c;
    const std::string m = "flow_network_temp";
    const char* branches[2] = {"down_lbl", "down_img"};
    Act cur = input; Norm ncur; bool have = false;
    for (int i = 0; i <= c.mask_down; ++i) {
      const ConvDef& cd = conv_of(h, m + "." + branches[b] + "." + std::to_string(i));
      const bool lastl = (i == c.mask_down);
      Act o = lastl ? CAT : act(cd.cout, i == 0 ? cur.H : cur.H / 2, i == 0 ? cur.W : cur.W / 2);
      Norm no = lastl ? ncat : norm(h->padc(cd.cout));
      ConvArgs a; a.cd = &cd; a.in = cur; a.out = o;
      if (have) { a.pro = &ncur; a.pro_lrelu = true; }
      if (i >= 1) {      // same kernel choice as the paired launch of this level (mask_branches_paired): bit-identical results
        a.no_split = true;
        a.choice_names = {m + ".down_pair." + std::to_string(i), m + ".down_img." + std::to_string(i)};
      }
      // the last level's (scale, shift) land in the shared arrays of the concatenated tensor: its finalize is emitted at
      // the producer (`no` is a local copy of ncat: a deferred finalize attached to it would be lost)
      if (lastl) { a.yoff = b * chm; a.stats_choff = (size_t)b * chm; a.stats_now = true; }
      a.want_stats = true; a.stats_out = &no; a.affine = true;
      if (!conv(a, cd.name)) return false;
      if (!lastl) tap(std::string("mask.") + (b == 0 ? "lbl_" : "img_") + std::to_string(i) + ".raw", o);
      cur = o; ncur = no; have = true;
    }
    return true;
  }

  // Both stride-2 encoders of the mask network, level by level (round 3): level i >= 1 of the label encoder and of the image
  // encoder are the same convolution on different tensors with different filters (32 -> 64 -> 128 -> 256 channels at
  // HSM.yaml's widths, 2.4 GFLOP each), ~35 us launches that cannot fill the chip one at a time: ONE launch computes both
  // (IgemmParams::pair).  Level 0 reads different inputs (22 / 9 channels): two launches into the two halves of a 2-image
  // buffer.  Batch 1 only; a chain keeps the encoders apart (its label encoder runs once per segment at batch T).
  bool mask_branches_paired(const Act& L, const Act& I9, const Act& CAT, const Norm& ncat, int chm) {
    const rib_config& c = h->g.c;
    const std::string m = "flow_network_temp";
    const Act* inputs[2] = {&L, &I9};
    const char* branches[2] = {"down_lbl", "down_img"};
    Act cur; Norm ncur;
    {
      const ConvDef& c0 = conv_of(h, m + ".down_lbl.0");
      cur = act_n(2, c0.cout, L.H, L.W); ncur = norm(h->padc(c0.cout), 2);
      const size_t img_bytes = (size_t)L.H * L.W * cur.Cp * h->esz();
      for (int b = 0; b < 2; ++b) {
        const ConvDef& cd = conv_of(h, m + "." + branches[b] + ".0");
        Act o = cur; o.off += b * img_bytes;                                   // image b of the pair buffer
        Norm no = ncur; no.sc += (size_t)b * ncur.ld * sizeof(float); no.sh += (size_t)b * ncur.ld * sizeof(float);
        ConvArgs a; a.cd = &cd; a.in = *inputs[b]; a.out = o; a.want_stats = true; a.stats_out = &no; a.affine = true; a.stats_now = true;
        if (!conv(a, cd.name)) return false;
        tap(std::string("mask.") + (b == 0 ? "lbl_0" : "img_0") + ".raw", o);
      }
    }
    for (int i = 1; i <= c.mask_down; ++i) {
      const ConvDef& cl = conv_of(h, m + ".down_lbl." + std::to_string(i));
      const ConvDef& ci = conv_of(h, m + ".down_img." + std::to_string(i));
      const bool lastl = (i == c.mask_down);
      Act o = lastl ? CAT : act_n(2, cl.cout, cur.H / 2, cur.W / 2);
      Norm no = lastl ? ncat : norm(h->padc(cl.cout), 2);
      ConvArgs a; a.cd = &cl; a.pair = &ci; a.in = cur; a.out = o; a.pro = &ncur; a.pro_lrelu = true;
      if (lastl) { a.pair_merge = true; a.pair_yoff = chm; }
      a.want_stats = true; a.stats_out = &no; a.affine = true; a.no_split = true;
      a.choice_names = {m + ".down_pair." + std::to_string(i), m + ".down_img." + std::to_string(i)};
      if (!conv(a, m + ".down_pair." + std::to_string(i))) return false;
      if (!lastl) {
        const size_t img_bytes = (size_t)o.H * o.W * o.Cp * h->esz();
        Act ol = o, oi = o; oi.off += img_bytes;
        tap("mask.lbl_" + std::to_string(i) + ".raw", ol);
        tap("mask.img_" + std::to_string(i) + ".raw", oi);
      }
      cur = o; ncur = no;
    }
    return true;
  }

  template <typename F> static void for_each_pref(Op& op, F f) {
    PRef* all[] = {&op.x, &op.pro_scale, &op.pro_shift, &op.w, &op.bias, &op.y, &op.res, &op.y_nchw, &op.stat, &op.xm, &op.m_scale,
                   &op.m_shift, &op.ys0, &op.ys1, &op.slab, &op.x2, &op.w2, &op.m_part,
                   &op.s_slab, &op.s_bias, &op.s_y, &op.s_res, &op.s_stat, &op.m_slab, &op.m_bias, &op.m_xm, &op.m_sc, &op.m_sh,
                   &op.m_ys0, &op.m_ys1, &op.f_part, &op.f_gamma, &op.f_beta, &op.f_scale, &op.f_shift, &op.p_x, &op.p_y, &op.p_stat,
                   &op.a_t1, &op.a_sc1, &op.a_sh1, &op.a_ts, &op.a_scs, &op.a_shs, &op.a_x, &op.a_out, &op.k_s0, &op.k_s1, &op.k_s2, &op.k_dst,
                   &op.wi_x, &op.wi_sc, &op.wi_sh, &op.wi_v, &op.wi_slab, &op.wi_x2, &op.wi_xres, &op.wi_o, &op.wi_sc2, &op.wi_sh2, &op.wo_m, &op.wo_bias, &op.wo_y, &op.wo_res, &op.wo_stat,
                   &op.lc_s0, &op.lc_s1, &op.lc_s2, &op.lc_w, &op.lc_bias, &op.lc_y, &op.lc_stat, &op.st_part[0], &op.st_part[1], &op.g_a, &op.g_c};
    for (PRef* r : all) if (r->sp == PS_WS) f(*r);
  }
  AllocRec* alloc_of(size_t voff) {
    // allocations are in increasing voff order
    size_t lo = 0, hi = allocs.size();
    while (lo + 1 < hi) { const size_t mid = (lo + hi) / 2; if (allocs[mid].voff <= voff) lo = mid; else hi = mid; }
    return (!allocs.empty() && voff >= allocs[lo].voff && voff < allocs[lo].voff + allocs[lo].bytes) ? &allocs[lo] : nullptr;
  }
  // Lifetime analysis + placement; rewrites every workspace reference of the plan.  The plan runs its launches in
  // order on one stream, so a buffer is live from the first to the last launch that names it; buffers written outside
  // the plan (the label-only results rib_chain gathers into their slots before it runs the frame) are live from the
  // start, tapped activations (debug) until the end.
  bool assign_physical() {
    const bool reuse = pol.ws_reuse && !P->labels_only;
    if (!reuse) { P->ws_bytes = ws; return true; }
    bool bad = false;
    for (size_t i = 0; i < P->ops.size(); ++i)
      for_each_pref(P->ops[i], [&](PRef& r) {
        AllocRec* a = alloc_of(r.off);
        if (!a) { bad = true; return; }
        a->first = std::min(a->first, (int)i); a->last = std::max(a->last, (int)i);
      });
    if (bad) { error = "workspace reference outside every allocation"; return false; }
    const LabelSlots& l = P->ls;
    for (size_t off : {l.x0, l.nx_sc, l.nx_sh, l.cat, l.ncat_sc, l.ncat_sh})
      if (AllocRec* a = alloc_of(off)) a->first = -1;
    if (h->keep_taps)
      for (const Tap& t : P->taps)
        if (AllocRec* a = alloc_of(t.off)) a->last = INT_MAX;
    // placement: largest first, each at the lowest offset where it does not meet a placed buffer with an overlapping lifetime
    std::vector<int> order;
    for (size_t i = 0; i < allocs.size(); ++i) if (allocs[i].last >= 0 || allocs[i].first == -1) order.push_back((int)i);
    std::sort(order.begin(), order.end(), [&](int a, int b) { return allocs[a].bytes != allocs[b].bytes ? allocs[a].bytes > allocs[b].bytes : a < b; });
    std::vector<int> placed;
    size_t top = 0;
    for (int i : order) {
      AllocRec& a = allocs[i];
      std::vector<std::pair<size_t, size_t>> busy;     // physical ranges of the placed buffers alive together with a
      for (int j : placed) {
        const AllocRec& b = allocs[j];
        if (b.first <= a.last && a.first <= b.last) busy.push_back({b.phys, b.phys + b.bytes});
      }
      std::sort(busy.begin(), busy.end());
      size_t at = 0;
      for (auto& r : busy) { if (at + a.bytes <= r.first) break; at = std::max(at, r.second); }
      a.phys = at;
      top = std::max(top, at + a.bytes);
      placed.push_back(i);
    }
    auto remap = [&](size_t voff) { AllocRec* a = alloc_of(voff); return a ? a->phys + (voff - a->voff) : voff; };
    for (Op& op : P->ops) for_each_pref(op, [&](PRef& r) { r.off = remap(r.off); });
    LabelSlots& m = P->ls;
    m.x0 = remap(m.x0); m.nx_sc = remap(m.nx_sc); m.nx_sh = remap(m.nx_sh); m.cat = remap(m.cat); m.ncat_sc = remap(m.ncat_sc); m.ncat_sh = remap(m.ncat_sh);
    for (Tap& t : P->taps) t.off = remap(t.off);
    P->ws_virtual = ws;
    P->ws_bytes = top;
    return true;
  }

  void record_label_slots(const Act& x, const Norm& nx, const Act& CAT, const Norm& ncat) {
    LabelSlots& l = P->ls;
    l.x0 = x.off; l.x0_b = (size_t)B * x.H * x.W * x.Cp * h->esz();
    l.nx_sc = nx.sc; l.nx_sh = nx.sh; l.nx_b = (size_t)B * nx.ld * sizeof(float);
    l.cat = CAT.off; l.cat_b = (size_t)B * CAT.H * CAT.W * CAT.Cp * h->esz();
    l.ncat_sc = ncat.sc; l.ncat_sh = ncat.sh; l.ncat_b = (size_t)B * ncat.ld * sizeof(float);
  }

  // the label-only launches alone (for rib_chain's batched pre-pass): pack.label, the mask network's label branch,
  // down_first; same tensors, same launch arguments as in build()
  bool build_labels() {
    const Cfg& g = h->g;
    const rib_config& c = g.c;
    const int H = P->H, W = P->W;
    P->labels_only = true;
    defer_stats = false;
    const bool vL = lowc_serves({"down_first", "flow_network_temp.down_lbl.0"});
    Act L = vL ? virt(c.label_nc, H, W, U_LABEL, c.label_nc) : act(c.label_nc, H, W);
    if (!vL && L.Cp > 32) { error = "input channel counts above 32 are not supported by the pack kernel"; return false; }
    if (!vL) {
      Op op; op.kind = OP_PACK; op.kclass = RIB_KC_PACK; op.name = "pack.label";
      memset(&op.kp, 0, sizeof op.kp);
      op.kp.c0 = c.label_nc; op.kp.c1 = 0; op.kp.c2 = 0; op.kp.dC = L.Cp; op.kp.HW = H * W;
      op.k_s0 = US(U_LABEL); op.k_dst = WS(L.off);
      op.grid = dim3((H * W + 63) / 64, B, 1);
      push(op);
    }
    const int chm = g.mask_nf(c.mask_down);
    if (h->padc(chm) != chm) { error = "mask network width must be a multiple of 8 (16 with bf16 storage)"; return false; }
    Act CAT = act(2 * chm, H >> c.mask_down, W >> c.mask_down);
    Norm ncat = norm(CAT.Cp);
    if (!mask_branch(0, L, CAT, ncat, chm)) return false;
    const ConvDef& df = conv_of(h, "down_first");
    Act x = act(df.cout, H, W); Norm nx = norm(x.Cp);
    ConvArgs a; a.cd = &df; a.in = L; a.out = x; a.want_stats = true; a.stats_out = &nx;
    if (!conv(a, "down_first")) return false;
    record_label_slots(x, nx, CAT, ncat);
    return assign_physical();
  }

  // ---- the frame plan, sub-network by sub-network (Generator.forward, PGNR/models/generator.py:181-234) ------------------
  // Tensors that more than one sub-network touches.  The planners below run in the order of `build()`, which is the order
  // of the launches AND of the workspace allocations (assign_physical() shares bytes by lifetime afterwards).
  struct Frame {
    Act L;                    // label map [B,22,H,W]: virtual (the caller's tensor, read in place by k_conv_lowc) or packed NHWC
    Act Ein;                  // cat([img_fake, img_prev]) (generator.py:197): the embedder's input
    Act I9;                   // cat([img_prev, img_fake, img]) (generator.py:232): the mask network's image input
    std::vector<Act> cond;    // condition maps of the embedder, level 0 (full resolution) .. emb_down
    Act CAT; Norm ncat;       // the mask network's concatenated encoder outputs [label | image] and their InstanceNorm
    int chm = 0, Hm = 0, Wm = 0;
  };

  // boundary: NCHW user tensors -> NHWC (+concat, +zero channel padding), or nothing at all where k_conv_lowc reads them in place
  bool plan_boundary(Frame& F) {
    const Cfg& g = h->g;
    const rib_config& c = g.c;
    const int H = P->H, W = P->W;
    // Where the first layers can read the caller's NCHW tensors in place (k_conv_lowc) there is no packed copy at all
    const bool vL = lowc_serves({"down_first", "flow_network_temp.down_lbl.0"});
    const bool vE = lowc_serves({"ref_embedding.conv_first"});
    const bool vI = lowc_serves({"flow_network_temp.down_img.0"}) && head_without_nhwc(conv_of(h, "conv_img"));
    F.L = vL ? virt(c.label_nc, H, W, U_LABEL, c.label_nc) : act(c.label_nc, H, W);
    F.Ein = vE ? virt(c.image_nc * 2, H, W, U_FAKE, c.image_nc, U_PREV, c.image_nc) : act(c.image_nc * 2, H, W);     // cat([img_fake, img_prev]) generator.py:197
    F.I9 = vI ? virt(c.image_nc * 3, H, W, U_PREV, c.image_nc, U_FAKE, c.image_nc, U_IMG, c.image_nc) : act(c.image_nc * 3, H, W);   // cat([img_prev, img_fake, img]) generator.py:232
    auto pack = [&](const std::string& nm, const Act& dst, int s0, int c0, int s1, int c1) {
      Op op; op.kind = OP_PACK; op.kclass = RIB_KC_PACK; op.name = nm;
      memset(&op.kp, 0, sizeof op.kp);
      op.kp.c0 = c0; op.kp.c1 = c1; op.kp.c2 = 0; op.kp.dC = dst.Cp; op.kp.HW = H * W;
      op.k_s0 = US(s0); if (c1) op.k_s1 = US(s1);
      op.k_dst = WS(dst.off);
      op.grid = dim3((H * W + 63) / 64, B, 1);
      push(op);
    };
    if (F.L.Cp > 32 || F.Ein.Cp > 32 || F.I9.Cp > 32) { error = "input channel counts above 32 are not supported by the pack kernel"; return false; }   // (Cp = 0 for the unpacked ones)
    mark_label = true;
    if (!vL) pack("pack.label", F.L, U_LABEL, c.label_nc, 0, 0);
    mark_label = false;
    if (!vI) pack("pack.img9", F.I9, U_PREV, c.image_nc, U_FAKE, c.image_nc);           // cat([img_prev, img_fake, .]) generator.py:232
    if (!vE) pack("pack.embed_in", F.Ein, U_FAKE, c.image_nc, U_PREV, c.image_nc);      // cat([img_fake, img_prev]) generator.py:197

    return true;
  }

  // ref_embedding (LabelEmbedder 'encoder', generator.py:360-387) + gamma/beta of the SPADEs of the small condition levels
  bool plan_embedder(Frame& F) {
    const Cfg& g = h->g;
    const rib_config& c = g.c;
    const int H = P->H, W = P->W;
    std::vector<Act>& cond = F.cond;
    cond.assign(c.emb_down + 1, Act());
    {
      const ConvDef& cf = conv_of(h, "ref_embedding.conv_first");
      cond[0] = act(cf.cout, H, W);
      ConvArgs a; a.cd = &cf; a.in = F.Ein; a.out = cond[0]; a.act = ACT_LRELU;
      if (!conv(a, "ref_embedding.conv_first")) return false;
      tap("cond_0", cond[0]);
      for (int i = 0; i < c.emb_down; ++i) {
        const ConvDef& cd = conv_of(h, "ref_embedding.down_" + std::to_string(i));
        cond[i + 1] = act(cd.cout, cond[i].H / 2, cond[i].W / 2);
        ConvArgs b; b.cd = &cd; b.in = cond[i]; b.out = cond[i + 1]; b.act = ACT_LRELU;
        if (!conv(b, cd.name)) return false;
        tap("cond_" + std::to_string(i + 1), cond[i + 1]);
      }
    }
    // gamma/beta of all SPADEs of the small condition levels, one GEMM per level (cond_level_gemm)
    for (int j = 0; j <= c.emb_down; ++j)
      if (!cond_level_gemm(j, cond[j])) return false;

    return true;
  }

  // label encoder of the mask network: depends only on the label map, so a chain runs it once per segment (labels-only plan);
  // in a paired frame plan it rides with the image encoder instead (plan_mask_net)
  bool plan_mask_label_branch(Frame& F) {
    const Cfg& g = h->g;
    const rib_config& c = g.c;
    const int H = P->H, W = P->W;
    F.chm = g.mask_nf(c.mask_down);
    F.Hm = H >> c.mask_down; F.Wm = W >> c.mask_down;
    const int chm = F.chm;
    if (h->padc(chm) != chm) { error = "mask network width must be a multiple of 8 (16 with bf16 storage)"; return false; }
    F.CAT = act(2 * chm, F.Hm, F.Wm);
    F.ncat = norm(F.CAT.Cp);
    if (!pair_mask) {
      mark_label = true;
      if (!mask_branch(0, F.L, F.CAT, F.ncat, chm)) return false;
      mark_label = false;
    }

    return true;
  }

  // main generator (generator.py:201-228): down_first, down_0..D with AvgPool, res blocks, up_D..0, conv_img + tanh
  bool plan_trunk(Frame& F) {
    const Cfg& g = h->g;
    const rib_config& c = g.c;
    const int H = P->H, W = P->W;
    const int D = c.num_down_img;
    const std::vector<Act>& cond = F.cond;
    Act x; Norm nx;
    {
      const ConvDef& df = conv_of(h, "down_first");
      x = act(df.cout, H, W); nx = norm(x.Cp);
      ConvArgs a; a.cd = &df; a.in = F.L; a.out = x; a.want_stats = true; a.stats_out = &nx;
      mark_label = true;
      if (!conv(a, "down_first")) return false;
      mark_label = false;
      tap("down_first", x);
    }
    record_label_slots(x, nx, F.CAT, F.ncat);
    for (int i = 0; i <= D; ++i) {
      Act out; Norm nout;
      const bool last = (i == D);
      if (!spade_block("down_" + std::to_string(i), x, false, nx, cond[std::min(c.emb_down, i)], &out, last ? &nout : nullptr)) return false;
      if (!last) {   // self.downsample = AvgPool2d(3, 2, 1) (generator.py:127,207-208)
        if (out.Cp % 4 != 0 || 256 % (out.Cp / 4) != 0) { error = "avgpool: unsupported channel count"; return false; }
        Act pooled = act(out.C, out.H / 2, out.W / 2);
        Norm np = norm(pooled.Cp);
        const int slots = 256 / (out.Cp / 4), ppb = slots * 4;
        const int blocks = (pooled.H * pooled.W + ppb - 1) / ppb;
        const size_t part = alloc((size_t)B * blocks * 2 * out.Cp * sizeof(double));
        Op op; op.kind = OP_POOL; op.kclass = RIB_KC_POOL; op.name = "down_" + std::to_string(i) + ".pool";
        memset(&op.pp, 0, sizeof op.pp);
        op.pp.H = out.H; op.pp.W = out.W; op.pp.C = out.Cp; op.pp.blocks = blocks;
        op.p_x = WS(out.off); op.p_y = WS(pooled.off); op.p_stat = WS(part);
        op.grid = dim3(blocks, B, 1);
        push(op);
        Op f; f.kind = OP_FINALIZE; f.kclass = RIB_KC_STATS; f.name = op.name + ".stats";
        memset(&f.fp, 0, sizeof f.fp);
        f.fp.tiles = blocks; f.fp.Cs = out.Cp; f.fp.C = out.Cp; f.fp.ld = np.ld; f.fp.off = 0;
        f.fp.inv_count = 1.0f / ((float)pooled.H * (float)pooled.W); f.fp.eps = 1e-5f;
        f.f_part = WS(part); f.f_scale = WS(np.sc); f.f_shift = WS(np.sh);
        f.grid = dim3((out.Cp + 15) / 16, B, 1);
        finalize_or_defer(f, &np, false, part, blocks, out.Cp, f.fp.inv_count, false, 0, 0);
        x = pooled; nx = np;
      } else { x = out; nx = nout; }
    }
    const int jres = std::min(c.emb_down, D + 1);
    for (int i = 0; i < g.num_res_blocks(); ++i) {
      Act out; Norm nout;
      if (!spade_block("res_" + std::to_string(i), x, false, nx, cond[jres], &out, &nout)) return false;
      x = out; nx = nout;
    }
    bool ups = false;
    for (int i = D; i >= 0; --i) {
      Act out; Norm nout;
      // statistics of a nearest-x2 upsampled tensor equal those of the tensor itself, so the
      // producer's (scale, shift) are reused and the upsample is folded into the consumer's read
      if (!spade_block("up_" + std::to_string(i), x, ups, nx, cond[std::min(i, c.emb_down)], &out, i != 0 ? &nout : nullptr)) return false;
      x = out; nx = nout; ups = true;
    }
    {   // conv_img 'AC' + tanh (generator.py:114-116,228); also lands in img9[6:9]
      const ConvDef& ci = conv_of(h, "conv_img");
      ConvArgs a; a.cd = &ci; a.in = x; a.pro_lrelu = true; a.out = F.I9; a.yoff = c.image_nc * 2;
      a.cout_store = c.image_nc; a.act = ACT_TANH; a.y_nchw = US(U_IMG);
      a.y_none = F.I9.virt;      // the mask network then reads the image from the caller's NCHW tensor
      if (!conv(a, "conv_img")) return false;
    }

    return true;
  }

  // MaskGenerator (generator.py:493-510): image encoder (paired with the label encoder where the plan pairs), join with the
  // label branch, res_flow blocks, phase-convolution decoder, sigmoid head (+ the driver's blend)
  bool plan_mask_net(Frame& F) {
    const Cfg& g = h->g;
    const rib_config& c = g.c;
    const int H = P->H, W = P->W;
    const std::string m = "flow_network_temp";
    const int chm = F.chm, Hm = F.Hm, Wm = F.Wm;
    const Act& CAT = F.CAT; const Norm& ncat = F.ncat;
    if (pair_mask ? !mask_branches_paired(F.L, F.I9, CAT, ncat, chm) : !mask_branch(1, F.I9, CAT, ncat, chm)) return false;
    tap("mask.cat.raw", CAT);
    Act r; bool first = true;
    for (int i = 0; i < c.mask_res_blocks; ++i) {
      const std::string bn = m + ".res_flow." + std::to_string(i);
      const ConvDef& c0 = conv_of(h, bn + ".conv_block_0");
      const ConvDef& c1 = conv_of(h, bn + ".conv_block_1");
      const bool learned = h->conv_index.count(bn + ".conv_block_s") > 0;
      const Act xin = first ? CAT : r;                              // (lazy: the previous block's join, computed by this block's first input transform)
      const Act xin_stored = xin.lazy ? xin.lazy->o : xin;          // ... which also stores it: the residual of this block's join
      Act t0 = act(c0.cout, Hm, Wm); Norm n0 = norm(t0.Cp);
      { ConvArgs a; a.cd = &c0; a.in = xin; a.out = t0; if (first) { a.pro = &ncat; a.pro_lrelu = true; }
        a.want_stats = true; a.stats_out = &n0; a.affine = true;
        if (!conv(a, c0.name)) return false; }
      Act t1 = act(c1.cout, Hm, Wm); Norm n1 = norm(t1.Cp);
      { ConvArgs a; a.cd = &c1; a.in = t0; a.out = t1; a.pro = &n0; a.pro_lrelu = true;
        a.want_stats = true; a.stats_out = &n1; a.affine = true;
        if (!conv(a, c1.name)) return false; }
      Act ts; Norm ns;
      if (learned) {
        const ConvDef& cs = conv_of(h, bn + ".conv_block_s");
        ts = act(cs.cout, Hm, Wm); ns = norm(ts.Cp);
        ConvArgs a; a.cd = &cs; a.in = xin; a.out = ts; if (first) { a.pro = &ncat; a.pro_lrelu = true; }
        a.want_stats = true; a.stats_out = &ns; a.affine = true;
        if (!conv(a, cs.name)) return false;
      } else if (first) { error = "mask res block 0 must have a learned shortcut"; return false; }
      Act o = act(c1.cout, Hm, Wm);
      {
        // the join feeds the next block's conv_block_0 (no prologue): when that runs in the Winograd domain its input
        // transform computes the join on the fly and stores it
        const bool no_lazy = !pol.lazy_sources;
        ConvArgs t;
        if (i + 1 < c.mask_res_blocks) t.cd = &conv_of(h, m + ".res_flow." + std::to_string(i + 1) + ".conv_block_0");
        if (t.cd && !no_lazy && h->prec() == PREC_F32 && runs_wino(t, Hm, Wm) && t.cd->cinp == o.Cp && n1.ld == o.Cp) {
          auto L = std::make_shared<LazySrc>();
          L->mode = WSRC_JOIN; L->name = bn + ".join"; L->x = t1; L->nx = n1; L->o = o;
          if (learned) { L->has2 = true; L->x2 = ts; L->n2 = ns; } else L->xres = xin_stored;
          Act y = o; y.lazy = L;
          tap("mask.res_" + std::to_string(i), o);
          r = y; first = false;
          continue;
        }
      }
      Op op; op.kind = OP_INADD; op.kclass = RIB_KC_ELTWISE; op.name = bn + ".join";
      memset(&op.ap, 0, sizeof op.ap);
      op.ap.C = o.Cp; op.ap.HW = Hm * Wm; op.ap.ld = n1.ld;
      op.a_t1 = WS(t1.off);
      if (!take_partials(n1, op.ap.st1, op, 0)) { materialize(n1); op.a_sc1 = WS(n1.sc); op.a_sh1 = WS(n1.sh); }
      if (learned) {
        op.a_ts = WS(ts.off);
        if (!take_partials(ns, op.ap.sts, op, 1)) { materialize(ns); op.a_scs = WS(ns.sc); op.a_shs = WS(ns.sh); }
      } else op.a_x = WS(xin_stored.off);
      op.a_out = WS(o.off);
      const int nsl = (o.Cp + 63) / 64;
      op.ap.nslices = nsl; op.ap.pblocks = std::max(1, std::min((Hm * Wm + 15) / 16, ((op.ap.st1.tiles > 0 || op.ap.sts.tiles > 0) ? 384 : 2048) / nsl));
      op.grid = dim3(op.ap.pblocks * nsl, B, 1);
      push(op);
      tap("mask.res_" + std::to_string(i), o);
      r = o; first = false;
    }
    Act cur = r; Norm ncur; bool have = false;
    for (int j = 0; j < c.mask_down; ++j) {
      const ConvDef& cd = conv_of(h, m + ".up_flow." + std::to_string(2 * j + 1));
      Act o = act(cd.cout, cur.H * 2, cur.W * 2); Norm no = norm(o.Cp);
      ConvArgs a; a.cd = &cd; a.in = cur; a.ups = true; a.out = o;
      if (have) { a.pro = &ncur; a.pro_lrelu = true; }
      a.want_stats = true; a.stats_out = &no; a.affine = true;
      if (!conv(a, cd.name)) return false;
      tap("mask.up_" + std::to_string(j) + ".raw", o);
      cur = o; ncur = no; have = true;
    }
    {
      const ConvDef& cm = conv_of(h, m + ".conv_mask.0");
      ConvArgs a; a.cd = &cm; a.in = cur; if (have) { a.pro = &ncur; a.pro_lrelu = true; }
      a.act = ACT_SIGMOID; a.y_user = US(U_MASK);
      if (!conv(a, cm.name)) return false;
    }
    return true;
  }

  bool build() {
    Frame F;
    return plan_boundary(F) && plan_embedder(F) && plan_mask_label_branch(F) && plan_trunk(F) && plan_mask_net(F) && assign_physical();
  }
};

// PLAN_LABELS: the label-only launches alone (rib_chain's batched pre-pass).  PLAN_UNPAIRED: a frame plan that keeps the mask
// network's label encoder in launches of its own (rib_chain skips them); the default frame plan pairs the two encoders
// level by level at batch 1 (Builder::mask_branches_paired) - same kernels, same choices, bit-identical frames.
enum { PLAN_LABELS = 1, PLAN_UNPAIRED = 2 };
inline bool plan_pairs(int B, int flags) {
  return FusionPolicy().pair_mask_encoders && B == 1 && !(flags & (PLAN_LABELS | PLAN_UNPAIRED));
}
Plan* get_plan(rib_handle* h, int B, int H, int W, int flags = 0, int tuneB = 0) {
  const bool labels_only = (flags & PLAN_LABELS) != 0;
  if (!labels_only && !plan_pairs(B, 0)) flags &= ~PLAN_UNPAIRED;      // one frame plan where nothing is paired anyway
  if (h->plan_batch > 0) tuneB = h->plan_batch;                       // batch-invariant policy (rib_set_plan_batch)
  if (tuneB == B) tuneB = 0;
  const std::array<int, 5> key = {flags & 3, tuneB, B, H, W};
  auto it = h->plans.find(key);
  if (it != h->plans.end()) return it->second.get();
  const int mult = 1 << std::max(h->g.c.num_down_img, h->g.c.mask_down);
  if (B < 1 || H < mult || W < mult || H % mult || W % mult) {
    h->err = fmt("unsupported shape B=%d H=%d W=%d: H and W must be positive multiples of %d (SURVEY F5)", B, H, W, mult);
    return nullptr;
  }
  {  // the kernels address one sample with 32-bit element offsets: H*W*C < 2^31 for every activation
    const rib_config& c = h->g.c;
    const size_t widest = (size_t)std::max(std::max(c.emb_filters, c.mask_filters * 2), std::max(c.num_filters * 2, 32));
    if ((size_t)H * W * widest >= (1ull << 31)) {
      h->err = fmt("unsupported shape H=%d W=%d: a full-resolution activation exceeds 2^31 elements", H, W);
      return nullptr;
    }
  }
  std::unique_ptr<Plan> P(new Plan());
  P->B = B; P->H = H; P->W = W;
  Builder b; b.h = h; b.P = P.get(); b.B = B; b.tuneB = tuneB; b.pair_mask = plan_pairs(B, flags);
  if (!(labels_only ? b.build_labels() : b.build())) { h->err = "plan: " + b.error; return nullptr; }
  Plan* raw = P.get();
  h->plans[key] = std::move(P);
  return raw;
}

// ------------------------------------------------------------------------------------------
// launch
// ------------------------------------------------------------------------------------------
struct Resolver {
  char* ws; float* blob; const void* user[U_COUNT];
  const rib_handle* h = nullptr;     // for PS_WINO
  template <typename T> T* get(const PRef& r) const {
    switch (r.sp) {
      case PS_WS: return reinterpret_cast<T*>(ws + r.off);
      case PS_WEIGHT: return reinterpret_cast<T*>(blob + r.off);
      case PS_WINO: return reinterpret_cast<T*>(h->wino_sets[r.off].d);
      case PS_USER: return reinterpret_cast<T*>(const_cast<void*>(user[r.off]));
      default: return nullptr;
    }
  }
};

// a kernel templated on the storage type alone, launched for the handle's precision mode
#define RIB_LAUNCH_ST(prec, KERNEL, grid, block, lds, st, ...)                                                        \
  do {                                                                                                                \
    if ((prec) == PREC_BF16) RIB_KLAUNCH((KERNEL<ST_BF16>), grid, block, lds, st, __VA_ARGS__);                \
    else if ((prec) == PREC_F16) RIB_KLAUNCH((KERNEL<ST_F16>), grid, block, lds, st, __VA_ARGS__);             \
    else RIB_KLAUNCH((KERNEL<ST_F32>), grid, block, lds, st, __VA_ARGS__);                                     \
  } while (0)

template <int CO, int CIN> void launch_head_t(int bf16, dim3 grid, hipStream_t st, const IgemmParams& p) {
  if (bf16 == PREC_BF16) RIB_KLAUNCH((k_conv_head<CO, CIN, ST_BF16>), grid, dim3(256), 0, st, p);
  else if (bf16 == PREC_F16) RIB_KLAUNCH((k_conv_head<CO, CIN, ST_F16>), grid, dim3(256), 0, st, p);
  else RIB_KLAUNCH((k_conv_head<CO, CIN, ST_F32>), grid, dim3(256), 0, st, p);
}
template <int CO> void launch_small_t(int bf16, dim3 grid, size_t lds, hipStream_t st, const IgemmParams& p) {
  if (bf16 == PREC_BF16) RIB_KLAUNCH((k_conv_small<CO, ST_BF16>), grid, dim3(256), lds, st, p);
  else if (bf16 == PREC_F16) RIB_KLAUNCH((k_conv_small<CO, ST_F16>), grid, dim3(256), lds, st, p);
  else RIB_KLAUNCH((k_conv_small<CO, ST_F32>), grid, dim3(256), lds, st, p);
}
void launch_head(int co, int cin, int bf16, dim3 grid, hipStream_t st, const IgemmParams& p) {
  if (cin == 16) { if (co == 1) launch_head_t<1, 16>(bf16, grid, st, p); else if (co == 2) launch_head_t<2, 16>(bf16, grid, st, p); else launch_head_t<3, 16>(bf16, grid, st, p); }
  else { if (co == 1) launch_head_t<1, 32>(bf16, grid, st, p); else if (co == 2) launch_head_t<2, 32>(bf16, grid, st, p); else launch_head_t<3, 32>(bf16, grid, st, p); }
}

int run_plan(rib_handle* h, Plan* P, const Resolver& R, hipStream_t st, bool skip_label_ops = false) {
  const int bf16 = h->prec();       // storage type of the activations: PREC_F32 / PREC_BF16 / PREC_F16
  for (Op& op : P->ops) {
    if (skip_label_ops && op.label_only) continue;     // done for the whole chain by the labels-only plan
    g_prof_pair = ProfPair();
    if (h->profiling && h->prof_kernels) {      // the launch below carries its own (start, stop) pair
      ProfPair pp;
      HIP_TRY(h, hipEventCreate(&pp.start));
      HIP_TRY(h, hipEventCreate(&pp.stop));
      h->prof_kernel_events.push_back({op.kclass, pp.start, pp.stop});
      g_prof_pair = pp;
    } else if (h->profiling) {      // one event in front of every launch (rib.h: a launch is charged the time to the next event)
      hipEvent_t e0 = nullptr;
      HIP_TRY(h, hipEventCreate(&e0));
      HIP_TRY(h, hipEventRecord(e0, st));
      h->prof_events.push_back({op.kclass, e0});
    }
    switch (op.kind) {
      case OP_IGEMM: {
        IgemmParams p = op.ip;
        p.x = R.get<const float>(op.x); p.pro_scale = R.get<const float>(op.pro_scale); p.pro_shift = R.get<const float>(op.pro_shift);
        p.w = R.get<const float>(op.w); p.bias = R.get<const float>(op.bias);
        p.y = R.get<float>(op.y); p.res = R.get<const float>(op.res); p.y_nchw = R.get<float>(op.y_nchw);
        p.stat_part = R.get<double>(op.stat); p.slab = R.get<float>(op.slab);
        p.x2 = R.get<const float>(op.x2); p.w2 = R.get<const float>(op.w2);
        p.xm = R.get<const float>(op.xm); p.m_scale = R.get<const float>(op.m_scale); p.m_shift = R.get<const float>(op.m_shift);
        p.ys0 = R.get<float>(op.ys0); p.ys1 = R.get<float>(op.ys1);
        p.m_part = R.get<const double>(op.m_part);
        p.zeros = R.blob + h->zero_off;
        if (op.small_co > 0 && op.head) {
          if (op.fuse_blend && R.user[U_FUSE]) {
            p.bl_img = reinterpret_cast<const float*>(R.user[U_IMG]); p.bl_dain = reinterpret_cast<const float*>(R.user[U_FAKE]);
            p.bl_fuse = reinterpret_cast<float*>(const_cast<void*>(R.user[U_FUSE]));
          }
          launch_head(op.small_co, p.Cin, bf16, op.grid, st, p);
        } else if (op.small_co > 0) {
          const size_t lds = ((size_t)18 * 18 * (p.Cin + 4) + (size_t)op.small_co * 9 * p.Cin) * sizeof(float);
          switch (op.small_co) {
            case 1: launch_small_t<1>(bf16, op.grid, lds, st, p); break;
            case 2: launch_small_t<2>(bf16, op.grid, lds, st, p); break;
            case 3: launch_small_t<3>(bf16, op.grid, lds, st, p); break;
            default: launch_small_t<4>(bf16, op.grid, lds, st, p); break;
          }
        } else
        RIB_KLAUNCH(pick_igemm_fn(op.var, p), op.grid, dim3(256 * op.var->KW), 0, st, p);
      } break;
      case OP_GEMM: {
        GemmDmaParams p = op.gp;
        p.A = R.get<const float>(op.g_a); p.B = R.get<const float>(op.g_b); p.C = R.get<float>(op.g_c);
        RIB_KLAUNCH(h->products == RIB_PRODUCTS_BF16X3 && op.var->gfn_x3 ? op.var->gfn_x3 : op.var->gfn, op.grid, dim3(256), 0, st, p);
      } break;
      case OP_FINALIZE: {
        FinalizeParams p = op.fp;
        p.part = R.get<const double>(op.f_part); p.gamma = R.get<const float>(op.f_gamma); p.beta = R.get<const float>(op.f_beta);
        p.scale = R.get<float>(op.f_scale); p.shift = R.get<float>(op.f_shift);
        if (p.tiles > 512) RIB_KLAUNCH(k_stats_finalize<4>, dim3((p.Cs + 3) / 4, op.grid.y, op.grid.z), dim3(1024), 0, st, p);
        else RIB_KLAUNCH(k_stats_finalize<16>, op.grid, dim3(1024), 0, st, p);
      } break;
      case OP_MODULATE: {
        ModulateParams p = op.mp;
        p.slab = R.get<const float>(op.m_slab); p.bias = R.get<const float>(op.m_bias); p.xm = R.get<const float>(op.m_xm);
        p.m_scale = R.get<const float>(op.m_sc); p.m_shift = R.get<const float>(op.m_sh);
        p.st.part = R.get<const double>(op.st_part[0]); p.st.gamma = R.get<const float>(op.st_gamma[0]); p.st.beta = R.get<const float>(op.st_beta[0]);
        p.ys0 = R.get<float>(op.m_ys0); p.ys1 = R.get<float>(op.m_ys1);
        RIB_LAUNCH_ST(bf16, k_spade_modulate, op.grid, dim3(256), 0, st, p);
      } break;
      case OP_SPLITEPI: {
        SplitEpiParams p = op.sp;
        p.slab = R.get<const float>(op.s_slab); p.bias = R.get<const float>(op.s_bias); p.y = R.get<float>(op.s_y);
        p.res = R.get<const float>(op.s_res); p.stat_part = R.get<double>(op.s_stat);
        RIB_LAUNCH_ST(bf16, k_splitk_epilogue, op.grid, dim3(256), 0, st, p);
      } break;
      case OP_POOL: {
        PoolParams p = op.pp;
        p.x = R.get<const float>(op.p_x); p.y = R.get<float>(op.p_y); p.stat_part = R.get<double>(op.p_stat);
        RIB_LAUNCH_ST(bf16, k_avgpool, op.grid, dim3(256), 0, st, p);
      } break;
      case OP_INADD: {
        InAddParams p = op.ap;
        p.t1 = R.get<const float>(op.a_t1); p.sc1 = R.get<const float>(op.a_sc1); p.sh1 = R.get<const float>(op.a_sh1);
        p.ts = R.get<const float>(op.a_ts); p.scs = R.get<const float>(op.a_scs); p.shs = R.get<const float>(op.a_shs);
        p.xres = R.get<const float>(op.a_x); p.out = R.get<float>(op.a_out);
        p.st1.part = R.get<const double>(op.st_part[0]); p.st1.gamma = R.get<const float>(op.st_gamma[0]); p.st1.beta = R.get<const float>(op.st_beta[0]);
        p.sts.part = R.get<const double>(op.st_part[1]); p.sts.gamma = R.get<const float>(op.st_gamma[1]); p.sts.beta = R.get<const float>(op.st_beta[1]);
        RIB_LAUNCH_ST(bf16, k_in_add, op.grid, dim3(256), 0, st, p);
      } break;
      case OP_WINO_IN: {
        WinoInParams p = op.wi;
        p.x = R.get<const float>(op.wi_x); p.pro_scale = R.get<const float>(op.wi_sc); p.pro_shift = R.get<const float>(op.wi_sh);
        p.v = R.get<float>(op.wi_v);
        p.st.part = R.get<const double>(op.st_part[0]); p.st.gamma = R.get<const float>(op.st_gamma[0]); p.st.beta = R.get<const float>(op.st_beta[0]);
        p.slab = R.get<const float>(op.wi_slab); p.sbias = R.get<const float>(op.wi_sbias);
        p.x2 = R.get<const float>(op.wi_x2); p.xres = R.get<const float>(op.wi_xres); p.o = R.get<float>(op.wi_o);
        p.pro2_scale = R.get<const float>(op.wi_sc2); p.pro2_shift = R.get<const float>(op.wi_sh2);
        p.st2.part = R.get<const double>(op.st_part[1]); p.st2.gamma = R.get<const float>(op.st_gamma[1]); p.st2.beta = R.get<const float>(op.st_beta[1]);
        if (op.wino_m == 4) {
          if (op.wi_mode == WSRC_SPADE) RIB_KLAUNCH(k_wino4_in<WSRC_SPADE>, op.grid, dim3(256), 0, st, p);
          else if (op.wi_mode == WSRC_JOIN) RIB_KLAUNCH(k_wino4_in<WSRC_JOIN>, op.grid, dim3(256), 0, st, p);
          else RIB_KLAUNCH(k_wino4_in<WSRC_PLAIN>, op.grid, dim3(256), 0, st, p);
        } else {
          if (op.wi_mode == WSRC_SPADE) RIB_KLAUNCH(k_wino_in<WSRC_SPADE>, op.grid, dim3(256), 0, st, p);
          else if (op.wi_mode == WSRC_JOIN) RIB_KLAUNCH(k_wino_in<WSRC_JOIN>, op.grid, dim3(256), 0, st, p);
          else RIB_KLAUNCH(k_wino_in<WSRC_PLAIN>, op.grid, dim3(256), 0, st, p);
        }
      } break;
      case OP_WINO_OUT: {
        WinoOutParams p = op.wo;
        p.m = R.get<const float>(op.wo_m); p.bias = R.get<const float>(op.wo_bias); p.y = R.get<float>(op.wo_y);
        p.res = R.get<const float>(op.wo_res); p.stat_part = R.get<double>(op.wo_stat);
        if (op.wino_m == 4) RIB_KLAUNCH(k_wino4_out, op.grid, dim3(256), 0, st, p);
        else RIB_KLAUNCH(k_wino_out, op.grid, dim3(256), 0, st, p);
      } break;
      case OP_LOWC: {
        LowcParams p = op.lc;
        p.s0 = R.get<const float>(op.lc_s0); p.s1 = R.get<const float>(op.lc_s1); p.s2 = R.get<const float>(op.lc_s2);
        p.w = R.get<const float>(op.lc_w); p.bias = R.get<const float>(op.lc_bias); p.y = R.get<float>(op.lc_y);
        p.stat_part = R.get<double>(op.lc_stat);
        launch_lowc(op.lowc_ce, op.lowc_ncol, op.lowc_tw, op.grid, st, p);
      } break;
      case OP_PACK: {
        PackParams p = op.kp;
        p.s0 = R.get<const float>(op.k_s0); p.s1 = R.get<const float>(op.k_s1); p.s2 = R.get<const float>(op.k_s2);
        p.dst = R.get<float>(op.k_dst);
        RIB_LAUNCH_ST(bf16, k_pack, op.grid, dim3(256), 0, st, p);
      } break;
    }
  }
  g_prof_pair = ProfPair();
  if (h->profiling && !h->prof_kernels) {        // closes the last launch's interval
    hipEvent_t e1 = nullptr;
    HIP_TRY(h, hipEventCreate(&e1));
    HIP_TRY(h, hipEventRecord(e1, st));
    h->prof_events.push_back({-1, e1});
  }
  HIP_TRY(h, hipGetLastError());
  return RIB_OK;
}

int check_ready(rib_handle* h) {
  if (!h) return RIB_ERR_INVALID;
  if (!h->weights_ready) return fail(h, RIB_ERR_STATE, "weights not loaded: call rib_finalize_weights() or rib_import_weights() first");
  return RIB_OK;
}

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

int rib_create(const rib_config* cfg, int device, rib_handle** out) {
  if (!cfg || !out) { g_create_error = "rib_create: null argument"; return RIB_ERR_INVALID; }
  *out = nullptr;
  const rib_config& c = *cfg;
  auto bad = [&](const std::string& m) { g_create_error = "rib_create: " + m; return RIB_ERR_UNSUPPORTED; };
  if (c.label_nc < 1 || c.image_nc < 1 || c.num_filters < 1 || c.num_down_img < 1 || c.mask_down < 1 ||
      c.emb_filters < 1 || c.mask_filters < 1 || c.mask_res_blocks < 1 || c.num_layers < c.num_down_img)
    return bad("invalid hyper-parameters");
  if (c.emb_down != c.num_down_img) return bad("embed.num_downsamples must equal num_downsamples_img (HSM.yaml: 4/4)");
  std::unique_ptr<rib_handle> h(new rib_handle());
  h->g.c = c; h->device = device;
  h->graph_replay = getenv("RIB_GRAPH") != nullptr && atoi(getenv("RIB_GRAPH")) != 0;
  for (int i = 0; i <= c.emb_down; ++i) {
    if (h->g.cond_ch(i) != h->g.emb_ch(i)) return bad("embed/generator max_num_filters disagree on the cond map width");
    if (h->g.emb_ch(i) % 32 != 0) return bad(fmt("embed width %d at level %d is not a multiple of 32", h->g.emb_ch(i), i));
  }
  for (int i = 1; i <= c.mask_down; ++i)
    if (h->g.mask_nf(i - 1) % 32 != 0) return bad("mask.num_filters must be a multiple of 32 (stride-2 kernels use 32-channel chunks)");
  for (int i = 0; i <= c.num_down_img + 1; ++i) {
    const int ch = h->g.nf(i);
    if (ch % 4 != 0 || 256 % (ch / 4 > 0 ? ch / 4 : 1) != 0 || ch > 1024) return bad(fmt("generator width %d unsupported (power-of-two multiples of 4 up to 1024)", ch));
  }
  h->convs = build_inventory(h->g);
  for (size_t i = 0; i < h->convs.size(); ++i) h->conv_index[h->convs[i].name] = (int)i;
  register_tensors(h.get());
  assign_weight_layout(h.get());
  if (device >= 0) {   // device < 0: host-only handle (inventory, plans, weight fold; no launches)
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&h->d_blob), h->blob_floats * sizeof(float));
    h->d_blob_floats = h->blob_floats;
    if (e != hipSuccess) { g_create_error = fmt("rib_create: device %d: %s", device, hipGetErrorString(e)); return RIB_ERR_HIP; }
  }
  *out = h.release();
  return RIB_OK;
}

void rib_destroy(rib_handle* h) {
  if (!h) return;
  drop_chain_graphs(h);
  for (auto& rs : h->raster_stage) { if (rs.host) (void)hipHostFree(rs.host); if (rs.done) (void)hipEventDestroy(rs.done); }
  if (h->d_blob) (void)hipFree(h->d_blob);
  free_wino_sets(h);
  for (auto& pe : h->prof_events) (void)hipEventDestroy(pe.second);
  for (auto& ke : h->prof_kernel_events) { (void)hipEventDestroy(ke.start); (void)hipEventDestroy(ke.stop); }
  delete h;
}

const char* rib_last_error(const rib_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int rib_num_tensors(const rib_handle* h) { return h ? (int)h->tensors.size() : RIB_ERR_INVALID; }

int rib_tensor_info(const rib_handle* h, int idx, const char** name, int* ndim, int64_t dims[4], int* used) {
  if (!h || idx < 0 || idx >= (int)h->tensors.size()) return RIB_ERR_INVALID;
  const TensorDef& t = h->tensors[idx];
  if (name) *name = t.name.c_str();
  if (ndim) *ndim = (int)t.dims.size();
  if (dims) for (size_t i = 0; i < 4; ++i) dims[i] = i < t.dims.size() ? t.dims[i] : 1;
  if (used) *used = t.used ? 1 : 0;
  return RIB_OK;
}

int rib_set_tensor(rib_handle* h, const char* name, const float* data, int ndim, const int64_t* dims) {
  if (!h || !name || !data || !dims) return RIB_ERR_INVALID;
  auto it = h->tensor_index.find(name);
  if (it == h->tensor_index.end()) return fail(h, RIB_ERR_INVALID, fmt("unexpected key '%s' in state_dict (strict load, PGNR/utils/utils.py:119)", name));
  TensorDef& t = h->tensors[it->second];
  if (ndim != (int)t.dims.size()) return fail(h, RIB_ERR_INVALID, fmt("'%s': rank %d, expected %zu", name, ndim, t.dims.size()));
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    if (dims[i] != t.dims[i]) return fail(h, RIB_ERR_INVALID, fmt("size mismatch for '%s' (dim %d: %lld vs %lld)", name, i, (long long)dims[i], (long long)t.dims[i]));
    n *= (size_t)dims[i];
  }
  if (t.used) t.data.assign(data, data + n);   // tensors of never-called modules are accepted and dropped
  t.set = true;
  h->weights_ready = false;
  return RIB_OK;
}

int rib_finalize_weights(rib_handle* h) {
  if (!h) return RIB_ERR_INVALID;
  for (auto& t : h->tensors)
    if (!t.set) return fail(h, RIB_ERR_MISSING, fmt("missing key '%s' in state_dict (strict load)", t.name.c_str()));
  std::vector<float> blob(h->blob_floats, 0.f);
  { const BlobHeader bh = blob_header_of(h); memcpy(blob.data(), &bh, sizeof bh); }
  for (auto& c : h->convs) {
    if (!c.used) continue;
    const std::string p = c.name + ".layers.conv";
    const int K = c.cin * c.ks * c.ks;
    const std::vector<float>* w = nullptr;
    double sigma = 1.0;
    if (c.spectral) {
      w = tensor_data(h, p + ".weight_orig");
      const std::vector<float>& u = *tensor_data(h, p + ".weight_u");
      const std::vector<float>& v = *tensor_data(h, p + ".weight_v");
      // sigma = u . (W_mat v): eval-mode spectral norm, u and v exactly as stored
      double s = 0.0;
      for (int o = 0; o < c.cout; ++o) {
        double row = 0.0;
        for (int k = 0; k < K; ++k) row += (double)(*w)[(size_t)o * K + k] * (double)v[k];
        s += (double)u[o] * row;
      }
      sigma = s;
      if (sigma == 0.0 || !std::isfinite(sigma)) return fail(h, RIB_ERR_INVALID, fmt("'%s': spectral norm sigma is %g", c.name.c_str(), sigma));
    } else {
      w = tensor_data(h, p + ".weight");
    }
    const std::vector<float>& b = *tensor_data(h, p + ".bias");
    const float inv = (float)sigma;
    const int taps = c.ks * c.ks;
    // OIHW -> [CoutPad][tap][CinPad]; torch computes weight_orig / sigma in fp32
    for (int o = 0; o < c.cout; ++o)
      for (int i = 0; i < c.cin; ++i)
        for (int t = 0; t < taps; ++t) {
          const float wv = (*w)[((size_t)o * c.cin + i) * taps + t];
          blob[c.w_off + ((size_t)o * taps + t) * c.cinp + i] = c.spectral ? wv / inv : wv;
        }
    for (int o = 0; o < c.cout; ++o) blob[c.b_off + o] = b[o];
    if (c.ups_in) {
      // phase filters of the upsample convolution (k_igemm, UPS): along each axis, phase p and tap a
      // sum the original taps   p = 0: a = 0 <- {0}, a = 1 <- {1, 2};   p = 1: a = 0 <- {0, 1}, a = 1 <- {2}
      // (fp32 sums of the folded filters, in a fixed dy-major order)
      static const int lo[2][2] = {{0, 1}, {0, 2}}, hi[2][2] = {{0, 2}, {1, 2}};   // [p][a]: inclusive tap range
      for (int o = 0; o < c.cout; ++o)
        for (int ph = 0; ph < 4; ++ph)
          for (int a = 0; a < 2; ++a)
            for (int bb = 0; bb < 2; ++bb) {
              const int py = ph >> 1, px = ph & 1;
              float* dst = &blob[c.wp_off + ((size_t)o * 16 + ph * 4 + a * 2 + bb) * c.cinp];
              for (int i = 0; i < c.cin; ++i) {
                float acc = 0.f;
                for (int dy = lo[py][a]; dy <= hi[py][a]; ++dy)
                  for (int dx = lo[px][bb]; dx <= hi[px][bb]; ++dx)
                    acc += blob[c.w_off + ((size_t)o * 9 + dy * 3 + dx) * c.cinp + i];
                dst[i] = acc;
              }
            }
    }
    if (c.in_affine) {
      const std::vector<float>& gm = *tensor_data(h, c.name + ".layers.norm.weight");
      const std::vector<float>& bt = *tensor_data(h, c.name + ".layers.norm.bias");
      for (int o = 0; o < c.cout; ++o) { blob[c.g_off + o] = gm[o]; blob[c.be_off + o] = bt[o]; }
    }
  }
  for (auto& c : h->convs) {
    if (!c.used || !c.fb_off) continue;
    const ConvDef& cs = conv_of(h, c.name.substr(0, c.name.size() - 1) + "s");
    for (int o = 0; o < c.cout; ++o) blob[c.fb_off + o] = blob[c.b_off + o] + blob[cs.b_off + o];
  }
  // SPADE gamma/beta filters: virtual column pairs [gamma(32) | beta(32)] per 32 virtual channels;
  // virtual channel v = set*Cp + c, set 0 = conv_block_0/1, set 1 = conv_block_s
  for (auto& sg : h->spades) {
    const std::string blk = sg.key.substr(0, sg.key.size() - 2);
    const std::string which = sg.key.substr(sg.key.size() - 1);
    const int condp = h->padc(sg.cond);
    for (int set = 0; set < sg.nsets; ++set) {
      const std::string cn = blk + ".conv_block_" + (set == 0 ? which : std::string("s"));
      const std::string sp = cn + ".layers.norm.mlps.0.0.layers.conv";
      const std::vector<float>& w = *tensor_data(h, sp + ".weight");   // [2C][cond]
      const std::vector<float>& b = *tensor_data(h, sp + ".bias");
      for (int ch = 0; ch < sg.C; ++ch) {
        const int v = set * sg.Cp + ch;
        const int colg = (v / 32) * 64 + (v % 32), colb = colg + 32;
        for (int k = 0; k < sg.cond; ++k) {
          blob[sg.w_off + (size_t)colg * condp + k] = w[(size_t)ch * sg.cond + k];              // gamma = first C rows
          blob[sg.w_off + (size_t)colb * condp + k] = w[(size_t)(sg.C + ch) * sg.cond + k];     // beta = last C rows
        }
        blob[sg.b_off + colg] = b[ch];
        blob[sg.b_off + colb] = b[sg.C + ch];
        if (sg.w1_off) {   // [gamma(16) | beta(16)]: v < 16
          for (int k = 0; k < sg.cond; ++k) {
            blob[sg.w1_off + (size_t)v * condp + k] = w[(size_t)ch * sg.cond + k];
            blob[sg.w1_off + (size_t)(16 + v) * condp + k] = w[(size_t)(sg.C + ch) * sg.cond + k];
          }
          blob[sg.b1_off + v] = b[ch];
          blob[sg.b1_off + 16 + v] = b[sg.C + ch];
        }
      }
    }
  }
  for (auto& c : h->convs) {
    if (!c.used || !c.wl_off) continue;
    // k_conv_lowc: [step][k within the MFMA's group][column], k = tap * CE + channel over the REAL channels (zero for the
    // channels that round Cin up to the group and for columns beyond Cout)
    const int CE = c.lowc_ce, N = c.lowc_ncol, KG = N == 16 ? 4 : 2;
    for (int k = 0; k < 9 * CE; ++k) {
      const int tap = k / CE, ch = k % CE;
      for (int col = 0; col < N; ++col)
        blob[c.wl_off + (size_t)k * N + col] = (col < c.cout && ch < c.cin) ? blob[c.w_off + ((size_t)col * 9 + tap) * c.cinp + ch] : 0.f;
    }
    (void)KG;   // (step s, slot j) = (k / KG, k % KG): the rows are already in k order
  }
  if (h->mc16()) {
    // rows of `rowlen` K-contiguous elements: [row][rowlen] bf16
    // IEEE half ends at 65504: a folded filter beyond it (a trained checkpoint's W / sigma can be far larger than the seed-defined
    // ones the mode was developed on) would become an infinity in the 16-bit copy and a NaN frame later - refuse it HERE, by name
    std::string out_of_range;
    auto to16 = [&](const std::string& name, size_t src, size_t dst, size_t rows, size_t rowlen) {
      uint16_t* d = reinterpret_cast<uint16_t*>(&blob[dst]);
      if (h->prec() == PREC_F16) {
        for (size_t i = 0; i < rows * rowlen; ++i) {
          const float v = blob[src + i];
          if (!(std::fabs(v) <= 65504.f) && out_of_range.empty()) out_of_range = fmt("%s: folded filter value %g", name.c_str(), (double)v);
          d[i] = host_f16(v);
        }
      } else for (size_t i = 0; i < rows * rowlen; ++i) d[i] = host_bf16(blob[src + i]);
    };
    for (auto& c : h->convs) {
      if (!c.used) continue;
      to16(c.name, c.w_off, c.w16_off, (size_t)c.coutp * c.ks * c.ks, c.cinp);      // a "row" is one (output channel, tap) slice
      if (c.ups_in) to16(c.name, c.wp_off, c.wp16_off, (size_t)c.coutp * 16, c.cinp);
    }
    for (auto& sg : h->spades) to16(sg.key, sg.w_off, sg.w16_off, sg.npad, h->padc(sg.cond));
    if (!out_of_range.empty())
      return fail(h, RIB_ERR_INVALID, "rib_finalize_weights: " + out_of_range + " is outside IEEE half's range (65504): this checkpoint cannot run in "
                                      "RIB_DTYPE_F16; use RIB_DTYPE_BF16 (same speed, fp32's range) or RIB_DTYPE_F32");
  }
  if (h->device < 0) {
    h->host_blob.swap(blob);
  } else {
    HIP_TRY(h, hipSetDevice(h->device));
    if (h->d_blob_floats != h->blob_floats) {
      drop_chain_graphs(h);
      if (h->d_blob) (void)hipFree(h->d_blob);
      h->d_blob = nullptr; h->d_blob_floats = 0;
      HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_blob), h->blob_floats * sizeof(float)));
      h->d_blob_floats = h->blob_floats;
    }
    HIP_TRY(h, hipMemcpy(h->d_blob, blob.data(), blob.size() * sizeof(float), hipMemcpyHostToDevice));
    h->weights_ready = true;
    const int rcw = refresh_wino_sets(h, nullptr);      // sets that plans already use follow the new weights
    if (rcw) return rcw;
    HIP_TRY(h, hipDeviceSynchronize());
  }
  // the host copies stay (138 MB at HSM.yaml's size): a later rib_set_tensor of a SUBSET of the tensors followed by
  // rib_finalize_weights (load_state_dict(strict=False), a fine-tuned head) folds the new values with the old ones
  return RIB_OK;
}

size_t rib_weights_bytes(const rib_handle* h) { return h ? h->blob_floats * sizeof(float) : 0; }

int rib_export_weights(rib_handle* h, void* dst, size_t bytes, void* hip_stream) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!dst || bytes != h->blob_floats * sizeof(float)) return fail(h, RIB_ERR_INVALID, "rib_export_weights: size mismatch");
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipMemcpyAsync(dst, h->d_blob, bytes, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(hip_stream)));
  return RIB_OK;
}

int rib_import_weights(rib_handle* h, const void* src, size_t bytes, void* hip_stream) {
  if (!h) return RIB_ERR_INVALID;
  if (!src || bytes != h->blob_floats * sizeof(float)) return fail(h, RIB_ERR_INVALID, "rib_import_weights: size mismatch");
  if (h->device < 0) return fail(h, RIB_ERR_INVALID, "rib_import_weights: host-only handle");
  HIP_TRY(h, hipSetDevice(h->device));
  drop_chain_graphs(h);      // (the blob may move)
  {   // the byte count alone does not identify a layout: check the header (a 32-byte read behind whatever filled src on this stream)
    BlobHeader got;
    HIP_TRY(h, hipMemcpyAsync(&got, src, sizeof got, hipMemcpyDeviceToHost, reinterpret_cast<hipStream_t>(hip_stream)));
    HIP_TRY(h, hipStreamSynchronize(reinterpret_cast<hipStream_t>(hip_stream)));
    const BlobHeader want = blob_header_of(h);
    if (got.magic != want.magic || got.version != want.version)
      return fail(h, RIB_ERR_INVALID, "rib_import_weights: not a weight blob of this library version (bad header)");
    if (got.prec != want.prec || got.floats != want.floats || got.layout_hash != want.layout_hash)
      return fail(h, RIB_ERR_INVALID, fmt("rib_import_weights: blob was exported by a handle with another precision mode or filter layout "
                                          "(mode %u vs %u, layout hash %016llx vs %016llx)", got.prec, want.prec,
                                          (unsigned long long)got.layout_hash, (unsigned long long)want.layout_hash));
  }
  if (h->d_blob_floats != h->blob_floats) {
    if (h->d_blob) (void)hipFree(h->d_blob);
    h->d_blob = nullptr; h->d_blob_floats = 0;
    HIP_TRY(h, hipMalloc(reinterpret_cast<void**>(&h->d_blob), h->blob_floats * sizeof(float)));
    h->d_blob_floats = h->blob_floats;
  }
  HIP_TRY(h, hipMemcpyAsync(h->d_blob, src, bytes, hipMemcpyDeviceToDevice, reinterpret_cast<hipStream_t>(hip_stream)));
  h->weights_ready = true;
  return refresh_wino_sets(h, reinterpret_cast<hipStream_t>(hip_stream));      // stream-ordered behind the copy
}

int rib_set_compute_dtype(rib_handle* h, int dtype) {
  if (!h || (dtype != RIB_DTYPE_F32 && dtype != RIB_DTYPE_BF16 && dtype != RIB_DTYPE_F16)) return RIB_ERR_INVALID;
  const int mode = dtype == RIB_DTYPE_BF16 ? PREC_BF16 : (dtype == RIB_DTYPE_F16 ? PREC_F16 : PREC_F32);
  if (h->prec_mode == mode) return RIB_OK;
  // The storage type decides the activation / filter layout (bf16: 16-channel minimum, bf16 filter copies in the
  // blob): plans and the weight layout are rebuilt, and the folded blob has to be produced again - by
  // rib_finalize_weights from the state-dict tensors the handle still holds, or by rib_import_weights from a blob
  // exported by a handle of the same storage type.
  h->plans.clear();
  drop_chain_graphs(h);
  free_wino_sets(h);       // (plans reference them by index; the bf16 mode has none)
  h->choices.clear();      // tuned variant indices belong to the previous precision's kernels: back to the cost model until re-pinned
  h->prec_mode = mode;
  assign_weight_layout(h);
  const bool had = h->weights_ready || !h->host_blob.empty();
  h->weights_ready = false;
  h->host_blob.clear();
  if (had) {
    bool all = true;
    for (auto& t : h->tensors) all = all && t.set && (!t.used || !t.data.empty());
    if (all) return rib_finalize_weights(h);
  }
  return RIB_OK;
}

int rib_set_products(rib_handle* h, int products) {
  if (!h || (products != RIB_PRODUCTS_F32 && products != RIB_PRODUCTS_BF16X3)) return RIB_ERR_INVALID;
  if (h->products == products) return RIB_OK;
  drop_chain_graphs(h);      // (captured segments hold the kernels of the other setting)
  h->products = products;
  return RIB_OK;
}

size_t rib_workspace_bytes(rib_handle* h, int B, int H, int W) {
  if (!h) return 0;
  Plan* P = get_plan(h, B, H, W);
  if (!P) return 0;
  size_t need = P->ws_bytes;
  if (plan_pairs(B, 0)) {      // rib_chain runs the unpaired twin of the frame plan on the same workspace
    Plan* PU = get_plan(h, B, H, W, PLAN_UNPAIRED);
    if (!PU) return 0;
    need = std::max(need, PU->ws_bytes);
  }
  // + scratch img/mask frames rib_chain uses when the caller does not ask for them
  return need + 2 * align256((size_t)B * h->g.c.image_nc * H * W * sizeof(float)) +
         align256((size_t)B * H * W * sizeof(float));
}

int rib_forward(rib_handle* h, int B, int H, int W, const float* label, const float* img_fake,
                const float* img_prev, float* img, float* mask, void* workspace, size_t workspace_bytes,
                void* hip_stream) {
  return rib_forward_blend(h, B, H, W, label, img_fake, img_prev, img, mask, nullptr, workspace, workspace_bytes, hip_stream);
}

int rib_forward_blend(rib_handle* h, int B, int H, int W, const float* label, const float* img_fake,
                      const float* img_prev, float* img, float* mask, float* fuse, void* workspace, size_t workspace_bytes,
                      void* hip_stream) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!label || !img_fake || !img_prev || !img || !mask || !workspace) return fail(h, RIB_ERR_INVALID, "rib_forward: null pointer");
  Plan* P = get_plan(h, B, H, W);
  if (!P) return RIB_ERR_INVALID;
  if (workspace_bytes < P->ws_bytes) return fail(h, RIB_ERR_WORKSPACE, fmt("workspace %zu < required %zu bytes", workspace_bytes, P->ws_bytes));
  Resolver R; R.ws = reinterpret_cast<char*>(workspace); R.blob = h->d_blob; R.h = h;
  R.user[U_LABEL] = label; R.user[U_FAKE] = img_fake; R.user[U_PREV] = img_prev; R.user[U_IMG] = img; R.user[U_MASK] = mask;
  R.user[U_FUSE] = nullptr;
  bool fused = false;       // does the plan's mask head carry the blend?
  if (fuse)
    for (const Op& op : P->ops) fused = fused || op.fuse_blend;
  if (fused) R.user[U_FUSE] = fuse;
  rc = run_plan(h, P, R, reinterpret_cast<hipStream_t>(hip_stream));
  if (rc == RIB_OK && fuse && !fused) rc = rib_blend(h, B, h->g.c.image_nc, H, W, img, mask, img_fake, fuse, hip_stream);
  return rc;
}

int rib_blend(rib_handle* h, int B, int C, int H, int W, const float* img, const float* mask,
              const float* dain, float* fuse, void* hip_stream) {
  if (!h || !img || !mask || !dain || !fuse) return RIB_ERR_INVALID;
  if (h->device >= 0) HIP_TRY(h, hipSetDevice(h->device));
  const size_t total = (size_t)B * C * H * W;
  const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 4096);
  RIB_KLAUNCH(k_blend, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream), img, mask, dain, fuse, C, H * W, total);
  HIP_TRY(h, hipGetLastError());
  return RIB_OK;
}

int rib_quantise(rib_handle* h, int B, int C, int H, int W, const float* img, uint8_t* out, void* hip_stream) {
  if (!h || !img || !out) return RIB_ERR_INVALID;
  if (h->device >= 0) HIP_TRY(h, hipSetDevice(h->device));
  const size_t total = (size_t)B * C * H * W;
  const unsigned blocks = (unsigned)std::min<size_t>((total + 255) / 256, 4096);
  RIB_KLAUNCH(k_quantise, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(hip_stream), img, out, C, H * W, total);
  HIP_TRY(h, hipGetLastError());
  return RIB_OK;
}

int rib_warp(rib_handle* h, int B, int C, int H, int W, const float* img, const float* flow, float* out, void* hip_stream) {
  if (!h || !img || !flow || !out) return RIB_ERR_INVALID;
  if (h->device >= 0) HIP_TRY(h, hipSetDevice(h->device));
  if (B < 1 || C < 1 || C > 8 || H < 1 || W < 1) return fail(h, RIB_ERR_INVALID, "rib_warp: 1 <= C <= 8 channels (the staged window must fit in LDS)");
  const int tilesX = (W + WARP_TW - 1) / WARP_TW, tilesY = (H + WARP_TH - 1) / WARP_TH;
  const size_t lds = (size_t)C * WARP_WH * WARP_PITCH * sizeof(float);      // 16 KB per channel
  // k_warp's dynamic-LDS limit is raised once per HANDLE, with the handle's device current (the attribute may be kept per
  // device: a process that drives two GPUs must not leave the second one at the 64 KB default), and a failure is reported
  // by the call that met it, not cached for the life of the process
  if (!h->warp_lds_ready) {
    HIP_TRY(h, hipFuncSetAttribute(reinterpret_cast<const void*>(&k_warp), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * WARP_WH * WARP_PITCH * (int)sizeof(float)));
    h->warp_lds_ready = true;
  }
  RIB_KLAUNCH(k_warp, dim3(tilesX * tilesY, B), dim3(256), lds, reinterpret_cast<hipStream_t>(hip_stream), img, flow, out, C, H, W, tilesX, xcd_chunk_of(tilesX * tilesY));
  HIP_TRY(h, hipGetLastError());
  return RIB_OK;
}

static_assert(sizeof(rib_stroke) == sizeof(RasterStroke) && sizeof(rib_stroke) == 48, "rib_stroke layout");

namespace {
struct RasterLayout { size_t strokes, colors, peaks, weights, canvas, total; };
static RasterLayout raster_layout(int T, int H, int W, int n_edges, int n_maps, int radius) {
  RasterLayout L; size_t o = 0;
  L.strokes = o; o += align256((size_t)T * n_edges * sizeof(rib_stroke));
  L.colors = o;  o += align256((size_t)n_edges * sizeof(uint32_t));
  L.peaks = o;   o += align256((size_t)T * n_maps * 2 * sizeof(int32_t));
  L.weights = o; o += align256((size_t)(radius + 1) * sizeof(double));
  L.canvas = o;  o += align256((size_t)T * H * W * sizeof(uint32_t));
  L.total = o;
  return L;
}
}  // namespace

size_t rib_rasterise_workspace_bytes(rib_handle* h, int T, int H, int W, int n_edges, int n_maps, int radius) {
  if (!h || T < 1 || H < 1 || W < 1 || n_edges < 0 || n_maps < 0 || radius < 0) return 0;
  return raster_layout(T, H, W, n_edges, n_maps, radius).total;
}

int rib_rasterise(rib_handle* h, int T, int H, int W, const rib_stroke* strokes, int n_edges,
                  const uint8_t* colors_rgb, int stroke_halfwidth, const int32_t* peaks, int n_maps,
                  const double* weights, int radius, float* labels, void* workspace, size_t workspace_bytes,
                  void* hip_stream) {
  if (!h) return RIB_ERR_INVALID;
  if (h->device < 0) return fail(h, RIB_ERR_INVALID, "rib_rasterise: host-only handle");
  if (T < 1 || H < 1 || W < 1 || !labels || !workspace || (n_edges > 0 && (!strokes || !colors_rgb)) ||
      (n_maps > 0 && (!peaks || !weights)))
    return fail(h, RIB_ERR_INVALID, "rib_rasterise: bad argument");
  if (3 + n_maps != h->g.c.label_nc)
    return fail(h, RIB_ERR_INVALID, fmt("rib_rasterise: 3 + %d maps != label_nc %d", n_maps, h->g.c.label_nc));
  if (H > RASTER_MAXPTS || W > RASTER_MAXPTS) return fail(h, RIB_ERR_INVALID, fmt("rib_rasterise: H, W <= %d", RASTER_MAXPTS));
  if (radius > 127 || stroke_halfwidth < 1 || stroke_halfwidth > 16) return fail(h, RIB_ERR_INVALID, "rib_rasterise: radius <= 127, 1 <= stroke half-width <= 16");
  for (size_t i = 0; i < (size_t)T * n_edges; ++i)
    if (strokes[i].n < 0 || strokes[i].n > RASTER_MAXPTS) return fail(h, RIB_ERR_INVALID, fmt("rib_rasterise: stroke %zu has %d samples", i, strokes[i].n));
  for (size_t i = 0; i < (size_t)T * n_maps; ++i) {
    const int32_t x = peaks[2 * i], y = peaks[2 * i + 1];
    if (x >= W || (x >= 0 && (y < 0 || y >= H))) return fail(h, RIB_ERR_INVALID, fmt("rib_rasterise: peak %zu (%d, %d) outside the frame", i, x, y));
  }
  const RasterLayout L = raster_layout(T, H, W, n_edges, n_maps, radius);
  if (workspace_bytes < L.total) return fail(h, RIB_ERR_WORKSPACE, fmt("workspace %zu < required %zu bytes", workspace_bytes, L.total));
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  char* ws = reinterpret_cast<char*>(workspace);
  HIP_TRY(h, hipSetDevice(h->device));
  // The four tables sit at the head of the workspace in one contiguous range [0, L.canvas): they are assembled in a page-locked
  // staging slot and go over in ONE asynchronous copy.  (Rounds 1-3 copied from the caller's pageable arrays and then
  // synchronised the stream so that the arrays could be reused: where the upload stream shares a hardware queue with the
  // stream a chain runs on, that wait was the previous segment's whole chain - 60 ms per call, profiles/r04_prof_driver.txt.)
  {
    rib_handle::RasterStage& rs = h->raster_stage[h->raster_next];
    h->raster_next ^= 1;
    if (rs.done) HIP_TRY(h, hipEventSynchronize(rs.done));      // the copy out of this slot two calls ago
    else HIP_TRY(h, hipEventCreateWithFlags(&rs.done, hipEventDisableTiming));
    if (rs.bytes < L.canvas) {
      if (rs.host) (void)hipHostFree(rs.host);
      rs.host = nullptr; rs.bytes = 0;
      HIP_TRY(h, hipHostMalloc(reinterpret_cast<void**>(&rs.host), L.canvas, hipHostMallocDefault));
      rs.bytes = L.canvas;
    }
    memset(rs.host, 0, L.canvas);
    if (n_edges > 0) {
      memcpy(rs.host + L.strokes, strokes, (size_t)T * n_edges * sizeof(rib_stroke));
      uint32_t* packed = reinterpret_cast<uint32_t*>(rs.host + L.colors);
      for (int e = 0; e < n_edges; ++e)
        packed[e] = (uint32_t)colors_rgb[3 * e] | ((uint32_t)colors_rgb[3 * e + 1] << 8) | ((uint32_t)colors_rgb[3 * e + 2] << 16);
    }
    if (n_maps > 0) {
      memcpy(rs.host + L.peaks, peaks, (size_t)T * n_maps * 2 * sizeof(int32_t));
      memcpy(rs.host + L.weights, weights, (size_t)(radius + 1) * sizeof(double));
    }
    HIP_TRY(h, hipMemcpyAsync(ws, rs.host, L.canvas, hipMemcpyHostToDevice, st));
    HIP_TRY(h, hipEventRecord(rs.done, st));
  }
  if (n_maps > 0) {
    HeatParams hp;
    hp.peaks = reinterpret_cast<const int32_t*>(ws + L.peaks); hp.w = reinterpret_cast<const double*>(ws + L.weights);
    hp.r = radius; hp.label = labels; hp.T = T; hp.H = H; hp.W = W; hp.nmaps = n_maps; hp.label_nc = 3 + n_maps; hp.ch0 = 3;
    RIB_KLAUNCH(k_heatmaps, dim3((H * W + 255) / 256, n_maps, T), dim3(256), 0, st, hp);
  }
  SkelParams sp;
  sp.strokes = reinterpret_cast<const RasterStroke*>(ws + L.strokes); sp.colors = reinterpret_cast<const uint32_t*>(ws + L.colors);
  sp.nedges = n_edges; sp.canvas = reinterpret_cast<uint32_t*>(ws + L.canvas); sp.label = labels;
  sp.T = T; sp.H = H; sp.W = W; sp.label_nc = 3 + n_maps; sp.bw = stroke_halfwidth;
  RIB_KLAUNCH(k_skeleton, dim3(T), dim3(256), 0, st, sp);
  HIP_TRY(h, hipGetLastError());
  return RIB_OK;
}

namespace {
// label-only work is batched over the chain when it has more than one frame, on one stream (RIB_NO_LABEL_BATCH=1 disables)
static bool chain_batches_labels(const rib_handle* h, int T, int B) {
  // T * B * split-K (<= 16) indexes blockIdx.z (< 65536) of the batched launches
  return T > 1 && B < 128 && (long)T * B < 4096 && !getenv("RIB_NO_LABEL_BATCH");
}
}  // namespace

size_t rib_chain_workspace_bytes(rib_handle* h, int T, int B, int H, int W) {
  if (!h || T < 1) return 0;
  const size_t base = rib_workspace_bytes(h, B, H, W);
  if (base == 0 || !chain_batches_labels(h, T, B)) return base;
  Plan* PL = get_plan(h, T * B, H, W, PLAN_LABELS, B);
  return PL ? base + align256(PL->ws_bytes) : 0;
}

// (inside extern "C" an unnamed namespace does not keep a function's name out of the dynamic symbol table: static does)
static int chain_enqueue(rib_handle* h, int T, int B, int H, int W, const float* key_frame, const float* labels,
                         const float* dains, float* imgs, float* masks, float* fuses, void* workspace,
                         size_t workspace_bytes, void* hip_stream);

int rib_set_graph_replay(rib_handle* h, int enable) {
  if (!h) return RIB_ERR_INVALID;
  h->graph_replay = enable != 0;
  if (!h->graph_replay) drop_chain_graphs(h);
  return RIB_OK;
}

int rib_set_plan_batch(rib_handle* h, int n) {
  if (!h || n < 0) return RIB_ERR_INVALID;
  if (h->plan_batch != n) drop_chain_graphs(h);     // (plans are keyed by the batch they follow: nothing else to drop)
  h->plan_batch = n;
  return RIB_OK;
}

int rib_get_plan_batch(const rib_handle* h) { return h ? h->plan_batch : RIB_ERR_INVALID; }

int rib_graph_stats(rib_handle* h, int64_t* captures, int64_t* replays) {
  if (!h) return RIB_ERR_INVALID;
  if (captures) *captures = (int64_t)h->graph_captures;
  if (replays) *replays = (int64_t)h->graph_replays;
  return RIB_OK;
}

int rib_chain(rib_handle* h, int T, int B, int H, int W, const float* key_frame, const float* labels,
              const float* dains, float* imgs, float* masks, float* fuses, void* workspace,
              size_t workspace_bytes, void* hip_stream) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (T < 1 || !key_frame || !labels || !dains || !fuses || !workspace) return fail(h, RIB_ERR_INVALID, "rib_chain: bad argument");
  // (the NULL stream cannot be captured: a caller on it gets the launch-by-launch path)
  if (!h->graph_replay || h->profiling || h->device < 0 || hip_stream == nullptr)
    return chain_enqueue(h, T, B, H, W, key_frame, labels, dains, imgs, masks, fuses, workspace, workspace_bytes, hip_stream);
  // ---- graph replay: the T x ~130 launches of the segment as ONE graph launch.  Every kernel parameter is a function of the
  // shape and of the pointers of this call, so that tuple is the key; a call with other tensors captures its own graph (the
  // enqueue code runs under stream capture, nothing executes), at most 8 are kept.  Same kernels, same parameters, same order
  // on the same stream: the frames are bit-identical to the launch-by-launch path.
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  const std::array<uintptr_t, 12> key = {(uintptr_t)T, (uintptr_t)B, (uintptr_t)H, (uintptr_t)W, (uintptr_t)key_frame, (uintptr_t)labels,
                                         (uintptr_t)dains, (uintptr_t)imgs, (uintptr_t)masks, (uintptr_t)fuses, (uintptr_t)workspace,
                                         (uintptr_t)workspace_bytes};
  for (auto& g : h->chain_graphs)
    if (g.key == key) {
      g.used = ++h->graph_clock; ++h->graph_replays; g.stream = st;
      HIP_TRY(h, hipGraphLaunch(g.exec, st));
      return RIB_OK;
    }
  // plans (and the Winograd filter sets they allocate) must exist before the capture starts: building one synchronises
  if (!get_plan(h, B, H, W, chain_batches_labels(h, T, B) ? PLAN_UNPAIRED : 0)) return RIB_ERR_INVALID;
  if (chain_batches_labels(h, T, B) && !get_plan(h, T * B, H, W, PLAN_LABELS, B)) return RIB_ERR_INVALID;
  hipGraph_t graph = nullptr;
  HIP_TRY(h, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  rc = chain_enqueue(h, T, B, H, W, key_frame, labels, dains, imgs, masks, fuses, workspace, workspace_bytes, hip_stream);
  const hipError_t ec = hipStreamEndCapture(st, &graph);
  if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  if (ec != hipSuccess) return fail(h, RIB_ERR_HIP, fmt("rib_chain: stream capture failed: %s", hipGetErrorString(ec)));
  hipGraphExec_t exec = nullptr;
  const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ei != hipSuccess) return fail(h, RIB_ERR_HIP, fmt("rib_chain: graph instantiation failed: %s", hipGetErrorString(ei)));
  if (h->chain_graphs.size() >= 8) {
    size_t lru = 0;
    for (size_t i = 1; i < h->chain_graphs.size(); ++i) if (h->chain_graphs[i].used < h->chain_graphs[lru].used) lru = i;
    destroy_chain_graph(h->chain_graphs[lru]);
    h->chain_graphs.erase(h->chain_graphs.begin() + lru);
  }
  h->chain_graphs.push_back({key, exec, ++h->graph_clock, st});
  ++h->graph_captures;
  HIP_TRY(h, hipGraphLaunch(exec, st));
  return RIB_OK;
}

static int chain_enqueue(rib_handle* h, int T, int B, int H, int W, const float* key_frame, const float* labels,
                         const float* dains, float* imgs, float* masks, float* fuses, void* workspace,
                         size_t workspace_bytes, void* hip_stream) {
  int rc = RIB_OK;
  // a chain that runs the label-only launches once per segment needs them apart from the image encoder's: the unpaired
  // twin of the frame plan (same kernels and choices: the frames equal rib_forward's bit for bit)
  Plan* P = get_plan(h, B, H, W, chain_batches_labels(h, T, B) ? PLAN_UNPAIRED : 0);
  if (!P) return RIB_ERR_INVALID;
  const rib_config& c = h->g.c;
  const size_t frame = (size_t)B * c.image_nc * H * W, mframe = (size_t)B * H * W, lframe = (size_t)B * c.label_nc * H * W;
  const size_t need = P->ws_bytes + 2 * align256(frame * sizeof(float)) + align256(mframe * sizeof(float));
  if (workspace_bytes < need) return fail(h, RIB_ERR_WORKSPACE, fmt("workspace %zu < required %zu bytes", workspace_bytes, need));
  char* wsb = reinterpret_cast<char*>(workspace);
  float* tmp_img = reinterpret_cast<float*>(wsb + P->ws_bytes);
  float* tmp_mask = reinterpret_cast<float*>(wsb + P->ws_bytes + align256(frame * sizeof(float)));
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  // The label maps of the whole segment are known up front: pack.label, down_first and the mask network's label
  // branch (8 + 9 launches, ~0.22 ms of a 3.0 ms frame at 512x512) run once at batch T*B, where the small maps fill
  // the chip, when the caller's workspace has room for it (rib_chain_workspace_bytes); each frame then copies its
  // slices into the frame plan's slots and skips those launches.
  Plan* PL = nullptr;
  char* lws = nullptr;
  if (chain_batches_labels(h, T, B)) {
    PL = get_plan(h, T * B, H, W, PLAN_LABELS, B);
    if (!PL) return RIB_ERR_INVALID;
    const size_t off = align256(need);
    if (workspace_bytes >= off + PL->ws_bytes) {
      lws = wsb + off;
      Resolver RL; RL.ws = lws; RL.blob = h->d_blob; RL.h = h;
      for (int u = 0; u < U_COUNT; ++u) RL.user[u] = nullptr;
      RL.user[U_LABEL] = labels;
      rc = run_plan(h, PL, RL, st);
      if (rc) return rc;
    } else PL = nullptr;     // a caller that sized the workspace with rib_workspace_bytes: per-frame label work
  }
  bool chain_fused = false;
  for (const Op& op : P->ops) chain_fused = chain_fused || op.fuse_blend;
  const float* prev = key_frame;   // evaluator.py:240-244: a segment starts from the ground-truth key frame
  for (int t = 0; t < T; ++t) {
    if (PL) {
      const LabelSlots& a = PL->ls; const LabelSlots& b = P->ls;   // a: [T*B] batch, b: [B] batch
      struct { size_t src, dst, bytes; } cp[6] = {{a.x0, b.x0, b.x0_b}, {a.nx_sc, b.nx_sc, b.nx_b}, {a.nx_sh, b.nx_sh, b.nx_b},
                                                   {a.cat, b.cat, b.cat_b}, {a.ncat_sc, b.ncat_sc, b.ncat_b}, {a.ncat_sh, b.ncat_sh, b.ncat_b}};
      Gather6Params gp;
      for (int k = 0; k < 6; ++k) {
        gp.src[k] = reinterpret_cast<const float4*>(lws + cp[k].src + (size_t)t * cp[k].bytes);
        gp.dst[k] = reinterpret_cast<float4*>(wsb + cp[k].dst);
        gp.n4[k] = (unsigned)(cp[k].bytes / 16);       // every slot is a whole number of 8-channel (32-byte) groups
      }
      RIB_KLAUNCH(k_gather6, dim3(1024, 6), dim3(256), 0, st, gp);
    }
    float* img_t = imgs ? imgs + (size_t)t * frame : tmp_img;
    float* mask_t = masks ? masks + (size_t)t * mframe : tmp_mask;
    float* fuse_t = fuses + (size_t)t * frame;
    Resolver R; R.ws = wsb; R.blob = h->d_blob; R.h = h;
    R.user[U_LABEL] = labels + (size_t)t * lframe; R.user[U_FAKE] = dains + (size_t)t * frame; R.user[U_PREV] = prev;
    R.user[U_IMG] = img_t; R.user[U_MASK] = mask_t;
    R.user[U_FUSE] = chain_fused ? fuse_t : nullptr;     // the mask head writes the blend itself when it can
    rc = run_plan(h, P, R, st, PL != nullptr);
    if (rc) return rc;
    if (!chain_fused) {
      rc = rib_blend(h, B, c.image_nc, H, W, img_t, mask_t, dains + (size_t)t * frame, fuse_t, hip_stream);
      if (rc) return rc;
    }
    prev = fuse_t;                 // evaluator.py:252: prev_img = results['fuse'][-1]
  }
  return RIB_OK;
}

int rib_set_debug_taps(rib_handle* h, int enable) {
  if (!h) return RIB_ERR_INVALID;
  if (h->keep_taps != (enable != 0)) { h->plans.clear(); drop_chain_graphs(h); }   // the workspace layout depends on it
  h->keep_taps = enable != 0;
  return RIB_OK;
}

int rib_num_taps(rib_handle* h, int B, int H, int W) {
  if (!h) return RIB_ERR_INVALID;
  Plan* P = get_plan(h, B, H, W);
  return P ? (int)P->taps.size() : RIB_ERR_INVALID;
}

int rib_tap_info(rib_handle* h, int B, int H, int W, int idx, const char** name, int* C, int* th, int* tw) {
  if (!h) return RIB_ERR_INVALID;
  Plan* P = get_plan(h, B, H, W);
  if (!P || idx < 0 || idx >= (int)P->taps.size()) return RIB_ERR_INVALID;
  const Tap& t = P->taps[idx];
  if (name) *name = t.name.c_str();
  if (C) *C = t.C;
  if (th) *th = t.H;
  if (tw) *tw = t.W;
  return RIB_OK;
}

int rib_read_tap(rib_handle* h, int B, int H, int W, int idx, const void* workspace, float* dst, void* hip_stream) {
  if (!h || !workspace || !dst) return RIB_ERR_INVALID;
  if (!h->keep_taps) return fail(h, RIB_ERR_STATE, "rib_read_tap: call rib_set_debug_taps(h, 1) before the forward (intermediate buffers are reused otherwise)");
  Plan* P = get_plan(h, B, H, W);
  if (!P || idx < 0 || idx >= (int)P->taps.size()) return RIB_ERR_INVALID;
  const Tap& t = P->taps[idx];
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  const float* src = reinterpret_cast<const float*>(reinterpret_cast<const char*>(workspace) + t.off);
  RIB_LAUNCH_ST(h->prec(), k_unpack, dim3((t.H * t.W + 255) / 256, B), dim3(256), 0, st, src, t.Cp, t.C, t.H * t.W, 0, t.H, t.W, dst);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(st));
  return RIB_OK;
}

// Inverse of the filter re-layout, from a host-only handle: used by the CPU tests to check the
// spectral-norm fold and the [CoutPad][tap][CinPad] / SPADE column layouts against the oracle.
int rib_debug_conv_weight(rib_handle* h, const char* conv_name, float* w_oihw, float* bias) {
  if (!h || !conv_name || !w_oihw || !bias) return RIB_ERR_INVALID;
  if (h->host_blob.empty()) return fail(h, RIB_ERR_STATE, "rib_debug_conv_weight needs a finalized host-only handle (device < 0)");
  auto it = h->conv_index.find(conv_name);
  if (it == h->conv_index.end() || !h->convs[it->second].used) return fail(h, RIB_ERR_INVALID, fmt("unknown conv '%s'", conv_name));
  const ConvDef& c = h->convs[it->second];
  const int taps = c.ks * c.ks;
  for (int o = 0; o < c.cout; ++o) {
    for (int i = 0; i < c.cin; ++i)
      for (int t = 0; t < taps; ++t)
        w_oihw[((size_t)o * c.cin + i) * taps + t] = h->host_blob[c.w_off + ((size_t)o * taps + t) * c.cinp + i];
    bias[o] = h->host_blob[c.b_off + o];
  }
  return RIB_OK;
}

int rib_debug_spade_weight(rib_handle* h, const char* conv_name, float* w_2c_by_cond, float* bias_2c) {
  if (!h || !conv_name || !w_2c_by_cond || !bias_2c) return RIB_ERR_INVALID;
  if (h->host_blob.empty()) return fail(h, RIB_ERR_STATE, "rib_debug_spade_weight needs a finalized host-only handle (device < 0)");
  const std::string cn = conv_name;
  const size_t pos = cn.rfind(".conv_block_");
  if (pos == std::string::npos) return fail(h, RIB_ERR_INVALID, "not a SPADE conv block");
  const std::string blk = cn.substr(0, pos), which = cn.substr(cn.size() - 1);
  const std::string key = blk + "." + (which == "s" ? std::string("0") : which);
  auto it = h->spade_index.find(key);
  if (it == h->spade_index.end()) return fail(h, RIB_ERR_INVALID, fmt("no SPADE group '%s'", key.c_str()));
  const SpadeGroup& sg = h->spades[it->second];
  const int set = which == "s" ? 1 : 0;
  const int condp = h->padc(sg.cond);
  for (int ch = 0; ch < sg.C; ++ch) {
    const int v = set * sg.Cp + ch;
    const int colg = (v / 32) * 64 + (v % 32), colb = colg + 32;
    for (int k = 0; k < sg.cond; ++k) {
      w_2c_by_cond[(size_t)ch * sg.cond + k] = h->host_blob[sg.w_off + (size_t)colg * condp + k];
      w_2c_by_cond[(size_t)(sg.C + ch) * sg.cond + k] = h->host_blob[sg.w_off + (size_t)colb * condp + k];
    }
    bias_2c[ch] = h->host_blob[sg.b_off + colg];
    bias_2c[sg.C + ch] = h->host_blob[sg.b_off + colb];
  }
  return RIB_OK;
}

// Launch list of a plan, for the CPU-side structure tests: "<name>|<kernel class>|<grid>|<tile>"
static int launch_info_head(const Op& op, char* buf, size_t buflen) {
  if (op.kind == OP_IGEMM && op.small_co > 0)
    snprintf(buf, buflen, "%s|%d|%u,%u,%u|%s 16x16 tile, %d output channels%s|%.0f", op.name.c_str(), op.kclass, op.grid.x, op.grid.y, op.grid.z,
             op.head ? "head (taps as MFMA columns)" : "direct (vector ALUs)", op.small_co, op.fuse_blend ? " + fused blend" : "", op.flops);
  else if (op.kind == OP_IGEMM)
    snprintf(buf, buflen, "%s|%d|%u,%u,%u|tile %dx%d BN %d BK %d s%d k%d ups%d ksplit%d kw%d tb%d%s v%d|%.0f", op.name.c_str(), op.kclass, op.grid.x, op.grid.y, op.grid.z,
             op.var->TH(), op.var->TW(), op.var->BN(), op.var->BK, op.var->STRIDE, op.var->KS, (int)op.var->UPS, op.ip.ksplit, op.var->KW, op.var->DMAK ? 100 + (op.var->TB == 9 ? 9 : 0) : op.var->TB,
             op.wino ? (op.wino_m == 4 ? " wino4" : " wino") : "", (int)(op.var - kVariants), op.flops);      // (v<n>: index into rib_variant_info)
  else if (op.kind == OP_GEMM)
    snprintf(buf, buflen, "%s|%d|%u,%u,%u|gemm (LDS-DMA staged operands) tile %dx%d BK 32, %d x [%d x %d x %d]%s|%.0f", op.name.c_str(), op.kclass, op.grid.x, op.grid.y, op.grid.z,
             op.var->BM(), op.var->BN(), (int)op.grid.z, op.gp.M, op.gp.N, op.gp.K, op.wino ? (op.wino_m == 4 ? " wino4" : " wino") : "", op.flops);
  else if (op.kind == OP_LOWC)
    snprintf(buf, buflen, "%s|%d|%u,%u,%u|lowc (caller's NCHW tensors, K = 9 x %d real channels) 8x%d tile, %d columns|%.0f", op.name.c_str(), op.kclass,
             op.grid.x, op.grid.y, op.grid.z, op.lowc_ce, op.lowc_tw, op.lowc_ncol, op.flops);
  else
    snprintf(buf, buflen, "%s|%d|%u,%u,%u||0", op.name.c_str(), op.kclass, op.grid.x, op.grid.y, op.grid.z);
  return RIB_OK;
}

// Algorithmic HBM bytes of one launch (tools/prof_ops.py prices every launch against max(FLOPs / MFMA peak, these bytes / HBM)):
// every operand the launch NEEDS read once, every result written once - activations in the storage type, filters and
// statistics as stored; halo re-reads, split-K re-reads and L2 misses are exactly what this leaves out.
static double op_algorithmic_bytes(const rib_handle* h, const Plan* P, const Op& op) {
  const double e = h->esz(), Bn = P->B;
  switch (op.kind) {
    case OP_IGEMM: {
      const IgemmParams& p = op.ip;
      const double samples = op.ip.pair ? 2.0 * Bn : Bn;      // a paired launch carries two convolutions per image
      const int taps = op.var ? op.var->KS * op.var->KS : 9;
      double b = samples * p.Hin * p.Win * (double)p.Cin * e                         // input (or SPADE condition map)
                 + (double)p.CoutPad * taps * p.Cin * e * (op.ip.pair ? 2.0 : 1.0)   // filters
                 + p.CoutPad * 4.0;                                                    // bias
      if (op.var && op.var->SPADE) {
        b += Bn * p.Hout * p.Wout * (double)p.C * e / (p.xm_ups ? 4.0 : 1.0);          // tensor being normalised
        b += Bn * p.Hout * p.Wout * (double)p.C * e * p.nsets;                         // modulated outputs
      } else if (p.slab) {
        b += (double)p.ksplit * samples * p.Hout * p.Wout * p.CoutPad * 4.0;           // split-K partial slabs (fp32)
      } else {
        b += samples * p.Hout * p.Wout * (double)p.Cout * (op.small_co > 0 || p.y_f32 ? 4.0 : e);
        if (p.res) b += samples * p.Hout * p.Wout * (double)p.Cout * e / (p.res_ups ? 4.0 : 1.0);
        if (p.x2) b += samples * p.Hout * p.Wout * (double)p.Cin2 * e + (double)p.CoutPad * p.Cin2 * e;
        if (p.bl_fuse) b += Bn * p.Hout * p.Wout * 3.0 * 4.0 * 3.0;                     // blend: img + dain read, fuse written (fp32 NCHW)
      }
      return b;
    }
    case OP_GEMM: {
      const GemmDmaParams& g = op.gp;
      const double Z = op.grid.z;
      return Z * g.M * (double)g.K * e + (g.modB ? (double)g.modB : 1.0) * g.N * (double)g.K * e + Z * g.M * (double)g.N * 4.0;
    }
    case OP_LOWC: {
      const LowcParams& l = op.lc;
      return Bn * l.H * l.W * (double)(l.c0 + l.c1 + l.c2) * 4.0 + Bn * l.H * l.W * (double)l.Cout * e + 9.0 * (l.c0 + l.c1 + l.c2) * l.Cout * 4.0;
    }
    case OP_POOL: return Bn * op.pp.H * op.pp.W * (double)op.pp.C * e * 1.25;
    case OP_INADD: return Bn * op.ap.HW * (double)op.ap.C * e * 3.0;
    case OP_SPLITEPI: {
      const SplitEpiParams& q = op.sp;
      return (double)q.ksplit * q.B * q.Hout * q.Wout * q.CoutPad * 4.0 + (double)q.B * q.Hout * q.Wout * q.Cout * e * (q.res ? 2.0 : 1.0);
    }
    case OP_MODULATE: {
      const ModulateParams& m = op.mp;
      const double px = (double)m.B * m.Hout * m.Wout;
      return px * 2.0 * m.C * m.nsets * 4.0 * m.ksplit + px * m.C * e / (m.xm_ups ? 4.0 : 1.0) + px * m.C * e * m.nsets;
    }
    case OP_WINO_IN: {
      const WinoInParams& w = op.wi;
      const int T = op.wino_m + 2;
      double b = Bn * w.H * w.W * (double)w.Cin * e / (w.x_ups ? 4.0 : 1.0) + Bn * w.tilesY * w.tilesX * (double)T * T * w.Cin * e;
      if (op.wi_mode != 0 && w.slab) b += Bn * w.H * w.W * 2.0 * w.Cin * 4.0;          // gamma/beta columns of the level slab
      if (w.x2 || w.xres) b += Bn * w.H * w.W * (double)w.Cin * e * 2.0;               // join: second operand read, join stored
      return b;
    }
    case OP_WINO_OUT: {
      const WinoOutParams& w = op.wo;
      const int T = op.wino_m + 2;
      return Bn * w.tilesY * w.tilesX * (double)T * T * w.CoutPad * 4.0 + Bn * w.Hout * w.Wout * (double)w.Cout * e * (w.res ? 2.0 : 1.0);
    }
    case OP_FINALIZE: return Bn * op.fp.tiles * 2.0 * op.fp.Cs * 8.0;
    default: return 0.0;
  }
}

int rib_debug_launch_info(rib_handle* h, int B, int H, int W, int idx, char* buf, size_t buflen) {
  if (!h || !buf) return RIB_ERR_INVALID;
  Plan* P = get_plan(h, B, H, W);
  if (!P || idx < 0 || idx >= (int)P->ops.size()) return RIB_ERR_INVALID;
  const Op& op = P->ops[idx];
  const int rc = launch_info_head(op, buf, buflen);
  const size_t n = strlen(buf);
  if (n + 24 < buflen) snprintf(buf + n, buflen - n, "|%.0f", op_algorithmic_bytes(h, P, op));
  return rc;
}

// ---- tuning hooks: enumerate kernel variants, pin a (variant, split-K) choice for one op of one
// shape, and time a single op of the plan in isolation ----
const char* rib_build_info(void) {
  static const std::string info = [] {
#define RIB_X(s) rib_stamp_section_##s,
    const char* sh[RIB_NSECTIONS] = {RIB_FOR_SECTIONS(RIB_X)};
#undef RIB_X
    std::string s = std::string("librib stamp=") + (kLibStamp + sizeof("rib-stamp lib ") - 1) + " shards=";
    bool ok = true;
    for (int i = 0; i < RIB_NSECTIONS; ++i) {
      const char* hash = strrchr(sh[i], ' ');          // "rib-stamp shard<i> <hash>"
      hash = hash ? hash + 1 : "?";
      ok = ok && strcmp(hash, RIB_SHARD_STAMP) == 0;
      s += std::string(i ? "," : "") + hash;
    }
    s += fmt(" consistent=%d variants=%d compiler=%s", ok ? 1 : 0, kNumVariants, __VERSION__);
    return s;
  }();
  return info.c_str();
}

int rib_num_variants(void) { return kNumVariants; }

int rib_variant_info(int idx, int geom[12]) {
  if (idx < 0 || idx >= kNumVariants || !geom) return RIB_ERR_INVALID;
  const Variant& v = kVariants[idx];
  const int g[12] = {v.FRW, v.WM, v.WN, v.MF, v.NF, v.BK, v.STRIDE, v.KS, v.UPS ? 1 : 0, v.SPADE ? 1 : 0, v.KW, v.DMAK ? 100 + (v.TB == 9 ? 9 : 0) : v.TB};
  for (int i = 0; i < 12; ++i) geom[i] = g[i];
  return v.BF16;   // precision of the instantiation: 0 fp32, 1 bf16 storage, 2 half storage
}

int rib_set_choice(rib_handle* h, int B, int H, int W, const char* op_name, int variant_idx, int ksplit) {
  if (!h || !op_name) return RIB_ERR_INVALID;
  const std::string key = fmt("%d,%d,%d|%s", B, H, W, op_name);
  if (variant_idx < 0) h->choices.erase(key);
  else {
    if (variant_idx >= kNumVariants || ksplit < 1) return fail(h, RIB_ERR_INVALID, "rib_set_choice: bad variant / ksplit");
    h->choices[key] = {variant_idx, ksplit};
  }
  drop_chain_graphs(h);
  // the frame plans of this shape (paired and unpaired) are rebuilt on next use, and so are the labels-only plans of
  // chains that follow this shape's choices (key {flags, tuneB, batch, H, W})
  for (auto it = h->plans.begin(); it != h->plans.end();) {
    const std::array<int, 5>& k = it->first;
    const bool same_hw = k[3] == H && k[4] == W;
    const bool follows = same_hw && (k[1] > 0 ? k[1] : k[2]) == B;      // the batch whose choices the plan follows
    if (follows) it = h->plans.erase(it); else ++it;
  }
  return RIB_OK;
}

int rib_time_op(rib_handle* h, int B, int H, int W, const char* op_name, const float* label, const float* img_fake,
                const float* img_prev, float* img, float* mask, void* workspace, size_t workspace_bytes, int iters,
                void* hip_stream, double* usec) {
  int rc = check_ready(h);
  if (rc) return rc;
  if (!op_name || !workspace || !usec || iters < 1) return RIB_ERR_INVALID;
  Plan* P = get_plan(h, B, H, W);
  if (!P) return RIB_ERR_INVALID;
  if (workspace_bytes < P->ws_bytes) return fail(h, RIB_ERR_WORKSPACE, "rib_time_op: workspace too small");
  // a sub-plan holding the op and, for split-K, its slab-summing epilogue
  Plan sub; sub.B = B; sub.H = H; sub.W = W;
  const std::string nm = op_name;
  for (const Op& op : P->ops)
    if (op.name == nm || op.name == nm + ".splitk_sum" || op.name == nm + ".modulate" || op.for_op == nm) sub.ops.push_back(op);
  if (sub.ops.empty()) return fail(h, RIB_ERR_INVALID, fmt("rib_time_op: no op named '%s'", op_name));
  Resolver R; R.ws = reinterpret_cast<char*>(workspace); R.blob = h->d_blob; R.h = h;
  R.user[U_LABEL] = label; R.user[U_FAKE] = img_fake; R.user[U_PREV] = img_prev; R.user[U_IMG] = img; R.user[U_MASK] = mask;
  R.user[U_FUSE] = nullptr;
  hipStream_t st = reinterpret_cast<hipStream_t>(hip_stream);
  const bool was = h->profiling; h->profiling = false;
  rc = run_plan(h, &sub, R, st);   // warm-up
  hipEvent_t e0, e1;
  HIP_TRY(h, hipEventCreate(&e0)); HIP_TRY(h, hipEventCreate(&e1));
  HIP_TRY(h, hipEventRecord(e0, st));
  for (int i = 0; i < iters && rc == RIB_OK; ++i) rc = run_plan(h, &sub, R, st);
  HIP_TRY(h, hipEventRecord(e1, st));
  HIP_TRY(h, hipEventSynchronize(e1));
  float ms = 0.f;
  HIP_TRY(h, hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  h->profiling = was;
  *usec = (double)ms * 1e3 / iters;
  return rc;
}

static void drop_prof_events(rib_handle* h) {
  for (auto& pe : h->prof_events) (void)hipEventDestroy(pe.second);
  h->prof_events.clear();
  for (auto& ke : h->prof_kernel_events) { (void)hipEventDestroy(ke.start); (void)hipEventDestroy(ke.stop); }
  h->prof_kernel_events.clear();
}

int rib_profile_begin(rib_handle* h) {
  if (!h) return RIB_ERR_INVALID;
  drop_prof_events(h);
  h->profiling = true; h->prof_kernels = false;
  return RIB_OK;
}

int rib_profile_begin_kernels(rib_handle* h) {
  if (!h) return RIB_ERR_INVALID;
  drop_prof_events(h);
  h->profiling = true; h->prof_kernels = true;
  return RIB_OK;
}

int rib_profile_collect(rib_handle* h, int64_t launches[RIB_KC_COUNT], double ms[RIB_KC_COUNT]) {
  if (!h || !launches || !ms) return RIB_ERR_INVALID;
  h->profiling = false;
  for (int i = 0; i < RIB_KC_COUNT; ++i) { launches[i] = 0; ms[i] = 0.0; }
  if (h->prof_kernels) {
    if (!h->prof_kernel_events.empty()) HIP_TRY(h, hipEventSynchronize(h->prof_kernel_events.back().stop));
    for (const auto& ke : h->prof_kernel_events) {
      float t = 0.f;
      HIP_TRY(h, hipEventElapsedTime(&t, ke.start, ke.stop));
      launches[ke.kclass] += 1; ms[ke.kclass] += (double)t;
    }
    drop_prof_events(h);
    h->prof_kernels = false;
    return RIB_OK;
  }
  if (!h->prof_events.empty()) HIP_TRY(h, hipEventSynchronize(h->prof_events.back().second));
  for (size_t i = 0; i + 1 < h->prof_events.size(); ++i) {
    const int kc = h->prof_events[i].first;
    if (kc < 0) continue;                       // end marker of a plan run: the gap to the next run belongs to no launch
    float t = 0.f;
    HIP_TRY(h, hipEventElapsedTime(&t, h->prof_events[i].second, h->prof_events[i + 1].second));
    launches[kc] += 1; ms[kc] += (double)t;
  }
  drop_prof_events(h);
  return RIB_OK;
}

int rib_forward_flops(rib_handle* h, int B, int H, int W, double flops[RIB_KC_COUNT]) {
  if (!h || !flops) return RIB_ERR_INVALID;
  Plan* P = get_plan(h, B, H, W);
  if (!P) return RIB_ERR_INVALID;
  for (int i = 0; i < RIB_KC_COUNT; ++i) flops[i] = P->flops[i];
  return RIB_OK;
}

int rib_num_launches(rib_handle* h, int B, int H, int W) {
  if (!h) return RIB_ERR_INVALID;
  Plan* P = get_plan(h, B, H, W);
  return P ? (int)P->ops.size() : RIB_ERR_INVALID;
}

}  // extern "C"
