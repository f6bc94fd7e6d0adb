// variants.hip.h - which k_igemm instantiations a table entry of variants.def stands for.
//
// RIB_I_<kind>(F, args...) applies F to the template-argument list of every instantiation of an entry:
//   RIB_F_EXTERN  -> `extern template` declaration (rib.hip: the code lives in a shard object)
//   RIB_F_TOUCH   -> `&k_igemm<...>,` (igemm_shard.hip: taking the address instantiates the kernel there)
// rib.hip's RIB_V / RIB_VK / ... table macros name the same instantiations; keep the two in step.
#pragma once

// The entries of variants.def are dealt to RIB_NSECTIONS shard objects (igemm_shard.hip compiled once per section with
// -DRIB_SECTION=<s> -DRIB_ON_<s>=RIB_KEEP; csrc/build.py reads the count from this line and runs the compiles as a job queue:
// more sections than cores, so that the few slow entries - the fully unrolled phase convolutions - do not make one object the
// long pole of the build).
#define RIB_NSECTIONS 24
#define RIB_FOR_SECTIONS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23)

// template parameters of k_igemm: FRW, WM, WN, MF, NF, BK, STRIDE, KS, UPS, SPADE, PREC (0 fp32 / 1 bf16 / 2 half; false = 0), AUX, PRO, KW, TB, DMA
#define RIB_F_EXTERN(...) extern template __global__ void rib::k_igemm<__VA_ARGS__>(const rib::IgemmParams);
#define RIB_F_TOUCH(...) &rib::k_igemm<__VA_ARGS__>,

// convolution geometry: generic (fused-shortcut loop + prologue) / no shortcut loop / lean
#define RIB_I_V(F, FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP)                                   \
  F(FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, false, (KS == 3 && S == 1 && !UPS), true)        \
  F(FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, false, false, true)                              \
  F(FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, false, false, false)
// in-workgroup split-K twin (KW wave groups)
#define RIB_I_VK(F, FRW, WM, WN, MF, NF, BK, S, KS, UPS, KW)                                       \
  F(FRW, WM, WN, MF, NF, BK, S, KS, UPS, false, false, (KS == 3 && S == 1 && !UPS), true, KW)      \
  F(FRW, WM, WN, MF, NF, BK, S, KS, UPS, false, false, false, true, KW)                            \
  F(FRW, WM, WN, MF, NF, BK, S, KS, UPS, false, false, false, false, KW)
// three filter slices per barrier
#define RIB_I_VT(F, FRW, WM, WN, MF, NF, BK, S, UPS)                                      \
  F(FRW, WM, WN, MF, NF, BK, S, 3, UPS, false, false, (S == 1 && !UPS), true, 1, 3)       \
  F(FRW, WM, WN, MF, NF, BK, S, 3, UPS, false, false, false, true, 1, 3)                  \
  F(FRW, WM, WN, MF, NF, BK, S, 3, UPS, false, false, false, false, 1, 3)
#define RIB_I_VTK(F, FRW, WM, WN, MF, NF, BK, S, KW)                                      \
  F(FRW, WM, WN, MF, NF, BK, S, 3, false, false, false, (S == 1), true, KW, 3)            \
  F(FRW, WM, WN, MF, NF, BK, S, 3, false, false, false, false, true, KW, 3)               \
  F(FRW, WM, WN, MF, NF, BK, S, 3, false, false, false, false, false, KW, 3)
// all nine filter slices per barrier pair
#define RIB_I_V9(F, FRW, WM, WN, MF, NF, BK, S, KW)                                       \
  F(FRW, WM, WN, MF, NF, BK, S, 3, false, false, false, (S == 1), true, KW, 9)            \
  F(FRW, WM, WN, MF, NF, BK, S, 3, false, false, false, false, true, KW, 9)               \
  F(FRW, WM, WN, MF, NF, BK, S, 3, false, false, false, false, false, KW, 9)
// phase-decomposed upsample convolution, four taps of a phase per barrier
#define RIB_I_VU4(F, FRW, WM, WN, MF, NF, BK)                                             \
  F(FRW, WM, WN, MF, NF, BK, 1, 3, true, false, false, false, true, 1, 4)                 \
  F(FRW, WM, WN, MF, NF, BK, 1, 3, true, false, false, false, false, 1, 4)
// SPADE geometry / its in-workgroup split-K twin / 16-bit matrix-core twins (bf16 and half)
#define RIB_I_VS(F, FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP) F(FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, false)
#define RIB_I_VSK(F, FRW, WM, WN, MF, NF, BK, KW) F(FRW, WM, WN, MF, NF, BK, 1, 1, false, true, false, true, true, KW)
#define RIB_I_VB(F, FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP) F(FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, 1) F(FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, 2)
// bf16-storage variant with explicit wave groups / slices per barrier (one instantiation: shortcut loop where it can exist, prologue)
#define RIB_I_VBX(F, FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, KW, TB) \
  F(FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, 1, (KS == 3 && S == 1 && !UPS && !SP), true, KW, TB) \
  F(FRW, WM, WN, MF, NF, BK, S, KS, UPS, SP, 2, (KS == 3 && S == 1 && !UPS && !SP), true, KW, TB)
// 1x1 with input tile and filter slice double-buffered (TB = 2: one barrier per chunk), optional wave groups; conv (generic = pro, lean) and SPADE
#define RIB_I_V1D(F, FRW, WM, WN, MF, NF, BK, KW)                                \
  F(FRW, WM, WN, MF, NF, BK, 1, 1, false, false, 0, false, true, KW, 2)          \
  F(FRW, WM, WN, MF, NF, BK, 1, 1, false, false, 0, false, false, KW, 2)
#define RIB_I_VS1D(F, FRW, WM, WN, MF, NF, BK, KW) F(FRW, WM, WN, MF, NF, BK, 1, 1, false, true, 0, true, true, KW, 2)
// operand tiles staged by LDS-DMA: prologue instantiation (filters by DMA) + lean instantiation (filters and input tile); fused SPADE
#define RIB_I_VD(F, FRW, WM, WN, MF, NF, BK, S, KS)                      \
  F(FRW, WM, WN, MF, NF, BK, S, KS, false, false, 0, false, true, 1, 1, 1) \
  F(FRW, WM, WN, MF, NF, BK, S, KS, false, false, 0, false, false, 1, 1, 3)
#define RIB_I_VSD(F, FRW, WM, WN, MF, NF, BK) F(FRW, WM, WN, MF, NF, BK, 1, 1, false, true, 0, false, false, 1, 1, 3)
#define RIB_I_VD9(F, FRW, WM, WN, MF, NF, BK, S) F(FRW, WM, WN, MF, NF, BK, S, 3, false, false, 0, false, false, 1, 9, 3)
