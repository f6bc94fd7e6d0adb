// igemm_shard.hip - one section of the k_igemm tile variants (variants.def), compiled once per section
// with -DRIB_SECTION=<s> -DRIB_ON_<s>=RIB_KEEP (s < RIB_NSECTIONS, variants.hip.h) so that the kernel code generation runs as parallel hipcc jobs.
// Taking a kernel's address instantiates it in this object; rib.hip refers to it through `extern template`.
#include "igemm.hip.h"
#include "variants.hip.h"

#ifndef RIB_SECTION
#error "compile with -DRIB_SECTION=<n>"
#endif

#define RIB_CAT_(a, b) a##b
#define RIB_CAT(a, b) RIB_CAT_(a, b)
#define RIB_KEEP(...) __VA_ARGS__
#define RIB_DROP(...)
// RIB_ON_<s> keeps its arguments only for this object's section: the build defines RIB_ON_<RIB_SECTION> as RIB_KEEP on the
// command line, every other section's selector falls back to RIB_DROP here
#ifndef RIB_ON_0
#define RIB_ON_0 RIB_DROP
#endif
#ifndef RIB_ON_1
#define RIB_ON_1 RIB_DROP
#endif
#ifndef RIB_ON_2
#define RIB_ON_2 RIB_DROP
#endif
#ifndef RIB_ON_3
#define RIB_ON_3 RIB_DROP
#endif
#ifndef RIB_ON_4
#define RIB_ON_4 RIB_DROP
#endif
#ifndef RIB_ON_5
#define RIB_ON_5 RIB_DROP
#endif
#ifndef RIB_ON_6
#define RIB_ON_6 RIB_DROP
#endif
#ifndef RIB_ON_7
#define RIB_ON_7 RIB_DROP
#endif
#ifndef RIB_ON_8
#define RIB_ON_8 RIB_DROP
#endif
#ifndef RIB_ON_9
#define RIB_ON_9 RIB_DROP
#endif
#ifndef RIB_ON_10
#define RIB_ON_10 RIB_DROP
#endif
#ifndef RIB_ON_11
#define RIB_ON_11 RIB_DROP
#endif
#ifndef RIB_ON_12
#define RIB_ON_12 RIB_DROP
#endif
#ifndef RIB_ON_13
#define RIB_ON_13 RIB_DROP
#endif
#ifndef RIB_ON_14
#define RIB_ON_14 RIB_DROP
#endif
#ifndef RIB_ON_15
#define RIB_ON_15 RIB_DROP
#endif
#ifndef RIB_ON_16
#define RIB_ON_16 RIB_DROP
#endif
#ifndef RIB_ON_17
#define RIB_ON_17 RIB_DROP
#endif
#ifndef RIB_ON_18
#define RIB_ON_18 RIB_DROP
#endif
#ifndef RIB_ON_19
#define RIB_ON_19 RIB_DROP
#endif
#ifndef RIB_ON_20
#define RIB_ON_20 RIB_DROP
#endif
#ifndef RIB_ON_21
#define RIB_ON_21 RIB_DROP
#endif
#ifndef RIB_ON_22
#define RIB_ON_22 RIB_DROP
#endif
#ifndef RIB_ON_23
#define RIB_ON_23 RIB_DROP
#endif
#if RIB_SECTION < 0 || RIB_SECTION >= RIB_NSECTIONS
#error "RIB_SECTION out of range"
#endif

#define RIB_V(sec, ...) RIB_CAT(RIB_ON_, sec)(RIB_I_V(RIB_F_TOUCH, __VA_ARGS__))
#define RIB_VK(sec, ...) RIB_CAT(RIB_ON_, sec)(RIB_I_VK(RIB_F_TOUCH, __VA_ARGS__))
#define RIB_VT(sec, ...) RIB_CAT(RIB_ON_, sec)(RIB_I_VT(RIB_F_TOUCH, __VA_ARGS__))
#define RIB_VTK(sec, ...) RIB_CAT(RIB_ON_, sec)(RIB_I_VTK(RIB_F_TOUCH, __VA_ARGS__))
#define RIB_V9(sec, ...) RIB_CAT(RIB_ON_, sec)(RIB_I_V9(RIB_F_TOUCH, __VA_ARGS__))
#define RIB_VU4(sec, ...) RIB_CAT(RIB_ON_, sec)(RIB_I_VU4(RIB_F_TOUCH, __VA_ARGS__))
#define RIB_VS(sec, ...) RIB_CAT(RIB_ON_, sec)(RIB_I_VS(RIB_F_TOUCH, __VA_ARGS__))
#define RIB_VSK(sec, ...) RIB_CAT(RIB_ON_, sec)(RIB_I_VSK(RIB_F_TOUCH, __VA_ARGS__))
#define RIB_VB(sec, ...) RIB_CAT(RIB_ON_, sec)(RIB_I_VB(RIB_F_TOUCH, __VA_ARGS__))
#define RIB_VBX(sec, ...) RIB_CAT(RIB_ON_, sec)(RIB_I_VBX(RIB_F_TOUCH, __VA_ARGS__))
#define RIB_V1D(sec, ...) RIB_CAT(RIB_ON_, sec)(RIB_I_V1D(RIB_F_TOUCH, __VA_ARGS__))
#define RIB_VS1D(sec, ...) RIB_CAT(RIB_ON_, sec)(RIB_I_VS1D(RIB_F_TOUCH, __VA_ARGS__))
#define RIB_VD(sec, ...) RIB_CAT(RIB_ON_, sec)(RIB_I_VD(RIB_F_TOUCH, __VA_ARGS__))
#define RIB_VSD(sec, ...) RIB_CAT(RIB_ON_, sec)(RIB_I_VSD(RIB_F_TOUCH, __VA_ARGS__))
#define RIB_VD9(sec, ...) RIB_CAT(RIB_ON_, sec)(RIB_I_VD9(RIB_F_TOUCH, __VA_ARGS__))

#ifndef RIB_BUILD_STAMP
#error "compile through csrc/build.py (-DRIB_BUILD_STAMP: content hash of this object's sources, see build.py)"
#endif
#define RIB_STR_(x) #x
#define RIB_STR(x) RIB_STR_(x)
// "rib-stamp shard<s> <hash>": what rib_build_info() reports for this object and what build.py looks for in it
extern "C" __attribute__((used, visibility("hidden"))) const char RIB_CAT(rib_stamp_section_, RIB_SECTION)[] =
    "rib-stamp shard" RIB_STR(RIB_SECTION) " " RIB_BUILD_STAMP;

typedef void (*IgemmFn)(const rib::IgemmParams);
extern "C" __attribute__((used, visibility("hidden"))) IgemmFn const RIB_CAT(rib_igemm_section_, RIB_SECTION)[] = {
#include "variants.def"
    nullptr};
