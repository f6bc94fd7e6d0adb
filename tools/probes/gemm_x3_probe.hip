// Split-bf16 ("x3") products on the LDS-DMA GEMMs of the frame (premise test).
//
// Every fp32 operand value is written as hi + mid + lo, three bf16 numbers (truncation split: x - hi and r - mid are exact
// fp32 subtractions), and a product a*b becomes NP bf16 matrix-core products accumulated in fp32:
//   NP = 6: hh, hm, mh, hl, lh, mm            (dropped: ml, lm, ll <= 2^-24 |a b| each)
//   NP = 8: + ml, lm                          (dropped: ll <= 2^-32 |a b|)
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32, so NP = 6 is 3/8 of the matrix time, NP = 8 1/2.
// Kernels, all on k_gemm_dma's staging (fp32 tiles by global_load_lds_dwordx4, XOR-swizzled rows, two stages, one barrier
// per 32-channel chunk) so that nothing around the GEMM changes format:
//   F32    production k_gemm_dma (exact fp32 matrix cores)
//   X3R    both operands split IN REGISTERS after the ds_read (11 vector instructions per pair of values and operand)
//   X3R2   the same with the hh products in their own accumulator (the other five never round against the large sum)
//   X3B    B (filters: constants) pre-split into three bf16 planes in memory, A split in registers; chunks of 64 channels
// Accuracy: every variant against an fp64 host GEMM on the same fp32 inputs (max-abs and rms over 64 rows of every batch).
// Shapes: the frame's GEMM-shaped launches (profiles/r05_prof_ops_512.txt).
//
// hipcc -O3 --offload-arch=gfx950 tools/probes/gemm_x3_probe.hip -o tools/probes/bin/gemm_x3_probe
#include "../../render-in-between_amd/csrc/kernels.hip.h"
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
using namespace rib;

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// 8 fp32 values (two 16-byte LDS slots) -> three bf16x8 operands by truncation
__device__ __forceinline__ void split3(const float4 p, const float4 q, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
  const float x[8] = {p.x, p.y, p.z, p.w, q.x, q.y, q.z, q.w};
  u32x4 h, m, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t u0 = __float_as_uint(x[2 * i]), u1 = __float_as_uint(x[2 * i + 1]);
    h[i] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x[2 * i] - __uint_as_float(u0 & 0xffff0000u), r1 = x[2 * i + 1] - __uint_as_float(u1 & 0xffff0000u);
    const uint32_t v0 = __float_as_uint(r0), v1 = __float_as_uint(r1);
    m[i] = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
    const float s0 = r0 - __uint_as_float(v0 & 0xffff0000u), s1 = r1 - __uint_as_float(v1 & 0xffff0000u);
    l[i] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
  }
  hi = __builtin_bit_cast(bf16x8, h); mid = __builtin_bit_cast(bf16x8, m); lo = __builtin_bit_cast(bf16x8, l);
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// MODE 0: X3R, 1: X3R2 (two accumulators); NP products; NST LDS stages (NST - 1 chunks in flight beyond the one being computed)
template <int WM, int WN, int NF, int MODE, int NP, int NST = 2>
__global__ __launch_bounds__(256) void k_gemm_x3r(const GemmDmaParams p) {
  static_assert(WM * WN == 4, "4 waves per workgroup");
  constexpr int BM = 32 * WM, BN = 32 * NF * WN, BK = 32;
  constexpr int STAGE = (BM + BN) * BK;
  constexpr int NFILL = (BM + BN) / 32;
  __shared__ __attribute__((aligned(16))) float smem[NST * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int z = blockIdx.z;
  const float* A = p.A + (size_t)z * p.sA;
  const float* B = p.B + (size_t)(p.modB ? z % p.modB : 0) * p.sB;
  constexpr int NACC = MODE == 1 ? 2 : 1;
  f32x16 acc[NACC][NF];
#pragma unroll
  for (int s = 0; s < NACC; ++s)
#pragma unroll
    for (int nf = 0; nf < NF; ++nf)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][nf][r] = 0.f;
  const int nch = p.K / BK;
  typedef __attribute__((address_space(3))) void lds_void;
  const uint32_t lds0 = (uint32_t)(size_t)(lds_void*)smem;
  const float* src[NFILL];
#pragma unroll
  for (int j = 0; j < NFILL; ++j) {
    const int trow = wave * 8 + 32 * j + (lane >> 3);
    const bool isA = 32 * j < BM;
    const int row = isA ? trow : trow - BM;
    const int ls = (lane & 7) ^ ((row >> 1) & 7);
    src[j] = isA ? A + (size_t)min(m0 + row, p.M - 1) * p.lda + ls * 4 : B + (size_t)min(n0 + row, p.N - 1) * p.K + ls * 4;
  }
  auto fill = [&](int st, int kc) {
#pragma unroll
    for (int j = 0; j < NFILL; ++j) {
      const uint32_t dst = lds0 + (uint32_t)(st * STAGE + (wave * 8 + 32 * j) * BK) * 4u;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src[j] + kc) : "memory");
    }
  };
  constexpr int AHEAD = NST - 1;
  for (int c = 0; c < AHEAD && c < nch; ++c) fill(c, c * BK);
  for (int c = 0; c < nch; ++c) {
    const int st = c % NST;
    // chunk c has landed when at most the fills of the chunks issued after it are outstanding (NFILL per chunk, in order)
    const int later = min(AHEAD - 1, nch - 1 - c);
    if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NFILL) : "memory");
    else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NFILL) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (c + AHEAD < nch) fill((c + AHEAD) % NST, (c + AHEAD) * BK);
    const float* sA = smem + st * STAGE;
    const float* sB = sA + BM * BK;
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      // this lane's 8 k values of the step: slots (4 ks + lh) and (4 ks + 2 + lh) - any 8 will do as long as A and B agree
      const int s0 = ks * 4 + lh, s1 = s0 + 2;
      const int ra = wm * 32 + li;
      bf16x8 ah, am, al;
      split3(*reinterpret_cast<const float4*>(sA + ra * BK + (s0 ^ ((ra >> 1) & 7)) * 4),
             *reinterpret_cast<const float4*>(sA + ra * BK + (s1 ^ ((ra >> 1) & 7)) * 4), ah, am, al);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const int rb = (wn * NF + nf) * 32 + li;
        bf16x8 bh, bm, bl;
        split3(*reinterpret_cast<const float4*>(sB + rb * BK + (s0 ^ ((rb >> 1) & 7)) * 4),
               *reinterpret_cast<const float4*>(sB + rb * BK + (s1 ^ ((rb >> 1) & 7)) * 4), bh, bm, bl);
        f32x16& lo_acc = acc[NACC - 1][nf];
        if constexpr (NP >= 8) { lo_acc = MFMA16(am, bl, lo_acc); lo_acc = MFMA16(al, bm, lo_acc); }
        lo_acc = MFMA16(am, bm, lo_acc);
        lo_acc = MFMA16(ah, bl, lo_acc);
        lo_acc = MFMA16(al, bh, lo_acc);
        lo_acc = MFMA16(ah, bm, lo_acc);
        lo_acc = MFMA16(am, bh, lo_acc);
        acc[0][nf] = MFMA16(ah, bh, acc[0][nf]);
      }
    }
  }
  float* C = p.C + (size_t)z * p.sC;
#pragma unroll
  for (int nf = 0; nf < NF; ++nf) {
    const int col = n0 + (wn * NF + nf) * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      float v = acc[0][nf][r];
      if constexpr (MODE == 1) v += acc[1][nf][r];
      if (row < p.M && col < p.N) C[(size_t)row * p.ldc + col] = v;
    }
  }
}

// X3B: B as three bf16 planes [plane][N][K] (plane stride sBp elements), A fp32 split in registers; chunks of 64 channels:
// A rows are two 128-byte sub-chunks, a B plane row is one (64 bf16).  Stage = A (BM x 64 floats) + 3 planes (BN x 32 words).
struct GemmX3bParams { GemmDmaParams g; const uint16_t* Bp; size_t sBp; };
template <int WM, int WN, int NF, int NP>
__global__ __launch_bounds__(256) void k_gemm_x3b(const GemmX3bParams q) {
  const GemmDmaParams& p = q.g;
  constexpr int BM = 32 * WM, BN = 32 * NF * WN, BK = 32;       // BK: words of a 128-byte row segment
  constexpr int STAGE = (2 * BM + 3 * BN) * BK;
  constexpr int NFILL = (2 * BM + 3 * BN) / 32;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int z = blockIdx.z;
  const float* A = p.A + (size_t)z * p.sA;
  const uint16_t* Bp = q.Bp + (size_t)(p.modB ? z % p.modB : 0) * p.sB;
  f32x16 acc[NF];
#pragma unroll
  for (int nf = 0; nf < NF; ++nf)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[nf][r] = 0.f;
  const int nch = p.K / 64;
  typedef __attribute__((address_space(3))) void lds_void;
  const uint32_t lds0 = (uint32_t)(size_t)(lds_void*)smem;
  // stacked tile rows: [A sub-chunk 0: BM][A sub-chunk 1: BM][plane 0: BN][plane 1: BN][plane 2: BN], 128 bytes each
  const char* src[NFILL]; int step[NFILL];
#pragma unroll
  for (int j = 0; j < NFILL; ++j) {
    const int trow = wave * 8 + 32 * j + (lane >> 3);
    if (32 * j < 2 * BM) {
      const int sub = trow / BM, row = trow % BM;
      const int ls = (lane & 7) ^ ((row >> 1) & 7);
      src[j] = reinterpret_cast<const char*>(A + (size_t)min(m0 + row, p.M - 1) * p.lda + sub * 32 + ls * 4);
    } else {
      const int t = trow - 2 * BM, pl = t / BN, row = t % BN;
      const int ls = (lane & 7) ^ ((row >> 1) & 7);
      src[j] = reinterpret_cast<const char*>(Bp + pl * q.sBp + (size_t)min(n0 + row, p.N - 1) * p.K + ls * 8);
    }
    step[j] = 32 * j < 2 * BM ? 256 : 128;      // bytes per 64-channel chunk
  }
  auto fill = [&](int st, int c) {
#pragma unroll
    for (int j = 0; j < NFILL; ++j) {
      const uint32_t dst = lds0 + (uint32_t)(st * STAGE + (wave * 8 + 32 * j) * BK) * 4u;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src[j] + (size_t)c * step[j]) : "memory");
    }
  };
  fill(0, 0);
  for (int c = 0; c < nch; ++c) {
    const int st = c & 1;
    asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (c + 1 < nch) fill(st ^ 1, c + 1);
    const float* sA = smem + st * STAGE;
    const float* sB = sA + 2 * BM * BK;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      // k = 16 ks + 8 lh .. + 8: A floats [sub = ks / 2] slots 4 (ks & 1) + 2 lh, + 1; B plane slot 2 ks + lh (8 bf16)
      const int ra = wm * 32 + li;
      const float* rowA = sA + ((ks >> 1) * BM + ra) * BK;
      const int sa = (ks & 1) * 4 + 2 * lh;
      bf16x8 ah, am, al;
      split3(*reinterpret_cast<const float4*>(rowA + (sa ^ ((ra >> 1) & 7)) * 4), *reinterpret_cast<const float4*>(rowA + ((sa + 1) ^ ((ra >> 1) & 7)) * 4), ah, am, al);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const int rb = (wn * NF + nf) * 32 + li;
        const int sb = ((2 * ks + lh) ^ ((rb >> 1) & 7)) * 4;
        const bf16x8 bh = *reinterpret_cast<const bf16x8*>(sB + (0 * BN + rb) * BK + sb);
        const bf16x8 bm = *reinterpret_cast<const bf16x8*>(sB + (1 * BN + rb) * BK + sb);
        const bf16x8 bl = *reinterpret_cast<const bf16x8*>(sB + (2 * BN + rb) * BK + sb);
        if constexpr (NP >= 8) { acc[nf] = MFMA16(am, bl, acc[nf]); acc[nf] = MFMA16(al, bm, acc[nf]); }
        acc[nf] = MFMA16(am, bm, acc[nf]);
        acc[nf] = MFMA16(ah, bl, acc[nf]);
        acc[nf] = MFMA16(al, bh, acc[nf]);
        acc[nf] = MFMA16(ah, bm, acc[nf]);
        acc[nf] = MFMA16(am, bh, acc[nf]);
        acc[nf] = MFMA16(ah, bh, acc[nf]);
      }
    }
  }
  float* C = p.C + (size_t)z * p.sC;
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

template <typename F> float time_us(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  hipEventDestroy(e0); hipEventDestroy(e1);
  return best / 20 * 1e3f;
}

struct Err { double mx, rms; };

int main() {
  struct Shape { const char* name; int batch, M, N, K; int wm, wn, nf; float prod_us; };
  // tiles as the r05 plan picks them (profiles/r05_prof_ops_512.txt)
  const Shape shapes[] = {{"res_0/1 wino F(2x2) 16x[256x512x512] 64x128", 16, 256, 512, 512, 2, 2, 2, 22.9f},
                          {"res_0/1 wino F(2x2) 16x[256x512x512] 64x64", 16, 256, 512, 512, 2, 2, 1, 23.0f},
                          {"cond_3 gamma/beta [4096x2048x512] 128x128", 1, 4096, 2048, 512, 4, 1, 4, 72.4f},
                          {"cond_4 gamma/beta [1024x8192x512] 128x128", 1, 1024, 8192, 512, 4, 1, 4, 72.0f},
                          {"res_flow wino4 36x[256x256x256] 64x64", 36, 256, 256, 256, 2, 2, 1, 18.3f},
                          {"res_flow.0 wino4 36x[256x256x512] 64x64", 36, 256, 256, 512, 2, 2, 1, 31.6f},
                          {"up_4 wino 16x[256x256x512] 64x64", 16, 256, 256, 512, 2, 2, 1, 14.9f},
                          {"up_3 wino4 36x[256x128x256] 128x64", 36, 256, 128, 256, 4, 1, 2, 13.3f}};
  size_t maxA = 0, maxB = 0, maxC = 0;
  for (auto& s : shapes) { maxA = std::max(maxA, (size_t)s.batch * s.M * s.K); maxB = std::max(maxB, (size_t)s.batch * s.N * s.K); maxC = std::max(maxC, (size_t)s.batch * s.M * s.N); }
  float *A, *B, *C; uint16_t* Bp;
  hipMalloc(&A, maxA * 4); hipMalloc(&B, maxB * 4); hipMalloc(&C, maxC * 4); hipMalloc(&Bp, maxB * 6);
  std::vector<float> hA(maxA), hB(maxB), hC(maxC);
  std::vector<uint16_t> hBp(maxB * 3);
  srand(1);
  // values of mixed magnitude (a normalised activation times a filter entry: not a grid of short decimals)
  auto rnd = [] { const float u = (rand() % 20001 - 10000) * 1e-4f, v = (rand() % 9973 + 1) / 9973.f; return u * v * 1.7f; };
  for (auto& v : hA) v = rnd();
  for (auto& v : hB) v = rnd() * 0.05f;
  hipMemcpy(A, hA.data(), maxA * 4, hipMemcpyHostToDevice); hipMemcpy(B, hB.data(), maxB * 4, hipMemcpyHostToDevice);
  printf("# split-bf16 products on the LDS-DMA GEMMs: us per launch (best of 5 x 20 back-to-back), error against an fp64 host GEMM (max-abs / rms, 64 rows per batch entry)\n");
  printf("# bar (VERDICT r05 item 1): the first and third shape >= 1.35x faster than F32\n");
  for (auto& s : shapes) {
    GemmDmaParams p{A, B, C, s.M, s.N, s.K, s.K, s.N, (size_t)s.M * s.K, (size_t)s.N * s.K, (size_t)s.M * s.N, s.batch > 1 ? s.batch : 0};
    const int BM = 32 * s.wm, BN = 32 * s.nf * s.wn;
    dim3 grid((s.M + BM - 1) / BM, (s.N + BN - 1) / BN, s.batch);
    // B planes by truncation split on the host (what a fold-time pre-split would store), per batch entry [plane][N][K]
    const size_t nB = (size_t)s.batch * s.N * s.K;
    for (size_t i = 0; i < nB; ++i) {
      const size_t zb = i / ((size_t)s.N * s.K), within = i % ((size_t)s.N * s.K);
      float x = hB[i];
      for (int pl = 0; pl < 3; ++pl) {
        uint32_t u; memcpy(&u, &x, 4); u &= 0xffff0000u; float h; memcpy(&h, &u, 4);
        hBp[(zb * 3 + pl) * (size_t)s.N * s.K + within] = (uint16_t)(u >> 16);
        x -= h;
      }
    }
    hipMemcpy(Bp, hBp.data(), nB * 6, hipMemcpyHostToDevice);
    // fp64 reference on 64 rows of every batch entry
    const int RS = 64;
    std::vector<double> ref((size_t)s.batch * RS * s.N);
    for (int b = 0; b < s.batch; ++b)
      for (int r = 0; r < RS; ++r) {
        const int m = (r * 37 + b * 5) % s.M;
        for (int n = 0; n < s.N; ++n) {
          double acc = 0;
          const float* a = &hA[((size_t)b * s.M + m) * s.K]; const float* w = &hB[((size_t)b * s.N + n) * s.K];
          for (int k = 0; k < s.K; ++k) acc += (double)a[k] * (double)w[k];
          ref[((size_t)b * RS + r) * s.N + n] = acc;
        }
      }
    auto err = [&]() {
      hipMemcpy(hC.data(), C, (size_t)s.batch * s.M * s.N * 4, hipMemcpyDeviceToHost);
      Err e{0, 0}; double ss = 0; size_t cnt = 0;
      for (int b = 0; b < s.batch; ++b)
        for (int r = 0; r < RS; ++r) {
          const int m = (r * 37 + b * 5) % s.M;
          for (int n = 0; n < s.N; ++n) {
            const double d = (double)hC[((size_t)b * s.M + m) * s.N + n] - ref[((size_t)b * RS + r) * s.N + n];
            e.mx = std::max(e.mx, std::abs(d)); ss += d * d; ++cnt;
          }
        }
      e.rms = std::sqrt(ss / cnt);
      return e;
    };
    struct Row { const char* name; float us; Err e; };
    std::vector<Row> rows;
    auto run = [&](const char* name, auto launch) {
      hipMemset(C, 0xff, (size_t)s.batch * s.M * s.N * 4);
      launch(); hipDeviceSynchronize();
      const Err e = err();
      rows.push_back({name, time_us(launch), e});
    };
#define DISPATCH(KERN, ...)                                                                                             \
    do {                                                                                                                \
      if (s.wm == 2 && s.wn == 2 && s.nf == 2) hipLaunchKernelGGL((KERN<2, 2, 2, __VA_ARGS__>), grid, dim3(256), 0, 0, p); \
      else if (s.wm == 2 && s.wn == 2 && s.nf == 1) hipLaunchKernelGGL((KERN<2, 2, 1, __VA_ARGS__>), grid, dim3(256), 0, 0, p); \
      else if (s.wm == 4 && s.wn == 1 && s.nf == 4) hipLaunchKernelGGL((KERN<4, 1, 4, __VA_ARGS__>), grid, dim3(256), 0, 0, p); \
      else hipLaunchKernelGGL((KERN<4, 1, 2, __VA_ARGS__>), grid, dim3(256), 0, 0, p);                                  \
    } while (0)
    run("F32", [&] { DISPATCH(k_gemm_dma, ST_F32); });
    run("X3R np6", [&] { DISPATCH(k_gemm_x3r, 0, 6); });
    run("X3R2 np6", [&] { DISPATCH(k_gemm_x3r, 1, 6); });
    run("X3R np8", [&] { DISPATCH(k_gemm_x3r, 0, 8); });
    run("X3R np6 3st", [&] { DISPATCH(k_gemm_x3r, 0, 6, 3); });
    if (BM + BN <= 192) run("X3R np6 4st", [&] { DISPATCH(k_gemm_x3r, 0, 6, 4); });
    run("X3R2 np6 3st", [&] { DISPATCH(k_gemm_x3r, 1, 6, 3); });
    {
      GemmX3bParams q{p, Bp, (size_t)s.N * s.K};
      q.g.sB = (size_t)3 * s.N * s.K;
      const size_t lds = (size_t)2 * (2 * BM + 3 * BN) * 32 * 4;
      typedef void (*kern_t)(const GemmX3bParams);
      auto pick = [&](int np) -> kern_t {
        if (s.wm == 2 && s.wn == 2 && s.nf == 2) return np == 6 ? k_gemm_x3b<2, 2, 2, 6> : k_gemm_x3b<2, 2, 2, 8>;
        if (s.wm == 2 && s.wn == 2 && s.nf == 1) return np == 6 ? k_gemm_x3b<2, 2, 1, 6> : k_gemm_x3b<2, 2, 1, 8>;
        if (s.wm == 4 && s.wn == 1 && s.nf == 4) return np == 6 ? k_gemm_x3b<4, 1, 4, 6> : k_gemm_x3b<4, 1, 4, 8>;
        return np == 6 ? k_gemm_x3b<4, 1, 2, 6> : k_gemm_x3b<4, 1, 2, 8>;
      };
      const kern_t k6 = pick(6), k8 = pick(8);
      hipFuncSetAttribute((const void*)k6, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipFuncSetAttribute((const void*)k8, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (lds <= 160 * 1024) {
        run("X3B np6", [&] { hipLaunchKernelGGL(k6, grid, dim3(256), lds, 0, q); });
        run("X3B np8", [&] { hipLaunchKernelGGL(k8, grid, dim3(256), lds, 0, q); });
      }
    }
    printf("\n%s   (frame: %.1f us)\n", s.name, s.prod_us);
    for (auto& r : rows)
      printf("  %-12s %7.1f us  x%.2f   max-abs %.3e  rms %.3e  (rms / F32 rms %.2f)\n", r.name, r.us, rows[0].us / r.us, r.e.mx, r.e.rms, r.e.rms / rows[0].e.rms);
  }
  return 0;
}
