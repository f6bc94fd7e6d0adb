// What bounds k_gemm_dma on the Winograd-domain GEMMs (36 x [256 x 256 x 256], 16 x [256 x 512 x 512]; 41-56 % matrix-pipe busy)?
// A copy of the kernel's loop with parts switched off, timed in one process:
//   0 full | 1 no MFMA (LDS reads folded with adds) | 2 no LDS reads (MFMA on registers) | 3 no fills | 4 MFMA only
//   5 full, no C store | 6 no barrier (wrong results; fills + reads + MFMA free-running)
//   7 full, the next chunk's fills issued BETWEEN the k steps of this chunk's MFMAs instead of in front of them
//   8 full, all of the next chunk's fills issued behind the FIRST k step's MFMAs (an MFMA-first head)
// Build: hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/probes/gemm_limits.hip -o tools/probes/bin/gemm_limits
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct P { const float* A; const float* B; float* C; int M, N, K; size_t sA, sB, sC; };

template <int WM, int WN, int NF, int ABL>
__global__ __launch_bounds__(256) void k(const P p) {
  constexpr int BM = 32 * WM, BN = 32 * NF * WN, BK = 32, STAGE = (BM + BN) * BK, NFILL = (BM + BN) / 32;
  __shared__ __attribute__((aligned(16))) float smem[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), wm = wave / WN, wn = wave % WN;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, z = blockIdx.z;
  const float* A = p.A + (size_t)z * p.sA;
  const float* B = p.B + (size_t)z * p.sB;
  f32x16 acc[NF];
  for (int nf = 0; nf < NF; ++nf) for (int r = 0; r < 16; ++r) acc[nf][r] = 0.f;
  const int nch = p.K / BK;
  typedef __attribute__((address_space(3))) void lds_void;
  const uint32_t lds0 = (uint32_t)(size_t)(lds_void*)smem;
  const float* src[NFILL];
  for (int j = 0; j < NFILL; ++j) {
    const int trow = wave * 8 + 32 * j + (lane >> 3);
    const bool isA = 32 * j < BM;
    const int row = isA ? trow : trow - BM;
    const int ls = (lane & 7) ^ ((row >> 1) & 7);
    src[j] = isA ? A + (size_t)min(m0 + row, p.M - 1) * p.K + ls * 4 : B + (size_t)min(n0 + row, p.N - 1) * p.K + ls * 4;
  }
  auto fill = [&](int st, int kc) {
    if (ABL == 3 || ABL == 4) return;
#pragma unroll
    for (int j = 0; j < NFILL; ++j) {
      const uint32_t dst = lds0 + (uint32_t)(st * STAGE + (wave * 8 + 32 * j) * BK) * 4u;
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src[j] + kc) : "memory");
    }
  };
  fill(0, 0);
  float4 ra = {1.f, 2.f, 3.f, 4.f}, rbv = {.5f, .25f, .125f, 2.f};
  for (int c = 0; c < nch; ++c) {
    const int st = c & 1;
    if (ABL != 4 && ABL != 6) asm volatile("s_waitcnt vmcnt(0)\n\ts_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (ABL == 6) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (ABL != 7 && ABL != 8 && c + 1 < nch) fill(st ^ 1, (c + 1) * BK);
    const float* sA = smem + st * STAGE;
    const float* sB = sA + BM * BK;
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      const int slot = kb * 2 + lh;
      const int rr = wm * 32 + li;
      float4 a = ra;
      float4 b[NF];
      if (ABL != 2 && ABL != 4) a = *reinterpret_cast<const float4*>(sA + rr * BK + (slot ^ ((rr >> 1) & 7)) * 4);
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        const int rb = (wn * NF + nf) * 32 + li;
        b[nf] = rbv;
        if (ABL != 2 && ABL != 4) b[nf] = *reinterpret_cast<const float4*>(sB + rb * BK + (slot ^ ((rb >> 1) & 7)) * 4);
      }
#pragma unroll
      for (int nf = 0; nf < NF; ++nf) {
        if (ABL == 1) { acc[nf][0] += a.x * b[nf].x + a.y * b[nf].y; acc[nf][1] += a.z * b[nf].z + a.w * b[nf].w; }
        else {
          acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[nf].x, acc[nf], 0, 0, 0);
          acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[nf].y, acc[nf], 0, 0, 0);
          acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[nf].z, acc[nf], 0, 0, 0);
          acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[nf].w, acc[nf], 0, 0, 0);
        }
      }
      if (ABL == 2 || ABL == 4) { ra.x += 1.f; rbv.y += 1.f; }
      if (ABL == 8 && kb == 0 && c + 1 < nch) fill(st ^ 1, (c + 1) * BK);
      if (ABL == 7 && c + 1 < nch) {
        constexpr int PER = (NFILL + 3) / 4;      // fill instructions per k step (4 k steps per chunk)
#pragma unroll
        for (int j = kb * PER; j < (kb + 1) * PER && j < NFILL; ++j) {
          const uint32_t dst = lds0 + (uint32_t)((st ^ 1) * STAGE + (wave * 8 + 32 * j) * BK) * 4u;
          asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src[j] + (c + 1) * BK) : "memory");
        }
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
      if (ABL == 5) { if (acc[nf][r] == 123.456f) C[(size_t)row * p.N + col] = acc[nf][r]; }
      else if (row < p.M && col < p.N) C[(size_t)row * p.N + col] = acc[nf][r];
    }
  }
}

template <typename F> float time_us(F launch) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) launch();
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    best = std::min(best, ms * 1000.f / 20);
  }
  return best;
}

template <int WM, int WN, int NF> void run(const char* name, int Z, int M, int N, int K) {
  constexpr int BM = 32 * WM, BN = 32 * NF * WN;
  float *A, *B, *C;
  hipMalloc(&A, (size_t)Z * M * K * 4); hipMalloc(&B, (size_t)Z * N * K * 4); hipMalloc(&C, (size_t)Z * M * N * 4);
  std::vector<float> h((size_t)Z * std::max(M, N) * K);
  for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
  hipMemcpy(A, h.data(), (size_t)Z * M * K * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), (size_t)Z * N * K * 4, hipMemcpyHostToDevice);
  P p{A, B, C, M, N, K, (size_t)M * K, (size_t)N * K, (size_t)M * N};
  dim3 g((M + BM - 1) / BM, (N + BN - 1) / BN, Z);
  const double fl = 2.0 * Z * M * N * K;
  float t[9];
  t[0] = time_us([&] { k<WM, WN, NF, 0><<<g, 256>>>(p); });
  t[1] = time_us([&] { k<WM, WN, NF, 1><<<g, 256>>>(p); });
  t[2] = time_us([&] { k<WM, WN, NF, 2><<<g, 256>>>(p); });
  t[3] = time_us([&] { k<WM, WN, NF, 3><<<g, 256>>>(p); });
  t[4] = time_us([&] { k<WM, WN, NF, 4><<<g, 256>>>(p); });
  t[5] = time_us([&] { k<WM, WN, NF, 5><<<g, 256>>>(p); });
  t[6] = time_us([&] { k<WM, WN, NF, 6><<<g, 256>>>(p); });
  t[7] = time_us([&] { k<WM, WN, NF, 7><<<g, 256>>>(p); });
  t[8] = time_us([&] { k<WM, WN, NF, 8><<<g, 256>>>(p); });
  printf("%-34s tile %3dx%-3d grid %4d  full %6.1f (%3.0f TF) | noMFMA %6.1f | noLDSread %6.1f | noFill %6.1f | MFMAonly %6.1f | noStore %6.1f | noBarrier %6.1f | fills interleaved %6.1f (%3.0f TF) | fills behind k step 0 %6.1f (%3.0f TF)   ideal %5.1f\n",
         name, BM, BN, g.x * g.y * g.z, t[0], fl / t[0] / 1e6, t[1], t[2], t[3], t[4], t[5], t[6], t[7], fl / t[7] / 1e6, t[8], fl / t[8] / 1e6, fl / 157.3e6);
  hipFree(A); hipFree(B); hipFree(C);
}

int main() {
  run<2, 2, 1>("36 x [256 x 256 x 256]", 36, 256, 256, 256);
  run<2, 2, 2>("36 x [256 x 256 x 256]", 36, 256, 256, 256);
  run<4, 1, 2>("36 x [256 x 256 x 256]", 36, 256, 256, 256);
  run<2, 2, 1>("16 x [256 x 512 x 512]", 16, 256, 512, 512);
  run<2, 2, 2>("16 x [256 x 512 x 512]", 16, 256, 512, 512);
  run<2, 2, 4>("16 x [256 x 512 x 512]", 16, 256, 512, 512);
  run<2, 2, 1>("16 x [256 x 256 x 256]", 16, 256, 256, 256);
  run<2, 2, 4>("1 x [4096 x 2048 x 512]", 1, 4096, 2048, 512);
  run<2, 2, 4>("1 x [1024 x 8192 x 512]", 1, 1024, 8192, 512);
  run<2, 2, 2>("1 x [4096 x 2048 x 512]", 1, 4096, 2048, 512);
  run<2, 2, 2>("36 x [256 x 256 x 512]", 36, 256, 256, 512);
  run<2, 2, 1>("36 x [256 x 256 x 512]", 36, 256, 256, 512);
  run<4, 1, 4>("1 x [4096 x 2048 x 512]", 1, 4096, 2048, 512);
  run<4, 1, 4>("1 x [1024 x 8192 x 512]", 1, 1024, 8192, 512);
  run<2, 2, 1>("empty-ish 36 x [256 x 256 x 32]", 36, 256, 256, 32);
  return 0;
}
