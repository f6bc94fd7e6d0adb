// LDS-DMA probe: does staging the operand tiles of the 1x1 / Winograd-domain GEMMs with
// global_load_lds_dwordx4 (no staging registers, no ds_write) beat k_igemm's register-staged double buffering, and does the
// deeper prefetch it makes affordable (a ring of LDS stages instead of registers) pay?
//
//   C[M][N] = sum_k A[M][K] * B[N][K]        (both K-contiguous: activations [pixels][Cin], filters [Cout][Cin]), fp32,
//   v_mfma_f32_32x32x2_f32, workgroup tile 128 x 64 (4 waves x (32 rows x 64 columns)), K chunks of 32.
//
//   REG   k_igemm's scheme: global_load_dwordx4 -> registers while the previous chunk computes, ds_write_b128 into the
//         other buffer after the MFMAs, one barrier per chunk; rows padded to 36 floats (conflict-free ds_read_b128)
//   DMA2  the same two buffers filled by global_load_lds_dwordx4: a wave instruction lands 64 consecutive 16-byte slots, so
//         rows cannot be padded; slot' = slot ^ ((row >> 1) & 7) is conflict-free for ds_read_b128's lane groups instead
//   DMA4  a four-stage ring: the fills of chunks k+1 .. k+3 are in flight while chunk k computes (s_waitcnt vmcnt(12))
// and the production kernel (k_igemm 1x1, 8x16 / BN 64 / BK 32, lean) on the same problem, all in one process.
// Shapes: the condition-level gamma/beta GEMM (4096 x 2048 x 512), the batched Winograd GEMMs of the frame
// (16 x [256 x 512 x 512], 36 x [256 x 256 x 256]) and a long launch (16384 x 2048 x 512).
//
// hipcc -O3 --offload-arch=gfx950 tools/probes/ldsdma_probe.hip -o tools/probes/bin/ldsdma_probe
#include "../../render-in-between_amd/csrc/kernels.hip.h"
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace rib;

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void glb_void;

struct GemmP { const float* A; const float* B; float* C; int M, N, K; size_t sA, sB, sC; };   // batch strides (elements)

template <int MODE>   // 0 REG, 2 DMA2, 4 DMA4
__global__ __launch_bounds__(256) void k_gemm(const GemmP p) {
  constexpr int BM = 128, BN = 64, BK = 32;
  constexpr int PITCH = MODE == 0 ? BK + 4 : BK;
  constexpr int NST = MODE == 4 ? 4 : 2;
  constexpr int STAGE = (BM + BN) * PITCH;
  __shared__ __attribute__((aligned(16))) float smem[NST * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const float* A = p.A + blockIdx.z * p.sA + (size_t)m0 * p.K;
  const float* B = p.B + blockIdx.z * p.sB + (size_t)n0 * p.K;
  f32x16 acc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc[0][r] = 0.f; acc[1][r] = 0.f; }
  const int nch = p.K / BK;

  auto compute = [&](const float* sA, const float* sB) {
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      const int ra = wave * 32 + li;
      const int slot = kb * 2 + lh;
      const int sa = MODE == 0 ? slot : (slot ^ ((ra >> 1) & 7));
      const float4 a = *reinterpret_cast<const float4*>(sA + ra * PITCH + sa * 4);
      float4 b[2];
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        const int rb = nf * 32 + li;
        const int sb = MODE == 0 ? slot : (slot ^ ((rb >> 1) & 7));
        b[nf] = *reinterpret_cast<const float4*>(sB + rb * PITCH + sb * 4);
      }
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) {
        acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[nf].x, acc[nf], 0, 0, 0);
        acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[nf].y, acc[nf], 0, 0, 0);
        acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[nf].z, acc[nf], 0, 0, 0);
        acc[nf] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[nf].w, acc[nf], 0, 0, 0);
      }
    }
  };

  if constexpr (MODE == 0) {
    // thread -> 6 float4 slots of the (128 + 64) x 8 slot tile
    float4 reg[6];
    auto gload = [&](int kc) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int idx = tid + i * 256, row = idx >> 3, c4 = idx & 7;
        reg[i] = row < BM ? *reinterpret_cast<const float4*>(A + (size_t)row * p.K + kc + c4 * 4)
                          : *reinterpret_cast<const float4*>(B + (size_t)(row - BM) * p.K + kc + c4 * 4);
      }
    };
    auto lstore = [&](int st) {
#pragma unroll
      for (int i = 0; i < 6; ++i) {
        const int idx = tid + i * 256, row = idx >> 3, c4 = idx & 7;
        *reinterpret_cast<float4*>(smem + st * STAGE + row * PITCH + c4 * 4) = reg[i];
      }
    };
    gload(0); lstore(0);
    for (int c = 0; c < nch; ++c) {
      const int st = c & 1;
      if (c + 1 < nch) gload((c + 1) * BK);
      __syncthreads();
      compute(smem + st * STAGE, smem + st * STAGE + BM * PITCH);
      if (c + 1 < nch) lstore(st ^ 1);
    }
  } else {
    // wave w fills rows [w*8 + 32 j, +8) of A (j = 0..3) and rows [w*8 + 32 j, +8) of B (j = 0..1): 6 DMA instructions per chunk;
    // lane i lands in slot i of its 8-row group: row i/8, physical slot i%8 = logical slot ^ ((row >> 1) & 7)
    // The DMA is issued through inline assembly: with the builtin the compiler's wait-count pass, which cannot tell the ring's
    // stages apart, puts s_waitcnt vmcnt(0) in front of every ds_read that follows a fill - no prefetch would survive.
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t lds0 = (uint32_t)(size_t)(lds_void*)smem;
    auto fill = [&](int st, int kc) {
      const int r8 = lane >> 3, ps = lane & 7;
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const bool isA = j < 4;
        const int row = wv * 8 + (isA ? j : j - 4) * 32 + r8;
        const int ls = ps ^ ((row >> 1) & 7);
        const float* src = (isA ? A : B) + (size_t)row * p.K + kc + ls * 4;
        const uint32_t dst = lds0 + (uint32_t)(st * STAGE + (isA ? 0 : BM * PITCH) + (wv * 8 + (isA ? j : j - 4) * 32) * PITCH) * 4u;   // wave-uniform byte address
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(dst), "v"(src) : "memory");
      }
    };
    constexpr int AHEAD = NST - 1;     // chunks in flight beyond the one being computed
    for (int c = 0; c < AHEAD && c < nch; ++c) fill(c, c * BK);
    for (int c = 0; c < nch; ++c) {
      const int st = c % NST;
      // chunk c has landed when at most the fills of the chunks issued after it are outstanding (6 per chunk, in order)
      const int later = min(AHEAD - 1, nch - 1 - c);
      if (later >= 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // every wave's part of chunk c is in LDS; everybody finished chunk c-1 (its stage is free)
      if (c + AHEAD < nch) fill((c + AHEAD) % NST, (c + AHEAD) * BK);
      compute(smem + st * STAGE, smem + st * STAGE + BM * PITCH);
    }
  }
  float* C = p.C + blockIdx.z * p.sC;
#pragma unroll
  for (int nf = 0; nf < 2; ++nf)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      C[(size_t)(m0 + wave * 32 + row) * p.N + n0 + nf * 32 + li] = acc[nf][r];
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
  return best / 20 * 1e3f;
}

int main() {
  struct Shape { const char* name; int batch, M, N, K; };
  const Shape shapes[] = {{"cond-level gamma/beta GEMM", 1, 4096, 2048, 512}, {"Winograd F(2x2) x16, 512->512 at 32x32", 16, 256, 512, 512},
                          {"Winograd F(4x4) x36, 256->256 at 64x64", 36, 256, 256, 256}, {"long launch", 1, 16384, 2048, 512}};
  size_t maxA = 0, maxB = 0, maxC = 0;
  for (auto& s : shapes) { maxA = std::max(maxA, (size_t)s.batch * s.M * s.K); maxB = std::max(maxB, (size_t)s.batch * s.N * s.K); maxC = std::max(maxC, (size_t)s.batch * s.M * s.N); }
  float *A, *B, *C, *C2, *bias;
  hipMalloc(&A, maxA * 4); hipMalloc(&B, maxB * 4); hipMalloc(&C, maxC * 4); hipMalloc(&C2, maxC * 4); hipMalloc(&bias, 1 << 16);
  hipMemset(bias, 0, 1 << 16);
  std::vector<float> hA(maxA), hB(maxB);
  srand(1);
  for (auto& v : hA) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hB) v = (rand() % 2001 - 1000) * 1e-3f;
  hipMemcpy(A, hA.data(), maxA * 4, hipMemcpyHostToDevice); hipMemcpy(B, hB.data(), maxB * 4, hipMemcpyHostToDevice);
  printf("%-44s %10s %10s %10s %10s   (us per launch, best of 5 x 20; TFLOP/s)\n", "shape", "k_igemm", "REG", "DMA2", "DMA4");
  for (auto& s : shapes) {
    GemmP p{A, B, C, s.M, s.N, s.K, (size_t)s.M * s.K, (size_t)s.N * s.K, (size_t)s.M * s.N};
    dim3 grid(s.M / 128, s.N / 64, s.batch);
    const double fl = 2.0 * s.batch * s.M * s.N * (double)s.K;
    // production: 1x1 k_igemm, 8x16 tile (128 pixels), BN 64, BK 32, lean; the batch rides in blockIdx.z with one filter set per sample
    IgemmParams ip{};
    const int W = 16, H = s.M / 16;
    ip.x = A; ip.Hin = H; ip.Win = W; ip.xC = s.K; ip.Cin = s.K; ip.w = B; ip.bias = bias; ip.CoutPad = s.N; ip.Hout = H; ip.Wout = W;
    ip.tilesX = 1; ip.tilesY = H / 8; ip.y = C2; ip.yC = s.N; ip.Cout = s.N; ip.ksplit = 1; ip.w_mod = s.batch > 1 ? s.batch : 0; ip.w_stride = (unsigned)((size_t)s.N * s.K);
    auto fn = k_igemm<16, 4, 1, 1, 2, 32, 1, 1, false, false, 0, false, false>;
    dim3 pgrid(ip.tilesX * ip.tilesY, s.N / 64, s.batch);
    const float t_prod = time_us([&] { hipLaunchKernelGGL(fn, pgrid, dim3(256), 0, 0, ip); });
    const float t0 = time_us([&] { hipLaunchKernelGGL(k_gemm<0>, grid, dim3(256), 0, 0, p); });
    hipMemcpy(C2, C, (size_t)s.batch * s.M * s.N * 4, hipMemcpyDeviceToDevice);
    const float t2 = time_us([&] { hipLaunchKernelGGL(k_gemm<2>, grid, dim3(256), 0, 0, p); });
    // correctness of the DMA paths: bit-identical to REG (same MFMA order)
    std::vector<float> h0((size_t)s.batch * s.M * s.N), h1(h0.size());
    hipMemcpy(h0.data(), C2, h0.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(h1.data(), C, h1.size() * 4, hipMemcpyDeviceToHost);
    size_t bad2 = 0; for (size_t i = 0; i < h0.size(); ++i) bad2 += h0[i] != h1[i];
    const float t4 = time_us([&] { hipLaunchKernelGGL(k_gemm<4>, grid, dim3(256), 0, 0, p); });
    hipMemcpy(h1.data(), C, h1.size() * 4, hipMemcpyDeviceToHost);
    size_t bad4 = 0; for (size_t i = 0; i < h0.size(); ++i) bad4 += h0[i] != h1[i];
    // and REG against a host dot product on a few entries
    double worst = 0;
    for (int t = 0; t < 64; ++t) {
      const int b = t % s.batch, m = (t * 37) % s.M, n = (t * 91) % s.N;
      double ref = 0; for (int k = 0; k < s.K; ++k) ref += (double)hA[((size_t)b * s.M + m) * s.K + k] * hB[((size_t)b * s.N + n) * s.K + k];
      worst = std::max(worst, std::abs(ref - h0[((size_t)b * s.M + m) * s.N + n]));
    }
    printf("%-44s %6.1f %4.0f %6.1f %4.0f %6.1f %4.0f %6.1f %4.0f   mismatches DMA2 %zu DMA4 %zu, |REG - host| %.1e\n", s.name, t_prod, fl / t_prod / 1e6,
           t0, fl / t0 / 1e6, t2, fl / t2 / 1e6, t4, fl / t4 / 1e6, bad2, bad4, worst);
  }
  return 0;
}
