// Probe (round 2): what can a dedicated fp32 GEMM kernel reach on the 1x1 shapes of the frame (SPADE gamma/beta GEMMs,
// Winograd-domain GEMMs), where k_igemm's 1x1 variants run at 75-88 TFLOP/s against 100 for the 3x3 layers?
//   C[M][N] = A[M][K] . B[N][K]^T   (A = NHWC pixels x channels, B = filters, both K-contiguous), batch of G problems
// Candidate: BM x BN workgroup tile, 4 waves as 2 x 2, wave tile (BM/2) x (BN/2) of 32x32 fragments, BK-float stages,
// LDS double-buffered with register prefetch of the next stage (one barrier per stage), K-permuted 16-byte LDS reads.
// hipcc -O3 --offload-arch=gfx950 tools/probes/gemm_probe.hip -o tools/probes/bin/gemm_probe && tools/probes/bin/gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int BM, int BN, int BK, int MINW>
__global__ __launch_bounds__(256, MINW) void k_gemm(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                    int M, int N, int K, size_t strideA, size_t strideB, size_t strideC) {
  constexpr int P = BK + 4;                      // LDS row pitch (floats)
  constexpr int MF = BM / 64, NF = BN / 64;      // 32x32 fragments per wave
  constexpr int LA = BM * BK / 4 / 256, LB = BN * BK / 4 / 256;   // float4 loads per thread per stage
  static_assert(LA >= 1 && LB >= 1, "tile too small");
  __shared__ __attribute__((aligned(16))) float sA[2][BM * P];
  __shared__ __attribute__((aligned(16))) float sB[2][BN * P];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // XCD-aware tile order: consecutive workgroup ids go to different XCDs, so give each XCD a contiguous range of tiles
  const int ntm = (M + BM - 1) / BM, ntn = N / BN;
  int t = blockIdx.x;
  const int total = ntm * ntn;
  if (total % 8 == 0) t = (t & 7) * (total / 8) + (t >> 3);
  const int tn = t % ntn, tm = t / ntn;
  A += blockIdx.y * strideA; B += blockIdx.y * strideB; C += blockIdx.y * strideC;
  const int m0 = tm * BM, n0 = tn * BN;
  constexpr int TPR = BK / 4;                    // threads per row (float4 each)
  const int lrow = tid / TPR, lcol = (tid % TPR) * 4;
  constexpr int RPP = 256 / TPR;                 // rows per pass
  float4 ra[LA], rb[LB];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int r = min(m0 + lrow + i * RPP, M - 1);
      ra[i] = *reinterpret_cast<const float4*>(A + (size_t)r * K + k0 + lcol);
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) rb[i] = *reinterpret_cast<const float4*>(B + (size_t)(n0 + lrow + i * RPP) * K + k0 + lcol);
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < LA; ++i) *reinterpret_cast<float4*>(&sA[buf][(lrow + i * RPP) * P + lcol]) = ra[i];
#pragma unroll
    for (int i = 0; i < LB; ++i) *reinterpret_cast<float4*>(&sB[buf][(lrow + i * RPP) * P + lcol]) = rb[i];
  };
  f32x16 acc[MF][NF];
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  gload(0);
  lstore(0);
  __syncthreads();
  const int nst = K / BK;
  for (int s = 0; s < nst; ++s) {
    const int buf = s & 1;
    if (s + 1 < nst) gload((s + 1) * BK);
    const float* pa = &sA[buf][(wm * (BM / 2) + li) * P + lh * 4];
    const float* pb = &sB[buf][(wn * (BN / 2) + li) * P + lh * 4];
#pragma unroll
    for (int kk = 0; kk < BK / 8; ++kk) {
      float4 fa[MF], fb[NF];
#pragma unroll
      for (int i = 0; i < MF; ++i) fa[i] = *reinterpret_cast<const float4*>(pa + i * 32 * P + kk * 8);
#pragma unroll
      for (int j = 0; j < NF; ++j) fb[j] = *reinterpret_cast<const float4*>(pb + j * 32 * P + kk * 8);
#pragma unroll
      for (int i = 0; i < MF; ++i)
#pragma unroll
        for (int j = 0; j < NF; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].x, fb[j].x, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].y, fb[j].y, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].z, fb[j].z, acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i].w, fb[j].w, acc[i][j], 0, 0, 0);
        }
    }
    if (s + 1 < nst) lstore(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int i = 0; i < MF; ++i)
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int col = n0 + wn * (BN / 2) + j * 32 + li;
        if (row < M) C[(size_t)row * N + col] = acc[i][j][r];
      }
}

template <int BM, int BN, int BK, int MINW>
void run(const char* name, int M, int N, int K, int G, const float* A, const float* B, float* C, std::vector<float>* check) {
  if (N % BN != 0) return;
  dim3 grid(((M + BM - 1) / BM) * (N / BN), G, 1);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  const int it = 20;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < it; ++i)
      hipLaunchKernelGGL((k_gemm<BM, BN, BK, MINW>), grid, dim3(256), 0, 0, A, B, C, M, N, K, (size_t)M * K, (size_t)N * K, (size_t)M * N);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  ms /= it;
  double err = -1;
  if (check) {   // first problem, a few entries
    std::vector<float> hc((size_t)M * N);
    hipMemcpy(hc.data(), C, hc.size() * 4, hipMemcpyDeviceToHost);
    const std::vector<float>& h = *check;
    err = 0;
    for (int q = 0; q < 64; ++q) {
      const int r = (q * 7919) % M, c = (q * 104729) % N;
      double ref = 0;
      for (int k = 0; k < K; ++k) ref += (double)h[(size_t)r * K + k] * (double)h[(size_t)c * K + k + 5000000];
      err = fmax(err, fabs(ref - hc[(size_t)r * N + c]));
    }
  }
  const double flops = 2.0 * M * N * (double)K * G;
  printf("%-22s M %6d N %4d K %3d G %2d grid %5d: %7.1f us %6.1f TFLOP/s  err %.1e\n", name, M, N, K, G, grid.x * grid.y, ms * 1e3, flops / ms / 1e9, err);
}

int main() {
  const size_t n = (size_t)64 << 20;           // floats
  float *A, *C;
  hipMalloc(&A, n * 4); hipMalloc(&C, n * 4);
  std::vector<float> h(n / 4);
  for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2001) / 1000.f - 1.f;
  for (int q = 0; q < 4; ++q) hipMemcpy(A + q * h.size(), h.data(), h.size() * 4, hipMemcpyHostToDevice);
  const float* B = A + 5000000;
  struct S { const char* what; int M, N, K, G; };
  const S shapes[] = {
      {"up_3.0 spade", 4096, 1024, 512, 1},   {"up_2.0 spade", 16384, 512, 256, 1}, {"up_1.0 spade", 65536, 256, 128, 1},
      {"up_0.0 spade", 262144, 128, 64, 1},   {"down_0.x spade", 262144, 64, 64, 1}, {"res spade", 1024, 1024, 512, 1},
      {"res_flow wino4", 256, 256, 256, 36},  {"res_flow wino2", 1024, 256, 256, 16}, {"res wino2", 256, 512, 512, 16},
      {"res wino4", 64, 512, 512, 36},
  };
  for (const S& s : shapes) {
    printf("-- %s\n", s.what);
    std::vector<float>* chk = s.G == 1 && s.M <= 16384 ? &h : nullptr;
    run<128, 128, 32, 1>("128x128 BK32", s.M, s.N, s.K, s.G, A, B, C, chk);
    run<128, 128, 16, 2>("128x128 BK16", s.M, s.N, s.K, s.G, A, B, C, nullptr);
    run<128, 64, 32, 2>("128x64 BK32", s.M, s.N, s.K, s.G, A, B, C, nullptr);
    run<64, 128, 32, 2>("64x128 BK32", s.M, s.N, s.K, s.G, A, B, C, nullptr);
    run<64, 64, 32, 2>("64x64 BK32", s.M, s.N, s.K, s.G, A, B, C, nullptr);
    run<64, 64, 64, 2>("64x64 BK64", s.M, s.N, s.K, s.G, A, B, C, nullptr);
    run<128, 64, 64, 1>("128x64 BK64", s.M, s.N, s.K, s.G, A, B, C, nullptr);
  }
  return 0;
}
