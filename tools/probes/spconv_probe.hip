// spconv_probe.hip - VERDICT r03 item 3: SPADE folded into its convolution's PROLOGUE (SURVEY 2a's original plan).
//
// The library runs a SPADE + LeakyReLU + 3x3 convolution as two launches on the >= 128x128 levels: k_igemm<SPADE> (gamma/beta
// 1x1 GEMM on the condition map, modulate in the epilogue, the modulated tensor WRITTEN to HBM) and k_igemm 3x3 (reads it back).
// Here ONE workgroup does both for an 8x16 output tile:
//   phase A  gamma|beta for the tile's 10x18 HALO pixels: GEMM [192 (180 used) x CC] x [CC x 2C] on the matrix cores, the
//            condition halo tile staged through LDS in 32-channel chunks;
//   modulate (x * rstd - mean * rstd) * (1 + gamma) + beta, LeakyReLU(0.2), zero outside the image (the reference pads the
//            ACTIVATED tensor: PGNR/models/layers/conv.py:77-91 runs norm -> act -> conv), into an LDS tile [180][C];
//   phase B  the 3x3 convolution from that tile (all nine filter slices resident in LDS), bias, NHWC store.
// The modulated tensor and gamma/beta never touch HBM.  Price: the gamma/beta GEMM runs on 192 rows per 128 output pixels (1.5x).
// What this probe leaves out of the library's pair (all of it favours the probe): the fused 1x1 shortcut chunks of conv_block_1,
// the fp64 statistics partials of the output, the consumer-side finalize of x's statistics (rstd / mean come in as arrays).
//
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/probes/spconv_probe.hip -o tools/probes/bin/spconv_probe
//   tools/probes/bin/spconv_probe            # checks against a plain two-pass GPU computation, then times three layer shapes
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct P {
  const float* x; const float* sc; const float* sh;      // [H][W][C], rstd[C], -mean*rstd[C]
  const float* cond;                                     // [H][W][CC]
  const float* wgb; const float* bgb;                    // [2C][CC] (rows: C<=16: [gamma(16)|beta(16)]; else gamma rows 0..C-1, beta rows C..2C-1), [2C]
  const float* w; const float* bias;                     // [COUT][9][C], [COUT]
  float* y;                                              // [H][W][COUT]
  float* ys;                                             // two-pass reference only: the modulated tensor [H][W][C]
  int H, W;
};

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : 0.2f * v; }

// ---- the fused kernel: workgroup = 4 waves, output tile 8 x 16, wave w owns output rows 2w, 2w + 1 (one 32-pixel fragment) ----
template <int C, int CC, int COUT>
__global__ __launch_bounds__(256) void k_spconv(const P p) {
  constexpr int TH = 8, TW = 16, IH = TH + 2, IW = TW + 2, NPX = IH * IW;      // 180 halo pixels
  constexpr int MFR = (NPX + 31) / 32;                                        // 6 row fragments of the gamma/beta GEMM
  constexpr int NG = (2 * C) / 32;                                            // column fragments of gamma|beta
  constexpr int BKC = 32, CKC = BKC + 4;                                      // condition chunk, padded LDS row
  constexpr int CPA = C + 4;                                                  // modulated tile row
  constexpr int TPS = C == 16 ? 9 : (C == 32 ? 3 : 1);                        // filter slices staged per barrier pair in phase B
  constexpr int KW = 9 * C, KS = TPS * C, CPW = KS + 4;                       // conv filter row of one stage
  constexpr int NO = COUT / 32;
  static_assert(C % 16 == 0 && CC % BKC == 0 && COUT % 32 == 0 && NG >= 1, "shapes");
  // LDS: [condition chunk | modulated tile] share one region (the tile is written after the last gamma/beta MFMA), then the
  // gamma/beta filter chunk, then the conv filter stage
  constexpr int SCA = (MFR * 32 * CKC > NPX * CPA) ? MFR * 32 * CKC : NPX * CPA;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sC = smem;
  float* sA = smem;
  float* sG = smem + SCA;
  float* sW = sG + 2 * C * CKC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int tilesX = (p.W + TW - 1) / TW;
  const int ty0 = (blockIdx.x / tilesX) * TH, tx0 = (blockIdx.x % tilesX) * TW;

  // conv filters of stage st (slices st * TPS ..): the first stage is issued here and consumed in phase B
  auto load_w = [&](int st) {
    for (int i = tid; i < COUT * KS / 4; i += 256) {
      const int row = (i * 4) / KS, k = (i * 4) % KS;
      *reinterpret_cast<float4*>(sW + row * CPW + k) = *reinterpret_cast<const float4*>(p.w + (size_t)row * KW + st * KS + k);
    }
  };
  load_w(0);
  // ---- phase A: gamma|beta of the halo pixels ----
  constexpr int MYF = (MFR + 3) / 4;                     // row fragments per wave: f = wave + 4 j
  f32x16 acc[MYF][NG];
#pragma unroll
  for (int j = 0; j < MYF; ++j)
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][g][r] = 0.f;
  for (int kc = 0; kc < CC; kc += BKC) {
    __syncthreads();
    for (int i = tid; i < MFR * 32 * (BKC / 4); i += 256) {
      const int px = i / (BKC / 4), c4 = i % (BKC / 4);
      const int hy = px / IW, hx = px % IW, gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (px < NPX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
        v = *reinterpret_cast<const float4*>(p.cond + ((size_t)gy * p.W + gx) * CC + kc + c4 * 4);
      *reinterpret_cast<float4*>(sC + px * CKC + c4 * 4) = v;
    }
    for (int i = tid; i < 2 * C * (BKC / 4); i += 256) {
      const int row = i / (BKC / 4), c4 = i % (BKC / 4);
      *reinterpret_cast<float4*>(sG + row * CKC + c4 * 4) = *reinterpret_cast<const float4*>(p.wgb + (size_t)row * CC + kc + c4 * 4);
    }
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < BKC / 8; ++kb) {
      float4 b[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) b[g] = *reinterpret_cast<const float4*>(sG + (g * 32 + li) * CKC + kb * 8 + lh * 4);
#pragma unroll
      for (int j = 0; j < MYF; ++j) {
        const int f = wave + 4 * j;
        if (f < MFR) {
          const float4 a = *reinterpret_cast<const float4*>(sC + (f * 32 + li) * CKC + kb * 8 + lh * 4);
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            acc[j][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[g].x, acc[j][g], 0, 0, 0);
            acc[j][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[g].y, acc[j][g], 0, 0, 0);
            acc[j][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[g].z, acc[j][g], 0, 0, 0);
            acc[j][g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[g].w, acc[j][g], 0, 0, 0);
          }
        }
      }
    }
  }
  __syncthreads();        // every wave is done reading the condition chunk: the modulated tile takes its place
  // ---- modulate into the LDS tile: accumulator element r of lane (li, lh) = row (r&3) + 8 (r>>2) + 4 lh, column li ----
#pragma unroll
  for (int j = 0; j < MYF; ++j) {
    const int f = wave + 4 * j;
    if (f >= MFR) continue;
    if constexpr (C == 16) {
      // one fragment [gamma(16) | beta(16)]: the halves exchange so that gamma lanes finish rows 0-7 of the 16, beta lanes rows 8-15
      const int hb = li >> 4, c = li & 15;
      const float bg = p.bgb[c], bb = p.bgb[16 + c], sc = p.sc[c], sh = p.sh[c];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float lo = acc[j][0][k], hi = acc[j][0][8 + k];
        const float olo = __shfl_xor(lo, 16), ohi = __shfl_xor(hi, 16);
        const float gamma = (hb ? ohi : lo) + bg, beta = (hb ? hi : olo) + bb;
        const int r = hb * 8 + k;
        const int px = f * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        const int hy = px / IW, hx = px % IW, gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
        if (px < NPX) {
          float o = 0.f;
          if (gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
            o = lrelu((p.x[((size_t)gy * p.W + gx) * C + c] * sc + sh) * (1.f + gamma) + beta);
          sA[px * CPA + c] = o;
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NG / 2; ++q) {                 // fragment q: gamma of channels 32 q + li, fragment NG/2 + q: their beta
        const int c = q * 32 + li;
        const float bg = p.bgb[c], bb = p.bgb[C + c], sc = p.sc[c], sh = p.sh[c];
        float xr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = f * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int hy = px / IW, hx = px % IW, gy = min(max(ty0 - 1 + hy, 0), p.H - 1), gx = min(max(tx0 - 1 + hx, 0), p.W - 1);
          xr[r] = p.x[((size_t)gy * p.W + gx) * C + c];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = f * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int hy = px / IW, hx = px % IW, gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
          if (px < NPX) {
            const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const float gamma = acc[j][q][r] + bg, beta = acc[j][NG / 2 + q][r] + bb;
            sA[px * CPA + c] = in ? lrelu((xr[r] * sc + sh) * (1.f + gamma) + beta) : 0.f;
          }
        }
      }
    }
  }
  __syncthreads();
  // ---- phase B: the 3x3 convolution out of the LDS tile ----
  f32x16 out[NO];
#pragma unroll
  for (int n = 0; n < NO; ++n)
#pragma unroll
    for (int r = 0; r < 16; ++r) out[n][r] = 0.f;
  const int fy = wave * 2 + li / TW, fx = li % TW;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    if (tap > 0 && tap % TPS == 0) { __syncthreads(); load_w(tap / TPS); __syncthreads(); }
    const float* pa = sA + ((fy + tap / 3) * IW + fx + tap % 3) * CPA + lh * 4;
#pragma unroll
    for (int kb = 0; kb < C / 8; ++kb) {
      const float4 a = *reinterpret_cast<const float4*>(pa + kb * 8);
#pragma unroll
      for (int n = 0; n < NO; ++n) {
        const float4 b = *reinterpret_cast<const float4*>(sW + (n * 32 + li) * CPW + (tap % TPS) * C + kb * 8 + lh * 4);
        out[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, out[n], 0, 0, 0);
        out[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, out[n], 0, 0, 0);
        out[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, out[n], 0, 0, 0);
        out[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, out[n], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int n = 0; n < NO; ++n) {
    const int col = n * 32 + li;
    const float bv = p.bias[col];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const int oy = ty0 + wave * 2 + row / TW, ox = tx0 + row % TW;
      if (oy < p.H && ox < p.W) p.y[((size_t)oy * p.W + ox) * COUT + col] = out[n][r] + bv;
    }
  }
}

// ---- plain two-pass reference on the vector ALUs (correctness only) ----
template <int C, int CC>
__global__ void k_ref_spade(const P p) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t)p.H * p.W * C) return;
  const int c = i % C; const size_t px = i / C;
  float g = p.bgb[C == 16 ? c : c], b = p.bgb[(C == 16 ? 16 : C) + c];
  const float* wg = p.wgb + (size_t)c * CC;
  const float* wb = p.wgb + (size_t)((C == 16 ? 16 : C) + c) * CC;
  for (int k = 0; k < CC; ++k) { const float v = p.cond[px * CC + k]; g += v * wg[k]; b += v * wb[k]; }
  p.ys[i] = lrelu((p.x[i] * p.sc[c] + p.sh[c]) * (1.f + g) + b);
}
template <int C, int COUT>
__global__ void k_ref_conv(const P p) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t)p.H * p.W * COUT) return;
  const int co = i % COUT; const size_t px = i / COUT; const int oy = px / p.W, ox = px % p.W;
  float a = p.bias[co];
  for (int t = 0; t < 9; ++t) {
    const int iy = oy - 1 + t / 3, ix = ox - 1 + t % 3;
    if (iy < 0 || iy >= p.H || ix < 0 || ix >= p.W) continue;
    for (int c = 0; c < C; ++c) a += p.ys[((size_t)iy * p.W + ix) * C + c] * p.w[((size_t)co * 9 + t) * C + c];
  }
  p.y[i] = a;
}

template <typename T> T* dev(size_t n, unsigned seed, float scale) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * (((s >> 8) & 0xffff) / 32768.f - 1.f); }
  T* d; CHECK(hipMalloc(&d, n * sizeof(float))); CHECK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}

template <int C, int CC, int COUT>
size_t lds_bytes() {
  constexpr int NPX = 180, MFR = 6, CKC = 36, CPA = C + 4, TPS = C == 16 ? 9 : (C == 32 ? 3 : 1), CPW = TPS * C + 4;
  constexpr int SCA = (MFR * 32 * CKC > NPX * CPA) ? MFR * 32 * CKC : NPX * CPA;
  return (size_t)(SCA + 2 * C * CKC + COUT * CPW) * sizeof(float);
}

template <int C, int CC, int COUT>
void run(const char* name, int H, int W, double pair_us) {
  const size_t lds = lds_bytes<C, CC, COUT>();
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spconv<C, CC, COUT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  P p; p.H = H; p.W = W;
  p.x = dev<float>((size_t)H * W * C, 1, 2.f); p.sc = dev<float>(C, 2, 1.f); p.sh = dev<float>(C, 3, 0.5f);
  p.cond = dev<float>((size_t)H * W * CC, 4, 1.f);
  p.wgb = dev<float>((size_t)2 * C * CC, 5, 0.1f); p.bgb = dev<float>(2 * C, 6, 0.1f);
  p.w = dev<float>((size_t)COUT * 9 * C, 7, 0.1f); p.bias = dev<float>(COUT, 8, 0.1f);
  float *y1, *y2, *ys;
  CHECK(hipMalloc(&y1, (size_t)H * W * COUT * 4)); CHECK(hipMalloc(&y2, (size_t)H * W * COUT * 4)); CHECK(hipMalloc(&ys, (size_t)H * W * C * 4));
  p.ys = ys;
  const int tiles = ((H + 7) / 8) * ((W + 15) / 16);
  p.y = y1;
  hipLaunchKernelGGL((k_spconv<C, CC, COUT>), dim3(tiles), dim3(256), lds, 0, p);
  p.y = y2;
  hipLaunchKernelGGL((k_ref_spade<C, CC>), dim3(((size_t)H * W * C + 255) / 256), dim3(256), 0, 0, p);
  hipLaunchKernelGGL((k_ref_conv<C, COUT>), dim3(((size_t)H * W * COUT + 255) / 256), dim3(256), 0, 0, p);
  CHECK(hipDeviceSynchronize());
  std::vector<float> a((size_t)H * W * COUT), b(a.size());
  CHECK(hipMemcpy(a.data(), y1, a.size() * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(b.data(), y2, b.size() * 4, hipMemcpyDeviceToHost));
  double md = 0, mv = 0;
  for (size_t i = 0; i < a.size(); ++i) { md = fmax(md, fabs((double)a[i] - b[i])); mv = fmax(mv, fabs((double)b[i])); }
  p.y = y1;
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_spconv<C, CC, COUT>), dim3(tiles), dim3(256), lds, 0, p);
  const int iters = 100;
  CHECK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_spconv<C, CC, COUT>), dim3(tiles), dim3(256), lds, 0, p);
  CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  const double gf = 2.0 * H * W * ((double)CC * 2 * C + 9.0 * C * COUT) / 1e9;
  const double mb = ((double)H * W * (C + CC + COUT) * 4) / 1e6;
  printf("%-34s %4dx%-4d C %3d cond %3d -> %3d : fused %7.1f us  (%.1f alg. TFLOP/s, %.2f TB/s of %.0f MB)   library pair %.1f us   LDS %zu KB   max|diff| %.2e (max|y| %.1f)\n",
         name, H, W, C, CC, COUT, us, gf / us / 1e3, mb / us / 1e6 * 1e3 / 1e3, mb, pair_us, lds / 1024, md, mv);
}

int main() {
  // library pair = SPADE launch + convolution launch of the same layer in the 512x512 B=1 frame (profiles/r03_prof_ops_512.txt)
  run<16, 64, 32>("down_0.1.spade + conv_block_1", 512, 512, 24.0 + 35.8);
  run<16, 64, 32>("same, ragged 200x136", 200, 136, 0);
  run<16, 64, 32>("up_0.1.spade + conv_block_1 (16 -> 16, columns padded to 32)", 512, 512, 23.7 + 25.0);
  run<32, 128, 64>("down_1.1.spade + conv_block_1", 256, 256, 18.3 + 33.4);
  run<32, 128, 32>("up_1.1.spade + conv_block_1", 256, 256, 17.6 + 22.2);
  run<64, 256, 128>("down_2.1.spade + conv_block_1", 128, 128, 19.4 + 35.6);
  run<64, 256, 64>("up_2.1.spade + conv_block_1", 128, 128, 20.7 + 20.9);
  return 0;
}
