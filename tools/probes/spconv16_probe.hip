// spconv16_probe.hip - VERDICT r04 item 2(a): SPADE folded into its convolution's PROLOGUE, retried with 16-bit operands.
//
// Round 4's fp32 probe (spconv_probe.hip) lost at every level because the gamma/beta GEMM on the HALO of a tile (192 rows per
// 128 output pixels, and the condition map is 4x wider than the tensor it modulates) is matrix-core-bound in fp32.  With bf16
// operands (v_mfma_f32_32x32x16_bf16: 16x the rate) that argument is gone and the saved round trip of the modulated tensor
// (+ one launch) could win.  Same structure as the fp32 probe, bf16 storage in HBM and LDS, fp32 accumulate and modulate:
//   phase A  gamma|beta for the tile's 10x18 halo pixels: GEMM [192 x CC] x [CC x 2C], condition tile staged in 64-channel chunks;
//   modulate (x * rstd - mean * rstd) * (1 + gamma) + beta, LeakyReLU(0.2), zero outside the image, rounded to bf16 into an
//            LDS tile [180][C] (the library rounds the modulated tensor to bf16 when it stores it: same values);
//   phase B  the 3x3 convolution from that tile, bias, bf16 NHWC store.
// Left out of the library's pair (all of it favours the probe): the fused 1x1 shortcut chunks of conv_block_1, the fp64
// statistics partials of the output, the consumer-side finalize of x's statistics.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/probes/spconv16_probe.hip -o tools/probes/bin/spconv16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct P {
  const uint16_t* x; const float* sc; const float* sh;   // bf16 [H][W][C], rstd[C], -mean*rstd[C]
  const uint16_t* cond;                                  // bf16 [H][W][CC]
  const uint16_t* wgb; const float* bgb;                 // bf16 [2C][CC] (C == 16: [gamma(16)|beta(16)]; else gamma rows 0..C-1, beta rows C..2C-1), [2C]
  const uint16_t* w; const float* bias;                  // bf16 [COUT][9][C], [COUT]
  uint16_t* y;                                           // bf16 [H][W][COUT]
  // two-pass reference (fp32 arithmetic on the bf16-rounded values, the modulated tensor rounded to bf16 as the library stores it)
  const float* xf; const float* condf; const float* wgbf; const float* wf; float* ysf; float* yf;
  int H, W;
};

__device__ __forceinline__ float lrelu(float v) { return v > 0.f ? v : 0.2f * v; }
__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float((uint32_t)b << 16); }
__device__ __forceinline__ uint16_t f2bf(float v) { const __bf16 b = (__bf16)v; return __builtin_bit_cast(uint16_t, b); }

template <int C, int CC, int COUT>
__global__ __launch_bounds__(256) void k_spconv16(const P p) {
  constexpr int TH = 8, TW = 16, IH = TH + 2, IW = TW + 2, NPX = IH * IW;      // 180 halo pixels
  constexpr int MFR = (NPX + 31) / 32;                                        // 6 row fragments of the gamma/beta GEMM
  constexpr int NG = (2 * C) / 32;                                            // column fragments of gamma|beta
  constexpr int BKC = 64, CKC = BKC + 8;                                      // condition chunk (bf16 elements), padded LDS row: 144 bytes
  constexpr int CPA = C + 8;                                                  // modulated tile row
  constexpr int TPS = C == 16 ? 9 : (C == 32 ? 3 : 1);                        // filter slices staged per barrier pair in phase B
  constexpr int KW = 9 * C, KS = TPS * C, CPW = KS + 8;
  constexpr int NO = COUT / 32;
  static_assert(C % 16 == 0 && CC % BKC == 0 && COUT % 32 == 0 && NG >= 1, "shapes");
  constexpr int SCA = (MFR * 32 * CKC > NPX * CPA) ? MFR * 32 * CKC : NPX * CPA;
  extern __shared__ __attribute__((aligned(16))) uint16_t smem16[];
  uint16_t* sC = smem16;
  uint16_t* sA = smem16;
  uint16_t* sG = smem16 + SCA;
  uint16_t* sW = sG + 2 * C * CKC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  const int tilesX = (p.W + TW - 1) / TW;
  const int ty0 = (blockIdx.x / tilesX) * TH, tx0 = (blockIdx.x % tilesX) * TW;
  auto load_w = [&](int st) {
    for (int i = tid; i < COUT * KS / 8; i += 256) {
      const int row = (i * 8) / KS, k = (i * 8) % KS;
      *reinterpret_cast<uint4*>(sW + row * CPW + k) = *reinterpret_cast<const uint4*>(p.w + (size_t)row * KW + st * KS + k);
    }
  };
  load_w(0);
  // ---- phase A: gamma|beta of the halo pixels ----
  constexpr int MYF = (MFR + 3) / 4;
  f32x16 acc[MYF][NG];
#pragma unroll
  for (int j = 0; j < MYF; ++j)
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][g][r] = 0.f;
  for (int kc = 0; kc < CC; kc += BKC) {
    __syncthreads();
    for (int i = tid; i < MFR * 32 * (BKC / 8); i += 256) {
      const int px = i / (BKC / 8), c8 = i % (BKC / 8);
      const int hy = px / IW, hx = px % IW, gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
      uint4 v = make_uint4(0u, 0u, 0u, 0u);
      if (px < NPX && gy >= 0 && gy < p.H && gx >= 0 && gx < p.W)
        v = *reinterpret_cast<const uint4*>(p.cond + ((size_t)gy * p.W + gx) * CC + kc + c8 * 8);
      *reinterpret_cast<uint4*>(sC + px * CKC + c8 * 8) = v;
    }
    for (int i = tid; i < 2 * C * (BKC / 8); i += 256) {
      const int row = i / (BKC / 8), c8 = i % (BKC / 8);
      *reinterpret_cast<uint4*>(sG + row * CKC + c8 * 8) = *reinterpret_cast<const uint4*>(p.wgb + (size_t)row * CC + kc + c8 * 8);
    }
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < BKC / 16; ++kb) {
      bf16x8 b[NG];
#pragma unroll
      for (int g = 0; g < NG; ++g) b[g] = *reinterpret_cast<const bf16x8*>(sG + (g * 32 + li) * CKC + kb * 16 + lh * 8);
#pragma unroll
      for (int j = 0; j < MYF; ++j) {
        const int f = wave + 4 * j;
        if (f < MFR) {
          const bf16x8 a = *reinterpret_cast<const bf16x8*>(sC + (f * 32 + li) * CKC + kb * 16 + lh * 8);
#pragma unroll
          for (int g = 0; g < NG; ++g) acc[j][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[g], acc[j][g], 0, 0, 0);
        }
      }
    }
  }
  __syncthreads();
  // ---- modulate into the LDS tile: accumulator element r of lane (li, lh) = row (r&3) + 8 (r>>2) + 4 lh, column li ----
#pragma unroll
  for (int j = 0; j < MYF; ++j) {
    const int f = wave + 4 * j;
    if (f >= MFR) continue;
    if constexpr (C == 16) {
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
            o = lrelu((bf2f(p.x[((size_t)gy * p.W + gx) * C + c]) * sc + sh) * (1.f + gamma) + beta);
          sA[px * CPA + c] = f2bf(o);
        }
      }
    } else {
#pragma unroll
      for (int q = 0; q < NG / 2; ++q) {
        const int c = q * 32 + li;
        const float bg = p.bgb[c], bb = p.bgb[C + c], sc = p.sc[c], sh = p.sh[c];
        float xr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = f * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int hy = px / IW, hx = px % IW, gy = min(max(ty0 - 1 + hy, 0), p.H - 1), gx = min(max(tx0 - 1 + hx, 0), p.W - 1);
          xr[r] = bf2f(p.x[((size_t)gy * p.W + gx) * C + c]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = f * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
          const int hy = px / IW, hx = px % IW, gy = ty0 - 1 + hy, gx = tx0 - 1 + hx;
          if (px < NPX) {
            const bool in = gy >= 0 && gy < p.H && gx >= 0 && gx < p.W;
            const float gamma = acc[j][q][r] + bg, beta = acc[j][NG / 2 + q][r] + bb;
            sA[px * CPA + c] = f2bf(in ? lrelu((xr[r] * sc + sh) * (1.f + gamma) + beta) : 0.f);
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
    const uint16_t* pa = sA + ((fy + tap / 3) * IW + fx + tap % 3) * CPA + lh * 8;
#pragma unroll
    for (int kb = 0; kb < C / 16; ++kb) {
      const bf16x8 a = *reinterpret_cast<const bf16x8*>(pa + kb * 16);
#pragma unroll
      for (int n = 0; n < NO; ++n) {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(sW + (n * 32 + li) * CPW + (tap % TPS) * C + kb * 16 + lh * 8);
        out[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, out[n], 0, 0, 0);
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
      if (oy < p.H && ox < p.W) p.y[((size_t)oy * p.W + ox) * COUT + col] = f2bf(out[n][r] + bv);
    }
  }
}

// ---- two-pass reference on the vector ALUs (correctness only): fp32 arithmetic on the bf16-rounded values ----
template <int C, int CC>
__global__ void k_ref_spade(const P p) {
  const size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  if (i >= (size_t)p.H * p.W * C) return;
  const int c = i % C; const size_t px = i / C;
  float g = p.bgb[c], b = p.bgb[(C == 16 ? 16 : C) + c];
  const float* wg = p.wgbf + (size_t)c * CC;
  const float* wb = p.wgbf + (size_t)((C == 16 ? 16 : C) + c) * CC;
  for (int k = 0; k < CC; ++k) { const float v = p.condf[px * CC + k]; g += v * wg[k]; b += v * wb[k]; }
  p.ysf[i] = bf2f(f2bf(lrelu((p.xf[i] * p.sc[c] + p.sh[c]) * (1.f + g) + b)));
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
    for (int c = 0; c < C; ++c) a += p.ysf[((size_t)iy * p.W + ix) * C + c] * p.wf[((size_t)co * 9 + t) * C + c];
  }
  p.yf[i] = a;
}

static uint16_t host_bf16(float f) { uint32_t u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1u); return (uint16_t)(u >> 16); }
static float host_bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

// values in [-scale, scale), rounded to bf16: the bf16 array and its float twin
static void dev_pair(size_t n, unsigned seed, float scale, const uint16_t** d16, const float** d32) {
  std::vector<uint16_t> h16(n); std::vector<float> h32(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h16[i] = host_bf16(scale * (((s >> 8) & 0xffff) / 32768.f - 1.f)); h32[i] = host_bf2f(h16[i]); }
  uint16_t* a; float* b;
  CHECK(hipMalloc(&a, n * 2)); CHECK(hipMemcpy(a, h16.data(), n * 2, hipMemcpyHostToDevice));
  CHECK(hipMalloc(&b, n * 4)); CHECK(hipMemcpy(b, h32.data(), n * 4, hipMemcpyHostToDevice));
  *d16 = a; *d32 = b;
}
static const float* dev_f32(size_t n, unsigned seed, float scale) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = scale * (((s >> 8) & 0xffff) / 32768.f - 1.f); }
  float* d; CHECK(hipMalloc(&d, n * 4)); CHECK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}

template <int C, int CC, int COUT>
size_t lds_bytes() {
  constexpr int NPX = 180, MFR = 6, CKC = 72, CPA = C + 8, TPS = C == 16 ? 9 : (C == 32 ? 3 : 1), CPW = TPS * C + 8;
  constexpr int SCA = (MFR * 32 * CKC > NPX * CPA) ? MFR * 32 * CKC : NPX * CPA;
  return (size_t)(SCA + 2 * C * CKC + COUT * CPW) * sizeof(uint16_t);
}

template <int C, int CC, int COUT>
void run(const char* name, int H, int W, double spade_us, double conv_us) {
  const size_t lds = lds_bytes<C, CC, COUT>();
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_spconv16<C, CC, COUT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  P p; p.H = H; p.W = W;
  dev_pair((size_t)H * W * C, 1, 2.f, &p.x, &p.xf); p.sc = dev_f32(C, 2, 1.f); p.sh = dev_f32(C, 3, 0.5f);
  dev_pair((size_t)H * W * CC, 4, 1.f, &p.cond, &p.condf);
  dev_pair((size_t)2 * C * CC, 5, 0.1f, &p.wgb, &p.wgbf); p.bgb = dev_f32(2 * C, 6, 0.1f);
  dev_pair((size_t)COUT * 9 * C, 7, 0.1f, &p.w, &p.wf); p.bias = dev_f32(COUT, 8, 0.1f);
  uint16_t* y16; float *yf, *ysf;
  CHECK(hipMalloc(&y16, (size_t)H * W * COUT * 2)); CHECK(hipMalloc(&yf, (size_t)H * W * COUT * 4)); CHECK(hipMalloc(&ysf, (size_t)H * W * C * 4));
  p.y = y16; p.yf = yf; p.ysf = ysf;
  const int tiles = ((H + 7) / 8) * ((W + 15) / 16);
  hipLaunchKernelGGL((k_spconv16<C, CC, COUT>), dim3(tiles), dim3(256), lds, 0, p);
  hipLaunchKernelGGL((k_ref_spade<C, CC>), dim3(((size_t)H * W * C + 255) / 256), dim3(256), 0, 0, p);
  hipLaunchKernelGGL((k_ref_conv<C, COUT>), dim3(((size_t)H * W * COUT + 255) / 256), dim3(256), 0, 0, p);
  CHECK(hipDeviceSynchronize());
  std::vector<uint16_t> a((size_t)H * W * COUT); std::vector<float> b(a.size());
  CHECK(hipMemcpy(a.data(), y16, a.size() * 2, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(b.data(), yf, b.size() * 4, hipMemcpyDeviceToHost));
  double md = 0, mv = 0, mrel = 0;
  for (size_t i = 0; i < a.size(); ++i) {
    const double d = fabs((double)host_bf2f(a[i]) - b[i]);
    md = fmax(md, d); mv = fmax(mv, fabs((double)b[i])); mrel = fmax(mrel, d / fmax(1.0, fabs((double)b[i])));
  }
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k_spconv16<C, CC, COUT>), dim3(tiles), dim3(256), lds, 0, p);
  const int iters = 100;
  CHECK(hipEventRecord(e0, 0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((k_spconv16<C, CC, COUT>), dim3(tiles), dim3(256), lds, 0, p);
  CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / iters;
  const double mb = ((double)H * W * (C + CC + COUT) * 2) / 1e6;
  printf("%-62s %4dx%-4d C %3d cond %3d -> %3d : fused %6.1f us (%.2f TB/s of %.0f MB)   library pair (bf16 frame) %.1f + %.1f = %.1f us   LDS %zu KB   max|diff| %.2e (bf16 output; max|y| %.1f, rel %.1e)\n",
         name, H, W, C, CC, COUT, us, mb / us / 1e6 * 1e3 / 1e3, mb, spade_us, conv_us, spade_us + conv_us, lds / 1024, md, mv, mrel);
}

int main() {
  // library pair = SPADE launch + convolution launch of the same layer in the bf16 512x512 B=1 frame (profiles/r03_prof_ops_512_bf16.txt)
  run<16, 64, 32>("down_0.1.spade + conv_block_1", 512, 512, 14.8, 19.8);
  run<16, 64, 32>("same, ragged 200x136", 200, 136, 0, 0);
  run<16, 64, 32>("up_0.1.spade + conv_block_1 (16 -> 16, columns padded to 32)", 512, 512, 13.9, 20.7);
  run<32, 128, 64>("down_1.1.spade + conv_block_1", 256, 256, 7.2, 13.2);
  run<32, 128, 32>("up_1.1.spade + conv_block_1", 256, 256, 9.7, 11.1);
  run<64, 256, 128>("down_2.1.spade + conv_block_1", 128, 128, 10.6, 11.1);
  run<64, 256, 64>("up_2.1.spade + conv_block_1", 128, 128, 11.3, 10.1);
  return 0;
}
