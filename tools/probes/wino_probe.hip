// Probe: Winograd F(2x2, 3x3) in fp32 on the matrix cores for the deep 3x3 stride-1 layers that are
// matrix-core-bound today (res / res_flow blocks: 256->256 at 64x64, 512->512 at 32x32), against the
// production k_igemm on the same layer.  2.25x fewer MACs; the question is whether the transforms and
// the larger LDS footprint (1 workgroup per CU) eat the gain.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/wino_probe.hip -o tools/probes/bin/wino_probe
#include "../../render-in-between_amd/csrc/kernels.hip.h"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
using namespace rib;

// Workgroup: 8 x 16 output pixels = 4 x 8 Winograd tiles (M = 32), BN = 32 * NFW output channels, 16 positions
// spread over 4 waves (wave w owns positions 4w .. 4w+3), BK input channels per chunk.
template <int BK, int NFW>
struct WinoGeom {
  static constexpr int CK = BK + 4;
  static constexpr int BN = 32 * NFW;
  static constexpr int SV = 16 * 32 * CK;        // V[pos][tile][CK]
  static constexpr int SU = 16 * BN * CK;        // U[pos][cout][CK]
  static constexpr int SR = 10 * 18 * CK;        // raw halo [10][18][CK]
  static constexpr int SM = 16 * 32 * (BN + 1);  // M[pos][tile][BN+1] for the output transform
  static constexpr int SMEM = (SV + SU + SR) > SM ? (SV + SU + SR) : SM;
};

struct WinoParams {
  const float* x; int H, W, C;        // input NHWC, C = Cin
  const float* u;                     // transformed filters [16][Cout][Cin]
  const float* bias; int Cout;
  float* y;                           // output NHWC
  int tilesX;
};

template <int BK, int NFW>
__global__ __launch_bounds__(256) void k_wino(const WinoParams p) {
  typedef WinoGeom<BK, NFW> G;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* sV = smem;
  float* sU = smem + G::SV;
  float* sR = sU + G::SU;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int tile = blockIdx.x;
  const int ty0 = (tile / p.tilesX) * 8, tx0 = (tile % p.tilesX) * 16;
  const int n0 = blockIdx.y * G::BN;
  constexpr int C4 = BK / 4;
  constexpr int NR4 = (10 * 18 * C4 + 255) / 256;      // raw float4 per thread
  constexpr int NU4 = 16 * G::BN * C4 / 256;           // filter float4 per thread
  float4 rreg[NR4], ureg[NU4];
  f32x16 acc[4][NFW];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int f = 0; f < NFW; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][f][r] = 0.f;

  auto load_chunk = [&](int kc) {
#pragma unroll
    for (int i = 0; i < NR4; ++i) {
      const int idx = tid + i * 256;
      const int pix = idx / C4, c4 = idx % C4;
      const int iy = ty0 - 1 + pix / 18, ix = tx0 - 1 + pix % 18;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < 10 * 18 * C4 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
        v = *reinterpret_cast<const float4*>(p.x + ((size_t)iy * p.W + ix) * p.C + kc + c4 * 4);
      rreg[i] = v;
    }
#pragma unroll
    for (int i = 0; i < NU4; ++i) {
      const int idx = tid + i * 256;
      const int c4 = idx % C4, row = (idx / C4) % G::BN, pos = idx / (C4 * G::BN);
      ureg[i] = *reinterpret_cast<const float4*>(p.u + ((size_t)pos * p.Cout + n0 + row) * p.C + kc + c4 * 4);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int i = 0; i < NR4; ++i) {
      const int idx = tid + i * 256;
      if (idx < 10 * 18 * C4) { const int pix = idx / C4, c4 = idx % C4; *reinterpret_cast<float4*>(sR + pix * G::CK + c4 * 4) = rreg[i]; }
    }
#pragma unroll
    for (int i = 0; i < NU4; ++i) {
      const int idx = tid + i * 256;
      const int c4 = idx % C4, row = (idx / C4) % G::BN, pos = idx / (C4 * G::BN);
      *reinterpret_cast<float4*>(sU + (pos * G::BN + row) * G::CK + c4 * 4) = ureg[i];
    }
  };
  // input transform V = B^T d B of one (tile, float4 channel group, row half): 32 tiles x C4 groups x 2 halves
  auto transform = [&]() {
    for (int item = tid; item < 32 * C4 * 2; item += 256) {
      const int half = item & 1, c4 = (item >> 1) % C4, t = (item >> 1) / C4;
      const int ti = t >> 3, tj = t & 7;
      const float* base = sR + ((2 * ti) * 18 + 2 * tj) * G::CK + c4 * 4;
      float4 d[3][4];
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) d[r][c] = *reinterpret_cast<const float4*>(base + ((r + half) * 18 + c) * G::CK);
      // rows of B^T d: half 0 -> (d0 - d2, d1 + d2); half 1 -> (d2 - d1, d1 - d3) with d indices shifted by `half`
      float4 t0[4], t1[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (half == 0) {
          t0[c] = make_float4(d[0][c].x - d[2][c].x, d[0][c].y - d[2][c].y, d[0][c].z - d[2][c].z, d[0][c].w - d[2][c].w);
          t1[c] = make_float4(d[1][c].x + d[2][c].x, d[1][c].y + d[2][c].y, d[1][c].z + d[2][c].z, d[1][c].w + d[2][c].w);
        } else {   // d[r] holds rows 1, 2, 3
          t0[c] = make_float4(d[1][c].x - d[0][c].x, d[1][c].y - d[0][c].y, d[1][c].z - d[0][c].z, d[1][c].w - d[0][c].w);
          t1[c] = make_float4(d[0][c].x - d[2][c].x, d[0][c].y - d[2][c].y, d[0][c].z - d[2][c].z, d[0][c].w - d[2][c].w);
        }
      }
      auto colt = [&](const float4* tr, int xi) {
        const float4 v0 = make_float4(tr[0].x - tr[2].x, tr[0].y - tr[2].y, tr[0].z - tr[2].z, tr[0].w - tr[2].w);
        const float4 v1 = make_float4(tr[1].x + tr[2].x, tr[1].y + tr[2].y, tr[1].z + tr[2].z, tr[1].w + tr[2].w);
        const float4 v2 = make_float4(tr[2].x - tr[1].x, tr[2].y - tr[1].y, tr[2].z - tr[1].z, tr[2].w - tr[1].w);
        const float4 v3 = make_float4(tr[1].x - tr[3].x, tr[1].y - tr[3].y, tr[1].z - tr[3].z, tr[1].w - tr[3].w);
        *reinterpret_cast<float4*>(sV + ((xi * 4 + 0) * 32 + t) * G::CK + c4 * 4) = v0;
        *reinterpret_cast<float4*>(sV + ((xi * 4 + 1) * 32 + t) * G::CK + c4 * 4) = v1;
        *reinterpret_cast<float4*>(sV + ((xi * 4 + 2) * 32 + t) * G::CK + c4 * 4) = v2;
        *reinterpret_cast<float4*>(sV + ((xi * 4 + 3) * 32 + t) * G::CK + c4 * 4) = v3;
      };
      colt(t0, half * 2 + 0);
      colt(t1, half * 2 + 1);
    }
  };

  load_chunk(0);
  for (int kc = 0; kc < p.C; kc += BK) {
    __syncthreads();            // previous chunk's MFMA reads of sV / sU are done
    store_chunk();
    if (kc + BK < p.C) load_chunk(kc + BK);
    __syncthreads();
    transform();
    __syncthreads();
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int pos = wave * 4 + a;
        const float4 av = *reinterpret_cast<const float4*>(sV + (pos * 32 + li) * G::CK + kb * 8 + lh * 4);
#pragma unroll
        for (int f = 0; f < NFW; ++f) {
          const float4 bv = *reinterpret_cast<const float4*>(sU + (pos * G::BN + f * 32 + li) * G::CK + kb * 8 + lh * 4);
          acc[a][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[a][f], 0, 0, 0);
          acc[a][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[a][f], 0, 0, 0);
          acc[a][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc[a][f], 0, 0, 0);
          acc[a][f] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc[a][f], 0, 0, 0);
        }
      }
    }
  }
  // ---- output transform: M[pos][tile][cout] through LDS, Y = A^T M A ----
  __syncthreads();
  float* sM = smem;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int f = 0; f < NFW; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * lh;      // tile index
        sM[((wave * 4 + a) * 32 + row) * (G::BN + 1) + f * 32 + li] = acc[a][f][r];
      }
  __syncthreads();
  for (int item = tid; item < 32 * G::BN; item += 256) {
    const int c = item % G::BN, t = item / G::BN;
    float m[4][4];
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) m[xi][nu] = sM[((xi * 4 + nu) * 32 + t) * (G::BN + 1) + c];
    float s[2][4];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) { s[0][nu] = m[0][nu] + m[1][nu] + m[2][nu]; s[1][nu] = m[1][nu] - m[2][nu] - m[3][nu]; }
    const float b = p.bias[n0 + c];
    const int ti = t >> 3, tj = t & 7;
#pragma unroll
    for (int oy = 0; oy < 2; ++oy) {
      const float y0 = s[oy][0] + s[oy][1] + s[oy][2] + b, y1 = s[oy][1] - s[oy][2] - s[oy][3] + b;
      const int py = ty0 + 2 * ti + oy, px = tx0 + 2 * tj;
      if (py < p.H && px < p.W) p.y[((size_t)py * p.W + px) * p.Cout + n0 + c] = y0;
      if (py < p.H && px + 1 < p.W) p.y[((size_t)py * p.W + px + 1) * p.Cout + n0 + c] = y1;
    }
  }
}

static void transform_filters(const std::vector<float>& w, int Cout, int Cin, std::vector<float>& u) {
  // w [Cout][9][Cin] -> u [16][Cout][Cin], U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]
  const float Gm[4][3] = {{1, 0, 0}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0, 0, 1}};
  u.assign((size_t)16 * Cout * Cin, 0.f);
  for (int o = 0; o < Cout; ++o)
    for (int i = 0; i < Cin; ++i) {
      float g[3][3], t[4][3];
      for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b) g[a][b] = w[((size_t)o * 9 + a * 3 + b) * Cin + i];
      for (int a = 0; a < 4; ++a) for (int b = 0; b < 3; ++b) t[a][b] = Gm[a][0] * g[0][b] + Gm[a][1] * g[1][b] + Gm[a][2] * g[2][b];
      for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b)
        u[((size_t)(a * 4 + b) * Cout + o) * Cin + i] = t[a][0] * Gm[b][0] + t[a][1] * Gm[b][1] + t[a][2] * Gm[b][2];
    }
}

template <int BK, int NFW>
void run(const char* name, int H, int W, int Cin, int Cout) {
  typedef WinoGeom<BK, NFW> G;
  const size_t nx = (size_t)H * W * Cin, ny = (size_t)H * W * Cout, nw = (size_t)Cout * 9 * Cin;
  std::vector<float> hx(nx), hw(nw), hb(Cout), hu;
  srand(1);
  for (auto& v : hx) v = (rand() % 2001 - 1000) / 1000.f;
  for (auto& v : hw) v = (rand() % 2001 - 1000) / 1000.f / sqrtf(9.f * Cin);
  for (auto& v : hb) v = (rand() % 2001 - 1000) / 5000.f;
  transform_filters(hw, Cout, Cin, hu);
  float *x, *w, *u, *b, *y0, *y1;
  hipMalloc(&x, nx * 4); hipMalloc(&w, nw * 4); hipMalloc(&u, hu.size() * 4); hipMalloc(&b, Cout * 4); hipMalloc(&y0, ny * 4); hipMalloc(&y1, ny * 4);
  hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
  hipMemcpy(u, hu.data(), hu.size() * 4, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), Cout * 4, hipMemcpyHostToDevice);
  // baseline: production direct kernel, 8x16 BN32 BK32 lean, no split-K
  IgemmParams ip{};
  typedef IgemmGeom<16, 4, 1, 1, 1, 32, 1, 3, false> DG;
  ip.x = x; ip.Hin = H; ip.Win = W; ip.xC = Cin; ip.Cin = Cin; ip.w = w; ip.bias = b; ip.CoutPad = Cout; ip.Hout = H; ip.Wout = W;
  ip.tilesX = (W + DG::TW - 1) / DG::TW; ip.tilesY = (H + DG::TH - 1) / DG::TH; ip.y = y0; ip.yC = Cout; ip.Cout = Cout; ip.ksplit = 1;
  dim3 dgrid(ip.tilesX * ip.tilesY, Cout / DG::BN, 1);
  auto dfn = k_igemm<16, 4, 1, 1, 1, 32, 1, 3, false, false, false, false, false>;
  WinoParams wp{x, H, W, Cin, u, b, Cout, y1, (W + 15) / 16};
  dim3 wgrid(((H + 7) / 8) * wp.tilesX, Cout / G::BN, 1);
  auto wfn = k_wino<BK, NFW>;
  const size_t lds = G::SMEM * sizeof(float);
  hipFuncSetAttribute(reinterpret_cast<const void*>(wfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float msd = 0, msw = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(dfn, dgrid, dim3(256), 0, 0, ip);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&msd, e0, e1);
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(wfn, wgrid, dim3(256), lds, 0, wp);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&msw, e0, e1);
  }
  hipError_t err = hipGetLastError();
  std::vector<float> r0(ny), r1(ny);
  hipMemcpy(r0.data(), y0, ny * 4, hipMemcpyDeviceToHost); hipMemcpy(r1.data(), y1, ny * 4, hipMemcpyDeviceToHost);
  double md = 0, mx = 0;
  for (size_t i = 0; i < ny; ++i) { md = fmax(md, fabs((double)r0[i] - r1[i])); mx = fmax(mx, fabs((double)r0[i])); }
  const double flops = 2.0 * Cin * 9 * Cout * (double)H * W;
  printf("%-22s %3dx%-3d %3d->%-3d direct(ksplit 1) %7.1f us %6.1f TF | winograd grid %4d lds %6zu B %7.1f us %6.1f TF (algorithmic) | max|diff| %.2e of %.2f  %s\n",
         name, H, W, Cin, Cout, msd / 20 * 1e3, flops / (msd / 20) / 1e9, wgrid.x * wgrid.y, lds, msw / 20 * 1e3, flops / (msw / 20) / 1e9, md, mx,
         err == hipSuccess ? "" : hipGetErrorString(err));
  hipFree(x); hipFree(w); hipFree(u); hipFree(b); hipFree(y0); hipFree(y1);
}

int main() {
  run<16, 1>("BK16 BN32", 64, 64, 256, 256);
  run<8, 1>("BK8 BN32", 64, 64, 256, 256);
  run<8, 2>("BK8 BN64", 64, 64, 256, 256);
  run<16, 1>("BK16 BN32", 32, 32, 512, 512);
  run<8, 1>("BK8 BN32", 32, 32, 512, 512);
  run<8, 2>("BK8 BN64", 32, 32, 512, 512);
  run<16, 1>("BK16 BN32", 128, 128, 128, 128);
  run<8, 2>("BK8 BN64", 128, 128, 128, 128);
  run<16, 1>("BK16 BN32", 256, 256, 64, 64);
  return 0;
}
