// motion.hip — stage 1 on the MI355X (gfx950): the motion transformer of Human_Motion_Modelling (HMM)
// behind the C ABI of include/rib_motion.h (libribmotion.so).  SURVEY 8 row f-4.
//
// The network is small (2 M parameters, clips of 17 .. 321 frames, 128-d tokens): every launch is
// latency-bound, so the design goal is FEW launches with everything elementwise fused into them:
//   km_linear     y = epi( pro(x) . W^T + b ):  prologue = LayerNorm (+ positional encoding for the first
//                 pos_cols output columns: q and k of an attention take x + pos, v takes x), epilogue =
//                 activation, residual add; strided row addressing so that the [N][C][L] clips and the
//                 [L][N][C] outputs of the reference are read / written in place (no permute kernels)
//   km_attention  one thread per query, keys / values staged through LDS in 64-key tiles, exact two-pass
//                 softmax (max, then exp / sum) with the boolean masks of the reference as -inf
//   km_layernorm  in place, for the post-norm variant and the encoder's final norm
//   km_interp     Transformer.interpolate_embedding (bit-exact: same operation order, no contraction)
// fp32 storage and arithmetic throughout (fp32 FMA on the vector ALUs; per the scope contract the
// matrix cores are reserved for the generator's convolutions, and these GEMMs are 10^-3 of its work).
//
// Reference semantics restated here:
//   Transformer.forward / encode / decode      HMM/models/transformer.py:78-133
//   encoder / decoder layers (pre- and post-norm)  HMM/models/transformer.py:202-346
//   nn.MultiheadAttention with separate q/k/v sources: torch.nn.functional.multi_head_attention_forward
//   (third-party, torch 2.10 in this image): q scaled by sqrt(1/head_dim), masks added as -inf, softmax
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/rib_motion.h"

namespace {

constexpr int TM = 8;          // rows per km_linear block
constexpr int KMAX = 1024;     // widest reduction (dim_feedforward)
constexpr float LN_EPS = 1e-5f;

struct Strides { long n, l, k; };   // element (n, l, k) of a logical [N][L][K] tensor

struct LinParams {
  const float* x; Strides xs; int K;
  const float* ln_g; const float* ln_b;       // LayerNorm over K before the product (nullptr: none)
  const float* pos; Strides ps; int pos_cols;  // x += pos for output columns < pos_cols (multiple of TN, or >= Nout)
  const float* wt; int ldw;                    // W^T [K][ldw], columns col0 .. col0+Nout of it
  const float* bias; int Nout;
  int act;                                     // -1 none, else RIBM_ACT_*
  const float* res; Strides rs;                // residual added after the activation (nullptr: none)
  float* y; Strides ys;
  int L, R;                                    // rows = N * L
};

__device__ __forceinline__ float act_fn(float v, int act) {
  if (act == RIBM_ACT_RELU) return v > 0.f ? v : 0.f;
  if (act == RIBM_ACT_LEAKY_RELU) return v > 0.f ? v : 0.01f * v;          // F.leaky_relu default slope
  if (act == RIBM_ACT_GELU) return 0.5f * v * (1.f + erff(v * 0.70710678118654752f));   // F.gelu (erf form)
  return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// grid (ceil(R / TM), ceil(Nout / TN)), block 256.  One global round trip per operand: the TM input rows
// (whole K) and a KC x TN chunk of W^T are staged in LDS by all threads at once, then thread (column =
// tid & 63, row pair = wave) runs its two dot products from LDS (conflict-free filter reads, broadcast
// input reads).  A first version that walked W^T from global memory in the k loop spent 27 us per launch
// waiting on 32 dependent batches of loads (profiles/r01_motion.md).
constexpr int TN = 64, KC = 128;
constexpr int WB = KC * TN / 256;      // filter values per thread and chunk
__global__ __launch_bounds__(256) void km_linear(const LinParams p) {
  __shared__ float xs[TM][KMAX + 4];
  __shared__ float wsm[KC][TN + 1];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int r0 = blockIdx.x * TM;
  const int n0 = blockIdx.y * TN;
  const int K = p.K;
  const int col = n0 + lane;
  const bool cvalid = col < p.Nout;
  // ---- every global load that does not depend on another is issued up front, so that the launch pays ONE
  // memory round trip (a load -> LDS store loop pays one per iteration: 15 us per launch, measured) ----
  float wv[WB];     // first KC x TN chunk of W^T
  const int kn0 = min(KC, K);
#pragma unroll
  for (int u = 0; u < WB; ++u) {
    const int idx = u * 256 + tid;
    const int kk = idx >> 6, cc = idx & 63;
    wv[u] = (kk < kn0 && n0 + cc < p.Nout) ? p.wt[(size_t)kk * p.ldw + n0 + cc] : 0.f;
  }
  const float b = cvalid ? p.bias[col] : 0.f;
  float resv[2] = {0.f, 0.f};
  if (p.res && cvalid) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = r0 + wave * 2 + i;
      if (r < p.R) { const int n = r / p.L, l = r - n * p.L; resv[i] = p.res[n * p.rs.n + l * p.rs.l + col * p.rs.k]; }
    }
  }
  const bool use_pos = p.pos && n0 < p.pos_cols;
  const bool small = K <= 256;       // LayerNorm / positional inputs are hidden_dim wide: at most 8 values per thread
  float xv[8], pv[8], gv[4], bv[4];
  if (small) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = u * 256 + tid;
      const int row = idx / K, k = idx - row * K;
      const int r = r0 + row;
      xv[u] = 0.f; pv[u] = 0.f;
      if (idx < TM * K && r < p.R) {
        const int n = r / p.L, l = r - n * p.L;
        xv[u] = p.x[n * p.xs.n + l * p.xs.l + k * p.xs.k];
        if (use_pos) pv[u] = p.pos[n * p.ps.n + l * p.ps.l + k * p.ps.k];
      }
    }
    if (p.ln_g) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int k = lane + u * 64; gv[u] = k < K ? p.ln_g[k] : 0.f; bv[u] = k < K ? p.ln_b[k] : 0.f; }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = u * 256 + tid;
      if (idx < TM * K) { const int row = idx / K; xs[row][idx - row * K] = xv[u]; }
    }
  } else {
    // wide reductions (the second FFN matrix): no LayerNorm / positional term on these inputs
    for (int base = 0; base < TM * K; base += 256 * 8) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 256 + tid;
        const int row = idx / K, k = idx - row * K;
        const int r = r0 + row;
        xv[u] = 0.f;
        if (idx < TM * K && r < p.R) { const int n = r / p.L, l = r - n * p.L; xv[u] = p.x[n * p.xs.n + l * p.xs.l + k * p.xs.k]; }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int idx = base + u * 256 + tid;
        if (idx < TM * K) { const int row = idx / K; xs[row][idx - row * K] = xv[u]; }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < WB; ++u) {
    const int idx = u * 256 + tid;
    wsm[idx >> 6][idx & 63] = wv[u];
  }
  __syncthreads();
  if (small && p.ln_g) {
    // nn.LayerNorm: biased variance over the last dim, (x - mean) / sqrt(var + eps) * g + b; two passes
    for (int row = wave * 2; row < wave * 2 + 2; ++row) {
      float s = 0.f;
      for (int k = lane; k < K; k += 64) s += xs[row][k];
      const float mean = wave_sum(s) / (float)K;
      float q = 0.f;
      for (int k = lane; k < K; k += 64) { const float d = xs[row][k] - mean; q += d * d; }
      const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)K + LN_EPS);
#pragma unroll
      for (int u = 0; u < 4; ++u) { const int k = lane + u * 64; if (k < K) xs[row][k] = (xs[row][k] - mean) * rstd * gv[u] + bv[u]; }
    }
    __syncthreads();
  }
  if (small && use_pos) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = u * 256 + tid;
      if (idx < TM * K) { const int row = idx / K; xs[row][idx - row * K] += pv[u]; }
    }
    __syncthreads();
  }
  float acc0 = b, acc1 = b;
  const float* x0 = xs[wave * 2];
  const float* x1 = xs[wave * 2 + 1];
  for (int kc = 0; kc < K; kc += KC) {
    const int kn = min(KC, K - kc);
    if (kc > 0) {
      __syncthreads();     // every wave is done with the previous chunk of W^T
#pragma unroll
      for (int u = 0; u < WB; ++u) {
        const int idx = u * 256 + tid;
        const int kk = idx >> 6, cc = idx & 63;
        wv[u] = (kk < kn && n0 + cc < p.Nout) ? p.wt[(size_t)(kc + kk) * p.ldw + n0 + cc] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < WB; ++u) {
        const int idx = u * 256 + tid;
        wsm[idx >> 6][idx & 63] = wv[u];
      }
      __syncthreads();
    }
#pragma unroll 8
    for (int kk = 0; kk < kn; ++kk) {
      const float w = wsm[kk][lane];
      acc0 = fmaf(x0[kc + kk], w, acc0);
      acc1 = fmaf(x1[kc + kk], w, acc1);
    }
  }
  if (!cvalid) return;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = r0 + wave * 2 + i;
    if (r >= p.R) break;
    const int n = r / p.L, l = r - n * p.L;
    float v = i == 0 ? acc0 : acc1;
    if (p.act >= 0) v = act_fn(v, p.act);
    if (p.res) v += resv[i];
    p.y[n * p.ys.n + l * p.ys.l + col * p.ys.k] = v;
  }
}

struct AttnParams {
  const float* q; int ldq;     // rows [N][Lq], this head's slice at column h * HD
  const float* k; int ldk;     // rows [N][Lk]
  const float* v; int ldv;
  const uint8_t* kpm;          // [N][Lk], 1 = key may not be attended (nullptr: none)
  int diag;                    // 1: query i may not attend key i (Transformer.encode's eye mask)
  float* out; int ldo;         // rows [N][Lq]
  int Lq, Lk;
  float scale;                 // sqrt(1 / head_dim), applied to q as torch does
};

// grid (ceil(Lq / 64), heads, N), block 64: thread = query
template <int HD>
__global__ __launch_bounds__(64) void km_attention(const AttnParams p) {
  __shared__ __attribute__((aligned(16))) float sk[64][HD + 1];
  __shared__ __attribute__((aligned(16))) float sv[64][HD + 1];
  __shared__ uint8_t sm[64];
  const int tid = threadIdx.x;
  const int h = blockIdx.y, n = blockIdx.z;
  const int i = blockIdx.x * 64 + tid;
  const bool active = i < p.Lq;
  float q[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) q[d] = active ? p.q[((size_t)n * p.Lq + i) * p.ldq + h * HD + d] * p.scale : 0.f;
  float m = -INFINITY;
  for (int pass = 0; pass < 2; ++pass) {
    float sum = 0.f, acc[HD];
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] = 0.f;
    for (int j0 = 0; j0 < p.Lk; j0 += 64) {
      __syncthreads();
      for (int idx = tid; idx < 64 * HD; idx += 64) {
        const int jj = idx / HD, d = idx - jj * HD;
        const int j = j0 + jj;
        sk[jj][d] = j < p.Lk ? p.k[((size_t)n * p.Lk + j) * p.ldk + h * HD + d] : 0.f;
        if (pass == 1) sv[jj][d] = j < p.Lk ? p.v[((size_t)n * p.Lk + j) * p.ldv + h * HD + d] : 0.f;
      }
      { const int j = j0 + tid; sm[tid] = (j >= p.Lk) || (p.kpm && p.kpm[(size_t)n * p.Lk + j]); }
      __syncthreads();
      const int jn = min(64, p.Lk - j0);
      for (int jj = 0; jj < jn; ++jj) {
        if (sm[jj] || (p.diag && j0 + jj == i)) continue;     // -inf score: no contribution
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < HD; ++d) s = fmaf(q[d], sk[jj][d], s);
        if (pass == 0) m = fmaxf(m, s);
        else {
          const float e = expf(s - m);
          sum += e;
#pragma unroll
          for (int d = 0; d < HD; ++d) acc[d] = fmaf(e, sv[jj][d], acc[d]);
        }
      }
    }
    if (pass == 1 && active) {
      // every key masked: 0 / 0 = NaN, as torch's softmax of a row of -inf
#pragma unroll
      for (int d = 0; d < HD; ++d) p.out[((size_t)n * p.Lq + i) * p.ldo + h * HD + d] = acc[d] / sum;
    }
  }
}

// km_attention_tile: the same attention for clips whose score tile fits LDS (Lk <= ~1000 frames at head_dim 16;
// the reference config trains on 321).  grid (ceil(Lq / QT), heads, N), block 256, QT = 256 / HD queries:
//   1. thread = key: k_j in registers, scores of the QT queries against it -> S[QT][Lk] in LDS (masks as -inf)
//   2. wave = query row: max, exp, sum over the row with wavefront shuffles (exact two-pass softmax)
//   3. thread = (query, channel): out = (sum_j e_j v_j[d]) / sum_j e_j, V staged once in LDS
// The thread-per-query kernel above needs 2 x Lk serial steps per wave (28 us at 65 frames, 135 us at 321);
// here the longest chain is Lk / 4 fused multiply-adds.
template <int HD>
__global__ __launch_bounds__(256) void km_attention_tile(const AttnParams p) {
  constexpr int QT = 256 / HD;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int Lk = p.Lk, SS = Lk + 1;
  float* S = smem;                       // [QT][Lk + 1]
  float* sv = S + QT * SS;               // [Lk][HD]
  float* sq = sv + (size_t)Lk * HD;      // [QT][HD]
  float* ssum = sq + QT * HD;            // [QT]
  const int tid = threadIdx.x;
  const int h = blockIdx.y, n = blockIdx.z;
  const int q0 = blockIdx.x * QT;
  // the first key of this thread is fetched together with q and V: one memory round trip before the scores
  float4 kpre[HD / 4];
  if (tid < Lk) {
    const float4* kp = reinterpret_cast<const float4*>(p.k + ((size_t)n * Lk + tid) * p.ldk + h * HD);
#pragma unroll
    for (int d4 = 0; d4 < HD / 4; ++d4) kpre[d4] = kp[d4];
  }
  const bool mpre = tid < Lk && p.kpm && p.kpm[(size_t)n * Lk + tid];
  {
    const int qi = tid / HD, d = tid - qi * HD;
    const int i = q0 + qi;
    sq[tid] = i < p.Lq ? p.q[((size_t)n * p.Lq + i) * p.ldq + h * HD + d] * p.scale : 0.f;
  }
  for (int base = 0; base < Lk * HD; base += 256 * 8) {   // batched loads, see km_linear
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * 256 + tid;
      const int j = idx / HD, d = idx - j * HD;
      v[u] = idx < Lk * HD ? p.v[((size_t)n * Lk + j) * p.ldv + h * HD + d] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * 256 + tid;
      if (idx < Lk * HD) sv[idx] = v[u];
    }
  }
  __syncthreads();
  for (int j = tid; j < Lk; j += 256) {
    float kreg[HD];
    const float4* kp = reinterpret_cast<const float4*>(p.k + ((size_t)n * Lk + j) * p.ldk + h * HD);
#pragma unroll
    for (int d4 = 0; d4 < HD / 4; ++d4) {
      const float4 t = j == tid ? kpre[d4] : kp[d4];
      kreg[d4 * 4] = t.x; kreg[d4 * 4 + 1] = t.y; kreg[d4 * 4 + 2] = t.z; kreg[d4 * 4 + 3] = t.w;
    }
    const bool masked = j == tid ? mpre : (p.kpm && p.kpm[(size_t)n * Lk + j]);
#pragma unroll 4
    for (int qi = 0; qi < QT; ++qi) {
      float s = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) s = fmaf(sq[qi * HD + d], kreg[d], s);
      if (masked || (p.diag && j == q0 + qi)) s = -INFINITY;
      S[qi * SS + j] = s;
    }
  }
  __syncthreads();
  const int wave = tid >> 6, lane = tid & 63;
  for (int qi = wave; qi < QT; qi += 4) {
    float* row = S + qi * SS;
    float m = -INFINITY;
    for (int j = lane; j < Lk; j += 64) m = fmaxf(m, row[j]);
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sum = 0.f;
    for (int j = lane; j < Lk; j += 64) { const float e = expf(row[j] - m); row[j] = e; sum += e; }   // all keys masked: NaN, as torch
    sum = wave_sum(sum);
    if (lane == 0) ssum[qi] = sum;
  }
  __syncthreads();
  {
    const int qi = tid / HD, d = tid - qi * HD;
    const float* row = S + qi * SS;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int j = 0;
    for (; j + 3 < Lk; j += 4) {
      a0 = fmaf(row[j], sv[(size_t)j * HD + d], a0);
      a1 = fmaf(row[j + 1], sv[(size_t)(j + 1) * HD + d], a1);
      a2 = fmaf(row[j + 2], sv[(size_t)(j + 2) * HD + d], a2);
      a3 = fmaf(row[j + 3], sv[(size_t)(j + 3) * HD + d], a3);
    }
    for (; j < Lk; ++j) a0 = fmaf(row[j], sv[(size_t)j * HD + d], a0);
    const int i = q0 + qi;
    if (i < p.Lq) p.out[((size_t)n * p.Lq + i) * p.ldo + h * HD + d] = ((a0 + a1) + (a2 + a3)) / ssum[qi];
  }
}

template <int HD>
size_t attn_tile_lds_bytes(int Lk) {
  constexpr int QT = 256 / HD;
  return ((size_t)QT * (Lk + 1) + (size_t)Lk * HD + QT * HD + QT) * sizeof(float);
}
constexpr size_t ATTN_TILE_LDS_MAX = 150 * 1024;   // of the 160 KB per CU

// in-place LayerNorm of rows [R][D]; grid ceil(R / 4), block 256 (one wave per row)
__global__ __launch_bounds__(256) void km_layernorm(float* x, int R, int D, const float* g, const float* b) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= R) return;
  float* xr = x + (size_t)row * D;
  float s = 0.f;
  for (int k = lane; k < D; k += 64) s += xr[k];
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int k = lane; k < D; k += 64) { const float d = xr[k] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + LN_EPS);
  for (int k = lane; k < D; k += 64) xr[k] = (xr[k] - mean) * rstd * g[k] + b[k];
}

// Transformer.interpolate_embedding (HMM/models/transformer.py:59-75) on reco [L][N][C] -> center [L][N][C]:
//   (prev / rate * (rate - remain)) + (next / rate * remain), the reference's operation order, unfused
__global__ __launch_bounds__(256) void km_interp(const float* reco, float* center, int L, int NC, int rate) {
#pragma clang fp contract(off)
  const size_t total = (size_t)L * NC;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    const int l = (int)(idx / NC), c = (int)(idx - (size_t)l * NC);
    const int chunk = l / rate, rem = l - chunk * rate;
    const int ln = l == L - 1 ? L - 1 : (chunk + 1) * rate;
    const float prev = reco[(size_t)chunk * rate * NC + c], next = reco[(size_t)ln * NC + c];
    const float a = (prev / (float)rate) * (float)(rate - rem);
    const float b = (next / (float)rate) * (float)rem;
    center[idx] = a + b;
  }
}

std::string fmt(const char* f, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, f);
  vsnprintf(buf, sizeof buf, f, ap);
  va_end(ap);
  return buf;
}

thread_local std::string g_create_error;

struct TensorDef {
  std::string name;
  std::vector<int64_t> dims;
  std::vector<float> data;
  bool set = false;
  size_t off = 0;    // floats into the device blob (matrices stored transposed: [in][out])
};

}  // namespace

struct ribm_handle {
  ribm_config c;
  int device = -1;
  std::string err;
  std::vector<TensorDef> tensors;
  std::map<std::string, int> index;
  float* d_blob = nullptr;
  size_t blob_floats = 0;
  bool ready = false;
  int launches = 0;
};

namespace {

int fail(ribm_handle* h, int code, const std::string& msg) { h->err = msg; return code; }

#define HIPM_TRY(h, expr)                                                                          \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) return fail(h, RIBM_ERR_HIP, fmt("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__)); \
  } while (0)

void add(ribm_handle* h, const std::string& name, std::vector<int64_t> dims) {
  TensorDef t; t.name = name; t.dims = dims;
  h->index[name] = (int)h->tensors.size();
  h->tensors.push_back(t);
}
void add_attn(ribm_handle* h, const std::string& p, int D) {
  add(h, p + ".in_proj_weight", {3 * D, D}); add(h, p + ".in_proj_bias", {3 * D});
  add(h, p + ".out_proj.weight", {D, D});    add(h, p + ".out_proj.bias", {D});
}
void add_ffn(ribm_handle* h, const std::string& p, int D, int F) {
  add(h, p + ".linear1.weight", {F, D}); add(h, p + ".linear1.bias", {F});
  add(h, p + ".linear2.weight", {D, F}); add(h, p + ".linear2.bias", {D});
}
void add_norm(ribm_handle* h, const std::string& p, int D) { add(h, p + ".weight", {D}); add(h, p + ".bias", {D}); }

// the reference's state_dict() order (HMM/models/transformer.py:20-46,184-200,257-276)
void build_inventory(ribm_handle* h) {
  const int D = h->c.hidden_dim, F = h->c.dim_feedforward, C = h->c.input_joints;
  add(h, "input_embed.weight", {D, C}); add(h, "input_embed.bias", {D});
  for (int i = 0; i < h->c.enc_layers; ++i) {
    const std::string p = "encoder.layers." + std::to_string(i);
    add_attn(h, p + ".self_attn", D); add_ffn(h, p, D, F); add_norm(h, p + ".norm1", D); add_norm(h, p + ".norm2", D);
  }
  if (h->c.pre_norm) add_norm(h, "encoder.norm", D);
  for (int i = 0; i < h->c.dec_layers; ++i) {
    const std::string p = "decoder.layers." + std::to_string(i);
    add_attn(h, p + ".self_attn", D); add_attn(h, p + ".multihead_attn", D); add_ffn(h, p, D, F);
    add_norm(h, p + ".norm1", D); add_norm(h, p + ".norm2", D); add_norm(h, p + ".norm3", D);
  }
  add_norm(h, "decoder.norm", D);
  add(h, "joints_embed.weight", {C, D}); add(h, "joints_embed.bias", {C});
  size_t off = 0;
  for (auto& t : h->tensors) {
    size_t n = 1;
    for (auto d : t.dims) n *= (size_t)d;
    t.off = off;
    off += (n + 63) / 64 * 64;
  }
  h->blob_floats = off;
}

// workspace layout (floats), rows R = N * L
struct WsLayout {
  size_t X, T, MEM, QKV, Q, KV, A, HID, RECO, CENTER, total;
};
WsLayout ws_layout(const ribm_config& c, int N, int L) {
  const size_t R = (size_t)N * L, D = c.hidden_dim;
  WsLayout w;
  size_t off = 0;
  auto take = [&](size_t n) { size_t o = off; off += (n + 63) / 64 * 64; return o; };
  w.X = take(R * D); w.T = take(R * D); w.MEM = take(R * D); w.QKV = take(R * 3 * D); w.Q = take(R * D);
  w.KV = take(R * 2 * D); w.A = take(R * D); w.HID = take(R * c.dim_feedforward);
  w.RECO = take(R * c.input_joints); w.CENTER = take(R * c.input_joints);
  w.total = off;
  return w;
}

}  // namespace

extern "C" {

int ribm_create(const ribm_config* cfg, int device, ribm_handle** out) {
  if (!cfg || !out) { g_create_error = "ribm_create: null argument"; return RIBM_ERR_INVALID; }
  const ribm_config& c = *cfg;
  auto bad = [&](const std::string& m) { g_create_error = m; return RIBM_ERR_UNSUPPORTED; };
  if (c.input_joints < 1 || c.input_joints > 128) return bad(fmt("input_joints %d: supported range 1..128", c.input_joints));
  if (c.hidden_dim < 8 || c.hidden_dim > 256 || c.hidden_dim % 4) return bad(fmt("hidden_dim %d: multiple of 4, <= 256", c.hidden_dim));
  if (c.nheads < 1 || c.hidden_dim % c.nheads) return bad("hidden_dim must be divisible by nheads");
  const int hd = c.hidden_dim / c.nheads;
  if (hd != 8 && hd != 16 && hd != 32 && hd != 64) return bad(fmt("head_dim %d: supported 8, 16, 32, 64", hd));
  if (c.dim_feedforward < 1 || c.dim_feedforward > KMAX) return bad(fmt("dim_feedforward %d: <= %d", c.dim_feedforward, KMAX));
  if (c.enc_layers < 1 || c.dec_layers < 1) return bad("enc_layers and dec_layers must be >= 1");
  if (c.activation < RIBM_ACT_RELU || c.activation > RIBM_ACT_LEAKY_RELU) return bad("activation should be relu/gelu/leaky_relu");
  ribm_handle* h = new ribm_handle();
  h->c = c;
  h->device = device;
  build_inventory(h);
  if (device >= 0) {
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) { g_create_error = fmt("hipSetDevice(%d): %s", device, hipGetErrorString(e)); delete h; return RIBM_ERR_HIP; }
  }
  if (device >= 0) {   // score tiles above the default 64 KB of dynamic LDS
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&km_attention_tile<8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATTN_TILE_LDS_MAX);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&km_attention_tile<16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATTN_TILE_LDS_MAX);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&km_attention_tile<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATTN_TILE_LDS_MAX);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&km_attention_tile<64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ATTN_TILE_LDS_MAX);
  }
  *out = h;
  return RIBM_OK;
}

void ribm_destroy(ribm_handle* h) {
  if (!h) return;
  if (h->d_blob) (void)hipFree(h->d_blob);
  delete h;
}

const char* ribm_last_error(const ribm_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

#ifndef RIB_BUILD_STAMP
#error "compile through csrc/build.py (-DRIB_BUILD_STAMP: content hash of the sources, see build.py)"
#endif
const char* ribm_build_info(void) {
  static const char stamp[] = "rib-stamp motion " RIB_BUILD_STAMP;      // (build.py looks for this string in the library)
  return stamp;
}

int ribm_num_tensors(const ribm_handle* h) { return h ? (int)h->tensors.size() : 0; }

int ribm_tensor_info(const ribm_handle* h, int idx, const char** name, int* ndim, int64_t dims[2]) {
  if (!h || idx < 0 || idx >= (int)h->tensors.size()) return RIBM_ERR_INVALID;
  const TensorDef& t = h->tensors[idx];
  if (name) *name = t.name.c_str();
  if (ndim) *ndim = (int)t.dims.size();
  if (dims) for (size_t i = 0; i < t.dims.size(); ++i) dims[i] = t.dims[i];
  return RIBM_OK;
}

int ribm_set_tensor(ribm_handle* h, const char* name, const float* data, int ndim, const int64_t* dims) {
  if (!h || !name || !data || !dims) return h ? fail(h, RIBM_ERR_INVALID, "ribm_set_tensor: null argument") : RIBM_ERR_INVALID;
  auto it = h->index.find(name);
  if (it == h->index.end()) return fail(h, RIBM_ERR_INVALID, fmt("unexpected key '%s' in state_dict (strict load, HMM/utils/utils.py:80)", name));
  TensorDef& t = h->tensors[it->second];
  if (ndim != (int)t.dims.size()) return fail(h, RIBM_ERR_INVALID, fmt("'%s': rank %d, expected %zu", name, ndim, t.dims.size()));
  size_t n = 1;
  for (int i = 0; i < ndim; ++i) {
    if (dims[i] != t.dims[i]) return fail(h, RIBM_ERR_INVALID, fmt("size mismatch for '%s' (dim %d: %lld vs %lld)", name, i, (long long)dims[i], (long long)t.dims[i]));
    n *= (size_t)dims[i];
  }
  t.data.assign(data, data + n);
  t.set = true;
  h->ready = false;
  return RIBM_OK;
}

int ribm_finalize_weights(ribm_handle* h) {
  if (!h) return RIBM_ERR_INVALID;
  for (auto& t : h->tensors)
    if (!t.set) return fail(h, RIBM_ERR_MISSING, fmt("missing key '%s' in state_dict (strict load)", t.name.c_str()));
  std::vector<float> blob(h->blob_floats, 0.f);
  for (auto& t : h->tensors) {
    if (t.dims.size() == 2) {   // nn.Linear weight [out][in] -> [in][out]: consecutive output columns are contiguous
      const int64_t O = t.dims[0], I = t.dims[1];
      for (int64_t o = 0; o < O; ++o)
        for (int64_t i = 0; i < I; ++i) blob[t.off + (size_t)i * O + o] = t.data[(size_t)o * I + i];
    } else {
      std::memcpy(&blob[t.off], t.data.data(), t.data.size() * sizeof(float));
    }
  }
  if (h->device >= 0) {
    HIPM_TRY(h, hipSetDevice(h->device));
    if (!h->d_blob) HIPM_TRY(h, hipMalloc(&h->d_blob, h->blob_floats * sizeof(float)));
    HIPM_TRY(h, hipMemcpy(h->d_blob, blob.data(), h->blob_floats * sizeof(float), hipMemcpyHostToDevice));
  }
  h->ready = true;
  return RIBM_OK;
}

size_t ribm_weights_bytes(const ribm_handle* h) { return h ? h->blob_floats * sizeof(float) : 0; }

size_t ribm_workspace_bytes(const ribm_handle* h, int N, int L) {
  if (!h || N < 1 || L < 1) return 0;
  return ws_layout(h->c, N, L).total * sizeof(float);
}

int ribm_num_launches(const ribm_handle* h) { return h ? h->launches : 0; }

int ribm_forward(ribm_handle* h, int N, int L, int rate, const float* src, const uint8_t* src_mask,
                 const float* src_pos, const float* tgt, const uint8_t* tgt_mask, const float* tgt_pos,
                 float* joints, float* reco, void* ws, size_t ws_bytes, void* stream_) {
  if (!h) return RIBM_ERR_INVALID;
  if (h->device < 0) return fail(h, RIBM_ERR_STATE, "host-only handle (device < 0) cannot launch");
  if (!h->ready) return fail(h, RIBM_ERR_STATE, "ribm_forward before ribm_finalize_weights");
  if (N < 1 || L < 2 || (long)N * L > (1 << 24)) return fail(h, RIBM_ERR_INVALID, fmt("unsupported clip batch N=%d L=%d", N, L));
  if (!src || !src_mask || !src_pos || !tgt_mask || !tgt_pos || !joints || !ws) return fail(h, RIBM_ERR_INVALID, "ribm_forward: null tensor");
  const ribm_config& c = h->c;
  if (c.two_stage) {
    if (rate < 1 || (L - 1) % rate != 0)
      return fail(h, RIBM_ERR_INVALID, fmt("two_stage: (L - 1) %% rate must be 0 (L=%d rate=%d): interpolate_embedding indexes key frames at multiples of rate", L, rate));
  } else if (!tgt) return fail(h, RIBM_ERR_INVALID, "two_stage=0 needs tgt");
  const WsLayout w = ws_layout(c, N, L);
  if (ws_bytes < w.total * sizeof(float)) return fail(h, RIBM_ERR_WORKSPACE, fmt("workspace %zu < %zu bytes", ws_bytes, w.total * sizeof(float)));
  hipStream_t st = (hipStream_t)stream_;
  float* W = (float*)ws;
  const int D = c.hidden_dim, F = c.dim_feedforward, C = c.input_joints, H = c.nheads, HD = D / H;
  const int R = N * L;
  int launches = 0;
  auto T = [&](const std::string& name) -> const float* { return h->d_blob + h->tensors[h->index.at(name)].off; };
  const Strides rowD{(long)L * D, D, 1}, row3D{(long)L * 3 * D, 3 * D, 1}, row2D{(long)L * 2 * D, 2 * D, 1}, rowF{(long)L * F, F, 1};
  const Strides clip{(long)C * L, 1, L};        // [N][C][L]
  const Strides lnc{C, (long)N * C, 1};          // [L][N][C]
  const Strides lnd{D, (long)N * D, 1};          // [L][N][D]

  auto linear = [&](const float* x, Strides xs, int K, const char* ln, const float* pos, int pos_cols, const std::string& wname,
                    int col0, int Nout, const std::string& bname, int act, const float* res, Strides rs, float* y, Strides ys) {
    LinParams p;
    p.x = x; p.xs = xs; p.K = K;
    p.ln_g = ln ? T(std::string(ln) + ".weight") : nullptr; p.ln_b = ln ? T(std::string(ln) + ".bias") : nullptr;
    p.pos = pos; p.ps = lnd; p.pos_cols = pos_cols;
    const TensorDef& wt = h->tensors[h->index.at(wname)];
    p.wt = h->d_blob + wt.off + col0; p.ldw = (int)wt.dims[0];
    p.bias = T(bname) + col0; p.Nout = Nout; p.act = act;
    p.res = res; p.rs = rs; p.y = y; p.ys = ys; p.L = L; p.R = R;
    if (pos && pos_cols < Nout && pos_cols % TN != 0) {
      // a TN-column block must not straddle the pos / no-pos boundary: two launches
      LinParams a = p; a.Nout = pos_cols;
      hipLaunchKernelGGL(km_linear, dim3((R + TM - 1) / TM, (a.Nout + TN - 1) / TN), dim3(256), 0, st, a);
      LinParams b = p; b.pos = nullptr; b.wt += pos_cols; b.bias += pos_cols; b.Nout = Nout - pos_cols; b.y = y + (size_t)pos_cols * ys.k;
      hipLaunchKernelGGL(km_linear, dim3((R + TM - 1) / TM, (b.Nout + TN - 1) / TN), dim3(256), 0, st, b);
      launches += 2;
      return;
    }
    hipLaunchKernelGGL(km_linear, dim3((R + TM - 1) / TM, (Nout + TN - 1) / TN), dim3(256), 0, st, p);
    ++launches;
  };
  auto attention = [&](const float* q, int ldq, const float* k, int ldk, const float* v, int ldv, const uint8_t* kpm, int diag, float* out) {
    AttnParams p;
    p.q = q; p.ldq = ldq; p.k = k; p.ldk = ldk; p.v = v; p.ldv = ldv; p.kpm = kpm; p.diag = diag; p.out = out; p.ldo = D;
    p.Lq = L; p.Lk = L; p.scale = sqrtf(1.0f / (float)HD);
    ++launches;
    const size_t lds = HD == 8 ? attn_tile_lds_bytes<8>(L) : HD == 16 ? attn_tile_lds_bytes<16>(L) : HD == 32 ? attn_tile_lds_bytes<32>(L)
                                                                                                              : attn_tile_lds_bytes<64>(L);
    if (lds <= ATTN_TILE_LDS_MAX && !getenv("RIBM_NO_TILE_ATTENTION")) {
      const int QT = 256 / HD;
      const dim3 grid((L + QT - 1) / QT, H, N);
      if (HD == 8) hipLaunchKernelGGL(km_attention_tile<8>, grid, dim3(256), lds, st, p);
      else if (HD == 16) hipLaunchKernelGGL(km_attention_tile<16>, grid, dim3(256), lds, st, p);
      else if (HD == 32) hipLaunchKernelGGL(km_attention_tile<32>, grid, dim3(256), lds, st, p);
      else hipLaunchKernelGGL(km_attention_tile<64>, grid, dim3(256), lds, st, p);
      return;
    }
    const dim3 grid((L + 63) / 64, H, N);   // long clips: thread-per-query kernel, keys streamed through LDS
    if (HD == 8) hipLaunchKernelGGL(km_attention<8>, grid, dim3(64), 0, st, p);
    else if (HD == 16) hipLaunchKernelGGL(km_attention<16>, grid, dim3(64), 0, st, p);
    else if (HD == 32) hipLaunchKernelGGL(km_attention<32>, grid, dim3(64), 0, st, p);
    else hipLaunchKernelGGL(km_attention<64>, grid, dim3(64), 0, st, p);
  };
  auto layernorm = [&](float* x, const std::string& name) {
    hipLaunchKernelGGL(km_layernorm, dim3((R + 3) / 4), dim3(256), 0, st, x, R, D, T(name + ".weight"), T(name + ".bias"));
    ++launches;
  };
  const Strides none{0, 0, 0};
  auto ffn = [&](float* x, const std::string& p, const char* ln) {
    linear(x, rowD, D, ln, nullptr, 0, p + ".linear1.weight", 0, F, p + ".linear1.bias", c.activation, nullptr, none, W + w.HID, rowF);
    linear(W + w.HID, rowF, F, nullptr, nullptr, 0, p + ".linear2.weight", 0, D, p + ".linear2.bias", -1, x, rowD, x, rowD);
  };

  // ---- encoder: trans_src = input_embed(src); memory = encoder(...) (transformer.py:85-87,113-119) ----
  float* X = W + w.X;
  linear(src, clip, C, nullptr, nullptr, 0, "input_embed.weight", 0, D, "input_embed.bias", -1, nullptr, none, X, rowD);
  for (int i = 0; i < c.enc_layers; ++i) {
    const std::string p = "encoder.layers." + std::to_string(i);
    const std::string n1 = p + ".norm1", n2 = p + ".norm2";
    linear(X, rowD, D, c.pre_norm ? n1.c_str() : nullptr, src_pos, 2 * D, p + ".self_attn.in_proj_weight", 0, 3 * D,
           p + ".self_attn.in_proj_bias", -1, nullptr, none, W + w.QKV, row3D);
    attention(W + w.QKV, 3 * D, W + w.QKV + D, 3 * D, W + w.QKV + 2 * D, 3 * D, src_mask, 1, W + w.A);
    linear(W + w.A, rowD, D, nullptr, nullptr, 0, p + ".self_attn.out_proj.weight", 0, D, p + ".self_attn.out_proj.bias", -1, X, rowD, X, rowD);
    if (!c.pre_norm) layernorm(X, n1);
    ffn(X, p, c.pre_norm ? n2.c_str() : nullptr);
    if (!c.pre_norm) layernorm(X, n2);
  }
  float* MEM = X;
  if (c.pre_norm) {   // encoder_norm exists only with normalize_before (transformer.py:31)
    HIPM_TRY(h, hipMemcpyAsync(W + w.MEM, X, (size_t)R * D * sizeof(float), hipMemcpyDeviceToDevice, st));
    MEM = W + w.MEM;
    layernorm(MEM, "encoder.norm");
  }
  // reco = joints_embed(mem) + input (transformer.py:88); kept as [L][N][C]
  float* RECO = W + w.RECO;
  linear(MEM, rowD, D, nullptr, nullptr, 0, "joints_embed.weight", 0, C, "joints_embed.bias", -1, src, clip, RECO, lnc);
  if (reco) HIPM_TRY(h, hipMemcpyAsync(reco, RECO, (size_t)R * C * sizeof(float), hipMemcpyDeviceToDevice, st));

  // ---- decoder input: two_stage interpolates the reconstruction between key frames (transformer.py:99-105) ----
  const float* center; Strides cs;
  if (c.two_stage) {
    const size_t total = (size_t)R * C;
    hipLaunchKernelGGL(km_interp, dim3((unsigned)std::min<size_t>((total + 255) / 256, 1024)), dim3(256), 0, st, RECO, W + w.CENTER, L, N * C, rate);
    ++launches;
    center = W + w.CENTER; cs = lnc;
  } else { center = tgt; cs = clip; }
  float* Tt = W + w.T;
  linear(center, cs, C, nullptr, nullptr, 0, "input_embed.weight", 0, D, "input_embed.bias", -1, nullptr, none, Tt, rowD);
  for (int i = 0; i < c.dec_layers; ++i) {
    const std::string p = "decoder.layers." + std::to_string(i);
    const std::string n1 = p + ".norm1", n2 = p + ".norm2", n3 = p + ".norm3";
    const std::string sa = p + ".self_attn", ca = p + ".multihead_attn";
    linear(Tt, rowD, D, c.pre_norm ? n1.c_str() : nullptr, tgt_pos, 2 * D, sa + ".in_proj_weight", 0, 3 * D, sa + ".in_proj_bias", -1,
           nullptr, none, W + w.QKV, row3D);
    attention(W + w.QKV, 3 * D, W + w.QKV + D, 3 * D, W + w.QKV + 2 * D, 3 * D, tgt_mask, 0, W + w.A);
    linear(W + w.A, rowD, D, nullptr, nullptr, 0, sa + ".out_proj.weight", 0, D, sa + ".out_proj.bias", -1, Tt, rowD, Tt, rowD);
    if (!c.pre_norm) layernorm(Tt, n1);
    // cross attention: q = (norm2(t) | t) + query_pos, k = memory + pos, v = memory
    linear(Tt, rowD, D, c.pre_norm ? n2.c_str() : nullptr, tgt_pos, D, ca + ".in_proj_weight", 0, D, ca + ".in_proj_bias", -1,
           nullptr, none, W + w.Q, rowD);
    linear(MEM, rowD, D, nullptr, src_pos, D, ca + ".in_proj_weight", D, 2 * D, ca + ".in_proj_bias", -1, nullptr, none, W + w.KV, row2D);
    attention(W + w.Q, D, W + w.KV, 2 * D, W + w.KV + D, 2 * D, src_mask, 0, W + w.A);
    linear(W + w.A, rowD, D, nullptr, nullptr, 0, ca + ".out_proj.weight", 0, D, ca + ".out_proj.bias", -1, Tt, rowD, Tt, rowD);
    if (!c.pre_norm) layernorm(Tt, n2);
    ffn(Tt, p, c.pre_norm ? n3.c_str() : nullptr);
    if (!c.pre_norm) layernorm(Tt, n3);
  }
  // joints = joints_embed(decoder_norm(output)) + center (transformer.py:107-109)
  linear(Tt, rowD, D, "decoder.norm", nullptr, 0, "joints_embed.weight", 0, C, "joints_embed.bias", -1, center, cs, joints, lnc);
  HIPM_TRY(h, hipGetLastError());
  h->launches = launches;
  return RIBM_OK;
}

}  // extern "C"
