// raster.hip.h — label-map rasterisation on the GPU.
//
// The 22-channel label of a frame is  [3-channel limb drawing in [-1,1] | 19 joint heat-maps in [0,1]]
// (PGNR/models/evaluator.py:221-229,250).  Both are integer/IEEE-exact restatements:
//
//   k_heatmaps   _generate_pose_map, test phase (PGNR/datasets/HSM_auto_dataset.py:205-236):
//                one-hot at (int(y), int(x)) -> scipy.ndimage.gaussian_filter(sigma) -> / max.
//                scipy filters axis 0 then axis 1 in fp64 with the symmetric branch of its 1-D
//                correlation (centre tap first, then tap pairs from the outermost inwards) and
//                'reflect' borders; a one-hot input makes every line hold a single non-zero, so a
//                pixel is two short tap walks.  Same order, same unfused fp64 ops -> bit-exact fp32.
//   k_skeleton   _generate_skeleton (HSM_auto_dataset.py:238-251) = connect_keypoints / drawEdge /
//                setColor (PGNR/utils/keypoint2img.py:36-64,132-147): a painter's algorithm whose
//                result depends on the order of its steps ("if every touched pixel is still black
//                paint, else average with what is there"), so the steps of one frame stay sequential
//                inside one workgroup and the FRAMES run in parallel (one workgroup per frame).
//                The curve points of a limb come from a host-fitted line (interpPoints, :66-88)
//                evaluated here exactly as numpy does (linspace, a*x+b, truncation).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace rib {

struct RasterStroke {      // == rib_stroke (include/rib.h)
  int32_t n;               // number of curve samples, 0 = limb not drawn
  int32_t swap;            // 1: the line was fitted as x = a*y + b (interpPoints' recursion)
  double start, step, stop;   // np.linspace(int(x0), int(x1), n): sample k = k*step + start, last = stop
  double a, b;
};

// scipy 'reflect' extension (d c b a | a b c d | d c b a), any overshoot
__device__ inline int reflect_idx(int i, int n) {
  if (n == 1) return 0;
  const int p = 2 * n;
  i %= p;
  if (i < 0) i += p;
  return i < n ? i : p - 1 - i;
}

// a*x + b with two roundings, as numpy evaluates it.  HIP's __dmul_rn / __dadd_rn are plain
// operators that the default -ffp-contract=fast may fuse into one fma, hence the pragma.
__device__ inline double mul_add_unfused(double a, double x, double b) {
#pragma clang fp contract(off)
  const double m = a * x;
  return m + b;
}

// One output sample of NI_Correlate1D's symmetric branch for a line whose only non-zero is `v` at j.
// w[d] = weight of taps +-d.  Unfused fp64 multiply/add in scipy's order.
__device__ inline double corr1d_onehot(int i, int j, int n, double v, const double* w, int r) {
  double acc = (i == j ? v : 0.0) * w[0];
  for (int d = r; d >= 1; --d) {
    const double s = (reflect_idx(i - d, n) == j ? v : 0.0) + (reflect_idx(i + d, n) == j ? v : 0.0);
    acc = mul_add_unfused(s, w[d], acc);
  }
  return acc;
}

struct HeatParams {
  const int32_t* peaks;    // [T][nmaps][2] (x, y) of the one-hot, x < 0: map is all zero
  const double* w;         // [r+1]
  int r;
  float* label;            // [T][label_nc][H][W]
  int T, H, W, nmaps, label_nc, ch0;
};

// grid (ceil(HW/256), nmaps, T).  radius <= 127.
__global__ __launch_bounds__(256) void k_heatmaps(const HeatParams p) {
  __shared__ double s_w[128];
  __shared__ double s_c[256];
  __shared__ double s_max;
  const int t = blockIdx.z, c = blockIdx.y, tid = threadIdx.x;
  const int x0 = p.peaks[((size_t)t * p.nmaps + c) * 2 + 0];
  const int y0 = p.peaks[((size_t)t * p.nmaps + c) * 2 + 1];
  const int pix0 = blockIdx.x * 256;
  const int pix = pix0 + tid;
  const int HW = p.H * p.W;
  float* dst = p.label + ((size_t)t * p.label_nc + p.ch0 + c) * HW;
  // blocks whose rows are farther than the radius from the peak only write zeros
  const bool near = x0 >= 0 && min(pix0 + 255, HW - 1) / p.W >= y0 - p.r && pix0 / p.W <= y0 + p.r;
  if (!near) {
    if (pix < HW) dst[pix] = 0.f;
    return;
  }
  for (int i = tid; i <= p.r; i += 256) s_w[i] = p.w[i];
  __syncthreads();
  // map.max(): with 'reflect' borders it need not sit on the peak, so search the footprint.  Every
  // output is monotone in the first-pass value, hence max = max_x pass2(x, max_y pass1(y)).
  {
    const int y = y0 - p.r + tid;
    s_c[tid] = (tid <= 2 * p.r && y >= 0 && y < p.H) ? corr1d_onehot(y, y0, p.H, 1.0, s_w, p.r) : 0.0;
  }
  __syncthreads();
  if (tid == 0) {
    double m = 0.0;
    for (int k = 0; k <= 2 * p.r; ++k) m = fmax(m, s_c[k]);
    s_max = m;
  }
  __syncthreads();
  const double v1max = s_max;
  {
    const int x = x0 - p.r + tid;
    s_c[tid] = (tid <= 2 * p.r && x >= 0 && x < p.W) ? corr1d_onehot(x, x0, p.W, v1max, s_w, p.r) : 0.0;
  }
  __syncthreads();
  if (tid == 0) {
    double m = 0.0;
    for (int k = 0; k <= 2 * p.r; ++k) m = fmax(m, s_c[k]);
    s_max = m;
  }
  __syncthreads();
  if (pix >= HW) return;
  const int y = pix / p.W, x = pix - y * p.W;
  float out = 0.f;
  if (abs(y - y0) <= p.r && abs(x - x0) <= p.r) {
    const double v1 = corr1d_onehot(y, y0, p.H, 1.0, s_w, p.r);
    if (v1 != 0.0) out = (float)(corr1d_onehot(x, x0, p.W, v1, s_w, p.r) / s_max);
  }
  dst[pix] = out;
}

struct SkelParams {
  const RasterStroke* strokes;   // [T][nedges]
  const uint32_t* colors;        // [nedges] r | g<<8 | b<<16
  int nedges;
  uint32_t* canvas;              // [T][H][W] scratch, packed like colors
  float* label;                  // [T][label_nc][H][W], channels 0..2 written
  int T, H, W, label_nc, bw;
};

constexpr int RASTER_MAXPTS = 2048;       // a limb has at most max(H, W) samples
constexpr int RASTER_PPT = RASTER_MAXPTS / 256;

__device__ inline uint32_t paint(uint32_t old, uint32_t color, bool average) {
  if (!average) return color;
  // ((old + c) / 2).astype(uint8) per channel (keypoint2img.py:41-43)
  const uint32_t r = ((old & 0xff) + (color & 0xff)) >> 1;
  const uint32_t g = (((old >> 8) & 0xff) + ((color >> 8) & 0xff)) >> 1;
  const uint32_t b = (((old >> 16) & 0xff) + ((color >> 16) & 0xff)) >> 1;
  return r | (g << 8) | (b << 16);
}

// grid (T), block 256: one workgroup paints one frame.
__global__ __launch_bounds__(256) void k_skeleton(const SkelParams p) {
  __shared__ int2 pts[RASTER_MAXPTS];
  const int t = blockIdx.x, tid = threadIdx.x;
  const int H = p.H, W = p.W, HW = H * W;
  uint32_t* cv = p.canvas + (size_t)t * HW;
  for (int i = tid; i < HW; i += 256) cv[i] = 0u;
  __syncthreads();

  for (int e = 0; e < p.nedges; ++e) {
    const RasterStroke S = p.strokes[(size_t)t * p.nedges + e];
    const int n = S.n;
    if (n <= 0) continue;                                  // uniform across the workgroup
    for (int k = tid; k < n; k += 256) {
      double lin = mul_add_unfused((double)k, S.step, S.start);          // np.linspace
      if (k == n - 1 && n > 1) lin = S.stop;
      if (n == 1) lin = S.start;
      const int u = (int)lin;                                            // .astype(int) truncates
      const int v = (int)mul_add_unfused(S.a, lin, S.b);                 // a * x + b
      pts[k] = S.swap ? make_int2(v, u) : make_int2(u, v);               // (x, y)
    }
    __syncthreads();
    const uint32_t color = p.colors[e];

    // the stroke: the whole curve shifted over a 2bw x 2bw square, one setColor per shift (:51-56)
    for (int i = -p.bw; i < p.bw; ++i) {
      for (int j = -p.bw; j < p.bw; ++j) {
        uint32_t old[RASTER_PPT];
        int idx[RASTER_PPT];
        int nz = 0;
#pragma unroll
        for (int q = 0; q < RASTER_PPT; ++q) {
          const int k = tid + q * 256;
          idx[q] = -1;
          if (k < n) {
            const int yy = max(0, min(H - 1, pts[k].y + i));
            const int xx = max(0, min(W - 1, pts[k].x + j));
            idx[q] = yy * W + xx;
            old[q] = cv[idx[q]];
            nz |= old[q] != 0u;
          }
        }
        const int any = __syncthreads_or(nz);               // (im[yy, xx] == 0).all() over the whole curve
#pragma unroll
        for (int q = 0; q < RASTER_PPT; ++q)
          if (idx[q] >= 0) cv[idx[q]] = paint(old[q], color, any != 0);
        __syncthreads();
      }
    }

    // the two end discs (:58-64): steps touch only {first, last} + (i, j)
    const int2 A = pts[0], B = pts[n - 1];
    const int R = 3 * p.bw;
    const bool inside = A.x - R >= 0 && A.x + R - 1 < W && A.y - R >= 0 && A.y + R - 1 < H &&
                        B.x - R >= 0 && B.x + R - 1 < W && B.y - R >= 0 && B.y + R - 1 < H;
    const bool apart = abs(A.x - B.x) >= 2 * R || abs(A.y - B.y) >= 2 * R;
    if (inside && apart) {
      // no pixel is touched by two different steps: the steps commute, one thread per shift
      for (int s = tid; s < 4 * R * R; s += 256) {
        const int i = s / (2 * R) - R, j = s % (2 * R) - R;
        if (i * i + j * j < 4 * p.bw * p.bw) {
          const int ia = (A.y + i) * W + A.x + j, ib = (B.y + i) * W + B.x + j;
          const uint32_t oa = cv[ia], ob = cv[ib];
          const bool avg = (oa | ob) != 0u;
          cv[ia] = paint(oa, color, avg);
          cv[ib] = paint(ob, color, avg);
        }
      }
    } else if (tid < 64) {
      // overlapping or clamped discs: the steps interact, walk them in order on one wavefront
      for (int i = -R; i < R; ++i) {
        for (int j = -R; j < R; ++j) {
          if (i * i + j * j >= 4 * p.bw * p.bw) continue;
          const int2 P = tid == 0 ? A : B;
          const int yy = max(0, min(H - 1, P.y + i)), xx = max(0, min(W - 1, P.x + j));
          uint32_t o = 0u;
          if (tid < 2) o = cv[yy * W + xx];
          const bool avg = __any(o != 0u);
          if (tid < 2) cv[yy * W + xx] = paint(o, color, avg);
          __threadfence_block();
        }
      }
    }
    __syncthreads();
  }

  // ToTensor + Normalize(0.5, 0.5): (u8 / 255 - 0.5) / 0.5 in fp32
  float* l0 = p.label + (size_t)t * p.label_nc * HW;
  for (int i = tid; i < HW; i += 256) {
    const uint32_t v = cv[i];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float f = __fdiv_rn((float)((v >> (8 * c)) & 0xff), 255.f);
      l0[(size_t)c * HW + i] = __fdiv_rn(__fsub_rn(f, 0.5f), 0.5f);
    }
  }
}

}  // namespace rib
