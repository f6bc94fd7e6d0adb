/*
 * rib.h — C ABI of the MI355X-native pose-guided generator ("Render-In-Between" hot path).
 *
 * The reference (azuxmioy/Render-In-Between) is pure Python and has no FFI layer: its boundary for
 * this path is the Python object protocol between the sequence driver and the generator module
 * (Pose_Guided_Neural_Rendering, abbreviated PGNR below).  Every entry point cites the reference
 * interface it replaces.  All functions return 0 on success or a negative rib_status; no C++
 * exception crosses this boundary.  rib_last_error() gives the message of the last failure.
 *
 * Ownership: the caller owns every tensor and the workspace; a handle owns only the weight blob
 * and its launch plans.  A handle is bound to one device, is single-stream and not re-entrant
 * (one handle per GPU / process).  Every launch of an entry point is enqueued, in order, on the caller's
 * stream; the handle owns no streams or events.
 * Synchronisation: rib_finalize_weights(), rib_read_tap() and rib_profile_collect() synchronise; rib_import_weights()
 * waits for its stream once (it reads the blob's 64-byte header back before accepting it); and the FIRST call that
 * needs the launch plan of a new (B,H,W) - rib_workspace_bytes / rib_forward / rib_chain / the introspection calls -
 * may allocate device memory for Winograd-domain filter sets and synchronise the device while it computes them
 * (one-time work per shape; query rib_workspace_bytes() up front to keep it out of a timed or captured region).
 * Steady-state calls of every other entry point only enqueue.
 *
 * Tensors at the boundary are dense fp32 NCHW on the handle's device, exactly what the reference
 * generator takes and returns (PGNR/models/evaluator.py:250-255).
 */
#ifndef RIB_H
#define RIB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rib_handle rib_handle;

typedef enum {
  RIB_OK = 0,
  RIB_ERR_INVALID = -1,      /* bad argument / unsupported shape */
  RIB_ERR_UNSUPPORTED = -2,  /* generator variant the path does not implement */
  RIB_ERR_STATE = -3,        /* e.g. forward before weights are loaded */
  RIB_ERR_MISSING = -4,      /* a required checkpoint tensor was never set */
  RIB_ERR_HIP = -5,          /* a HIP runtime call failed */
  RIB_ERR_WORKSPACE = -6     /* workspace too small */
} rib_status;

/* Resolved hyper-parameters of Generator(gen_cfg): the keys PGNR/models/generator.py:46-65,
 * 317-324, 431-440 reads from configs/HSM.yaml:35-67 (after getattr defaults). */
typedef struct {
  int32_t label_nc;          /* gen.input_label_nc   (22) */
  int32_t image_nc;          /* gen.input_image_nc   (3)  */
  int32_t num_filters;       /* gen.num_filters      (16) */
  int32_t max_num_filters;   /* gen.max_num_filters  (512) */
  int32_t num_layers;        /* gen.num_layers       (6)  */
  int32_t num_down_img;      /* gen.num_downsamples_img, default 4 (generator.py:50) */
  int32_t emb_filters;       /* gen.embed.num_filters      (64)  */
  int32_t emb_max_filters;   /* gen.embed.max_num_filters  (512) */
  int32_t emb_down;          /* gen.embed.num_downsamples  (4)   */
  int32_t mask_filters;      /* gen.mask.num_filters       (32)  */
  int32_t mask_max_filters;  /* gen.mask.max_num_filters   (512) */
  int32_t mask_down;         /* gen.mask.num_downsamples   (3)   */
  int32_t mask_res_blocks;   /* gen.mask.num_res_blocks    (4)   */
} rib_config;

/* ---- construction: replaces Generator(cfg.gen).to(device) (PGNR/models/trainer.py:61) ---- */
int rib_create(const rib_config* cfg, int device, rib_handle** out);
void rib_destroy(rib_handle* h);
/* h may be NULL: message of the last failed rib_create on this thread. */
const char* rib_last_error(const rib_handle* h);

/* ---- weights: replaces load_state_dict(net_G, path) (PGNR/utils/utils.py:107-119) ----
 * The handle speaks the reference checkpoint's vocabulary: the caller hands over the raw
 * state-dict tensors by their reference names ('down_0.conv_block_0.layers.conv.weight_orig',
 * '...weight_u', '...layers.norm.mlps.0.0.layers.conv.weight', ...).  rib_finalize_weights()
 * folds the eval-mode spectral norm  W = W_orig / (u . W_mat v)  (hook of
 * PGNR/models/layers/weight_norm.py:84-85), re-lays the filters out for the kernels and uploads
 * one contiguous device blob.  Tensors of modules the forward never calls (label_embedding.*,
 * conv_mask.*) are accepted and ignored; unknown names are an error (strict load). */
int rib_num_tensors(const rib_handle* h);
int rib_tensor_info(const rib_handle* h, int idx, const char** name, int* ndim, int64_t dims[4],
                    int* used);
int rib_set_tensor(rib_handle* h, const char* name, const float* host_data, int ndim,
                   const int64_t* dims);
int rib_finalize_weights(rib_handle* h);
/* The folded device blob, for the one RCCL broadcast of the multi-GPU path: rank 0 finalizes and
 * exports the blob into a caller-owned device buffer, the caller broadcasts that buffer
 * (torch.distributed / RCCL over xGMI), every other rank imports it.  Both copies are
 * device-to-device and asynchronous on the given stream. */
size_t rib_weights_bytes(const rib_handle* h);
int rib_export_weights(rib_handle* h, void* dst_device, size_t bytes, void* hip_stream);
int rib_import_weights(rib_handle* h, const void* src_device, size_t bytes, void* hip_stream);

/* ---- storage type and arithmetic of the matrix-core contractions (no reference counterpart: the reference is
 * fp32 only; PGNR/models/trainer.py:51 builds a plain fp32 module).
 *   RIB_DTYPE_F32 (default)  fp32 activations and filters, exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): the mode every
 *                            parity claim and the bench headline are made in.
 *   RIB_DTYPE_BF16           BASELINE.json configs[2]: bf16 NHWC activations and bf16 filters in HBM and LDS, bf16
 *                            MFMA operands (v_mfma_f32_32x32x16_bf16), fp32 accumulation, fp32 InstanceNorm statistics
 *                            and SPADE arithmetic; the caller's tensors stay fp32 NCHW.
 *   RIB_DTYPE_F16            the same 16-bit storage layouts and kernels with IEEE half elements
 *                            (v_mfma_f32_32x32x16_f16): 11 significant bits instead of 8 at the same speed.  The bf16
 *                            mode's deviation from fp32 (1e-1 max / 8e-3 mean on a [-1,1] frame) is the format's, not the
 *                            kernels' (DESIGN 6); half brings it to ~1e-2 / 1e-3, and the network's activations and
 *                            filters sit well inside half's range (finite outputs are asserted by the tests).
 * The mode decides the weight-blob layout (16-bit filter copies are appended): set it before
 * rib_finalize_weights / rib_import_weights (a handle that still holds the state-dict tensors re-folds by itself;
 * tuned choices pinned with rib_set_choice are dropped, they name kernels of the other mode).  Blobs are
 * exchangeable only between handles of the same mode and layout: the blob starts with a 64-byte header (magic,
 * mode, size, a hash of the layout offsets) that rib_import_weights checks. ---- */
enum { RIB_DTYPE_F32 = 0, RIB_DTYPE_BF16 = 1, RIB_DTYPE_F16 = 3 };   /* (2 was round 2's retired split-bf16 mode) */
int rib_set_compute_dtype(rib_handle* h, int dtype);

/* ---- arithmetic of the GEMM-shaped launches in RIB_DTYPE_F32 mode (no reference counterpart; OPT-IN, round 6).
 *   RIB_PRODUCTS_F32 (default)  exact-fp32 matrix-core products everywhere: the reference's arithmetic, the mode every parity
 *                               claim and the bench headline are made in.
 *   RIB_PRODUCTS_BF16X3         the plain GEMMs of the frame (k_gemm_dma: the Winograd-domain GEMMs of the deep 3x3 layers and the
 *                               gamma/beta GEMM of a condition level) split every fp32 operand value into three bf16 numbers
 *                               (hi + mid + lo, exact) in registers and form a product from six bf16 matrix-core products
 *                               accumulated in fp32 (dropped cross terms < 2^-24 |a b|).  Storage, layouts, the weight blob
 *                               and every other kernel are unchanged.  Against an fp64 GEMM it is no less accurate than the
 *                               exact-fp32 MFMA chain (rms error 0.4-0.9x: fewer roundings of the running sum), but it is
 *                               not the reference's arithmetic: results differ from the default's by fp32 rounding noise.
 * Takes effect at the next launch (plans and tuned choices are shared between the two); ignored by the 16-bit modes. ---- */
enum { RIB_PRODUCTS_F32 = 0, RIB_PRODUCTS_BF16X3 = 1 };
int rib_set_products(rib_handle* h, int products);

/* ---- forward: replaces  img, mask = net_G(label, label_prev, img_fake, img_prev)
 *      (PGNR/models/generator.py:181-234; call site PGNR/models/evaluator.py:255) ----
 * label [B,label_nc,H,W], img_fake/img_prev [B,image_nc,H,W] -> img [B,image_nc,H,W] (tanh),
 * mask [B,1,H,W] (sigmoid).  label_prev is dead in the reference and has no
 * parameter.  H and W must be multiples of 2^max(num_down_img, mask_down). */
size_t rib_workspace_bytes(rib_handle* h, int B, int H, int W);
int rib_forward(rib_handle* h, int B, int H, int W, const float* label, const float* img_fake,
                const float* img_prev, float* img, float* mask, void* workspace,
                size_t workspace_bytes, void* hip_stream);
/* The same forward with the driver's blend fused in: fuse = img*mask + img_fake*(1-mask), mask broadcast over the
 * image channels (PGNR/models/evaluator.py:256-258), written by the kernel that computes the mask.  fuse [B,3,H,W]
 * fp32 NCHW device memory; may be null (then this IS rib_forward). */
int rib_forward_blend(rib_handle* h, int B, int H, int W, const float* label, const float* img_fake,
                      const float* img_prev, float* img, float* mask, float* fuse,
                      void* workspace, size_t workspace_bytes, void* hip_stream);

/* ---- autoregressive segment: replaces the inference loop body of
 *      Evaluator.evaluate_from_folder (PGNR/models/evaluator.py:238-262) ----
 * prev <- key_frame; for t in 0..T-1:
 *   img_t, mask_t = G(labels[t], ., dains[t], prev); fuse_t = img_t*mask_t + dains[t]*(1-mask_t);
 *   prev <- fuse_t.
 * labels [T,B,label_nc,H,W], dains [T,B,image_nc,H,W], key_frame [B,image_nc,H,W];
 * outputs imgs/fuses [T,B,image_nc,H,W], masks [T,B,1,H,W]; imgs and masks may be NULL. */
/* Workspace for a T-frame rib_chain: rib_workspace_bytes plus room for the label-only launches of the whole
 * segment at batch T*B (they run once per chain instead of once per frame).  A caller that passes only
 * rib_workspace_bytes still gets a correct chain with per-frame label work. */
size_t rib_chain_workspace_bytes(rib_handle* h, int T, int B, int H, int W);
int rib_chain(rib_handle* h, int T, int B, int H, int W, const float* key_frame,
              const float* labels, const float* dains, float* imgs, float* masks, float* fuses,
              void* workspace, size_t workspace_bytes, void* hip_stream);

/* fuse = img*mask + dain*(1-mask) (PGNR/models/evaluator.py:256-258); n = B*H*W pixels. */
int rib_blend(rib_handle* h, int B, int C, int H, int W, const float* img, const float* mask,
              const float* dain, float* fuse, void* hip_stream);
/* uint8 HWC frame = uint8(clip(x*0.5+0.5,0,1)*255), truncating (PGNR/utils/utils.py:129-142). */
int rib_quantise(rib_handle* h, int B, int C, int H, int W, const float* img_nchw,
                 uint8_t* out_nhwc, void* hip_stream);
/* ---- label-map rasterisation --------------------------------------------------
 * Replaces, per frame, Dataset._generate_skeleton + _generate_pose_map
 * (PGNR/datasets/HSM_auto_dataset.py:205-251; drawing rules PGNR/utils/keypoint2img.py:36-88,132-147)
 * as called from the inference pre-load loop (PGNR/models/evaluator.py:221-229,250): writes the
 * 22-channel label [3-ch limb drawing in [-1,1] | n_maps gaussian joint maps in [0,1]] of T frames
 * straight into device memory.  The host keeps what is file parsing and scalar work: reading the
 * OpenPose json, thresholding the joints and fitting one line per limb (interpPoints,
 * keypoint2img.py:66-88); the pixels are produced here, bit-identical to the reference.
 *
 * rib_stroke: one limb of one frame.  n = int(x1 - x0) samples of np.linspace(int(x0), int(x1), n)
 * (sample k = k*step + start, the last one = stop), each mapped through a*x + b; swap = 1 when the
 * line was fitted with the roles of x and y exchanged.  n = 0: limb not drawn.
 * peaks: (x, y) = (int(x), int(y)) of each joint's one-hot, x = -1 when the joint is off.
 * weights: the radius+1 half of scipy's normalised gaussian kernel (weights[d] = tap +-d).
 * All four tables are HOST pointers (a few KB per frame): they are copied into page-locked staging memory of the handle
 * before the call returns and go to the workspace in one asynchronous copy on `stream` - the call only enqueues. */
typedef struct rib_stroke {
  int32_t n, swap;
  double start, step, stop, a, b;
} rib_stroke;
size_t rib_rasterise_workspace_bytes(rib_handle* h, int T, int H, int W, int n_edges, int n_maps, int radius);
int rib_rasterise(rib_handle* h, int T, int H, int W,
                  const rib_stroke* strokes, int n_edges, const uint8_t* colors_rgb, int stroke_halfwidth,
                  const int32_t* peaks, int n_maps, const double* weights, int radius,
                  float* labels, void* workspace, size_t workspace_bytes, void* hip_stream);

/* Extension op named by the north star but absent from the reference: bilinear
 * flow-grid warp, semantics of torch.nn.functional.grid_sample(img, base+flow*2/(size-1),
 * 'bilinear', padding_mode='border', align_corners=True).  flow [B,2,H,W] in pixels (x,y). */
int rib_warp(rib_handle* h, int B, int C, int H, int W, const float* img, const float* flow,
             float* out, void* hip_stream);

/* ---- introspection for parity tests (no reference counterpart) ----
 * After a rib_forward on `workspace`, intermediate activations can be read back as NCHW - the reference-side
 * equivalent is a forward hook on a leaf module.  The plan gives buffers with disjoint lifetimes
 * the same workspace bytes, so taps must be switched on BEFORE the forward: rib_set_debug_taps(h, 1) keeps every
 * tapped activation intact until the end of the forward (and rebuilds the launch plans). */
int rib_set_debug_taps(rib_handle* h, int enable);
int rib_num_taps(rib_handle* h, int B, int H, int W);
int rib_tap_info(rib_handle* h, int B, int H, int W, int idx, const char** name, int* C, int* th,
                 int* tw);
int rib_read_tap(rib_handle* h, int B, int H, int W, int idx, const void* workspace,
                 float* dst_nchw_device, void* hip_stream);

/* ---- measurement: per-kernel-class device time with HIP events on the caller's stream ----
 * rib_profile_begin() makes subsequent forwards record ONE event in front of every launch and one behind the
 * last (slower; never enable inside a timed region).  A launch is charged the time from its event to the next
 * one, so the classes add up to the forward's time on the stream, dispatch gaps included.
 * rib_profile_collect() synchronises and returns, per class, the number of launches and total milliseconds
 * since begin.  RIB_KC_CONVAUX holds the launches that are part of computing a convolution of class
 * RIB_KC_IGEMM but are not matrix-core kernels: the Winograd input / output transforms and the split-K slab
 * sums - a convolution's time is IGEMM + CONVAUX. */
enum { RIB_KC_IGEMM = 0, RIB_KC_SPADE = 1, RIB_KC_STATS = 2, RIB_KC_POOL = 3, RIB_KC_ELTWISE = 4,
       RIB_KC_PACK = 5, RIB_KC_CONVAUX = 6, RIB_KC_COUNT = 7 };
int rib_profile_begin(rib_handle* h);
/* rib_profile_begin_kernels(): the same bookkeeping, but every launch carries a (start, stop) event pair bound to the
 * dispatch itself (hipExtLaunchKernelGGL): rib_profile_collect() then returns the kernels' OWN execution times from the
 * queue's timestamps - the durations rocprofv3 --kernel-trace reports, with no event packet between two launches; the
 * classes no longer add up to the step (the dependent-launch gaps belong to no kernel). */
int rib_profile_begin_kernels(rib_handle* h);
int rib_profile_collect(rib_handle* h, int64_t launches[RIB_KC_COUNT], double ms[RIB_KC_COUNT]);
/* Algorithmic FLOPs (2*MAC) one rib_forward spends in class RIB_KC_IGEMM / RIB_KC_SPADE. */
int rib_forward_flops(rib_handle* h, int B, int H, int W, double flops[RIB_KC_COUNT]);
int rib_num_launches(rib_handle* h, int B, int H, int W);

/* ---- tuning hooks (tools/autotune.py): the launch plan picks, per convolution, one of a small set
 * of tile geometries and a split-K factor from an analytic cost model; a measured choice can be
 * pinned per (B,H,W, op name).  geom = {FRW,WM,WN,MF,NF,BK,STRIDE,KS,UPS,SPADE,KW,TB} (KW: wave
 * groups per workgroup, in-workgroup split-K; TB: filter slices staged per barrier). ---- */
int rib_num_variants(void);
int rib_variant_info(int idx, int geom[12]);   /* returns the precision of the instantiation (0 fp32, 1 bf16, 2 half), <0 on error */
int rib_set_choice(rib_handle* h, int B, int H, int W, const char* op_name, int variant_idx, int ksplit);
int rib_time_op(rib_handle* h, int B, int H, int W, const char* op_name, const float* label,
                const float* img_fake, const float* img_prev, float* img, float* mask, void* workspace,
                size_t workspace_bytes, int iters, void* hip_stream, double* usec);

/* ---- graph replay of rib_chain (no reference counterpart; a host-side option).  With it on (or RIB_GRAPH=1 in the
 * environment at rib_create), the first rib_chain call with a given (T,B,H,W) AND a given set of pointers captures the
 * segment's launches into one HIP graph and every later call with the same arguments is ONE hipGraphLaunch on the
 * caller's stream (which must not be the NULL stream - that one cannot be captured and keeps the launch-by-launch
 * path); calls with other tensors capture their own graph (at most 8 are kept, least recently used first out).
 * Same kernels, parameters and order: frames are bit-identical to the launch-by-launch path (tested).  Meant for hosts
 * that drive many GPUs from few cores, where enqueueing ~130 launches per frame per GPU becomes the limiter.
 * rib_graph_stats: how many calls captured / replayed since rib_create.
 * A captured segment holds the plans' kernels and parameters, so every call that changes them - rib_set_choice,
 * rib_set_plan_batch, rib_set_products, rib_set_compute_dtype, rib_set_debug_taps, rib_finalize_weights / rib_import_weights,
 * turning replay off - destroys the handle's captured graphs first, and so does the eviction of the least recently used one:
 * these calls BLOCK the host until the handle's device is idle (hipDeviceSynchronize: a graph may still be running on any
 * stream it was replayed on) whenever captured graphs exist.  With no captured graph they do not synchronise. ---- */
int rib_set_graph_replay(rib_handle* h, int enable);
int rib_graph_stats(rib_handle* h, int64_t* captures, int64_t* replays);

/* ---- batch-invariant launch plans (no reference counterpart: the reference renders one frame per call,
 * PGNR/models/evaluator.py:238-262, so its frames cannot depend on a grouping).  By default every batch size has its own
 * kernel choices (measured table or cost model: tile variant, split-K, Winograd tile, fused / level-wise SPADE), and a
 * sample rendered in a batch of 4 differs from the same sample rendered alone by ~1e-5 (other summation order).
 * rib_set_plan_batch(h, n) with n > 0 makes every plan built afterwards, whatever its batch, follow the choices of batch n:
 * a launch only grows in its per-sample grid dimension, a sample's arithmetic is the same in every grouping, and the
 * frames of rib_forward / rib_chain at B = 1, 2, 3, ... are bit-identical per sample (GPU test).  The folder driver
 * sets n to its group size, so that ragged last groups, other world sizes and --batch 1 all write the same bytes.
 * n = 0 restores the default.  Plans are rebuilt on demand; the required workspace of a shape may change with n
 * (query rib_workspace_bytes / rib_chain_workspace_bytes again). ---- */
int rib_set_plan_batch(rib_handle* h, int n);
int rib_get_plan_batch(const rib_handle* h);

/* ---- build identity (no reference counterpart: the reference is interpreted Python).  A static string
 *   "librib stamp=<lib> shards=<s0>,...,<s23> consistent=<0|1> variants=<n> compiler=<...>"
 * where <lib> is the content hash (csrc/build.py: sha256 over rib.hip, kernels.hip.h, igemm.hip.h, raster.hip.h,
 * variants.def, variants.hip.h, igemm_shard.hip, include/rib.h + flags + compiler version) rib.o was compiled with and <si> the hash
 * shard object i carries (24 of them, csrc/variants.hip.h); consistent=1 when all carry the hash rib.o expected of them.  bench.py
 * prints it in its JSON line; tests/test_native_host.py compares it with the hashes of the tracked tree. ---- */
const char* rib_build_info(void);

/* ---- host-only debugging (CPU tests): rib_create(cfg, device = -1, ..) builds a handle that owns
 * no device memory; it supports the tensor inventory, rib_set_tensor / rib_finalize_weights (the
 * folded blob stays on the host), plans (workspace size, launch list, FLOPs) and these readers,
 * which undo the filter re-layout so the fold can be compared with the oracle. ---- */
int rib_debug_conv_weight(rib_handle* h, const char* conv_name, float* w_oihw, float* bias);
int rib_debug_spade_weight(rib_handle* h, const char* conv_name, float* w_2c_by_cond, float* bias_2c);
int rib_debug_launch_info(rib_handle* h, int B, int H, int W, int idx, char* buf, size_t buflen);

#ifdef __cplusplus
}
#endif
#endif /* RIB_H */
