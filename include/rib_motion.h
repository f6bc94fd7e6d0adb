/*
 * rib_motion.h — C ABI of stage 1 on the MI355X: the motion transformer that turns low-frame-rate
 * OpenPose key frames into the interpolated pose sequence the generator is conditioned on
 * (SURVEY 8 row f-4).  HMM = the reference's Human_Motion_Modelling directory.
 *
 * The reference is pure Python and has no FFI layer; the boundary for this stage is the module
 * protocol between HMM/inference.py:Model_inference and HMM/models/transformer.py:Transformer.
 * Every entry point cites the reference interface it replaces.  Conventions as in rib.h: 0 on
 * success or a negative status, no C++ exception crosses the boundary, ribm_last_error() gives the
 * message of the last failure; the caller owns every tensor and the workspace, a handle owns only
 * its weight blob; work is enqueued on the caller's stream and nothing synchronises the device
 * except ribm_finalize_weights().
 *
 * Tensors at the boundary are dense fp32 on the handle's device in the reference's own layouts:
 * clips [N][C][L] (C = input_joints, L frames), padding masks uint8 [N][L] (1 = padded, what the
 * reference passes as bool key_padding_mask), positional encodings and outputs [L][N][*].
 */
#ifndef RIB_MOTION_H
#define RIB_MOTION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ribm_handle ribm_handle;

typedef enum {
  RIBM_OK = 0,
  RIBM_ERR_INVALID = -1,      /* bad argument / unsupported shape */
  RIBM_ERR_UNSUPPORTED = -2,  /* transformer variant the path does not implement */
  RIBM_ERR_STATE = -3,        /* forward before weights are loaded */
  RIBM_ERR_MISSING = -4,      /* a required checkpoint tensor was never set */
  RIBM_ERR_HIP = -5,          /* a HIP runtime call failed */
  RIBM_ERR_WORKSPACE = -6     /* workspace too small */
} ribm_status;

enum { RIBM_ACT_RELU = 0, RIBM_ACT_GELU = 1, RIBM_ACT_LEAKY_RELU = 2 };   /* HMM/models/transformer.py:365-375 */

/* build_transformer(args) arguments (HMM/models/transformer.py:349-362; HMM/configs/config.yaml
 * 'transformer:' section).  Dropout is inference-inert; 'intermediate' (stacked decoder
 * activations) is not supported. */
typedef struct {
  int32_t input_joints;      /* transformer.input_joints     (38)  */
  int32_t hidden_dim;        /* transformer.hidden_dim       (128): multiple of 4, <= 256 */
  int32_t nheads;            /* transformer.nheads           (8):  head_dim in {8, 16, 32, 64} */
  int32_t dim_feedforward;   /* transformer.dim_feedforward  (256): <= 1024 */
  int32_t enc_layers;        /* transformer.enc_layers       (6)   */
  int32_t dec_layers;        /* transformer.dec_layers       (6)   */
  int32_t activation;        /* transformer.activation       (RIBM_ACT_LEAKY_RELU) */
  int32_t pre_norm;          /* transformer.pre_norm         (1)   */
  int32_t two_stage;         /* transformer.two_stage        (1)   */
} ribm_config;

/* ---- construction: replaces build_transformer(cfg.transformer).to(device)
 * (HMM/models/transformer.py:349-362).  device < 0: host-only handle (inventory and strict-load
 * checks; no launches). ---- */
int ribm_create(const ribm_config* cfg, int device, ribm_handle** out);
void ribm_destroy(ribm_handle* h);
const char* ribm_last_error(const ribm_handle* h);   /* h may be NULL: last failed ribm_create of this thread */
/* "rib-stamp motion <hash>": content hash of motion.hip + this header + flags + compiler (csrc/build.py), as rib_build_info() */
const char* ribm_build_info(void);

/* ---- weights: replaces load_state_dict(net, path) (HMM/utils/utils.py:66-80: strict) ----
 * The caller hands over the raw state-dict tensors by their reference names
 * ('encoder.layers.0.self_attn.in_proj_weight', 'decoder.norm.bias', ...).  Unknown names and
 * wrong shapes are errors; ribm_finalize_weights() fails if any tensor is missing, transposes the
 * matrices for coalesced reads and uploads one device blob. */
int ribm_num_tensors(const ribm_handle* h);
int ribm_tensor_info(const ribm_handle* h, int idx, const char** name, int* ndim, int64_t dims[2]);
int ribm_set_tensor(ribm_handle* h, const char* name, const float* host_data, int ndim, const int64_t* dims);
int ribm_finalize_weights(ribm_handle* h);
size_t ribm_weights_bytes(const ribm_handle* h);

/* ---- forward: replaces Transformer.forward(src, src_mask, src_pos, tgt, tgt_mask, tgt_pos, rate)
 * (HMM/models/transformer.py:78-111), called by Model_inference.inference (HMM/inference.py:20-41).
 *   src, tgt      [N][C][L]   key-frame clip (non-key frames zeroed) / linearly interpolated clip;
 *                             tgt is only read when two_stage == 0 (may be NULL otherwise)
 *   src_mask      [N][L] u8   1 = frame is not a key frame / padded (encoder + memory key padding)
 *   tgt_mask      [N][L] u8   decoder key padding (all 0 in the OpenPose path)
 *   src_pos, tgt_pos [L][N][hidden_dim]   PositionEmbeddingSine_1D outputs (HMM/models/position_encoding.py:25-50)
 *   rate          key-frame spacing; (L - 1) % rate must be 0 when two_stage (interpolate_embedding
 *                 indexes the key frames, transformer.py:59-75)
 *   joints, reco  [L][N][C]   outputs; reco may be NULL
 * As in the reference, an encoder frame may not attend to itself (transformer.py:113-119) and a
 * query whose keys are all masked yields NaN.  ws: ribm_workspace_bytes(N, L) bytes of scratch. */
size_t ribm_workspace_bytes(const ribm_handle* h, int N, int L);
int ribm_num_launches(const ribm_handle* h);
int ribm_forward(ribm_handle* h, int N, int L, int rate, const float* src, const uint8_t* src_mask,
                 const float* src_pos, const float* tgt, const uint8_t* tgt_mask, const float* tgt_pos,
                 float* joints, float* reco, void* ws, size_t ws_bytes, void* stream /* hipStream_t */);

#ifdef __cplusplus
}
#endif
#endif
