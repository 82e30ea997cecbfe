/* ocrl_hip — C ABI of the MI355X-native SLATE / Slot-Attention pre-training path.
 *
 * The reference (ugadiarov-la-phystech-edu/OCRL) is pure Python and has no native boundary of
 * its own; each entry point below names the reference function(s) whose arithmetic it replaces.
 * Conventions: every function returns 0 on success, non-zero on error with a thread-local message
 * from ocrl_last_error(); nothing throws across the ABI.  All pointers are DEVICE pointers to
 * fp32 (unless noted), 16-byte aligned, caller-owned; `stream` is a hipStream_t passed as void*
 * (NULL = default stream).  No call synchronises the device unless stated.  One handle per
 * process per GPU; calls on a handle are not thread-safe.
 */
#ifndef OCRL_HIP_H
#define OCRL_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OCRL_ABI_VERSION 5

const char* ocrl_last_error(void);
int ocrl_abi_version(void);

/* ---- model handle: SLATE_Module + SLATE/Base optimiser (ocrs/slate/slate_module.py:23-121,
 *      ocrs/slate/slate.py:13-34, ocrs/base.py:8-25) */
typedef struct ocrl_slate ocrl_slate;
typedef struct ocrl_slate_config {
    int obs_size, obs_channels;                 /* env_config.obs_size / obs_channels */
    int vocab_size, d_model;                    /* ocr_config.dvae.* */
    int cnn_hidden;                             /* ocr_config.cnn.hidden_size (must be 64) */
    int num_slots, num_iterations, slot_size, mlp_hidden;   /* ocr_config.slotattr.* */
    int num_dec_blocks, num_dec_heads;          /* ocr_config.tfdec.* */
    float dropout;                              /* ocr_config.learning.dropout */
    int max_batch;                              /* workspace is sized for this many images */
    int use_bcdec;                              /* ocr_config.use_bcdec: Slot-Attention configuration (broadcast decoder) */
    int hard;                                   /* ocr_config.hard: straight-through Gumbel sample into the dVAE decoder */
    int num_slot_heads;                         /* ocr_config.slotattr.num_slot_heads (ocrs/common/slot_attn.py:28,54-92); 0 is read as 1.
                                                 * heads > 1: heads * num_slots <= 16, num_slots <= 8, (slot_size / heads) % 16 == 0 */
} ocrl_slate_config;

/* sizeof(ocrl_slate_config) as this library was built: a binding checks its own struct against it before ocrl_slate_create */
size_t ocrl_slate_config_size(void);
int ocrl_slate_create(const ocrl_slate_config* cfg, ocrl_slate** out);
void ocrl_slate_destroy(ocrl_slate* h);

/* Parameter inventory in the reference's state_dict names and optimiser-group order
 * (slate_module.py:94-121): group 0 dvae, 1 enc/enc_pos/slotattn/slotproj, 2 dict/bos/pos/tfdec/out.
 * offset/numel are in floats inside the flat buffers handed to ocrl_slate_bind; every tensor
 * starts 16-byte aligned (gaps are zero and inert). */
int ocrl_slate_param_count(const ocrl_slate* h);
int ocrl_slate_param_info(const ocrl_slate* h, int i, char* name, int name_cap, int shape[4], int* ndim,
                          long long* offset, long long* numel, int* group);
long long ocrl_slate_flat_size(const ocrl_slate* h);
long long ocrl_slate_group_begin(const ocrl_slate* h, int group);      /* group in 0..3 (3 = end) */
size_t ocrl_slate_workspace_bytes(const ocrl_slate* h);

/* Adopt caller-allocated (e.g. torch) flat parameter / gradient / Adam-moment buffers of
 * ocrl_slate_flat_size() floats and a workspace of ocrl_slate_workspace_bytes() bytes, all
 * 256-byte aligned.  m/v may be NULL for inference.  Synchronises the device once. */
int ocrl_slate_bind(ocrl_slate* h, float* params, float* grads, float* adam_m, float* adam_v, void* workspace, size_t workspace_bytes);

/* SLATE_Module.get_loss (slate_module.py:198-241), masks=None path.  obs: [B,3,S,S] NCHW in [0,1].
 * noise_z / noise_zh: optional injected Exp(1) draws laid out [B,T,V] (the reference's two
 * torch.empty_like(logits).exponential_() tensors permuted to channel-last, utils.py:77);
 * noise_slots: optional injected N(0,1) [B,K,D] (slot_attn.py:155).  NULL = draw on device from
 * `seed`.  train != 0 enables dropout (masks derived from `seed`).  Results: ocrl_slate_metrics. */
int ocrl_slate_forward(ocrl_slate* h, const float* obs, int B, float tau, int train, unsigned long long seed,
                       const float* noise_z, const float* noise_zh, const float* noise_slots, void* stream);
/* loss.backward() of the last forward: fills the flat gradient buffer (overwrites). */
int ocrl_slate_backward(ocrl_slate* h, void* stream);
/* SLATE_Module.forward (slate_module.py:181-196): slots [B,K,D] and attention [B,N,K] only. */
int ocrl_slate_encode(ocrl_slate* h, const float* obs, int B, unsigned long long seed, const float* noise_slots, void* stream);
/* Backward of the last ocrl_slate_encode for a downstream loss (poolings/base.py:53-55, learn_downstream_loss=True: the slots enter the
 * pooling head undetached): dslots [B,K,D] -> the flat gradient buffer, overwritten: gradients of the CNN encoder, the positional
 * embedding and the slot-attention module; zeros elsewhere.  The next ocrl_slate_clip_adam then steps those tensors only (learning rate
 * lr[1]), as torch's Adam skips parameters without a gradient. */
int ocrl_slate_encode_backward(ocrl_slate* h, const float* dslots, void* stream);
/* Serving with a frozen encoder (sb3s/ocr_extractor.py:33-36 with a pre-trained checkpoint and finetuning off): on != 0 promises that
 * the parameters do not change until the next call with on = 0 (or a clip_adam / bind), so ocrl_slate_encode builds the derived
 * weight images once instead of at every call.  Writing the flat parameter buffer while frozen leaves them stale. */
int ocrl_slate_freeze_weights(ocrl_slate* h, int on);
/* SLATE_Module._gen_imgs (slate_module.py:163-179): greedy autoregressive token decode from the slots of the last
 * forward/encode, then dVAE decode into the "recon" tensor; metrics[4] = sum (obs - recon_tf)^2 / B.  Destroys the
 * activations of the last forward (no ocrl_slate_backward afterwards). */
int ocrl_slate_generate(ocrl_slate* h, void* stream);
/* clip_grad_norm_(params, clip, "inf") + Adam(3 groups).step() (base.py:65-72, slate.py:19-34):
 * grads are first scaled by grad_scale (1/world_size after an all-reduce sum), clip <= 0 disables
 * clipping, `step` is the 1-based Adam step count.  metrics[3] receives max|g| before scaling. */
int ocrl_slate_clip_adam(ocrl_slate* h, const float lr[3], float clip, int step, float grad_scale, void* stream);
int ocrl_slate_grad_norm(ocrl_slate* h, void* stream);
/* device float[8]: [0] dvae_mse (mse with use_bcdec), [1] cross_entropy, [2] loss, [3] max|grad|, [4] mse of ocrl_slate_generate */
float* ocrl_slate_metrics(const ocrl_slate* h);
/* Named tensors of the last step ("slots" [B,K,D], "attn" [B,N,K], "recon" [B,S,S,4], "tokens"
 * (int32) [B,T], "zraw" [B,T,V] (soft models: the Gumbel scores (logits + g)/tau until the backward overwrites them with d logits;
 * hard=True: the raw logits), "z" [B,T,V] (see ocrl_slate_soft_z), "z_lse"/"ce_lse" [B,T], "feats" [B,N,64], "dec_out" [B,T,d],
 * "pred" (the output-head logits), "mem", "emb",
 * "slots0", "sa_inputs") or any parameter name; count = capacity in elements at max_batch. */
int ocrl_slate_tensor(const ocrl_slate* h, const char* name, float** ptr, long long* count);
/* The soft Gumbel sample z = softmax((logits + g) / tau) of the last forward (ocrs/slate/slate_module.py:126, the `z` that
 * get_loss(with_rep=True) returns, :239-241) written into the named tensor "z".  The training step itself never materialises it:
 * the vocabulary products rebuild it from the stored scores ("zraw") and their row log-sum-exp ("z_lse").  Call between forward and
 * backward; a no-op for hard=True models, whose forward writes "z" / "z_st" itself. */
int ocrl_slate_soft_z(ocrl_slate* h, void* stream);
/* The keep-mask (float 0/1) the kernels used for a dropout site in the last forward; site ids:
 * 1 = z_pos, 16 + 8*block + {0 self.attn, 1 self.out, 2 cross.attn, 3 cross.out, 4 ffn}. */
int ocrl_slate_dropout_mask(const ocrl_slate* h, unsigned site, long long n, float* out, void* stream);

/* ---- unit entry points (parity tests of the MFMA kernels)
 * C[M,N] = alpha * op(A) op(B) (+bias[n]) (relu) (* (mask>0)) (+resid); akc/bkc select storage:
 * akc=1: A is [M,K] row-major, else [K,M]; bkc=1: B is [N,K] row-major (torch Linear weight), else [K,N].
 * splitk > 1 needs ws of splitk*M*N floats and ldc == N. */
int ocrl_gemm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int akc, int bkc,
              float alpha, const float* bias, int relu, const float* mask, int ldmask, const float* resid, int ldr,
              int splitk, float* ws, void* stream);
/* F.conv2d(x, w, b, stride 1, padding ks/2) on NHWC x [B,H,W,cin_pad] (cin_pad = 8 or 64; channels >= cin are
 * zero) with the reference-layout weight w [64,cin,ks,ks]; y [B,H,W,64] NHWC.  ws: ks*ks*cin_pad*64 floats. */
int ocrl_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int cin, int cin_pad, int ks,
                    int relu, float* ws, void* stream);
/* The same convolution for a few images at a time (the RL extractor's encode(): sb3s/ocr_extractor.py:45 at num_envs images): 5x5 /
 * 64-channel layers whose grid would leave most of the GPU idle run a kernel that splits one output tile over four waves and two
 * workgroups (k-split, ordered sum).  Same arguments and result up to fp32 summation order; other shapes take the kernel above. */
int ocrl_conv2d_fwd_lowlat(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int cin, int cin_pad, int ks,
                           int relu, float* ws, void* stream);
/* EXPLORATORY, not on any default path: the ks x ks (5 or 3) / 64 -> 64 layer (forward, and backward-data with the ReLU mask) on the bf16
 * matrix pipe, every fp32 operand split exactly into three bf16 numbers and six products accumulated in fp32 (csrc/conv_x3.hip).  NHWC
 * [B,H,W,64], reference-layout weight [64,64,ks,ks]; ws: ocrl_conv2d_x3_ws_floats() floats. */
size_t ocrl_conv2d_x3_ws_floats(void);
int ocrl_conv2d_fwd_x3(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int ks, int relu, float* ws, void* stream);
int ocrl_conv2d_bwd_data_x3(const float* dy, const float* w, const float* mask, float* dx, int B, int H, int W, int ks, float* ws, void* stream);
/* its weight gradient [64,64,ks,ks]; ws from ocrl_conv2d_wgrad_ws_floats(B, H, W, ks, 64) */
int ocrl_conv2d_bwd_weight_x3(const float* x, const float* dy, float* dw, int B, int H, int W, int ks, float* ws, size_t ws_floats, void* stream);
/* grad wrt input of the same conv (square 64->64 layers): dx = conv_transpose(dy, w) * (mask > 0 if mask). ws: 2*ks*ks*64*64 floats. */
int ocrl_conv2d_bwd_data(const float* dy, const float* w, const float* mask, float* dx, int B, int H, int W, int ks, float* ws, void* stream);
/* grad wrt weight (reference layout [64,cin,ks,ks]) and bias [64] (may be NULL); ws from ocrl_conv2d_wgrad_ws_floats. */
size_t ocrl_conv2d_wgrad_ws_floats(int B, int H, int W, int ks, int cin_pad);
int ocrl_conv2d_bwd_weight(const float* x, const float* dy, float* dw, float* db, int B, int H, int W, int cin, int cin_pad, int ks,
                           float* ws, size_t ws_floats, void* stream);
/* nn.LayerNorm(F) forward / backward over R rows (F in {64,128,192,256}); dgb = [dgamma | dbeta]. */
int ocrl_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, long long R, int F, void* stream);
int ocrl_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma, float* dx, float* dgb,
                       long long R, int F, float* ws, size_t ws_floats, void* stream);

/* Causal multi-head self-attention core of MultiHeadAttention.forward (ocrs/common/transformer.py:31-47): q,k,v are the
 * projected [B,T,d] tensors (row stride ld >= d, heads side by side, q unscaled); o = dropout(softmax(mask(q k^T / sqrt(dh)))) v
 * as [B,T,d]; lse [B,h,T] is saved for the backward.  Dropout decisions come from (seed, site) as in ocrl_slate_forward. */
int ocrl_attention_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int T, int d, int h, int ld,
                       float p, unsigned long long seed, unsigned site, void* stream);
/* gradients dq,dk,dv (row stride ld) from dO [B,T,d]; delta is scratch [B,h,T]. */
int ocrl_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* lse, const float* dO,
                       float* dq, float* dk, float* dv, float* delta, int B, int T, int d, int h, int ld,
                       float p, unsigned long long seed, unsigned site, void* stream);

/* ---- optional HIP-event timing of kernel families on the launch stream (bench.py's roofline line).
 * tag bits: 0 conv5x5/64ch fwd+bwd-data, 1 other convs, 2 conv weight-grad, 3 gemm, 4 slot-attn fwd, 5 slot-attn bwd,
 * 6 self-attention fwd, 7 self-attention bwd.
 * ocrl_prof_collect synchronises the device and returns total milliseconds / launch counts per tag. */
/* Input pipeline on the device: utils/datasets.py:13-24 (`torch.Tensor(obss[i]).permute(2, 0, 1) / 255.0`) + to_device
 * (utils/tools.py:182-191, train_ocr.py:52-53).  The host uploads the dataset's uint8 HWC images as they are (a quarter of the
 * fp32 bytes over PCIe); obs_chw = obs_hwc / 255 as correctly rounded fp32, bit-identical to the reference's conversion. */
int ocrl_obs_u8_to_f32(const unsigned char* obs_hwc, float* obs_chw, int B, int H, int W, int C, void* stream);

int ocrl_prof_enable(unsigned tag_mask);
int ocrl_prof_collect(double* ms, long long* count, int ntags);

/* ---- the slot-attention loop on its own (SlotAttention.forward, ocrs/common/slot_attn.py:47-102; single head) ----
 * x [B,N,64] inputs (before norm_inputs), slots0 [B,K,D]; `w` = 17 device pointers in the reference's parameter order:
 * norm_inputs.{weight,bias}, norm_slots.{weight,bias}, norm_mlp.{weight,bias}, project_q / project_k / project_v .weight,
 * gru.weight_ih, gru.weight_hh, gru.bias_ih, gru.bias_hh, mlp.0.{weight,bias}, mlp.2.{weight,bias}.
 * Outputs: slots [B,K,D], attn [B,N,K] (last iteration, before the epsilon; may be NULL).  The workspace keeps the saved
 * activations: call _bwd with the same ws right after _fwd.  _bwd: dslots [B,K,D] -> dx [B,N,64], dslots0 [B,K,D] and the
 * 17 weight gradients `dw` (same order and shapes).  Slot / MLP widths: multiples of 64 up to 256; 1 <= K <= 16. */
size_t ocrl_slot_attention_ws_floats(int B, int K, int D, int H, int I);
int ocrl_slot_attention_fwd(const float* x, const float* slots0, const float* const* w, float* slots, float* attn, int B, int N, int K, int D,
                            int H, int I, float* ws, size_t ws_floats, void* stream);
int ocrl_slot_attention_bwd(const float* x, const float* dslots, float* dx, float* dslots0, float* const* dw, int B, int N, int K, int D, int H,
                            int I, float* ws, size_t ws_floats, void* stream);
/* the same with `heads` attention heads (ocrs/common/slot_attn.py:28,54-92: q / k / v split into heads, soft-max over heads * K columns,
 * attn = sum over the heads): heads * K <= 16, K <= 8 and (D / heads) % 16 == 0 when heads > 1; heads = 1 is the pair above. */
size_t ocrl_slot_attention_mh_ws_floats(int B, int N, int K, int D, int H, int I, int heads);
int ocrl_slot_attention_mh_fwd(const float* x, const float* slots0, const float* const* w, float* slots, float* attn, int B, int N, int K, int D,
                               int H, int I, int heads, float* ws, size_t ws_floats, void* stream);
int ocrl_slot_attention_mh_bwd(const float* x, const float* dslots, float* dx, float* dslots0, float* const* dw, int B, int N, int K, int D, int H,
                               int I, int heads, float* ws, size_t ws_floats, void* stream);

/* ---- slot-set pooling head: poolings/common/transformer.py:9-33 (Transformer: Linear -> [CLS; tokens] (+pos) ->
 * nn.TransformerEncoder of post-norm ReLU layers -> CLS row), as built by poolings/transformer/transformer_module.py:27-117 with its
 * default switches; the consumer of the slots in sb3s/ocr_extractor.py:45.  SURVEY.md §8(f) rank 4.
 * slots [B,K,Din]; `w` = 3 + 12 L device pointers in the module's state_dict order: _linear.{weight [d,Din], bias},
 * _cls_token._cls_token [d], then per layer self_attn.in_proj_{weight [3d,d], bias}, self_attn.out_proj.{weight,bias},
 * linear1.{weight [ff,d], bias}, linear2.{weight [d,ff], bias}, norm1.{weight,bias}, norm2.{weight,bias}.
 * pos: [K+1,d] positional table added to the token sequence (the reference's `pe` buffers), or NULL (pos_emb "None").
 * out [B,d] = encoder output of the CLS token.  drop_p = nn.TransformerEncoderLayer's dropout in train mode (0 in eval); the keep
 * decisions are a pure function of (seed, layer, site, element) so _bwd regenerates them: pass the same drop_p and seed.
 * _bwd: dout [B,d] -> dw (same order / shapes as w; every entry is overwritten) and dslots [B,K,Din] (NULL = the slots are detached,
 * poolings/base.py:53).  Call it with the same ws right after _fwd.  d_model: multiple of 64 up to 256; K + 1 <= 32. */
#define OCRL_POOL_MAX_LAYERS 8
size_t ocrl_pool_transformer_ws_floats(int B, int K, int d, int nhead, int ff, int L);
int ocrl_pool_transformer_fwd(const float* slots, const float* const* w, const float* pos, float* out, int B, int K, int Din, int d, int nhead,
                              int ff, int L, float drop_p, unsigned long long seed, float* ws, size_t ws_floats, void* stream);
int ocrl_pool_transformer_bwd(const float* slots, const float* dout, const float* const* w, float* dslots, float* const* dw, int B, int K, int Din,
                              int d, int nhead, int ff, int L, float drop_p, unsigned long long seed, float* ws, size_t ws_floats, void* stream);
/* keep-mask (1 = kept) of one dropout site, for parity tests: which = 0 attention weights [B,h,S,S], 1 dropout1 [B,S,d],
 * 2 FFN hidden [B,S,ff], 3 dropout2 [B,S,d]; n = element count. */
int ocrl_pool_transformer_dropout_mask(int layer, int which, long long n, float drop_p, unsigned long long seed, float* out, void* stream);

/* ---- IODINE (ocrs/iodine/iodine_module.py:14-271, ocrs/iodine/iodine.py:4-14, ocrs/base.py:60-74): SURVEY.md §8 row a20 ----
 * Same conventions as the SLATE handle: flat fp32 parameter / gradient / Adam buffers in the reference's
 * _module.parameters() order and state_dict names, adopted from the caller; one workspace; all work on the caller's stream. */
typedef struct ocrl_iodine ocrl_iodine;
typedef struct ocrl_iodine_config {
    int obs_size, obs_channels;                 /* env_config.obs_size / obs_channels (3) */
    int slot_size, num_iterations, num_slots;   /* ocr_config.slot_size / num_iterations / num_slots */
    float sigma, beta;                          /* ocr_config.sigma / beta */
    int layer_norm;                             /* ocr_config.layer_norm */
    int ref_mlp_hidden;                         /* ocr_config.ref_mlp_hidden_size; conv widths 64, 4 layers, 3x3 (stride 2 / 1) are fixed */
    int max_batch;
} ocrl_iodine_config;
int ocrl_iodine_create(const ocrl_iodine_config* cfg, ocrl_iodine** out);
void ocrl_iodine_destroy(ocrl_iodine* h);
int ocrl_iodine_param_count(const ocrl_iodine* h);
int ocrl_iodine_param_info(const ocrl_iodine* h, int i, char* name, int name_cap, int shape[4], int* ndim, long long* offset, long long* numel);
long long ocrl_iodine_flat_size(const ocrl_iodine* h);
size_t ocrl_iodine_workspace_bytes(const ocrl_iodine* h);
int ocrl_iodine_bind(ocrl_iodine* h, float* params, float* grads, float* adam_m, float* adam_v, void* workspace, size_t workspace_bytes);
/* Iodine_Module._forward (iodine_module.py:79-252): obs [B,3,S,S] NCHW; noise: optional injected N(0,1) draws of the
 * per-iteration rsample, [I,B,K,L] (NULL = device RNG stream `seed`).  Results: ocrl_iodine_metrics / ocrl_iodine_tensor
 * ("slots" [B,K,L], "masks" [B,K,S,S], "recon" [B,3,S,S], "recons_masked" [B,K,3,S,S], "out4" [B*K,S,S,4]). */
int ocrl_iodine_forward(ocrl_iodine* h, const float* obs, int B, unsigned long long seed, const float* noise, void* stream);
/* loss.backward() of the last forward (loss = -sum_i (i+1)/I ELBO_i) into the flat gradient buffer */
int ocrl_iodine_backward(ocrl_iodine* h, void* stream);
/* clip_grad_norm_(params, clip, 2.0) + Adam(lr) (base.py:60-74); gscale = 1/world for a data-parallel mean */
int ocrl_iodine_clip_adam(ocrl_iodine* h, float lr, float clip, int step, float gscale, void* stream);
int ocrl_iodine_grad_norm(ocrl_iodine* h, void* stream);
/* device float[8]: [0] loss, [1] mse (last iteration), [2] kld (last iteration), [3] gradient L2 norm */
float* ocrl_iodine_metrics(const ocrl_iodine* h);
int ocrl_iodine_tensor(const ocrl_iodine* h, const char* name, float** ptr, long long* count);

/* ---- data-parallel exchange: one in-place SUM all-reduce of a flat fp32 gradient buffer per step over RCCL
 * (xGMI); pass gscale = 1/world to ocrl_*_clip_adam for the mean.  For hosts without torch.distributed; librccl is opened
 * lazily (dlopen).  unique id: 128 bytes produced on rank 0 and broadcast by the host's own means (as ncclGetUniqueId). */
typedef struct ocrl_comm ocrl_comm;
int ocrl_comm_unique_id(void* out128, size_t cap);
int ocrl_comm_init(ocrl_comm** out, int rank, int world, const void* unique_id128);   /* current HIP device = this rank's GPU */
int ocrl_comm_allreduce(ocrl_comm* c, float* device_buf, long long n, void* stream);
int ocrl_comm_world(const ocrl_comm* c);
void ocrl_comm_destroy(ocrl_comm* c);

#ifdef __cplusplus
}
#endif
#endif
