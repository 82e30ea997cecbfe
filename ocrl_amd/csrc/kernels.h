// Internal launcher declarations shared by the kernel translation units, the model
// orchestration (slate_model.cpp) and the C ABI (capi.cpp).  Not part of the public ABI.
#pragma once
#include "common.h"

// ------------------------------------------------------------------ prof.cpp
enum { PROF_CONV5 = 0, PROF_CONV_OTHER = 1, PROF_WGRAD = 2, PROF_GEMM = 3, PROF_SA_FWD = 4, PROF_SA_BWD = 5, PROF_ATTN_FWD = 6, PROF_ATTN_BWD = 7, PROF_NTAGS = 8 };
int prof_begin(int tag, hipStream_t st);
void prof_end(int idx, hipStream_t st);

// ------------------------------------------------------------------ gemm.hip
struct GemmArgs {
    const float* A = nullptr;
    const float* B = nullptr;
    float* C = nullptr;
    int M = 0, N = 0, K = 0;
    int lda = 0, ldb = 0, ldc = 0;
    int akc = 1, bkc = 1;                 // operand storage, see gemm.hip
    int batch = 1;                        // total batches; batch index z -> (z / batch_inner, z % batch_inner)
    int batch_inner = 1;
    long long sA = 0, sB = 0, sC = 0;     // outer batch strides (elements)
    long long sAi = 0, sBi = 0, sCi = 0;  // inner batch strides (e.g. attention heads inside a [B,T,d] tensor)
    int splitk = 1;
    long long sCsplit = 0;                // slab stride when splitk > 1 (raw alpha*acc partials)
    // epilogue (ignored when splitk > 1):  v = alpha*acc + bias[n]; relu; dropout; *(mask>0); + resid
    float alpha = 1.f;
    const float* bias = nullptr;
    int relu = 0;                         // 0 none, 1 ReLU, 2 ELU
    float drop_p = 0.f;
    unsigned long long drop_seed = 0;
    unsigned drop_site = 0;
    const float* mask = nullptr; int ldmask = 0; long long sMask = 0;
    int mask_elu = 0;                     // mask holds ELU outputs: v *= (m > 0 ? 1 : m + 1) instead of the ReLU gate
    const float* resid = nullptr; int ldr = 0; long long sR = 0;
    // A-operand dropout (backward of y = x + dropout(branch)): A element (m, n) of the logical row-major [rows, adrop_ld]
    // tensor is multiplied by keep/(1-p) of dropout site adrop_site while it is staged (no separate mask pass)
    float adrop_p = 0.f; unsigned adrop_site = 0; int adrop_ld = 0;
    // fused bias gradient for the dW form (akc == 0): bias_out[m] = sum_k A(m,k)  (partials per split at stride sBias)
    float* bias_out = nullptr; long long sBias = 0;
    // ---- soft-max heads fused into the product (the [rows, vocabulary] tensor is written once and never re-read by an elementwise pass)
    // operand transforms, applied while an operand tile is staged; `row` is the token row (the m index of a k-contiguous A, the k index
    // of an m/n-contiguous A or B), `col` the vocabulary index:
    //   2: exp(x - x_lse[row])                                    (soft-max probabilities rebuilt from the stored scores)
    //   3: (exp(x - x_lse[row]) - [col == x_tok[row]]) * x_scale  (cross-entropy gradient rebuilt from the stored logits; A only)
    int a_mode = 0, b_mode = 0;
    const float* x_lse = nullptr; const int* x_tok = nullptr; float x_scale = 1.f;
    // epilogue modes (128x128 tiles, k-contiguous A, N % 4 == 0, no split-k / batches):
    //   1: C = alpha*acc (+bias) and per (row, 64-column segment) soft-max statistics  stat[(row*nseg + seg)*2 + {0,1}] = (max, sum exp(v - max))
    //   2: Gumbel head: l = acc + bias;  C = (l + g1) * e_scale  with statistics as in 1;  the hard sample's segment maximum of
    //      l + g2 and its column go to hstat / hidx.  g = -log(E + tiny), E from e1 / e2 (injected Exp(1) noise, [M,N]) or
    //      from the counter RNG (e_seed, the same draws as gumbel_softmax_kernel)
    //   3: soft-max backward: C = exp(mask - e_lse[row]) * (acc - e_rowvec[row]) * e_scale   (mask = the stored scores; C may alias mask)
    int epi_mode = 0;
    float* stat = nullptr; float* hstat = nullptr; int* hidx = nullptr;
    const float* e1 = nullptr; const float* e2 = nullptr; unsigned long long e_seed = 0;
    const float* e_lse = nullptr; const float* e_rowvec = nullptr; float e_scale = 1.f;
};
inline int gemm_stat_segments(int N) { return 2 * ((N + 127) / 128); }
int gemm_launch(const GemmArgs& a, hipStream_t st);
int splitk_reduce_launch(const float* part, float* out, long long n, int splits, long long stride,
                         int accumulate, hipStream_t st);

// ------------------------------------------------------------------ pool.hip
int pool_embed_launch(const float* lin, const float* cls, const float* pos, float* x0, int B, int K, int d, hipStream_t st);
int pool_rows_launch(const float* in, float* out, int B, int K, int d, int mode, hipStream_t st);
int pool_attn_launch(const float* qkv, float* P, float* O, const float* dO, float* dqkv, int B, int S, int d, int h, float p, unsigned long long seed,
                     unsigned site, int backward, hipStream_t st);

// ------------------------------------------------------------------ conv.hip
struct ConvArgs {
    const float* X = nullptr;       // [B,H,W,CIN] NHWC
    const float* Wp = nullptr;      // packed weights [KS*KS][CIN/8][COUT][8]
    float* Y = nullptr;             // [B,H,W,COUT]
    int B = 0, H = 0, W = 0;
    const float* bias = nullptr;    // [COUT]
    int relu = 0;                   // 0 none, 1 ReLU, 2 ELU
    const float* posmap = nullptr;  // [H,W,COUT] added after bias/relu
    const float* mask = nullptr;    // [B,H,W,COUT]: output zeroed where mask <= 0 (ReLU backward)
    int mask_elu = 0;               // mask holds ELU outputs: v *= (m > 0 ? 1 : m + 1)
};
// conv_x3.hip (exploratory): the 5x5 / 64-channel layer on the bf16 matrix pipe with every fp32 operand split exactly into three bf16
size_t conv_x3_pack_floats(int KS = 5);
int conv_pack_x3_launch(const float* W, float* fwd3, float* bwd3, hipStream_t st, int KS = 5);          // W [64][64][KS][KS]; bwd3 may be null
int conv_x3_launch(const ConvArgs& a, const float* pack3, hipStream_t st, int KS = 5);                   // KS = 5 or 3
struct WgradArgs;
int conv_wgrad_x3_stage(const WgradArgs& a, int nchunk, hipStream_t st, int KS = 5);       // first stage of conv_wgrad_launch (same partial slabs)
struct WgradArgs {
    const float* X = nullptr;       // [B,H,W,CIN]
    const float* dY = nullptr;      // [B,H,W,COUT]
    float* part = nullptr;          // workspace, conv_wgrad_ws_floats()
    int B = 0, H = 0, W = 0;
};
// low_latency: the caller's grid is small (inference at a few images): 5x5 / 64-channel forward layers whose throughput grid would leave
// most CUs idle go to the k-split kernel (conv_lat_kernel); results differ from the throughput kernel by fp32 summation order
int conv_fwd_launch(const ConvArgs& a, int KS, int CIN, int COUT, hipStream_t st, int low_latency = 0);
// x3: 1 = the split-precision first stage for the 5x5 / 64-channel layer (exploratory), 0 = fp32 MFMA, -1 = by OCRL_CONV_X3
int conv_wgrad_launch(const WgradArgs& a, int KS, int CIN, int COUT, int cin_real, float* dW, int accumulate, hipStream_t st, int x3 = -1);
size_t conv_wgrad_ws_floats(int B, int H, int W, int KS, int CIN);
int conv_pack_launch(const float* W, float* fwd, float* bwd, int KS, int CIN, int COUT, int cin_real, hipStream_t st);

// ------------------------------------------------------------------ elementwise.hip
int nchw_to_nhwc8_launch(const float* in, float* out, int B, int C, int H, int W, hipStream_t st);
int patchify4_launch(const float* in, float* out, int B, int C, int S, hipStream_t st);
int pixel_shuffle_launch(const float* in, float* out, int B, int h, int w, int Cout, int forward, const float* mask, hipStream_t st);
int layernorm_fwd_launch(const float* x, const float* g, const float* b, float* y, float* mean, float* rstd, long long R, int F, hipStream_t st);
int layernorm_bwd_launch(const float* dy, const float* x, const float* mean, const float* rstd, const float* g, float* dx,
                         float* dgb, long long R, int F, int accumulate_dx, int accumulate_dgb, float* ws, size_t ws_floats, hipStream_t st);
int colsum_launch(const float* X, long long ld, float* out, long long R, int F, int accumulate, float scale, float* ws, size_t ws_floats, hipStream_t st);
int reduce_partials_launch(const float* part, int n, float* out, float scale, int accumulate, hipStream_t st);
int mse_launch(const float* obs, const float* recon, float* drecon, float* out, int B, int C, int H, int W, float* ws, size_t ws_floats, hipStream_t st);
int gumbel_softmax_launch(const float* raw, const float* e1, const float* e2, float* z, int* tokens, long long R, int V, float tau,
                          unsigned long long seed, hipStream_t st, float* zst = nullptr);
int softmax_bwd_rows_launch(const float* z, float* d, long long R, int V, float scale, hipStream_t st);
int softmax_stat_combine_launch(const float* stat, int nseg, long long R, float* lse, const float* hstat, const int* hidx, int* tokens,
                                const float* pred, int ldp, const int* tok, float* out, float scale, float* ws, size_t ws_floats, hipStream_t st,
                                const float* cdf_scores = nullptr, int cdf_V = 0, float cdf_tau = 1.f, unsigned long long cdf_seed = 0);
int exp_rows_launch(const float* y, const float* lse, float* z, long long R, int V, hipStream_t st);
int rowdot_bias64_launch(const float* g, const float* act, const float* bias, long long R, float* out, hipStream_t st);
int ce_launch(float* pred, const int* tokens, float* out, long long R, int V, int B, int write_grad, float* ws, size_t ws_floats, hipStream_t st);
int embed_fwd_launch(const int* tokens, const float* dict, const float* bos, const float* pe, float* out, int B, int T, int d, float p,
                     unsigned long long seed, hipStream_t st);
size_t embed_bwd_ws_floats(long long BT, int V, int d);
int embed_bwd_launch(float* g, const int* tokens, float* ddict, int B, int T, int V, int d, float p, unsigned long long seed, float* ws,
                     size_t ws_floats, hipStream_t st);
int dropout_apply_launch(const float* x, float* y, long long n, float p, unsigned long long seed, unsigned site, hipStream_t st);
int dropout_mask_launch(float* y, long long n, float p, unsigned long long seed, unsigned site, hipStream_t st);
int cross_attn_fwd_launch(const float* Q, const float* Km, const float* Vm, float* O, float* P, int B, int T, int K, int d, int h, float p,
                          unsigned long long seed, unsigned site, hipStream_t st);
size_t cross_attn_bwd_ws_floats(int B, int T, int K, int d, int h);
int cross_attn_bwd_launch(const float* dO, const float* Q, const float* Km, const float* Vm, const float* P, float* dQ, float* dKm, float* dVm,
                          int B, int T, int K, int d, int h, float p, unsigned long long seed, unsigned site, float* part, size_t part_floats,
                          hipStream_t st);
int obs_u8_to_f32_launch(const unsigned char* in, float* out, int B, int H, int W, int C, hipStream_t st);
int fill_launch(float* x, long long n, float v, hipStream_t st);
int axpy_launch(const float* x, float* y, long long n, float a, hipStream_t st);
int posmap_launch(const float* Wpos, const float* bpos, float* out, int S, int C, hipStream_t st);
int posgrid_launch(float* out, int S, hipStream_t st);
int im2col5_launch(const float* x8, float* col, long long npix, int H, int W, int C, int ldc, hipStream_t st);
int unpack5_launch(const float* dWp, float* dW, int C, int ldc, hipStream_t st);
int pad_cols_launch(const float* in, int ldi, float* out, int ldo, long long R, int C, int Cout, hipStream_t st);

// ------------------------------------------------------------------ xattn.hip: cross attention to the slots, folded (transformer.py:23-50,181-185)
bool xattn_supported(int K, int d, int h);
int xattn_kp(int K, int h);                       // slots per head after padding (8 or 16); columns NC = h * KP
struct XaFoldHost {                               // per-image operands of all decoder blocks from the projected slots, one launch
    const float* mem = nullptr;                   // [B,K,d]
    const float* Wq[8]; const float* Wk[8]; const float* Wv[8]; const float* Wo[8];     // [d,d] (out, in) per block
    float* ck[8]; float* cv[8];                   // [B,K,d]
    float* Ab[8]; float* AbT[8]; float* Vo[8]; float* VoT[8];     // [B,NC,d] / [B,d,NC]; padding columns must be zero (never written)
    int B = 0, K = 0, d = 0, h = 0, nblk = 0;
};
int xattn_fold_fwd_launch(const XaFoldHost& f, hipStream_t st);
struct XaHost {
    const float* x = nullptr;                     // [B,T,d] attention input (LN of the residual stream)
    const float* resid = nullptr;                 // forward: residual stream
    float* y = nullptr;                           // forward: resid + dropout(out);  backward: d x
    float* P = nullptr;                           // [B,h,T,K] probabilities before dropout
    const float *Ab = nullptr, *AbT = nullptr, *Vo = nullptr, *VoT = nullptr;
    const float* gd = nullptr;                    // backward: gradient wrt out
    float *Pd = nullptr, *dS = nullptr;           // backward: [B*T, NC]
    int B = 0, T = 0, K = 0, d = 0, h = 0;
    float p = 0.f; unsigned long long seed = 0; unsigned site_p = 0, site_o = 0;
};
int xattn_launch(const XaHost& h, int backward, hipStream_t st);


// ------------------------------------------------------------------ slot_attn.hip
// Packed weight block (built once per step by pack_launch): originals and transposed copies.
struct SaWts {
    int ln_in_g, ln_in_b, ln_s_g, ln_s_b, ln_m_g, ln_m_b;
    int Wq, WqT, Wk, WkT, Wv, WvT, Wih, WihT, Whh, WhhT, bih, bhh, W0, W0T, b0, W2, W2T, b2;
    int total;
};
static inline SaWts sa_wts_layout(int C, int D, int H) {
    SaWts o; int a = 0;
    auto take = [&](int n) { int r = a; a += (n + 3) & ~3; return r; };
    o.ln_in_g = take(C); o.ln_in_b = take(C); o.ln_s_g = take(D); o.ln_s_b = take(D); o.ln_m_g = take(D); o.ln_m_b = take(D);
    o.Wq = take(D * D); o.WqT = take(D * D); o.Wk = take(D * C); o.WkT = take(C * D); o.Wv = take(D * C); o.WvT = take(C * D);
    o.Wih = take(3 * D * D); o.WihT = take(3 * D * D); o.Whh = take(3 * D * D); o.WhhT = take(3 * D * D);
    o.bih = take(3 * D); o.bhh = take(3 * D);
    o.W0 = take(H * D); o.W0T = take(H * D); o.b0 = take(H); o.W2 = take(D * H); o.W2T = take(D * H); o.b2 = take(D);
    o.total = a;
    return o;
}
// Saved-activation matrix: one row per (image, iteration, slot), row = (b*I + t)*K + j, fields at fixed
// column offsets, so every field is a strided [B*I*K, dim] GEMM operand with ld = sa_save_ld().
// With NH attention heads (ocrs/common/slot_attn.py:54-61) the folded query, the weighted means and the weight sums exist per
// (slot, head): qp / up / upn are NH blocks of C columns (head h at + h*C), csum NH values.
struct SaSave { int sprev, sn, q, u, r, z, n, hn, sg, m, hid, qp, up, upn, csum, ld; };
static inline SaSave sa_save_layout(int C, int D, int H, int NH = 1) {
    SaSave o;
    o.sprev = 0; o.sn = D; o.q = 2 * D; o.u = 3 * D; o.r = 4 * D; o.z = 5 * D; o.n = 6 * D; o.hn = 7 * D; o.sg = 8 * D; o.m = 9 * D;
    o.hid = 10 * D; o.qp = 10 * D + H; o.up = o.qp + NH * C; o.upn = o.up + NH * C; o.csum = o.upn + NH * C; o.ld = o.csum + ((NH + 3) & ~3);
    return o;
}
// Gradient rows emitted by the backward kernel (same row index), consumed by the weight-gradient GEMMs.
struct SaGrad { int out, hid, gi, gh, u, q, qp, ld; };
static inline SaGrad sa_grad_layout(int C, int D, int H, int NH = 1) {
    SaGrad o;
    o.out = 0; o.hid = D; o.gi = D + H; o.gh = 4 * D + H; o.u = 7 * D + H; o.q = 8 * D + H; o.qp = 9 * D + H; o.ld = 9 * D + H + NH * C;      // qp: NH blocks of C
    return o;
}
struct SlotAttnArgs {
    int B = 0, N = 0, C = 64, K = 0, D = 0, H = 0, I = 0;
    int NH = 1;                      // attention heads (ocrs/common/slot_attn.py:28): the soft-max runs over NH*K columns; NH*K <= 16, K <= 8 when NH > 1
    float eps = 1e-8f, scale = 1.f;  // scale = (D / NH)^-1/2
    const float* x = nullptr;        // [B,N,C]
    const float* slots0 = nullptr;   // [B,K,D]
    const float* wts = nullptr;      // packed weights, sa_wts_layout
    float* slots = nullptr;          // [B,K,D]
    float* attn = nullptr;           // [B,N,K] (last iteration, pre-eps; summed over the heads), may be null
    float* attn_heads = nullptr;     // NH > 1 and attn wanted: [B,N,NH*K] scratch for the per-head maps
    float* save = nullptr;           // [B*I*K, sa_save_layout().ld], may be null for inference
    const float* dslots = nullptr;   // [B,K,D]
    float* dx = nullptr;             // [B,N,C]
    float* dslots0 = nullptr;        // [B,K,D]
    float* grows = nullptr;          // [B*I*K, sa_grad_layout().ld]
    float* g_small = nullptr;        // [B][4D + 2C]: dgamma/dbeta of norm_slots, norm_mlp, norm_inputs
    // pipelined form (streaming launches with several workgroups per image alternate with slot-side launches): per-image exchange
    // buffer of B * sa_xchg_floats_host() floats and the partial-sum buffer of sa_parts_floats_host() floats; either null = the fused
    // one-workgroup-per-image kernels
    float* xchg = nullptr;
    float* parts = nullptr;
    // forward only: 0 = whole chain; 1 = the preparation launch alone (slots0 -> slots / folded query of iteration 0 in xchg; needs the
    // weights and slots0 but not x, so a caller can issue it long before the features exist); 2 = the chain without that launch
    int phase = 0;
};
size_t sa_xchg_floats_host(int K, int D);      // per image; K = num_slots * heads
size_t sa_parts_floats_host(int B, int K);     // K = num_slots * heads
int slot_attn_launch(const SlotAttnArgs& a, int backward, hipStream_t st);

// generic gather/transposing pack: entry e copies src[rows][cols] to dst + dst_off (transposed if requested)
struct PackEntry { const float* src; int rows, cols, dst_off, transpose; };
int pack_launch(const PackEntry* entries_dev, int n_entries, int max_elems, float* dst, hipStream_t st);

// ------------------------------------------------------------------ optim.hip
int absmax_launch(const float* g, long long n, float* out, float* ws, size_t ws_floats, hipStream_t st);
// g is first scaled by gscale (1/world for data parallel mean), then clipped by the inf-norm in norm[0]*gscale
int clip_adam_launch(float* p, const float* g, float* m, float* v, long long n, const float* norm, float clip, float lr, float b1,
                     float b2, float eps, int step, float gscale, hipStream_t st);
int store_u64_launch(unsigned long long* dst, unsigned long long v, hipStream_t st);
int slot_init_launch(const float* mu, const float* logsig, const float* noise, float* slots0, int BK, int D, unsigned long long seed, hipStream_t st,
                     const unsigned long long* seed_dev = nullptr);
int slot_init_bwd_launch(const float* dslots0, const float* logsig, const float* noise, float* dmu, float* dlogsig, int BK, int D, unsigned long long seed, hipStream_t st);
int copy_launch(const float* src, float* dst, long long n, hipStream_t st);
int argmax_pos_launch(const float* pred, int* tokens, int B, int T, int V, int t, hipStream_t st, int dense_rows = 0);
int embed_step_launch(const int* tokens, const float* dict, const float* bos, const float* pe, float* out, int B, int T, int d, int t, hipStream_t st);
int decode_attn_launch(const float* qkv, float* out, int B, int T, int t, int d, int h, int ld, hipStream_t st);
int onehot_launch(const int* tokens, float* z, long long rows, int V, hipStream_t st);

// ------------------------------------------------------------------ attention.hip
struct AttnArgs {
    const float *q = nullptr, *k = nullptr, *v = nullptr;   // projected (q unscaled), heads side by side, row stride ld (>= d)
    float* o = nullptr;                                     // [B,T,d]
    float* lse = nullptr;                                   // [B,h,T] log-sum-exp of the scaled scores
    int B = 0, T = 0, d = 0, h = 0;
    int ld = 0;                                             // row stride of q,k,v and dq,dk,dv (o, dO are dense [B,T,d])
    float p = 0.f;                                          // dropout on the probabilities
    unsigned long long seed = 0;
    unsigned site = 0;
    const float* dO = nullptr;                              // backward
    float *dq = nullptr, *dk = nullptr, *dv = nullptr, *delta = nullptr;
};
int attn_launch(const AttnArgs& a, int mode, hipStream_t st);

// ------------------------------------------------------------------ bcdec.hip (Slot-Attention broadcast decoder)
int bc_compose_launch(const float* W1, const float* Wpos, const float* bpos, float* Wc, float* W1r, int D, hipStream_t st);
int bc_posconv_launch(const float* Wc, float* P1, int S, hipStream_t st);
int bc_class_sum_launch(const float* in, float* out, int BK, int forward, hipStream_t st);
int bc_layer1_launch(const float* P1, const float* Tc, const float* b1, float* c1, int BK, int S, hipStream_t st);
size_t bc_layer1_bwd_ws_floats(int BK, int S);
int bc_layer1_bwd_launch(const float* g, float* dT, int BK, int S, float* ws, size_t ws_floats, hipStream_t st);
int bc_posconv_bwd_launch(const float* G, float* dWc, int S, hipStream_t st);
int bc_compose_bwd_launch(const float* W1, const float* Wpos, const float* bpos, const float* dWc, const float* dW1r, float* dW1,
                          float* dWpos, float* dbpos, int D, hipStream_t st);
int bc_c4_pack_launch(const float* W, float* Wk, float* Wb, int co_n, hipStream_t st);
int bc_c4_fwd_launch(const float* X, const float* Wk, const float* bias4, float* Y, int Bn, int S, hipStream_t st);
int bc_c4_bwd_data_launch(const float* dY, const float* Wb, const float* act, float* dX, int Bn, int S, hipStream_t st, int elu = 0);
int bc_c4_wgrad_blocks(int Bn, int S);
int bc_c4_wgrad_launch(const float* X, const float* dY, float* part, int Bn, int S, hipStream_t st);
int bc_mix_launch(const float* out4, const float* obs, float* recon, float* dout4, float* loss_out, int B, int K, int S, int C, float* ws,
                  size_t ws_floats, hipStream_t st);

// ------------------------------------------------------------------ iodine.hip (IODINE: ocrs/iodine/iodine_module.py)
int io_sample_launch(const float* mu, const float* ls, const float* noise, float* eps_out, float* slots, float* kl_out, long long n,
                     unsigned long long seed, unsigned site, float* ws, size_t ws_floats, hipStream_t st);
int io_w1_pack_launch(const float* W1, float* W1r, float* Wxy, int L, hipStream_t st);
int io_p1_launch(const float* Wxy, const float* b1, float* P1, int S, hipStream_t st);
int io_class_sum_launch(const float* in, float* out, long long BK, int forward, hipStream_t st);
int io_layer1_launch(const float* P1, const float* T, float* c1, long long BK, int S, hipStream_t st);
int io_layer1_bwd_launch(const float* g, float* dT, long long BK, int S, float* ws, size_t ws_floats, hipStream_t st);
int io_w1_grad_launch(const float* dW1r, const float* G, float* dW1, float* db1, int S, int L, hipStream_t st);
int io_elbo_launch(const float* out4, const float* obs, int B, int K, int S, float sigma, float* enc, float* st1, float* dout4, float* part,
                   float* masks_out, float* recon_out, float* rmasked_out, float* ws, size_t ws_floats, hipStream_t st);
int io_enc_norm_launch(float* enc, const float* st1, float* st2, long long BK, int N, float* ws, size_t ws_floats, hipStream_t st);
int io_elbo_bwd_launch(const float* out4, const float* obs, const float* denc, int B, int K, int S, float sigma, float cw, float* dout4, hipStream_t st);
int io_latent_launch(const float* mu, const float* ls, const float* eps, const float* ds, float* latent, long long BK, int L, float beta, int layer_norm,
                     int ld, hipStream_t st);
int io_im2col_launch(const float* x, float* col, long long Bn, int C, int Hi, int Wi, int ldc, hipStream_t st);
int io_col2im_launch(const float* dcol, const float* act, float* dx, long long Bn, int C, int Hi, int Wi, int ldc, hipStream_t st);
int io_refw_pack_launch(const float* W, float* Wp, int C, int ldc, hipStream_t st);
int io_refw_unpack_launch(const float* dWp, float* dW, int C, int ldc, hipStream_t st);
int io_pool_launch(const float* r, float* pool, long long BK, int n, hipStream_t st);
int io_pool_bwd_launch(const float* dpool, const float* r, float* dpre, long long BK, int n, hipStream_t st);
int io_elu2_launch(const float* a, int lda, float* y, int ldy, long long rows, int F, const float* dy, int lddy, hipStream_t st);
int io_lstm_fwd_launch(const float* gates, const float* c0, float* acts, float* c1, float* h1, long long rows, int H, hipStream_t st);
int io_lstm_bwd_launch(const float* acts, const float* c0, const float* c1, const float* dh1, const float* dc1, float* dgates, float* dc0,
                       long long rows, int H, hipStream_t st);
int io_post_grad_launch(const float* mu, const float* ls, const float* eps, const float* ds, const float* dlat, int ldl, float kw, float* gmu,
                        float* gls, long long BK, int L, hipStream_t st);
int io_l2norm_launch(const float* g, long long n, float* out, float* ws, size_t ws_floats, hipStream_t st);
int io_loss_launch(const float* parts, float* metrics, int I, int B, float beta, hipStream_t st);
