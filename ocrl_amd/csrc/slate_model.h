// SLATE training-step orchestration over the HIP kernels (host code).  One object per process
// per GPU; not thread-safe; every launch goes to the caller's stream.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "kernels.h"

struct SlateConfig {
    int obs_size = 64, obs_channels = 3, vocab = 4096, d_model = 192, cnn_hidden = 64;
    int num_slots = 5, num_iters = 3, slot_size = 192, mlp_hidden = 192;
    int num_blocks = 4, num_heads = 4;
    float dropout = 0.1f;
    int max_batch = 1;
    int use_bcdec = 0;      // Slot-Attention configuration: spatial-broadcast decoder instead of dVAE + transformer
    int hard = 0;           // ocr_config.hard: straight-through Gumbel sample for the dVAE decoder (utils.py:81-83)
    int slot_heads = 1;     // ocr_config.slotattr.num_slot_heads (ocrs/common/slot_attn.py:28)
};

struct ParamInfo {
    std::string name;
    int shape[4] = {1, 1, 1, 1};
    int ndim = 1;
    long long numel = 0;
    long long offset = 0;   // element offset into the flat parameter buffer (16-byte aligned)
    int group = 0;          // optimiser group: 0 dvae, 1 slot-attention side, 2 transformer decoder
};

struct StepInputs {
    const float* obs = nullptr;        // [B,3,S,S] NCHW fp32 in [0,1]
    int B = 0;
    float tau = 1.f;
    int train = 1;                     // dropout on/off
    unsigned long long seed = 0;       // device RNG stream for this step (noise + dropout)
    const float* noise_z = nullptr;    // optional injected Exp(1) draws [B,T,V] (both or none)
    const float* noise_zh = nullptr;
    const float* noise_slots = nullptr;  // optional injected N(0,1) [B,K,D]
    const unsigned long long* seed_dev = nullptr;   // internal: the slot-noise seed read from device memory (captured encode graphs)
};

// dropout-backward mask applied to dy while the GEMM stages it
struct Drop { float p = 0.f; unsigned site = 0; };
// soft-max operand transform of a product (GemmArgs::a_mode / b_mode): the operand is rebuilt from stored scores + per-row log-sum-exp
struct Xf { int a_mode = 0, b_mode = 0; const float* lse = nullptr; const int* tok = nullptr; float scale = 1.f; };

class SlateModel {
public:
    explicit SlateModel(const SlateConfig& c);
    ~SlateModel();
    SlateModel(const SlateModel&) = delete;
    SlateModel& operator=(const SlateModel&) = delete;
    const std::vector<ParamInfo>& params() const { return params_; }
    long long flat_size() const { return flat_size_; }
    long long group_begin(int g) const { return group_begin_[g]; }   // group g = [begin(g), begin(g+1))
    size_t workspace_bytes() const { return ws_bytes_; }
    int bind(float* p, float* g, float* m, float* v, void* ws, size_t ws_bytes);
    int forward(const StepInputs& in, hipStream_t st);          // loss terms -> metrics()
    int backward(hipStream_t st);                               // fills the flat gradient buffer
    int encode(const StepInputs& in, hipStream_t st);           // slots + attention only (inference path)
    void freeze_weights(bool on) { frozen_ = on; packs_valid_ = false; clear_encode_graphs(); }
    void clear_encode_graphs();
    int encode_backward(const float* dslots, hipStream_t st);   // gradient of the last encode() wrt the encoder's parameters (downstream fine-tuning)
    int generate(hipStream_t st);                               // greedy autoregressive image from the last step's slots
    int clip_adam(const float lr[3], float clip, int step, float gscale, hipStream_t st);
    int grad_norm(hipStream_t st);                              // metrics()[3] = max |g|
    float* metrics() const { return metrics_; }                 // device float[8]: dvae_mse, ce, loss, grad absmax
    int tensor(const char* name, float** ptr, long long* count) const;
    int dropout_mask(unsigned site, long long n, float* out, hipStream_t st) const;
    // writes the soft sample z = softmax(scores) of the last forward into the named tensor "z" (the fused heads keep only the scores;
    // valid between forward and backward -- the backward builds d logits in place of the scores)
    int soft_z(hipStream_t st);
    const SlateConfig cfg;

private:
    float* P(const std::string& n) const;
    float* G(const std::string& n) const;
    float* carve(const char* name, size_t n);
    void layout_workspace(bool commit);
    int lin_fwd(const float* x, int ldx, const float* W, const float* b, float* y, int ldy, long long M, int N, int K, int relu,
                const float* resid, int ldr, float drop_p, unsigned site, hipStream_t st);
    int lin_bwd_x(const float* dy, int ld_dy, const float* W, float* dx, int ldx, long long M, int N_out, int K_in, const float* mask,
                  int ldmask, const float* resid, int ldr, hipStream_t st, Drop dr = Drop(), Xf xf = Xf());
    int lin_bwd_w(const float* dy, int ld_dy, const float* x, int ldx, float* dW, float* db, long long M, int N_out, int K_in,
                  float alpha, hipStream_t st, Drop dr = Drop(), Xf xf = Xf());
    // the Gumbel / cross-entropy soft-max heads live in the vocabulary GEMMs (soft samples; the straight-through `hard` form keeps z)
    bool fused_heads() const { return !cfg.hard; }
    int conv_layer_fwd(const float* x, const float* pack, const float* bias, float* y, int Bn, int Hh, int Ww, int KS, int CIN, int relu,
                       const float* posmap, const float* mask, hipStream_t st);
    int conv_layer_wgrad(const float* x, const float* dy, float* dW, float* db, int Bn, int Hh, int Ww, int KS, int CIN, int cin_real,
                         hipStream_t st);
    int pack_weights(hipStream_t st, bool encoder_only = false);
    int fwd_encoder(const StepInputs& in, hipStream_t st, int fork_dvae = 0);
    int fwd_dvae(const StepInputs& in, hipStream_t st);
    int fwd_decoder(hipStream_t st, bool with_ce = true);
    int dvae_decode(int B, float* drecon, hipStream_t st, const float* zin = nullptr);
    int bwd_decoder(hipStream_t st);
    int bwd_encoder(hipStream_t st, bool fork_dvae = false);
    int bwd_dvae(hipStream_t st);
    int pack_bcdec(hipStream_t st);
    int fwd_bcdec(hipStream_t st);
    int bwd_bcdec(hipStream_t st);

    std::vector<ParamInfo> params_;
    std::map<std::string, int> index_;
    long long flat_size_ = 0;
    long long group_begin_[4] = {0, 0, 0, 0};
    float *p_ = nullptr, *g_ = nullptr, *m_ = nullptr, *v_ = nullptr;
    char* ws_ = nullptr;
    size_t ws_bytes_ = 0, ws_off_ = 0;
    bool ws_commit_ = false;
    std::map<std::string, std::pair<float*, size_t>> named_;
    float* metrics_ = nullptr;

    // dims
    int S, E, T, N, V, d, C, K, I, D, H, NB, NH, DH, Bmax;
    int SH = 1;             // slot-attention heads
    // last step
    StepInputs last_;
    float pdrop_ = 0.f;
    bool have_fwd_ = false;
    int conv_lowlat_ = 0;           // set around the encoder of encode(): small grids take the low-latency convolution
    bool frozen_ = false, packs_valid_ = false;   // freeze_weights(): encode() re-uses the derived weight images (serving: the parameters do not change)
    bool have_enc_ = false;         // the last call was encode(): its activations are what encode_backward() differentiates
    bool enc_only_grads_ = false;   // the gradient buffer holds an encode_backward(): only the encoder tensors have gradients
    bool have_scores_ = false;        // zraw_ holds the Gumbel scores of last_ (not yet overwritten by their gradient)

    // ---- workspace tensors
    float *scratch_ = nullptr;            // transient: split-k slabs, column-sum partials, wgrad slabs
    size_t scratch_floats_ = 0;
    float* scratch2_ = nullptr;           // scratch of the dVAE branch when it runs on the side stream
    int overlap_mode_ = 0;                // OCRL_OVERLAP 0..5 (default 5), described where it is read in SlateModel::bind
    hipStream_t side_ = nullptr;          // dVAE forward / backward overlap the encoder + decoder work (independent branches)
    hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr, ev_tokens_ = nullptr;
    int fork_side(hipStream_t st);
    int join_side(hipStream_t st);
    float *zstat_ = nullptr, *zhstat_ = nullptr, *zlse_ = nullptr, *zdot_ = nullptr, *cestat_ = nullptr, *celse_ = nullptr, *cepart_ = nullptr;
    int* zhidx_ = nullptr;
    float *obs8_, *patches_, *de_[7], *zraw_, *z_, *zdec_;      // zdec_: what the dVAE decoder consumes (z_ or its straight-through form)
    int* tokens_;
    float *dd0_, *dd1_, *dd2_, *dd3_, *dd4_, *ps1_, *dd6_, *dd7_, *dd8_, *dd9_, *ps2_, *recon_, *drecon_;
    float *e1_, *e2_, *e3_, *e4_, *posmap_, *gridT_, *ln0_, *ln0_mean_, *ln0_rstd_, *h1_, *x_;
    float *slots0_, *slot_noise_, *slots_, *attn_, *sa_save_, *sa_wts_, *sa_grows_, *sa_small_, *sa_xchg_;
    float* attn_heads_ = nullptr;         // [B,N,SH*K] per-head attention maps (SH > 1 only)
    float* sa_parts_;
    PackEntry* sa_pack_dev_ = nullptr;
    int sa_pack_n_ = 0, sa_pack_max_ = 0;
    float *cw_fwd_[4], *cw_bwd_[4];       // CNN encoder conv packs
    int conv_x3_ = -1;                    // OCRL_CONV_X3=1 (exploratory): the 5x5 / 64-channel layers on the split-precision bf16 kernel
    std::map<const float*, const float*> x3_of_;      // fp32 pack -> its split-precision pack
    float *dw_fwd_[2], *dw_bwd_[2];       // dVAE decoder 3x3 conv packs
    float *w11p_;                         // [4,64] padded copy of the dVAE output conv
    float *mem_, *emb_;
    struct Blk {
        float *ln1, *ln1_mean, *ln1_rstd, *q, *k, *v, *lse, *ao, *x1;
        float *ln2, *ln2_mean, *ln2_rstd, *cq, *ck, *cv, *cP, *cao, *x2;
        float *ln3, *ln3_mean, *ln3_rstd, *f1, *x3;
        float *xaAb, *xaAbT, *xaVo, *xaVoT;       // folded cross-attention operands per image (xattn.hip): [B,NC,d], [B,d,NC], [B,NC,d], [B,d,NC]
    };
    bool xattn_ = false;                  // OCRL_XATTN (default 1): cross attention in its folded form (one launch per block and direction)
    float *xa_dAb_ = nullptr, *xa_dVo_ = nullptr, *xa_pq_ = nullptr, *xa_po_ = nullptr;
    size_t xa_zero_floats_ = 0;           // the per-block operand buffers form one contiguous region that bind() zeroes (padding columns)
    float* xa_zero_base_ = nullptr;
    std::vector<Blk> blk_;
    // per-block gradient temporaries that a weight-gradient product on the side stream may still be reading while the main stream has
    // moved on: the dropout-backward copies of the residual gradient at the three branch outputs, d ffn-hidden, d cross-attention query,
    // d q|k|v (OCRL_DW_SIDE; without it every block uses the first set)
    struct BlkG { float *gbr[3], *gf1, *gt2, *gqkv, *xaPd, *xaDs; };
    std::vector<BlkG> bg_;
    // inference path of the RL feature extractor (sb3s/ocr_extractor.py:45) at tiny batches: the ~30 launches of encode() are captured once
    // per batch size into a hipGraph and replayed (OCRL_ENCODE_GRAPH=1, opt-in: measured no faster; B <= 32, device RNG).  The observation is staged into a
    // fixed buffer and the seed passed through device memory, so a replay sees new inputs.
    struct EncGraph { hipGraphExec_t exec = nullptr; int warm = 0; };
    std::map<int, EncGraph> enc_graphs_;
    int enc_graph_mode_ = -1;
    hipStream_t cap_ = nullptr;
    float* obs_stage_ = nullptr;
    unsigned long long* seed_dev_ = nullptr;
    int dw_mode_ = 0;                     // OCRL_DW_SIDE: 0 weight gradients of the decoder on the main stream, 1 on the dVAE side stream, 2 on a stream of their own
    hipStream_t side2_ = nullptr;
    float* scratch3_ = nullptr;
    hipEvent_t ev_dw_[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_join2_ = nullptr;
    int ev_dw_next_ = 0;
    float *attn_delta_;                   // [B,h,T] scratch of the attention backward
    float *lnf_, *lnf_mean_, *lnf_rstd_, *pred_;
    // gradient temporaries
    float *gqkv_;                         // [BT, 3d] fused dq|dk|dv
    float *gx_, *gbr_, *gt1_, *gt2_, *gt3_, *gf1_, *gmem_, *gck_, *gcv_, *gslots_, *gslots0_;
    float *gA_, *gB_, *gC_;               // [B*N,64] (CNN encoder / slot-attention input gradients)
    float *gdA_, *gdB_;                   // dVAE decoder gradient ping-pong (up to [B*4T,256])
    float *gmap_;
    float *col0_, *dw0p_;                 // first conv layer: im2col of the observation and the [64, 25*ch (+pad)] gradient product
    // broadcast decoder (use_bcdec)
    float *bc_Wc_, *bc_W1r_, *bc_P1_, *bc_M_, *bc_T_, *bc_c1_, *bc_c2_, *bc_c3_, *bc_out4_, *bc_dout4_, *bc_gA_, *bc_gB_;
    float *bc_pk_[2], *bc_pkb_[2], *bc_Wk4_, *bc_Wb4_, *bc_dW1r_, *bc_dWc_, *bc_dT_, *bc_dM_, *bc_G1_;
};
