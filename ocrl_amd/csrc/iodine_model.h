// IODINE training-step orchestration over the HIP kernels (host code): reference ocrs/iodine/iodine_module.py:79-252 and
// ocrs/base.py:60-74.  One object per process per GPU; not thread-safe; every launch goes to the caller's stream.
#pragma once
#include <map>
#include <string>
#include <vector>

#include "kernels.h"
#include "slate_model.h"      // ParamInfo

struct IodineConfig {
    int obs_size = 64, obs_channels = 3, slot_size = 64, num_iters = 5, num_slots = 7;
    float sigma = 0.35f, beta = 1.f;
    int layer_norm = 1;
    int ref_mlp_hidden = 256;     // refinement MLP / LSTM width; the convolution widths are fixed at 64 (reference configs)
    int max_batch = 1;
};

class IodineModel {
public:
    explicit IodineModel(const IodineConfig& c);
    const std::vector<ParamInfo>& params() const { return params_; }
    long long flat_size() const { return flat_size_; }
    size_t workspace_bytes() const { return ws_bytes_; }
    int bind(float* p, float* g, float* m, float* v, void* ws, size_t ws_bytes);
    // obs [B,3,S,S] NCHW; noise: optional injected N(0,1) draws [I,B,K,L] (else the device RNG stream `seed`)
    int forward(const float* obs, int B, unsigned long long seed, const float* noise, hipStream_t st);
    int backward(hipStream_t st);
    int grad_norm(hipStream_t st);                                       // metrics()[3] = ||g||_2
    int clip_adam(float lr, float clip, int step, float gscale, hipStream_t st);
    float* metrics() const { return metrics_; }                          // device float[8]: loss, mse, kld, grad norm
    int tensor(const char* name, float** ptr, long long* count) const;
    const IodineConfig cfg;

private:
    float* P(const std::string& n) const { return p_ + params_[index_.at(n)].offset; }
    float* G(const std::string& n) const { return g_ + params_[index_.at(n)].offset; }
    float* carve(const char* name, size_t n);
    void layout_workspace(bool commit);
    int pack_weights(hipStream_t st);
    int gemm_nt(const float* x, int ldx, const float* W, int ldw, const float* b, float* y, int ldy, long long M, int N, int K, int act,
                const float* resid, int ldr, hipStream_t st);
    int gemm_nn(const float* dy, int ld_dy, const float* W, int ldw, float* dx, int ldx, long long M, int K_out, int N_in, const float* resid, int ldr,
                hipStream_t st);
    int gemm_tn(const float* dy, int ld_dy, const float* x, int ldx, float* dW, float* db, long long M, int N_out, int K_in, int accumulate,
                hipStream_t st);
    int decoder_fwd(int i, hipStream_t st);
    int decoder_bwd(int i, const float* dout4, bool weights, hipStream_t st);      // -> dslots_
    int refine_fwd(int i, hipStream_t st);
    int refine_bwd(int i, hipStream_t st);                                         // -> denc_, dxin_ (latent part)

    std::vector<ParamInfo> params_;
    std::map<std::string, int> index_;
    long long flat_size_ = 0;
    float *p_ = nullptr, *g_ = nullptr, *m_ = nullptr, *v_ = nullptr;
    char* ws_ = nullptr;
    size_t ws_bytes_ = 0, ws_off_ = 0;
    bool ws_commit_ = false;
    std::map<std::string, std::pair<float*, size_t>> named_;
    float* metrics_ = nullptr;

    int S, N, K, I, L, Hm, Bmax, XW;          // XW = Hm + 4L (LSTM input width)
    int rs_[5];                               // spatial side of the refinement feature maps: S, S/2, ...
    int B_ = 0;
    const float* obs_ = nullptr;
    bool have_fwd_ = false;

    float *scratch_ = nullptr; size_t scratch_floats_ = 0;
    int conv_x3_ = -1;                    // OCRL_CONV_X3=1 (exploratory): the decoder's 3x3 / 64-channel layers on the split-precision bf16 kernels
    float *pk3_[3] = {nullptr, nullptr, nullptr}, *pkb3_[3] = {nullptr, nullptr, nullptr};
    float* parts_ = nullptr;                  // [I][4]: ll, mse, kl sums
    float *st1_, *st2_;
    std::vector<float*> mu_, ls_, eps_, slots_, c_[4], out4_;
    std::vector<float*> enc_, r_[4], pool_, mlpa_, xin_, acts_, cst_, hst_;
    float *zero_state_, *M_, *T_, *P1_, *W1r_, *Wxy_, *pk_[3], *pkb_[3], *Wk4_, *Wb4_, *Wp_[4], *dWp_[4];
    float *col_, *dcol_, *gates_, *dgates_, *dxin_, *dslots_, *gmu_, *gls_, *dh_, *dc_, *dcH_, *dpool_, *da_, *dr_[2], *denc_, *dout4_, *gA_, *gB_;
    float *G1_, *dW1r_, *masks_, *recon_, *rmasked_;
    int ldc_[4];
};
