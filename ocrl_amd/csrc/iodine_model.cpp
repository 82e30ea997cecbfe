// IODINE forward / backward orchestration (reference: ocrs/iodine/iodine_module.py:79-252).
//
// Per iteration i: sample slots -> decoder -> mixture ELBO; for i < I-1 also the in-forward gradients of B*ELBO wrt the
// posterior (a decoder backward-data pass), the 17-channel encoding and the refinement network (LSTM) that updates the
// posterior.  The gradients fed to the refinement network are detached in the reference (:138-143), so the training
// backward is a plain reverse sweep over the saved activations of every iteration.
#include "iodine_model.h"

#include <stdarg.h>
#include <stdlib.h>

#define RC(x)                 \
    do {                      \
        int rc__ = (x);       \
        if (rc__) return rc__; \
    } while (0)

static std::string ifmt(const char* f, ...) {
    char buf[128];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}

IodineModel::IodineModel(const IodineConfig& c) : cfg(c) {
    S = c.obs_size; N = S * S; K = c.num_slots; I = c.num_iters; L = c.slot_size; Hm = c.ref_mlp_hidden; Bmax = c.max_batch;
    XW = Hm + 4 * L;
    rs_[0] = S;
    for (int l = 1; l <= 4; ++l) rs_[l] = (rs_[l - 1] - 1) / 2 + 1;
    ldc_[0] = (9 * 17 + 3) & ~3;
    ldc_[1] = ldc_[2] = ldc_[3] = 9 * 64;
    auto add = [&](const std::string& name, std::vector<int> shp) {
        ParamInfo p;
        p.name = name; p.ndim = (int)shp.size(); p.group = 0; p.numel = 1;
        for (size_t i = 0; i < shp.size(); ++i) { p.shape[i] = shp[i]; p.numel *= shp[i]; }
        params_.push_back(p);
    };
    // order = the reference module's parameters() order (own Parameters first, then refine, then decoder)
    add("slot_mean_init", {1, 1, L}); add("slot_logsig_init", {1, 1, L}); add("slot_init", {1, 1, L});
    for (int l = 0; l < 4; ++l) { add(ifmt("refine.mlc.layers.%d.weight", l), {64, l ? 64 : 17, 3, 3}); add(ifmt("refine.mlc.layers.%d.bias", l), {64}); }
    add("refine.mlp.layers.0.weight", {Hm, 64}); add("refine.mlp.layers.0.bias", {Hm});
    add("refine.lstm.weight_ih", {4 * Hm, XW}); add("refine.lstm.weight_hh", {4 * Hm, Hm});
    add("refine.lstm.bias_ih", {4 * Hm}); add("refine.lstm.bias_hh", {4 * Hm});
    add("refine.mean_update.weight", {L, Hm}); add("refine.mean_update.bias", {L});
    add("refine.logsig_update.weight", {L, Hm}); add("refine.logsig_update.bias", {L});
    for (int l = 0; l < 4; ++l) { add(ifmt("decoder.mlc.layers.%d.weight", l), {64, l ? 64 : L + 2, 3, 3}); add(ifmt("decoder.mlc.layers.%d.bias", l), {64}); }
    add("decoder.conv.weight", {4, 64, 3, 3}); add("decoder.conv.bias", {4});
    long long off = 0;
    for (size_t i = 0; i < params_.size(); ++i) {
        params_[i].offset = off;
        index_[params_[i].name] = (int)i;
        off += (params_[i].numel + 3) & ~3ll;
    }
    flat_size_ = off;
    ws_ = nullptr;
    layout_workspace(false);
}

float* IodineModel::carve(const char* name, size_t n) {
    const size_t bytes = (n * 4 + 255) & ~(size_t)255;
    float* p = reinterpret_cast<float*>(ws_ + ws_off_);
    ws_off_ += bytes;
    if (ws_commit_ && name) named_[name] = std::make_pair(p, n);
    return p;
}

void IodineModel::layout_workspace(bool commit) {
    ws_commit_ = commit;
    ws_off_ = 0;
    const size_t BK = (size_t)Bmax * K, BKN = BK * N;
    metrics_ = carve("metrics", 64);
    parts_ = carve("parts", (size_t)I * 4 + 4);
    st1_ = carve(nullptr, BK * 4); st2_ = carve(nullptr, BK * 4);
    {
        size_t need = conv_wgrad_ws_floats((int)BK, S, S, 3, 64) + (size_t)512 * 2304 * 2;
        const size_t sk = (size_t)1024 * 64 * 576 / 4 + (size_t)4 * Hm * XW * 8 + (1 << 20);
        if (sk > need) need = sk;
        const size_t cs = (size_t)N * 64 * 4 + (1 << 20);
        if (cs > need) need = cs;
        const size_t rp = BK * (size_t)S * 3 * 64;        // row partials of the first decoder layer's backward
        if (rp > need) need = rp;
        scratch_floats_ = need + (1 << 20);
    }
    scratch_ = carve(nullptr, scratch_floats_);
    mu_.assign(I, nullptr); ls_.assign(I, nullptr); eps_.assign(I, nullptr); slots_.assign(I, nullptr); out4_.assign(I, nullptr);
    for (int l = 0; l < 4; ++l) c_[l].assign(I, nullptr);
    for (int i = 0; i < I; ++i) {
        mu_[i] = carve(nullptr, BK * L); ls_[i] = carve(nullptr, BK * L); eps_[i] = carve(nullptr, BK * L);
        slots_[i] = carve(i == I - 1 ? "slots" : nullptr, BK * L);
        for (int l = 0; l < 4; ++l) c_[l][i] = carve(nullptr, BKN * 64);
        out4_[i] = carve(i == I - 1 ? "out4" : nullptr, BKN * 4);
    }
    const int R = I > 1 ? I - 1 : 0;
    enc_.assign(R, nullptr); pool_.assign(R, nullptr); mlpa_.assign(R, nullptr); xin_.assign(R, nullptr); acts_.assign(R, nullptr);
    cst_.assign(R, nullptr); hst_.assign(R, nullptr);
    for (int l = 0; l < 4; ++l) r_[l].assign(R, nullptr);
    for (int i = 0; i < R; ++i) {
        enc_[i] = carve(i == 0 ? "enc0" : nullptr, BKN * 17);
        for (int l = 0; l < 4; ++l) r_[l][i] = carve(nullptr, BK * rs_[l + 1] * rs_[l + 1] * 64);
        pool_[i] = carve(nullptr, BK * 64); mlpa_[i] = carve(nullptr, BK * Hm); xin_[i] = carve(i == 0 ? "xin0" : nullptr, BK * XW);
        acts_[i] = carve(nullptr, BK * 4 * Hm); cst_[i] = carve(nullptr, BK * Hm); hst_[i] = carve(nullptr, BK * Hm);
    }
    zero_state_ = carve(nullptr, BK * Hm);
    M_ = carve(nullptr, BK * 576); T_ = carve(nullptr, BK * 576);
    P1_ = carve(nullptr, (size_t)N * 64); W1r_ = carve(nullptr, (size_t)576 * L); Wxy_ = carve(nullptr, 576 * 2);
    for (int l = 0; l < 3; ++l) { pk_[l] = carve(nullptr, 9 * 64 * 64); pkb_[l] = carve(nullptr, 9 * 64 * 64); }
    if (conv_x3_ < 0) { const char* e = getenv("OCRL_CONV_X3"); conv_x3_ = e ? atoi(e) : 0; }
    if (conv_x3_ > 0)
        for (int l = 0; l < 3; ++l) { pk3_[l] = carve(nullptr, conv_x3_pack_floats(3)); pkb3_[l] = carve(nullptr, conv_x3_pack_floats(3)); }
    Wk4_ = carve(nullptr, 9 * 64 * 4); Wb4_ = carve(nullptr, 9 * 4 * 64);
    size_t colmax = 0;
    for (int l = 0; l < 4; ++l) {
        Wp_[l] = carve(nullptr, (size_t)64 * ldc_[l]); dWp_[l] = carve(nullptr, (size_t)64 * ldc_[l]);
        const size_t c = BK * rs_[l + 1] * rs_[l + 1] * ldc_[l];
        if (c > colmax) colmax = c;
    }
    col_ = carve(nullptr, colmax); dcol_ = carve(nullptr, colmax);
    gates_ = carve(nullptr, BK * 4 * Hm); dgates_ = carve(nullptr, BK * 4 * Hm); dxin_ = carve(nullptr, BK * XW);
    dslots_ = carve(nullptr, BK * L); gmu_ = carve(nullptr, BK * L); gls_ = carve(nullptr, BK * L);
    dh_ = carve(nullptr, BK * Hm); dc_ = carve(nullptr, BK * Hm); dcH_ = carve(nullptr, BK * Hm);
    dpool_ = carve(nullptr, BK * 64); da_ = carve(nullptr, BK * Hm);
    dr_[0] = carve(nullptr, BK * rs_[1] * rs_[1] * 64); dr_[1] = carve(nullptr, BK * rs_[1] * rs_[1] * 64);
    denc_ = carve(nullptr, BKN * 17); dout4_ = carve(nullptr, BKN * 4); gA_ = carve(nullptr, BKN * 64); gB_ = carve(nullptr, BKN * 64);
    G1_ = carve(nullptr, (size_t)N * 64); dW1r_ = carve(nullptr, (size_t)576 * L);
    masks_ = carve("masks", BKN); recon_ = carve("recon", (size_t)Bmax * 3 * N); rmasked_ = carve("recons_masked", BKN * 3);
    if (!commit) ws_bytes_ = ws_off_ + 4096;
}

int IodineModel::bind(float* p, float* g, float* m, float* v, void* ws, size_t ws_bytes) {
    OCRL_REQUIRE(p && g && ws, "bind: null buffer");
    OCRL_REQUIRE(ws_bytes >= ws_bytes_, "bind: workspace too small (%zu < %zu)", ws_bytes, ws_bytes_);
    OCRL_REQUIRE(((uintptr_t)p & 255) == 0 && ((uintptr_t)g & 255) == 0 && ((uintptr_t)ws & 255) == 0, "bind: buffers must be 256-byte aligned");
    OCRL_REQUIRE(cfg.obs_channels == 3 && S % 16 == 0 && S >= 16 && L % 4 == 0 && L >= 4 && L <= 256 && Hm % 64 == 0 && K >= 1 && K <= 16 && I >= 1,
                 "iodine: unsupported configuration (obs_size %% 16, slot_size %% 4, mlp hidden %% 64, 1..16 slots)");
    p_ = p; g_ = g; m_ = m; v_ = v;
    ws_ = static_cast<char*>(ws);
    named_.clear();
    layout_workspace(true);
    RC(fill_launch(zero_state_, (long long)Bmax * K * Hm, 0.f, 0));
    OCRL_HIP(hipDeviceSynchronize());
    have_fwd_ = false;
    return 0;
}

int IodineModel::tensor(const char* name, float** ptr, long long* count) const {
    auto it = named_.find(name);
    if (it == named_.end()) {
        auto pi = index_.find(name);
        OCRL_REQUIRE(pi != index_.end(), "tensor: unknown name '%s'", name);
        *ptr = p_ + params_[pi->second].offset;
        *count = params_[pi->second].numel;
        return 0;
    }
    *ptr = it->second.first;
    *count = (long long)it->second.second;
    return 0;
}

// y[M,N] = act(x[M,K] W[N,K]^T + b) + resid
int IodineModel::gemm_nt(const float* x, int ldx, const float* W, int ldw, const float* b, float* y, int ldy, long long M, int Nn, int Kk, int act,
                         const float* resid, int ldr, hipStream_t st) {
    GemmArgs a;
    a.A = x; a.B = W; a.C = y; a.M = (int)M; a.N = Nn; a.K = Kk; a.lda = ldx; a.ldb = ldw; a.ldc = ldy; a.akc = 1; a.bkc = 1;
    a.bias = b; a.relu = act; a.resid = resid; a.ldr = ldr;
    return gemm_launch(a, st);
}
// dx[M,N_in] = dy[M,K_out] W[K_out,N_in] + resid
int IodineModel::gemm_nn(const float* dy, int ld_dy, const float* W, int ldw, float* dx, int ldx, long long M, int K_out, int N_in, const float* resid,
                         int ldr, hipStream_t st) {
    GemmArgs a;
    a.A = dy; a.B = W; a.C = dx; a.M = (int)M; a.N = N_in; a.K = K_out; a.lda = ld_dy; a.ldb = ldw; a.ldc = ldx; a.akc = 1; a.bkc = 0;
    a.resid = resid; a.ldr = ldr;
    return gemm_launch(a, st);
}
// dW[N_out,K_in] (+)= dy[M,N_out]^T x[M,K_in];  db[N_out] (+)= column sums of dy
int IodineModel::gemm_tn(const float* dy, int ld_dy, const float* x, int ldx, float* dW, float* db, long long M, int N_out, int K_in, int accumulate,
                         hipStream_t st) {
    GemmArgs a;
    a.A = dy; a.B = x; a.C = dW; a.M = N_out; a.N = K_in; a.K = (int)M; a.lda = ld_dy; a.ldb = ldx; a.ldc = K_in; a.akc = 0; a.bkc = 0;
    const int tiles = cdiv(N_out, 128) * cdiv(K_in, (K_in % 128 == 0) ? 128 : 64);
    long long splits = 1024 / tiles;
    if (splits > M / 256) splits = M / 256;
    if (splits < 1) splits = 1;
    const long long slab = (long long)N_out * K_in;
    if (splits * slab > (long long)scratch_floats_) splits = (long long)scratch_floats_ / slab;
    if (splits > 1) {
        a.splitk = (int)splits; a.C = scratch_; a.sCsplit = slab;
        RC(gemm_launch(a, st));
        RC(splitk_reduce_launch(scratch_, dW, slab, (int)splits, slab, accumulate, st));
    } else {
        if (accumulate) { a.resid = dW; a.ldr = K_in; }
        RC(gemm_launch(a, st));
    }
    if (db) RC(colsum_launch(dy, ld_dy, db, M, N_out, accumulate, 1.f, scratch_, scratch_floats_, st));
    return 0;
}

int IodineModel::pack_weights(hipStream_t st) {
    RC(io_w1_pack_launch(P("decoder.mlc.layers.0.weight"), W1r_, Wxy_, L, st));
    RC(io_p1_launch(Wxy_, P("decoder.mlc.layers.0.bias"), P1_, S, st));
    for (int l = 0; l < 3; ++l) RC(conv_pack_launch(P(ifmt("decoder.mlc.layers.%d.weight", l + 1)), pk_[l], pkb_[l], 3, 64, 64, 64, st));
    if (conv_x3_ > 0)
        for (int l = 0; l < 3; ++l) RC(conv_pack_x3_launch(P(ifmt("decoder.mlc.layers.%d.weight", l + 1)), pk3_[l], pkb3_[l], st, 3));
    RC(bc_c4_pack_launch(P("decoder.conv.weight"), Wk4_, Wb4_, 4, st));
    for (int l = 0; l < 4; ++l) RC(io_refw_pack_launch(P(ifmt("refine.mlc.layers.%d.weight", l)), Wp_[l], l ? 64 : 17, ldc_[l], st));
    return 0;
}

int IodineModel::decoder_fwd(int i, hipStream_t st) {
    const long long BK = (long long)B_ * K;
    RC(gemm_nt(slots_[i], L, W1r_, L, nullptr, M_, 576, BK, 576, L, 0, nullptr, 0, st));          // M[bk][tap][co] = W_tap s
    RC(io_class_sum_launch(M_, T_, BK, 1, st));
    RC(io_layer1_launch(P1_, T_, c_[0][i], BK, S, st));
    for (int l = 0; l < 3; ++l) {
        ConvArgs a;
        a.X = c_[l][i]; a.Wp = pk_[l]; a.Y = c_[l + 1][i]; a.B = (int)BK; a.H = S; a.W = S; a.relu = 2;
        a.bias = P(ifmt("decoder.mlc.layers.%d.bias", l + 1));
        if (conv_x3_ > 0) RC(conv_x3_launch(a, pk3_[l], st, 3)); else
        RC(conv_fwd_launch(a, 3, 64, 64, st));
    }
    RC(bc_c4_fwd_launch(c_[3][i], Wk4_, P("decoder.conv.bias"), out4_[i], (int)BK, S, st));
    return 0;
}

// dout4 -> dslots_; with `weights` also the decoder weight gradients (accumulated over the iterations)
int IodineModel::decoder_bwd(int i, const float* dout4, bool weights, hipStream_t st) {
    const long long BK = (long long)B_ * K, BKN = BK * N;
    if (weights) {
        const int nb = bc_c4_wgrad_blocks((int)BK, S);
        RC(bc_c4_wgrad_launch(c_[3][i], dout4, scratch_, (int)BK, S, st));
        RC(colsum_launch(scratch_, 2304, G("decoder.conv.weight"), nb, 2304, 1, 1.f, scratch_ + (size_t)nb * 2304, scratch_floats_ - (size_t)nb * 2304, st));
        RC(colsum_launch(dout4, 4, G("decoder.conv.bias"), BKN, 4, 1, 1.f, scratch_, scratch_floats_, st));
    }
    RC(bc_c4_bwd_data_launch(dout4, Wb4_, c_[3][i], gA_, (int)BK, S, st, 1));                        // gA = d pre-activation of layer 3
    float* cur = gA_;
    float* nxt = gB_;
    for (int l = 2; l >= 0; --l) {
        if (weights) {
            WgradArgs w;
            w.X = c_[l][i]; w.dY = cur; w.part = scratch_; w.B = (int)BK; w.H = S; w.W = S;
            OCRL_REQUIRE(conv_wgrad_ws_floats((int)BK, S, S, 3, 64) <= scratch_floats_, "conv wgrad: scratch too small");
            RC(conv_wgrad_launch(w, 3, 64, 64, 64, G(ifmt("decoder.mlc.layers.%d.weight", l + 1)), 1, st, conv_x3_ > 0 ? 1 : 0));
            RC(colsum_launch(cur, 64, G(ifmt("decoder.mlc.layers.%d.bias", l + 1)), BKN, 64, 1, 1.f, scratch_, scratch_floats_, st));
        }
        ConvArgs a;
        a.X = cur; a.Wp = pkb_[l]; a.Y = nxt; a.B = (int)BK; a.H = S; a.W = S; a.mask = c_[l][i]; a.mask_elu = 1;
        if (conv_x3_ > 0) RC(conv_x3_launch(a, pkb3_[l], st, 3)); else
        RC(conv_fwd_launch(a, 3, 64, 64, st));                                                        // d pre-activation of layer l
        float* t = cur; cur = nxt; nxt = t;
    }
    // first layer through the broadcast shortcut (cur = d pre-activation of layer 0)
    RC(io_layer1_bwd_launch(cur, T_, BK, S, scratch_, scratch_floats_, st));
    RC(io_class_sum_launch(T_, M_, BK, 0, st));
    RC(gemm_nn(M_, 576, W1r_, L, dslots_, L, BK, 576, L, nullptr, 0, st));
    if (weights) {
        RC(gemm_tn(M_, 576, slots_[i], L, dW1r_, nullptr, BK, 576, L, 1, st));
        RC(colsum_launch(cur, (long long)N * 64, G1_, BK, N * 64, 1, 1.f, scratch_, scratch_floats_, st));
    }
    return 0;
}

int IodineModel::refine_fwd(int i, hipStream_t st) {
    const long long BK = (long long)B_ * K;
    const float* x = enc_[i];
    for (int l = 0; l < 4; ++l) {
        const int C = l ? 64 : 17, so = rs_[l + 1];
        RC(io_im2col_launch(x, col_, BK, C, rs_[l], rs_[l], ldc_[l], st));
        RC(gemm_nt(col_, ldc_[l], Wp_[l], ldc_[l], P(ifmt("refine.mlc.layers.%d.bias", l)), r_[l][i], 64, BK * so * so, 64, ldc_[l], 2, nullptr, 0, st));
        x = r_[l][i];
    }
    RC(io_pool_launch(r_[3][i], pool_[i], BK, rs_[4] * rs_[4], st));
    RC(gemm_nt(pool_[i], 64, P("refine.mlp.layers.0.weight"), 64, P("refine.mlp.layers.0.bias"), mlpa_[i], Hm, BK, Hm, 64, 0, nullptr, 0, st));
    RC(io_elu2_launch(mlpa_[i], Hm, xin_[i], XW, BK, Hm, nullptr, 0, st));
    const float* hp = i ? hst_[i - 1] : zero_state_;
    const float* cp = i ? cst_[i - 1] : zero_state_;
    RC(gemm_nt(xin_[i], XW, P("refine.lstm.weight_ih"), XW, P("refine.lstm.bias_ih"), gates_, 4 * Hm, BK, 4 * Hm, XW, 0, nullptr, 0, st));
    RC(gemm_nt(hp, Hm, P("refine.lstm.weight_hh"), Hm, P("refine.lstm.bias_hh"), gates_, 4 * Hm, BK, 4 * Hm, Hm, 0, gates_, 4 * Hm, st));
    RC(io_lstm_fwd_launch(gates_, cp, acts_[i], cst_[i], hst_[i], BK, Hm, st));
    // the reference binds the LSTMCell outputs as (c, h): the update heads read the CELL state (iodine_module.py:418-422)
    RC(gemm_nt(cst_[i], Hm, P("refine.mean_update.weight"), Hm, P("refine.mean_update.bias"), mu_[i + 1], L, BK, L, Hm, 0, mu_[i], L, st));
    RC(gemm_nt(cst_[i], Hm, P("refine.logsig_update.weight"), Hm, P("refine.logsig_update.bias"), ls_[i + 1], L, BK, L, Hm, 0, ls_[i], L, st));
    return 0;
}

int IodineModel::forward(const float* obs, int B, unsigned long long seed, const float* noise, hipStream_t st) {
    OCRL_REQUIRE(p_ && ws_, "forward: buffers not bound");
    OCRL_REQUIRE(obs && B >= 1 && B <= Bmax, "forward: bad batch %d (max %d)", B, Bmax);
    B_ = B; obs_ = obs;
    const long long BK = (long long)B * K;
    RC(pack_weights(st));
    RC(fill_launch(parts_, I * 4, 0.f, st));
    // posterior initialisation: every (image, slot) row starts from the shared init vectors
    {
        GemmArgs a;      // rows of ones would be a GEMM; a strided copy is simpler: use pad_cols with ldi = 0 (broadcast row)
        RC(pad_cols_launch(P("slot_mean_init"), 0, mu_[0], L, BK, L, L, st));
        RC(pad_cols_launch(P("slot_logsig_init"), 0, ls_[0], L, BK, L, L, st));
        (void)a;
    }
    for (int i = 0; i < I; ++i) {
        const bool last = i == I - 1;
        RC(io_sample_launch(mu_[i], ls_[i], noise ? noise + (size_t)i * BK * L : nullptr, eps_[i], slots_[i], parts_ + i * 4 + 2, BK * L, seed,
                            300u + (unsigned)i, scratch_, scratch_floats_, st));
        RC(decoder_fwd(i, st));
        RC(io_elbo_launch(out4_[i], obs, B, K, S, cfg.sigma, last ? nullptr : enc_[i], st1_, last ? nullptr : dout4_, parts_ + i * 4,
                          last ? masks_ : nullptr, last ? recon_ : nullptr, last ? rmasked_ : nullptr, scratch_, scratch_floats_, st));
        if (last) break;
        if (cfg.layer_norm) {
            RC(io_enc_norm_launch(enc_[i], st1_, st2_, BK, N, scratch_, scratch_floats_, st));
        }
        RC(decoder_bwd(i, dout4_, false, st));                                  // d(B*elbo)/d slots
        RC(io_latent_launch(mu_[i], ls_[i], eps_[i], dslots_, xin_[i] + Hm, BK, L, cfg.beta, cfg.layer_norm, XW, st));
        RC(refine_fwd(i, st));
    }
    RC(io_loss_launch(parts_, metrics_, I, B, cfg.beta, st));
    have_fwd_ = true;
    return 0;
}

// refinement step i backward: consumes gmu_/gls_ (gradients wrt the posterior of iteration i+1) and the carried LSTM state
// gradients; leaves denc_ (gradient of the encoding) and dxin_ (its latent columns feed the posterior gradient)
int IodineModel::refine_bwd(int i, hipStream_t st) {
    const long long BK = (long long)B_ * K;
    const bool top = i == I - 2;                  // the last refinement step has no successor: no carried state gradients
    // heads: mu[i+1] = mu[i] + c W_m^T + b_m (same for logsig)
    RC(gemm_nn(gmu_, L, P("refine.mean_update.weight"), Hm, dcH_, Hm, BK, L, Hm, top ? nullptr : dc_, Hm, st));
    RC(gemm_nn(gls_, L, P("refine.logsig_update.weight"), Hm, dcH_, Hm, BK, L, Hm, dcH_, Hm, st));
    RC(gemm_tn(gmu_, L, cst_[i], Hm, G("refine.mean_update.weight"), G("refine.mean_update.bias"), BK, L, Hm, 1, st));
    RC(gemm_tn(gls_, L, cst_[i], Hm, G("refine.logsig_update.weight"), G("refine.logsig_update.bias"), BK, L, Hm, 1, st));
    const float* hp = i ? hst_[i - 1] : zero_state_;
    const float* cp = i ? cst_[i - 1] : zero_state_;
    RC(io_lstm_bwd_launch(acts_[i], cp, cst_[i], top ? nullptr : dh_, dcH_, dgates_, dc_, BK, Hm, st));     // dc_ = gradient wrt c_{i-1}
    RC(gemm_nn(dgates_, 4 * Hm, P("refine.lstm.weight_ih"), XW, dxin_, XW, BK, 4 * Hm, XW, nullptr, 0, st));
    RC(gemm_nn(dgates_, 4 * Hm, P("refine.lstm.weight_hh"), Hm, dh_, Hm, BK, 4 * Hm, Hm, nullptr, 0, st));  // dh_ = gradient wrt h_{i-1}
    RC(gemm_tn(dgates_, 4 * Hm, xin_[i], XW, G("refine.lstm.weight_ih"), G("refine.lstm.bias_ih"), BK, 4 * Hm, XW, 1, st));
    RC(gemm_tn(dgates_, 4 * Hm, hp, Hm, G("refine.lstm.weight_hh"), G("refine.lstm.bias_hh"), BK, 4 * Hm, Hm, 1, st));
    // MLP with the double ELU
    RC(io_elu2_launch(mlpa_[i], Hm, da_, Hm, BK, Hm, dxin_, XW, st));
    RC(gemm_nn(da_, Hm, P("refine.mlp.layers.0.weight"), 64, dpool_, 64, BK, Hm, 64, nullptr, 0, st));
    RC(gemm_tn(da_, Hm, pool_[i], 64, G("refine.mlp.layers.0.weight"), G("refine.mlp.layers.0.bias"), BK, Hm, 64, 1, st));
    // stride-2 convolutions, last to first
    float* dpre = dr_[0];
    float* other = dr_[1];
    RC(io_pool_bwd_launch(dpool_, r_[3][i], dpre, BK, rs_[4] * rs_[4], st));
    for (int l = 3; l >= 0; --l) {
        const int C = l ? 64 : 17, so = rs_[l + 1];
        const long long rows = BK * so * so;
        const float* x = l ? r_[l - 1][i] : enc_[i];
        RC(io_im2col_launch(x, col_, BK, C, rs_[l], rs_[l], ldc_[l], st));
        RC(gemm_tn(dpre, 64, col_, ldc_[l], dWp_[l], G(ifmt("refine.mlc.layers.%d.bias", l)), rows, 64, ldc_[l], 1, st));
        RC(gemm_nn(dpre, 64, Wp_[l], ldc_[l], dcol_, ldc_[l], rows, 64, ldc_[l], nullptr, 0, st));
        if (l) {
            RC(io_col2im_launch(dcol_, r_[l - 1][i], other, BK, 64, rs_[l], rs_[l], ldc_[l], st));
            float* t = dpre; dpre = other; other = t;
        } else {
            RC(io_col2im_launch(dcol_, nullptr, denc_, BK, 17, S, S, ldc_[0], st));
        }
    }
    return 0;
}

int IodineModel::backward(hipStream_t st) {
    OCRL_REQUIRE(have_fwd_, "backward: no forward pass to differentiate");
    const long long BK = (long long)B_ * K;
    RC(fill_launch(g_, flat_size_, 0.f, st));
    RC(fill_launch(gmu_, BK * L, 0.f, st));
    RC(fill_launch(gls_, BK * L, 0.f, st));
    RC(fill_launch(G1_, (long long)N * 64, 0.f, st));
    RC(fill_launch(dW1r_, 576ll * L, 0.f, st));
    for (int l = 0; l < 4; ++l) RC(fill_launch(dWp_[l], 64ll * ldc_[l], 0.f, st));
    for (int i = I - 1; i >= 0; --i) {
        const bool last = i == I - 1;
        const float w = (float)(i + 1) / (float)I;
        if (!last) RC(refine_bwd(i, st));
        RC(io_elbo_bwd_launch(out4_[i], obs_, last ? nullptr : denc_, B_, K, S, cfg.sigma, -w / (float)B_, dout4_, st));
        RC(decoder_bwd(i, dout4_, true, st));
        RC(io_post_grad_launch(mu_[i], ls_[i], eps_[i], dslots_, last ? nullptr : dxin_ + Hm, XW, w * cfg.beta / (float)B_, gmu_, gls_, BK, L, st));
    }
    RC(colsum_launch(gmu_, L, G("slot_mean_init"), BK, L, 0, 1.f, scratch_, scratch_floats_, st));
    RC(colsum_launch(gls_, L, G("slot_logsig_init"), BK, L, 0, 1.f, scratch_, scratch_floats_, st));
    RC(io_w1_grad_launch(dW1r_, G1_, G("decoder.mlc.layers.0.weight"), G("decoder.mlc.layers.0.bias"), S, L, st));
    for (int l = 0; l < 4; ++l) RC(io_refw_unpack_launch(dWp_[l], G(ifmt("refine.mlc.layers.%d.weight", l)), l ? 64 : 17, ldc_[l], st));
    have_fwd_ = false;
    return 0;
}

int IodineModel::grad_norm(hipStream_t st) { return io_l2norm_launch(g_, flat_size_, metrics_ + 3, scratch_, scratch_floats_, st); }

// clip_grad_norm_(params, clip, 2.0) + Adam over the one parameter group (ocrs/base.py:60-74).  `slot_init` never receives a
// gradient in the reference, so torch's Adam skips it: the update covers the two ranges around it.
int IodineModel::clip_adam(float lr, float clip, int step, float gscale, hipStream_t st) {
    OCRL_REQUIRE(m_ && v_, "clip_adam: optimiser state not bound");
    RC(grad_norm(st));
    const ParamInfo& skip = params_[index_.at("slot_init")];
    const long long a1 = skip.offset, b0 = skip.offset + ((skip.numel + 3) & ~3ll);
    RC(clip_adam_launch(p_, g_, m_, v_, a1, metrics_ + 3, clip, lr, 0.9f, 0.999f, 1e-8f, step, gscale, st));
    RC(clip_adam_launch(p_ + b0, g_ + b0, m_ + b0, v_ + b0, flat_size_ - b0, metrics_ + 3, clip, lr, 0.9f, 0.999f, 1e-8f, step, gscale, st));
    return 0;
}
