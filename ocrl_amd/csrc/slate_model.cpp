// SLATE training step on the HIP kernels: parameter table, workspace, forward, backward, optimiser.
// Follows the reference's SLATE_Module.get_loss (ocrs/slate/slate_module.py:198-241) and
// Base.update (ocrs/base.py:60-74); maths restated in SURVEY.md Appendix A.
#include "slate_model.h"

#include <stdlib.h>
#include <functional>
#include <utility>

#include <math.h>
#include <stdarg.h>
#include <string.h>

#define RC(x)                 \
    do {                      \
        int rc__ = (x);       \
        if (rc__) return rc__; \
    } while (0)

static std::string fmt(const char* f, ...) {
    char buf[256];
    va_list ap;
    va_start(ap, f);
    vsnprintf(buf, sizeof buf, f, ap);
    va_end(ap);
    return buf;
}

// ---------------------------------------------------------------------------------------------
SlateModel::SlateModel(const SlateConfig& c) : cfg(c) {
    S = c.obs_size; E = S / 4; T = E * E; N = S * S; V = c.vocab; d = c.d_model; C = c.cnn_hidden;
    K = c.num_slots; I = c.num_iters; D = c.slot_size; H = c.mlp_hidden; NB = c.num_blocks; NH = c.num_heads;
    DH = d / NH; Bmax = c.max_batch; SH = c.slot_heads > 0 ? c.slot_heads : 1;
    const int ch = c.obs_channels;
    auto add = [&](const std::string& name, std::vector<int> shp, int g) {
        ParamInfo p;
        p.name = name; p.ndim = (int)shp.size(); p.group = g; p.numel = 1;
        for (size_t i = 0; i < shp.size(); ++i) { p.shape[i] = shp[i]; p.numel *= shp[i]; }
        params_.push_back(p);
    };
    // group 0: dVAE (ocrs/common/models.py:10-37) — order = reference module.parameters() order
    add("_dvae._encoder.0.m.weight", {64, ch, 4, 4}, 0); add("_dvae._encoder.0.m.bias", {64}, 0);
    for (int i = 1; i < 7; ++i) { add(fmt("_dvae._encoder.%d.m.weight", i), {64, 64, 1, 1}, 0); add(fmt("_dvae._encoder.%d.m.bias", i), {64}, 0); }
    add("_dvae._encoder.7.weight", {V, 64, 1, 1}, 0); add("_dvae._encoder.7.bias", {V}, 0);
    const int di[9] = {0, 1, 2, 3, 4, 6, 7, 8, 9};
    const int dshape[9][4] = {{64, V, 1, 1}, {64, 64, 3, 3}, {64, 64, 1, 1}, {64, 64, 1, 1}, {256, 64, 1, 1},
                              {64, 64, 3, 3}, {64, 64, 1, 1}, {64, 64, 1, 1}, {256, 64, 1, 1}};
    for (int i = 0; i < 9; ++i) {
        add(fmt("_dvae._decoder.%d.m.weight", di[i]), {dshape[i][0], dshape[i][1], dshape[i][2], dshape[i][3]}, 0);
        add(fmt("_dvae._decoder.%d.m.bias", di[i]), {dshape[i][0]}, 0);
    }
    add("_dvae._decoder.11.weight", {ch, 64, 1, 1}, 0); add("_dvae._decoder.11.bias", {ch}, 0);
    // group 1
    add("_enc._encoder.0.m.weight", {C, ch, 5, 5}, 1); add("_enc._encoder.0.m.bias", {C}, 1);
    for (int i = 1; i < 3; ++i) { add(fmt("_enc._encoder.%d.m.weight", i), {C, C, 5, 5}, 1); add(fmt("_enc._encoder.%d.m.bias", i), {C}, 1); }
    add("_enc._encoder.3.weight", {C, C, 5, 5}, 1); add("_enc._encoder.3.bias", {C}, 1);
    add("_enc_pos.channels_map.weight", {C, 4, 1, 1}, 1); add("_enc_pos.channels_map.bias", {C}, 1);
    add("_slotattn.slot_mu", {1, 1, D}, 1); add("_slotattn.slot_log_sigma", {1, 1, D}, 1);
    add("_slotattn.layer_norm.weight", {C}, 1); add("_slotattn.layer_norm.bias", {C}, 1);
    add("_slotattn.mlp.0.weight", {C, C}, 1); add("_slotattn.mlp.0.bias", {C}, 1);
    add("_slotattn.mlp.2.weight", {C, C}, 1); add("_slotattn.mlp.2.bias", {C}, 1);
    const std::string sa = "_slotattn.slot_attention.";
    add(sa + "norm_inputs.weight", {C}, 1); add(sa + "norm_inputs.bias", {C}, 1);
    add(sa + "norm_slots.weight", {D}, 1); add(sa + "norm_slots.bias", {D}, 1);
    add(sa + "norm_mlp.weight", {D}, 1); add(sa + "norm_mlp.bias", {D}, 1);
    add(sa + "project_q.weight", {D, D}, 1); add(sa + "project_k.weight", {D, C}, 1); add(sa + "project_v.weight", {D, C}, 1);
    add(sa + "gru.weight_ih", {3 * D, D}, 1); add(sa + "gru.weight_hh", {3 * D, D}, 1);
    add(sa + "gru.bias_ih", {3 * D}, 1); add(sa + "gru.bias_hh", {3 * D}, 1);
    add(sa + "mlp.0.weight", {H, D}, 1); add(sa + "mlp.0.bias", {H}, 1);
    add(sa + "mlp.2.weight", {D, H}, 1); add(sa + "mlp.2.bias", {D}, 1);
    add("_slotproj.weight", {d, D}, 1);
    if (c.use_bcdec) {      // ocrs/common/models.py:110-126, appended to the slot-attention group (slate_module.py:96-103)
        add("_dec._decoder.0.m.weight", {C, D, 5, 5}, 1); add("_dec._decoder.0.m.bias", {C}, 1);
        for (int i = 1; i < 3; ++i) { add(fmt("_dec._decoder.%d.m.weight", i), {C, C, 5, 5}, 1); add(fmt("_dec._decoder.%d.m.bias", i), {C}, 1); }
        add("_dec._decoder.3.weight", {ch + 1, C, 3, 3}, 1); add("_dec._decoder.3.bias", {ch + 1}, 1);
        add("_dec._pos_emb.channels_map.weight", {D, 4, 1, 1}, 1); add("_dec._pos_emb.channels_map.bias", {D}, 1);
    }
    // group 2
    add("_dict.dictionary.weight", {V, d}, 2);
    add("_bos_token._bos_token", {1, 1, d}, 2);
    add("_z_pos.pe", {1, 1 + T, d}, 2);
    for (int b = 0; b < NB; ++b) {
        const std::string p = fmt("_tfdec.blocks.%d.", b);
        add(p + "self_attn_layer_norm.weight", {d}, 2); add(p + "self_attn_layer_norm.bias", {d}, 2);
        for (const char* q : {"q", "k", "v", "o"}) add(p + "self_attn.proj_" + q + ".weight", {d, d}, 2);
        add(p + "encoder_decoder_attn_layer_norm.weight", {d}, 2); add(p + "encoder_decoder_attn_layer_norm.bias", {d}, 2);
        for (const char* q : {"q", "k", "v", "o"}) add(p + "encoder_decoder_attn.proj_" + q + ".weight", {d, d}, 2);
        add(p + "ffn_layer_norm.weight", {d}, 2); add(p + "ffn_layer_norm.bias", {d}, 2);
        add(p + "ffn.0.weight", {4 * d, d}, 2); add(p + "ffn.0.bias", {4 * d}, 2);
        add(p + "ffn.2.weight", {d, 4 * d}, 2); add(p + "ffn.2.bias", {d}, 2);
    }
    add("_tfdec.layer_norm.weight", {d}, 2); add("_tfdec.layer_norm.bias", {d}, 2);
    add("_out.weight", {V, d}, 2);

    long long off = 0;
    int g = 0;
    group_begin_[0] = 0;
    for (size_t i = 0; i < params_.size(); ++i) {
        while (g < params_[i].group) group_begin_[++g] = off;
        params_[i].offset = off;
        off += (params_[i].numel + 3) & ~3ll;       // keep every tensor 16-byte aligned (float4 / MFMA staging)
        index_[params_[i].name] = (int)i;
    }
    while (g < 3) group_begin_[++g] = off;
    flat_size_ = off;
    blk_.resize(NB);
    layout_workspace(false);
}

SlateModel::~SlateModel() {
    if (ev_fork_) (void)hipEventDestroy(ev_fork_);
    if (ev_join_) (void)hipEventDestroy(ev_join_);
    if (ev_tokens_) (void)hipEventDestroy(ev_tokens_);
    if (side_) (void)hipStreamDestroy(side_);
    for (auto& kv : enc_graphs_) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    for (hipEvent_t e : ev_dw_) if (e) (void)hipEventDestroy(e);
    if (ev_join2_) (void)hipEventDestroy(ev_join2_);
    if (side2_) (void)hipStreamDestroy(side2_);
    if (cap_) (void)hipStreamDestroy(cap_);
}

float* SlateModel::P(const std::string& n) const { return p_ + params_[index_.at(n)].offset; }
float* SlateModel::G(const std::string& n) const { return g_ + params_[index_.at(n)].offset; }

float* SlateModel::carve(const char* name, size_t n) {
    const size_t bytes = (n * 4 + 255) & ~(size_t)255;
    float* p = reinterpret_cast<float*>(ws_ + ws_off_);
    ws_off_ += bytes;
    if (ws_commit_ && name) named_[name] = std::make_pair(p, n);
    return p;
}

void SlateModel::layout_workspace(bool commit) {
    ws_commit_ = commit;
    ws_off_ = 0;
    x3_of_.clear();
    const size_t B = Bmax, BT = B * T, BN = B * N, BK = B * K;
    metrics_ = carve("metrics", 64);
    // transient scratch: split-k slabs (<= 1024 slabs of the largest weight tile set), wgrad slabs, column sums
    scratch_floats_ = 0;
    {
        size_t need = conv_wgrad_ws_floats((int)(B * (cfg.use_bcdec ? K : 1)), S, S, 5, 64) + (size_t)512 * 2304 * 2;
        size_t sk = (size_t)32 * V * d + (size_t)(1 << 20);        // split-k slabs ([V,d] weights x16; 170 slabs of the [4d,d] FFN weights)
        if (sk > need) need = sk;
        size_t cs = (size_t)N * C * 8 + (size_t)T * d * 8 + (1 << 20);   // column-sum partials (pos map / pe)
        if (cs > need) need = cs;
        if (cfg.use_bcdec && bc_layer1_bwd_ws_floats((int)(B * K), S) > need) need = bc_layer1_bwd_ws_floats((int)(B * K), S);
        const size_t ca = cross_attn_bwd_ws_floats((int)B, T, K, d, NH), eb = embed_bwd_ws_floats((long long)B * T, V, d);
        if (ca > need) need = ca;
        if (eb > need) need = eb;
        scratch_floats_ = need + (1 << 20);
    }
    scratch_ = carve(nullptr, scratch_floats_);
    scratch2_ = cfg.use_bcdec ? scratch_ : carve(nullptr, scratch_floats_);
    obs8_ = carve("obs8", BN * 8);
    obs_stage_ = carve(nullptr, (B < 32 ? B : 32) * (size_t)cfg.obs_channels * N);
    seed_dev_ = reinterpret_cast<unsigned long long*>(carve(nullptr, 64));
    e1_ = carve("enc1", BN * 64); e2_ = carve("enc2", BN * 64); e3_ = carve("enc3", BN * 64); e4_ = carve("feats", BN * 64);
    posmap_ = carve(nullptr, (size_t)N * C); gridT_ = carve(nullptr, (size_t)N * 4);
    ln0_ = carve(nullptr, BN * 64); ln0_mean_ = carve(nullptr, BN); ln0_rstd_ = carve(nullptr, BN);
    h1_ = carve("sa_mlp_hidden", BN * 64); x_ = carve("sa_inputs", BN * 64);
    slots0_ = carve("slots0", BK * D); slot_noise_ = carve(nullptr, BK * D); slots_ = carve("slots", BK * D);
    attn_ = carve("attn", BN * K);
    attn_heads_ = SH > 1 ? carve(nullptr, BN * K * SH) : nullptr;
    const SaSave so = sa_save_layout(C, D, H, SH);
    const SaGrad go = sa_grad_layout(C, D, H, SH);
    const SaWts wo = sa_wts_layout(C, D, H);
    sa_save_ = carve("sa_save", BK * I * so.ld);
    sa_grows_ = carve(nullptr, BK * I * go.ld);
    sa_wts_ = carve(nullptr, wo.total);
    sa_small_ = carve(nullptr, B * (4 * D + 2 * C));
    sa_xchg_ = carve(nullptr, B * sa_xchg_floats_host(K * SH, D));
    sa_parts_ = carve(nullptr, sa_parts_floats_host((int)B, K * SH));
    sa_pack_dev_ = reinterpret_cast<PackEntry*>(carve(nullptr, 64 * sizeof(PackEntry) / 4 + 64));
    for (int i = 0; i < 4; ++i) {
        const int cin = i == 0 ? 8 : 64;
        cw_fwd_[i] = carve(nullptr, (size_t)25 * cin * 64);
        cw_bwd_[i] = i == 0 ? nullptr : carve(nullptr, (size_t)25 * 64 * 64);
        if (conv_x3_ < 0) { const char* e = getenv("OCRL_CONV_X3"); conv_x3_ = e ? atoi(e) : 0; }
        if (conv_x3_ && i > 0) {          // exploratory split-precision packs beside the fp32 ones (csrc/conv_x3.hip)
            float* f3 = carve(nullptr, conv_x3_pack_floats());
            float* b3 = carve(nullptr, conv_x3_pack_floats());
            if (ws_commit_) { x3_of_[cw_fwd_[i]] = f3; x3_of_[cw_bwd_[i]] = b3; }
        }
    }
    gslots_ = carve(nullptr, BK * D); gslots0_ = carve(nullptr, BK * D);
    gA_ = carve(nullptr, BN * 64); gB_ = carve(nullptr, BN * 64); gC_ = cfg.use_bcdec ? gA_ : carve(nullptr, BN * 64);
    gmap_ = carve(nullptr, (size_t)N * C);
    if (cfg.use_bcdec) {
        const size_t BKN = BK * (size_t)N;
        bc_Wc_ = carve(nullptr, 25 * 64 * 5 + 64); bc_W1r_ = carve(nullptr, (size_t)25 * 64 * D); bc_P1_ = carve(nullptr, (size_t)N * 64);
        bc_M_ = carve(nullptr, BK * 1600); bc_T_ = carve(nullptr, BK * 1600);
        bc_c1_ = carve("bc_c1", BKN * 64); bc_c2_ = carve("bc_c2", BKN * 64); bc_c3_ = carve("bc_c3", BKN * 64);
        bc_out4_ = carve(nullptr, BKN * 4); bc_dout4_ = carve(nullptr, BKN * 4);
        bc_gA_ = carve(nullptr, BKN * 64); bc_gB_ = carve(nullptr, BKN * 64);
        for (int i = 0; i < 2; ++i) {
            bc_pk_[i] = carve(nullptr, 25 * 64 * 64); bc_pkb_[i] = carve(nullptr, 25 * 64 * 64);
            if (conv_x3_ > 0) {           // exploratory split-precision packs of the broadcast decoder's 5x5 / 64-channel layers
                float* f3 = carve(nullptr, conv_x3_pack_floats());
                float* b3 = carve(nullptr, conv_x3_pack_floats());
                if (ws_commit_) { x3_of_[bc_pk_[i]] = f3; x3_of_[bc_pkb_[i]] = b3; }
            }
        }
        bc_Wk4_ = carve(nullptr, 9 * 64 * 4); bc_Wb4_ = carve(nullptr, 9 * 64 * 4);
        bc_dW1r_ = carve(nullptr, (size_t)25 * 64 * D); bc_dWc_ = carve(nullptr, 25 * 64 * 5 + 64);
        bc_dT_ = carve(nullptr, BK * 1600); bc_dM_ = carve(nullptr, BK * 1600); bc_G1_ = carve(nullptr, (size_t)N * 64);
        recon_ = carve("recon", BN * 4);
    } else {
    patches_ = carve("patches", BT * 16 * cfg.obs_channels);
    for (int i = 0; i < 7; ++i) de_[i] = carve(fmt("dvae_enc%d", i).c_str(), BT * 64);
    zraw_ = carve("zraw", BT * V);
    z_ = carve("z", BT * V);
    zdec_ = cfg.hard ? carve("z_st", BT * V) : z_;
    tokens_ = reinterpret_cast<int*>(carve("tokens", BT));
    // soft-max heads fused into the vocabulary GEMMs: per-row / per-segment statistics (gemm.hip epi_mode 1-3)
    {
        const size_t nseg = (size_t)gemm_stat_segments(V);
        zstat_ = carve(nullptr, BT * nseg * 2); zhstat_ = carve(nullptr, BT * nseg); zhidx_ = reinterpret_cast<int*>(carve(nullptr, BT * nseg));
        zlse_ = carve("z_lse", BT); zdot_ = carve(nullptr, BT);
        cestat_ = carve(nullptr, BT * nseg * 2); celse_ = carve("ce_lse", BT); cepart_ = carve(nullptr, BT / 16 + 16);
    }
    // post-activation outputs of the dVAE decoder blocks (named: the parity tests read their ReLU masks)
    dd0_ = carve("dvae_dec0", BT * 64); dd1_ = carve("dvae_dec1", BT * 64); dd2_ = carve("dvae_dec2", BT * 64); dd3_ = carve("dvae_dec3", BT * 64);
    dd4_ = carve("dvae_dec4", BT * 256); ps1_ = carve(nullptr, BT * 256);
    dd6_ = carve("dvae_dec6", BT * 256); dd7_ = carve("dvae_dec7", BT * 256); dd8_ = carve("dvae_dec8", BT * 256);
    dd9_ = carve("dvae_dec9", BT * 1024); ps2_ = carve(nullptr, BN * 64);
    recon_ = carve("recon", BN * 4); drecon_ = carve(nullptr, BN * 4);
    for (int i = 0; i < 2; ++i) {
        dw_fwd_[i] = carve(nullptr, 9 * 64 * 64); dw_bwd_[i] = carve(nullptr, 9 * 64 * 64);
        if (conv_x3_ > 0) {           // exploratory split-precision packs of the dVAE decoder's 3x3 / 64-channel layers
            float* f3 = carve(nullptr, conv_x3_pack_floats(3));
            float* b3 = carve(nullptr, conv_x3_pack_floats(3));
            if (ws_commit_) { x3_of_[dw_fwd_[i]] = f3; x3_of_[dw_bwd_[i]] = b3; }
        }
    }
    w11p_ = carve(nullptr, 4 * 64);
    mem_ = carve("mem", BK * d); emb_ = carve("emb", BT * d);
    for (int b = 0; b < NB; ++b) {
        Blk& k = blk_[b];
        k.ln1 = carve(nullptr, BT * d); k.ln1_mean = carve(nullptr, BT); k.ln1_rstd = carve(nullptr, BT);
        k.q = carve(nullptr, BT * 3 * d); k.k = k.q + d; k.v = k.q + 2 * d;      // fused [BT, 3d] projection output
        k.lse = carve(nullptr, B * NH * (size_t)T); k.ao = carve(nullptr, BT * d); k.x1 = carve(nullptr, BT * d);
        k.ln2 = carve(nullptr, BT * d); k.ln2_mean = carve(nullptr, BT); k.ln2_rstd = carve(nullptr, BT);
        k.cq = carve(nullptr, BT * d); k.ck = carve(nullptr, BK * d); k.cv = carve(nullptr, BK * d);
        k.cP = carve(nullptr, B * NH * (size_t)T * K); k.cao = carve(nullptr, BT * d); k.x2 = carve(nullptr, BT * d);
        k.ln3 = carve(nullptr, BT * d); k.ln3_mean = carve(nullptr, BT); k.ln3_rstd = carve(nullptr, BT);
        k.f1 = carve(fmt("blk%d.ffn_hidden", b).c_str(), BT * 4 * d); k.x3 = carve(nullptr, BT * d);
    }
    bg_.resize(NB);
    {   // folded cross attention (xattn.hip)
        const size_t NC = xattn_supported(K, d, NH) ? (size_t)NH * xattn_kp(K, NH) : 16;
        xa_zero_base_ = reinterpret_cast<float*>(ws_ + ws_off_);
        for (int b = 0; b < NB; ++b) {
            Blk& k = blk_[b];
            k.xaAb = carve(nullptr, B * NC * d); k.xaAbT = carve(nullptr, B * NC * d); k.xaVo = carve(nullptr, B * NC * d); k.xaVoT = carve(nullptr, B * NC * d);
        }
        xa_zero_floats_ = (size_t)(reinterpret_cast<float*>(ws_ + ws_off_) - xa_zero_base_);
        for (int b = 0; b < NB; ++b) { bg_[b].xaPd = carve(nullptr, BT * NC); bg_[b].xaDs = carve(nullptr, BT * NC); }
        xa_dAb_ = carve(nullptr, B * NC * d); xa_dVo_ = carve(nullptr, B * NC * d);
        xa_pq_ = carve(nullptr, B * (size_t)d * d); xa_po_ = carve(nullptr, B * (size_t)d * d);
    }
    attn_delta_ = carve(nullptr, B * NH * (size_t)T);
    lnf_ = carve("dec_out", BT * d); lnf_mean_ = carve(nullptr, BT); lnf_rstd_ = carve(nullptr, BT);
    pred_ = carve("pred", BT * V);
    gx_ = carve(nullptr, BT * d); gbr_ = carve(nullptr, BT * d); gt1_ = carve(nullptr, BT * d); gt2_ = carve(nullptr, BT * d);
    gt3_ = carve(nullptr, BT * d); gf1_ = carve(nullptr, BT * 4 * d); gqkv_ = carve(nullptr, BT * 3 * d);
    for (int b = 0; b < NB; ++b) {
        BlkG& q = bg_[b];
        if (b == 0) { q.gbr[0] = gbr_; q.gf1 = gf1_; q.gt2 = gt2_; q.gqkv = gqkv_; }
        else { q.gbr[0] = carve(nullptr, BT * d); q.gf1 = carve(nullptr, BT * 4 * d); q.gt2 = carve(nullptr, BT * d); q.gqkv = carve(nullptr, BT * 3 * d); }
        q.gbr[1] = carve(nullptr, BT * d); q.gbr[2] = carve(nullptr, BT * d);
    }
    scratch3_ = carve(nullptr, scratch_floats_);
    gmem_ = carve(nullptr, BK * d); gck_ = carve(nullptr, BK * d); gcv_ = carve(nullptr, BK * d);
    gdA_ = carve(nullptr, BN * 64); gdB_ = carve(nullptr, BN * 64);
    }
    {
        const int ldc0 = (25 * cfg.obs_channels + 3) & ~3;
        col0_ = carve(nullptr, BN * ldc0); dw0p_ = carve(nullptr, (size_t)64 * ldc0);
    }
    if (!commit) ws_bytes_ = ws_off_ + 4096;
}

int SlateModel::bind(float* p, float* g, float* m, float* v, void* ws, size_t ws_bytes) {
    OCRL_REQUIRE(p && g && ws, "bind: null buffer");
    OCRL_REQUIRE(ws_bytes >= ws_bytes_, "bind: workspace too small (%zu < %zu)", ws_bytes, ws_bytes_);
    OCRL_REQUIRE(((uintptr_t)p & 255) == 0 && ((uintptr_t)g & 255) == 0 && ((uintptr_t)ws & 255) == 0, "bind: buffers must be 256-byte aligned");
    OCRL_REQUIRE(d % 64 == 0 && d <= 256 && D % 64 == 0 && C == 64 && V % 256 == 0 && S % 4 == 0 && d % NH == 0 && DH % 4 == 0 && DH <= 64,
                 "unsupported configuration (d_model/slot_size multiples of 64 <= 256, cnn hidden 64, vocab %% 256, obs_size %% 4)");
    OCRL_REQUIRE(cfg.obs_channels == 3, "obs_channels must be 3");
    OCRL_REQUIRE(T % 4 == 0, "obs_size/4 squared must be a multiple of 4");
    p_ = p; g_ = g; m_ = m; v_ = v;
    ws_ = static_cast<char*>(ws);
    named_.clear();
    clear_encode_graphs();       // captured against the old buffers
    packs_valid_ = false;
    layout_workspace(true);
    if (!side_) {
        // OCRL_OVERLAP (default 5): the dVAE branch (many short 64-wide products that do not fill the machine) runs on a side stream.
        //   5 = forward forked right after the encoder convolutions (beside the input LayerNorm / MLP and the slot-attention chain),
        //       backward beside the transformer-decoder backward and joined before the slot-attention backward and the 5x5
        //       convolutions, so those never share the GPU and their timings stay comparable
        //   4 = as 5 with the forward forked at the start of the step (beside the encoder convolutions)
        //   3 = as 5 with the forward forked at the slot-attention launches
        //   2 = forward and backward beside the slot-attention launches only (round 1's default, when slot attention left half the CUs idle)
        //   1 = whole dVAE branch beside encoder + decoder (per-kernel timings of both branches stop being comparable); 0 = single stream
        // In modes 3-5 the main stream waits for the dVAE tokens only (OCRL_TOKENS_LATE=1 restores the full join before the decoder).
        const char* e = getenv("OCRL_OVERLAP");
        overlap_mode_ = e ? atoi(e) : 5;
        if (overlap_mode_) {
            OCRL_HIP(hipStreamCreateWithFlags(&side_, hipStreamNonBlocking));
            OCRL_HIP(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
            OCRL_HIP(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
            OCRL_HIP(hipEventCreateWithFlags(&ev_tokens_, hipEventDisableTiming));
            // OCRL_DW_SIDE (default 2): the weight-gradient products of the transformer decoder leave the main stream -- nothing later in
            // the step reads them, while the input-gradient chain they would otherwise interrupt is the step's critical path.
            //   1 = on the dVAE side stream (behind the dVAE backward), 2 = on a stream of their own, 0 = on the main stream (round 2)
            const char* w = getenv("OCRL_DW_SIDE");
            dw_mode_ = (overlap_mode_ >= 3 && !cfg.use_bcdec) ? (w ? atoi(w) : 2) : 0;
            if (dw_mode_) {
                for (hipEvent_t& ev : ev_dw_) OCRL_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                OCRL_HIP(hipEventCreateWithFlags(&ev_join2_, hipEventDisableTiming));
                if (dw_mode_ == 2) OCRL_HIP(hipStreamCreateWithFlags(&side2_, hipStreamNonBlocking));
            }
        }
    }
    {
        const char* e = getenv("OCRL_XATTN");
        xattn_ = !cfg.use_bcdec && xattn_supported(K, d, NH) && (e ? atoi(e) != 0 : true);
        if (!cfg.use_bcdec && xa_zero_floats_) OCRL_HIP(hipMemset(xa_zero_base_, 0, xa_zero_floats_ * sizeof(float)));      // padding columns of the folded operands
    }
    // static tables
    RC(posgrid_launch(gridT_, S, 0));
    RC(fill_launch(recon_, (long long)Bmax * N * 4, 0.f, 0));
    // slot-attention weight pack table
    const SaWts wo = sa_wts_layout(C, D, H);
    const std::string sa = "_slotattn.slot_attention.";
    std::vector<PackEntry> ent;
    auto e = [&](const std::string& name, int rows, int cols, int off, int tr) {
        PackEntry x; x.src = P(name); x.rows = rows; x.cols = cols; x.dst_off = off; x.transpose = tr;
        ent.push_back(x);
    };
    e(sa + "norm_inputs.weight", 1, C, wo.ln_in_g, 0); e(sa + "norm_inputs.bias", 1, C, wo.ln_in_b, 0);
    e(sa + "norm_slots.weight", 1, D, wo.ln_s_g, 0); e(sa + "norm_slots.bias", 1, D, wo.ln_s_b, 0);
    e(sa + "norm_mlp.weight", 1, D, wo.ln_m_g, 0); e(sa + "norm_mlp.bias", 1, D, wo.ln_m_b, 0);
    e(sa + "project_q.weight", D, D, wo.Wq, 2); e(sa + "project_q.weight", D, D, wo.WqT, 3);
    e(sa + "project_k.weight", D, C, wo.Wk, 2); e(sa + "project_k.weight", D, C, wo.WkT, 3);
    e(sa + "project_v.weight", D, C, wo.Wv, 2); e(sa + "project_v.weight", D, C, wo.WvT, 3);
    e(sa + "gru.weight_ih", 3 * D, D, wo.Wih, 2); e(sa + "gru.weight_ih", 3 * D, D, wo.WihT, 3);
    e(sa + "gru.weight_hh", 3 * D, D, wo.Whh, 2); e(sa + "gru.weight_hh", 3 * D, D, wo.WhhT, 3);
    e(sa + "gru.bias_ih", 1, 3 * D, wo.bih, 0); e(sa + "gru.bias_hh", 1, 3 * D, wo.bhh, 0);
    e(sa + "mlp.0.weight", H, D, wo.W0, 2); e(sa + "mlp.0.weight", H, D, wo.W0T, 3); e(sa + "mlp.0.bias", 1, H, wo.b0, 0);
    e(sa + "mlp.2.weight", D, H, wo.W2, 2); e(sa + "mlp.2.weight", D, H, wo.W2T, 3); e(sa + "mlp.2.bias", 1, D, wo.b2, 0);
    sa_pack_n_ = (int)ent.size();
    sa_pack_max_ = 3 * D * D;
    OCRL_REQUIRE(sa_pack_n_ <= 64, "pack table overflow");
    OCRL_HIP(hipMemcpy(sa_pack_dev_, ent.data(), ent.size() * sizeof(PackEntry), hipMemcpyHostToDevice));
    OCRL_HIP(hipDeviceSynchronize());
    have_fwd_ = false;
    return 0;
}

int SlateModel::tensor(const char* name, float** ptr, long long* count) const {
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

int SlateModel::soft_z(hipStream_t st) {
    if (!fused_heads() || cfg.use_bcdec) return 0;          // z_ is written by the forward itself
    OCRL_REQUIRE(have_scores_, "soft_z: no forward since the last backward (the scores have been overwritten by their gradient)");
    return exp_rows_launch(zraw_, zlse_, z_, (long long)last_.B * T, V, st);
}
int SlateModel::dropout_mask(unsigned site, long long n, float* out, hipStream_t st) const {
    return dropout_mask_launch(out, n, pdrop_, last_.seed, site, st);
}

// ---------------------------------------------------------------------------------------------
int SlateModel::lin_fwd(const float* x, int ldx, const float* W, const float* b, float* y, int ldy, long long M, int Nn, int Kk, int relu,
                        const float* resid, int ldr, float drop_p, unsigned site, hipStream_t st) {
    GemmArgs a;
    a.A = x; a.B = W; a.C = y; a.M = (int)M; a.N = Nn; a.K = Kk; a.lda = ldx; a.ldb = Kk; a.ldc = ldy; a.akc = 1; a.bkc = 1;
    a.bias = b; a.relu = relu; a.resid = resid; a.ldr = ldr; a.drop_p = drop_p; a.drop_seed = last_.seed; a.drop_site = site;
    return gemm_launch(a, st);
}
// dx[M,K_in] = (drop(dy)[M,N_out] W[N_out,K_in]) * (mask > 0) + resid
int SlateModel::lin_bwd_x(const float* dy, int ld_dy, const float* W, float* dx, int ldx, long long M, int N_out, int K_in,
                          const float* mask, int ldmask, const float* resid, int ldr, hipStream_t st, Drop dr, Xf xf) {
    GemmArgs a;
    a.a_mode = xf.a_mode; a.b_mode = xf.b_mode; a.x_lse = xf.lse; a.x_tok = xf.tok; a.x_scale = xf.scale;
    a.A = dy; a.B = W; a.C = dx; a.M = (int)M; a.N = K_in; a.K = N_out; a.lda = ld_dy; a.ldb = K_in; a.ldc = ldx; a.akc = 1; a.bkc = 0;
    a.mask = mask; a.ldmask = ldmask; a.resid = resid; a.ldr = ldr;
    if (dr.p > 0.f) { a.adrop_p = dr.p; a.adrop_site = dr.site; a.adrop_ld = N_out; a.drop_seed = last_.seed; }
    return gemm_launch(a, st);
}
// dW[N_out,K_in] = alpha * drop(dy)^T x (split over the M rows);  db[N_out] = column sums of drop(dy), fused into the GEMM
int SlateModel::lin_bwd_w(const float* dy, int ld_dy, const float* x, int ldx, float* dW, float* db, long long M, int N_out, int K_in,
                          float alpha, hipStream_t st, Drop dr, Xf xf) {
    GemmArgs a;
    a.a_mode = xf.a_mode; a.b_mode = xf.b_mode; a.x_lse = xf.lse; a.x_tok = xf.tok; a.x_scale = xf.scale;
    a.A = dy; a.B = x; a.C = dW; a.M = N_out; a.N = K_in; a.K = (int)M; a.lda = ld_dy; a.ldb = ldx; a.ldc = K_in; a.akc = 0; a.bkc = 0;
    a.alpha = alpha;
    if (dr.p > 0.f) { a.adrop_p = dr.p; a.adrop_site = dr.site; a.adrop_ld = N_out; a.drop_seed = last_.seed; }
    // output tiles as gemm.hip will cut them (128x192 for 192 input features over >= 4096 rows, else 128x128 / 128x64): the split count
    // aims at ~1024 workgroups -- counting 64-wide tiles for the 128x192 case left 340 workgroups on 256 CUs (PMC: 0.95 waves per SIMD)
    const int col_tiles = (K_in == 192 && M >= 4096) ? 1 : cdiv(K_in, (K_in % 128 == 0) ? 128 : 64);
    const int tiles = cdiv(N_out, 128) * col_tiles;
    long long splits = 1024 / tiles;
    if (splits > M / 256) splits = M / 256;
    if (splits < 1) splits = 1;
    const long long slab = (long long)N_out * K_in;
    const long long bslab = (N_out + 3) & ~3;
    if (splits * (slab + bslab) > (long long)scratch_floats_) splits = (long long)scratch_floats_ / (slab + bslab);
    if (splits > 1) {
        a.splitk = (int)splits; a.C = scratch_; a.sCsplit = slab;
        float* bpart = scratch_ + splits * slab;
        if (db) { a.bias_out = bpart; a.sBias = bslab; }
        RC(gemm_launch(a, st));
        RC(splitk_reduce_launch(scratch_, dW, slab, (int)splits, slab, 0, st));
        if (db) {
            if (N_out % 4 == 0) RC(splitk_reduce_launch(bpart, db, N_out, (int)splits, bslab, 0, st));
            else RC(colsum_launch(bpart, bslab, db, splits, N_out, 0, 1.f, bpart + splits * bslab, scratch_floats_ - (size_t)(splits * (slab + bslab)), st));
        }
    } else {
        if (db) a.bias_out = db;
        RC(gemm_launch(a, st));
    }
    return 0;
}
int SlateModel::conv_layer_fwd(const float* x, const float* pack, const float* bias, float* y, int Bn, int Hh, int Ww, int KS, int CIN,
                               int relu, const float* posmap, const float* mask, hipStream_t st) {
    ConvArgs a;
    a.X = x; a.Wp = pack; a.Y = y; a.B = Bn; a.H = Hh; a.W = Ww; a.bias = bias; a.relu = relu; a.posmap = posmap; a.mask = mask;
    if (conv_x3_ > 0 && (KS == 5 || KS == 3) && CIN == 64 && !conv_lowlat_) {
        auto it = x3_of_.find(pack);
        if (it != x3_of_.end()) return conv_x3_launch(a, it->second, st, KS);
    }
    return conv_fwd_launch(a, KS, CIN, 64, st, conv_lowlat_);
}
int SlateModel::conv_layer_wgrad(const float* x, const float* dy, float* dW, float* db, int Bn, int Hh, int Ww, int KS, int CIN,
                                 int cin_real, hipStream_t st) {
    WgradArgs a;
    a.X = x; a.dY = dy; a.part = scratch_; a.B = Bn; a.H = Hh; a.W = Ww;
    OCRL_REQUIRE(conv_wgrad_ws_floats(Bn, Hh, Ww, KS, CIN) <= scratch_floats_, "conv wgrad: scratch too small");
    RC(conv_wgrad_launch(a, KS, CIN, 64, cin_real, dW, 0, st, conv_x3_ > 0 ? 1 : 0));
    if (db) RC(colsum_launch(dy, 64, db, (long long)Bn * Hh * Ww, 64, 0, 1.f, scratch_, scratch_floats_, st));
    return 0;
}

int SlateModel::pack_weights(hipStream_t st, bool encoder_only) {
    RC(conv_pack_launch(P("_enc._encoder.0.m.weight"), cw_fwd_[0], nullptr, 5, 8, 64, cfg.obs_channels, st));
    RC(conv_pack_launch(P("_enc._encoder.1.m.weight"), cw_fwd_[1], cw_bwd_[1], 5, 64, 64, 64, st));
    RC(conv_pack_launch(P("_enc._encoder.2.m.weight"), cw_fwd_[2], cw_bwd_[2], 5, 64, 64, 64, st));
    RC(conv_pack_launch(P("_enc._encoder.3.weight"), cw_fwd_[3], cw_bwd_[3], 5, 64, 64, 64, st));
    if (conv_x3_ > 0) {
        const char* names[4] = {nullptr, "_enc._encoder.1.m.weight", "_enc._encoder.2.m.weight", "_enc._encoder.3.weight"};
        for (int i = 1; i < 4; ++i) {
            auto f = x3_of_.find(cw_fwd_[i]), b = x3_of_.find(cw_bwd_[i]);
            if (f != x3_of_.end()) RC(conv_pack_x3_launch(P(names[i]), const_cast<float*>(f->second), b != x3_of_.end() ? const_cast<float*>(b->second) : nullptr, st));
        }
    }
    if (!cfg.use_bcdec && !encoder_only) {
        RC(conv_pack_launch(P("_dvae._decoder.1.m.weight"), dw_fwd_[0], dw_bwd_[0], 3, 64, 64, 64, st));
        RC(conv_pack_launch(P("_dvae._decoder.6.m.weight"), dw_fwd_[1], dw_bwd_[1], 3, 64, 64, 64, st));
        if (conv_x3_ > 0) {
            const char* names[2] = {"_dvae._decoder.1.m.weight", "_dvae._decoder.6.m.weight"};
            for (int i = 0; i < 2; ++i) {
                auto f = x3_of_.find(dw_fwd_[i]), b = x3_of_.find(dw_bwd_[i]);
                if (f != x3_of_.end()) RC(conv_pack_x3_launch(P(names[i]), const_cast<float*>(f->second), b != x3_of_.end() ? const_cast<float*>(b->second) : nullptr, st, 3));
            }
        }
        RC(copy_launch(P("_dvae._decoder.11.weight"), w11p_, cfg.obs_channels * 64, st));    // [3,64] -> [4,64], row 3 zero
        RC(fill_launch(w11p_ + cfg.obs_channels * 64, (4 - cfg.obs_channels) * 64, 0.f, st));
    }
    RC(pack_launch(sa_pack_dev_, sa_pack_n_, sa_pack_max_, sa_wts_, st));
    RC(posmap_launch(P("_enc_pos.channels_map.weight"), P("_enc_pos.channels_map.bias"), posmap_, S, C, st));
    return 0;
}

// ---------------------------------------------------------------------------------------------
// CNN encoder + slot attention (ocrs/common/models.py:96-107, slot_attn.py:147-161)
int SlateModel::fwd_encoder(const StepInputs& in, hipStream_t st, int fork_dvae) {
    const int B = in.B;
    const long long BN = (long long)B * N;
    auto fork_here = [&]() -> int {
        RC(fork_side(st));
        std::swap(scratch_, scratch2_);
        const int rc = fwd_dvae(in, side_);
        std::swap(scratch_, scratch2_);
        return rc;
    };
    // slot initialisation and the preparation launch of the slot-attention chain (LayerNorm, q, folded query of iteration 0) need the
    // weights and the noise only: issued first, they run on an otherwise idle machine.  After the fork their whole-CU-LDS workgroups
    // queue behind the Gumbel head's 32 768 small workgroups on the side stream (measured: 1.46 ms instead of 60 us, on the critical
    // path of the decoder)
    RC(slot_init_launch(P("_slotattn.slot_mu"), P("_slotattn.slot_log_sigma"), in.noise_slots, slots0_, B * K, D, in.seed, st, in.seed_dev));
    SlotAttnArgs a;
    a.B = B; a.N = N; a.C = C; a.K = K; a.D = D; a.H = H; a.I = I; a.NH = SH; a.eps = 1e-8f; a.scale = 1.0f / sqrtf((float)(D / SH));
    a.x = x_; a.slots0 = slots0_; a.wts = sa_wts_; a.slots = slots_; a.attn = attn_; a.attn_heads = attn_heads_; a.save = sa_save_;
    if (conv_lowlat_ && frozen_) a.save = nullptr;       // encode() of a frozen encoder: no backward can follow, the saved-activation rows are not written
    a.xchg = sa_xchg_; a.parts = sa_parts_;
    a.phase = 1;
    RC(slot_attn_launch(a, 0, st));
    if (fork_dvae == 2) RC(fork_here());
    RC(nchw_to_nhwc8_launch(in.obs, obs8_, B, cfg.obs_channels, S, S, st));
    RC(conv_layer_fwd(obs8_, cw_fwd_[0], P("_enc._encoder.0.m.bias"), e1_, B, S, S, 5, 8, 1, nullptr, nullptr, st));
    RC(conv_layer_fwd(e1_, cw_fwd_[1], P("_enc._encoder.1.m.bias"), e2_, B, S, S, 5, 64, 1, nullptr, nullptr, st));
    RC(conv_layer_fwd(e2_, cw_fwd_[2], P("_enc._encoder.2.m.bias"), e3_, B, S, S, 5, 64, 1, nullptr, nullptr, st));
    RC(conv_layer_fwd(e3_, cw_fwd_[3], P("_enc._encoder.3.bias"), e4_, B, S, S, 5, 64, 0, posmap_, nullptr, st));
    if (fork_dvae == 3) RC(fork_here());
    RC(layernorm_fwd_launch(e4_, P("_slotattn.layer_norm.weight"), P("_slotattn.layer_norm.bias"), ln0_, ln0_mean_, ln0_rstd_, BN, C, st));
    RC(lin_fwd(ln0_, C, P("_slotattn.mlp.0.weight"), P("_slotattn.mlp.0.bias"), h1_, C, BN, C, C, 1, nullptr, 0, 0.f, 0, st));
    RC(lin_fwd(h1_, C, P("_slotattn.mlp.2.weight"), P("_slotattn.mlp.2.bias"), x_, C, BN, C, C, 0, nullptr, 0, 0.f, 0, st));
    if (fork_dvae == 1) RC(fork_here());
    a.phase = 2;
    RC(slot_attn_launch(a, 0, st));
    return 0;
}

// dVAE encode -> Gumbel softmax -> decode -> reconstruction loss (models.py:14-45, utils.py:75-85)
int SlateModel::fwd_dvae(const StepInputs& in, hipStream_t st) {
    const int B = in.B;
    const long long BT = (long long)B * T, BN = (long long)B * N;
    const int ch = cfg.obs_channels;
    RC(patchify4_launch(in.obs, patches_, B, ch, S, st));
    RC(lin_fwd(patches_, 16 * ch, P("_dvae._encoder.0.m.weight"), P("_dvae._encoder.0.m.bias"), de_[0], 64, BT, 64, 16 * ch, 1, nullptr, 0, 0.f, 0, st));
    for (int i = 1; i < 7; ++i)
        RC(lin_fwd(de_[i - 1], 64, P(fmt("_dvae._encoder.%d.m.weight", i)), P(fmt("_dvae._encoder.%d.m.bias", i)), de_[i], 64, BT, 64, 64, 1, nullptr, 0, 0.f, 0, st));
    if (fused_heads()) {
        // 64 -> vocabulary head with both Gumbel samples in its epilogue (ocrs/slate/slate_module.py:125-127, ocrs/common/utils.py:72-85):
        // zraw_ <- (logits + g1) / tau, the soft sample's scores.  z = softmax(zraw_) is never written: the decoder product and the
        // backward rebuild it from zraw_ and z_lse while staging their operand tiles.  (The reference adds the noise to
        // log_softmax(logits); the row shift cancels in the soft-max and in the argmax.)
        GemmArgs a;
        a.A = de_[6]; a.B = P("_dvae._encoder.7.weight"); a.C = zraw_; a.M = (int)BT; a.N = V; a.K = 64; a.lda = 64; a.ldb = 64; a.ldc = V;
        a.bias = P("_dvae._encoder.7.bias");
        a.epi_mode = 2; a.stat = zstat_; a.hstat = zhstat_; a.hidx = zhidx_; a.e1 = in.noise_z; a.e2 = in.noise_zh; a.e_seed = in.seed;
        a.e_scale = 1.0f / in.tau;
        RC(gemm_launch(a, st));
        // hard sample: with injected noise the arg-max of l + g2 (segment maxima from the epilogue); with the device RNG a draw from
        // Categorical(soft-max(l)) by inverse CDF over the segment masses (one pair of uniforms per row instead of a Gumbel draw per entry)
        RC(softmax_stat_combine_launch(zstat_, gemm_stat_segments(V), BT, zlse_, zhstat_, zhidx_, tokens_, nullptr, 0, nullptr, nullptr, 0.f, nullptr, 0, st,
                                       in.noise_z ? nullptr : zraw_, V, in.tau, in.seed));
        have_scores_ = true;
    } else {
    RC(lin_fwd(de_[6], 64, P("_dvae._encoder.7.weight"), P("_dvae._encoder.7.bias"), zraw_, V, BT, V, 64, 0, nullptr, 0, 0.f, 0, st));
    RC(gumbel_softmax_launch(zraw_, in.noise_z, in.noise_zh, z_, tokens_, BT, V, in.tau, in.seed, st, cfg.hard ? zdec_ : nullptr));
    }
    // the transformer decoder only needs the tokens: it may start on the main stream while the dVAE decoder and the reconstruction loss
    // are still running here (tokens_early mode of forward())
    if (side_ && st == side_ && ev_tokens_) OCRL_HIP(hipEventRecord(ev_tokens_, st));
    RC(dvae_decode(B, drecon_, st));
    return 0;
}

// dVAE decoder on z_ -> recon_ (ocrs/common/models.py:24-37,44-45) and the reconstruction loss into metrics[0]
int SlateModel::dvae_decode(int B, float* drecon, hipStream_t st, const float* zin) {
    const long long BT = (long long)B * T, BN = (long long)B * N;
    const int ch = cfg.obs_channels;
    // decoder
    if (!zin && fused_heads()) {
        GemmArgs a;        // A = softmax(zraw_) rebuilt on the fly
        a.A = zraw_; a.B = P("_dvae._decoder.0.m.weight"); a.C = dd0_; a.M = (int)BT; a.N = 64; a.K = V; a.lda = V; a.ldb = V; a.ldc = 64;
        a.bias = P("_dvae._decoder.0.m.bias"); a.relu = 1; a.a_mode = 2; a.x_lse = zlse_;
        RC(gemm_launch(a, st));
    } else {
    if (!zin) zin = zdec_;
    RC(lin_fwd(zin, V, P("_dvae._decoder.0.m.weight"), P("_dvae._decoder.0.m.bias"), dd0_, 64, BT, 64, V, 1, nullptr, 0, 0.f, 0, st));
    }
    RC(conv_layer_fwd(dd0_, dw_fwd_[0], P("_dvae._decoder.1.m.bias"), dd1_, B, E, E, 3, 64, 1, nullptr, nullptr, st));
    RC(lin_fwd(dd1_, 64, P("_dvae._decoder.2.m.weight"), P("_dvae._decoder.2.m.bias"), dd2_, 64, BT, 64, 64, 1, nullptr, 0, 0.f, 0, st));
    RC(lin_fwd(dd2_, 64, P("_dvae._decoder.3.m.weight"), P("_dvae._decoder.3.m.bias"), dd3_, 64, BT, 64, 64, 1, nullptr, 0, 0.f, 0, st));
    RC(lin_fwd(dd3_, 64, P("_dvae._decoder.4.m.weight"), P("_dvae._decoder.4.m.bias"), dd4_, 256, BT, 256, 64, 1, nullptr, 0, 0.f, 0, st));
    RC(pixel_shuffle_launch(dd4_, ps1_, B, E, E, 64, 1, nullptr, st));
    RC(conv_layer_fwd(ps1_, dw_fwd_[1], P("_dvae._decoder.6.m.bias"), dd6_, B, 2 * E, 2 * E, 3, 64, 1, nullptr, nullptr, st));
    RC(lin_fwd(dd6_, 64, P("_dvae._decoder.7.m.weight"), P("_dvae._decoder.7.m.bias"), dd7_, 64, 4 * BT, 64, 64, 1, nullptr, 0, 0.f, 0, st));
    RC(lin_fwd(dd7_, 64, P("_dvae._decoder.8.m.weight"), P("_dvae._decoder.8.m.bias"), dd8_, 64, 4 * BT, 64, 64, 1, nullptr, 0, 0.f, 0, st));
    RC(lin_fwd(dd8_, 64, P("_dvae._decoder.9.m.weight"), P("_dvae._decoder.9.m.bias"), dd9_, 256, 4 * BT, 256, 64, 1, nullptr, 0, 0.f, 0, st));
    RC(pixel_shuffle_launch(dd9_, ps2_, B, 2 * E, 2 * E, 64, 1, nullptr, st));
    RC(lin_fwd(ps2_, 64, P("_dvae._decoder.11.weight"), P("_dvae._decoder.11.bias"), recon_, 4, BN, ch, 64, 0, nullptr, 0, 0.f, 0, st));
    RC(mse_launch(last_.obs, recon_, drecon, metrics_ + (drecon ? 0 : 4), B, ch, S, S, scratch_, scratch_floats_, st));
    return 0;
}

// token embedding + transformer decoder + cross entropy (slate_module.py:141-156, transformer.py)
int SlateModel::fwd_decoder(hipStream_t st, bool with_ce) {
    const int B = last_.B;
    const long long BT = (long long)B * T;
    const float p = pdrop_;
    const float scale = 1.0f / sqrtf((float)DH);
    RC(lin_fwd(slots_, D, P("_slotproj.weight"), nullptr, mem_, d, (long long)B * K, d, D, 0, nullptr, 0, 0.f, 0, st));
    RC(embed_fwd_launch(tokens_, P("_dict.dictionary.weight"), P("_bos_token._bos_token"), P("_z_pos.pe"), emb_, B, T, d, p, last_.seed, st));
    bool fold_pending = false;
    if (xattn_) {       // per-image cross-attention operands of every block from the projected slots: one launch
        XaFoldHost f;
        f.mem = mem_; f.B = B; f.K = K; f.d = d; f.h = NH; f.nblk = NB;
        OCRL_REQUIRE(NB <= 8, "more than 8 decoder blocks: set OCRL_XATTN=0");
        for (int b = 0; b < NB; ++b) {
            const std::string pre = fmt("_tfdec.blocks.%d.encoder_decoder_attn.", b);
            f.Wq[b] = P(pre + "proj_q.weight"); f.Wk[b] = P(pre + "proj_k.weight"); f.Wv[b] = P(pre + "proj_v.weight"); f.Wo[b] = P(pre + "proj_o.weight");
            f.ck[b] = blk_[b].ck; f.cv[b] = blk_[b].cv; f.Ab[b] = blk_[b].xaAb; f.AbT[b] = blk_[b].xaAbT; f.Vo[b] = blk_[b].xaVo; f.VoT[b] = blk_[b].xaVoT;
        }
        // it only needs the projected slots and is first consumed by block 0's cross attention, after the embedding, a LayerNorm, the
        // q|k|v projection and the self attention: on the (idle) weight-gradient stream it runs beside them
        if (side2_) {
            hipEvent_t ev = ev_dw_[ev_dw_next_++ & 7];
            OCRL_HIP(hipEventRecord(ev, st));
            OCRL_HIP(hipStreamWaitEvent(side2_, ev, 0));
            RC(xattn_fold_fwd_launch(f, side2_));
            OCRL_HIP(hipEventRecord(ev_join2_, side2_));
            fold_pending = true;
        } else RC(xattn_fold_fwd_launch(f, st));
    }
    const float* xin = emb_;
    for (int b = 0; b < NB; ++b) {
        Blk& k = blk_[b];
        const std::string pre = fmt("_tfdec.blocks.%d.", b);
        const unsigned site = SITE_BLK_BASE + 8 * b;
        RC(layernorm_fwd_launch(xin, P(pre + "self_attn_layer_norm.weight"), P(pre + "self_attn_layer_norm.bias"), k.ln1, k.ln1_mean, k.ln1_rstd, BT, d, st));
        const float* res = (b == 0) ? k.ln1 : xin;      // block 0 normalises the residual stream itself (transformer.py:175-178)
        // proj_q / proj_k / proj_v are adjacent in the flat buffer: one [3d, d] weight, one GEMM
        RC(lin_fwd(k.ln1, d, P(pre + "self_attn.proj_q.weight"), nullptr, k.q, 3 * d, BT, 3 * d, d, 0, nullptr, 0, 0.f, 0, st));
        {   // causal self attention, flash style (scores never leave the chip)
            AttnArgs a;
            a.q = k.q; a.k = k.k; a.v = k.v; a.o = k.ao; a.lse = k.lse; a.B = B; a.T = T; a.d = d; a.h = NH; a.ld = 3 * d;
            a.p = p; a.seed = last_.seed; a.site = site + 0;
            RC(attn_launch(a, 0, st));
        }
        RC(lin_fwd(k.ao, d, P(pre + "self_attn.proj_o.weight"), nullptr, k.x1, d, BT, d, d, 0, res, d, p, site + 1, st));
        // cross attention to the projected slots
        RC(layernorm_fwd_launch(k.x1, P(pre + "encoder_decoder_attn_layer_norm.weight"), P(pre + "encoder_decoder_attn_layer_norm.bias"), k.ln2, k.ln2_mean, k.ln2_rstd, BT, d, st));
        if (xattn_) {       // folded form: scores, soft-max, dropout, output, dropout and the residual add in one launch (xattn.hip)
            if (fold_pending) { OCRL_HIP(hipStreamWaitEvent(st, ev_join2_, 0)); fold_pending = false; }
            XaHost hx;
            hx.x = k.ln2; hx.resid = k.x1; hx.y = k.x2; hx.P = k.cP; hx.Ab = k.xaAb; hx.AbT = k.xaAbT; hx.Vo = k.xaVo; hx.VoT = k.xaVoT;
            hx.B = B; hx.T = T; hx.K = K; hx.d = d; hx.h = NH; hx.p = p; hx.seed = last_.seed; hx.site_p = site + 2; hx.site_o = site + 3;
            RC(xattn_launch(hx, 0, st));
        } else {
        RC(lin_fwd(k.ln2, d, P(pre + "encoder_decoder_attn.proj_q.weight"), nullptr, k.cq, d, BT, d, d, 0, nullptr, 0, 0.f, 0, st));
        RC(lin_fwd(mem_, d, P(pre + "encoder_decoder_attn.proj_k.weight"), nullptr, k.ck, d, (long long)B * K, d, d, 0, nullptr, 0, 0.f, 0, st));
        RC(lin_fwd(mem_, d, P(pre + "encoder_decoder_attn.proj_v.weight"), nullptr, k.cv, d, (long long)B * K, d, d, 0, nullptr, 0, 0.f, 0, st));
        RC(cross_attn_fwd_launch(k.cq, k.ck, k.cv, k.cao, k.cP, B, T, K, d, NH, p, last_.seed, site + 2, st));
        RC(lin_fwd(k.cao, d, P(pre + "encoder_decoder_attn.proj_o.weight"), nullptr, k.x2, d, BT, d, d, 0, k.x1, d, p, site + 3, st));
        }
        // feed forward
        RC(layernorm_fwd_launch(k.x2, P(pre + "ffn_layer_norm.weight"), P(pre + "ffn_layer_norm.bias"), k.ln3, k.ln3_mean, k.ln3_rstd, BT, d, st));
        RC(lin_fwd(k.ln3, d, P(pre + "ffn.0.weight"), P(pre + "ffn.0.bias"), k.f1, 4 * d, BT, 4 * d, d, 1, nullptr, 0, 0.f, 0, st));
        RC(lin_fwd(k.f1, 4 * d, P(pre + "ffn.2.weight"), P(pre + "ffn.2.bias"), k.x3, d, BT, d, 4 * d, 0, k.x2, d, p, site + 4, st));
        xin = k.x3;
    }
    RC(layernorm_fwd_launch(xin, P("_tfdec.layer_norm.weight"), P("_tfdec.layer_norm.bias"), lnf_, lnf_mean_, lnf_rstd_, BT, d, st));
    if (with_ce) {
        // vocabulary head with the cross-entropy statistics in its epilogue (ocrs/slate/slate_module.py:150-156): pred_ keeps the
        // logits; the gradient (softmax - onehot) / B is rebuilt from pred_ and ce_lse while the backward products stage it
        GemmArgs a;
        a.A = lnf_; a.B = P("_out.weight"); a.C = pred_; a.M = (int)BT; a.N = V; a.K = d; a.lda = d; a.ldb = d; a.ldc = V;
        a.epi_mode = 1; a.stat = cestat_;
        RC(gemm_launch(a, st));
        RC(softmax_stat_combine_launch(cestat_, gemm_stat_segments(V), BT, celse_, nullptr, nullptr, nullptr, pred_, V, tokens_, metrics_ + 1, 1.0f / B,
                                       cepart_, (size_t)(BT / 16 + 16), st));
    } else RC(lin_fwd(lnf_, d, P("_out.weight"), nullptr, pred_, V, BT, V, d, 0, nullptr, 0, 0.f, 0, st));
    return 0;
}

int SlateModel::forward(const StepInputs& in, hipStream_t st) {
    OCRL_REQUIRE(p_ && ws_, "forward: model not bound");
    OCRL_REQUIRE(in.B >= 1 && in.B <= Bmax, "forward: batch %d outside [1,%d]", in.B, Bmax);
    OCRL_REQUIRE(in.obs && in.tau > 0.f, "forward: bad inputs");
    last_ = in;
    have_enc_ = false;
    pdrop_ = in.train ? cfg.dropout : 0.f;
    RC(pack_weights(st));
    if (cfg.use_bcdec) {      // slate_module.py:218-225: loss = mse of the broadcast-decoder reconstruction
        RC(fwd_encoder(in, st));
        RC(pack_bcdec(st));
        RC(fwd_bcdec(st));
        RC(fill_launch(metrics_ + 1, 1, 0.f, st));
        RC(copy_launch(metrics_ + 0, metrics_ + 2, 1, st));
        have_fwd_ = true;
        return 0;
    }
    // the dVAE branch (tokens, reconstruction loss) and the CNN encoder + slot attention are independent until the decoder:
    // the dVAE runs on the side stream, filling the CUs the one-workgroup-per-image slot-attention kernel leaves idle
    if (side_ && overlap_mode_ >= 2) {
        // forks the dVAE forward right before the slot-attention launch (modes 2, 3), at the start of the step (4) or after the convolutions (5)
        RC(fwd_encoder(in, st, overlap_mode_ == 4 ? 2 : (overlap_mode_ == 5 ? 3 : 1)));
        if (overlap_mode_ >= 3 && !getenv("OCRL_TOKENS_LATE")) {
            // wait for the tokens only; the rest of the dVAE branch (its decoder, the reconstruction loss) overlaps the transformer decoder
            OCRL_HIP(hipStreamWaitEvent(st, ev_tokens_, 0));
            RC(fwd_decoder(st));
            RC(join_side(st));
            RC(copy_launch(metrics_ + 0, metrics_ + 2, 1, st));
            RC(axpy_launch(metrics_ + 1, metrics_ + 2, 1, 1.f, st));      // loss = dvae_mse + cross_entropy
            have_fwd_ = true;
            return 0;
        }
        RC(join_side(st));
    } else if (side_) {
        RC(fork_side(st));
        std::swap(scratch_, scratch2_);
        const int rc = fwd_dvae(in, side_);
        std::swap(scratch_, scratch2_);
        RC(rc);
        RC(fwd_encoder(in, st));
        RC(join_side(st));
    } else {
        RC(fwd_dvae(in, st));
        RC(fwd_encoder(in, st));
    }
    RC(fwd_decoder(st));
    RC(copy_launch(metrics_ + 0, metrics_ + 2, 1, st));
    RC(axpy_launch(metrics_ + 1, metrics_ + 2, 1, 1.f, st));      // loss = dvae_mse + cross_entropy
    have_fwd_ = true;
    return 0;
}

int SlateModel::encode(const StepInputs& in, hipStream_t st) {
    OCRL_REQUIRE(p_ && ws_, "encode: model not bound");
    OCRL_REQUIRE(in.B >= 1 && in.B <= Bmax && in.obs, "encode: bad inputs");
    last_ = in;
    pdrop_ = 0.f;
    have_fwd_ = false;
    have_enc_ = true;
    // measured (tools/bench_encode.py, B = 1 / 8 / 32 at 64x64): 0.457 / 0.483 / 1.006 ms replayed against 0.450 / 0.473 / 0.988 ms eager -- the
    // call is bound by its chain of ~30 dependent small kernels on the GPU, not by the host's launches, so the replay is opt-in
    if (enc_graph_mode_ < 0) { const char* e = getenv("OCRL_ENCODE_GRAPH"); enc_graph_mode_ = e ? atoi(e) : 0; }
    const bool graph = enc_graph_mode_ && in.B <= 32 && !in.noise_slots && obs_stage_;
    // frozen weights (ocrl_slate_freeze_weights: serving with a pre-trained encoder): the derived weight images -- convolution packs,
    // slot-attention block, position map; 8 launches, ~45 us of a 0.47 ms call at B = 1 -- are built once
    const bool need_pack = !(frozen_ && packs_valid_);
    struct LowLat { int& f; LowLat(int& x) : f(x) { f = 1; } ~LowLat() { f = 0; } } lowlat(conv_lowlat_);      // for every fwd_encoder below
    if (!graph) {
        if (need_pack) RC(pack_weights(st, true));
        packs_valid_ = frozen_;
        return fwd_encoder(in, st);
    }
    EncGraph& eg = enc_graphs_[in.B];
    if (!eg.warm || need_pack) {            // first call at this batch size runs eagerly: one-time kernel attributes are set outside the capture
        eg.warm = 1;
        if (need_pack) RC(pack_weights(st, true));
        packs_valid_ = frozen_;
        return fwd_encoder(in, st);
    }
    const size_t bytes = (size_t)in.B * cfg.obs_channels * N * sizeof(float);
    OCRL_HIP(hipMemcpyAsync(obs_stage_, in.obs, bytes, hipMemcpyDeviceToDevice, st));
    RC(store_u64_launch(seed_dev_, in.seed, st));
    if (!eg.exec) {
        StepInputs gi = in;
        gi.obs = obs_stage_; gi.seed_dev = seed_dev_;
        hipGraph_t g = nullptr;
        // captured on a stream of its own (the caller's may be the legacy default stream, which cannot capture); replayed on the caller's
        if (!cap_) OCRL_HIP(hipStreamCreateWithFlags(&cap_, hipStreamNonBlocking));
        OCRL_HIP(hipStreamBeginCapture(cap_, hipStreamCaptureModeThreadLocal));
        int rc = frozen_ ? 0 : pack_weights(cap_, true);
        if (!rc) rc = fwd_encoder(gi, cap_);
        const hipError_t ce = hipStreamEndCapture(cap_, &g);
        if (rc) { if (g) (void)hipGraphDestroy(g); return rc; }
        OCRL_HIP(ce);
        OCRL_HIP(hipGraphInstantiate(&eg.exec, g, nullptr, nullptr, 0));
        OCRL_HIP(hipGraphDestroy(g));
    }
    OCRL_HIP(hipGraphLaunch(eg.exec, st));
    return 0;
}

// SLATE_Module._gen_imgs (slate_module.py:163-179): greedy autoregressive token decode from the projected slots, then the dVAE
// decoder.  The reference re-runs the whole decoder on the growing prefix for every token (O(T^2) decoder rows); the causal mask makes
// the rows of earlier positions independent of later tokens, so here every step pushes ONE new row per image through the blocks and
// attends to the keys / values of the earlier rows kept in the fused q|k|v buffer of each block (KV cache): T steps of B rows instead
// of T passes of up to B*T rows.  Clobbers the decoder activations: no backward() afterwards.  metrics[4] = mse of the generated image.
// Fine-tuning the encoder through a downstream loss (poolings/base.py:53-55, learn_downstream_loss: the slots reach the pooling head
// undetached): d loss / d slots of the last encode() -> gradients of the CNN encoder, the positional embedding and the slot-attention
// module.  Every other tensor of the flat gradient buffer is zero and is skipped by the next clip_adam(), as torch's Adam skips
// parameters whose .grad is None.
void SlateModel::clear_encode_graphs() {
    for (auto& kv : enc_graphs_) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
    enc_graphs_.clear();
}

int SlateModel::encode_backward(const float* dslots, hipStream_t st) {
    OCRL_REQUIRE(have_enc_, "encode_backward: call encode first (its activations are what is differentiated)");
    OCRL_REQUIRE(!frozen_, "encode_backward: the weights are frozen (ocrl_slate_freeze_weights): encode() kept no activations");
    OCRL_REQUIRE(g_ && dslots, "encode_backward: no gradient buffer bound / null dslots");
    RC(fill_launch(g_, flat_size_, 0.f, st));
    RC(copy_launch(dslots, gslots_, (long long)last_.B * K * D, st));
    RC(bwd_encoder(st));
    have_enc_ = false;
    enc_only_grads_ = true;
    return 0;
}

int SlateModel::generate(hipStream_t st) {
    OCRL_REQUIRE(!cfg.use_bcdec, "generate: the autoregressive decoder is not part of the use_bcdec configuration");
    OCRL_REQUIRE(last_.B > 0 && last_.obs, "generate: run forward or encode first");
    have_enc_ = false;
    const int B = last_.B;
    const long long BK = (long long)B * K;
    pdrop_ = 0.f;
    have_fwd_ = false;
    OCRL_HIP(hipMemsetAsync(tokens_, 0, sizeof(int) * (size_t)B * T, st));
    // slot side, once: projected slots and every block's cross-attention keys / values
    RC(lin_fwd(slots_, D, P("_slotproj.weight"), nullptr, mem_, d, BK, d, D, 0, nullptr, 0, 0.f, 0, st));
    for (int b = 0; b < NB; ++b) {
        const std::string pre = fmt("_tfdec.blocks.%d.", b);
        RC(lin_fwd(mem_, d, P(pre + "encoder_decoder_attn.proj_k.weight"), nullptr, blk_[b].ck, d, BK, d, d, 0, nullptr, 0, 0.f, 0, st));
        RC(lin_fwd(mem_, d, P(pre + "encoder_decoder_attn.proj_v.weight"), nullptr, blk_[b].cv, d, BK, d, d, 0, nullptr, 0, 0.f, 0, st));
    }
    float* x0 = emb_;           // [B, d] rows of the current step (the first rows of the full-size training buffers)
    for (int t = 0; t < T; ++t) {
        RC(embed_step_launch(tokens_, P("_dict.dictionary.weight"), P("_bos_token._bos_token"), P("_z_pos.pe"), x0, B, T, d, t, st));
        const float* xin = x0;
        for (int b = 0; b < NB; ++b) {
            Blk& k = blk_[b];
            const std::string pre = fmt("_tfdec.blocks.%d.", b);
            RC(layernorm_fwd_launch(xin, P(pre + "self_attn_layer_norm.weight"), P(pre + "self_attn_layer_norm.bias"), k.ln1, k.ln1_mean, k.ln1_rstd, B, d, st));
            const float* res = (b == 0) ? k.ln1 : xin;
            // q | k | v of the new row go straight into row t of each image's cache: output row m lives at (m*T + t) * 3d
            RC(lin_fwd(k.ln1, d, P(pre + "self_attn.proj_q.weight"), nullptr, k.q + (size_t)t * 3 * d, T * 3 * d, B, 3 * d, d, 0, nullptr, 0, 0.f, 0, st));
            RC(decode_attn_launch(k.q, k.ao, B, T, t, d, NH, 3 * d, st));
            RC(lin_fwd(k.ao, d, P(pre + "self_attn.proj_o.weight"), nullptr, k.x1, d, B, d, d, 0, res, d, 0.f, 0, st));
            RC(layernorm_fwd_launch(k.x1, P(pre + "encoder_decoder_attn_layer_norm.weight"), P(pre + "encoder_decoder_attn_layer_norm.bias"), k.ln2, k.ln2_mean, k.ln2_rstd, B, d, st));
            RC(lin_fwd(k.ln2, d, P(pre + "encoder_decoder_attn.proj_q.weight"), nullptr, k.cq, d, B, d, d, 0, nullptr, 0, 0.f, 0, st));
            RC(cross_attn_fwd_launch(k.cq, k.ck, k.cv, k.cao, k.cP, B, 1, K, d, NH, 0.f, 0, 0, st));
            RC(lin_fwd(k.cao, d, P(pre + "encoder_decoder_attn.proj_o.weight"), nullptr, k.x2, d, B, d, d, 0, k.x1, d, 0.f, 0, st));
            RC(layernorm_fwd_launch(k.x2, P(pre + "ffn_layer_norm.weight"), P(pre + "ffn_layer_norm.bias"), k.ln3, k.ln3_mean, k.ln3_rstd, B, d, st));
            RC(lin_fwd(k.ln3, d, P(pre + "ffn.0.weight"), P(pre + "ffn.0.bias"), k.f1, 4 * d, B, 4 * d, d, 1, nullptr, 0, 0.f, 0, st));
            RC(lin_fwd(k.f1, 4 * d, P(pre + "ffn.2.weight"), P(pre + "ffn.2.bias"), k.x3, d, B, d, 4 * d, 0, k.x2, d, 0.f, 0, st));
            xin = k.x3;
        }
        RC(layernorm_fwd_launch(xin, P("_tfdec.layer_norm.weight"), P("_tfdec.layer_norm.bias"), lnf_, lnf_mean_, lnf_rstd_, B, d, st));
        RC(lin_fwd(lnf_, d, P("_out.weight"), nullptr, pred_, V, B, V, d, 0, nullptr, 0, 0.f, 0, st));
        RC(argmax_pos_launch(pred_, tokens_, B, T, V, t, st, 1));
    }
    RC(onehot_launch(tokens_, z_, (long long)B * T, V, st));
    RC(dvae_decode(B, nullptr, st, z_));
    return 0;
}

// ---------------------------------------------------------------------------------------------
int SlateModel::bwd_decoder(hipStream_t st) {
    const int B = last_.B;
    const long long BT = (long long)B * T, BK = (long long)B * K;
    const float p = pdrop_;
    const float scale = 1.0f / sqrtf((float)DH);
    // Weight-gradient products on a side stream (OCRL_DW_SIDE): `dw_sync` makes the side stream wait for everything the main stream has
    // enqueued so far, `dw` enqueues one product there (with the side stream's own split-k scratch).  The operands such a product reads
    // are never rewritten by the main stream during this backward: saved activations, and the per-block gradient temporaries bg_[b].
    const bool dws = dw_mode_ != 0 && side_ != nullptr;
    hipStream_t sw = dws ? (dw_mode_ == 2 ? side2_ : side_) : st;
    float* const sw_scratch = dw_mode_ == 2 ? scratch3_ : scratch2_;
    auto dw_sync = [&]() -> int {
        if (!dws) return 0;
        hipEvent_t ev = ev_dw_[ev_dw_next_++ & 7];
        OCRL_HIP(hipEventRecord(ev, st));
        OCRL_HIP(hipStreamWaitEvent(sw, ev, 0));
        return 0;
    };
    auto dw = [&](const float* dy, int ld_dy, const float* x, int ldx, float* dW, float* db, long long M, int N_out, int K_in, Xf xf = Xf()) -> int {
        float* keep = scratch_;
        if (dws) scratch_ = sw_scratch;
        const int rc = lin_bwd_w(dy, ld_dy, x, ldx, dW, db, M, N_out, K_in, 1.f, sw, Drop(), xf);
        scratch_ = keep;
        return rc;
    };
    auto on_sw = [&](const std::function<int()>& body) -> int {      // run `body` (launches on sw) with the side stream's scratch
        float* keep = scratch_;
        if (dws) scratch_ = sw_scratch;
        const int rc = body();
        scratch_ = keep;
        return rc;
    };
    // output head: d loss / d pred = (softmax(pred_) - onehot(tokens)) / B, rebuilt from the logits as both products stage their A tiles
    Xf ce; ce.a_mode = 3; ce.lse = celse_; ce.tok = tokens_; ce.scale = 1.0f / last_.B;
    RC(dw_sync());
    RC(dw(pred_, V, lnf_, d, G("_out.weight"), nullptr, BT, V, d, ce));
    RC(lin_bwd_x(pred_, V, P("_out.weight"), gt1_, d, BT, V, d, nullptr, 0, nullptr, 0, st, Drop(), ce));
    const float* xlast = blk_[NB - 1].x3;
    RC(layernorm_bwd_launch(gt1_, xlast, lnf_mean_, lnf_rstd_, P("_tfdec.layer_norm.weight"), gx_, G("_tfdec.layer_norm.weight"), BT, d, 0, 0,
                            scratch_, scratch_floats_, st));
    RC(fill_launch(gmem_, BK * d, 0.f, st));
    for (int b = NB - 1; b >= 0; --b) {
        Blk& k = blk_[b];
        const BlkG& q = bg_[dws ? b : 0];
        const std::string pre = fmt("_tfdec.blocks.%d.", b);
        const unsigned site = SITE_BLK_BASE + 8 * b;
        const float* xin = (b == 0) ? emb_ : blk_[b - 1].x3;
        // ---- feed forward:  x3 = x2 + drop(W2 relu(W1 ln3 + b1) + b2)
        // the gradient entering a residual branch is dropout-backward(gx) of the branch's site: applied once into a buffer of its own and
        // fed to both the weight-gradient and the input-gradient product (drawing the mask inside the products made each 40-90 % slower).
        // With the weight gradients on the side stream the copy is also taken when there is no dropout: gx_ itself moves on.
        const float* gd = gx_;
        auto drop_gx = [&](unsigned s_, float* into) -> int {
            gd = gx_;
            if (p > 0.f || dws) { RC(dropout_apply_launch(gx_, into, BT * d, p, last_.seed, s_, st)); gd = into; }
            return 0;
        };
        RC(drop_gx(site + 4, q.gbr[0]));
        RC(lin_bwd_x(gd, d, P(pre + "ffn.2.weight"), q.gf1, 4 * d, BT, d, 4 * d, k.f1, 4 * d, nullptr, 0, st));
        RC(dw_sync());
        RC(dw(gd, d, k.f1, 4 * d, G(pre + "ffn.2.weight"), G(pre + "ffn.2.bias"), BT, d, 4 * d));
        RC(dw(q.gf1, 4 * d, k.ln3, d, G(pre + "ffn.0.weight"), G(pre + "ffn.0.bias"), BT, 4 * d, d));
        RC(lin_bwd_x(q.gf1, 4 * d, P(pre + "ffn.0.weight"), gt1_, d, BT, 4 * d, d, nullptr, 0, nullptr, 0, st));
        RC(layernorm_bwd_launch(gt1_, k.x2, k.ln3_mean, k.ln3_rstd, P(pre + "ffn_layer_norm.weight"), gx_, G(pre + "ffn_layer_norm.weight"), BT, d, 1, 0,
                                scratch_, scratch_floats_, st));
        // ---- cross attention
        RC(drop_gx(site + 3, q.gbr[1]));
        const float* gd_co = gd;
        if (xattn_) {
            // folded backward (xattn.hip): d probabilities, soft-max backward and d LN(x) in one launch; the per-image sums
            // d Vo_b = Pd^T d out and d A_b = dS^T LN(x) are two batched products; a per-image kernel takes them back to the projection
            // weights (partials summed over the images in a fixed order) and to d ck / d cv
            const int NC = NH * xattn_kp(K, NH);
            XaHost hx;
            hx.x = k.ln2; hx.y = gt1_; hx.P = k.cP; hx.Ab = k.xaAb; hx.AbT = k.xaAbT; hx.Vo = k.xaVo; hx.VoT = k.xaVoT; hx.gd = gd_co; hx.Pd = q.xaPd; hx.dS = q.xaDs;
            hx.B = B; hx.T = T; hx.K = K; hx.d = d; hx.h = NH; hx.p = p; hx.seed = last_.seed; hx.site_p = site + 2; hx.site_o = site + 3;
            RC(xattn_launch(hx, 1, st));                                                                                  // gt1 = d ln2
            // everything below feeds weight gradients and d mem only: it runs on the weight-gradient stream (dw_sync / on_sw), in order
            RC(dw_sync());
            RC(on_sw([&]() -> int {
                const int KP = xattn_kp(K, NH), dh = DH;
                const float* Wq = P(pre + "encoder_decoder_attn.proj_q.weight");
                const float* Wo = P(pre + "encoder_decoder_attn.proj_o.weight");
                GemmArgs ga;      // per-image sums over the tokens: d Vo_b = Pd^T d out,  d A_b = dS^T LN(x)      [NC, d] each
                ga.M = NC; ga.N = d; ga.K = T; ga.lda = NC; ga.ldb = d; ga.ldc = d; ga.akc = 0; ga.bkc = 0; ga.batch = B;
                ga.sA = (long long)T * NC; ga.sB = (long long)T * d; ga.sC = (long long)NC * d;
                ga.A = q.xaPd; ga.B = gd_co; ga.C = xa_dVo_;
                RC(gemm_launch(ga, sw));
                ga.A = q.xaDs; ga.B = k.ln2; ga.C = xa_dAb_;
                RC(gemm_launch(ga, sw));
                // back through the fold, batched over (image, head):
                GemmArgs gb;      // d ck[k, h dh + j] = scale sum_e dA_b[(h,k), e] Wq[h dh + j, e]
                gb.M = K; gb.N = dh; gb.K = d; gb.akc = 1; gb.bkc = 1; gb.lda = d; gb.ldb = d; gb.ldc = d; gb.batch = B * NH; gb.batch_inner = NH; gb.alpha = scale;
                gb.A = xa_dAb_; gb.sA = (long long)NC * d; gb.sAi = (long long)KP * d;
                gb.B = Wq; gb.sB = 0; gb.sBi = (long long)dh * d;
                gb.C = gck_; gb.sC = (long long)K * d; gb.sCi = dh;
                RC(gemm_launch(gb, sw));
                GemmArgs gc;      // d cv[k, h dh + j] = sum_o dVo_b[(h,k), o] Wo[o, h dh + j]
                gc.M = K; gc.N = dh; gc.K = d; gc.akc = 1; gc.bkc = 0; gc.lda = d; gc.ldb = d; gc.ldc = d; gc.batch = B * NH; gc.batch_inner = NH;
                gc.A = xa_dVo_; gc.sA = (long long)NC * d; gc.sAi = (long long)KP * d;
                gc.B = Wo; gc.sB = 0; gc.sBi = dh;
                gc.C = gcv_; gc.sC = (long long)K * d; gc.sCi = dh;
                RC(gemm_launch(gc, sw));
                GemmArgs gq;      // this image's d Wq[h dh + j, e] = scale sum_k ck[k, h dh + j] dA_b[(h,k), e]
                gq.M = dh; gq.N = d; gq.K = K; gq.akc = 0; gq.bkc = 0; gq.lda = d; gq.ldb = d; gq.ldc = d; gq.batch = B * NH; gq.batch_inner = NH; gq.alpha = scale;
                gq.A = k.ck; gq.sA = (long long)K * d; gq.sAi = dh;
                gq.B = xa_dAb_; gq.sB = (long long)NC * d; gq.sBi = (long long)KP * d;
                gq.C = xa_pq_; gq.sC = (long long)d * d; gq.sCi = (long long)dh * d;
                RC(gemm_launch(gq, sw));
                GemmArgs go;      // this image's d Wo[o, h dh + j] = sum_k dVo_b[(h,k), o] cv[k, h dh + j]
                go.M = d; go.N = dh; go.K = K; go.akc = 0; go.bkc = 0; go.lda = d; go.ldb = d; go.ldc = d; go.batch = B * NH; go.batch_inner = NH;
                go.A = xa_dVo_; go.sA = (long long)NC * d; go.sAi = (long long)KP * d;
                go.B = k.cv; go.sB = (long long)K * d; go.sBi = dh;
                go.C = xa_po_; go.sC = (long long)d * d; go.sCi = dh;
                RC(gemm_launch(go, sw));
                RC(colsum_launch(xa_pq_, (long long)d * d, G(pre + "encoder_decoder_attn.proj_q.weight"), B, d * d, 0, 1.f, scratch_, scratch_floats_, sw));
                RC(colsum_launch(xa_po_, (long long)d * d, G(pre + "encoder_decoder_attn.proj_o.weight"), B, d * d, 0, 1.f, scratch_, scratch_floats_, sw));
                return 0;
            }));
        } else {
        RC(lin_bwd_x(gd, d, P(pre + "encoder_decoder_attn.proj_o.weight"), gt1_, d, BT, d, d, nullptr, 0, nullptr, 0, st));   // d cao
        RC(cross_attn_bwd_launch(gt1_, k.cq, k.ck, k.cv, k.cP, q.gt2, gck_, gcv_, B, T, K, d, NH, p, last_.seed, site + 2, scratch_, scratch_floats_, st));   // gt2 = d cq
        RC(dw_sync());
        RC(dw(gd_co, d, k.cao, d, G(pre + "encoder_decoder_attn.proj_o.weight"), nullptr, BT, d, d));
        RC(dw(q.gt2, d, k.ln2, d, G(pre + "encoder_decoder_attn.proj_q.weight"), nullptr, BT, d, d));
        RC(lin_bwd_x(q.gt2, d, P(pre + "encoder_decoder_attn.proj_q.weight"), gt1_, d, BT, d, d, nullptr, 0, nullptr, 0, st));   // d ln2
        }
        {   // slot-side projections k / v: their weight gradients and d mem (accumulated over the blocks)
            hipStream_t sx = xattn_ ? sw : st;
            auto body = [&]() -> int {
                RC(lin_bwd_w(gck_, d, mem_, d, G(pre + "encoder_decoder_attn.proj_k.weight"), nullptr, BK, d, d, 1.f, sx));
                RC(lin_bwd_w(gcv_, d, mem_, d, G(pre + "encoder_decoder_attn.proj_v.weight"), nullptr, BK, d, d, 1.f, sx));
                RC(lin_bwd_x(gck_, d, P(pre + "encoder_decoder_attn.proj_k.weight"), gmem_, d, BK, d, d, nullptr, 0, gmem_, d, sx));
                RC(lin_bwd_x(gcv_, d, P(pre + "encoder_decoder_attn.proj_v.weight"), gmem_, d, BK, d, d, nullptr, 0, gmem_, d, sx));
                return 0;
            };
            if (xattn_) RC(on_sw(body)); else RC(body());
        }
        RC(layernorm_bwd_launch(gt1_, k.x1, k.ln2_mean, k.ln2_rstd, P(pre + "encoder_decoder_attn_layer_norm.weight"), gx_,
                                G(pre + "encoder_decoder_attn_layer_norm.weight"), BT, d, 1, 0, scratch_, scratch_floats_, st));
        // ---- causal self attention
        RC(drop_gx(site + 1, q.gbr[2]));
        const float* gd_o = gd;
        RC(lin_bwd_x(gd, d, P(pre + "self_attn.proj_o.weight"), gt1_, d, BT, d, d, nullptr, 0, nullptr, 0, st));   // gt1 = d ao
        {
            AttnArgs a;
            a.q = k.q; a.k = k.k; a.v = k.v; a.o = k.ao; a.lse = k.lse; a.B = B; a.T = T; a.d = d; a.h = NH; a.ld = 3 * d;
            a.p = p; a.seed = last_.seed; a.site = site + 0;
            a.dO = gt1_; a.dq = q.gqkv; a.dk = q.gqkv + d; a.dv = q.gqkv + 2 * d; a.delta = attn_delta_;
            RC(attn_launch(a, 1, st));
        }
        // fused [3d, d] weight: dW = [dq|dk|dv]^T ln1 ;  d ln1 = [dq|dk|dv] W
        RC(dw_sync());
        RC(dw(gd_o, d, k.ao, d, G(pre + "self_attn.proj_o.weight"), nullptr, BT, d, d));
        RC(dw(q.gqkv, 3 * d, k.ln1, d, G(pre + "self_attn.proj_q.weight"), nullptr, BT, 3 * d, d));
        RC(lin_bwd_x(q.gqkv, 3 * d, P(pre + "self_attn.proj_q.weight"), gt1_, d, BT, 3 * d, d, nullptr, 0, nullptr, 0, st));      // gt1 = d ln1
        if (b == 0) {
            // ln1 is both the attention input and the residual stream: d ln1_total = gx + gt2, then LN backward to emb
            RC(axpy_launch(gx_, gt1_, BT * d, 1.f, st));
            RC(layernorm_bwd_launch(gt1_, xin, k.ln1_mean, k.ln1_rstd, P(pre + "self_attn_layer_norm.weight"), gx_,
                                    G(pre + "self_attn_layer_norm.weight"), BT, d, 0, 0, scratch_, scratch_floats_, st));
        } else {
            RC(layernorm_bwd_launch(gt1_, xin, k.ln1_mean, k.ln1_rstd, P(pre + "self_attn_layer_norm.weight"), gx_,
                                    G(pre + "self_attn_layer_norm.weight"), BT, d, 1, 0, scratch_, scratch_floats_, st));
        }
    }
    // ---- embedding: gx_ = d emb (after dropout);  dictionary (rows sorted by token, summed in a fixed order), pe / bos (sum over the batch)
    RC(embed_bwd_launch(gx_, tokens_, G("_dict.dictionary.weight"), B, T, V, d, p, last_.seed, scratch_, scratch_floats_, st));
    RC(fill_launch(G("_z_pos.pe"), (long long)(T + 1) * d, 0.f, st));
    RC(colsum_launch(gx_, (long long)T * d, G("_z_pos.pe"), B, T * d, 0, 1.f, scratch_, scratch_floats_, st));
    RC(copy_launch(G("_z_pos.pe"), G("_bos_token._bos_token"), d, st));
    // ---- slot projection (d mem was accumulated on the weight-gradient stream in the folded cross-attention form)
    if (dws && xattn_) {
        hipEvent_t ev = ev_dw_[ev_dw_next_++ & 7];
        OCRL_HIP(hipEventRecord(ev, sw));
        OCRL_HIP(hipStreamWaitEvent(st, ev, 0));
    }
    RC(lin_bwd_w(gmem_, d, slots_, D, G("_slotproj.weight"), nullptr, BK, d, D, 1.f, st));
    RC(lin_bwd_x(gmem_, d, P("_slotproj.weight"), gslots_, D, BK, d, D, nullptr, 0, nullptr, 0, st));
    return 0;
}

int SlateModel::bwd_encoder(hipStream_t st, bool fork_dvae) {
    const int B = last_.B;
    const long long BN = (long long)B * N, R = (long long)B * I * K;
    const SaSave so = sa_save_layout(C, D, H, SH);
    const SaGrad go = sa_grad_layout(C, D, H, SH);
    const std::string sa = "_slotattn.slot_attention.";
    SlotAttnArgs a;
    a.B = B; a.N = N; a.C = C; a.K = K; a.D = D; a.H = H; a.I = I; a.NH = SH; a.eps = 1e-8f; a.scale = 1.0f / sqrtf((float)(D / SH));
    a.x = x_; a.wts = sa_wts_; a.save = sa_save_; a.dslots = gslots_; a.dx = gA_; a.dslots0 = gslots0_; a.grows = sa_grows_; a.g_small = sa_small_;
    a.xchg = sa_xchg_; a.parts = sa_parts_;
    if (fork_dvae) {
        RC(fork_side(st));
        std::swap(scratch_, scratch2_);
        const int rc = bwd_dvae(side_);
        std::swap(scratch_, scratch2_);
        RC(rc);
    }
    RC(slot_attn_launch(a, 1, st));
    // Everything below that only consumes the slot-attention backward's outputs (gradient rows, dx) and feeds no later kernel of the
    // step -- the slot-side weight gradients, the LayerNorm partials, the slot-initialisation gradients and the weight gradient of the
    // input MLP's second layer -- goes to the side stream; the main stream continues with the input-gradient chain towards the
    // convolutions.  Joined at the end of the backward.
    const bool side_w = side_ && overlap_mode_ >= 3 && !cfg.use_bcdec && !fork_dvae;
    hipStream_t sw = side_w ? side_ : st;
    if (side_w) { RC(fork_side(st)); std::swap(scratch_, scratch2_); }
    auto sa_weight_grads = [&]() -> int {
    // weight gradients: contract the emitted gradient rows with the saved activations over (image, iteration, slot)
    RC(lin_bwd_w(sa_grows_ + go.out, go.ld, sa_save_ + so.hid, so.ld, G(sa + "mlp.2.weight"), G(sa + "mlp.2.bias"), R, D, H, 1.f, sw));
    RC(lin_bwd_w(sa_grows_ + go.hid, go.ld, sa_save_ + so.m, so.ld, G(sa + "mlp.0.weight"), G(sa + "mlp.0.bias"), R, H, D, 1.f, sw));
    RC(lin_bwd_w(sa_grows_ + go.gi, go.ld, sa_save_ + so.u, so.ld, G(sa + "gru.weight_ih"), G(sa + "gru.bias_ih"), R, 3 * D, D, 1.f, sw));
    RC(lin_bwd_w(sa_grows_ + go.gh, go.ld, sa_save_ + so.sprev, so.ld, G(sa + "gru.weight_hh"), G(sa + "gru.bias_hh"), R, 3 * D, D, 1.f, sw));
    RC(lin_bwd_w(sa_grows_ + go.q, go.ld, sa_save_ + so.sn, so.ld, G(sa + "project_q.weight"), nullptr, R, D, D, 1.f, sw));
    for (int h = 0, dh = D / SH; h < SH; ++h) {       // rows of head h of project_v / project_k meet that head's weighted means / folded-query gradients
        RC(lin_bwd_w(sa_grows_ + go.u + h * dh, go.ld, sa_save_ + so.up + h * C, so.ld, G(sa + "project_v.weight") + (size_t)h * dh * C, nullptr, R, dh, C, 1.f, sw));
        RC(lin_bwd_w(sa_save_ + so.q + h * dh, so.ld, sa_grows_ + go.qp + h * C, go.ld, G(sa + "project_k.weight") + (size_t)h * dh * C, nullptr, R, dh, C, a.scale, sw));
    }
    const int SM = 4 * D + 2 * C;
    RC(colsum_launch(sa_small_ + 0, SM, G(sa + "norm_slots.weight"), B, 2 * D, 0, 1.f, scratch_, scratch_floats_, sw));
    RC(colsum_launch(sa_small_ + 2 * D, SM, G(sa + "norm_mlp.weight"), B, 2 * D, 0, 1.f, scratch_, scratch_floats_, sw));
    RC(colsum_launch(sa_small_ + 4 * D, SM, G(sa + "norm_inputs.weight"), B, 2 * C, 0, 1.f, scratch_, scratch_floats_, sw));
    RC(slot_init_bwd_launch(gslots0_, P("_slotattn.slot_log_sigma"), last_.noise_slots, G("_slotattn.slot_mu"), G("_slotattn.slot_log_sigma"),
                            B * K, D, last_.seed, sw));
    // ---- input MLP: x = W2 relu(W0 LN(e4) + b0) + b2 ; gA = dx
    RC(lin_bwd_w(gA_, C, h1_, C, G("_slotattn.mlp.2.weight"), G("_slotattn.mlp.2.bias"), BN, C, C, 1.f, sw));
    // the first convolution's weight gradient is a product with im2col(obs): the patch matrix only needs the observation
    if (side_w) RC(im2col5_launch(obs8_, col0_, BN, S, S, cfg.obs_channels, (25 * cfg.obs_channels + 3) & ~3, sw));
    return 0;
    };
    {
        const int rc = sa_weight_grads();
        if (side_w) std::swap(scratch_, scratch2_);
        RC(rc);
    }
    RC(lin_bwd_x(gA_, C, P("_slotattn.mlp.2.weight"), gB_, C, BN, C, C, h1_, C, nullptr, 0, st));              // gB = d h1 (pre-relu)
    RC(lin_bwd_w(gB_, C, ln0_, C, G("_slotattn.mlp.0.weight"), G("_slotattn.mlp.0.bias"), BN, C, C, 1.f, st));
    // d ln0 goes to gC_ when the side stream may still be reading gA_ (= dx) for the second layer's weight gradient
    float* gL = side_w ? gC_ : gA_;
    RC(lin_bwd_x(gB_, C, P("_slotattn.mlp.0.weight"), gL, C, BN, C, C, nullptr, 0, nullptr, 0, st));            // gL = d ln0
    RC(layernorm_bwd_launch(gL, e4_, ln0_mean_, ln0_rstd_, P("_slotattn.layer_norm.weight"), gB_, G("_slotattn.layer_norm.weight"), BN, C, 0, 0,
                            scratch_, scratch_floats_, st));                                                      // gB = d e4
    // ---- positional embedding (added to every image): d map = sum over images
    RC(colsum_launch(gB_, (long long)N * C, gmap_, B, N * C, 0, 1.f, scratch_, scratch_floats_, st));
    RC(lin_bwd_w(gmap_, C, gridT_, 4, G("_enc_pos.channels_map.weight"), G("_enc_pos.channels_map.bias"), N, C, 4, 1.f, st));
    if (fork_dvae || side_w) RC(join_side(st));          // the side stream has read gA_ (= dx): the convolution chain may reuse it
    if (dw_mode_ == 2 && side2_) {                       // ... and the decoder's weight gradients are done: the convolutions run alone
        OCRL_HIP(hipEventRecord(ev_join2_, side2_));
        OCRL_HIP(hipStreamWaitEvent(st, ev_join2_, 0));
    }
    // ---- CNN encoder, last layer first
    // the last layer's bias gradient is the column sum of the per-position sums gmap_ just taken for the positional embedding (4 MB),
    // not another pass over the [B*N, 64] gradient (537 MB)
    RC(colsum_launch(gmap_, C, G("_enc._encoder.3.bias"), N, C, 0, 1.f, scratch_, scratch_floats_, st));
    RC(conv_layer_wgrad(e3_, gB_, G("_enc._encoder.3.weight"), nullptr, B, S, S, 5, 64, 64, st));
    RC(conv_layer_fwd(gB_, cw_bwd_[3], nullptr, gA_, B, S, S, 5, 64, 0, nullptr, e3_, st));                       // gA = d e3 (pre-relu)
    RC(conv_layer_wgrad(e2_, gA_, G("_enc._encoder.2.m.weight"), G("_enc._encoder.2.m.bias"), B, S, S, 5, 64, 64, st));
    RC(conv_layer_fwd(gA_, cw_bwd_[2], nullptr, gB_, B, S, S, 5, 64, 0, nullptr, e2_, st));
    RC(conv_layer_wgrad(e1_, gB_, G("_enc._encoder.1.m.weight"), G("_enc._encoder.1.m.bias"), B, S, S, 5, 64, 64, st));
    RC(conv_layer_fwd(gB_, cw_bwd_[1], nullptr, gA_, B, S, S, 5, 64, 0, nullptr, e1_, st));
    // first layer (3 input channels): on the conv wgrad kernel its 8-of-64 useful MFMA columns cost 1.5 ms; as
    // dW = dY^T im2col(obs) it is one [64 x 75] split-K product over the B*N pixels
    {
        const int ch = cfg.obs_channels, ldc0 = (25 * ch + 3) & ~3;
        if (!side_w) RC(im2col5_launch(obs8_, col0_, BN, S, S, ch, ldc0, st));
        RC(lin_bwd_w(gA_, 64, col0_, ldc0, dw0p_, G("_enc._encoder.0.m.bias"), BN, 64, ldc0, 1.f, st));
        RC(unpack5_launch(dw0p_, G("_enc._encoder.0.m.weight"), ch, ldc0, st));
    }
    return 0;
}

int SlateModel::bwd_dvae(hipStream_t st) {
    const int B = last_.B;
    const long long BT = (long long)B * T, BN = (long long)B * N;
    const int ch = cfg.obs_channels;
    // ---- output conv 64 -> 3 (padded to 4 columns): dW through a [4,64] scratch
    float* w4 = scratch_ + scratch_floats_ - 1024;      // tail of the scratch region, not touched by split-k / colsum
    {
        GemmArgs a;
        a.A = drecon_; a.B = ps2_; a.C = w4; a.M = 4; a.N = 64; a.K = (int)BN; a.lda = 4; a.ldb = 64; a.ldc = 64; a.akc = 0; a.bkc = 0;
        long long splits = BN / 512; if (splits > 512) splits = 512; if (splits < 1) splits = 1;
        if (splits > 1) {
            a.splitk = (int)splits; a.C = scratch_; a.sCsplit = 256;
            RC(gemm_launch(a, st));
            RC(splitk_reduce_launch(scratch_, w4, 256, (int)splits, 256, 0, st));
        } else RC(gemm_launch(a, st));
        RC(copy_launch(w4, G("_dvae._decoder.11.weight"), ch * 64, st));
        // bias gradient: the padded rows are summed as float4 (the 3-wide scalar form ran one 64-lane column group at 0.05 TB/s)
        RC(colsum_launch(drecon_, 4, w4 + 256, BN, 4, 0, 1.f, scratch_, scratch_floats_ - 1024, st));
        RC(copy_launch(w4 + 256, G("_dvae._decoder.11.bias"), ch, st));
    }
    RC(lin_bwd_x(drecon_, 4, w11p_, gdA_, 64, BN, 4, 64, nullptr, 0, nullptr, 0, st));                            // gdA = d ps2
    RC(pixel_shuffle_launch(gdA_, gdB_, B, 2 * E, 2 * E, 64, 0, dd9_, st));                                        // gdB = d dd9 (pre-relu) [4BT,256]
    RC(lin_bwd_w(gdB_, 256, dd8_, 64, G("_dvae._decoder.9.m.weight"), G("_dvae._decoder.9.m.bias"), 4 * BT, 256, 64, 1.f, st));
    RC(lin_bwd_x(gdB_, 256, P("_dvae._decoder.9.m.weight"), gdA_, 64, 4 * BT, 256, 64, dd8_, 64, nullptr, 0, st));
    RC(lin_bwd_w(gdA_, 64, dd7_, 64, G("_dvae._decoder.8.m.weight"), G("_dvae._decoder.8.m.bias"), 4 * BT, 64, 64, 1.f, st));
    RC(lin_bwd_x(gdA_, 64, P("_dvae._decoder.8.m.weight"), gdB_, 64, 4 * BT, 64, 64, dd7_, 64, nullptr, 0, st));
    RC(lin_bwd_w(gdB_, 64, dd6_, 64, G("_dvae._decoder.7.m.weight"), G("_dvae._decoder.7.m.bias"), 4 * BT, 64, 64, 1.f, st));
    RC(lin_bwd_x(gdB_, 64, P("_dvae._decoder.7.m.weight"), gdA_, 64, 4 * BT, 64, 64, dd6_, 64, nullptr, 0, st));   // gdA = d dd6 (pre-relu)
    RC(conv_layer_wgrad(ps1_, gdA_, G("_dvae._decoder.6.m.weight"), G("_dvae._decoder.6.m.bias"), B, 2 * E, 2 * E, 3, 64, 64, st));
    RC(conv_layer_fwd(gdA_, dw_bwd_[1], nullptr, gdB_, B, 2 * E, 2 * E, 3, 64, 0, nullptr, ps1_, st));             // gdB = d ps1 (relu mask of dd4)
    RC(pixel_shuffle_launch(gdB_, gdA_, B, E, E, 64, 0, nullptr, st));                                             // gdA = d dd4 (pre-relu) [BT,256]
    RC(lin_bwd_w(gdA_, 256, dd3_, 64, G("_dvae._decoder.4.m.weight"), G("_dvae._decoder.4.m.bias"), BT, 256, 64, 1.f, st));
    RC(lin_bwd_x(gdA_, 256, P("_dvae._decoder.4.m.weight"), gdB_, 64, BT, 256, 64, dd3_, 64, nullptr, 0, st));
    RC(lin_bwd_w(gdB_, 64, dd2_, 64, G("_dvae._decoder.3.m.weight"), G("_dvae._decoder.3.m.bias"), BT, 64, 64, 1.f, st));
    RC(lin_bwd_x(gdB_, 64, P("_dvae._decoder.3.m.weight"), gdA_, 64, BT, 64, 64, dd2_, 64, nullptr, 0, st));
    RC(lin_bwd_w(gdA_, 64, dd1_, 64, G("_dvae._decoder.2.m.weight"), G("_dvae._decoder.2.m.bias"), BT, 64, 64, 1.f, st));
    RC(lin_bwd_x(gdA_, 64, P("_dvae._decoder.2.m.weight"), gdB_, 64, BT, 64, 64, dd1_, 64, nullptr, 0, st));       // gdB = d dd1 (pre-relu)
    RC(conv_layer_wgrad(dd0_, gdB_, G("_dvae._decoder.1.m.weight"), G("_dvae._decoder.1.m.bias"), B, E, E, 3, 64, 64, st));
    RC(conv_layer_fwd(gdB_, dw_bwd_[0], nullptr, gdA_, B, E, E, 3, 64, 0, nullptr, dd0_, st));                     // gdA = d dd0 (pre-relu)
    float* dz = zraw_;      // the logits / scores are not needed any more: d raw = d logp is built in their place
    have_scores_ = false;
    if (fused_heads()) {
        Xf zx; zx.b_mode = 2; zx.lse = zlse_;
        RC(lin_bwd_w(gdA_, 64, zraw_, V, G("_dvae._decoder.0.m.weight"), G("_dvae._decoder.0.m.bias"), BT, 64, V, 1.f, st, Drop(), zx));
        // Gumbel soft-max backward in the epilogue of the dz product: d = z (dz - sum_v z_v dz_v) / tau with
        // sum_v z_v dz_v = sum_c g_c (z W^T)_c = sum_c g_c (dd0 - bias)_c  (g = gdA_ is zero where the ReLU of dd0 is closed)
        RC(rowdot_bias64_launch(gdA_, dd0_, P("_dvae._decoder.0.m.bias"), BT, zdot_, st));
        GemmArgs a;
        a.A = gdA_; a.B = P("_dvae._decoder.0.m.weight"); a.C = dz; a.M = (int)BT; a.N = V; a.K = 64; a.lda = 64; a.ldb = V; a.ldc = V; a.bkc = 0;
        a.epi_mode = 3; a.mask = zraw_; a.ldmask = V; a.e_lse = zlse_; a.e_rowvec = zdot_; a.e_scale = 1.0f / last_.tau;
        RC(gemm_launch(a, st));
    } else {
    RC(lin_bwd_w(gdA_, 64, zdec_, V, G("_dvae._decoder.0.m.weight"), G("_dvae._decoder.0.m.bias"), BT, 64, V, 1.f, st));
    RC(lin_bwd_x(gdA_, 64, P("_dvae._decoder.0.m.weight"), dz, V, BT, 64, V, nullptr, 0, nullptr, 0, st));
    // ---- Gumbel softmax + log_softmax backward (row sums of the soft-max gradient vanish, so d raw = d logp)
    RC(softmax_bwd_rows_launch(z_, dz, BT, V, 1.0f / last_.tau, st));
    }
    // ---- encoder
    RC(lin_bwd_w(dz, V, de_[6], 64, G("_dvae._encoder.7.weight"), G("_dvae._encoder.7.bias"), BT, V, 64, 1.f, st));
    RC(lin_bwd_x(dz, V, P("_dvae._encoder.7.weight"), gdA_, 64, BT, V, 64, de_[6], 64, nullptr, 0, st));
    float* cur = gdA_;
    float* oth = gdB_;
    for (int i = 6; i >= 1; --i) {
        RC(lin_bwd_w(cur, 64, de_[i - 1], 64, G(fmt("_dvae._encoder.%d.m.weight", i)), G(fmt("_dvae._encoder.%d.m.bias", i)), BT, 64, 64, 1.f, st));
        RC(lin_bwd_x(cur, 64, P(fmt("_dvae._encoder.%d.m.weight", i)), oth, 64, BT, 64, 64, de_[i - 1], 64, nullptr, 0, st));
        float* t = cur; cur = oth; oth = t;
    }
    RC(lin_bwd_w(cur, 64, patches_, 16 * ch, G("_dvae._encoder.0.m.weight"), G("_dvae._encoder.0.m.bias"), BT, 64, 16 * ch, 1.f, st));
    return 0;
}

int SlateModel::backward(hipStream_t st) {
    OCRL_REQUIRE(have_fwd_, "backward: call forward first");
    OCRL_REQUIRE(g_, "backward: no gradient buffer bound");
    enc_only_grads_ = false;
    if (cfg.use_bcdec) {
        RC(fill_launch(g_, flat_size_, 0.f, st));       // dVAE / transformer / slotproj parameters get no gradient in this mode
        RC(bwd_bcdec(st));
        RC(bwd_encoder(st));
        have_fwd_ = false;
        return 0;
    }
    if (side_ && overlap_mode_ == 2) {
        RC(bwd_decoder(st));
        RC(bwd_encoder(st, true));           // forks the dVAE backward at the slot-attention launch, joins before the 5x5 convolutions
    } else if (side_ && overlap_mode_ >= 3) {
        // the dVAE backward (many short 64-wide products) runs beside the transformer-decoder backward and is joined before the
        // slot-attention / convolution kernels, which then have the machine to themselves (their timings stay comparable)
        RC(fork_side(st));
        std::swap(scratch_, scratch2_);
        const int rc = bwd_dvae(side_);
        std::swap(scratch_, scratch2_);
        RC(rc);
        RC(bwd_decoder(st));
        // with the decoder's weight gradients on the side stream (OCRL_DW_SIDE) that stream may still be busy: it is joined inside
        // bwd_encoder, before the convolutions, so the slot-attention backward and the input MLP overlap what is left of it
        if (!dw_mode_) RC(join_side(st));
        RC(bwd_encoder(st));
    } else if (side_) {         // the dVAE backward only needs the forward's reconstruction gradient: it overlaps decoder + encoder
        RC(fork_side(st));
        std::swap(scratch_, scratch2_);
        const int rc = bwd_dvae(side_);
        std::swap(scratch_, scratch2_);
        RC(rc);
        RC(bwd_decoder(st));
        RC(bwd_encoder(st));
        RC(join_side(st));
    } else {
        RC(bwd_decoder(st));
        RC(bwd_encoder(st));
        RC(bwd_dvae(st));
    }
    have_fwd_ = false;
    return 0;
}

int SlateModel::fork_side(hipStream_t st) {
    OCRL_HIP(hipEventRecord(ev_fork_, st));
    OCRL_HIP(hipStreamWaitEvent(side_, ev_fork_, 0));
    return 0;
}
int SlateModel::join_side(hipStream_t st) {
    OCRL_HIP(hipEventRecord(ev_join_, side_));
    OCRL_HIP(hipStreamWaitEvent(st, ev_join_, 0));
    return 0;
}

int SlateModel::grad_norm(hipStream_t st) {
    return absmax_launch(g_, flat_size_, metrics_ + 3, scratch_, scratch_floats_, st);
}

int SlateModel::clip_adam(const float lr[3], float clip, int step, float gscale, hipStream_t st) {
    OCRL_REQUIRE(m_ && v_, "clip_adam: optimiser state not bound");
    packs_valid_ = false;
    RC(grad_norm(st));
    if (enc_only_grads_) {       // after encode_backward(): the encoder tensors only (group 1 up to the slot projection / broadcast decoder)
        long long end = group_begin_[2];
        for (const ParamInfo& q : params_)
            if (q.group == 1 && (q.name.rfind("_slotproj.", 0) == 0 || q.name.rfind("_dec.", 0) == 0) && q.offset < end) end = q.offset;
        const long long b0 = group_begin_[1];
        return clip_adam_launch(p_ + b0, g_ + b0, m_ + b0, v_ + b0, end - b0, metrics_ + 3, clip, lr[1], 0.9f, 0.999f, 1e-8f, step, gscale, st);
    }
    for (int g = 0; g < 3; ++g) {
        if (cfg.use_bcdec && g != 1) continue;      // parameters without gradients are skipped, as torch's Adam does
        const long long b0 = group_begin_[g], n = group_begin_[g + 1] - b0;
        RC(clip_adam_launch(p_ + b0, g_ + b0, m_ + b0, v_ + b0, n, metrics_ + 3, clip, lr[g], 0.9f, 0.999f, 1e-8f, step, gscale, st));
    }
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Slot-Attention configuration: spatial-broadcast decoder (ocrs/common/models.py:110-141)
int SlateModel::pack_bcdec(hipStream_t st) {
    RC(bc_compose_launch(P("_dec._decoder.0.m.weight"), P("_dec._pos_emb.channels_map.weight"), P("_dec._pos_emb.channels_map.bias"), bc_Wc_,
                         bc_W1r_, D, st));
    RC(conv_pack_launch(P("_dec._decoder.1.m.weight"), bc_pk_[0], bc_pkb_[0], 5, 64, 64, 64, st));
    RC(conv_pack_launch(P("_dec._decoder.2.m.weight"), bc_pk_[1], bc_pkb_[1], 5, 64, 64, 64, st));
    if (conv_x3_ > 0) {
        const char* names[2] = {"_dec._decoder.1.m.weight", "_dec._decoder.2.m.weight"};
        for (int i = 0; i < 2; ++i) {
            auto f = x3_of_.find(bc_pk_[i]), b = x3_of_.find(bc_pkb_[i]);
            if (f != x3_of_.end()) RC(conv_pack_x3_launch(P(names[i]), const_cast<float*>(f->second), b != x3_of_.end() ? const_cast<float*>(b->second) : nullptr, st));
        }
    }
    RC(bc_c4_pack_launch(P("_dec._decoder.3.weight"), bc_Wk4_, bc_Wb4_, cfg.obs_channels + 1, st));
    return 0;
}

int SlateModel::fwd_bcdec(hipStream_t st) {
    const int B = last_.B, BK = B * K;
    RC(bc_posconv_launch(bc_Wc_, bc_P1_, S, st));
    RC(lin_fwd(slots_, D, bc_W1r_, nullptr, bc_M_, 1600, BK, 1600, D, 0, nullptr, 0, 0.f, 0, st));       // M[bk][tap][co] = W_tap s
    RC(bc_class_sum_launch(bc_M_, bc_T_, BK, 1, st));
    RC(bc_layer1_launch(bc_P1_, bc_T_, P("_dec._decoder.0.m.bias"), bc_c1_, BK, S, st));
    RC(conv_layer_fwd(bc_c1_, bc_pk_[0], P("_dec._decoder.1.m.bias"), bc_c2_, BK, S, S, 5, 64, 1, nullptr, nullptr, st));
    RC(conv_layer_fwd(bc_c2_, bc_pk_[1], P("_dec._decoder.2.m.bias"), bc_c3_, BK, S, S, 5, 64, 1, nullptr, nullptr, st));
    RC(bc_c4_fwd_launch(bc_c3_, bc_Wk4_, P("_dec._decoder.3.bias"), bc_out4_, BK, S, st));
    RC(bc_mix_launch(bc_out4_, last_.obs, recon_, bc_dout4_, metrics_ + 0, B, K, S, cfg.obs_channels, scratch_, scratch_floats_, st));
    return 0;
}

int SlateModel::bwd_bcdec(hipStream_t st) {
    const int B = last_.B, BK = B * K;
    const long long BKN = (long long)BK * N;
    // output conv 64 -> 4
    const int nb = bc_c4_wgrad_blocks(BK, S);
    RC(bc_c4_wgrad_launch(bc_c3_, bc_dout4_, scratch_, BK, S, st));
    RC(colsum_launch(scratch_, 4 * 64 * 9, G("_dec._decoder.3.weight"), nb, 4 * 64 * 9, 0, 1.f, scratch_ + (size_t)nb * 2304, scratch_floats_ - (size_t)nb * 2304, st));
    RC(colsum_launch(bc_dout4_, 4, G("_dec._decoder.3.bias"), BKN, cfg.obs_channels + 1, 0, 1.f, scratch_, scratch_floats_, st));
    RC(bc_c4_bwd_data_launch(bc_dout4_, bc_Wb4_, bc_c3_, bc_gA_, BK, S, st));                                      // gA = d c3 (pre-relu)
    RC(conv_layer_wgrad(bc_c2_, bc_gA_, G("_dec._decoder.2.m.weight"), G("_dec._decoder.2.m.bias"), BK, S, S, 5, 64, 64, st));
    RC(conv_layer_fwd(bc_gA_, bc_pkb_[1], nullptr, bc_gB_, BK, S, S, 5, 64, 0, nullptr, bc_c2_, st));               // gB = d c2 (pre-relu)
    RC(conv_layer_wgrad(bc_c1_, bc_gB_, G("_dec._decoder.1.m.weight"), G("_dec._decoder.1.m.bias"), BK, S, S, 5, 64, 64, st));
    RC(conv_layer_fwd(bc_gB_, bc_pkb_[0], nullptr, bc_gA_, BK, S, S, 5, 64, 0, nullptr, bc_c1_, st));               // gA = d c1 (pre-relu)
    // first layer through the shortcut
    RC(bc_layer1_bwd_launch(bc_gA_, bc_dT_, BK, S, scratch_, scratch_floats_, st));
    RC(colsum_launch(bc_dT_, 64, G("_dec._decoder.0.m.bias"), (long long)BK * 25, 64, 0, 1.f, scratch_, scratch_floats_, st));
    RC(bc_class_sum_launch(bc_dT_, bc_dM_, BK, 0, st));
    RC(lin_bwd_x(bc_dM_, 1600, bc_W1r_, gslots_, D, BK, 1600, D, nullptr, 0, nullptr, 0, st));                      // d slots
    RC(lin_bwd_w(bc_dM_, 1600, slots_, D, bc_dW1r_, nullptr, BK, 1600, D, 1.f, st));
    RC(colsum_launch(bc_gA_, (long long)N * 64, bc_G1_, BK, N * 64, 0, 1.f, scratch_, scratch_floats_, st));        // sum over (image, slot)
    RC(bc_posconv_bwd_launch(bc_G1_, bc_dWc_, S, st));
    RC(bc_compose_bwd_launch(P("_dec._decoder.0.m.weight"), P("_dec._pos_emb.channels_map.weight"), P("_dec._pos_emb.channels_map.bias"), bc_dWc_,
                             bc_dW1r_, G("_dec._decoder.0.m.weight"), G("_dec._pos_emb.channels_map.weight"), G("_dec._pos_emb.channels_map.bias"), D, st));
    return 0;
}
