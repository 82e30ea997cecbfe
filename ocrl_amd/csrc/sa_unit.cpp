// Unit entry points of the slot-attention kernels (include/ocrl_hip.h: ocrl_slot_attention_*): the north-star kernel on its own,
// with the reference's weight tensors as they are (ocrs/common/slot_attn.py:11-45), for parity tests and for hosts that only
// want this operator.  Same code path as the model: pack -> slot_attn_launch -> weight-gradient GEMMs over (image, iteration, slot).
#include <math.h>
#include <string.h>

#include "../../include/ocrl_hip.h"
#include "kernels.h"

#define RC(x)                 \
    do {                      \
        int rc__ = (x);       \
        if (rc__) return rc__; \
    } while (0)

namespace {
struct Lay {
    size_t wts, save, grows, small, table, xchg, parts, scratch, scratch_floats, total;
};
Lay layout(int B, int K, int D, int H, int I, int NH) {
    const int C = 64;
    const SaWts wo = sa_wts_layout(C, D, H);
    const SaSave so = sa_save_layout(C, D, H, NH);
    const SaGrad go = sa_grad_layout(C, D, H, NH);
    Lay l;
    size_t a = 0;
    auto take = [&](size_t n) { size_t r = a; a += (n + 63) & ~(size_t)63; return r; };
    l.wts = take(wo.total);
    l.save = take((size_t)B * I * K * so.ld);
    l.grows = take((size_t)B * I * K * go.ld);
    l.small = take((size_t)B * (4 * D + 2 * C));
    l.table = take(64 * sizeof(PackEntry) / 4 + 64);
    l.xchg = take((size_t)B * sa_xchg_floats_host(K * NH, D));
    l.parts = take(sa_parts_floats_host(B, K * NH));
    l.scratch_floats = (size_t)1024 * 3 * D * D / 4 + (1 << 18);
    l.scratch = take(l.scratch_floats);
    l.total = a;
    return l;
}
// dW[N_out,K_in] = alpha * dy^T x over M rows (split-K through `scr`), db[N_out] = column sums of dy
int tn(const float* dy, int ld_dy, const float* x, int ldx, float* dW, float* db, long long M, int N_out, int K_in, float alpha, float* scr, size_t scr_floats,
       hipStream_t st) {
    GemmArgs a;
    a.A = dy; a.B = x; a.C = dW; a.M = N_out; a.N = K_in; a.K = (int)M; a.lda = ld_dy; a.ldb = ldx; a.ldc = K_in; a.akc = 0; a.bkc = 0; a.alpha = alpha;
    long long splits = M / 256;
    const long long slab = (long long)N_out * K_in;
    if (splits > 64) splits = 64;
    if (splits * slab > (long long)scr_floats) splits = (long long)scr_floats / slab;
    if (splits > 1) {
        a.splitk = (int)splits; a.C = scr; a.sCsplit = slab;
        RC(gemm_launch(a, st));
        RC(splitk_reduce_launch(scr, dW, slab, (int)splits, slab, 0, st));
    } else RC(gemm_launch(a, st));
    if (db) RC(colsum_launch(dy, ld_dy, db, M, N_out, 0, 1.f, scr, scr_floats, st));
    return 0;
}
}  // namespace

extern "C" {

size_t ocrl_slot_attention_ws_floats(int B, int K, int D, int H, int I) { return layout(B, K, D, H, I, 1).total; }
// heads > 1: + the per-head attention maps [B,N,heads*K] behind the single-head layout
size_t ocrl_slot_attention_mh_ws_floats(int B, int N, int K, int D, int H, int I, int heads) {
    return layout(B, K, D, H, I, heads).total + (heads > 1 ? (size_t)B * N * K * heads : 0);
}

int ocrl_slot_attention_fwd(const float* x, const float* slots0, const float* const* w, float* slots, float* attn, int B, int N, int K, int D, int H, int I,
                            float* ws, size_t ws_floats, void* stream) {
    return ocrl_slot_attention_mh_fwd(x, slots0, w, slots, attn, B, N, K, D, H, I, 1, ws, ws_floats, stream);
}
int ocrl_slot_attention_bwd(const float* x, const float* dslots, float* dx, float* dslots0, float* const* dw, int B, int N, int K, int D, int H, int I,
                            float* ws, size_t ws_floats, void* stream) {
    return ocrl_slot_attention_mh_bwd(x, dslots, dx, dslots0, dw, B, N, K, D, H, I, 1, ws, ws_floats, stream);
}

int ocrl_slot_attention_mh_fwd(const float* x, const float* slots0, const float* const* w, float* slots, float* attn, int B, int N, int K, int D, int H, int I,
                               int NH, float* ws, size_t ws_floats, void* stream) {
    OCRL_REQUIRE(x && slots0 && w && slots && ws, "ocrl_slot_attention_fwd: null argument");
    OCRL_REQUIRE(NH >= 1 && D % NH == 0, "ocrl_slot_attention_fwd: %d heads do not divide the slot size %d", NH, D);
    const int C = 64;
    const Lay l = layout(B, K, D, H, I, NH);
    OCRL_REQUIRE(ws_floats >= l.total + (NH > 1 ? (size_t)B * N * K * NH : 0), "ocrl_slot_attention_fwd: workspace too small");
    OCRL_REQUIRE(ws_floats >= l.total, "ocrl_slot_attention_fwd: workspace too small (%zu < %zu floats)", ws_floats, l.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const SaWts wo = sa_wts_layout(C, D, H);
    // weights in the reference's order: norm_inputs.{weight,bias}, norm_slots.{w,b}, norm_mlp.{w,b}, project_q, project_k, project_v,
    // gru.weight_ih, gru.weight_hh, gru.bias_ih, gru.bias_hh, mlp.0.{w,b}, mlp.2.{w,b}
    PackEntry e[32];
    int n = 0;
    auto add = [&](const float* src, int rows, int cols, int off, int tr) { e[n].src = src; e[n].rows = rows; e[n].cols = cols; e[n].dst_off = off; e[n].transpose = tr; ++n; };
    add(w[0], 1, C, wo.ln_in_g, 0); add(w[1], 1, C, wo.ln_in_b, 0);
    add(w[2], 1, D, wo.ln_s_g, 0); add(w[3], 1, D, wo.ln_s_b, 0);
    add(w[4], 1, D, wo.ln_m_g, 0); add(w[5], 1, D, wo.ln_m_b, 0);
    add(w[6], D, D, wo.Wq, 2); add(w[6], D, D, wo.WqT, 3);
    add(w[7], D, C, wo.Wk, 2); add(w[7], D, C, wo.WkT, 3);
    add(w[8], D, C, wo.Wv, 2); add(w[8], D, C, wo.WvT, 3);
    add(w[9], 3 * D, D, wo.Wih, 2); add(w[9], 3 * D, D, wo.WihT, 3);
    add(w[10], 3 * D, D, wo.Whh, 2); add(w[10], 3 * D, D, wo.WhhT, 3);
    add(w[11], 1, 3 * D, wo.bih, 0); add(w[12], 1, 3 * D, wo.bhh, 0);
    add(w[13], H, D, wo.W0, 2); add(w[13], H, D, wo.W0T, 3); add(w[14], 1, H, wo.b0, 0);
    add(w[15], D, H, wo.W2, 2); add(w[15], D, H, wo.W2T, 3); add(w[16], 1, D, wo.b2, 0);
    OCRL_HIP(hipMemcpyAsync(ws + l.table, e, n * sizeof(PackEntry), hipMemcpyHostToDevice, st));
    OCRL_HIP(hipStreamSynchronize(st));            // `e` lives on this stack frame
    int mx = 3 * D * D;
    if (H * D > mx) mx = H * D;
    RC(pack_launch(reinterpret_cast<const PackEntry*>(ws + l.table), n, mx, ws + l.wts, st));
    SlotAttnArgs a;
    a.B = B; a.N = N; a.C = C; a.K = K; a.D = D; a.H = H; a.I = I; a.NH = NH; a.eps = 1e-8f; a.scale = 1.0f / sqrtf((float)(D / NH));
    a.x = x; a.slots0 = slots0; a.wts = ws + l.wts; a.slots = slots; a.attn = attn; a.attn_heads = NH > 1 ? ws + l.total : nullptr; a.save = ws + l.save;
    a.xchg = ws + l.xchg; a.parts = ws + l.parts;
    return slot_attn_launch(a, 0, st);
}

int ocrl_slot_attention_mh_bwd(const float* x, const float* dslots, float* dx, float* dslots0, float* const* dw, int B, int N, int K, int D, int H, int I,
                               int NH, float* ws, size_t ws_floats, void* stream) {
    OCRL_REQUIRE(x && dslots && dx && dslots0 && dw && ws, "ocrl_slot_attention_bwd: null argument");
    OCRL_REQUIRE(NH >= 1 && D % NH == 0, "ocrl_slot_attention_bwd: %d heads do not divide the slot size %d", NH, D);
    const int C = 64;
    const Lay l = layout(B, K, D, H, I, NH);
    OCRL_REQUIRE(ws_floats >= l.total, "ocrl_slot_attention_bwd: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const SaSave so = sa_save_layout(C, D, H, NH);
    const SaGrad go = sa_grad_layout(C, D, H, NH);
    float *save = ws + l.save, *grows = ws + l.grows, *small = ws + l.small, *scr = ws + l.scratch;
    SlotAttnArgs a;
    a.B = B; a.N = N; a.C = C; a.K = K; a.D = D; a.H = H; a.I = I; a.NH = NH; a.eps = 1e-8f; a.scale = 1.0f / sqrtf((float)(D / NH));
    a.x = x; a.wts = ws + l.wts; a.save = save; a.dslots = dslots; a.dx = dx; a.dslots0 = dslots0; a.grows = grows; a.g_small = small;
    a.xchg = ws + l.xchg; a.parts = ws + l.parts;
    RC(slot_attn_launch(a, 1, st));
    const long long R = (long long)B * I * K;
    const size_t sf = l.scratch_floats;
    RC(tn(grows + go.out, go.ld, save + so.hid, so.ld, dw[15], dw[16], R, D, H, 1.f, scr, sf, st));        // mlp.2
    RC(tn(grows + go.hid, go.ld, save + so.m, so.ld, dw[13], dw[14], R, H, D, 1.f, scr, sf, st));          // mlp.0
    RC(tn(grows + go.gi, go.ld, save + so.u, so.ld, dw[9], dw[11], R, 3 * D, D, 1.f, scr, sf, st));        // gru ih
    RC(tn(grows + go.gh, go.ld, save + so.sprev, so.ld, dw[10], dw[12], R, 3 * D, D, 1.f, scr, sf, st));   // gru hh
    RC(tn(grows + go.q, go.ld, save + so.sn, so.ld, dw[6], nullptr, R, D, D, 1.f, scr, sf, st));           // project_q
    for (int h = 0, dh = D / NH; h < NH; ++h) {       // head h: rows h*dh .. of project_v / project_k against that head's means / folded-query gradients
        RC(tn(grows + go.u + h * dh, go.ld, save + so.up + h * C, so.ld, dw[8] + (size_t)h * dh * C, nullptr, R, dh, C, 1.f, scr, sf, st));           // project_v
        RC(tn(save + so.q + h * dh, so.ld, grows + go.qp + h * C, go.ld, dw[7] + (size_t)h * dh * C, nullptr, R, dh, C, a.scale, scr, sf, st));       // project_k
    }
    // LayerNorm gammas / betas: per-image partials [B][ln_s (2D) | ln_m (2D) | ln_in (2C)]; weight and bias are separate tensors here
    const int SM = 4 * D + 2 * C;
    RC(colsum_launch(small + 0, SM, dw[2], B, D, 0, 1.f, scr, sf, st));
    RC(colsum_launch(small + D, SM, dw[3], B, D, 0, 1.f, scr, sf, st));
    RC(colsum_launch(small + 2 * D, SM, dw[4], B, D, 0, 1.f, scr, sf, st));
    RC(colsum_launch(small + 3 * D, SM, dw[5], B, D, 0, 1.f, scr, sf, st));
    RC(colsum_launch(small + 4 * D, SM, dw[0], B, C, 0, 1.f, scr, sf, st));
    RC(colsum_launch(small + 4 * D + C, SM, dw[1], B, C, 0, 1.f, scr, sf, st));
    return 0;
}

}  // extern "C"
