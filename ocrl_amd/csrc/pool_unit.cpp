// C ABI of the slot-set pooling head (include/ocrl_hip.h: ocrl_pool_transformer_*): poolings/common/transformer.py:9-33
// (Linear -> [CLS; tokens] (+pos) -> nn.TransformerEncoder(num_layers x post-norm TransformerEncoderLayer, ReLU) -> CLS row), the
// consumer of the slots on the RL side (sb3s/ocr_extractor.py:45).  Stateless: the caller owns parameters, gradients and the
// workspace (the parameters belong to the RL policy's optimiser in the reference); forward leaves what backward needs in `ws`.
#include <math.h>

#include "../../include/ocrl_hip.h"
#include "kernels.h"

#define RC(x)                 \
    do {                      \
        int rc__ = (x);       \
        if (rc__) return rc__; \
    } while (0)

namespace {
constexpr unsigned SITE_POOL = 300;      // + 8 * layer + {0: attention weights, 1: dropout1, 2: FFN hidden, 3: dropout2}
constexpr size_t TMP_FLOATS = (size_t)1 << 17;
constexpr int WS_DIN = 256;               // the workspace query has no rep_dim: split-k slabs are sized for rep_dim <= 256 (wider inputs get fewer splits)

struct LayerLay {
    size_t x, qkv, P, o, y1, mr1, x1, hdn, y2, mr2;      // x = layer input; output = next layer's x
};
struct Lay {
    size_t lin, xlast, gA, gB, gC, dqkv, dhdn, dgb, dlin, tmp, sk, sk_floats, total;
    LayerLay l[OCRL_POOL_MAX_LAYERS];
};
Lay layout(int B, int K, int Din, int d, int h, int ff, int L) {
    Lay y;
    size_t a = 0;
    auto take = [&](size_t n) { size_t r = a; a += (n + 63) & ~(size_t)63; return r; };
    const size_t S = K + 1, R = (size_t)B * S;
    y.lin = take((size_t)B * K * d);
    for (int l = 0; l < L; ++l) {
        LayerLay& q = y.l[l];
        q.x = take(R * d); q.qkv = take(R * 3 * d); q.P = take((size_t)B * h * S * S); q.o = take(R * d); q.y1 = take(R * d); q.mr1 = take(2 * R);
        q.x1 = take(R * d); q.hdn = take(R * ff); q.y2 = take(R * d); q.mr2 = take(2 * R);
    }
    y.xlast = take(R * d);
    y.gA = take(R * d); y.gB = take(R * d); y.gC = take(R * d);
    y.dqkv = take(R * 3 * d); y.dhdn = take(R * ff); y.dgb = take(2 * (size_t)d); y.dlin = take((size_t)B * K * d);
    y.tmp = take(TMP_FLOATS);
    // split-k slabs of the weight-gradient GEMMs (rows of a PPO minibatch are the k dimension): up to 32 slabs of the largest weight
    size_t slab = (size_t)ff * d;
    if ((size_t)3 * d * d > slab) slab = (size_t)3 * d * d;
    if ((size_t)d * Din > slab) slab = (size_t)d * Din;
    size_t splits = R / 256;
    if (splits > 32) splits = 32;
    y.sk_floats = splits > 1 ? splits * (slab + (size_t)ff + 3 * (size_t)d + 8) : 0;
    y.sk = take(y.sk_floats);
    y.total = a;
    return y;
}

// y = epilogue(x W^T + b): relu, dropout at `site`, + resid
int lin_fwd(const float* x, const float* W, const float* b, float* y, long long M, int N, int Kk, int relu, const float* resid, float p, unsigned long long seed,
            unsigned site, hipStream_t st) {
    GemmArgs a;
    a.A = x; a.B = W; a.C = y; a.M = (int)M; a.N = N; a.K = Kk; a.lda = Kk; a.ldb = Kk; a.ldc = N; a.akc = 1; a.bkc = 1;
    a.bias = b; a.relu = relu; a.resid = resid; a.ldr = N; a.drop_p = p; a.drop_seed = seed; a.drop_site = site;
    return gemm_launch(a, st);
}
// dx = alpha * (drop(dy) W) * (mask > 0) + resid
int lin_bwd_x(const float* dy, const float* W, float* dx, long long M, int N_out, int K_in, float alpha, const float* mask, const float* resid, float p,
              unsigned long long seed, unsigned site, hipStream_t st) {
    GemmArgs a;
    a.A = dy; a.B = W; a.C = dx; a.M = (int)M; a.N = K_in; a.K = N_out; a.lda = N_out; a.ldb = K_in; a.ldc = K_in; a.akc = 1; a.bkc = 0;
    a.alpha = alpha; a.mask = mask; a.ldmask = K_in; a.resid = resid; a.ldr = K_in;
    if (p > 0.f) { a.adrop_p = p; a.adrop_site = site; a.adrop_ld = N_out; a.drop_seed = seed; }
    return gemm_launch(a, st);
}
// dW = drop(dy)^T x, db = column sums of drop(dy); split over the M rows through `sk` when there are enough of them
int lin_bwd_w(const float* dy, const float* x, float* dW, float* db, long long M, int N_out, int K_in, float p, unsigned long long seed, unsigned site,
              float* sk, size_t sk_floats, hipStream_t st) {
    GemmArgs a;
    a.A = dy; a.B = x; a.C = dW; a.M = N_out; a.N = K_in; a.K = (int)M; a.lda = N_out; a.ldb = K_in; a.ldc = K_in; a.akc = 0; a.bkc = 0;
    if (p > 0.f) { a.adrop_p = p; a.adrop_site = site; a.adrop_ld = N_out; a.drop_seed = seed; }
    const long long slab = (long long)N_out * K_in, bslab = (N_out + 3) & ~3;
    const int tiles = cdiv(N_out, 128) * cdiv(K_in, (K_in % 128 == 0) ? 128 : 64);
    long long splits = 1024 / tiles;
    if (splits > M / 256) splits = M / 256;
    if (splits * (slab + bslab) > (long long)sk_floats) splits = (long long)sk_floats / (slab + bslab);
    if (splits > 1 && N_out % 4 == 0) {
        a.splitk = (int)splits; a.C = sk; a.sCsplit = slab;
        float* bpart = sk + splits * slab;
        a.bias_out = bpart; a.sBias = bslab;
        RC(gemm_launch(a, st));
        RC(splitk_reduce_launch(sk, dW, slab, (int)splits, slab, 0, st));
        return splitk_reduce_launch(bpart, db, N_out, (int)splits, bslab, 0, st);
    }
    a.bias_out = db;
    return gemm_launch(a, st);
}
int check_dims(int B, int K, int Din, int d, int h, int ff, int L) {
    OCRL_REQUIRE(B > 0 && K >= 1 && K + 1 <= 32, "pool_transformer: 1 <= num_slots <= 31 (got %d)", K);
    OCRL_REQUIRE(L >= 1 && L <= OCRL_POOL_MAX_LAYERS, "pool_transformer: 1 <= num_layers <= %d (got %d)", OCRL_POOL_MAX_LAYERS, L);
    OCRL_REQUIRE(Din % 4 == 0 && ff % 4 == 0 && d % 64 == 0 && d <= 256, "pool_transformer: d_model must be a multiple of 64 <= 256, rep_dim/ff multiples of 4");
    OCRL_REQUIRE(h >= 1 && d % h == 0, "pool_transformer: d_model %d not divisible by nhead %d", d, h);
    return 0;
}
}  // namespace

extern "C" {

size_t ocrl_pool_transformer_ws_floats(int B, int K, int d, int nhead, int ff, int L) {
    if (L < 1 || L > OCRL_POOL_MAX_LAYERS) return 0;
    return layout(B, K, WS_DIN, d, nhead, ff, L).total;
}

int ocrl_pool_transformer_fwd(const float* slots, const float* const* w, const float* pos, float* out, int B, int K, int Din, int d, int nhead, int ff, int L,
                              float drop_p, unsigned long long seed, float* ws, size_t ws_floats, void* stream) {
    OCRL_REQUIRE(slots && w && out && ws, "ocrl_pool_transformer_fwd: null argument");
    RC(check_dims(B, K, Din, d, nhead, ff, L));
    const Lay y = layout(B, K, WS_DIN, d, nhead, ff, L);
    OCRL_REQUIRE(ws_floats >= y.total, "ocrl_pool_transformer_fwd: workspace too small (%zu < %zu floats)", ws_floats, y.total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int S = K + 1;
    const long long R = (long long)B * S;
    RC(lin_fwd(slots, w[0], w[1], ws + y.lin, (long long)B * K, d, Din, 0, nullptr, 0.f, 0, 0, st));
    RC(pool_embed_launch(ws + y.lin, w[2], pos, ws + y.l[0].x, B, K, d, st));
    for (int l = 0; l < L; ++l) {
        const float* const* q = w + 3 + 12 * l;
        const LayerLay& a = y.l[l];
        const unsigned site = SITE_POOL + 8 * l;
        float* xn = ws + (l + 1 < L ? y.l[l + 1].x : y.xlast);
        RC(lin_fwd(ws + a.x, q[0], q[1], ws + a.qkv, R, 3 * d, d, 0, nullptr, 0.f, 0, 0, st));
        RC(pool_attn_launch(ws + a.qkv, ws + a.P, ws + a.o, nullptr, nullptr, B, S, d, nhead, drop_p, seed, site + 0, 0, st));
        RC(lin_fwd(ws + a.o, q[2], q[3], ws + a.y1, R, d, d, 0, ws + a.x, drop_p, seed, site + 1, st));                 // x + dropout1(attn)
        RC(layernorm_fwd_launch(ws + a.y1, q[8], q[9], ws + a.x1, ws + a.mr1, ws + a.mr1 + R, R, d, st));
        RC(lin_fwd(ws + a.x1, q[4], q[5], ws + a.hdn, R, ff, d, 1, nullptr, drop_p, seed, site + 2, st));              // dropout(relu(linear1))
        RC(lin_fwd(ws + a.hdn, q[6], q[7], ws + a.y2, R, d, ff, 0, ws + a.x1, drop_p, seed, site + 3, st));             // x1 + dropout2(linear2)
        RC(layernorm_fwd_launch(ws + a.y2, q[10], q[11], xn, ws + a.mr2, ws + a.mr2 + R, R, d, st));
    }
    RC(pool_rows_launch(ws + y.xlast, out, B, K, d, 2, st));
    return 0;
}

int ocrl_pool_transformer_bwd(const float* slots, const float* dout, const float* const* w, float* dslots, float* const* dw, int B, int K, int Din, int d,
                              int nhead, int ff, int L, float drop_p, unsigned long long seed, float* ws, size_t ws_floats, void* stream) {
    OCRL_REQUIRE(slots && dout && w && dw && ws, "ocrl_pool_transformer_bwd: null argument");
    RC(check_dims(B, K, Din, d, nhead, ff, L));
    const Lay y = layout(B, K, WS_DIN, d, nhead, ff, L);
    OCRL_REQUIRE(ws_floats >= y.total, "ocrl_pool_transformer_bwd: workspace too small");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int S = K + 1;
    const long long R = (long long)B * S;
    const float inv_keep = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    float *g2 = ws + y.gC, *gA = ws + y.gA, *gB = ws + y.gB, *tmp = ws + y.tmp, *dgb = ws + y.dgb;
    RC(pool_rows_launch(dout, g2, B, K, d, 1, st));                                                           // only the CLS row is consumed
    for (int l = L - 1; l >= 0; --l) {
        const float* const* q = w + 3 + 12 * l;
        float* const* g = dw + 3 + 12 * l;
        const LayerLay& a = y.l[l];
        const unsigned site = SITE_POOL + 8 * l;
        // x2 = LN2(y2)
        RC(layernorm_bwd_launch(g2, ws + a.y2, ws + a.mr2, ws + a.mr2 + R, q[10], gA, dgb, R, d, 0, 0, tmp, TMP_FLOATS, st));
        RC(copy_launch(dgb, g[10], d, st)); RC(copy_launch(dgb + d, g[11], d, st));
        // y2 = x1 + dropout2(hdn W2^T + b2),  hdn = dropout(relu(x1 W1^T + b1))
        RC(lin_bwd_w(gA, ws + a.hdn, g[6], g[7], R, d, ff, drop_p, seed, site + 3, ws + y.sk, y.sk_floats, st));
        RC(lin_bwd_x(gA, q[6], ws + y.dhdn, R, d, ff, inv_keep, ws + a.hdn, nullptr, drop_p, seed, site + 3, st));
        RC(lin_bwd_w(ws + y.dhdn, ws + a.x1, g[4], g[5], R, ff, d, 0.f, 0, 0, ws + y.sk, y.sk_floats, st));
        RC(lin_bwd_x(ws + y.dhdn, q[4], gB, R, ff, d, 1.f, nullptr, gA, 0.f, 0, 0, st));                    // + residual
        // x1 = LN1(y1)
        RC(layernorm_bwd_launch(gB, ws + a.y1, ws + a.mr1, ws + a.mr1 + R, q[8], gA, dgb, R, d, 0, 0, tmp, TMP_FLOATS, st));
        RC(copy_launch(dgb, g[8], d, st)); RC(copy_launch(dgb + d, g[9], d, st));
        // y1 = x + dropout1(o Wo^T + bo)
        RC(lin_bwd_w(gA, ws + a.o, g[2], g[3], R, d, d, drop_p, seed, site + 1, ws + y.sk, y.sk_floats, st));
        RC(lin_bwd_x(gA, q[2], gB, R, d, d, 1.f, nullptr, nullptr, drop_p, seed, site + 1, st));
        RC(pool_attn_launch(ws + a.qkv, ws + a.P, nullptr, gB, ws + y.dqkv, B, S, d, nhead, drop_p, seed, site + 0, 1, st));
        RC(lin_bwd_w(ws + y.dqkv, ws + a.x, g[0], g[1], R, 3 * d, d, 0.f, 0, 0, ws + y.sk, y.sk_floats, st));
        RC(lin_bwd_x(ws + y.dqkv, q[0], g2, R, 3 * d, d, 1.f, nullptr, gA, 0.f, 0, 0, st));                  // + residual
    }
    // x0 = [cls; Linear(slots)] (+pos)
    RC(colsum_launch(g2, (long long)S * d, dw[2], B, d, 0, 1.f, tmp, TMP_FLOATS, st));
    RC(pool_rows_launch(g2, ws + y.dlin, B, K, d, 0, st));
    RC(lin_bwd_w(ws + y.dlin, slots, dw[0], dw[1], (long long)B * K, d, Din, 0.f, 0, 0, ws + y.sk, y.sk_floats, st));
    if (dslots) RC(lin_bwd_x(ws + y.dlin, w[0], dslots, (long long)B * K, d, Din, 1.f, nullptr, nullptr, 0.f, 0, 0, st));
    return 0;
}

// keep-mask of one dropout site (1 = kept), for parity tests: which = 0 attention [B,h,S,S], 1 dropout1 [B,S,d], 2 FFN hidden [B,S,ff], 3 dropout2 [B,S,d]
int ocrl_pool_transformer_dropout_mask(int layer, int which, long long n, float drop_p, unsigned long long seed, float* out, void* stream) {
    OCRL_REQUIRE(out && layer >= 0 && which >= 0 && which < 4, "ocrl_pool_transformer_dropout_mask: bad argument");
    return dropout_mask_launch(out, n, drop_p, seed, SITE_POOL + 8 * layer + which, static_cast<hipStream_t>(stream));
}

}  // extern "C"
