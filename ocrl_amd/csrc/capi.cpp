// extern "C" boundary of libocrl_hip.so (see include/ocrl_hip.h).
#include <stdarg.h>
#include <string.h>

#include <new>

#include "../../include/ocrl_hip.h"
#include "iodine_model.h"
#include "slate_model.h"

static thread_local char g_err[512] = "";

void ocrl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

struct ocrl_slate {
    SlateModel* m;
};

struct ocrl_iodine {
    IodineModel* m;
};

#define ST(s) static_cast<hipStream_t>(s)
#define GUARD(h) \
    if (!(h) || !(h)->m) { ocrl_set_error("null handle"); return 1; }

extern "C" {

const char* ocrl_last_error(void) { return g_err; }
int ocrl_obs_u8_to_f32(const unsigned char* obs_hwc, float* obs_chw, int B, int H, int W, int C, void* stream) {
    if (!obs_hwc || !obs_chw || B < 1 || H < 1 || W < 1 || C < 1) { ocrl_set_error("ocrl_obs_u8_to_f32: bad arguments"); return 1; }
    return obs_u8_to_f32_launch(obs_hwc, obs_chw, B, H, W, C, ST(stream));
}
int ocrl_abi_version(void) { return OCRL_ABI_VERSION; }

size_t ocrl_slate_config_size(void) { return sizeof(ocrl_slate_config); }

int ocrl_slate_create(const ocrl_slate_config* c, ocrl_slate** out) {
    if (!c || !out) { ocrl_set_error("ocrl_slate_create: null argument"); return 1; }
    if (c->obs_size < 8 || c->obs_size % 4 || c->vocab_size < 256 || c->num_slots < 1 || c->num_iterations < 1 || c->max_batch < 1 ||
        c->num_dec_blocks < 1 || c->num_dec_heads < 1 || c->d_model % c->num_dec_heads) {
        ocrl_set_error("ocrl_slate_create: invalid configuration");
        return 1;
    }
    SlateConfig k;
    k.obs_size = c->obs_size; k.obs_channels = c->obs_channels; k.vocab = c->vocab_size; k.d_model = c->d_model;
    k.cnn_hidden = c->cnn_hidden; k.num_slots = c->num_slots; k.num_iters = c->num_iterations; k.slot_size = c->slot_size;
    k.mlp_hidden = c->mlp_hidden; k.num_blocks = c->num_dec_blocks; k.num_heads = c->num_dec_heads; k.dropout = c->dropout;
    k.max_batch = c->max_batch;
    k.hard = c->hard ? 1 : 0; k.use_bcdec = c->use_bcdec ? 1 : 0;
    k.slot_heads = c->num_slot_heads > 0 ? c->num_slot_heads : 1;
    if (k.slot_heads > 1 && (c->num_slots > 8 || k.slot_heads * c->num_slots > 16 || c->slot_size % k.slot_heads || (c->slot_size / k.slot_heads) % 16)) {
        ocrl_set_error("ocrl_slate_create: num_slot_heads %d needs heads * num_slots <= 16, num_slots <= 8 and a head width that is a multiple of 16", k.slot_heads);
        return 1;
    }
    if (k.use_bcdec && (c->num_slots > 16 || c->obs_size < 5)) { ocrl_set_error("ocrl_slate_create: broadcast decoder needs num_slots <= 16 and obs_size >= 5"); return 1; }
    ocrl_slate* h = new (std::nothrow) ocrl_slate;
    if (!h) { ocrl_set_error("out of memory"); return 1; }
    h->m = new (std::nothrow) SlateModel(k);
    if (!h->m) { delete h; ocrl_set_error("out of memory"); return 1; }
    *out = h;
    return 0;
}
void ocrl_slate_destroy(ocrl_slate* h) {
    if (!h) return;
    delete h->m;
    delete h;
}
int ocrl_slate_param_count(const ocrl_slate* h) { return (h && h->m) ? (int)h->m->params().size() : -1; }
int ocrl_slate_param_info(const ocrl_slate* h, int i, char* name, int name_cap, int shape[4], int* ndim, long long* offset,
                          long long* numel, int* group) {
    GUARD(h);
    if (i < 0 || i >= (int)h->m->params().size()) { ocrl_set_error("param index out of range"); return 1; }
    const ParamInfo& p = h->m->params()[i];
    if (name && name_cap > 0) { strncpy(name, p.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (shape) for (int k = 0; k < 4; ++k) shape[k] = p.shape[k];
    if (ndim) *ndim = p.ndim;
    if (offset) *offset = p.offset;
    if (numel) *numel = p.numel;
    if (group) *group = p.group;
    return 0;
}
long long ocrl_slate_flat_size(const ocrl_slate* h) { return (h && h->m) ? h->m->flat_size() : -1; }
long long ocrl_slate_group_begin(const ocrl_slate* h, int g) { return (h && h->m && g >= 0 && g <= 3) ? h->m->group_begin(g) : -1; }
size_t ocrl_slate_workspace_bytes(const ocrl_slate* h) { return (h && h->m) ? h->m->workspace_bytes() : 0; }
int ocrl_slate_bind(ocrl_slate* h, float* p, float* g, float* m, float* v, void* ws, size_t n) { GUARD(h); return h->m->bind(p, g, m, v, ws, n); }

int ocrl_slate_forward(ocrl_slate* h, const float* obs, int B, float tau, int train, unsigned long long seed, const float* nz,
                       const float* nzh, const float* ns, void* stream) {
    GUARD(h);
    StepInputs in;
    in.obs = obs; in.B = B; in.tau = tau; in.train = train; in.seed = seed; in.noise_z = nz; in.noise_zh = nzh; in.noise_slots = ns;
    return h->m->forward(in, ST(stream));
}
int ocrl_slate_backward(ocrl_slate* h, void* stream) { GUARD(h); return h->m->backward(ST(stream)); }
int ocrl_slate_encode(ocrl_slate* h, const float* obs, int B, unsigned long long seed, const float* ns, void* stream) {
    GUARD(h);
    StepInputs in;
    in.obs = obs; in.B = B; in.seed = seed; in.noise_slots = ns; in.train = 0;
    return h->m->encode(in, ST(stream));
}
int ocrl_slate_encode_backward(ocrl_slate* h, const float* dslots, void* stream) { GUARD(h); return h->m->encode_backward(dslots, ST(stream)); }
int ocrl_slate_freeze_weights(ocrl_slate* h, int on) { GUARD(h); h->m->freeze_weights(on != 0); return 0; }
int ocrl_slate_generate(ocrl_slate* h, void* stream) { GUARD(h); return h->m->generate(ST(stream)); }
int ocrl_slate_clip_adam(ocrl_slate* h, const float lr[3], float clip, int step, float gscale, void* stream) {
    GUARD(h);
    return h->m->clip_adam(lr, clip, step, gscale, ST(stream));
}
int ocrl_slate_grad_norm(ocrl_slate* h, void* stream) { GUARD(h); return h->m->grad_norm(ST(stream)); }
float* ocrl_slate_metrics(const ocrl_slate* h) { return (h && h->m) ? h->m->metrics() : nullptr; }
int ocrl_slate_tensor(const ocrl_slate* h, const char* name, float** ptr, long long* count) { GUARD(h); return h->m->tensor(name, ptr, count); }
int ocrl_slate_soft_z(ocrl_slate* h, void* stream) { GUARD(h); return h->m->soft_z(ST(stream)); }
int ocrl_slate_dropout_mask(const ocrl_slate* h, unsigned site, long long n, float* out, void* stream) {
    GUARD(h);
    return h->m->dropout_mask(site, n, out, ST(stream));
}

// ---------------------------------------------------------------- unit entry points
int ocrl_gemm(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc, int akc, int bkc, float alpha,
              const float* bias, int relu, const float* mask, int ldmask, const float* resid, int ldr, int splitk, float* ws, void* stream) {
    GemmArgs a;
    a.A = A; a.B = B; a.C = C; a.M = M; a.N = N; a.K = K; a.lda = lda; a.ldb = ldb; a.ldc = ldc; a.akc = akc; a.bkc = bkc; a.alpha = alpha;
    a.bias = bias; a.relu = relu; a.mask = mask; a.ldmask = ldmask; a.resid = resid; a.ldr = ldr;
    if (splitk > 1) {
        if (!ws || ldc != N) { ocrl_set_error("ocrl_gemm: split-k needs a workspace and ldc == N"); return 1; }
        a.splitk = splitk; a.C = ws; a.sCsplit = (long long)M * N;
        if (gemm_launch(a, ST(stream))) return 1;
        return splitk_reduce_launch(ws, C, (long long)M * N, splitk, (long long)M * N, 0, ST(stream));
    }
    return gemm_launch(a, ST(stream));
}
int ocrl_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int cin, int cin_pad, int ks, int relu,
                    float* ws, void* stream) {
    if (conv_pack_launch(w, ws, nullptr, ks, cin_pad, 64, cin, ST(stream))) return 1;
    ConvArgs a;
    a.X = x; a.Wp = ws; a.Y = y; a.B = B; a.H = H; a.W = W; a.bias = bias; a.relu = relu;
    return conv_fwd_launch(a, ks, cin_pad, 64, ST(stream));
}
int ocrl_conv2d_fwd_lowlat(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int cin, int cin_pad, int ks, int relu,
                           float* ws, void* stream) {
    if (conv_pack_launch(w, ws, nullptr, ks, cin_pad, 64, cin, ST(stream))) return 1;
    ConvArgs a;
    a.X = x; a.Wp = ws; a.Y = y; a.B = B; a.H = H; a.W = W; a.bias = bias; a.relu = relu;
    return conv_fwd_launch(a, ks, cin_pad, 64, ST(stream), 1);
}
size_t ocrl_conv2d_x3_ws_floats(void) { return 2 * conv_x3_pack_floats(5); }
int ocrl_conv2d_fwd_x3(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int ks, int relu, float* ws, void* stream) {
    if (ks != 3 && ks != 5) { ocrl_set_error("ocrl_conv2d_fwd_x3: ks must be 3 or 5"); return 1; }
    if (conv_pack_x3_launch(w, ws, nullptr, ST(stream), ks)) return 1;
    ConvArgs a;
    a.X = x; a.Y = y; a.B = B; a.H = H; a.W = W; a.bias = bias; a.relu = relu;
    return conv_x3_launch(a, ws, ST(stream), ks);
}
int ocrl_conv2d_bwd_data_x3(const float* dy, const float* w, const float* mask, float* dx, int B, int H, int W, int ks, float* ws, void* stream) {
    if (ks != 3 && ks != 5) { ocrl_set_error("ocrl_conv2d_bwd_data_x3: ks must be 3 or 5"); return 1; }
    if (conv_pack_x3_launch(w, ws, ws + conv_x3_pack_floats(5), ST(stream), ks)) return 1;
    ConvArgs a;
    a.X = dy; a.Y = dx; a.B = B; a.H = H; a.W = W; a.mask = mask;
    return conv_x3_launch(a, ws + conv_x3_pack_floats(5), ST(stream), ks);
}
int ocrl_conv2d_bwd_data(const float* dy, const float* w, const float* mask, float* dx, int B, int H, int W, int ks, float* ws, void* stream) {
    float* bw = ws + (size_t)ks * ks * 64 * 64;
    if (conv_pack_launch(w, ws, bw, ks, 64, 64, 64, ST(stream))) return 1;
    ConvArgs a;
    a.X = dy; a.Wp = bw; a.Y = dx; a.B = B; a.H = H; a.W = W; a.mask = mask;
    return conv_fwd_launch(a, ks, 64, 64, ST(stream));
}
size_t ocrl_conv2d_wgrad_ws_floats(int B, int H, int W, int ks, int cin_pad) { return conv_wgrad_ws_floats(B, H, W, ks, cin_pad) + (1 << 16); }
int ocrl_conv2d_bwd_weight(const float* x, const float* dy, float* dw, float* db, int B, int H, int W, int cin, int cin_pad, int ks, float* ws,
                           size_t ws_floats, void* stream) {
    if (ws_floats < ocrl_conv2d_wgrad_ws_floats(B, H, W, ks, cin_pad)) { ocrl_set_error("ocrl_conv2d_bwd_weight: workspace too small"); return 1; }
    WgradArgs a;
    a.X = x; a.dY = dy; a.part = ws; a.B = B; a.H = H; a.W = W;
    if (conv_wgrad_launch(a, ks, cin_pad, 64, cin, dw, 0, ST(stream), 0)) return 1;
    if (db) return colsum_launch(dy, 64, db, (long long)B * H * W, 64, 0, 1.f, ws, ws_floats, ST(stream));
    return 0;
}
int ocrl_conv2d_bwd_weight_x3(const float* x, const float* dy, float* dw, int B, int H, int W, int ks, float* ws, size_t ws_floats, void* stream) {
    if (ks != 3 && ks != 5) { ocrl_set_error("ocrl_conv2d_bwd_weight_x3: ks must be 3 or 5"); return 1; }
    if (ws_floats < ocrl_conv2d_wgrad_ws_floats(B, H, W, ks, 64)) { ocrl_set_error("ocrl_conv2d_bwd_weight_x3: workspace too small"); return 1; }
    WgradArgs a;
    a.X = x; a.dY = dy; a.part = ws; a.B = B; a.H = H; a.W = W;
    return conv_wgrad_launch(a, ks, 64, 64, 64, dw, 0, ST(stream), 1);
}
int ocrl_layernorm_fwd(const float* x, const float* g, const float* b, float* y, float* mean, float* rstd, long long R, int F, void* stream) {
    return layernorm_fwd_launch(x, g, b, y, mean, rstd, R, F, ST(stream));
}
int ocrl_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* g, float* dx, float* dgb, long long R,
                       int F, float* ws, size_t ws_floats, void* stream) {
    return layernorm_bwd_launch(dy, x, mean, rstd, g, dx, dgb, R, F, 0, 0, ws, ws_floats, ST(stream));
}

int ocrl_attention_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int T, int d, int h, int ld, float p,
                       unsigned long long seed, unsigned site, void* stream) {
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.o = o; a.lse = lse; a.B = B; a.T = T; a.d = d; a.h = h; a.ld = ld; a.p = p; a.seed = seed; a.site = site;
    return attn_launch(a, 0, ST(stream));
}
int ocrl_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* lse, const float* dO, float* dq,
                       float* dk, float* dv, float* delta, int B, int T, int d, int h, int ld, float p, unsigned long long seed,
                       unsigned site, void* stream) {
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.o = const_cast<float*>(o); a.lse = const_cast<float*>(lse); a.B = B; a.T = T; a.d = d; a.h = h; a.ld = ld;
    a.p = p; a.seed = seed; a.site = site; a.dO = dO; a.dq = dq; a.dk = dk; a.dv = dv; a.delta = delta;
    return attn_launch(a, 1, ST(stream));
}


// ---------------------------------------------------------------- IODINE
int ocrl_iodine_create(const ocrl_iodine_config* c, ocrl_iodine** out) {
    if (!c || !out) { ocrl_set_error("ocrl_iodine_create: null argument"); return 1; }
    if (c->obs_size < 16 || c->obs_size % 16 || c->obs_channels != 3 || c->slot_size < 4 || c->slot_size % 4 || c->slot_size > 256 ||
        c->num_iterations < 1 || c->num_slots < 1 || c->num_slots > 16 || c->ref_mlp_hidden < 64 || c->ref_mlp_hidden % 64 || c->max_batch < 1 ||
        !(c->sigma > 0.f)) {
        ocrl_set_error("ocrl_iodine_create: invalid configuration");
        return 1;
    }
    IodineConfig k;
    k.obs_size = c->obs_size; k.obs_channels = c->obs_channels; k.slot_size = c->slot_size; k.num_iters = c->num_iterations;
    k.num_slots = c->num_slots; k.sigma = c->sigma; k.beta = c->beta; k.layer_norm = c->layer_norm; k.ref_mlp_hidden = c->ref_mlp_hidden;
    k.max_batch = c->max_batch;
    ocrl_iodine* h = new (std::nothrow) ocrl_iodine;
    if (!h) { ocrl_set_error("out of memory"); return 1; }
    h->m = new (std::nothrow) IodineModel(k);
    if (!h->m) { delete h; ocrl_set_error("out of memory"); return 1; }
    *out = h;
    return 0;
}
void ocrl_iodine_destroy(ocrl_iodine* h) {
    if (!h) return;
    delete h->m;
    delete h;
}
int ocrl_iodine_param_count(const ocrl_iodine* h) { return (h && h->m) ? (int)h->m->params().size() : -1; }
int ocrl_iodine_param_info(const ocrl_iodine* h, int i, char* name, int name_cap, int shape[4], int* ndim, long long* offset, long long* numel) {
    GUARD(h);
    if (i < 0 || i >= (int)h->m->params().size()) { ocrl_set_error("param index out of range"); return 1; }
    const ParamInfo& p = h->m->params()[i];
    if (name && name_cap > 0) { strncpy(name, p.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (shape) for (int k = 0; k < 4; ++k) shape[k] = p.shape[k];
    if (ndim) *ndim = p.ndim;
    if (offset) *offset = p.offset;
    if (numel) *numel = p.numel;
    return 0;
}
long long ocrl_iodine_flat_size(const ocrl_iodine* h) { return (h && h->m) ? h->m->flat_size() : -1; }
size_t ocrl_iodine_workspace_bytes(const ocrl_iodine* h) { return (h && h->m) ? h->m->workspace_bytes() : 0; }
int ocrl_iodine_bind(ocrl_iodine* h, float* p, float* g, float* m, float* v, void* ws, size_t n) { GUARD(h); return h->m->bind(p, g, m, v, ws, n); }
int ocrl_iodine_forward(ocrl_iodine* h, const float* obs, int B, unsigned long long seed, const float* noise, void* stream) {
    GUARD(h);
    return h->m->forward(obs, B, seed, noise, ST(stream));
}
int ocrl_iodine_backward(ocrl_iodine* h, void* stream) { GUARD(h); return h->m->backward(ST(stream)); }
int ocrl_iodine_clip_adam(ocrl_iodine* h, float lr, float clip, int step, float gscale, void* stream) {
    GUARD(h);
    return h->m->clip_adam(lr, clip, step, gscale, ST(stream));
}
int ocrl_iodine_grad_norm(ocrl_iodine* h, void* stream) { GUARD(h); return h->m->grad_norm(ST(stream)); }
float* ocrl_iodine_metrics(const ocrl_iodine* h) { return (h && h->m) ? h->m->metrics() : nullptr; }
int ocrl_iodine_tensor(const ocrl_iodine* h, const char* name, float** ptr, long long* count) { GUARD(h); return h->m->tensor(name, ptr, count); }
}  // extern "C"
