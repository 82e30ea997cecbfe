// Slot-set pooling (poolings/common/transformer.py:9-33, poolings/transformer/transformer_module.py:27-117): the small kernels around
// the GEMMs of the CLS-token transformer encoder over S = K + 1 tokens.  Rows are batch-major here ([B][S][d]; the reference
// permutes to [S][B][d], the arithmetic is per sample and identical).
//   pool_embed      x0[b][0] = cls (+pos[0]),  x0[b][1+k] = Linear(slots)[b][k] (+pos[1+k])
//   pool_attn_fwd   multi-head self attention of one sample in one workgroup: thread = (head, query); S <= 32 keys in registers
//   pool_attn_bwd   its gradient: phase 1 per (head, query) -> dS, dq; phase 2 per (head, key) -> dk, dv
#include "common.h"
#include "kernels.h"

__global__ void pool_embed_kernel(const float* __restrict__ lin, const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ x0,
                                  long long n4, int K, int d4) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)(i % d4);
    const long long row = i / d4;
    const int S = K + 1, s = (int)(row % S);
    const long long b = row / S;
    float4 v = s == 0 ? reinterpret_cast<const float4*>(cls)[c] : reinterpret_cast<const float4*>(lin)[(b * K + s - 1) * d4 + c];
    if (pos) {
        const float4 p = reinterpret_cast<const float4*>(pos)[(long long)s * d4 + c];
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
    }
    reinterpret_cast<float4*>(x0)[i] = v;
}
// mode 0: dlin[b][k] = dx0[b][1+k];  mode 1: dx[b][0] = dout[b], other rows 0;  mode 2: out[b] = x[b][0]
__global__ void pool_rows_kernel(const float* __restrict__ in, float* __restrict__ out, long long n4, int K, int d4, int mode) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c = (int)(i % d4), S = K + 1;
    const long long row = i / d4;
    const float4* src = reinterpret_cast<const float4*>(in);
    float4* dst = reinterpret_cast<float4*>(out);
    if (mode == 0) {                      // i enumerates dlin
        const long long b = row / K;
        const int k = (int)(row % K);
        dst[i] = src[(b * S + 1 + k) * d4 + c];
    } else if (mode == 1) {               // i enumerates dx
        const long long b = row / S;
        dst[i] = (row % S) == 0 ? src[b * d4 + c] : make_float4(0.f, 0.f, 0.f, 0.f);
    } else {                              // i enumerates out
        dst[i] = src[row * S * d4 + c];
    }
}
int pool_embed_launch(const float* lin, const float* cls, const float* pos, float* x0, int B, int K, int d, hipStream_t st) {
    OCRL_REQUIRE(d % 4 == 0, "pool_embed: d %% 4");
    const long long n4 = (long long)B * (K + 1) * d / 4;
    hipLaunchKernelGGL(pool_embed_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, st, lin, cls, pos, x0, n4, K, d / 4);
    OCRL_CHECK_LAUNCH("pool_embed");
    return 0;
}
int pool_rows_launch(const float* in, float* out, int B, int K, int d, int mode, hipStream_t st) {
    OCRL_REQUIRE(d % 4 == 0, "pool_rows: d %% 4");
    const long long rows = mode == 0 ? (long long)B * K : (mode == 1 ? (long long)B * (K + 1) : B);
    const long long n4 = rows * d / 4;
    hipLaunchKernelGGL(pool_rows_kernel, dim3(cdiv(n4, 256)), dim3(256), 0, st, in, out, n4, K, d / 4, mode);
    OCRL_CHECK_LAUNCH("pool_rows");
    return 0;
}

#define PA_MAXS 32
#define PA_T 256
__device__ __forceinline__ bool pa_keep(unsigned long long seed, unsigned site, unsigned long long idx, uint32_t thr) {
    return rng_keep(rng_bits4(seed, site, idx >> 2), (int)(idx & 3), thr);
}

// qkv [B*S][3d] (q | k | v, heads of HD channels inside each), P [B][h][S][S] softmax (before dropout), O [B*S][d]
template <int HD>
__global__ __launch_bounds__(PA_T) void pool_attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ P, float* __restrict__ O, int S, int h, float p,
                                                             unsigned long long seed, unsigned site) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int d = h * HD, b = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < S * 3 * d / 4; i += PA_T) reinterpret_cast<float4*>(sm)[i] = reinterpret_cast<const float4*>(qkv + (size_t)b * S * 3 * d)[i];
    __syncthreads();
    const float scale = rsqrtf((float)HD), keep_sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const uint32_t thr = drop_thresh(p);
    for (int pr = tid; pr < h * S; pr += PA_T) {
        const int hd = pr / S, i = pr - hd * S;
        float q[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) q[c] = sm[i * 3 * d + hd * HD + c] * scale;
        float sc[PA_MAXS];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < PA_MAXS; ++j) {
            sc[j] = -INFINITY;
            if (j < S) {
                const float* kj = sm + j * 3 * d + d + hd * HD;
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < HD; ++c) a += q[c] * kj[c];
                sc[j] = a;
                mx = fmaxf(mx, a);
            }
        }
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < PA_MAXS; ++j) { sc[j] = j < S ? __expf(sc[j] - mx) : 0.f; sum += sc[j]; }
        const float inv = 1.f / sum;
        float o[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) o[c] = 0.f;
        const size_t prow = (((size_t)b * h + hd) * S + i) * S;
#pragma unroll
        for (int j = 0; j < PA_MAXS; ++j) {
            if (j < S) {
                const float a = sc[j] * inv;
                P[prow + j] = a;
                float w = a;
                if (p > 0.f) w = pa_keep(seed, site, prow + j, thr) ? a * keep_sc : 0.f;
                const float* vj = sm + j * 3 * d + 2 * d + hd * HD;
#pragma unroll
                for (int c = 0; c < HD; ++c) o[c] += w * vj[c];
            }
        }
        float* orow = O + ((size_t)b * S + i) * d + hd * HD;
#pragma unroll
        for (int c = 0; c < HD; c += 4) *reinterpret_cast<float4*>(orow + c) = make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]);
    }
}

// dqkv [B*S][3d] from dO [B*S][d]
template <int HD>
__global__ __launch_bounds__(PA_T) void pool_attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ P, const float* __restrict__ dO,
                                                             float* __restrict__ dqkv, int S, int h, float p, unsigned long long seed, unsigned site) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int d = h * HD, b = blockIdx.x, tid = threadIdx.x;
    float* s_do = sm + S * 3 * d;          // [S][d]
    float* s_ds = s_do + S * d;            // [h][S][S] dS
    float* s_pd = s_ds + h * S * S;        // [h][S][S] dropped P
    for (int i = tid; i < S * 3 * d / 4; i += PA_T) reinterpret_cast<float4*>(sm)[i] = reinterpret_cast<const float4*>(qkv + (size_t)b * S * 3 * d)[i];
    for (int i = tid; i < S * d / 4; i += PA_T) reinterpret_cast<float4*>(s_do)[i] = reinterpret_cast<const float4*>(dO + (size_t)b * S * d)[i];
    __syncthreads();
    const float scale = rsqrtf((float)HD), keep_sc = p > 0.f ? 1.f / (1.f - p) : 1.f;
    const uint32_t thr = drop_thresh(p);
    for (int pr = tid; pr < h * S; pr += PA_T) {
        const int hd = pr / S, i = pr - hd * S;
        float g[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) g[c] = s_do[i * d + hd * HD + c];
        const size_t prow = (((size_t)b * h + hd) * S + i) * S;
        float dp[PA_MAXS], pj[PA_MAXS];
        float dot = 0.f;
#pragma unroll
        for (int j = 0; j < PA_MAXS; ++j) {
            dp[j] = 0.f; pj[j] = 0.f;
            if (j < S) {
                const float a = P[prow + j];
                const float kp = p > 0.f ? (pa_keep(seed, site, prow + j, thr) ? keep_sc : 0.f) : 1.f;
                const float* vj = sm + j * 3 * d + 2 * d + hd * HD;
                float t = 0.f;
#pragma unroll
                for (int c = 0; c < HD; ++c) t += g[c] * vj[c];
                pj[j] = a;
                dp[j] = t * kp;
                dot += dp[j] * a;
                s_pd[(hd * S + i) * S + j] = a * kp;
            }
        }
        float dq[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) dq[c] = 0.f;
#pragma unroll
        for (int j = 0; j < PA_MAXS; ++j) {
            if (j < S) {
                const float ds = pj[j] * (dp[j] - dot);
                s_ds[(hd * S + i) * S + j] = ds;
                const float* kj = sm + j * 3 * d + d + hd * HD;
#pragma unroll
                for (int c = 0; c < HD; ++c) dq[c] += ds * kj[c];
            }
        }
        float* out = dqkv + ((size_t)b * S + i) * 3 * d + hd * HD;
#pragma unroll
        for (int c = 0; c < HD; c += 4) *reinterpret_cast<float4*>(out + c) = make_float4(dq[c] * scale, dq[c + 1] * scale, dq[c + 2] * scale, dq[c + 3] * scale);
    }
    __syncthreads();
    for (int pr = tid; pr < h * S; pr += PA_T) {
        const int hd = pr / S, j = pr - hd * S;
        float dk[HD], dv[HD];
#pragma unroll
        for (int c = 0; c < HD; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
        for (int i = 0; i < S; ++i) {
            const float ds = s_ds[(hd * S + i) * S + j], pd = s_pd[(hd * S + i) * S + j];
            const float* qi = sm + i * 3 * d + hd * HD;
            const float* gi = s_do + i * d + hd * HD;
#pragma unroll
            for (int c = 0; c < HD; ++c) { dk[c] += ds * qi[c]; dv[c] += pd * gi[c]; }
        }
        float* out = dqkv + ((size_t)b * S + j) * 3 * d + hd * HD;
#pragma unroll
        for (int c = 0; c < HD; c += 4) {
            *reinterpret_cast<float4*>(out + d + c) = make_float4(dk[c] * scale, dk[c + 1] * scale, dk[c + 2] * scale, dk[c + 3] * scale);
            *reinterpret_cast<float4*>(out + 2 * d + c) = make_float4(dv[c], dv[c + 1], dv[c + 2], dv[c + 3]);
        }
    }
}

template <int HD>
static int pool_attn_k(const float* qkv, float* P, float* O, const float* dO, float* dqkv, int B, int S, int h, float p, unsigned long long seed, unsigned site,
                       int backward, hipStream_t st) {
    const int d = h * HD;
    if (backward) {
        const size_t smem = ((size_t)S * 4 * d + 2 * (size_t)h * S * S) * 4;
        OCRL_REQUIRE(smem <= 160 * 1024, "pool_attn bwd: LDS request %zu too large", smem);
        OCRL_HIP(hipFuncSetAttribute((const void*)pool_attn_bwd_kernel<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL((pool_attn_bwd_kernel<HD>), dim3(B), dim3(PA_T), smem, st, qkv, P, dO, dqkv, S, h, p, seed, site);
    } else {
        const size_t smem = (size_t)S * 3 * d * 4;
        OCRL_REQUIRE(smem <= 160 * 1024, "pool_attn fwd: LDS request %zu too large", smem);
        OCRL_HIP(hipFuncSetAttribute((const void*)pool_attn_fwd_kernel<HD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL((pool_attn_fwd_kernel<HD>), dim3(B), dim3(PA_T), smem, st, qkv, P, O, S, h, p, seed, site);
    }
    OCRL_CHECK_LAUNCH("pool_attn");
    return 0;
}
int pool_attn_launch(const float* qkv, float* P, float* O, const float* dO, float* dqkv, int B, int S, int d, int h, float p, unsigned long long seed,
                     unsigned site, int backward, hipStream_t st) {
    OCRL_REQUIRE(B > 0 && S >= 1 && S <= PA_MAXS, "pool_attn: 1 <= tokens <= %d supported (got %d)", PA_MAXS, S);
    OCRL_REQUIRE(h >= 1 && d % h == 0, "pool_attn: d_model %d not divisible by nhead %d", d, h);
    switch (d / h) {
        case 8: return pool_attn_k<8>(qkv, P, O, dO, dqkv, B, S, h, p, seed, site, backward, st);
        case 16: return pool_attn_k<16>(qkv, P, O, dO, dqkv, B, S, h, p, seed, site, backward, st);
        case 32: return pool_attn_k<32>(qkv, P, O, dO, dqkv, B, S, h, p, seed, site, backward, st);
        case 64: return pool_attn_k<64>(qkv, P, O, dO, dqkv, B, S, h, p, seed, site, backward, st);
    }
    OCRL_REQUIRE(false, "pool_attn: head size %d not supported (8, 16, 32, 64)", d / h);
    return -1;
}
