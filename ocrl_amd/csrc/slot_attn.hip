// Fused slot attention (reference: ocrs/common/slot_attn.py:47-102), forward and backward.
//
// One workgroup (512 threads) per image runs ALL iterations in one launch; the slots, the query
// and every slot-side intermediate stay in LDS between iterations.  The k/v projections are
// folded algebraically so the [N,D] k and v tensors are never materialised:
//     logits[n,j] = LN(x)[n] . q'[j],   q' = scale * q Wk        (q' is [K,C], C = 64)
//     updates[j]  = (sum_n w[n,j] LN(x)[n] / sum_n w[n,j]) Wv^T  (w = softmax_j(logits) + eps)
// so each iteration streams x (N x 64 floats, 256 B per position) exactly once — 3x less HBM
// traffic than re-reading k||v (N x 384) as the reference does; only the summation order differs.
// Streaming layout: 8 lanes per position (two float4 each, 128 B contiguous per instruction),
// per-row LayerNorm / logits reduced with 3 xor-shuffles inside the 8-lane group, the K x 64
// weighted sums accumulated in registers and reduced across the workgroup once per iteration.
//
// The backward recomputes attn from x and the saved q', streams x once per iteration (reverse
// order) accumulating dq' in registers and d(LN(x)) into the dx buffer, and emits the per-slot
// gradient rows that the weight-gradient GEMMs (slate_model.cpp) contract over (image, iter, slot).
//
// All weights come from ONE packed block, all saved activations / gradient rows go to ONE
// row-matrix each (kernels.h: sa_*_layout): few base pointers keep the kernels out of scratch.
#include "common.h"
#include "kernels.h"

#define SA_THREADS 512
#define SA_WAVES 8
#define SA_C 64
#define SA_PF 2      // positions in flight per 8-lane group (x2: current + next)
#define SA_PFB 1     // same for the backward pass (register budget)

__device__ inline float grp8_sum(float v) {
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    return v;
}
__device__ inline float rows_sum(float v) {   // across the 8 row-groups of a wave (same l8)
    v += __shfl_xor(v, 8, 64);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ inline float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }

// out[j][col] = scale * sum_e in[j][e] * Wt[e*ldw + col] + bias[col],  col < NC, j < K   (in/out in LDS)
template <int K>
__device__ __noinline__ void matvec(const float* __restrict__ Wt, int ldw, int E, int NC, const float* in, int ldin,
                                    float* out, int ldout, const float* __restrict__ bias, float scale) {
    for (int col = threadIdx.x; col < NC; col += SA_THREADS) {
        float acc[K];
#pragma unroll
        for (int j = 0; j < K; ++j) acc[j] = 0.f;
#pragma unroll 4
        for (int e = 0; e < E; ++e) {
            const float w = Wt[e * ldw + col];
#pragma unroll
            for (int j = 0; j < K; ++j) acc[j] += in[j * ldin + e] * w;
        }
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int j = 0; j < K; ++j) out[j * ldout + col] = acc[j] * scale + bv;
    }
}
// same for NC == 64 outputs: the E range is split over the 8 waves and reduced through `scr` ([8][K][64])
template <int K>
__device__ __noinline__ void matvec64_split(const float* __restrict__ Wt, int ldw, int E, const float* in, int ldin,
                                            float* out, int ldout, float scale, float* scr) {
    const int col = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int per = (E + SA_WAVES - 1) / SA_WAVES;
    const int e0 = g * per, e1 = min(E, e0 + per);
    float acc[K];
#pragma unroll
    for (int j = 0; j < K; ++j) acc[j] = 0.f;
    for (int e = e0; e < e1; ++e) {
        const float w = Wt[e * ldw + col];
#pragma unroll
        for (int j = 0; j < K; ++j) acc[j] += in[j * ldin + e] * w;
    }
#pragma unroll
    for (int j = 0; j < K; ++j) scr[(g * K + j) * 64 + col] = acc[j];
    __syncthreads();
    for (int i = threadIdx.x; i < K * 64; i += SA_THREADS) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < SA_WAVES; ++w) s += scr[w * K * 64 + i];
        out[(i >> 6) * ldout + (i & 63)] = s * scale;
    }
    __syncthreads();
}

// LayerNorm of K rows of width D held in LDS (one wave per row).
__device__ __noinline__ void ln_rows(const float* in, float* out, const float* __restrict__ g, const float* __restrict__ b, int K, int D) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int j = wv; j < K; j += SA_WAVES) {
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += in[j * D + c];
        const float mu = wave_sum(s) / D;
        float q = 0.f;
        for (int c = lane; c < D; c += 64) { const float d = in[j * D + c] - mu; q += d * d; }
        const float rs = rsqrtf(wave_sum(q) / D + 1e-5f);
        for (int c = lane; c < D; c += 64) out[j * D + c] = (in[j * D + c] - mu) * rs * g[c] + b[c];
    }
}
// dx[j] (+)= LN backward of rows; accumulates dgamma/dbeta (LDS, atomics)
__device__ __noinline__ void ln_rows_bwd(const float* dy, const float* xin, float* dx, int accumulate, const float* __restrict__ g,
                                         float* dgam, float* dbet, int K, int D) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int j = wv; j < K; j += SA_WAVES) {
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += xin[j * D + c];
        const float mu = wave_sum(s) / D;
        float q = 0.f;
        for (int c = lane; c < D; c += 64) { const float d = xin[j * D + c] - mu; q += d * d; }
        const float rs = rsqrtf(wave_sum(q) / D + 1e-5f);
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < D; c += 64) {
            const float xh = (xin[j * D + c] - mu) * rs;
            const float d = dy[j * D + c];
            atomicAdd(&dgam[c], d * xh);
            atomicAdd(&dbet[c], d);
            s1 += d * g[c];
            s2 += d * g[c] * xh;
        }
        s1 = wave_sum(s1) / D;
        s2 = wave_sum(s2) / D;
        for (int c = lane; c < D; c += 64) {
            const float xh = (xin[j * D + c] - mu) * rs;
            const float v = rs * (dy[j * D + c] * g[c] - s1 - xh * s2);
            dx[j * D + c] = accumulate ? dx[j * D + c] + v : v;
        }
    }
}

// copy K rows of width W between LDS ([K][W]) and a row matrix (row stride ld)
__device__ inline void rows_to_global(const float* lds, float* g, int ld, int K, int W) {
    for (int i = threadIdx.x; i < K * W; i += SA_THREADS) { const int j = i / W, c = i - j * W; g[j * ld + c] = lds[i]; }
}
__device__ inline void rows_from_global(float* lds, const float* g, int ld, int K, int W) {
    for (int i = threadIdx.x; i < K * W; i += SA_THREADS) { const int j = i / W, c = i - j * W; lds[i] = g[j * ld + c]; }
}

// ---- per-position LayerNorm(norm_inputs) for this lane's 8 channels; returns rstd, fills xh (normalised) and v (affine)
__device__ inline float ln8(const float4& a, const float4& b, const float* gam, const float* bet, float* xh, float* v) {
    constexpr int C = SA_C;
    float t[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    float s1 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s1 += t[i];
    const float mu = grp8_sum(s1) * (1.0f / C);
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) { t[i] -= mu; s2 += t[i] * t[i]; }
    const float rs = rsqrtf(grp8_sum(s2) * (1.0f / C) + 1e-5f);
#pragma unroll
    for (int i = 0; i < 8; ++i) { xh[i] = t[i] * rs; v[i] = xh[i] * gam[i] + bet[i]; }
    return rs;
}

// ------------------------------------------------------------------------------------------- forward streaming pass
// accumulates  scr[wave][j][0..63] = sum_n w[n,j] LN(x)[n],  scr[wave][j][64] = sum_n w[n,j]
template <int K>
__device__ __forceinline__ void sa_stream_fwd(const float4* __restrict__ xb, int N, const float* qp, const float* __restrict__ ln_g,
                                           const float* __restrict__ ln_b, float eps, float* attn_out, float* scr) {
    constexpr int C = SA_C;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, grp = tid >> 3, l8 = tid & 7;
    constexpr bool QREG = (K <= 6);          // q' in registers while they last, else broadcast LDS reads
    float gam[8], bet[8], qr[QREG ? K : 1][8], acc[K][8], csum[K];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = (i >> 2) * 32 + 4 * l8 + (i & 3);
        gam[i] = ln_g[c];
        bet[i] = ln_b[c];
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        csum[j] = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (QREG) qr[j][i] = qp[j * C + (i >> 2) * 32 + 4 * l8 + (i & 3)];
            acc[j][i] = 0.f;
        }
    }
    const float4* qp4 = reinterpret_cast<const float4*>(qp);
    float4 c0[SA_PF], c1[SA_PF];
#pragma unroll
    for (int k = 0; k < SA_PF; ++k) {
        const int n = grp + 64 * k;
        if (n < N) { c0[k] = xb[n * 16 + l8]; c1[k] = xb[n * 16 + 8 + l8]; }
    }
#pragma unroll 1
    for (int base = grp; base < N; base += 64 * SA_PF) {
        float4 n0[SA_PF], n1[SA_PF];
#pragma unroll
        for (int k = 0; k < SA_PF; ++k) {
            const int n = base + 64 * (SA_PF + k);
            if (n < N) { n0[k] = xb[n * 16 + l8]; n1[k] = xb[n * 16 + 8 + l8]; }
        }
#pragma unroll
        for (int k = 0; k < SA_PF; ++k) {
            const int n = base + 64 * k;
            if (n < N) {
                float xh[8], v[8];
                ln8(c0[k], c1[k], gam, bet, xh, v);
                int lo = l8;
                asm volatile("" : "+v"(lo));
                float lg[K];
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    float d = 0.f;
                    if (QREG) {
#pragma unroll
                        for (int i = 0; i < 8; ++i) d += v[i] * qr[j][i];
                    } else {
                        const float4 qa = qp4[j * 16 + lo], qb = qp4[j * 16 + 8 + lo];
                        d = v[0] * qa.x + v[1] * qa.y + v[2] * qa.z + v[3] * qa.w + v[4] * qb.x + v[5] * qb.y + v[6] * qb.z + v[7] * qb.w;
                    }
                    lg[j] = grp8_sum(d);
                    mx = fmaxf(mx, lg[j]);
                }
                float se = 0.f;
#pragma unroll
                for (int j = 0; j < K; ++j) { lg[j] = __expf(lg[j] - mx); se += lg[j]; }
                const float inv = 1.0f / se;
                float mine = 0.f;
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const float a = lg[j] * inv;
                    if (l8 == j) mine = a;
                    const float w = a + eps;
                    csum[j] += w;
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[j][i] += w * v[i];
                }
                if (attn_out && l8 < K) attn_out[n * K + l8] = mine;
            }
        }
#pragma unroll
        for (int k = 0; k < SA_PF; ++k) { c0[k] = n0[k]; c1[k] = n1[k]; }
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        csum[j] = rows_sum(csum[j]);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = rows_sum(acc[j][i]);
    }
    if (lane < 8) {
#pragma unroll
        for (int j = 0; j < K; ++j) {
#pragma unroll
            for (int i = 0; i < 8; ++i) scr[(wv * K + j) * (C + 1) + (i >> 2) * 32 + 4 * l8 + (i & 3)] = acc[j][i];
            if (lane == 0) scr[(wv * K + j) * (C + 1) + C] = csum[j];
        }
    }
}

template <int K>
__global__ __launch_bounds__(SA_THREADS) void slot_attn_fwd_kernel(SlotAttnArgs p, SaWts wo, SaSave so) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int D = p.D, H = p.H, N = p.N;
    constexpr int C = SA_C;
    float* s = sm;                       // [K][D] slots
    float* sn = s + K * D;               // [K][D]
    float* q = sn + K * D;               // [K][D]
    float* u = q + K * D;                // [K][D]
    float* gi = u + K * D;               // [K][3D]
    float* gh = gi + K * 3 * D;          // [K][3D]
    float* hid = gh + K * 3 * D;         // [K][H]
    float* qp = hid + K * H;             // [K][C]
    float* up = qp + K * C;              // [K][C]
    float* cs = up + K * C;              // [16]
    float* scr = cs + 16;                // [8][K][C+1] (+ csum column)

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int KD = K * D;
    const float* W = p.wts;
    for (int i = tid; i < KD; i += SA_THREADS) s[i] = p.slots0[(size_t)b * KD + i];
    __syncthreads();
    const float4* xb = reinterpret_cast<const float4*>(p.x + (size_t)b * N * C);

    for (int t = 0; t < p.I; ++t) {
        float* sv = p.save ? p.save + ((size_t)b * p.I + t) * K * so.ld : nullptr;   // K rows of the save matrix
        // ---- slot side: LN, q, q'
        if (sv) rows_to_global(s, sv + so.sprev, so.ld, K, D);
        ln_rows(s, sn, W + wo.ln_s_g, W + wo.ln_s_b, K, D);
        __syncthreads();
        matvec<K>(W + wo.WqT, D, D, D, sn, D, q, D, nullptr, 1.f);
        __syncthreads();
        matvec64_split<K>(W + wo.Wk, C, D, q, D, qp, C, p.scale, scr);
        if (sv) {
            rows_to_global(sn, sv + so.sn, so.ld, K, D);
            rows_to_global(q, sv + so.q, so.ld, K, D);
            rows_to_global(qp, sv + so.qp, so.ld, K, C);
        }
        // ---- streaming pass over the N positions
        sa_stream_fwd<K>(xb, N, qp, W + wo.ln_in_g, W + wo.ln_in_b, p.eps,
                         (t == p.I - 1 && p.attn) ? p.attn + (size_t)b * N * K : nullptr, scr);
        __syncthreads();
        if (tid < K) {
            float c0 = 0.f;
            for (int w = 0; w < SA_WAVES; ++w) c0 += scr[(w * K + tid) * (C + 1) + C];
            cs[tid] = c0;
        }
        __syncthreads();
        for (int i = tid; i < K * C; i += SA_THREADS) {
            const int j = i >> 6, c = i & 63;
            float a = 0.f;
            for (int w = 0; w < SA_WAVES; ++w) a += scr[(w * K + j) * (C + 1) + c];
            up[i] = a / cs[j];
        }
        __syncthreads();
        if (sv) {
            rows_to_global(up, sv + so.up, so.ld, K, C);
            if (tid < K) sv[tid * so.ld + so.csum] = cs[tid];
        }
        // ---- updates = U' Wv^T ; GRU ; residual MLP
        matvec<K>(W + wo.WvT, D, C, D, up, C, u, D, nullptr, 1.f);
        __syncthreads();
        matvec<K>(W + wo.WihT, 3 * D, D, 3 * D, u, D, gi, 3 * D, W + wo.bih, 1.f);
        matvec<K>(W + wo.WhhT, 3 * D, D, 3 * D, s, D, gh, 3 * D, W + wo.bhh, 1.f);
        __syncthreads();
        for (int i = tid; i < KD; i += SA_THREADS) {
            const int j = i / D, c = i - j * D;
            const float r = sigmoidf_(gi[j * 3 * D + c] + gh[j * 3 * D + c]);
            const float z = sigmoidf_(gi[j * 3 * D + D + c] + gh[j * 3 * D + D + c]);
            const float hn = gh[j * 3 * D + 2 * D + c];
            const float nn = tanhf(gi[j * 3 * D + 2 * D + c] + r * hn);
            const float sg = (1.f - z) * nn + z * s[i];
            if (sv) {
                float* row = sv + j * so.ld + c;
                row[so.u] = u[i]; row[so.r] = r; row[so.z] = z; row[so.n] = nn; row[so.hn] = hn; row[so.sg] = sg;
            }
            q[i] = sg;      // q is free now: holds s_gru
        }
        __syncthreads();
        ln_rows(q, sn, W + wo.ln_m_g, W + wo.ln_m_b, K, D);      // sn = m
        __syncthreads();
        matvec<K>(W + wo.W0T, H, D, H, sn, D, hid, H, W + wo.b0, 1.f);
        __syncthreads();
        for (int i = tid; i < K * H; i += SA_THREADS) hid[i] = fmaxf(hid[i], 0.f);
        __syncthreads();
        matvec<K>(W + wo.W2T, D, H, D, hid, H, u, D, W + wo.b2, 1.f);      // u = mlp out
        __syncthreads();
        for (int i = tid; i < KD; i += SA_THREADS) s[i] = q[i] + u[i];
        if (sv) {
            rows_to_global(sn, sv + so.m, so.ld, K, D);
            rows_to_global(hid, sv + so.hid, so.ld, K, H);
        }
        __syncthreads();
    }
    for (int i = tid; i < KD; i += SA_THREADS) p.slots[(size_t)b * KD + i] = s[i];
}

// ------------------------------------------------------------------------------------------- backward streaming pass
// recomputes attn; dqp partials -> scr[wave][j][c]; d LN(x) written (FIRST), accumulated, or (FINAL) pushed through
// the LayerNorm backward into dx, with the norm_inputs gamma/beta gradients accumulated into dg_in/db_in (LDS).
template <int K, bool FIRST, bool FINAL>
__device__ __forceinline__ void sa_stream_bwd(const float4* __restrict__ xb, float4* __restrict__ dxb, int N, const float* qp, const float* dup,
                                           const float* cs, const float* ud, const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                           float eps, float* scr, float* dg_in, float* db_in) {
    constexpr int C = SA_C;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, grp = tid >> 3, l8 = tid & 7;
    float gam[8], bet[8], acc[K][8], icv[K], udv[K];
    float dgin[8], dbin[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = (i >> 2) * 32 + 4 * l8 + (i & 3);
        gam[i] = ln_g[c];
        bet[i] = ln_b[c];
        dgin[i] = 0.f;
        dbin[i] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < K; ++j) {
        icv[j] = 1.0f / cs[j];
        udv[j] = ud[j];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;
    }
    // q' and dU' stay in LDS (broadcast float4 reads) so the registers go to the dq' accumulators
    const float4* qp4 = reinterpret_cast<const float4*>(qp);
    const float4* du4 = reinterpret_cast<const float4*>(dup);
    float4 c0[SA_PFB], c1[SA_PFB];
#pragma unroll
    for (int k = 0; k < SA_PFB; ++k) {
        const int n = grp + 64 * k;
        if (n < N) { c0[k] = xb[n * 16 + l8]; c1[k] = xb[n * 16 + 8 + l8]; }
    }
#pragma unroll 1
    for (int base = grp; base < N; base += 64 * SA_PFB) {
        float4 n0[SA_PFB], n1[SA_PFB];
#pragma unroll
        for (int k = 0; k < SA_PFB; ++k) {
            const int n = base + 64 * (SA_PFB + k);
            if (n < N) { n0[k] = xb[n * 16 + l8]; n1[k] = xb[n * 16 + 8 + l8]; }
        }
#pragma unroll
        for (int k = 0; k < SA_PFB; ++k) {
            const int n = base + 64 * k;
            if (n < N) {
                float xh[8], v[8];
                const float rs = ln8(c0[k], c1[k], gam, bet, xh, v);
                int lo = l8;                       // opaque copy: keeps the q'/dU' LDS reads inside the loop
                asm volatile("" : "+v"(lo));
                float lg[K], da[K];
                float mx = -INFINITY;
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const float4 qa = qp4[j * 16 + lo], qb = qp4[j * 16 + 8 + lo];
                    const float4 ua = du4[j * 16 + lo], ub = du4[j * 16 + 8 + lo];
                    const float d0 = v[0] * qa.x + v[1] * qa.y + v[2] * qa.z + v[3] * qa.w + v[4] * qb.x + v[5] * qb.y + v[6] * qb.z + v[7] * qb.w;
                    const float d1 = v[0] * ua.x + v[1] * ua.y + v[2] * ua.z + v[3] * ua.w + v[4] * ub.x + v[5] * ub.y + v[6] * ub.z + v[7] * ub.w;
                    lg[j] = grp8_sum(d0);
                    da[j] = (grp8_sum(d1) - udv[j]) * icv[j];       // d attn[n,j]
                    mx = fmaxf(mx, lg[j]);
                    __builtin_amdgcn_sched_barrier(0);               // keep each slot's LDS reads next to their use
                }
                float se = 0.f;
#pragma unroll
                for (int j = 0; j < K; ++j) { lg[j] = __expf(lg[j] - mx); se += lg[j]; }
                const float inv = 1.0f / se;
                float dot = 0.f;
#pragma unroll
                for (int j = 0; j < K; ++j) { lg[j] *= inv; dot += lg[j] * da[j]; }
                float dv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) dv[i] = 0.f;
#pragma unroll
                for (int j = 0; j < K; ++j) {
                    const float dl = lg[j] * (da[j] - dot);          // d logits[n,j]
                    const float wn = (lg[j] + eps) * icv[j];          // normalised weight
                    const float4 qa = qp4[j * 16 + lo], qb = qp4[j * 16 + 8 + lo];
                    const float4 ua = du4[j * 16 + lo], ub = du4[j * 16 + 8 + lo];
                    const float qv[8] = {qa.x, qa.y, qa.z, qa.w, qb.x, qb.y, qb.z, qb.w};
                    const float uv[8] = {ua.x, ua.y, ua.z, ua.w, ub.x, ub.y, ub.z, ub.w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        acc[j][i] += dl * v[i];
                        dv[i] += wn * uv[i] + dl * qv[i];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (!FIRST) {
                    const float4 o0 = dxb[n * 16 + l8], o1 = dxb[n * 16 + 8 + l8];
                    dv[0] += o0.x; dv[1] += o0.y; dv[2] += o0.z; dv[3] += o0.w;
                    dv[4] += o1.x; dv[5] += o1.y; dv[6] += o1.z; dv[7] += o1.w;
                }
                if (FINAL) {   // LayerNorm(norm_inputs) backward
                    float m1 = 0.f, m2 = 0.f;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        dgin[i] += dv[i] * xh[i];
                        dbin[i] += dv[i];
                        dv[i] *= gam[i];
                        m1 += dv[i];
                        m2 += dv[i] * xh[i];
                    }
                    m1 = grp8_sum(m1) * (1.0f / C);
                    m2 = grp8_sum(m2) * (1.0f / C);
#pragma unroll
                    for (int i = 0; i < 8; ++i) dv[i] = rs * (dv[i] - m1 - xh[i] * m2);
                }
                dxb[n * 16 + l8] = make_float4(dv[0], dv[1], dv[2], dv[3]);
                dxb[n * 16 + 8 + l8] = make_float4(dv[4], dv[5], dv[6], dv[7]);
            }
        }
#pragma unroll
        for (int k = 0; k < SA_PFB; ++k) { c0[k] = n0[k]; c1[k] = n1[k]; }
    }
#pragma unroll
    for (int j = 0; j < K; ++j)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[j][i] = rows_sum(acc[j][i]);
    if (lane < 8) {
#pragma unroll
        for (int j = 0; j < K; ++j)
#pragma unroll
            for (int i = 0; i < 8; ++i) scr[(wv * K + j) * C + (i >> 2) * 32 + 4 * l8 + (i & 3)] = acc[j][i];
    }
    if (FINAL) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { dgin[i] = rows_sum(dgin[i]); dbin[i] = rows_sum(dbin[i]); }
        if (lane < 8) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = (i >> 2) * 32 + 4 * l8 + (i & 3);
                atomicAdd(&dg_in[c], dgin[i]);
                atomicAdd(&db_in[c], dbin[i]);
            }
        }
    }
}

template <int K>
__global__ __launch_bounds__(SA_THREADS) void slot_attn_bwd_kernel(SlotAttnArgs p, SaWts wo, SaSave so, SaGrad go) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int D = p.D, H = p.H, N = p.N;
    constexpr int C = SA_C;
    float* ds = sm;                      // [K][D] gradient wrt the iteration output
    float* t0 = ds + K * D;              // [K][D] scratch
    float* t1 = t0 + K * D;              // [K][D] scratch
    float* t2 = t1 + K * D;              // [K][D] scratch
    float* dgi = t2 + K * D;             // [K][3D]
    float* dgh = dgi + K * 3 * D;        // [K][3D]
    float* dhid = dgh + K * 3 * D;       // [K][H]
    float* qp = dhid + K * H;            // [K][C]
    float* up = qp + K * C;              // [K][C]
    float* dup = up + K * C;             // [K][C]
    float* dqp = dup + K * C;            // [K][C]
    float* cs = dqp + K * C;             // [16] csum
    float* ud = cs + 16;                 // [16] up . dup
    float* gacc = ud + 16;               // dgamma/dbeta accumulators: ln_s (2D), ln_m (2D), ln_in (2C)
    float* scr = gacc + 4 * D + 2 * C;   // [8][K][C]

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const int KD = K * D;
    const float* W = p.wts;
    for (int i = tid; i < KD; i += SA_THREADS) ds[i] = p.dslots[(size_t)b * KD + i];
    for (int i = tid; i < 4 * D + 2 * C; i += SA_THREADS) gacc[i] = 0.f;
    __syncthreads();
    float* dg_s = gacc;  float* db_s = gacc + D;
    float* dg_m = gacc + 2 * D;  float* db_m = gacc + 3 * D;
    float* dg_in = gacc + 4 * D;  float* db_in = gacc + 4 * D + C;
    const float4* xb = reinterpret_cast<const float4*>(p.x + (size_t)b * N * C);
    float4* dxb = reinterpret_cast<float4*>(p.dx + (size_t)b * N * C);

    for (int t = p.I - 1; t >= 0; --t) {
        const float* sv = p.save + ((size_t)b * p.I + t) * K * so.ld;
        float* gr = p.grows + ((size_t)b * p.I + t) * K * go.ld;
        // ---- residual MLP backward: s_new = sg + W2 relu(W0 m + b0) + b2,  m = LN_m(sg)
        rows_to_global(ds, gr + go.out, go.ld, K, D);
        rows_from_global(t0, sv + so.sg, so.ld, K, D);
        matvec<K>(W + wo.W2, H, D, H, ds, D, dhid, H, nullptr, 1.f);            // dhid[h] = sum_d ds[d] W2[d][h]
        __syncthreads();
        for (int i = tid; i < K * H; i += SA_THREADS) {
            const int j = i / H, c = i - j * H;
            const float v = sv[j * so.ld + so.hid + c] > 0.f ? dhid[i] : 0.f;
            dhid[i] = v;
            gr[j * go.ld + go.hid + c] = v;
        }
        __syncthreads();
        matvec<K>(W + wo.W0, D, H, D, dhid, H, t1, D, nullptr, 1.f);            // dm[e] = sum_h dhid[h] W0[h][e]
        __syncthreads();
        ln_rows_bwd(t1, t0, ds, 1, W + wo.ln_m_g, dg_m, db_m, K, D);            // ds = d s_gru
        __syncthreads();
        // ---- GRU backward
        for (int i = tid; i < KD; i += SA_THREADS) {
            const int j = i / D, c = i - j * D;
            const float* row = sv + j * so.ld + c;
            const float r = row[so.r], z = row[so.z], nn = row[so.n], hn = row[so.hn], h = row[so.sprev];
            const float g = ds[i];
            const float dn_pre = g * (1.f - z) * (1.f - nn * nn);
            const float dz_pre = g * (h - nn) * z * (1.f - z);
            const float dr_pre = dn_pre * hn * r * (1.f - r);
            dgi[j * 3 * D + c] = dr_pre; dgi[j * 3 * D + D + c] = dz_pre; dgi[j * 3 * D + 2 * D + c] = dn_pre;
            dgh[j * 3 * D + c] = dr_pre; dgh[j * 3 * D + D + c] = dz_pre; dgh[j * 3 * D + 2 * D + c] = dn_pre * r;
            t2[i] = g * z;                 // dh (direct path)
            t0[i] = h;                     // s_prev, for LN_s backward below
        }
        __syncthreads();
        rows_to_global(dgi, gr + go.gi, go.ld, K, 3 * D);
        rows_to_global(dgh, gr + go.gh, go.ld, K, 3 * D);
        matvec<K>(W + wo.Wih, D, 3 * D, D, dgi, 3 * D, t1, D, nullptr, 1.f);    // du[e] = sum_g dgi[g] Wih[g][e]
        matvec<K>(W + wo.Whh, D, 3 * D, D, dgh, 3 * D, ds, D, nullptr, 1.f);    // ds = dgh Whh (dh via recurrent weights)
        __syncthreads();
        for (int i = tid; i < KD; i += SA_THREADS) t2[i] += ds[i];
        rows_to_global(t1, gr + go.u, go.ld, K, D);
        // ---- u = up Wv^T
        rows_from_global(qp, sv + so.qp, so.ld, K, C);
        rows_from_global(up, sv + so.up, so.ld, K, C);
        if (tid < K) cs[tid] = sv[tid * so.ld + so.csum];
        __syncthreads();
        matvec64_split<K>(W + wo.Wv, C, D, t1, D, dup, C, 1.f, scr);            // dup[c] = sum_d du[d] Wv[d][c]
        if (tid < K) {
            float a = 0.f;
            for (int c = 0; c < C; ++c) a += up[tid * C + c] * dup[tid * C + c];
            ud[tid] = a;
        }
        __syncthreads();
        // ---- streaming pass: recompute attn, accumulate dq', write / accumulate d LN(x)
        const bool first = (t == p.I - 1), final_ = (t == 0);
        if (first && final_) sa_stream_bwd<K, true, true>(xb, dxb, N, qp, dup, cs, ud, W + wo.ln_in_g, W + wo.ln_in_b, p.eps, scr, dg_in, db_in);
        else if (first) sa_stream_bwd<K, true, false>(xb, dxb, N, qp, dup, cs, ud, W + wo.ln_in_g, W + wo.ln_in_b, p.eps, scr, dg_in, db_in);
        else if (final_) sa_stream_bwd<K, false, true>(xb, dxb, N, qp, dup, cs, ud, W + wo.ln_in_g, W + wo.ln_in_b, p.eps, scr, dg_in, db_in);
        else sa_stream_bwd<K, false, false>(xb, dxb, N, qp, dup, cs, ud, W + wo.ln_in_g, W + wo.ln_in_b, p.eps, scr, dg_in, db_in);
        __syncthreads();
        for (int i = tid; i < K * C; i += SA_THREADS) {
            float a = 0.f;
            for (int w = 0; w < SA_WAVES; ++w) a += scr[w * K * C + i];
            dqp[i] = a;
        }
        __syncthreads();
        rows_to_global(dqp, gr + go.qp, go.ld, K, C);
        // ---- q' = scale q Wk  ->  dq[d] = scale sum_c dqp[c] Wk[d][c] ;  q = sn Wq^T  ->  dsn[e] = sum_d dq[d] Wq[d][e]
        matvec<K>(W + wo.WkT, D, C, D, dqp, C, t1, D, nullptr, p.scale);        // t1 = dq
        __syncthreads();
        rows_to_global(t1, gr + go.q, go.ld, K, D);
        matvec<K>(W + wo.Wq, D, D, D, t1, D, ds, D, nullptr, 1.f);              // ds = dsn
        __syncthreads();
        for (int i = tid; i < KD; i += SA_THREADS) { t1[i] = ds[i]; ds[i] = t2[i]; }
        __syncthreads();
        ln_rows_bwd(t1, t0, ds, 1, W + wo.ln_s_g, dg_s, db_s, K, D);            // ds = dh + LN_s backward
        __syncthreads();
    }
    for (int i = tid; i < KD; i += SA_THREADS) p.dslots0[(size_t)b * KD + i] = ds[i];
    for (int i = tid; i < 4 * D + 2 * C; i += SA_THREADS) p.g_small[(size_t)b * (4 * D + 2 * C) + i] = gacc[i];
}

static size_t sa_fwd_smem(int K, int D, int H) { return (size_t)(K * D * 4 + K * 3 * D * 2 + K * H + K * SA_C * 2 + 16 + 8 * K * (SA_C + 1)) * 4; }
static size_t sa_bwd_smem(int K, int D, int H) { return (size_t)(K * D * 4 + K * 3 * D * 2 + K * H + K * SA_C * 4 + 32 + 4 * D + 2 * SA_C + 8 * K * SA_C) * 4; }

template <int K>
static int sa_launch_k(const SlotAttnArgs& a, int backward, hipStream_t st) {
    const size_t smem = backward ? sa_bwd_smem(K, a.D, a.H) : sa_fwd_smem(K, a.D, a.H);
    OCRL_REQUIRE(smem <= 160 * 1024, "slot_attn: LDS request %zu too large", smem);
    const SaWts wo = sa_wts_layout(a.C, a.D, a.H);
    const SaSave so = sa_save_layout(a.C, a.D, a.H);
    const SaGrad go = sa_grad_layout(a.C, a.D, a.H);
    const int pi = prof_begin(backward ? PROF_SA_BWD : PROF_SA_FWD, st);
    if (backward) {
        OCRL_HIP(hipFuncSetAttribute((const void*)slot_attn_bwd_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL((slot_attn_bwd_kernel<K>), dim3(a.B), dim3(SA_THREADS), smem, st, a, wo, so, go);
    } else {
        OCRL_HIP(hipFuncSetAttribute((const void*)slot_attn_fwd_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL((slot_attn_fwd_kernel<K>), dim3(a.B), dim3(SA_THREADS), smem, st, a, wo, so);
    }
    prof_end(pi, st);
    OCRL_CHECK_LAUNCH("slot_attn");
    return 0;
}

int slot_attn_launch(const SlotAttnArgs& a, int backward, hipStream_t st) {
    OCRL_REQUIRE(a.C == SA_C, "slot_attn: input width must be %d (got %d)", SA_C, a.C);
    OCRL_REQUIRE(a.K >= 1 && a.K <= 8, "slot_attn: 1 <= num_slots <= 8 supported (got %d)", a.K);
    OCRL_REQUIRE(a.D % 64 == 0 && a.H % 64 == 0 && a.D <= 256 && a.H <= 256, "slot_attn: slot/mlp size must be multiples of 64, <= 256");
    OCRL_REQUIRE(a.B > 0 && a.N > 0 && a.I >= 1, "slot_attn: empty problem");
    OCRL_REQUIRE((long long)a.N * 16 < (1ll << 31), "slot_attn: N too large for 32-bit row offsets");
    OCRL_REQUIRE(a.x && a.wts && ((uintptr_t)a.x & 15) == 0, "slot_attn: x/wts missing or x not 16-byte aligned");
    if (backward) OCRL_REQUIRE(a.dx && ((uintptr_t)a.dx & 15) == 0 && a.save && a.grows && a.dslots && a.dslots0 && a.g_small, "slot_attn bwd: missing buffers");
    else OCRL_REQUIRE(a.slots0 && a.slots, "slot_attn fwd: missing buffers");
    switch (a.K) {
        case 1: return sa_launch_k<1>(a, backward, st);
        case 2: return sa_launch_k<2>(a, backward, st);
        case 3: return sa_launch_k<3>(a, backward, st);
        case 4: return sa_launch_k<4>(a, backward, st);
        case 5: return sa_launch_k<5>(a, backward, st);
        case 6: return sa_launch_k<6>(a, backward, st);
        case 7: return sa_launch_k<7>(a, backward, st);
        default: return sa_launch_k<8>(a, backward, st);
    }
}

// ------------------------------------------------------------------------------------------- packing
__global__ void pack_kernel(const PackEntry* __restrict__ ent, float* __restrict__ dst) {
    const PackEntry e = ent[blockIdx.y];
    const int n = e.rows * e.cols;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (e.transpose) {      // dst[c][r] = src[r][c], i enumerates dst
            const int c = i / e.rows, r = i - c * e.rows;
            dst[e.dst_off + i] = e.src[r * e.cols + c];
        } else {
            dst[e.dst_off + i] = e.src[i];
        }
    }
}
int pack_launch(const PackEntry* entries_dev, int n_entries, int max_elems, float* dst, hipStream_t st) {
    int gx = cdiv(max_elems, 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(pack_kernel, dim3(gx, n_entries), dim3(256), 0, st, entries_dev, dst);
    OCRL_CHECK_LAUNCH("pack");
    return 0;
}
