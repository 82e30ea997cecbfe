// Cross attention of the transformer decoder to the K projected slots (reference: ocrs/common/transformer.py:23-50 called from
// TransformerDecoderBlock.forward :181-185 with k = v = the slot memory), folded the way slot attention is (slot_attn.hip):
// the key / value side has only K <= 16 rows per image, so the two d x d projections around the attention collapse into two small
// per-image matrices
//     scores[t, (h,k)] = LN(x)[t] . A_b[:, (h,k)],   A_b[e, (h,k)]  = dh^-1/2 * sum_j Wq[h dh + j, e] * ck_b[k, h dh + j]
//     out[t]           = sum_(h,k) Pd[t, (h,k)] Vo_b[(h,k)],   Vo_b[(h,k), o] = sum_j cv_b[k, h dh + j] * Wo[o, h dh + j]
// (ck = mem Wk^T, cv = mem Wv^T as before).  One launch per block replaces: the query projection, the attention kernel and the
// output projection (2 x 2 d^2 FLOP per token become 2 x 2 d h K) -- it reads LN(x) and the residual once and writes the new
// residual stream once.  The backward mirrors it: d Pd = d out Vo^T, soft-max backward per head, d LN(x) = d scores A^T in one
// launch; the per-image sums d Vo_b = Pd^T d out and d A_b = d scores^T LN(x) are two batched products of the GEMM family, and four
// more (batched over image and head, slate_model.cpp) take them back to Wq, Wo, ck, cv.  Only the summation order differs from the reference's three products.
//
// Column layout: col = head * KP + slot with the heads padded to KP = 8 (K <= 8) or 16 slots, so that on the matrix cores
// (scores^T = A^T x^T: accumulator lane = (token li, column group g'), register r = column 16 tt + 4 g' + r) a head is two (KP = 8)
// or four (KP = 16) consecutive 4-column groups of one 16-column tile: its soft-max is in-lane arithmetic plus one or two cross-row
// exchanges (v_permlane16/32_swap), and the probabilities are already the B operand of the second product.
#include "common.h"
#include "kernels.h"

typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define XA_T 256          // threads per workgroup (4 waves, 16 tokens each per step)

__device__ __forceinline__ float xa_xrow16(float v, bool is_max) {
    const int a = __builtin_bit_cast(int, v);
    const auto r = __builtin_amdgcn_permlane16_swap(a, a, false, false);
    const float x = __builtin_bit_cast(float, (int)r[0]), y = __builtin_bit_cast(float, (int)r[1]);
    return is_max ? __builtin_amdgcn_fmed3f(x, y, INFINITY) : x + y;
}
__device__ __forceinline__ float xa_xrow32(float v, bool is_max) {
    const int a = __builtin_bit_cast(int, v);
    const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    const float x = __builtin_bit_cast(float, (int)r[0]), y = __builtin_bit_cast(float, (int)r[1]);
    return is_max ? __builtin_amdgcn_fmed3f(x, y, INFINITY) : x + y;
}
template <int KP>
__device__ __forceinline__ float xa_head_reduce(float v, bool is_max) {      // over the KP / 4 column groups of a head
    v = xa_xrow16(v, is_max);
    if (KP == 16) v = xa_xrow32(v, is_max);
    return v;
}

// ------------------------------------------------------------------------------------------- per-image operands (all blocks, one launch)
struct XaFoldArgs {
    const float* mem;            // [B,K,d] projected slots
    const float* Wq[8]; const float* Wk[8]; const float* Wv[8]; const float* Wo[8];      // per block, [d,d] row-major (out, in)
    float* ck[8]; float* cv[8];              // [B,K,d]
    float* Ab[8]; float* AbT[8]; float* Vo[8]; float* VoT[8];      // [B,NC,d], [B,d,NC], [B,NC,d], [B,d,NC]; padding columns stay zero
    int B, K, d, h, KP, NC;
};
__global__ __launch_bounds__(256) void xattn_fold_fwd_kernel(XaFoldArgs a) {
    extern __shared__ float sm[];          // mem [K][d] | ck [K][d] | cv [K][d]
    const int b = blockIdx.x, blk = blockIdx.y, K = a.K, d = a.d, dh = d / a.h, tid = threadIdx.x;
    float* mem = sm;
    float* ck = mem + K * d;
    float* cv = ck + K * d;
    for (int i = tid; i < K * d; i += 256) mem[i] = a.mem[(size_t)b * K * d + i];
    __syncthreads();
    const float *Wq = a.Wq[blk], *Wk = a.Wk[blk], *Wv = a.Wv[blk], *Wo = a.Wo[blk];
    // ck = mem Wk^T, cv = mem Wv^T: output (which, o) per thread, all K rows at once (the weight row is read once)
    for (int i = tid; i < 2 * d; i += 256) {
        const int which = i / d, o = i - which * d;
        const float* w = (which ? Wv : Wk) + (size_t)o * d;
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        for (int e = 0; e < d; e += 4) {
            const float4 w4 = *reinterpret_cast<const float4*>(w + e);
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < K) {
                    const float4 m4 = *reinterpret_cast<const float4*>(mem + k * d + e);
                    acc[k] += (w4.x * m4.x + w4.y * m4.y) + (w4.z * m4.z + w4.w * m4.w);
                }
        }
        float* dst = which ? cv : ck;
        float* gdst = (which ? a.cv[blk] : a.ck[blk]) + (size_t)b * K * d;
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < K) { dst[k * d + o] = acc[k]; gdst[k * d + o] = acc[k]; }
    }
    __syncthreads();
    const float scale = rsqrtf((float)dh);
    float* Ab = a.Ab[blk] + (size_t)b * a.NC * d;
    float* AbT = a.AbT[blk] + (size_t)b * d * a.NC;
    float* Vo = a.Vo[blk] + (size_t)b * a.NC * d;
    float* VoT = a.VoT[blk] + (size_t)b * d * a.NC;
    // A_b[col][e] = scale sum_j ck[k][h dh + j] Wq[h dh + j][e]      (threads over e: coalesced weight reads)
    for (int i = tid; i < a.h * d; i += 256) {
        const int hh = i / d, e = i - hh * d;
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        for (int j = 0; j < dh; ++j) {
            const float w = Wq[(size_t)(hh * dh + j) * d + e];
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < K) acc[k] += ck[k * d + hh * dh + j] * w;
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < K) {
                const int col = hh * a.KP + k;
                Ab[(size_t)col * d + e] = acc[k] * scale;
                AbT[(size_t)e * a.NC + col] = acc[k] * scale;
            }
    }
    // Vo_b[col][o] = sum_j cv[k][h dh + j] Wo[o][h dh + j]
    for (int i = tid; i < a.h * d; i += 256) {
        const int hh = i / d, o = i - hh * d;
        float acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.f;
        const float* w = Wo + (size_t)o * d + hh * dh;
        for (int j = 0; j < dh; j += 4) {
            const float4 w4 = *reinterpret_cast<const float4*>(w + j);
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (k < K) {
                    const float4 c4 = *reinterpret_cast<const float4*>(cv + k * d + hh * dh + j);
                    acc[k] += (w4.x * c4.x + w4.y * c4.y) + (w4.z * c4.z + w4.w * c4.w);
                }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k)
            if (k < K) {
                const int col = hh * a.KP + k;
                Vo[(size_t)col * d + o] = acc[k];
                VoT[(size_t)o * a.NC + col] = acc[k];
            }
    }
}

// ------------------------------------------------------------------------------------------- forward
struct XaArgs {
    const float* x;          // [B,T,d] LN(x): the attention input
    const float* resid;      // [B,T,d] residual stream (forward) / unused (backward)
    float* y;                // forward: new residual stream = resid + dropout(out);   backward: d LN(x)
    float* P;                // [B,h,T,K] probabilities before dropout (written by the forward, read by the backward)
    const float* Ab; const float* AbT; const float* Vo; const float* VoT;       // per-image operands of this block
    const float* gd;         // backward: gradient wrt out (after the output dropout's backward) [B,T,d]
    float* Pd; float* dS;    // backward: [B*T, NC] dropped probabilities and score gradients (operands of the per-image sums)
    int B, T, K, d, h, NS;
    float p; unsigned long long seed; unsigned site_p, site_o;
};
// LDS row strides: operands read as ds_read_b128 at [row][16c + 4g] (row = lane & 15): stride = cols + 4 spreads the 16 rows over the banks
template <int NM, int NT, int KP>
__global__ __launch_bounds__(XA_T, 3) void xattn_fwd_kernel(XaArgs a) {
    constexpr int D = 16 * NM, NC = 16 * NT, LDA = D + 4, LDV = NC + 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* As = sm;                    // [NC][LDA]   A_b: column-major scores operand
    float* Vs = As + NC * LDA;         // [D][LDV]    Vo_b^T
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.NS, hs = blockIdx.x % a.NS;
    const int T = a.T, K = a.K;
    {
        const float* Ab = a.Ab + (size_t)b * NC * D;
        for (int i = tid; i < NC * (D / 4); i += XA_T) {
            const int r = i / (D / 4), c4 = i - r * (D / 4);
            *reinterpret_cast<float4*>(As + r * LDA + 4 * c4) = *reinterpret_cast<const float4*>(Ab + (size_t)r * D + 4 * c4);
        }
        const float* VoT = a.VoT + (size_t)b * D * NC;
        for (int i = tid; i < D * (NC / 4); i += XA_T) {
            const int r = i / (NC / 4), c4 = i - r * (NC / 4);
            *reinterpret_cast<float4*>(Vs + r * LDV + 4 * c4) = *reinterpret_cast<const float4*>(VoT + (size_t)r * NC + 4 * c4);
        }
    }
    __syncthreads();
    const int ntile = (T + 15) / 16, per = (ntile + a.NS - 1) / a.NS;
    const int t0 = hs * per, t1 = (t0 + per < ntile) ? t0 + per : ntile;
    const uint32_t thr = drop_thresh(a.p);
    const float dsc = a.p > 0.f ? 1.0f / (1.0f - a.p) : 1.f;
    // column bookkeeping of this lane: column 16 tt + 4 g + r  ->  head = col / KP, slot = col % KP
    const int slot0 = (4 * g) % KP;                   // slot of r = 0 (the same for every tile: 16 % KP == 0)
    float bias[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias[r] = slot0 + r < K ? 0.f : -1e30f;
#pragma unroll 1
    for (int t = t0 + wv; t < t1; t += XA_T / 64) {
        const int tok = t * 16 + li;
        const bool live = tok < T;
        const size_t row = ((size_t)b * T + (live ? tok : 0)) * D;
        float4 xv[NM];
#pragma unroll
        for (int c = 0; c < NM; ++c) xv[c] = live ? *reinterpret_cast<const float4*>(a.x + row + 16 * c + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        // ---- scores^T[col][tok] = A_b^T x^T
        f32x4_t s[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) s[tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NM; ++c) {
            float4 av[NT];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) av[tt] = *reinterpret_cast<const float4*>(As + (16 * tt + li) * LDA + 16 * c + 4 * g);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) s[tt] = MFMA16(av[tt].x, xv[c].x, s[tt]);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) s[tt] = MFMA16(av[tt].y, xv[c].y, s[tt]);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) s[tt] = MFMA16(av[tt].z, xv[c].z, s[tt]);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) s[tt] = MFMA16(av[tt].w, xv[c].w, s[tt]);
            if ((c & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        // ---- soft-max over the slots of each head, dropout
        f32x4_t pd[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int head = (16 * tt + 4 * g) / KP;
            float l[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) l[r] = s[tt][r] + bias[r];
            float mx = fmaxf(fmaxf(l[0], l[1]), fmaxf(l[2], l[3]));
            mx = xa_head_reduce<KP>(mx, true);
            float e[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r] = __expf(l[r] - mx);
            float sum = (e[0] + e[1]) + (e[2] + e[3]);
            sum = xa_head_reduce<KP>(sum, false);
            const float inv = __builtin_amdgcn_rcpf(sum);
            const long long prow = (((long long)b * a.h + head) * T + (live ? tok : 0)) * K;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pr = e[r] * inv;
                float v = pr;
                if (slot0 + r < K) {
                    if (live) a.P[prow + slot0 + r] = pr;
                    if (a.p > 0.f) {
                        const uint64_t idx = (uint64_t)prow + slot0 + r;
                        v = rng_keep(rng_bits4(a.seed, a.site_p, idx >> 2), (int)(idx & 3), thr) ? pr * dsc : 0.f;
                    }
                } else v = 0.f;
                pd[tt][r] = v;
            }
        }
        // ---- out^T[o][tok] = Vo_b^T Pd^T
        f32x4_t o[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) o[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
            for (int m = 0; m < NM; m += 4) {          // four operand rows in flight, then their 16 MFMAs (the barrier keeps later reads from being hoisted)
                float4 v4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v4[u] = *reinterpret_cast<const float4*>(Vs + (16 * (m + u) + li) * LDV + 16 * tt + 4 * g);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    o[m + u] = MFMA16(v4[u].x, pd[tt][0], o[m + u]);
                    o[m + u] = MFMA16(v4[u].y, pd[tt][1], o[m + u]);
                    o[m + u] = MFMA16(v4[u].z, pd[tt][2], o[m + u]);
                    o[m + u] = MFMA16(v4[u].w, pd[tt][3], o[m + u]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- y = resid + dropout(out): lane (token li, g) holds out[16 m + 4 g .. + 3]
        if (live) {
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const float4 rvm = *reinterpret_cast<const float4*>(a.resid + row + 16 * m + 4 * g);
                float4 v = make_float4(o[m][0], o[m][1], o[m][2], o[m][3]);
                if (a.p > 0.f) {
                    const uint64_t idx = (uint64_t)row + 16 * m + 4 * g;          // multiple of 4
                    const uint2 bits = rng_bits4(a.seed, a.site_o, idx >> 2);
                    v.x = rng_keep(bits, 0, thr) ? v.x * dsc : 0.f; v.y = rng_keep(bits, 1, thr) ? v.y * dsc : 0.f;
                    v.z = rng_keep(bits, 2, thr) ? v.z * dsc : 0.f; v.w = rng_keep(bits, 3, thr) ? v.w * dsc : 0.f;
                }
                v.x += rvm.x; v.y += rvm.y; v.z += rvm.z; v.w += rvm.w;
                *reinterpret_cast<float4*>(a.y + row + 16 * m + 4 * g) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------- backward
template <int NM, int NT, int KP>
__global__ __launch_bounds__(XA_T, 3) void xattn_bwd_kernel(XaArgs a) {
    constexpr int D = 16 * NM, NC = 16 * NT, LDA = D + 4, LDV = NC + 4;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* Vs = sm;                    // [NC][LDA]   Vo_b: d Pd^T = Vo gd^T
    float* As = Vs + NC * LDA;         // [D][LDV]    A_b^T ([e][col]): d x^T = A dS^T
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, li = lane & 15, g = lane >> 4;
    const int b = blockIdx.x / a.NS, hs = blockIdx.x % a.NS;
    const int T = a.T, K = a.K;
    {
        const float* Vo = a.Vo + (size_t)b * NC * D;
        for (int i = tid; i < NC * (D / 4); i += XA_T) {
            const int r = i / (D / 4), c4 = i - r * (D / 4);
            *reinterpret_cast<float4*>(Vs + r * LDA + 4 * c4) = *reinterpret_cast<const float4*>(Vo + (size_t)r * D + 4 * c4);
        }
        const float* AbT = a.AbT + (size_t)b * D * NC;
        for (int i = tid; i < D * (NC / 4); i += XA_T) {
            const int r = i / (NC / 4), c4 = i - r * (NC / 4);
            *reinterpret_cast<float4*>(As + r * LDV + 4 * c4) = *reinterpret_cast<const float4*>(AbT + (size_t)r * NC + 4 * c4);
        }
    }
    __syncthreads();
    const int ntile = (T + 15) / 16, per = (ntile + a.NS - 1) / a.NS;
    const int t0 = hs * per, t1 = (t0 + per < ntile) ? t0 + per : ntile;
    const uint32_t thr = drop_thresh(a.p);
    const float dsc = a.p > 0.f ? 1.0f / (1.0f - a.p) : 1.f;
    const int slot0 = (4 * g) % KP;
#pragma unroll 1
    for (int t = t0 + wv; t < t1; t += XA_T / 64) {
        const int tok = t * 16 + li;
        const bool live = tok < T;
        const size_t trow = (size_t)b * T + (live ? tok : 0);
        const size_t row = trow * D;
        float4 gv[NM];
#pragma unroll
        for (int c = 0; c < NM; ++c) gv[c] = live ? *reinterpret_cast<const float4*>(a.gd + row + 16 * c + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        // the saved probabilities of this lane's columns
        float pr[NT][4];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int head = (16 * tt + 4 * g) / KP;
            const long long prow = (((long long)b * a.h + head) * T + (live ? tok : 0)) * K;
#pragma unroll
            for (int r = 0; r < 4; ++r) pr[tt][r] = (live && slot0 + r < K) ? a.P[prow + slot0 + r] : 0.f;
        }
        // ---- d Pd^T[col][tok] = Vo_b gd^T
        f32x4_t dp[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) dp[tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NM; ++c) {
            float4 av[NT];
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) av[tt] = *reinterpret_cast<const float4*>(Vs + (16 * tt + li) * LDA + 16 * c + 4 * g);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) dp[tt] = MFMA16(av[tt].x, gv[c].x, dp[tt]);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) dp[tt] = MFMA16(av[tt].y, gv[c].y, dp[tt]);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) dp[tt] = MFMA16(av[tt].z, gv[c].z, dp[tt]);
#pragma unroll
            for (int tt = 0; tt < NT; ++tt) dp[tt] = MFMA16(av[tt].w, gv[c].w, dp[tt]);
            if ((c & 1) == 1) __builtin_amdgcn_sched_barrier(0);
        }
        // ---- dropout backward, soft-max backward per head
        f32x4_t ds[NT];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
            const int head = (16 * tt + 4 * g) / KP;
            const long long prow = (((long long)b * a.h + head) * T + (live ? tok : 0)) * K;
            float keep[4], dpr[4];
            float dot = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                keep[r] = 1.f;
                if (a.p > 0.f && slot0 + r < K) {
                    const uint64_t idx = (uint64_t)prow + slot0 + r;
                    keep[r] = rng_keep(rng_bits4(a.seed, a.site_p, idx >> 2), (int)(idx & 3), thr) ? dsc : 0.f;
                }
                dpr[r] = dp[tt][r] * keep[r];
                dot += pr[tt][r] * dpr[r];
            }
            dot = xa_head_reduce<KP>(dot, false);
            float4 pdv, dsv;
            float* pdp = &pdv.x;
            float* dsp = &dsv.x;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pdp[r] = pr[tt][r] * keep[r];
                dsp[r] = pr[tt][r] * (dpr[r] - dot);
                ds[tt][r] = dsp[r];
            }
            if (live) {
                *reinterpret_cast<float4*>(a.Pd + trow * NC + 16 * tt + 4 * g) = pdv;
                *reinterpret_cast<float4*>(a.dS + trow * NC + 16 * tt + 4 * g) = dsv;
            }
        }
        // ---- d x^T[e][tok] = A_b dS^T
        f32x4_t o[NM];
#pragma unroll
        for (int m = 0; m < NM; ++m) o[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) {
#pragma unroll
            for (int m = 0; m < NM; m += 4) {
                float4 v4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v4[u] = *reinterpret_cast<const float4*>(As + (16 * (m + u) + li) * LDV + 16 * tt + 4 * g);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    o[m + u] = MFMA16(v4[u].x, ds[tt][0], o[m + u]);
                    o[m + u] = MFMA16(v4[u].y, ds[tt][1], o[m + u]);
                    o[m + u] = MFMA16(v4[u].z, ds[tt][2], o[m + u]);
                    o[m + u] = MFMA16(v4[u].w, ds[tt][3], o[m + u]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (live) {
#pragma unroll
            for (int m = 0; m < NM; ++m) *reinterpret_cast<float4*>(a.y + row + 16 * m + 4 * g) = make_float4(o[m][0], o[m][1], o[m][2], o[m][3]);
        }
    }
}

// ------------------------------------------------------------------------------------------- launchers
bool xattn_supported(int K, int d, int h) {
    if (K < 1 || K > 16 || d % 16 || d > 256 || h < 1 || d % h || (d / h) % 4) return false;
    const int KP = (K <= 8 && h > 1) ? 8 : 16, NC = h * KP;
    return NC == 16 || NC == 32 || NC == 64;
}
int xattn_kp(int K, int h) { return (K <= 8 && h > 1) ? 8 : 16; }

int xattn_fold_fwd_launch(const XaFoldHost& f, hipStream_t st) {
    OCRL_REQUIRE(xattn_supported(f.K, f.d, f.h) && f.nblk >= 1 && f.nblk <= 8, "xattn_fold: unsupported shape");
    XaFoldArgs a;
    a.mem = f.mem; a.B = f.B; a.K = f.K; a.d = f.d; a.h = f.h; a.KP = xattn_kp(f.K, f.h); a.NC = f.h * a.KP;
    for (int i = 0; i < f.nblk; ++i) {
        a.Wq[i] = f.Wq[i]; a.Wk[i] = f.Wk[i]; a.Wv[i] = f.Wv[i]; a.Wo[i] = f.Wo[i];
        a.ck[i] = f.ck[i]; a.cv[i] = f.cv[i]; a.Ab[i] = f.Ab[i]; a.AbT[i] = f.AbT[i]; a.Vo[i] = f.Vo[i]; a.VoT[i] = f.VoT[i];
    }
    hipLaunchKernelGGL(xattn_fold_fwd_kernel, dim3(f.B, f.nblk), dim3(256), 3 * f.K * f.d * sizeof(float), st, a);
    OCRL_CHECK_LAUNCH("xattn_fold_fwd");
    return 0;
}

static int xa_splits(int B, int T) {
    int ncu = 256, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ncu = prop.multiProcessorCount;
    const int ntile = (T + 15) / 16;
    int ns = B <= 2 * ncu ? (2 * ncu) / B : 1;          // two resident workgroups per CU, one round
    if (ns > ntile / 4) ns = ntile / 4;                  // at least one 16-token tile per wave
    return ns < 1 ? 1 : ns;
}

template <int NM, int NT, int KP>
static int xattn_launch_t(XaArgs& a, int backward, hipStream_t st) {
    constexpr int D = 16 * NM, NC = 16 * NT;
    const size_t smem = (size_t)(NC * (D + 4) + D * (NC + 4)) * sizeof(float);
    static bool attr[2] = {false, false};
    if (!attr[backward]) {
        if (backward) OCRL_HIP(hipFuncSetAttribute((const void*)xattn_bwd_kernel<NM, NT, KP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        else OCRL_HIP(hipFuncSetAttribute((const void*)xattn_fwd_kernel<NM, NT, KP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr[backward] = true;
    }
    a.NS = xa_splits(a.B, a.T);
    if (backward) hipLaunchKernelGGL((xattn_bwd_kernel<NM, NT, KP>), dim3(a.B * a.NS), dim3(XA_T), smem, st, a);
    else hipLaunchKernelGGL((xattn_fwd_kernel<NM, NT, KP>), dim3(a.B * a.NS), dim3(XA_T), smem, st, a);
    OCRL_CHECK_LAUNCH("xattn");
    return 0;
}
template <int NM>
static int xattn_launch_m(XaArgs& a, int KP, int NC, int backward, hipStream_t st) {
    if (KP == 8 && NC == 16) return xattn_launch_t<NM, 1, 8>(a, backward, st);
    if (KP == 8 && NC == 32) return xattn_launch_t<NM, 2, 8>(a, backward, st);
    if (KP == 8 && NC == 64) return xattn_launch_t<NM, 4, 8>(a, backward, st);
    if (KP == 16 && NC == 16) return xattn_launch_t<NM, 1, 16>(a, backward, st);
    if (KP == 16 && NC == 32) return xattn_launch_t<NM, 2, 16>(a, backward, st);
    if (KP == 16 && NC == 64) return xattn_launch_t<NM, 4, 16>(a, backward, st);
    OCRL_REQUIRE(false, "xattn: unsupported column count %d", NC);
}
int xattn_launch(const XaHost& h, int backward, hipStream_t st) {
    OCRL_REQUIRE(xattn_supported(h.K, h.d, h.h), "xattn: unsupported shape (K %d, d %d, heads %d)", h.K, h.d, h.h);
    OCRL_REQUIRE(h.x && h.y && h.P && h.Ab && h.AbT && h.Vo && h.VoT, "xattn: missing buffers");
    OCRL_REQUIRE((long long)h.B * h.T * h.d < (1ll << 40), "xattn: too large");
    XaArgs a;
    a.x = h.x; a.resid = h.resid; a.y = h.y; a.P = h.P; a.Ab = h.Ab; a.AbT = h.AbT; a.Vo = h.Vo; a.VoT = h.VoT; a.gd = h.gd; a.Pd = h.Pd; a.dS = h.dS;
    a.B = h.B; a.T = h.T; a.K = h.K; a.d = h.d; a.h = h.h; a.p = h.p; a.seed = h.seed; a.site_p = h.site_p; a.site_o = h.site_o; a.NS = 1;
    if (backward) OCRL_REQUIRE(h.gd && h.Pd && h.dS, "xattn backward: missing buffers");
    else OCRL_REQUIRE(h.resid, "xattn forward: missing residual");
    const int KP = xattn_kp(h.K, h.h), NC = h.h * KP;
    switch (h.d / 16) {
        case 4: return xattn_launch_m<4>(a, KP, NC, backward, st);
        case 8: return xattn_launch_m<8>(a, KP, NC, backward, st);
        case 12: return xattn_launch_m<12>(a, KP, NC, backward, st);
        case 16: return xattn_launch_m<16>(a, KP, NC, backward, st);
        default: OCRL_REQUIRE(false, "xattn: d_model %d not built (64, 128, 192, 256)", h.d);
    }
}
