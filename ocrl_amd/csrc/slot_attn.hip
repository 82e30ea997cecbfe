// Fused slot attention (reference: ocrs/common/slot_attn.py:47-102), forward and backward.
//
// One workgroup per image runs ALL iterations in one launch; the slots, the query and every slot-side
// intermediate stay in LDS between iterations.  The k/v projections are folded algebraically so the [N,D] k and v
// tensors are never materialised:
//     logits[n,j] = LN(x)[n] . q'[j],   q' = scale * q Wk        (q' is [K,C], C = 64)
//     updates[j]  = (sum_n w[n,j] LN(x)[n] / sum_n w[n,j]) Wv^T  (w = softmax_j(logits) + eps)
// so each iteration streams x (N x 64 floats, 256 B per position) exactly once — 3x less HBM traffic than
// re-reading k||v (N x 384) as the reference does; only the summation order differs.
//
// Streaming pass (the HBM-bound part) on the matrix cores: a wave owns tiles of 16 positions; lane (i, g) loads
// x[pos i][16c+4g .. +3] (c = 0..3), LayerNorm is reduced over the 4 lanes of a position with two xor-shuffles, and
//     logits[16 pos x 16 slots]   = 16 x v_mfma_f32_16x16x4_f32 (slots padded to 16, q' in registers)
//     P^T[64 ch x 16 slots]      += 16 x MFMA with the softmax weights straight from the accumulator registers
// (LN(x)^T comes back through a per-wave LDS tile).  The backward adds d attn = LN(x).dU', dq'^T and
// d LN(x) = w dU' + dlogits q' (64 MFMAs per tile), pushes d LN(x) through the LayerNorm backward in the load
// layout and accumulates it into the dx buffer.  Per-slot gradient rows are emitted for the weight-gradient GEMMs
// (slate_model.cpp) which contract over (image, iteration, slot).
//
// All weights come from ONE packed block, all saved activations / gradient rows go to ONE row-matrix each
// (kernels.h: sa_*_layout): few base pointers keep the kernels out of scratch.
#include "common.h"
#include "kernels.h"

#define SA_C 64
#include <stdlib.h>
#define SA_TF 1024        // forward threads per workgroup (16 waves)
#define SA_TB 512         // backward threads per workgroup (8 waves)
#define SA_TLD 68         // row stride of the per-wave [16 positions][64 channels] LDS tile
#define SA_WLD 17         // row stride of the per-wave [16 positions][16 slots] LDS tiles

typedef float f32x4_t __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__device__ inline float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
// All-reduce over the 16 lanes sharing lane>>4 (one DPP row) by row rotations 8, 4, 2, 1: VALU-speed DPP moves, no trip
// through the LDS crossbar (ds_bpermute), which is what __shfl_xor costs.
#define SA_ROR(v, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (v)), 0x120 + (n), 0xF, 0xF, false))
__device__ inline float red16_sum(float v) {
    v += SA_ROR(v, 8); v += SA_ROR(v, 4); v += SA_ROR(v, 2); v += SA_ROR(v, 1);
    return v;
}
__device__ inline float red16_max(float v) {
    v = fmaxf(v, SA_ROR(v, 8)); v = fmaxf(v, SA_ROR(v, 4)); v = fmaxf(v, SA_ROR(v, 2)); v = fmaxf(v, SA_ROR(v, 1));
    return v;
}
__device__ inline float redg_sum(float v) {           // over the 4 lanes of one position (same lane & 15)
    v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
    return v;
}

// out[j][col] = scale * sum_e in[j][e] * Wn[col*ldw + e] + bias[col],  col < NC, j < K   (in/out in LDS; E, NC multiples of 16).
// The K <= 16 slot rows are the M side of v_mfma_f32_16x16x4_f32 (rows >= K fed as zeros), a wave owns 16-column tiles of the
// output; per 16 values of e a lane issues one float4 weight load (the k-contiguous orientation of the weight, L2 resident) and
// one ds_read_b128 of its slot row and feeds 4 MFMAs — the slot-side products take microseconds instead of walking e serially.
// Ends with a workgroup barrier.
template <int K>
__device__ __forceinline__ void matvec(const float* __restrict__ Wn, int ldw, int E, int NC, const float* in, int ldin, float* out, int ldout,
                                       const float* __restrict__ bias, float scale) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6, li = lane & 15, g = lane >> 4;
    const int ntile = NC >> 4, nks = E >> 4;
    const float* arow = in + (li < K ? li : 0) * ldin + 4 * g;
    for (int tile = wv; tile < ntile; tile += nw) {
        const int col = tile * 16 + li;
        const float* wrow = Wn + (size_t)col * ldw + 4 * g;
        f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
        for (int s = 0; s < nks; ++s) {
            const float4 b = *reinterpret_cast<const float4*>(wrow + 16 * s);
            float4 a = *reinterpret_cast<const float4*>(arow + 16 * s);
            if (li >= K) a = make_float4(0.f, 0.f, 0.f, 0.f);
            acc = MFMA16(a.x, b.x, acc);
            acc = MFMA16(a.y, b.y, acc);
            acc = MFMA16(a.z, b.z, acc);
            acc = MFMA16(a.w, b.w, acc);
        }
        const float bv = bias ? bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * g + r < K) out[(4 * g + r) * ldout + col] = acc[r] * scale + bv;      // acc[r] = row (slot) 4g + r, column li
    }
    __syncthreads();
}

// LayerNorm of K rows of width D held in LDS (one wave per row).
__device__ __noinline__ void ln_rows(const float* in, float* out, const float* __restrict__ g, const float* __restrict__ b, int K, int D) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int j = wv; j < K; j += nw) {
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += in[j * D + c];
        const float mu = wave_sum(s) / D;
        float q = 0.f;
        for (int c = lane; c < D; c += 64) { const float d = in[j * D + c] - mu; q += d * d; }
        const float rs = rsqrtf(wave_sum(q) / D + 1e-5f);
        for (int c = lane; c < D; c += 64) out[j * D + c] = (in[j * D + c] - mu) * rs * g[c] + b[c];
    }
}
// dx[j] (+)= LN backward of rows; accumulates dgamma/dbeta into LDS accumulators.  The column sums over the rows run in a fixed
// order (thread c adds rows 0..K-1), not as atomics: run-to-run bitwise reproducibility is the repo's race detector.  Block-uniform call.
__device__ __noinline__ void ln_rows_bwd(const float* dy, const float* xin, float* dx, int accumulate, const float* __restrict__ g,
                                         float* dgam, float* dbet, int K, int D) {
    __shared__ float s_mu[16], s_rs[16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int j = wv; j < K; j += nw) {
        float s = 0.f;
        for (int c = lane; c < D; c += 64) s += xin[j * D + c];
        const float mu = wave_sum(s) / D;
        float q = 0.f;
        for (int c = lane; c < D; c += 64) { const float d = xin[j * D + c] - mu; q += d * d; }
        const float rs = rsqrtf(wave_sum(q) / D + 1e-5f);
        if (lane == 0) { s_mu[j] = mu; s_rs[j] = rs; }
        float s1 = 0.f, s2 = 0.f;
        for (int c = lane; c < D; c += 64) {
            const float xh = (xin[j * D + c] - mu) * rs;
            const float d = dy[j * D + c];
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
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float ag = 0.f, ab = 0.f;
        for (int j = 0; j < K; ++j) {
            const float d = dy[j * D + c];
            ag += d * (xin[j * D + c] - s_mu[j]) * s_rs[j];
            ab += d;
        }
        dgam[c] += ag;
        dbet[c] += ab;
    }
}

// copy K rows of width W between LDS ([K][W]) and a row matrix (row stride ld)
__device__ inline void rows_to_global(const float* lds, float* g, int ld, int K, int W) {
    for (int i = threadIdx.x; i < K * W; i += blockDim.x) { const int j = i / W, c = i - j * W; g[j * ld + c] = lds[i]; }
}
__device__ inline void rows_from_global(float* lds, const float* g, int ld, int K, int W) {
    for (int i = threadIdx.x; i < K * W; i += blockDim.x) { const int j = i / W, c = i - j * W; lds[i] = g[j * ld + c]; }
}

// one 16-position tile for lane (i = lane & 15, g = lane >> 4): x[pos i][16c + 4g .. +3], c = 0..3 (zeros beyond N)
__device__ inline void sa_load_tile(const float4* __restrict__ xb, int N, int t, int li, int g, float4 (&v)[4]) {
    const int pos = t * 16 + li;
#pragma unroll
    for (int c = 0; c < 4; ++c) v[c] = pos < N ? xb[pos * 16 + 4 * c + g] : make_float4(0.f, 0.f, 0.f, 0.f);
}
// LayerNorm(norm_inputs) statistics of a position spread over its 4 lanes: xn = (x - mean) * rstd; returns rstd.
// The affine part is folded into the slot-side operands: LN(x).q' = xn.(gamma q') + beta.q'.
__device__ inline float sa_ln16(const float4 (&v)[4], float* xn) {
    float t[16] = {v[0].x, v[0].y, v[0].z, v[0].w, v[1].x, v[1].y, v[1].z, v[1].w, v[2].x, v[2].y, v[2].z, v[2].w, v[3].x, v[3].y, v[3].z, v[3].w};
    float s1 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s1 += t[k];
    const float mu = redg_sum(s1) * (1.0f / SA_C);
    float s2 = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { t[k] -= mu; s2 += t[k] * t[k]; }
    const float rs = rsqrtf(redg_sum(s2) * (1.0f / SA_C) + 1e-5f);
#pragma unroll
    for (int k = 0; k < 16; ++k) xn[k] = t[k] * rs;
    return rs;
}
// qg[j][c] = gamma[c] v[j][c],  qb[j] = sum_c beta[c] v[j][c]   (v, qg: [K][64] in LDS; one wave per row)
__device__ inline void sa_fold_affine(const float* v, const float* __restrict__ gam, const float* __restrict__ bet, float* qg, float* qb, int K) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int j = wv; j < K; j += nw) {
        const float x = v[j * SA_C + lane];
        qg[j * SA_C + lane] = x * gam[lane];
        const float sb = wave_sum(x * bet[lane]);
        if (lane == 0) qb[j] = sb;
    }
}

// ------------------------------------------------------------------------------------------- forward streaming pass
// scr[wave][j][0..63] = sum_n w[n,j] xn[n],  scr[wave][j][64] = sum_n w[n,j]   (scr aliases the tiles: barrier inside)
template <int K>
__device__ __forceinline__ void sa_stream_fwd(const float4* __restrict__ xb, int N, const float* qg, const float* qb, float eps, float* attn_out,
                                              float* tiles, float* scr, int tile0 = 0, int tile1 = -1) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6, li = lane & 15, g = lane >> 4;
    float* tile = tiles + wv * 16 * SA_TLD;
    float qpr[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) qpr[k] = li < K ? qg[li * SA_C + 16 * (k >> 2) + 4 * g + (k & 3)] : 0.f;
    const float l0 = li < K ? qb[li] : 0.f;
    f32x4_t acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float csum = 0.f;
    const int ntile = tile1 < 0 ? (N + 15) / 16 : tile1;      // this workgroup's tiles: [tile0, ntile)
    // two register stages per wave: while tile t is processed, tiles t + nw and t + 2 nw are in flight
    float4 bufA[4], bufB[4];
    if (tile0 + wv < ntile) sa_load_tile(xb, N, tile0 + wv, li, g, bufA);
    if (tile0 + wv + nw < ntile) sa_load_tile(xb, N, tile0 + wv + nw, li, g, bufB);
    auto tile_step = [&](float4 (&cur)[4], int t) {
        float xn[16];
        sa_ln16(cur, xn);
        if (t + 2 * nw < ntile) sa_load_tile(xb, N, t + 2 * nw, li, g, cur);
        f32x4_t L = (f32x4_t){l0, l0, l0, l0};
#pragma unroll
        for (int k = 0; k < 16; ++k) L = MFMA16(xn[k], qpr[k], L);           // L[r] = logits[pos 4g+r][slot li]
#pragma unroll
        for (int c = 0; c < 4; ++c)
            *reinterpret_cast<float4*>(tile + li * SA_TLD + 16 * c + 4 * g) = make_float4(xn[4 * c], xn[4 * c + 1], xn[4 * c + 2], xn[4 * c + 3]);
        f32x4_t w;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = li < K ? L[r] : -INFINITY;
            const float mx = red16_max(x);
            const float e = li < K ? __expf(x - mx) : 0.f;
            const float a = e / red16_sum(e);
            const int pos = t * 16 + 4 * g + r;
            const bool ok = li < K && pos < N;
            w[r] = ok ? a + eps : 0.f;
            csum += w[r];
            if (attn_out && ok) attn_out[pos * K + li] = a;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = MFMA16(tile[(4 * g + r) * SA_TLD + 16 * m + li], w[r], acc[m]);   // sum_n w xn^T: [ch][slot]
        __builtin_amdgcn_wave_barrier();
    };
#pragma unroll 1
    for (int t = tile0 + wv; t < ntile; t += 2 * nw) {
        tile_step(bufA, t);
        if (t + nw < ntile) tile_step(bufB, t + nw);
    }
    csum = redg_sum(csum);
    __syncthreads();          // every wave is done with its tile: the region is reused for the partials
    if (li < K) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) scr[(wv * K + li) * (SA_C + 1) + 16 * m + 4 * g + r] = acc[m][r];
        if (g == 0) scr[(wv * K + li) * (SA_C + 1) + SA_C] = csum;
    }
}

// Slot-side work is independent per slot, so for K > 8 it runs in two blocks of KB = ceil(K/2) slots: the per-block
// temporaries (LN output, q, GRU gates, MLP hidden) are sized for KB rows and the kernel fits the 160 KB LDS up to K = 16.
template <int K> struct SaBlk { static constexpr int NB = K > 8 ? 2 : 1, KB = (K + NB - 1) / NB, KP = NB * KB; };

// LDS map of the forward kernels
template <int K>
struct SaFwdLds {
    float *s, *sn, *q, *u, *gi, *gh, *hid, *qp, *up, *qg, *cs, *qb, *tiles;
    __device__ SaFwdLds(float* sm, int D, int H) {
        constexpr int C = SA_C, KB = SaBlk<K>::KB, KP = SaBlk<K>::KP;
        s = sm;                       // [KP][D] slots (rows >= K are padding)
        sn = s + KP * D;              // [KB][D]
        q = sn + KB * D;              // [KB][D]
        u = q + KB * D;               // [KB][D]
        gi = u + KB * D;              // [KB][3D]
        gh = gi + KB * 3 * D;         // [KB][3D]
        hid = gh + KB * 3 * D;        // [KB][H]
        qp = hid + KB * H;            // [KP][C]
        up = qp + KP * C;             // [KP][C]
        qg = up + KP * C;             // [KP][C] gamma_in * q'
        cs = qg + KP * C;             // [16] weight sums
        qb = cs + 16;                 // [16] beta_in . q'
        tiles = qb + 16;              // [waves][16][SA_TLD]: streaming tiles, reduction scratch
    }
};

// slot side before the streaming pass of iteration t: LN(slots), q, q' = scale q Wk, and q' folded with the norm_inputs affine
template <int K>
__device__ __forceinline__ void sa_phase_a(const SaFwdLds<K>& L, const SlotAttnArgs& p, const SaWts& wo, const SaSave& so, float* sv0) {
    constexpr int C = SA_C, NB = SaBlk<K>::NB, KB = SaBlk<K>::KB;
    const int D = p.D;
    const float* W = p.wts;
#pragma unroll 1
    for (int hb = 0; hb < NB; ++hb) {
        const int j0 = hb * KB, kv = (K - j0) < KB ? (K - j0) : KB;
        float* sB = L.s + j0 * D;
        float* sv = sv0 ? sv0 + (size_t)j0 * so.ld : nullptr;
        if (sv) rows_to_global(sB, sv + so.sprev, so.ld, kv, D);
        ln_rows(sB, L.sn, W + wo.ln_s_g, W + wo.ln_s_b, KB, D);
        __syncthreads();
        matvec<KB>(W + wo.Wq, D, D, D, L.sn, D, L.q, D, nullptr, 1.f);
        matvec<KB>(W + wo.WkT, D, D, C, L.q, D, L.qp + j0 * C, C, nullptr, p.scale);
        if (sv) {
            rows_to_global(L.sn, sv + so.sn, so.ld, kv, D);
            rows_to_global(L.q, sv + so.q, so.ld, kv, D);
            rows_to_global(L.qp + j0 * C, sv + so.qp, so.ld, kv, C);
        }
        __syncthreads();
    }
    sa_fold_affine(L.qp, W + wo.ln_in_g, W + wo.ln_in_b, L.qg, L.qb, K);
    __syncthreads();
}

// weighted means from `nparts` partial sums part[w][j][0..63] = sum w xn, part[w][j][64] = sum w  (LDS or global memory)
template <int K>
__device__ __forceinline__ void sa_reduce_parts(const SaFwdLds<K>& L, const SlotAttnArgs& p, const SaWts& wo, const SaSave& so, const float* part, int nparts,
                                                float* sv0) {
    constexpr int C = SA_C;
    const int tid = threadIdx.x, nt = blockDim.x;
    const float* W = p.wts;
    if (tid < K) {
        float c0 = 0.f;
        for (int w = 0; w < nparts; ++w) c0 += part[(w * K + tid) * (C + 1) + C];
        L.cs[tid] = c0;
    }
    __syncthreads();
    for (int i = tid; i < K * C; i += nt) {
        const int j = i >> 6, c = i & 63;
        float a = 0.f;
        for (int w = 0; w < nparts; ++w) a += part[(w * K + j) * (C + 1) + c];
        a /= L.cs[j];                                                  // sum_n w xn / sum_n w
        const float v = a * W[wo.ln_in_g + c] + W[wo.ln_in_b + c];
        L.up[i] = v;
        if (sv0) { sv0[j * so.ld + so.upn + c] = a; sv0[j * so.ld + so.up + c] = v; }
    }
    if (sv0 && tid < K) sv0[tid * so.ld + so.csum] = L.cs[tid];
    __syncthreads();
}

// slot side after the streaming pass: updates = U' Wv^T ; GRU ; residual MLP
template <int K>
__device__ __forceinline__ void sa_phase_u(const SaFwdLds<K>& L, const SlotAttnArgs& p, const SaWts& wo, const SaSave& so, float* sv0) {
    constexpr int C = SA_C, NB = SaBlk<K>::NB, KB = SaBlk<K>::KB;
    const int D = p.D, H = p.H, tid = threadIdx.x, nt = blockDim.x;
    const float* W = p.wts;
    float *q = L.q, *u = L.u, *gi = L.gi, *gh = L.gh, *hid = L.hid, *sn = L.sn;
#pragma unroll 1
    for (int hb = 0; hb < NB; ++hb) {
        const int j0 = hb * KB, kv = (K - j0) < KB ? (K - j0) : KB;
        float* sB = L.s + j0 * D;
        float* sv = sv0 ? sv0 + (size_t)j0 * so.ld : nullptr;
        matvec<KB>(W + wo.Wv, C, C, D, L.up + j0 * C, C, u, D, nullptr, 1.f);
        matvec<KB>(W + wo.Wih, D, D, 3 * D, u, D, gi, 3 * D, W + wo.bih, 1.f);
        matvec<KB>(W + wo.Whh, D, D, 3 * D, sB, D, gh, 3 * D, W + wo.bhh, 1.f);
        for (int i = tid; i < KB * D; i += nt) {
            const int j = i / D, c = i - j * D;
            const float r = sigmoidf_(gi[j * 3 * D + c] + gh[j * 3 * D + c]);
            const float z = sigmoidf_(gi[j * 3 * D + D + c] + gh[j * 3 * D + D + c]);
            const float hn = gh[j * 3 * D + 2 * D + c];
            const float nn = tanhf(gi[j * 3 * D + 2 * D + c] + r * hn);
            const float sg = (1.f - z) * nn + z * sB[i];
            if (sv && j < kv) {
                float* row = sv + j * so.ld + c;
                row[so.u] = u[i]; row[so.r] = r; row[so.z] = z; row[so.n] = nn; row[so.hn] = hn; row[so.sg] = sg;
            }
            q[i] = sg;      // q is free now: holds s_gru
        }
        __syncthreads();
        ln_rows(q, sn, W + wo.ln_m_g, W + wo.ln_m_b, KB, D);      // sn = m
        __syncthreads();
        matvec<KB>(W + wo.W0, D, D, H, sn, D, hid, H, W + wo.b0, 1.f);
        for (int i = tid; i < KB * H; i += nt) hid[i] = fmaxf(hid[i], 0.f);
        __syncthreads();
        matvec<KB>(W + wo.W2, H, H, D, hid, H, u, D, W + wo.b2, 1.f);      // u = mlp out
        for (int i = tid; i < kv * D; i += nt) sB[i] = q[i] + u[i];
        if (sv) {
            rows_to_global(sn, sv + so.m, so.ld, kv, D);
            rows_to_global(hid, sv + so.hid, so.ld, kv, H);
        }
        __syncthreads();
    }
}

// One workgroup per image, all iterations in one launch, slots resident in LDS throughout.
template <int K>
__global__ __launch_bounds__(SA_TF) void slot_attn_fwd_kernel(SlotAttnArgs p, SaWts wo, SaSave so) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int C = SA_C, KP = SaBlk<K>::KP;
    const int D = p.D, N = p.N, nt = blockDim.x, nw = nt >> 6, tid = threadIdx.x;
    const SaFwdLds<K> L(sm, D, p.H);
    const int b = blockIdx.x;
    for (int i = tid; i < KP * D; i += nt) L.s[i] = i < K * D ? p.slots0[(size_t)b * K * D + i] : 0.f;
    __syncthreads();
    const float4* xb = reinterpret_cast<const float4*>(p.x + (size_t)b * N * C);
    for (int t = 0; t < p.I; ++t) {
        float* sv0 = p.save ? p.save + ((size_t)b * p.I + t) * K * so.ld : nullptr;   // K rows of the save matrix
        sa_phase_a<K>(L, p, wo, so, sv0);
        sa_stream_fwd<K>(xb, N, L.qg, L.qb, p.eps, (t == p.I - 1 && p.attn) ? p.attn + (size_t)b * N * K : nullptr, L.tiles, L.tiles);
        __syncthreads();
        sa_reduce_parts<K>(L, p, wo, so, L.tiles, nw, sv0);
        sa_phase_u<K>(L, p, wo, so, sv0);
    }
    for (int i = tid; i < K * D; i += nt) p.slots[(size_t)b * K * D + i] = L.s[i];
}

// Split form for batches that leave CUs idle (B < #CUs): NS workgroups per image and ONE LAUNCH PER ITERATION.  Every workgroup
// streams 1/NS of the positions and writes its partial sums; the workgroup that arrives last at the image's counter (agent-scope
// release / acquire around one atomic, no waiting on anybody) reduces the partials, runs the slot update and prepares q' of the
// next iteration.  Between launches an image's slots and folded query live in `xchg` (a few KB per image); within a launch they
// are in LDS as in the fused kernel.  t = -1 is the preparation launch (one workgroup per image: q' of iteration 0).
//   xchg per image: slots [KP*D] | qg [KP*64] | qb [16] | parts [NS][K][65]
__host__ __device__ inline size_t sa_xchg_floats(int K, int D, int NS) {
    const int KP = K > 8 ? 2 * ((K + 1) / 2) : K;
    return (size_t)KP * D + (size_t)KP * SA_C + 16 + (size_t)NS * K * (SA_C + 1) + 15;
}
template <int K>
__global__ __launch_bounds__(SA_TF) void slot_attn_fwd_split_kernel(SlotAttnArgs p, SaWts wo, SaSave so, int t, int NS) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int C = SA_C, KP = SaBlk<K>::KP;
    const int D = p.D, N = p.N, nt = blockDim.x, nw = nt >> 6, tid = threadIdx.x;
    const SaFwdLds<K> L(sm, D, p.H);
    const int b = t < 0 ? blockIdx.x : blockIdx.x / NS, h = t < 0 ? 0 : blockIdx.x % NS;
    float* xg = p.xchg + (size_t)b * sa_xchg_floats(K, D, NS);
    float* g_slots = xg;
    float* g_qg = g_slots + KP * D;
    float* g_qb = g_qg + KP * C;
    float* g_parts = g_qb + 16;
    if (t < 0) {
        for (int i = tid; i < KP * D; i += nt) L.s[i] = i < K * D ? p.slots0[(size_t)b * K * D + i] : 0.f;
        __syncthreads();
        sa_phase_a<K>(L, p, wo, so, p.save ? p.save + (size_t)b * p.I * K * so.ld : nullptr);
        for (int i = tid; i < KP * D; i += nt) g_slots[i] = L.s[i];
        for (int i = tid; i < K * C; i += nt) g_qg[i] = L.qg[i];
        if (tid < K) g_qb[tid] = L.qb[tid];
        return;
    }
    for (int i = tid; i < K * C; i += nt) L.qg[i] = g_qg[i];
    if (tid < K) L.qb[tid] = g_qb[tid];
    __syncthreads();
    const float4* xb = reinterpret_cast<const float4*>(p.x + (size_t)b * N * C);
    const int ntile = (N + 15) / 16, per = (ntile + NS - 1) / NS;
    const int t0 = h * per, t1 = (t0 + per < ntile) ? t0 + per : ntile;
    sa_stream_fwd<K>(xb, N, L.qg, L.qb, p.eps, (t == p.I - 1 && p.attn) ? p.attn + (size_t)b * N * K : nullptr, L.tiles, L.tiles, t0, t1 > t0 ? t1 : t0);
    __syncthreads();
    for (int i = tid; i < K * (C + 1); i += nt) {          // this workgroup's partial: sum over its waves
        float a = 0.f;
        for (int w = 0; w < nw; ++w) a += L.tiles[w * K * (C + 1) + i];
        g_parts[(size_t)h * K * (C + 1) + i] = a;
    }
    __threadfence();                                       // release (agent scope) of every thread's share of the partial ...
    __syncthreads();                                       // ... ordered before the count below; the tile region is free again
    int* s_last = reinterpret_cast<int*>(L.tiles);
    if (tid == 0) *s_last = atomicAdd(&p.counters[(size_t)b * p.I + t], 1) == NS - 1;
    __syncthreads();
    if (!*s_last) return;
    __syncthreads();
    __threadfence();                                       // acquire: the other workgroups' partials
    float* sv0 = p.save ? p.save + ((size_t)b * p.I + t) * K * so.ld : nullptr;
    for (int i = tid; i < KP * D; i += nt) L.s[i] = g_slots[i];
    __syncthreads();
    sa_reduce_parts<K>(L, p, wo, so, g_parts, NS, sv0);
    sa_phase_u<K>(L, p, wo, so, sv0);
    if (t == p.I - 1) {
        for (int i = tid; i < K * D; i += nt) p.slots[(size_t)b * K * D + i] = L.s[i];
        return;
    }
    sa_phase_a<K>(L, p, wo, so, p.save ? p.save + ((size_t)b * p.I + t + 1) * K * so.ld : nullptr);
    for (int i = tid; i < KP * D; i += nt) g_slots[i] = L.s[i];
    for (int i = tid; i < K * C; i += nt) g_qg[i] = L.qg[i];
    if (tid < K) g_qb[tid] = L.qb[tid];
}

// ------------------------------------------------------------------------------------------- backward streaming pass
// recomputes attn from xn and the folded operands (qg = gamma q', qb = beta.q', dug = gamma dU', dub = beta.dU');
// partials scr[wave][j][0..63] = sum_n dlogits[n,j] xn[n], scr[wave][j][64] = sum_n dlogits[n,j] (scr aliases the
// tiles: barrier inside); d xn written (FIRST), accumulated, or (FINAL) pushed through the LayerNorm backward into dx.
template <int K, bool FIRST, bool FINAL>
__device__ __forceinline__ void sa_stream_bwd(const float4* __restrict__ xb, float4* __restrict__ dxb, int N, const float* qg, const float* qb,
                                              const float* dug, const float* dub, const float* cs, const float* ud, float eps, float* tiles,
                                              float* scr) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6, li = lane & 15, g = lane >> 4;
    float* tile = tiles + wv * (16 * SA_TLD + 2 * 16 * SA_WLD);
    float* wt0 = tile + 16 * SA_TLD;
    float* wt1 = wt0 + 16 * SA_WLD;
    float qpr[16], dur[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int ch = 16 * (k >> 2) + 4 * g + (k & 3);
        qpr[k] = li < K ? qg[li * SA_C + ch] : 0.f;
        dur[k] = li < K ? dug[li * SA_C + ch] : 0.f;
    }
    // B operands of d xn[pos][ch] = sum_slot (wn[pos][slot] dug[slot][ch] + dl[pos][slot] qg[slot][ch]), slot = 4s + g
    constexpr int NS = (K + 3) / 4;
    float duB[NS][4], qpB[NS][4];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int slot = 4 * s + g;
            duB[s][m] = slot < K ? dug[slot * SA_C + 16 * m + li] : 0.f;
            qpB[s][m] = slot < K ? qg[slot * SA_C + 16 * m + li] : 0.f;
        }
    const float icv = li < K ? 1.0f / cs[li] : 0.f, udv = li < K ? ud[li] : 0.f;
    const float l0 = li < K ? qb[li] : 0.f, d0 = li < K ? dub[li] : 0.f;
    f32x4_t acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float sdl = 0.f;
    const int ntile = (N + 15) / 16;
    float4 cur[4];
    if (wv < ntile) sa_load_tile(xb, N, wv, li, g, cur);
#pragma unroll 1
    for (int t = wv; t < ntile; t += nw) {
        float4 nxt[4];
        if (t + nw < ntile) sa_load_tile(xb, N, t + nw, li, g, nxt);
        float xn[16];
        const float rs = sa_ln16(cur, xn);
        // the running d xn of this tile (earlier iterations) is fetched now and added after the tile's 64 MFMAs
        float4 od[4];
        if (!FIRST) {
            const int pos0 = t * 16 + li;
#pragma unroll
            for (int c = 0; c < 4; ++c) od[c] = pos0 < N ? dxb[pos0 * 16 + 4 * c + g] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        f32x4_t L = (f32x4_t){l0, l0, l0, l0}, DA = (f32x4_t){d0, d0, d0, d0};
#pragma unroll
        for (int k = 0; k < 16; ++k) { L = MFMA16(xn[k], qpr[k], L); DA = MFMA16(xn[k], dur[k], DA); }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            *reinterpret_cast<float4*>(tile + li * SA_TLD + 16 * c + 4 * g) = make_float4(xn[4 * c], xn[4 * c + 1], xn[4 * c + 2], xn[4 * c + 3]);
        f32x4_t dl, wn;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x = li < K ? L[r] : -INFINITY;
            const float mx = red16_max(x);
            const float e = li < K ? __expf(x - mx) : 0.f;
            const float a = e / red16_sum(e);
            const float da = (DA[r] - udv) * icv;                 // d attn[pos][slot] (0 for padded slots: icv = 0)
            const float dot = red16_sum(a * da);
            const bool ok = li < K && (t * 16 + 4 * g + r) < N;
            dl[r] = ok ? a * (da - dot) : 0.f;                    // d logits
            wn[r] = ok ? (a + eps) * icv : 0.f;                   // normalised weight
            sdl += dl[r];
            wt0[(4 * g + r) * SA_WLD + li] = wn[r];
            wt1[(4 * g + r) * SA_WLD + li] = dl[r];
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = MFMA16(tile[(4 * g + r) * SA_TLD + 16 * m + li], dl[r], acc[m]);   // sum_n dl xn^T: [ch][slot]
        f32x4_t Dx[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) Dx[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const float wa = wt0[li * SA_WLD + 4 * s + g], da2 = wt1[li * SA_WLD + 4 * s + g];     // A layout: [pos li][slot 4s+g]
#pragma unroll
            for (int m = 0; m < 4; ++m) { Dx[m] = MFMA16(wa, duB[s][m], Dx[m]); Dx[m] = MFMA16(da2, qpB[s][m], Dx[m]); }
        }
        __builtin_amdgcn_wave_barrier();          // all reads of the xn tile are issued: reuse it to transpose d xn
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[(4 * g + r) * SA_TLD + 16 * m + li] = Dx[m][r];
        __builtin_amdgcn_wave_barrier();
        float dv[16];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 d4 = *reinterpret_cast<const float4*>(tile + li * SA_TLD + 16 * c + 4 * g);
            dv[4 * c] = d4.x; dv[4 * c + 1] = d4.y; dv[4 * c + 2] = d4.z; dv[4 * c + 3] = d4.w;
        }
        __builtin_amdgcn_wave_barrier();
        const int pos = t * 16 + li;
        if (!FIRST) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { dv[4 * c] += od[c].x; dv[4 * c + 1] += od[c].y; dv[4 * c + 2] += od[c].z; dv[4 * c + 3] += od[c].w; }
        }
        if (FINAL) {      // LayerNorm(norm_inputs) backward from d xn, in the load layout
            float m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) { m1 += dv[k]; m2 += dv[k] * xn[k]; }
            m1 = redg_sum(m1) * (1.0f / SA_C);
            m2 = redg_sum(m2) * (1.0f / SA_C);
#pragma unroll
            for (int k = 0; k < 16; ++k) dv[k] = rs * (dv[k] - m1 - xn[k] * m2);
        }
        if (pos < N) {
#pragma unroll
            for (int c = 0; c < 4; ++c) dxb[pos * 16 + 4 * c + g] = make_float4(dv[4 * c], dv[4 * c + 1], dv[4 * c + 2], dv[4 * c + 3]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) cur[c] = nxt[c];
    }
    sdl = redg_sum(sdl);
    __syncthreads();
    if (li < K) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) scr[(wv * K + li) * (SA_C + 1) + 16 * m + 4 * g + r] = acc[m][r];
        if (g == 0) scr[(wv * K + li) * (SA_C + 1) + SA_C] = sdl;
    }
}

template <int K>
__global__ __launch_bounds__(SA_TB) void slot_attn_bwd_kernel(SlotAttnArgs p, SaWts wo, SaSave so, SaGrad go) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int D = p.D, H = p.H, N = p.N;
    constexpr int C = SA_C, NB = SaBlk<K>::NB, KB = SaBlk<K>::KB, KP = SaBlk<K>::KP;
    const int nt = blockDim.x, nw = nt >> 6;
    float* ds = sm;                      // [KP][D] gradient wrt the iteration output
    float* t2 = ds + KP * D;             // [KP][D] dh (direct GRU path), kept across the streaming pass
    float* t0 = t2 + KP * D;             // [KB][D] scratch
    float* t1 = t0 + KB * D;             // [KB][D] scratch
    float* dgi = t1 + KB * D;            // [KB][3D]
    float* dgh = dgi + KB * 3 * D;       // [KB][3D]
    float* dhid = dgh + KB * 3 * D;      // [KB][H]
    float* qp = dhid + KB * H;           // [KP][C]
    float* dup = qp + KP * C;            // [KP][C]
    float* dqp = dup + KP * C;           // [KP][C]
    float* qg = dqp + KP * C;            // [KP][C] gamma_in * q'
    float* dug = qg + KP * C;            // [KP][C] gamma_in * dU'
    float* cs = dug + KP * C;            // [16] csum
    float* ud = cs + 16;                 // [16] up . dup
    float* qb = ud + 16;                 // [16] beta_in . q'
    float* dub = qb + 16;                // [16] beta_in . dU'
    float* sdl = dub + 16;               // [16] sum_n dlogits
    float* gacc = sdl + 16;              // dgamma/dbeta accumulators: ln_s (2D), ln_m (2D), ln_in (2C)
    float* tiles = gacc + 4 * D + 2 * C; // [waves][16*SA_TLD + 2*16*SA_WLD]: streaming tiles, matvec / reduction scratch

    const int b = blockIdx.x;
    const int tid = threadIdx.x;
    const float* W = p.wts;
    for (int i = tid; i < KP * D; i += nt) ds[i] = i < K * D ? p.dslots[(size_t)b * K * D + i] : 0.f;
    for (int i = tid; i < 4 * D + 2 * C; i += nt) gacc[i] = 0.f;
    __syncthreads();
    float* dg_s = gacc;  float* db_s = gacc + D;
    float* dg_m = gacc + 2 * D;  float* db_m = gacc + 3 * D;
    float* dg_in = gacc + 4 * D;  float* db_in = gacc + 4 * D + C;
    const float4* xb = reinterpret_cast<const float4*>(p.x + (size_t)b * N * C);
    float4* dxb = reinterpret_cast<float4*>(p.dx + (size_t)b * N * C);

    for (int t = p.I - 1; t >= 0; --t) {
        const float* sv0 = p.save + ((size_t)b * p.I + t) * K * so.ld;
        float* gr0 = p.grows + ((size_t)b * p.I + t) * K * go.ld;
#pragma unroll 1
        for (int hb = 0; hb < NB; ++hb) {
            const int j0 = hb * KB, kv = (K - j0) < KB ? (K - j0) : KB;
            const float* sv = sv0 + (size_t)j0 * so.ld;
            float* gr = gr0 + (size_t)j0 * go.ld;
            float* dsB = ds + j0 * D;
            float* t2B = t2 + j0 * D;
            // ---- residual MLP backward: s_new = sg + W2 relu(W0 m + b0) + b2,  m = LN_m(sg)
            rows_to_global(dsB, gr + go.out, go.ld, kv, D);
            rows_from_global(t0, sv + so.sg, so.ld, kv, D);
            matvec<KB>(W + wo.W2T, D, D, H, dsB, D, dhid, H, nullptr, 1.f);          // dhid[h] = sum_d ds[d] W2[d][h]
            for (int i = tid; i < kv * H; i += nt) {
                const int j = i / H, c = i - j * H;
                const float v = sv[j * so.ld + so.hid + c] > 0.f ? dhid[i] : 0.f;
                dhid[i] = v;
                gr[j * go.ld + go.hid + c] = v;
            }
            __syncthreads();
            matvec<KB>(W + wo.W0T, H, H, D, dhid, H, t1, D, nullptr, 1.f);           // dm[e] = sum_h dhid[h] W0[h][e]
            ln_rows_bwd(t1, t0, dsB, 1, W + wo.ln_m_g, dg_m, db_m, kv, D);                 // ds = d s_gru
            __syncthreads();
            // ---- GRU backward
            for (int i = tid; i < kv * D; i += nt) {
                const int j = i / D, c = i - j * D;
                const float* row = sv + j * so.ld + c;
                const float r = row[so.r], z = row[so.z], nn = row[so.n], hn = row[so.hn], h = row[so.sprev];
                const float gg = dsB[i];
                const float dn_pre = gg * (1.f - z) * (1.f - nn * nn);
                const float dz_pre = gg * (h - nn) * z * (1.f - z);
                const float dr_pre = dn_pre * hn * r * (1.f - r);
                dgi[j * 3 * D + c] = dr_pre; dgi[j * 3 * D + D + c] = dz_pre; dgi[j * 3 * D + 2 * D + c] = dn_pre;
                dgh[j * 3 * D + c] = dr_pre; dgh[j * 3 * D + D + c] = dz_pre; dgh[j * 3 * D + 2 * D + c] = dn_pre * r;
                t2B[i] = gg * z;               // dh (direct path)
            }
            __syncthreads();
            rows_to_global(dgi, gr + go.gi, go.ld, kv, 3 * D);
            rows_to_global(dgh, gr + go.gh, go.ld, kv, 3 * D);
            matvec<KB>(W + wo.WihT, 3 * D, 3 * D, D, dgi, 3 * D, t1, D, nullptr, 1.f);   // du[e] = sum_g dgi[g] Wih[g][e]
            matvec<KB>(W + wo.WhhT, 3 * D, 3 * D, D, dgh, 3 * D, t0, D, nullptr, 1.f);   // dh via the recurrent weights
            for (int i = tid; i < kv * D; i += nt) t2B[i] += t0[i];
            rows_to_global(t1, gr + go.u, go.ld, kv, D);
            // ---- u = up Wv^T
            matvec<KB>(W + wo.WvT, D, D, C, t1, D, dup + j0 * C, C, nullptr, 1.f);   // dup[c] = sum_d du[d] Wv[d][c]
        }
        rows_from_global(qp, sv0 + so.qp, so.ld, K, C);
        if (tid < K) {
            cs[tid] = sv0[tid * so.ld + so.csum];
            float a = 0.f;
            for (int c = 0; c < C; ++c) a += sv0[tid * so.ld + so.up + c] * dup[tid * C + c];
            ud[tid] = a;
        }
        __syncthreads();
        sa_fold_affine(qp, W + wo.ln_in_g, W + wo.ln_in_b, qg, qb, K);
        sa_fold_affine(dup, W + wo.ln_in_g, W + wo.ln_in_b, dug, dub, K);
        __syncthreads();
        // ---- streaming pass: recompute attn, accumulate sum dl xn, write / accumulate d xn
        const bool first = (t == p.I - 1), final_ = (t == 0);
        if (first && final_) sa_stream_bwd<K, true, true>(xb, dxb, N, qg, qb, dug, dub, cs, ud, p.eps, tiles, tiles);
        else if (first) sa_stream_bwd<K, true, false>(xb, dxb, N, qg, qb, dug, dub, cs, ud, p.eps, tiles, tiles);
        else if (final_) sa_stream_bwd<K, false, true>(xb, dxb, N, qg, qb, dug, dub, cs, ud, p.eps, tiles, tiles);
        else sa_stream_bwd<K, false, false>(xb, dxb, N, qg, qb, dug, dub, cs, ud, p.eps, tiles, tiles);
        __syncthreads();
        if (tid < K) {
            float a = 0.f;
            for (int w = 0; w < nw; ++w) a += tiles[(w * K + tid) * (C + 1) + C];
            sdl[tid] = a;
        }
        for (int i = tid; i < K * C; i += nt) {
            const int j = i >> 6, c = i & 63;
            float a = 0.f;
            for (int w = 0; w < nw; ++w) a += tiles[(w * K + j) * (C + 1) + c];
            dqp[i] = a;                                          // sum_n dl xn (pre-affine)
        }
        __syncthreads();
        // norm_inputs gamma/beta:  d xa = wn dU' + dl q'  =>  dgamma = sum_j dU' upn + q' dqn,  dbeta = sum_j dU' + q' sdl
        if (tid < C) {
            float dg = 0.f, db = 0.f;
            for (int j = 0; j < K; ++j) {
                dg += dup[j * C + tid] * sv0[j * so.ld + so.upn + tid] + qp[j * C + tid] * dqp[j * C + tid];
                db += dup[j * C + tid] + qp[j * C + tid] * sdl[j];
            }
            dg_in[tid] += dg;
            db_in[tid] += db;
        }
        __syncthreads();
        for (int i = tid; i < K * C; i += nt) {
            const int j = i >> 6, c = i & 63;
            dqp[i] = dqp[i] * W[wo.ln_in_g + c] + W[wo.ln_in_b + c] * sdl[j];   // d q' = sum_n dl LN(x)
        }
        __syncthreads();
        rows_to_global(dqp, gr0 + go.qp, go.ld, K, C);
#pragma unroll 1
        for (int hb = 0; hb < NB; ++hb) {
            const int j0 = hb * KB, kv = (K - j0) < KB ? (K - j0) : KB;
            const float* sv = sv0 + (size_t)j0 * so.ld;
            float* gr = gr0 + (size_t)j0 * go.ld;
            float* dsB = ds + j0 * D;
            // ---- q' = scale q Wk  ->  dq[d] = scale sum_c dqp[c] Wk[d][c] ;  q = sn Wq^T  ->  dsn[e] = sum_d dq[d] Wq[d][e]
            matvec<KB>(W + wo.Wk, C, C, D, dqp + j0 * C, C, t1, D, nullptr, p.scale);   // t1 = dq
            rows_to_global(t1, gr + go.q, go.ld, kv, D);
            rows_from_global(t0, sv + so.sprev, so.ld, kv, D);
            matvec<KB>(W + wo.WqT, D, D, D, t1, D, dsB, D, nullptr, 1.f);                 // ds = dsn
            for (int i = tid; i < kv * D; i += nt) { t1[i] = dsB[i]; dsB[i] = t2[j0 * D + i]; }
            __syncthreads();
            ln_rows_bwd(t1, t0, dsB, 1, W + wo.ln_s_g, dg_s, db_s, kv, D);                      // ds = dh + LN_s backward
            __syncthreads();
        }
    }
    for (int i = tid; i < K * D; i += nt) p.dslots0[(size_t)b * K * D + i] = ds[i];
    for (int i = tid; i < 4 * D + 2 * C; i += nt) p.g_small[(size_t)b * (4 * D + 2 * C) + i] = gacc[i];
}

static size_t sa_fwd_smem(int K, int D, int H) {
    const int NB = K > 8 ? 2 : 1, KB = (K + NB - 1) / NB, KP = NB * KB;
    const size_t tiles = (size_t)(SA_TF / 64) * 16 * SA_TLD;
    return (size_t)(KP * D + KB * D * 3 + KB * 3 * D * 2 + KB * H + KP * SA_C * 3 + 32 + tiles) * 4;
}
size_t sa_xchg_floats_host(int K, int D) { return sa_xchg_floats(K, D, SA_MAX_SPLIT); }
static size_t sa_bwd_smem(int K, int D, int H) {
    const int NB = K > 8 ? 2 : 1, KB = (K + NB - 1) / NB, KP = NB * KB;
    const size_t tiles = (size_t)(SA_TB / 64) * (16 * SA_TLD + 2 * 16 * SA_WLD);
    return (size_t)(KP * D * 2 + KB * D * 2 + KB * 3 * D * 2 + KB * H + KP * SA_C * 5 + 80 + 4 * D + 2 * SA_C + tiles) * 4;
}

// workgroups per image of the split forward: fill the CUs (one workgroup per CU: the kernel owns most of the LDS), at least 32
// tiles per workgroup; 1 = fused single-launch kernel.  OCRL_SA_SPLIT=0 disables the split form.
int sa_fwd_splits(const SlotAttnArgs& a) {
    static int ncu = 0, mode = -1;
    if (mode < 0) { const char* e = getenv("OCRL_SA_SPLIT"); mode = e ? atoi(e) : 1; }
    if (!mode || !a.xchg || !a.counters) return 1;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 1;
        ncu = prop.multiProcessorCount;
    }
    int ns = 1;
    const int ntile = (a.N + 15) / 16;
    while (ns < SA_MAX_SPLIT && a.B * ns * 2 <= ncu && ntile / (ns * 2) >= 32) ns *= 2;
    return ns;
}

template <int K>
static int sa_launch_k(const SlotAttnArgs& a, int backward, hipStream_t st) {
    const size_t smem = backward ? sa_bwd_smem(K, a.D, a.H) : sa_fwd_smem(K, a.D, a.H);
    OCRL_REQUIRE(smem <= 160 * 1024, "slot_attn: LDS request %zu too large (num_slots %d, slot size %d)", smem, K, a.D);
    // the shared tile region doubles as matvec / reduction scratch: check it is large enough
    constexpr int KB = SaBlk<K>::KB;
    const size_t tiles_f = (size_t)(SA_TF / 64) * 16 * SA_TLD, tiles_b = (size_t)(SA_TB / 64) * (16 * SA_TLD + 2 * 16 * SA_WLD);
    const size_t part_f = (size_t)(SA_TF / 64) * K * (SA_C + 1), part_b = (size_t)(SA_TB / 64) * K * (SA_C + 1);
    const size_t mv_f = (size_t)16 * KB * 64 > (size_t)(SA_TF / 64) * KB * 64 ? (size_t)16 * KB * 64 : (size_t)(SA_TF / 64) * KB * 64;   // G * KB * NC <= threads * KB
    const size_t mv_b = (size_t)(SA_TB / 64) * KB * 64;
    OCRL_REQUIRE(tiles_f >= part_f && tiles_f >= mv_f && tiles_b >= part_b && tiles_b >= mv_b, "slot_attn: scratch region too small");
    const SaWts wo = sa_wts_layout(a.C, a.D, a.H);
    const SaSave so = sa_save_layout(a.C, a.D, a.H);
    const SaGrad go = sa_grad_layout(a.C, a.D, a.H);
    const int pi = prof_begin(backward ? PROF_SA_BWD : PROF_SA_FWD, st);
    if (backward) {
        OCRL_HIP(hipFuncSetAttribute((const void*)slot_attn_bwd_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        hipLaunchKernelGGL((slot_attn_bwd_kernel<K>), dim3(a.B), dim3(SA_TB), smem, st, a, wo, so, go);
    } else {
        const int NS = sa_fwd_splits(a);
        if (NS > 1) {
            OCRL_HIP(hipFuncSetAttribute((const void*)slot_attn_fwd_split_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            OCRL_HIP(hipMemsetAsync(a.counters, 0, sizeof(int) * (size_t)a.B * a.I, st));
            hipLaunchKernelGGL((slot_attn_fwd_split_kernel<K>), dim3(a.B), dim3(SA_TF), smem, st, a, wo, so, -1, NS);
            for (int t = 0; t < a.I; ++t) hipLaunchKernelGGL((slot_attn_fwd_split_kernel<K>), dim3(a.B * NS), dim3(SA_TF), smem, st, a, wo, so, t, NS);
        } else {
            OCRL_HIP(hipFuncSetAttribute((const void*)slot_attn_fwd_kernel<K>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            hipLaunchKernelGGL((slot_attn_fwd_kernel<K>), dim3(a.B), dim3(SA_TF), smem, st, a, wo, so);
        }
    }
    prof_end(pi, st);
    OCRL_CHECK_LAUNCH("slot_attn");
    return 0;
}

int slot_attn_launch(const SlotAttnArgs& a, int backward, hipStream_t st) {
    OCRL_REQUIRE(a.C == SA_C, "slot_attn: input width must be %d (got %d)", SA_C, a.C);
    OCRL_REQUIRE(a.K >= 1 && a.K <= 16, "slot_attn: 1 <= num_slots <= 16 supported (got %d)", a.K);
    OCRL_REQUIRE(a.D % 64 == 0 && a.H % 64 == 0 && a.D <= 256 && a.H <= 256, "slot_attn: slot/mlp size must be multiples of 64, <= 256");
    OCRL_REQUIRE(a.B > 0 && a.N > 0 && a.I >= 1, "slot_attn: empty problem");
    OCRL_REQUIRE((long long)a.N * 16 < (1ll << 31), "slot_attn: N too large for 32-bit row offsets");
    OCRL_REQUIRE(a.x && a.wts && ((uintptr_t)a.x & 15) == 0, "slot_attn: x/wts missing or x not 16-byte aligned");
    if (backward) OCRL_REQUIRE(a.dx && ((uintptr_t)a.dx & 15) == 0 && a.save && a.grows && a.dslots && a.dslots0 && a.g_small, "slot_attn bwd: missing buffers");
    else OCRL_REQUIRE(a.slots0 && a.slots, "slot_attn fwd: missing buffers");
    switch (a.K) {
#define SA_CASE(k) case k: return sa_launch_k<k>(a, backward, st);
        SA_CASE(1) SA_CASE(2) SA_CASE(3) SA_CASE(4) SA_CASE(5) SA_CASE(6) SA_CASE(7) SA_CASE(8)
        SA_CASE(9) SA_CASE(10) SA_CASE(11) SA_CASE(12) SA_CASE(13) SA_CASE(14) SA_CASE(15) SA_CASE(16)
#undef SA_CASE
    }
    return -1;
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
