// Slot attention (reference: ocrs/common/slot_attn.py:47-102), forward and backward.
//
// Each iteration is two launches: a STREAMING launch over the positions (several workgroups per image, every CU busy at any batch
// size) and a SLOT-SIDE launch (GRU, MLP, LayerNorms, projections of the K slots; one workgroup per group of images, slot state in
// LDS).  The k/v projections are folded algebraically so the [N,D] k and v tensors are never materialised:
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
#define SA_MAX_BLOCKS 2048 // upper bound of streaming workgroups per launch the partial buffer is sized for (plus one per image)

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
// The weights are stored tiled for this loop (pack_kernel modes 2 / 3): matrix Wn[col][e] (col < NCt, e < Et) as [col / 16][e / 16][lane][4]
// with lane = 16 * ((e % 16) / 4) + col % 16 -- the float4 a lane needs for 16 output columns x 16 values of e is one contiguous 1 KiB
// block per load instruction (the row-major form touched 16 half-used cache lines per instruction).  A job may address a sub-block:
// column tiles from t0, k-steps from s0 (the per-head products of multi-head slot attention).
struct MvJob {           // one product of the slot-side chain: out[j][col] = scale * in[j] . Wn[col] + bias[col]
    const float* Wt; int nkt, t0, s0, E, NC; const float* in; int ldin; float* out; int ldout; const float* bias; float scale;
};
// one 16-column output tile of a job (all K rows)
template <int K>
__device__ __forceinline__ void mv_tile(const MvJob& J, int tile) {
    const int lane = threadIdx.x & 63, li = lane & 15, g = lane >> 4;
    const int nks = J.E >> 4;
    const float* arow = J.in + (li < K ? li : 0) * J.ldin + 4 * g;
    const int col = tile * 16 + li;
    const float* wrow = J.Wt + ((size_t)(J.t0 + tile) * J.nkt + J.s0) * 256 + lane * 4;
    f32x4_t acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    // the weight row is fetched twelve k-steps (192 values of e) at a time, all loads issued before the first MFMA: one L2 round
    // trip per 192 e instead of one per 16
    for (int s0 = 0; s0 < nks; s0 += 12) {
        float4 bw[12];
#pragma unroll
        for (int u = 0; u < 12; ++u)
            if (s0 + u < nks) bw[u] = *reinterpret_cast<const float4*>(wrow + 256 * (s0 + u));
        __builtin_amdgcn_sched_barrier(0);          // keep the loads above: the scheduler otherwise sinks them next to their MFMAs (2 in flight)
#pragma unroll
        for (int u = 0; u < 12; ++u)
            if (s0 + u < nks) {
                float4 a = *reinterpret_cast<const float4*>(arow + 16 * (s0 + u));
                if (li >= K) a = make_float4(0.f, 0.f, 0.f, 0.f);
                acc = MFMA16(a.x, bw[u].x, acc);
                acc = MFMA16(a.y, bw[u].y, acc);
                acc = MFMA16(a.z, bw[u].z, acc);
                acc = MFMA16(a.w, bw[u].w, acc);
            }
    }
    const float bv = J.bias ? J.bias[col] : 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (4 * g + r < K) J.out[(4 * g + r) * J.ldout + col] = acc[r] * J.scale + bv;      // acc[r] = row (slot) 4g + r, column li
}
// out[j][col] = scale * sum_e in[j][e] * Wn[col*ldw + e] + bias[col],  col < NC, j < K   (in/out in LDS; E, NC multiples of 16).
// The K <= 16 slot rows are the M side of v_mfma_f32_16x16x4_f32 (rows >= K fed as zeros), a wave owns 16-column tiles of the
// output; per 16 values of e a lane issues one float4 weight load (the k-contiguous orientation of the weight, L2 resident) and
// one ds_read_b128 of its slot row and feeds 4 MFMAs.  Ends with a workgroup barrier.
template <int K>
__device__ __forceinline__ void matvec(const float* __restrict__ Wt, int nkt, int t0, int s0, int E, int NC, const float* in, int ldin, float* out, int ldout,
                                       const float* __restrict__ bias, float scale) {
    const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const MvJob J = {Wt, nkt, t0, s0, E, NC, in, ldin, out, ldout, bias, scale};
    for (int tile = wv; tile < (NC >> 4); tile += nw) mv_tile<K>(J, tile);
    __syncthreads();
}
// two independent products in one pass: their tiles are dealt to the waves as one list, so the waves that would idle in the last
// round of the first product already work on the second (GRU input / hidden gates: 36 + 36 tiles = 4.5 rounds of 16 waves, not 3 + 3)
template <int K>
__device__ __forceinline__ void matvec2(const MvJob& A, const MvJob& B) {
    const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int na = A.NC >> 4, nb = B.NC >> 4;
    for (int tile = wv; tile < na + nb; tile += nw) {
        if (tile < na) mv_tile<K>(A, tile);
        else mv_tile<K>(B, tile - na);
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

// ------------------------------------------------------------------------------------------- forward streaming pass, second form
// Same products, transposed: logits^T[slot][pos] = q' . d^T, so that a lane holds its OWN position's logits of four slots in its four
// accumulator registers (lane = (pos li, slot group g'), register r = slot 4g' + r).  The soft-max over the slots is then in-lane
// arithmetic plus one or two cross-row exchanges (v_permlane16/32_swap, VALU speed) instead of two 4-step DPP reductions per
// register; the LayerNorm scale rstd[pos] is lane-local and is folded into the logits (acc * rstd + bias) and into the weights of the
// second product (w * rstd), so x is only centred, not normalised; q' arrives pre-multiplied by log2(e) and the exponentials are bare
// v_exp_f32; the quotient is a v_rcp_f32.  The weights go through a 16x16 LDS patch to reach the operand layout of the second product
// (4 ds_write_b32 + 1 ds_read_b128).  Measured on the round-2 form: ~300 VALU instructions per 16-position tile against 32 MFMAs, i.e.
// VALU-bound with the matrix pipe 38 % busy; this form needs ~120.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
#define SA_WT_LD 20       // row stride of the per-wave [16 slots][16 positions] weight patch
#define SA_LOG2E 1.4426950408889634f
__device__ __forceinline__ float sa_xrow16(float v, bool is_max) {      // combine with the lane 16 apart (rows 0<->1, 2<->3)
    const int a = __builtin_bit_cast(int, v);
    const auto r = __builtin_amdgcn_permlane16_swap(a, a, false, false);
    const float x = __builtin_bit_cast(float, (int)r[0]), y = __builtin_bit_cast(float, (int)r[1]);
    return is_max ? __builtin_amdgcn_fmed3f(x, y, INFINITY) : x + y;
}
__device__ __forceinline__ float sa_xrow32(float v, bool is_max) {      // combine with the lane 32 apart (rows 0<->2, 1<->3)
    const int a = __builtin_bit_cast(int, v);
    const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    const float x = __builtin_bit_cast(float, (int)r[0]), y = __builtin_bit_cast(float, (int)r[1]);
    return is_max ? __builtin_amdgcn_fmed3f(x, y, INFINITY) : x + y;
}
__device__ __forceinline__ float sa_xrow_sum4(float v) { return sa_xrow32(sa_xrow16(v, false), false); }    // over the 4 lanes of a position
template <int K>
__device__ __forceinline__ void sa_stream_fwd2(const float4* __restrict__ xb, int N, const float* qg, const float* qb, float eps, float* attn_out,
                                               float* tiles, float* scr, int tile0, int tile1) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6, li = lane & 15, g = lane >> 4;
    float* tile = tiles + wv * (16 * SA_TLD + 16 * SA_WT_LD);
    float* wt = tile + 16 * SA_TLD;
    float qpr[16];                                       // A operand of the logits: q'[slot li][channel 16(k>>2) + 4g + (k&3)] * log2(e)
#pragma unroll
    for (int k = 0; k < 16; ++k) qpr[k] = li < K ? qg[li * SA_C + 16 * (k >> 2) + 4 * g + (k & 3)] * SA_LOG2E : 0.f;
    float qbr[4];                                        // bias of slot 4g + r in log2 units; padding slots sit at -1e30 (their weight is 0)
#pragma unroll
    for (int r = 0; r < 4; ++r) qbr[r] = 4 * g + r < K ? qb[4 * g + r] * SA_LOG2E : -1e30f;
    f32x4_t acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float cs[4] = {0.f, 0.f, 0.f, 0.f};                  // sum over this lane's positions of w[slot 4g + r]
    const int ntile = tile1;
    float4 bufA[4], bufB[4];
    if (tile0 + wv < ntile) sa_load_tile(xb, N, tile0 + wv, li, g, bufA);
    if (tile0 + wv + nw < ntile) sa_load_tile(xb, N, tile0 + wv + nw, li, g, bufB);
    auto tile_step = [&](float4 (&cur)[4], int t) {
        // ---- LayerNorm statistics of position li (its 64 channels sit in the four lanes li, li+16, li+32, li+48)
        f32x2_t v2[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) { v2[2 * c] = (f32x2_t){cur[c].x, cur[c].y}; v2[2 * c + 1] = (f32x2_t){cur[c].z, cur[c].w}; }
        if (t + 2 * nw < ntile) sa_load_tile(xb, N, t + 2 * nw, li, g, cur);
        f32x2_t s2 = ((v2[0] + v2[1]) + (v2[2] + v2[3])) + ((v2[4] + v2[5]) + (v2[6] + v2[7]));
        const float mu = sa_xrow_sum4(s2.x + s2.y) * (1.0f / SA_C);
        const f32x2_t mu2 = (f32x2_t){mu, mu};
#pragma unroll
        for (int k = 0; k < 8; ++k) v2[k] -= mu2;
        f32x2_t q2 = v2[0] * v2[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) q2 = __builtin_elementwise_fma(v2[k], v2[k], q2);
        const float rs = __builtin_amdgcn_rsqf(sa_xrow_sum4(q2.x + q2.y) * (1.0f / SA_C) + 1e-5f);
        // ---- logits^T = q' d^T in two independent chains (the centred x is the B operand: [k = channel group][n = position])
        f32x4_t La = (f32x4_t){0.f, 0.f, 0.f, 0.f}, Lb = La;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            La = MFMA16(qpr[2 * k], v2[k].x, La);
            Lb = MFMA16(qpr[2 * k + 1], v2[k].y, Lb);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            *reinterpret_cast<float4*>(tile + li * SA_TLD + 16 * c + 4 * g) = make_float4(v2[2 * c].x, v2[2 * c].y, v2[2 * c + 1].x, v2[2 * c + 1].y);
        // ---- soft-max over the slots of position li: registers, then across the slot groups
        float l[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) l[r] = __builtin_fmaf(La[r] + Lb[r], rs, qbr[r]);
        float mx = fmaxf(fmaxf(l[0], l[1]), fmaxf(l[2], l[3]));
        mx = sa_xrow16(mx, true);
        if (K > 8) mx = sa_xrow32(mx, true);
        float e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(l[r] - mx);
        float sm = (e[0] + e[1]) + (e[2] + e[3]);
        sm = sa_xrow16(sm, false);
        if (K > 8) sm = sa_xrow32(sm, false);
        const float inv = __builtin_amdgcn_rcpf(sm);
        const int pos = t * 16 + li;
        const float live = pos < N ? 1.f : 0.f;              // only the ragged last tile has dead positions
        float w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            w[r] = __builtin_fmaf(e[r], inv, eps) * live;
            cs[r] += w[r];
            wt[(4 * g + r) * SA_WT_LD + li] = w[r] * rs;      // [slot][position]
        }
        if (attn_out && pos < N) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (4 * g + r < K) attn_out[pos * K + 4 * g + r] = e[r] * inv;
        }
        __builtin_amdgcn_wave_barrier();
        const float4 wb4 = *reinterpret_cast<const float4*>(wt + li * SA_WT_LD + 4 * g);      // w rstd of slot li at positions 4g .. 4g+3
        const float wb[4] = {wb4.x, wb4.y, wb4.z, wb4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = MFMA16(tile[(4 * g + r) * SA_TLD + 16 * m + li], wb[r], acc[m]);   // sum_n (w rstd) d^T: [ch][slot]
        __builtin_amdgcn_wave_barrier();
    };
#pragma unroll 1
    for (int t = tile0 + wv; t < ntile; t += 2 * nw) {
        tile_step(bufA, t);
        if (t + nw < ntile) tile_step(bufB, t + nw);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) cs[r] = red16_sum(cs[r]);      // over the 16 positions of the row: every lane of row g holds slot 4g + r
    __syncthreads();          // every wave is done with its tile: the region is reused for the partials
    if (li < K) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) scr[(wv * K + li) * (SA_C + 1) + 16 * m + 4 * g + r] = acc[m][r];
    }
    if (li == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * g + r < K) scr[(wv * K + 4 * g + r) * (SA_C + 1) + SA_C] = cs[r];
    }
}

// ------------------------------------------------------------------------------------------- slot side: geometry
// The slot-side kernels run their matrix products on 16-row MFMA tiles.  With K <= 8 slots per image a workgroup therefore takes
// G = 16 / K images at once: the G*K slot rows fill the tile, so every weight element fetched from L2 and every MFMA issued serves G
// images instead of one (the slot side is bound by exactly those two: 1.4 MB of weights per update through one CU's 64 B/clk fill
// path and ~5.5k MFMAs on its four matrix pipes).  For K > 8 a workgroup takes one image and walks its slots in two row blocks so
// the per-row temporaries fit the LDS.  Row r of a workgroup = (image r / K of the group, slot r % K).
template <int K> struct SaBlk { static constexpr int NB = K > 8 ? 2 : 1, KB = (K + NB - 1) / NB, KP = NB * KB; };   // per image (streaming kernels, exchange layout)
template <int K, int G> struct SaGeo {
    static_assert(G >= 1 && (K <= 8 ? G * K <= 16 : G == 1), "slot rows of a group must fit one 16-row tile");
    static constexpr int NB = K > 8 ? 2 : 1;                    // row blocks processed one after the other
    static constexpr int KB = K > 8 ? (K + 1) / 2 : G * K;      // rows per block
    static constexpr int KP = NB * KB;                          // rows held in LDS
    static constexpr int KI = SaBlk<K>::KP;                     // rows per image in the exchange buffers
};
// maps a workgroup row to its row of a [B*I*K, ld] row matrix (saved activations, gradient rows) at one iteration
struct SaRowMap {
    float* base;           // row (first image of the group, iteration t, slot 0); null = nothing to save
    size_t img;            // distance between the row blocks of consecutive images: I*K*ld
    int K, ld;
    __device__ float* operator()(int r) const { return base + (size_t)(r / K) * img + (size_t)(r % K) * ld; }
};
__device__ inline SaRowMap sa_rows(float* mat, int ld, int b0, int t, int I, int K) {
    SaRowMap m;
    m.base = mat ? mat + ((size_t)b0 * I + t) * K * ld : nullptr; m.img = (size_t)I * K * ld; m.K = K; m.ld = ld;
    return m;
}
// LDS rows [nr][W] <-> rows r0 .. r0+nr of the row matrix, columns off .. off+W
__device__ inline void rows_put(const float* lds, const SaRowMap& m, int off, int r0, int nr, int W) {
    for (int i = threadIdx.x; i < nr * W; i += blockDim.x) { const int j = i / W, c = i - j * W; m(r0 + j)[off + c] = lds[i]; }
}
__device__ inline void rows_get(float* lds, const SaRowMap& m, int off, int r0, int nr, int W) {
    for (int i = threadIdx.x; i < nr * W; i += blockDim.x) { const int j = i / W, c = i - j * W; lds[i] = m(r0 + j)[off + c]; }
}

// LDS map of the forward slot-side kernel
template <int K, int G>
struct SaFwdLds {
    float *s, *sn, *q, *u, *gi, *gh, *hid, *qp, *up, *qg, *cs, *qb;
    // NH heads: the C-wide per-row fields exist per (head, row); virtual row v = h * KP + r, NH * KP <= 16
    __device__ SaFwdLds(float* sm, int D, int H, int NH) {
        constexpr int C = SA_C, KB = SaGeo<K, G>::KB;
        const int KP = SaGeo<K, G>::KP, KV = NH * KP;
        s = sm;                       // [KP][D] slots (rows beyond the valid ones are padding)
        sn = s + KP * D;              // [KB][D]
        q = sn + KB * D;              // [KB][D]
        u = q + KB * D;               // [KB][D]
        gi = u + KB * D;              // [KB][3D]
        gh = gi + KB * 3 * D;         // [KB][3D]
        hid = gh + KB * 3 * D;        // [KB][H]
        qp = hid + KB * H;            // [KV][C]
        up = qp + KV * C;             // [KV][C]
        qg = up + KV * C;             // [KV][C] gamma_in * q'
        cs = qg + KV * C;             // [16] weight sums
        qb = cs + 16;                 // [16] beta_in . q'
    }
};
// rows per image of the exchange buffers as the streaming kernels see them (SaBlk<KS>::KP for KS = heads * slots columns)
__host__ __device__ inline int sa_ki(int KS) { return KS > 8 ? 2 * ((KS + 1) / 2) : KS; }

// slot side before the streaming pass of iteration t: LN(slots), q, q' = scale q Wk, and q' folded with the norm_inputs affine
template <int K, int G>
__device__ __forceinline__ void sa_phase_a(const SaFwdLds<K, G>& L, const SlotAttnArgs& p, const SaWts& wo, const SaSave& so, const SaRowMap& sv, int nvalid) {
    constexpr int C = SA_C, NB = SaGeo<K, G>::NB, KB = SaGeo<K, G>::KB, KP = SaGeo<K, G>::KP;
    const int D = p.D, NH = p.NH, dh = D / NH;
    const float* W = p.wts;
#pragma unroll 1
    for (int hb = 0; hb < NB; ++hb) {
        const int j0 = hb * KB, kv = (nvalid - j0) < KB ? (nvalid - j0 > 0 ? nvalid - j0 : 0) : KB;
        float* sB = L.s + j0 * D;
        if (sv.base) rows_put(sB, sv, so.sprev, j0, kv, D);
        ln_rows(sB, L.sn, W + wo.ln_s_g, W + wo.ln_s_b, KB, D);
        __syncthreads();
        matvec<KB>(W + wo.Wq, D / 16, 0, 0, D, D, L.sn, D, L.q, D, nullptr, 1.f);
        // q'_h = scale q[:, head h] Wk[head h, :]  (one head: the whole row)
#pragma unroll 1
        for (int h = 0; h < NH; ++h) matvec<KB>(W + wo.WkT, D / 16, 0, h * dh / 16, dh, C, L.q + h * dh, D, L.qp + (h * KP + j0) * C, C, nullptr, p.scale);
        if (sv.base) {
            rows_put(L.sn, sv, so.sn, j0, kv, D);
            rows_put(L.q, sv, so.q, j0, kv, D);
            for (int h = 0; h < NH; ++h) rows_put(L.qp + (h * KP + j0) * C, sv, so.qp + h * C, j0, kv, C);
        }
        __syncthreads();
    }
    sa_fold_affine(L.qp, W + wo.ln_in_g, W + wo.ln_in_b, L.qg, L.qb, NH * KP);
    __syncthreads();
}

// fixed-order sum over the nparts partials of one image, eight loads in flight per thread (nparts reaches 128 for a single image)
__device__ inline float sa_psum(const float* __restrict__ part, int nparts, int stride, int off) {
    float a = 0.f;
    int w = 0;
    for (; w + 8 <= nparts; w += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(w + u) * stride + off];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    for (; w < nparts; ++w) a += part[(size_t)w * stride + off];
    return a;
}

// weighted means from the partial sums of the streaming launch: parts[b][w][j][0..63] = sum w xn, [..][64] = sum w
template <int K, int G>
__device__ __forceinline__ void sa_reduce_parts(const SaFwdLds<K, G>& L, const SlotAttnArgs& p, const SaWts& wo, const SaSave& so, int b0, int NS,
                                                const SaRowMap& sv, int nvalid) {
    constexpr int C = SA_C, KP = SaGeo<K, G>::KP;
    const int tid = threadIdx.x, nt = blockDim.x, NH = p.NH, KV = NH * KP;
    const float* W = p.wts;
    const int pstride = NH * K * (C + 1);          // an image's partial: [NH*K columns][65], column = head * K + slot
    if (tid < KV) {
        const int h = tid / KP, r = tid - h * KP;
        if (r < nvalid) L.cs[tid] = sa_psum(p.parts + (size_t)(b0 + r / K) * NS * pstride, NS, pstride, (h * K + r % K) * (C + 1) + C);
    }
    __syncthreads();
    for (int i = tid; i < KV * C; i += nt) {
        const int v_ = i >> 6, c = i & 63, h = v_ / KP, r = v_ - h * KP;
        if (r >= nvalid) { L.up[i] = 0.f; continue; }
        float a = sa_psum(p.parts + (size_t)(b0 + r / K) * NS * pstride, NS, pstride, (h * K + r % K) * (C + 1) + c);
        a /= L.cs[v_];                                                 // sum_n w xn / sum_n w
        const float v = a * W[wo.ln_in_g + c] + W[wo.ln_in_b + c];
        L.up[i] = v;
        if (sv.base) { float* row = sv(r); row[so.upn + h * C + c] = a; row[so.up + h * C + c] = v; }
    }
    if (sv.base && tid < KV) {
        const int h = tid / KP, r = tid - h * KP;
        if (r < nvalid) sv(r)[so.csum + h] = L.cs[tid];
    }
    __syncthreads();
}

// slot side after the streaming pass: updates = U' Wv^T ; GRU ; residual MLP
template <int K, int G>
__device__ __forceinline__ void sa_phase_u(const SaFwdLds<K, G>& L, const SlotAttnArgs& p, const SaWts& wo, const SaSave& so, const SaRowMap& sv, int nvalid) {
    constexpr int C = SA_C, NB = SaGeo<K, G>::NB, KB = SaGeo<K, G>::KB, KP = SaGeo<K, G>::KP;
    const int D = p.D, H = p.H, tid = threadIdx.x, nt = blockDim.x, NH = p.NH, dh = D / NH;
    const float* W = p.wts;
    float *q = L.q, *u = L.u, *gi = L.gi, *gh = L.gh, *hid = L.hid, *sn = L.sn;
#pragma unroll 1
    for (int hb = 0; hb < NB; ++hb) {
        const int j0 = hb * KB, kv = (nvalid - j0) < KB ? (nvalid - j0 > 0 ? nvalid - j0 : 0) : KB;
        float* sB = L.s + j0 * D;
        // updates[:, head h] = U'_h Wv[head h, :]^T
#pragma unroll 1
        for (int h = 0; h < NH; ++h) matvec<KB>(W + wo.Wv, C / 16, h * dh / 16, 0, C, dh, L.up + (h * KP + j0) * C, C, u + h * dh, D, nullptr, 1.f);
        {
            const MvJob ji = {W + wo.Wih, D / 16, 0, 0, D, 3 * D, u, D, gi, 3 * D, W + wo.bih, 1.f}, jh = {W + wo.Whh, D / 16, 0, 0, D, 3 * D, sB, D, gh, 3 * D, W + wo.bhh, 1.f};
            matvec2<KB>(ji, jh);
        }
        for (int i = tid; i < KB * D; i += nt) {
            const int j = i / D, c = i - j * D;
            const float r = sigmoidf_(gi[j * 3 * D + c] + gh[j * 3 * D + c]);
            const float z = sigmoidf_(gi[j * 3 * D + D + c] + gh[j * 3 * D + D + c]);
            const float hn = gh[j * 3 * D + 2 * D + c];
            const float nn = tanhf(gi[j * 3 * D + 2 * D + c] + r * hn);
            const float sg = (1.f - z) * nn + z * sB[i];
            if (sv.base && j < kv) {
                float* row = sv(j0 + j) + c;
                row[so.u] = u[i]; row[so.r] = r; row[so.z] = z; row[so.n] = nn; row[so.hn] = hn; row[so.sg] = sg;
            }
            q[i] = sg;      // q is free now: holds s_gru
        }
        __syncthreads();
        ln_rows(q, sn, W + wo.ln_m_g, W + wo.ln_m_b, KB, D);      // sn = m
        __syncthreads();
        matvec<KB>(W + wo.W0, D / 16, 0, 0, D, H, sn, D, hid, H, W + wo.b0, 1.f);
        for (int i = tid; i < KB * H; i += nt) hid[i] = fmaxf(hid[i], 0.f);
        __syncthreads();
        matvec<KB>(W + wo.W2, H / 16, 0, 0, H, D, hid, H, u, D, W + wo.b2, 1.f);      // u = mlp out
        for (int i = tid; i < kv * D; i += nt) sB[i] = q[i] + u[i];
        if (sv.base) {
            rows_put(sn, sv, so.m, j0, kv, D);
            rows_put(hid, sv, so.hid, j0, kv, H);
        }
        __syncthreads();
    }
}

// The forward is a pipeline of two kinds of launches:
//   sa_stream_fwd_kernel  NS workgroups of 4 waves per image, each streams 1/NS of the positions against the image's folded query and
//                         writes one partial [K][65]; B*NS is sized to fill every CU several workgroups deep at any batch size
//   sa_slot_fwd_kernel    one workgroup per group of G images: sums the partials in a fixed order, runs GRU + MLP, prepares q' of the
//                         next iteration
// Between launches an image's slots and folded query live in `xchg` (a few KB per image).
//   xchg per image: slots [KI*D] | qg [KI*64] | qb [16]            parts: [B][NS][K][65]
__host__ __device__ inline size_t sa_xchg_floats(int K, int D) {
    const int KI = K > 8 ? 2 * ((K + 1) / 2) : K;
    return ((size_t)KI * D + (size_t)KI * SA_C + 16 + 15) & ~(size_t)15;
}
#define SA_TS 256        // streaming workgroup: 4 waves
template <int K, bool V2>
__global__ __launch_bounds__(SA_TS, V2 ? 4 : 3) void sa_stream_fwd_kernel(SlotAttnArgs p, int t, int NS) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int C = SA_C, KP = SaBlk<K>::KP;
    const int D = p.D, N = p.N, nt = blockDim.x, nw = nt >> 6, tid = threadIdx.x;
    float* qg = sm;                 // [K][C]
    float* qb = qg + KP * C;        // [16]
    float* tiles = qb + 16;         // [waves][16][SA_TLD]
    const int b = blockIdx.x / NS, h = blockIdx.x % NS;
    const float* xg = p.xchg + (size_t)b * sa_xchg_floats(K, D);
    for (int i = tid; i < K * C; i += nt) qg[i] = xg[KP * D + i];
    if (tid < K) qb[tid] = xg[KP * D + KP * C + tid];
    __syncthreads();
    const float4* xb = reinterpret_cast<const float4*>(p.x + (size_t)b * N * C);
    const int ntile = (N + 15) / 16, per = (ntile + NS - 1) / NS;
    const int t0 = h * per, t1 = (t0 + per < ntile) ? t0 + per : ntile;
    float* attn_out = (t == p.I - 1 && p.attn) ? p.attn + (size_t)b * N * K : nullptr;
    if (V2) sa_stream_fwd2<K>(xb, N, qg, qb, p.eps, attn_out, tiles, tiles, t0, t1 > t0 ? t1 : t0);
    else sa_stream_fwd<K>(xb, N, qg, qb, p.eps, attn_out, tiles, tiles, t0, t1 > t0 ? t1 : t0);
    __syncthreads();
    float* part = p.parts + ((size_t)b * NS + h) * K * (C + 1);
    for (int i = tid; i < K * (C + 1); i += nt) {          // this workgroup's partial: sum over its waves, fixed order
        float a = 0.f;
        for (int w = 0; w < nw; ++w) a += tiles[w * K * (C + 1) + i];
        part[i] = a;
    }
}
// t = -1: preparation (q' of iteration 0 from slots0); t >= 0: slot update of iteration t (+ q' of iteration t + 1)
template <int K, int G>
__global__ __launch_bounds__(SA_TF) void sa_slot_fwd_kernel(SlotAttnArgs p, SaWts wo, SaSave so, int t, int NS) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int C = SA_C, KP = SaGeo<K, G>::KP;
    const int D = p.D, nt = blockDim.x, tid = threadIdx.x, NH = p.NH, KI = sa_ki(NH * K);
    const SaFwdLds<K, G> L(sm, D, p.H, NH);
    const int b0 = blockIdx.x * G;
    const int nimg = (p.B - b0) < G ? (p.B - b0) : G, nvalid = nimg * K;
    const size_t XF = sa_xchg_floats(NH * K, D);
    float* xg0 = p.xchg + (size_t)b0 * XF;
    float* save = p.save;
    if (t < 0) {
        for (int i = tid; i < KP * D; i += nt) L.s[i] = i < nvalid * D ? p.slots0[(size_t)b0 * K * D + i] : 0.f;
        __syncthreads();
        sa_phase_a<K, G>(L, p, wo, so, sa_rows(save, so.ld, b0, 0, p.I, K), nvalid);
    } else {
        const SaRowMap sv = sa_rows(save, so.ld, b0, t, p.I, K);
        for (int i = tid; i < KP * D; i += nt) {
            const int r = i / D, c = i - r * D;
            L.s[i] = r < nvalid ? xg0[(size_t)(r / K) * XF + (r % K) * D + c] : 0.f;
        }
        __syncthreads();
        sa_reduce_parts<K, G>(L, p, wo, so, b0, NS, sv, nvalid);
        sa_phase_u<K, G>(L, p, wo, so, sv, nvalid);
        if (t == p.I - 1) {
            for (int i = tid; i < nvalid * D; i += nt) p.slots[(size_t)b0 * K * D + i] = L.s[i];
            return;
        }
        sa_phase_a<K, G>(L, p, wo, so, sa_rows(save, so.ld, b0, t + 1, p.I, K), nvalid);
    }
    for (int i = tid; i < nvalid * D; i += nt) { const int r = i / D, c = i - r * D; xg0[(size_t)(r / K) * XF + (r % K) * D + c] = L.s[i]; }
    for (int i = tid; i < NH * KP * C; i += nt) {
        const int v = i >> 6, c = i & 63, h = v / KP, r = v - h * KP;
        if (r < nvalid) xg0[(size_t)(r / K) * XF + KI * D + (h * K + r % K) * C + c] = L.qg[i];
    }
    if (tid < NH * KP) {
        const int h = tid / KP, r = tid - h * KP;
        if (r < nvalid) xg0[(size_t)(r / K) * XF + KI * D + KI * C + h * K + r % K] = L.qb[tid];
    }
}

// ------------------------------------------------------------------------------------------- backward streaming pass
// recomputes attn from xn and the folded operands (qg = gamma q', qb = beta.q', dug = gamma dU', dub = beta.dU');
// partials scr[wave][j][0..63] = sum_n dlogits[n,j] xn[n], scr[wave][j][64] = sum_n dlogits[n,j] (scr aliases the
// tiles: barrier inside); d xn written (FIRST), accumulated, or (FINAL) pushed through the LayerNorm backward into dx.
template <int K, bool FIRST, bool FINAL>
__device__ __forceinline__ void sa_stream_bwd(const float4* __restrict__ xb, float4* __restrict__ dxb, int N, const float* qg, const float* qb,
                                              const float* dug, const float* dub, const float* cs, const float* ud, float eps, float* tiles,
                                              float* scr, int tile0 = 0, int tile1 = -1) {
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
    const int ntile = tile1 < 0 ? (N + 15) / 16 : tile1;      // this workgroup's tiles: [tile0, ntile)
    float4 cur[4];
    if (tile0 + wv < ntile) sa_load_tile(xb, N, tile0 + wv, li, g, cur);
#pragma unroll 1
    for (int t = tile0 + wv; t < ntile; t += nw) {
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

// ------------------------------------------------------------------------------------------- backward streaming pass, second form
// The transposed form of sa_stream_fwd2 applied to the backward: logits^T and (d attn)^T arrive with the position on the lane and four
// slots in the registers, so the soft-max recomputation, its backward and the weights are in-lane arithmetic plus cross-row exchanges;
// d xn^T[ch][pos] = dug^T wn^T + qg^T dl^T contracts over the slots with wn / dl straight from those registers as B operands and lands
// in the load layout of x (no LDS transposition of the 16 x 64 result); only sum_n dl xn^T needs dl through the 16 x 16 LDS patch.
template <int K, bool FIRST, bool FINAL>
__device__ __forceinline__ void sa_stream_bwd2(const float4* __restrict__ xb, float4* __restrict__ dxb, int N, const float* qg, const float* qb,
                                               const float* dug, const float* dub, const float* cs, const float* ud, float eps, float* tiles,
                                               float* scr, int tile0, int tile1) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nw = blockDim.x >> 6, li = lane & 15, g = lane >> 4;
    float* tile = tiles + wv * (16 * SA_TLD + 2 * 16 * SA_WLD);
    float* wt = tile + 16 * SA_TLD;                      // [16 slots][SA_WT_LD] (the region of the first form's two [16][17] patches)
    float qpr[16], dur[16];                              // A operands of the two [slot][pos] products: lane (slot li, channel group g)
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int ch = 16 * (k >> 2) + 4 * g + (k & 3);
        qpr[k] = li < K ? qg[li * SA_C + ch] * SA_LOG2E : 0.f;
        dur[k] = li < K ? dug[li * SA_C + ch] : 0.f;
    }
    float duA[4][4], qgA[4][4];                          // A operands of d xn^T: [channel 16 m + li][slot 4 g + r]
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int slot = 4 * g + r;
            duA[m][r] = slot < K ? dug[slot * SA_C + 16 * m + li] : 0.f;
            qgA[m][r] = slot < K ? qg[slot * SA_C + 16 * m + li] : 0.f;
        }
    float qbr[4], dbr[4], icr[4];                        // per-slot constants of this lane's four slots
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int slot = 4 * g + r;
        qbr[r] = slot < K ? qb[slot] * SA_LOG2E : -1e30f;
        icr[r] = slot < K ? 1.0f / cs[slot] : 0.f;
        dbr[r] = slot < K ? dub[slot] - ud[slot] : 0.f;
    }
    f32x4_t acc[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float sdl[4] = {0.f, 0.f, 0.f, 0.f};
    const int ntile = tile1;
    float4 cur[4];
    if (tile0 + wv < ntile) sa_load_tile(xb, N, tile0 + wv, li, g, cur);
#pragma unroll 1
    for (int t = tile0 + wv; t < ntile; t += nw) {
        float4 nxt[4];
        if (t + nw < ntile) sa_load_tile(xb, N, t + nw, li, g, nxt);
        const int pos = t * 16 + li;
        float4 od[4];
        if (!FIRST) {
#pragma unroll
            for (int c = 0; c < 4; ++c) od[c] = pos < N ? dxb[pos * 16 + 4 * c + g] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        // ---- LayerNorm statistics, centred x
        f32x2_t v2[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) { v2[2 * c] = (f32x2_t){cur[c].x, cur[c].y}; v2[2 * c + 1] = (f32x2_t){cur[c].z, cur[c].w}; }
        f32x2_t s2 = ((v2[0] + v2[1]) + (v2[2] + v2[3])) + ((v2[4] + v2[5]) + (v2[6] + v2[7]));
        const float mu = sa_xrow_sum4(s2.x + s2.y) * (1.0f / SA_C);
        const f32x2_t mu2 = (f32x2_t){mu, mu};
#pragma unroll
        for (int k = 0; k < 8; ++k) v2[k] -= mu2;
        f32x2_t q2 = v2[0] * v2[0];
#pragma unroll
        for (int k = 1; k < 8; ++k) q2 = __builtin_elementwise_fma(v2[k], v2[k], q2);
        const float rs = __builtin_amdgcn_rsqf(sa_xrow_sum4(q2.x + q2.y) * (1.0f / SA_C) + 1e-5f);
        // ---- logits^T and (d attn)^T: four independent chains
        f32x4_t La = (f32x4_t){0.f, 0.f, 0.f, 0.f}, Lb = La, Da = La, Db = La;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            La = MFMA16(qpr[2 * k], v2[k].x, La);
            Da = MFMA16(dur[2 * k], v2[k].x, Da);
            Lb = MFMA16(qpr[2 * k + 1], v2[k].y, Lb);
            Db = MFMA16(dur[2 * k + 1], v2[k].y, Db);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
            *reinterpret_cast<float4*>(tile + li * SA_TLD + 16 * c + 4 * g) = make_float4(v2[2 * c].x, v2[2 * c].y, v2[2 * c + 1].x, v2[2 * c + 1].y);
        // ---- soft-max of position li, its backward, the normalised weights
        float l[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) l[r] = __builtin_fmaf(La[r] + Lb[r], rs, qbr[r]);
        float mx = fmaxf(fmaxf(l[0], l[1]), fmaxf(l[2], l[3]));
        mx = sa_xrow16(mx, true);
        if (K > 8) mx = sa_xrow32(mx, true);
        float a[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) a[r] = __builtin_amdgcn_exp2f(l[r] - mx);
        float sm = (a[0] + a[1]) + (a[2] + a[3]);
        sm = sa_xrow16(sm, false);
        if (K > 8) sm = sa_xrow32(sm, false);
        const float inv = __builtin_amdgcn_rcpf(sm);
        float da[4], dot = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            a[r] *= inv;
            da[r] = (__builtin_fmaf(Da[r] + Db[r], rs, dbr[r])) * icr[r];          // d attn[pos][slot] (0 for padded slots: icr = 0)
            dot = __builtin_fmaf(a[r], da[r], dot);
        }
        dot = sa_xrow16(dot, false);
        if (K > 8) dot = sa_xrow32(dot, false);
        const float live = pos < N ? 1.f : 0.f;
        float dl[4], wn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            dl[r] = a[r] * (da[r] - dot) * live;                 // d logits
            wn[r] = (a[r] + eps) * icr[r] * live;                // normalised weight
            sdl[r] += dl[r];
            wt[(4 * g + r) * SA_WT_LD + li] = dl[r] * rs;
        }
        __builtin_amdgcn_wave_barrier();
        const float4 wb4 = *reinterpret_cast<const float4*>(wt + li * SA_WT_LD + 4 * g);      // dl rstd of slot li at positions 4g .. 4g+3
        const float wb[4] = {wb4.x, wb4.y, wb4.z, wb4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < 4; ++m) acc[m] = MFMA16(tile[(4 * g + r) * SA_TLD + 16 * m + li], wb[r], acc[m]);   // sum_n dl xn^T: [ch][slot]
        // ---- d xn^T[ch][pos] = sum_slot dug[slot][ch] wn[pos][slot] + qg[slot][ch] dl[pos][slot]
        f32x4_t Dx[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) Dx[m] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                Dx[m] = MFMA16(duA[m][r], wn[r], Dx[m]);
                Dx[m] = MFMA16(qgA[m][r], dl[r], Dx[m]);
            }
        __builtin_amdgcn_wave_barrier();          // the tile and the patch may be overwritten by the next step
        float dv[16];
#pragma unroll
        for (int m = 0; m < 4; ++m) { dv[4 * m] = Dx[m][0]; dv[4 * m + 1] = Dx[m][1]; dv[4 * m + 2] = Dx[m][2]; dv[4 * m + 3] = Dx[m][3]; }
        if (!FIRST) {
#pragma unroll
            for (int c = 0; c < 4; ++c) { dv[4 * c] += od[c].x; dv[4 * c + 1] += od[c].y; dv[4 * c + 2] += od[c].z; dv[4 * c + 3] += od[c].w; }
        }
        if (FINAL) {      // LayerNorm(norm_inputs) backward from d xn, in the load layout (xn = centred x * rstd)
            float m1 = 0.f, m2 = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                m1 += dv[2 * k] + dv[2 * k + 1];
                m2 += dv[2 * k] * v2[k].x + dv[2 * k + 1] * v2[k].y;
            }
            m1 = sa_xrow_sum4(m1) * (1.0f / SA_C);
            m2 = sa_xrow_sum4(m2) * rs * (1.0f / SA_C);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                dv[2 * k] = rs * (dv[2 * k] - m1 - v2[k].x * rs * m2);
                dv[2 * k + 1] = rs * (dv[2 * k + 1] - m1 - v2[k].y * rs * m2);
            }
        }
        if (pos < N) {
#pragma unroll
            for (int c = 0; c < 4; ++c) dxb[pos * 16 + 4 * c + g] = make_float4(dv[4 * c], dv[4 * c + 1], dv[4 * c + 2], dv[4 * c + 3]);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) cur[c] = nxt[c];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) sdl[r] = red16_sum(sdl[r]);
    __syncthreads();
    if (li < K) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) scr[(wv * K + li) * (SA_C + 1) + 16 * m + 4 * g + r] = acc[m][r];
    }
    if (li == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (4 * g + r < K) scr[(wv * K + 4 * g + r) * (SA_C + 1) + SA_C] = sdl[r];
    }
}

// LDS map of the backward slot-side kernel
template <int K, int G>
struct SaBwdLds {
    float *ds, *t2, *t0, *t1, *dgi, *dgh, *dhid, *qp, *dup, *dqp, *qg, *dug, *cs, *ud, *qb, *dub, *sdl, *gacc;
    __device__ SaBwdLds(float* sm, int D, int H, int NH) {
        constexpr int C = SA_C, KB = SaGeo<K, G>::KB, KP = SaGeo<K, G>::KP;
        const int KV = NH * KP;          // virtual rows (head, row) of the C-wide fields
        ds = sm;                      // [KP][D] gradient wrt the iteration output
        t2 = ds + KP * D;             // [KP][D] dh (direct GRU path), kept across the streaming pass
        t0 = t2 + KP * D;             // [KB][D] scratch
        t1 = t0 + KB * D;             // [KB][D] scratch
        dgi = t1 + KB * D;            // [KB][3D]
        dgh = dgi + KB * 3 * D;       // [KB][3D]
        dhid = dgh + KB * 3 * D;      // [KB][H]
        qp = dhid + KB * H;           // [KV][C]
        dup = qp + KV * C;            // [KV][C]
        dqp = dup + KV * C;           // [KV][C]
        qg = dqp + KV * C;            // [KV][C] gamma_in * q'
        dug = qg + KV * C;            // [KV][C] gamma_in * dU'
        cs = dug + KV * C;            // [16] csum
        ud = cs + 16;                 // [16] up . dup
        qb = ud + 16;                 // [16] beta_in . q'
        dub = qb + 16;                // [16] beta_in . dU'
        sdl = dub + 16;               // [16] sum_n dlogits
        gacc = sdl + 16;              // dgamma/dbeta accumulators of the group: ln_s (2D), ln_m (2D), ln_in (2C)
    }
};

// slot side of iteration t before the streaming pass: residual MLP, LN_m, GRU backward -> d updates -> dU' = du Wv; then the folded
// operands of the streaming pass (qg, qb, dug, dub, cs, ud).  In: L.ds = gradient wrt the iteration's output.  Out: L.t2 = dh.
template <int K, int G>
__device__ __forceinline__ void sa_bwd_part1(const SaBwdLds<K, G>& L, const SlotAttnArgs& p, const SaWts& wo, const SaSave& so, const SaGrad& go,
                                             const SaRowMap& sv, const SaRowMap& gr, int nvalid) {
    constexpr int C = SA_C, NB = SaGeo<K, G>::NB, KB = SaGeo<K, G>::KB, KP = SaGeo<K, G>::KP;
    const int D = p.D, H = p.H, tid = threadIdx.x, nt = blockDim.x, NH = p.NH, dh = D / NH, KV = NH * KP;
    const float* W = p.wts;
    float* dg_m = L.gacc + 2 * D;  float* db_m = L.gacc + 3 * D;
#pragma unroll 1
    for (int hb = 0; hb < NB; ++hb) {
        const int j0 = hb * KB, kv = (nvalid - j0) < KB ? (nvalid - j0 > 0 ? nvalid - j0 : 0) : KB;
        float* dsB = L.ds + j0 * D;
        float* t2B = L.t2 + j0 * D;
        float *t0 = L.t0, *t1 = L.t1, *dgi = L.dgi, *dgh = L.dgh, *dhid = L.dhid;
        // ---- residual MLP backward: s_new = sg + W2 relu(W0 m + b0) + b2,  m = LN_m(sg)
        rows_put(dsB, gr, go.out, j0, kv, D);
        rows_get(t0, sv, so.sg, j0, kv, D);
        for (int i = kv * D + tid; i < KB * D; i += nt) t0[i] = 0.f;                // padding rows: defined values for the LayerNorm statistics
        matvec<KB>(W + wo.W2T, D / 16, 0, 0, D, H, dsB, D, dhid, H, nullptr, 1.f);          // dhid[h] = sum_d ds[d] W2[d][h]
        for (int i = tid; i < kv * H; i += nt) {
            const int j = i / H, c = i - j * H;
            const float v = sv(j0 + j)[so.hid + c] > 0.f ? dhid[i] : 0.f;
            dhid[i] = v;
            gr(j0 + j)[go.hid + c] = v;
        }
        __syncthreads();
        matvec<KB>(W + wo.W0T, H / 16, 0, 0, H, D, dhid, H, t1, D, nullptr, 1.f);           // dm[e] = sum_h dhid[h] W0[h][e]
        ln_rows_bwd(t1, t0, dsB, 1, W + wo.ln_m_g, dg_m, db_m, kv, D);           // ds = d s_gru
        __syncthreads();
        // ---- GRU backward
        for (int i = tid; i < kv * D; i += nt) {
            const int j = i / D, c = i - j * D;
            const float* row = sv(j0 + j) + c;
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
        rows_put(dgi, gr, go.gi, j0, kv, 3 * D);
        rows_put(dgh, gr, go.gh, j0, kv, 3 * D);
        {   // du[e] = sum_g dgi[g] Wih[g][e]  and  dh via the recurrent weights, one pass
            const MvJob ji = {W + wo.WihT, 3 * D / 16, 0, 0, 3 * D, D, dgi, 3 * D, t1, D, nullptr, 1.f}, jh = {W + wo.WhhT, 3 * D / 16, 0, 0, 3 * D, D, dgh, 3 * D, t0, D, nullptr, 1.f};
            matvec2<KB>(ji, jh);
        }
        for (int i = tid; i < kv * D; i += nt) t2B[i] += t0[i];
        rows_put(t1, gr, go.u, j0, kv, D);
        // ---- u = up Wv^T
#pragma unroll 1
        for (int h = 0; h < NH; ++h)          // dU'_h[c] = sum_{d in head h} du[d] Wv[d][c]
            matvec<KB>(W + wo.WvT, D / 16, 0, h * dh / 16, dh, C, t1 + h * dh, D, L.dup + (h * KP + j0) * C, C, nullptr, 1.f);
    }
    for (int h = 0; h < NH; ++h) rows_get(L.qp + h * KP * C, sv, so.qp + h * C, 0, nvalid, C);
    for (int i = tid; i < KV * C; i += nt)
        if (((i >> 6) % KP) >= nvalid) { L.qp[i] = 0.f; L.dup[i] = 0.f; }
    __syncthreads();
    if (tid < KV) {
        const int h = tid / KP, r = tid - h * KP;
        if (r < nvalid) {
            const float* row = sv(r);
            L.cs[tid] = row[so.csum + h];
            float a = 0.f;
            for (int c = 0; c < C; ++c) a += row[so.up + h * C + c] * L.dup[tid * C + c];
            L.ud[tid] = a;
        }
    }
    __syncthreads();
    sa_fold_affine(L.qp, W + wo.ln_in_g, W + wo.ln_in_b, L.qg, L.qb, KV);
    sa_fold_affine(L.dup, W + wo.ln_in_g, W + wo.ln_in_b, L.dug, L.dub, KV);
    __syncthreads();
}

// slot side of iteration t after the streaming pass, from the partials parts[b][w][j][0..63] = sum_n dl xn, [..][64] = sum_n dl:
// norm_inputs gamma/beta, d q', d q, d LN_s input.  In: L.qp, L.dup, L.t2.  Out: L.ds = gradient wrt the output of iteration t - 1.
template <int K, int G>
__device__ __forceinline__ void sa_bwd_part2(const SaBwdLds<K, G>& L, const SlotAttnArgs& p, const SaWts& wo, const SaSave& so, const SaGrad& go,
                                             const SaRowMap& sv, const SaRowMap& gr, int b0, int NS, int nvalid) {
    constexpr int C = SA_C, NB = SaGeo<K, G>::NB, KB = SaGeo<K, G>::KB, KP = SaGeo<K, G>::KP;
    const int D = p.D, tid = threadIdx.x, nt = blockDim.x, NH = p.NH, dh = D / NH, KV = NH * KP;
    const float* W = p.wts;
    float* dg_s = L.gacc;  float* db_s = L.gacc + D;
    float* dg_in = L.gacc + 4 * D;  float* db_in = L.gacc + 4 * D + C;
    float *dqp = L.dqp, *sdl = L.sdl, *t0 = L.t0, *t1 = L.t1;
    const int pstride = NH * K * (C + 1);
    if (tid < KV) {
        const int h = tid / KP, r = tid - h * KP;
        sdl[tid] = r < nvalid ? sa_psum(p.parts + (size_t)(b0 + r / K) * NS * pstride, NS, pstride, (h * K + r % K) * (C + 1) + C) : 0.f;
    }
    for (int i = tid; i < KV * C; i += nt) {
        const int v = i >> 6, c = i & 63, h = v / KP, r = v - h * KP;
        dqp[i] = r < nvalid ? sa_psum(p.parts + (size_t)(b0 + r / K) * NS * pstride, NS, pstride, (h * K + r % K) * (C + 1) + c) : 0.f;   // sum_n dl xn (pre-affine)
    }
    __syncthreads();
    // norm_inputs gamma/beta:  d xa = wn dU' + dl q'  =>  dgamma = sum_j dU' upn + q' dqn,  dbeta = sum_j dU' + q' sdl
    if (tid < C) {
        float dg = 0.f, db = 0.f;
        for (int h = 0; h < NH; ++h)
            for (int r = 0; r < nvalid; ++r) {
                const int j = h * KP + r;
                dg += L.dup[j * C + tid] * sv(r)[so.upn + h * C + tid] + L.qp[j * C + tid] * dqp[j * C + tid];
                db += L.dup[j * C + tid] + L.qp[j * C + tid] * sdl[j];
            }
        dg_in[tid] += dg;
        db_in[tid] += db;
    }
    __syncthreads();
    for (int i = tid; i < KV * C; i += nt) {
        const int j = i >> 6, c = i & 63;
        if ((j % KP) < nvalid) dqp[i] = dqp[i] * W[wo.ln_in_g + c] + W[wo.ln_in_b + c] * sdl[j];   // d q' = sum_n dl LN(x)
    }
    __syncthreads();
    for (int h = 0; h < NH; ++h) rows_put(dqp + h * KP * C, gr, go.qp + h * C, 0, nvalid, C);
#pragma unroll 1
    for (int hb = 0; hb < NB; ++hb) {
        const int j0 = hb * KB, kv = (nvalid - j0) < KB ? (nvalid - j0 > 0 ? nvalid - j0 : 0) : KB;
        float* dsB = L.ds + j0 * D;
        // ---- q' = scale q Wk  ->  dq[d] = scale sum_c dqp[c] Wk[d][c] ;  q = sn Wq^T  ->  dsn[e] = sum_d dq[d] Wq[d][e]
#pragma unroll 1
        for (int h = 0; h < NH; ++h)          // dq[:, head h] = scale dq'_h Wk[head h, :]^T
            matvec<KB>(W + wo.Wk, C / 16, h * dh / 16, 0, C, dh, dqp + (h * KP + j0) * C, C, t1 + h * dh, D, nullptr, p.scale);   // t1 = dq
        rows_put(t1, gr, go.q, j0, kv, D);
        rows_get(t0, sv, so.sprev, j0, kv, D);
        for (int i = kv * D + tid; i < KB * D; i += nt) t0[i] = 0.f;
        matvec<KB>(W + wo.WqT, D / 16, 0, 0, D, D, t1, D, dsB, D, nullptr, 1.f);                 // ds = dsn
        for (int i = tid; i < kv * D; i += nt) { t1[i] = dsB[i]; dsB[i] = L.t2[j0 * D + i]; }
        __syncthreads();
        ln_rows_bwd(t1, t0, dsB, 1, W + wo.ln_s_g, dg_s, db_s, kv, D);                      // ds = dh + LN_s backward
        __syncthreads();
    }
}

// The backward mirrors the forward pipeline: sa_slot_bwd_kernel (one workgroup per group of G images) alternates with
// sa_stream_bwd_kernel (NS workgroups of 4 waves per image).  A slot launch runs the second half of iteration t_hi (from the partials
// of its streaming pass) and the first half of iteration t_lo = t_hi - 1; -1 marks the missing half of the first / last launch.
//   xchg per image (backward layout): t2 [KI*D] | dup [KI*C] | qg [KI*C] | dug [KI*C] | qb[16] | dub[16] | cs[16] | ud[16]
// The LayerNorm gamma / beta partials of a group are kept in the g_small row of its first image (the other rows of the group are zero).
__host__ __device__ inline size_t sa_xchg_bwd_floats(int K, int D) {
    const int KI = K > 8 ? 2 * ((K + 1) / 2) : K;
    return ((size_t)KI * D + 3 * (size_t)KI * SA_C + 64 + 15) & ~(size_t)15;
}
template <int K, int G>
__global__ __launch_bounds__(SA_TB) void sa_slot_bwd_kernel(SlotAttnArgs p, SaWts wo, SaSave so, SaGrad go, int t_hi, int t_lo, int NS) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int D = p.D;
    constexpr int C = SA_C, KP = SaGeo<K, G>::KP;
    const int nt = blockDim.x, tid = threadIdx.x, NH = p.NH, KV = NH * KP, KI = sa_ki(NH * K);
    const SaBwdLds<K, G> L(sm, D, p.H, NH);
    const int b0 = blockIdx.x * G;
    const int nimg = (p.B - b0) < G ? (p.B - b0) : G, nvalid = nimg * K;
    const size_t XF = sa_xchg_bwd_floats(NH * K, D);
    float* xg0 = p.xchg + (size_t)b0 * XF;
    const int SM = 4 * D + 2 * C;
    float* gs = p.g_small + (size_t)b0 * SM;
    if (t_hi < 0) {         // first launch: gradient wrt the final slots, zero accumulators
        for (int i = tid; i < KP * D; i += nt) L.ds[i] = i < nvalid * D ? p.dslots[(size_t)b0 * K * D + i] : 0.f;
        for (int i = tid; i < SM; i += nt) L.gacc[i] = 0.f;
    } else {
        for (int i = tid; i < KP * D; i += nt) { const int r = i / D, c = i - r * D; L.t2[i] = r < nvalid ? xg0[(size_t)(r / K) * XF + (r % K) * D + c] : 0.f; }
        for (int i = tid; i < KV * C; i += nt) {
            const int v = i >> 6, c = i & 63, h = v / KP, r = v - h * KP;
            L.dup[i] = r < nvalid ? xg0[(size_t)(r / K) * XF + KI * D + (h * K + r % K) * C + c] : 0.f;
        }
        for (int i = tid; i < SM; i += nt) L.gacc[i] = gs[i];
        for (int h = 0; h < NH; ++h) rows_get(L.qp + h * KP * C, sa_rows(p.save, so.ld, b0, t_hi, p.I, K), so.qp + h * C, 0, nvalid, C);
        for (int i = tid; i < KV * C; i += nt)
            if (((i >> 6) % KP) >= nvalid) L.qp[i] = 0.f;
    }
    __syncthreads();
    if (t_hi >= 0)
        sa_bwd_part2<K, G>(L, p, wo, so, go, sa_rows(p.save, so.ld, b0, t_hi, p.I, K), sa_rows(p.grows, go.ld, b0, t_hi, p.I, K), b0, NS, nvalid);
    if (t_lo >= 0) {
        sa_bwd_part1<K, G>(L, p, wo, so, go, sa_rows(p.save, so.ld, b0, t_lo, p.I, K), sa_rows(p.grows, go.ld, b0, t_lo, p.I, K), nvalid);
        for (int i = tid; i < nvalid * D; i += nt) { const int r = i / D, c = i - r * D; xg0[(size_t)(r / K) * XF + (r % K) * D + c] = L.t2[i]; }
        for (int i = tid; i < KV * C; i += nt) {
            const int v = i >> 6, c = i & 63, h = v / KP, r = v - h * KP;
            if (r >= nvalid) continue;
            float* x = xg0 + (size_t)(r / K) * XF + KI * D + (h * K + r % K) * C + c;
            x[0] = L.dup[i]; x[KI * C] = L.qg[i]; x[2 * KI * C] = L.dug[i];
        }
        if (tid < KV) {
            const int h = tid / KP, r = tid - h * KP;
            if (r < nvalid) {
                float* x = xg0 + (size_t)(r / K) * XF + KI * D + 3 * KI * C + h * K + r % K;
                x[0] = L.qb[tid]; x[16] = L.dub[tid]; x[32] = L.cs[tid]; x[48] = L.ud[tid];
            }
        }
    } else {
        for (int i = tid; i < nvalid * D; i += nt) p.dslots0[(size_t)b0 * K * D + i] = L.ds[i];
    }
    for (int i = tid; i < SM; i += nt) gs[i] = L.gacc[i];
    for (int i = tid; i < (nimg - 1) * SM; i += nt) gs[SM + i] = 0.f;
}
// 2 workgroups per CU = 256 VGPRs: at 3 (170 VGPRs) the FINAL variant spills 12 registers and the backward chain is 86 us slower
template <int K, bool FIRST, bool FINAL, bool V2>
__global__ __launch_bounds__(SA_TS, 2) void sa_stream_bwd_kernel(SlotAttnArgs p, int NS) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    constexpr int C = SA_C, KP = SaBlk<K>::KP;
    const int D = p.D, N = p.N, nt = blockDim.x, nw = nt >> 6, tid = threadIdx.x;
    float* qg = sm;                 // [KP][C]
    float* dug = qg + KP * C;       // [KP][C]
    float* sv4 = dug + KP * C;      // qb[16] | dub[16] | cs[16] | ud[16]
    float* tiles = sv4 + 64;        // [waves][16*SA_TLD + 2*16*SA_WLD]
    const int b = blockIdx.x / NS, h = blockIdx.x % NS;
    const float* xg = p.xchg + (size_t)b * sa_xchg_bwd_floats(K, D);
    const float *g_qg = xg + KP * D + KP * C, *g_dug = g_qg + KP * C, *g_small4 = g_dug + KP * C;
    for (int i = tid; i < K * C; i += nt) { qg[i] = g_qg[i]; dug[i] = g_dug[i]; }
    if (tid < 64) sv4[tid] = g_small4[tid];
    __syncthreads();
    const float4* xb = reinterpret_cast<const float4*>(p.x + (size_t)b * N * C);
    float4* dxb = reinterpret_cast<float4*>(p.dx + (size_t)b * N * C);
    const int ntile = (N + 15) / 16, per = (ntile + NS - 1) / NS;
    const int t0 = h * per, t1 = (t0 + per < ntile) ? t0 + per : ntile;
    if (V2) sa_stream_bwd2<K, FIRST, FINAL>(xb, dxb, N, qg, sv4, dug, sv4 + 16, sv4 + 32, sv4 + 48, p.eps, tiles, tiles, t0, t1 > t0 ? t1 : t0);
    else sa_stream_bwd<K, FIRST, FINAL>(xb, dxb, N, qg, sv4, dug, sv4 + 16, sv4 + 32, sv4 + 48, p.eps, tiles, tiles, t0, t1 > t0 ? t1 : t0);
    __syncthreads();
    float* part = p.parts + ((size_t)b * NS + h) * K * (C + 1);
    for (int i = tid; i < K * (C + 1); i += nt) {
        float a = 0.f;
        for (int w = 0; w < nw; ++w) a += tiles[w * K * (C + 1) + i];
        part[i] = a;
    }
}

static size_t sa_fwd_smem(int K, int G, int D, int H, int NH = 1) {
    const int NB = K > 8 ? 2 : 1, KB = K > 8 ? (K + 1) / 2 : G * K, KP = NB * KB;
    return (size_t)(KP * D + KB * D * 3 + KB * 3 * D * 2 + KB * H + NH * KP * SA_C * 3 + 32) * 4;
}
static size_t sa_bwd_smem(int K, int G, int D, int H, int NH = 1) {
    const int NB = K > 8 ? 2 : 1, KB = K > 8 ? (K + 1) / 2 : G * K, KP = NB * KB;
    return (size_t)(KP * D * 2 + KB * D * 2 + KB * 3 * D * 2 + KB * H + NH * KP * SA_C * 5 + 80 + 4 * D + 2 * SA_C) * 4;
}
#define SA_LDS_MAX (160 * 1024 - 256)       // dynamic LDS available to a workgroup (the kernels hold 128 bytes of static LDS)
size_t sa_xchg_floats_host(int K, int D) {
    const size_t f = sa_xchg_floats(K, D), g = sa_xchg_bwd_floats(K, D);
    return f > g ? f : g;
}
// partial sums of the streaming launches: one [K][65] block per streaming workgroup; B * NS <= SA_MAX_BLOCKS + B by construction
size_t sa_parts_floats_host(int B, int K) { return (size_t)(SA_MAX_BLOCKS + B) * K * (SA_C + 1); }

// streaming workgroups per image: `per_cu` workgroups per CU in total, so that a launch is a whole number of resident rounds (a launch
// bound of 3 with 4 workgroups per CU ran one full round and a second one a third full: the round-2 forward); at least 8 tiles of 16
// positions per workgroup (2 per wave)
static int sa_splits(int B, int N, int per_cu) {
    static int ncu = 0, mult = 0;
    if (!ncu) {
        int dev = 0;
        hipDeviceProp_t prop;
        ncu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        const char* e = getenv("OCRL_SA_WGS");          // streaming workgroups per CU per launch (development knob; 0 = the kernel's own)
        mult = e ? atoi(e) : 0;
    }
    const int ntile = (N + 15) / 16;
    int target = (mult > 0 ? mult : per_cu) * ncu;
    if (target > SA_MAX_BLOCKS) target = SA_MAX_BLOCKS;
    int ns = B <= target ? target / B : 1;              // rounded down: never more workgroups than one full round
    // OCRL_SA_NS (read at every call): a fixed split count, so that a small batch groups its partial sums exactly like a large one --
    // tests/test_gpu_benchshape.py compares a B = 128 step with a B = 1 step of the same image bit for bit
    if (const char* f = getenv("OCRL_SA_NS")) { const int v = atoi(f); if (v > 0 && (long long)B * v <= SA_MAX_BLOCKS) ns = v; }
    if (ns > ntile / 8) ns = ntile / 8;
    return ns < 1 ? 1 : ns;
}

template <int K, int G>
static int sa_launch_kg(const SlotAttnArgs& a, int backward, hipStream_t st) {
    constexpr int KP = SaBlk<K>::KP;
    const size_t smem = backward ? sa_bwd_smem(K, G, a.D, a.H) : sa_fwd_smem(K, G, a.D, a.H);
    OCRL_REQUIRE(smem <= SA_LDS_MAX, "slot_attn: LDS request %zu too large (num_slots %d, slot size %d)", smem, K, a.D);
    static_assert((SA_TS / 64) * 16 * SA_TLD >= (SA_TS / 64) * 16 * (SA_C + 1), "streaming workgroup: partial scratch must fit the tile region");
    const SaWts wo = sa_wts_layout(a.C, a.D, a.H);
    const SaSave so = sa_save_layout(a.C, a.D, a.H);
    const SaGrad go = sa_grad_layout(a.C, a.D, a.H);
    static size_t attr_f = 0, attr_b = 0;       // dynamic LDS limits granted so far (per instantiation)
    if (!backward && smem > attr_f) {
        OCRL_HIP(hipFuncSetAttribute((const void*)sa_slot_fwd_kernel<K, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_f = smem;
    }
    if (backward && smem > attr_b) {
        OCRL_HIP(hipFuncSetAttribute((const void*)sa_slot_bwd_kernel<K, G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_b = smem;
    }
    static int fwd_form = -1;          // OCRL_SA_FWD=1: the round-2 streaming forward (slots on the lanes); default: positions on the lanes
    if (fwd_form < 0) { const char* e = getenv("OCRL_SA_FWD"); fwd_form = e ? atoi(e) : 2; }
    // resident streaming workgroups per CU (launch bounds): forward 4 (3 for the round-2 form), backward 2 -- the backward runs two rounds
    const int NS = sa_splits(a.B, a.N, backward ? 4 : (fwd_form == 1 ? 3 : 4));
    const int ngrp = (a.B + G - 1) / G;
    const int pi = prof_begin(backward ? PROF_SA_BWD : PROF_SA_FWD, st);
    if (!backward) {
        const size_t smem_stream = (size_t)(KP * SA_C + 16 + (SA_TS / 64) * (16 * SA_TLD + 16 * SA_WT_LD)) * 4;
        if (a.phase != 2) hipLaunchKernelGGL((sa_slot_fwd_kernel<K, G>), dim3(ngrp), dim3(SA_TF), smem, st, a, wo, so, -1, NS);
        for (int t = 0; t < a.I && a.phase != 1; ++t) {
            if (fwd_form == 1) hipLaunchKernelGGL((sa_stream_fwd_kernel<K, false>), dim3(a.B * NS), dim3(SA_TS), smem_stream, st, a, t, NS);
            else hipLaunchKernelGGL((sa_stream_fwd_kernel<K, true>), dim3(a.B * NS), dim3(SA_TS), smem_stream, st, a, t, NS);
            hipLaunchKernelGGL((sa_slot_fwd_kernel<K, G>), dim3(ngrp), dim3(SA_TF), smem, st, a, wo, so, t, NS);
        }
    } else {
        static int bwd_form = -1;          // OCRL_SA_BWD=1: the round-2 streaming backward; default: positions on the lanes
        if (bwd_form < 0) { const char* e = getenv("OCRL_SA_BWD"); bwd_form = e ? atoi(e) : 2; }
        static_assert(16 * SA_WT_LD <= 2 * 16 * SA_WLD, "the weight patch of the second form must fit the first form's two patches");
        const size_t smem_stream = (size_t)(2 * KP * SA_C + 64 + (SA_TS / 64) * (16 * SA_TLD + 2 * 16 * SA_WLD)) * 4;
        hipLaunchKernelGGL((sa_slot_bwd_kernel<K, G>), dim3(ngrp), dim3(SA_TB), smem, st, a, wo, so, go, -1, a.I - 1, NS);
        for (int t = a.I - 1; t >= 0; --t) {
            const bool first = (t == a.I - 1), final_ = (t == 0);
            if (bwd_form == 1) {
                if (first && final_) hipLaunchKernelGGL((sa_stream_bwd_kernel<K, true, true, false>), dim3(a.B * NS), dim3(SA_TS), smem_stream, st, a, NS);
                else if (first) hipLaunchKernelGGL((sa_stream_bwd_kernel<K, true, false, false>), dim3(a.B * NS), dim3(SA_TS), smem_stream, st, a, NS);
                else if (final_) hipLaunchKernelGGL((sa_stream_bwd_kernel<K, false, true, false>), dim3(a.B * NS), dim3(SA_TS), smem_stream, st, a, NS);
                else hipLaunchKernelGGL((sa_stream_bwd_kernel<K, false, false, false>), dim3(a.B * NS), dim3(SA_TS), smem_stream, st, a, NS);
            } else {
                if (first && final_) hipLaunchKernelGGL((sa_stream_bwd_kernel<K, true, true, true>), dim3(a.B * NS), dim3(SA_TS), smem_stream, st, a, NS);
                else if (first) hipLaunchKernelGGL((sa_stream_bwd_kernel<K, true, false, true>), dim3(a.B * NS), dim3(SA_TS), smem_stream, st, a, NS);
                else if (final_) hipLaunchKernelGGL((sa_stream_bwd_kernel<K, false, true, true>), dim3(a.B * NS), dim3(SA_TS), smem_stream, st, a, NS);
                else hipLaunchKernelGGL((sa_stream_bwd_kernel<K, false, false, true>), dim3(a.B * NS), dim3(SA_TS), smem_stream, st, a, NS);
            }
            hipLaunchKernelGGL((sa_slot_bwd_kernel<K, G>), dim3(ngrp), dim3(SA_TB), smem, st, a, wo, so, go, t, t - 1, NS);
        }
    }
    prof_end(pi, st);
    OCRL_CHECK_LAUNCH("slot_attn");
    return 0;
}
// ---- several attention heads (ocrs/common/slot_attn.py:54-92).  The soft-max runs over heads * K columns and every column has its own
// folded query and weighted mean, so the streaming kernels are the single-head ones instantiated for KS = heads * K "slots"; the
// slot-side kernels (one image per workgroup) build / consume the per-head folded operands.  attn_vis = sum over the heads.
typedef void (*SaStreamFwdFn)(SlotAttnArgs, int, int);
typedef void (*SaStreamBwdFn)(SlotAttnArgs, int);
template <int KS> static SaStreamBwdFn sa_stream_bwd_pick(bool first, bool final_) {
    if (first && final_) return sa_stream_bwd_kernel<KS, true, true, true>;
    if (first) return sa_stream_bwd_kernel<KS, true, false, true>;
    if (final_) return sa_stream_bwd_kernel<KS, false, true, true>;
    return sa_stream_bwd_kernel<KS, false, false, true>;
}
static SaStreamFwdFn sa_stream_fwd_fn(int KS) {
    switch (KS) {
#define SA_CASE(k) case k: return sa_stream_fwd_kernel<k, true>;
        SA_CASE(2) SA_CASE(3) SA_CASE(4) SA_CASE(5) SA_CASE(6) SA_CASE(7) SA_CASE(8) SA_CASE(9) SA_CASE(10) SA_CASE(11) SA_CASE(12) SA_CASE(13) SA_CASE(14) SA_CASE(15) SA_CASE(16)
#undef SA_CASE
    }
    return nullptr;
}
static SaStreamBwdFn sa_stream_bwd_fn(int KS, bool first, bool final_) {
    switch (KS) {
#define SA_CASE(k) case k: return sa_stream_bwd_pick<k>(first, final_);
        SA_CASE(2) SA_CASE(3) SA_CASE(4) SA_CASE(5) SA_CASE(6) SA_CASE(7) SA_CASE(8) SA_CASE(9) SA_CASE(10) SA_CASE(11) SA_CASE(12) SA_CASE(13) SA_CASE(14) SA_CASE(15) SA_CASE(16)
#undef SA_CASE
    }
    return nullptr;
}
__global__ void sa_attn_heads_sum_kernel(const float* __restrict__ in, float* __restrict__ out, long long rows, int K, int NH) {
    const long long n = rows * K;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / K; const int k = (int)(i - r * K);
        float a = 0.f;
        for (int h = 0; h < NH; ++h) a += in[r * (NH * K) + h * K + k];
        out[i] = a;
    }
}
template <int K>
static int sa_launch_heads(const SlotAttnArgs& a0, int backward, hipStream_t st) {
    const int NH = a0.NH, KS = NH * K, KPS = sa_ki(KS);
    const size_t smem = backward ? sa_bwd_smem(K, 1, a0.D, a0.H, NH) : sa_fwd_smem(K, 1, a0.D, a0.H, NH);
    OCRL_REQUIRE(smem <= SA_LDS_MAX, "slot_attn: LDS request %zu too large (num_slots %d, heads %d, slot size %d)", smem, K, NH, a0.D);
    OCRL_REQUIRE(!a0.attn || a0.attn_heads, "slot_attn: the per-head attention scratch is missing");
    SlotAttnArgs a = a0;
    if (a.attn) a.attn = a0.attn_heads;          // the streaming kernel writes [B,N,KS]
    const SaWts wo = sa_wts_layout(a.C, a.D, a.H);
    const SaSave so = sa_save_layout(a.C, a.D, a.H, NH);
    const SaGrad go = sa_grad_layout(a.C, a.D, a.H, NH);
    OCRL_HIP(hipFuncSetAttribute((const void*)(backward ? (const void*)sa_slot_bwd_kernel<K, 1> : (const void*)sa_slot_fwd_kernel<K, 1>),
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)SA_LDS_MAX));
    const int NS = sa_splits(a.B, a.N, 4);
    const int pi = prof_begin(backward ? PROF_SA_BWD : PROF_SA_FWD, st);
    if (!backward) {
        const size_t smem_stream = (size_t)(KPS * SA_C + 16 + (SA_TS / 64) * (16 * SA_TLD + 16 * SA_WT_LD)) * 4;
        const SaStreamFwdFn fs = sa_stream_fwd_fn(KS);
        if (a.phase != 2) hipLaunchKernelGGL((sa_slot_fwd_kernel<K, 1>), dim3(a.B), dim3(SA_TF), smem, st, a, wo, so, -1, NS);
        for (int t = 0; t < a.I && a.phase != 1; ++t) {
            hipLaunchKernelGGL(fs, dim3(a.B * NS), dim3(SA_TS), smem_stream, st, a, t, NS);
            hipLaunchKernelGGL((sa_slot_fwd_kernel<K, 1>), dim3(a.B), dim3(SA_TF), smem, st, a, wo, so, t, NS);
        }
        if (a0.attn && a.phase != 1) {
            const long long rows = (long long)a.B * a.N;
            hipLaunchKernelGGL(sa_attn_heads_sum_kernel, dim3(cdiv(rows * K, 256) > 4096 ? 4096 : cdiv(rows * K, 256)), dim3(256), 0, st, a0.attn_heads, a0.attn, rows, K, NH);
        }
    } else {
        const size_t smem_stream = (size_t)(2 * KPS * SA_C + 64 + (SA_TS / 64) * (16 * SA_TLD + 2 * 16 * SA_WLD)) * 4;
        hipLaunchKernelGGL((sa_slot_bwd_kernel<K, 1>), dim3(a.B), dim3(SA_TB), smem, st, a, wo, so, go, -1, a.I - 1, NS);
        for (int t = a.I - 1; t >= 0; --t) {
            hipLaunchKernelGGL(sa_stream_bwd_fn(KS, t == a.I - 1, t == 0), dim3(a.B * NS), dim3(SA_TS), smem_stream, st, a, NS);
            hipLaunchKernelGGL((sa_slot_bwd_kernel<K, 1>), dim3(a.B), dim3(SA_TB), smem, st, a, wo, so, go, t, t - 1, NS);
        }
    }
    prof_end(pi, st);
    OCRL_CHECK_LAUNCH("slot_attn (heads)");
    return 0;
}

template <int K>
static int sa_launch_k(const SlotAttnArgs& a, int backward, hipStream_t st) {
    OCRL_REQUIRE(a.xchg && a.parts, "slot_attn: exchange / partial buffers missing");
    // images per slot-side workgroup: as many as fill the 16 MFMA rows, unless the group's rows do not fit the LDS (wide slots)
    constexpr int GM = K <= 8 ? 16 / K : 1;
    static int gmode = -1;
    if (gmode < 0) { const char* e = getenv("OCRL_SA_GROUP"); gmode = e ? atoi(e) : 1; }      // 0: one image per workgroup (development comparison)
    if (GM > 1 && gmode && sa_fwd_smem(K, GM, a.D, a.H) <= SA_LDS_MAX && sa_bwd_smem(K, GM, a.D, a.H) <= SA_LDS_MAX) return sa_launch_kg<K, GM>(a, backward, st);
    return sa_launch_kg<K, 1>(a, backward, st);
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
    OCRL_REQUIRE(a.xchg && a.parts && ((uintptr_t)a.xchg & 15) == 0, "slot_attn: exchange / partial buffers missing or xchg not 16-byte aligned");
    OCRL_REQUIRE(a.NH >= 1 && a.D % a.NH == 0, "slot_attn: the slot size %d does not divide into %d heads", a.D, a.NH);
    if (a.NH > 1) {
        OCRL_REQUIRE(a.K <= 8 && a.NH * a.K <= 16 && (a.D / a.NH) % 16 == 0,
                     "slot_attn: with %d heads, heads * num_slots <= 16 and a head width that is a multiple of 16 are supported (num_slots %d, slot size %d)", a.NH, a.K, a.D);
        switch (a.K) {
#define SA_CASE(k) case k: return sa_launch_heads<k>(a, backward, st);
            SA_CASE(1) SA_CASE(2) SA_CASE(3) SA_CASE(4) SA_CASE(5) SA_CASE(6) SA_CASE(7) SA_CASE(8)
#undef SA_CASE
        }
        return -1;
    }
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
        if (e.transpose >= 2) {     // tiled for the slot-side matrix products (MvJob): Wn[col][k] -> [col / 16][k / 16][16 * ((k % 16) / 4) + col % 16][k % 4]
            // mode 2: Wn = src ([rows = columns of the product][cols = k]);  mode 3: Wn = src^T
            const int Et = e.transpose == 2 ? e.cols : e.rows, nks = Et >> 4;
            const int j = i & 3, lane = (i >> 2) & 63, rest = i >> 8, s = rest % nks, tile = rest / nks;
            const int col = tile * 16 + (lane & 15), k = s * 16 + (lane >> 4) * 4 + j;
            dst[e.dst_off + i] = e.transpose == 2 ? e.src[col * Et + k] : e.src[k * e.cols + col];
        } else if (e.transpose) {      // dst[c][r] = src[r][c], i enumerates dst
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
