// Causal multi-head self-attention of the SLATE transformer decoder (reference:
// ocrs/common/transformer.py:23-50), flash-style: scores are never written to HBM.
// fp32 on v_mfma_f32_16x16x4_f32; softmax statistics online; dropout on the probabilities from the
// stateless counter RNG (identical decisions in forward and backward, index = ((b*h+head)*T+q)*T+key).
//
// Orientation trick (no LDS round trip for P): the forward and the dQ kernel compute S^T = K Q^T, so a lane
// holds, for its query (lane & 15), the four consecutive keys 4*(lane>>4)+r in accumulator registers r = 0..3 —
// exactly the B-operand layout of the following  O^T += V^T P^T  (resp. dQ^T += K^T dS^T) MFMA steps.  The
// dK/dV kernel owns 16 keys per wave (K, V fragments in registers), computes S = Q K^T with queries on rows
// and feeds P / dS the same way into  dV^T += dO^T P  and  dK^T += Q^T dS.
//
// One workgroup = 4 waves = 64 queries (or 64 keys); tiles of 64 keys (queries) stream through LDS.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

#define FA_BLK 64
#ifndef FA_LPT
#define FA_LPT 1          // 1: longest-first block order; 0: (image, head)-major order with the block index rotated per pair
#endif
#ifndef FA_BWD_WGS
#define FA_BWD_WGS 3      // workgroups per CU the backward kernels are compiled for (register budget 512 / FA_BWD_WGS per lane)
#endif

typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <int DH>
struct FaCfg {
    static constexpr int NC = DH / 16;       // 16-wide chunks of the head dimension
    static constexpr int LD = DH + 4;        // LDS row stride (floats)
};

// cooperative load of a [64 x DH] tile (rows row0.., zero beyond T) of one head into LDS, optionally scaled
template <int DH>
__device__ inline void fa_load_tile(float* s, const float* __restrict__ g, long long bt0, int row0, int T, int d, int hoff, float scale) {
    constexpr int F4 = DH / 4;
    constexpr int LD = FaCfg<DH>::LD;
    for (int idx = threadIdx.x; idx < FA_BLK * F4; idx += 256) {
        const int c4 = idx % F4, r = idx / F4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row0 + r < T) v = *reinterpret_cast<const float4*>(g + (bt0 + row0 + r) * d + hoff + c4 * 4);
        v.x *= scale; v.y *= scale; v.z *= scale; v.w *= scale;
        *reinterpret_cast<float4*>(s + r * LD + c4 * 4) = v;
    }
}

// Register-staged variant: fa_fetch issues the global loads of a tile into registers, fa_commit writes them to LDS one iteration
// later, so the memory round trip of tile t+1 overlaps the MFMA work on tile t instead of sitting between two barriers.
template <int DH> struct FaRegs { static constexpr int N = (FA_BLK * (DH / 4) + 255) / 256; };
template <int DH>
__device__ inline void fa_fetch(float4 (&r)[FaRegs<DH>::N], const float* __restrict__ g, long long bt0, int row0, int T, int d, int hoff) {
    constexpr int F4 = DH / 4;
#pragma unroll
    for (int i = 0; i < FaRegs<DH>::N; ++i) {
        const int idx = threadIdx.x + i * 256;
        const int c4 = idx % F4, row = idx / F4;
        r[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < FA_BLK * F4 && row0 + row < T) r[i] = *reinterpret_cast<const float4*>(g + (bt0 + row0 + row) * d + hoff + c4 * 4);
    }
}
template <int DH>
__device__ inline void fa_commit(float* s, const float4 (&r)[FaRegs<DH>::N], float scale) {
    constexpr int F4 = DH / 4, LD = FaCfg<DH>::LD;
#pragma unroll
    for (int i = 0; i < FaRegs<DH>::N; ++i) {
        const int idx = threadIdx.x + i * 256;
        if (idx < FA_BLK * F4) *reinterpret_cast<float4*>(s + (idx / F4) * LD + (idx % F4) * 4) = make_float4(r[i].x * scale, r[i].y * scale, r[i].z * scale, r[i].w * scale);
    }
}

// Blocks are dealt round-robin over the 8 XCDs, and causal work per block grows with its index: a fixed
// blockIdx -> block-index map gives each XCD always the same (heavy or light) classes — measured 2.4x imbalance,
// CUs 67 % busy.  Rotating the block index by the (batch, head) index hands every XCD a uniform mix.
__device__ inline int fa_block_index(int j, long long bh, int nblk) {
    const long long rot = nblk >= 8 ? bh : bh / (8 / (nblk > 0 ? nblk : 1) > 0 ? 8 / nblk : 1);
    return (int)((j + rot) % nblk);
}

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ------------------------------------------------------------------------------------------- forward
template <int DH>
__global__ __launch_bounds__(256, 4) void attn_fwd_kernel(AttnArgs p) {
    constexpr int NC = FaCfg<DH>::NC, LD = FaCfg<DH>::LD;
    __shared__ __attribute__((aligned(16))) float Ks[FA_BLK * LD];
    __shared__ __attribute__((aligned(16))) float Vs[FA_BLK * LD];
    const int nqb = (p.T + FA_BLK - 1) / FA_BLK;
    // longest-first dispatch: blockIdx runs over the tile-count classes from the heaviest (last query block: nqb key tiles) to the
    // lightest, all (image, head) pairs of a class together, so the launch ends on one-tile blocks instead of on a 16-tile block
    // that started late (with the rotated (image, head)-major order a third of the wave slots sat empty: SQ_WAVE_CYCLES).
    // Every class has B*h blocks dealt round-robin to the XCDs, so each XCD still gets the same mix.
    int bid = blockIdx.x;
    const int nbh = gridDim.x / nqb;
    const int qb = FA_LPT ? nqb - 1 - bid / nbh : fa_block_index(bid % nqb, bid / nqb, nqb);
    bid = FA_LPT ? bid % nbh : bid / nqb;
    const int hd = bid % p.h;
    const long long b = bid / p.h;
    const int T = p.T, d = p.ld, hoff = hd * DH;
    const long long bt0 = b * T;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int q_abs = qb * FA_BLK + wv * 16 + li;            // this lane's query (column of S^T)
    const float scale = rsqrtf((float)DH);
    const long long bh = b * p.h + hd;

    float4 qf[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q_abs < T) v = *reinterpret_cast<const float4*>(p.q + (bt0 + q_abs) * d + hoff + 16 * c + 4 * g);
        qf[c] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
    }
    f32x4_t o[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) o[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    const uint32_t thr = drop_thresh(p.p);
    const float dsc = p.p > 0.f ? 1.0f / (1.0f - p.p) : 1.f;

    float4 rk[FaRegs<DH>::N], rv[FaRegs<DH>::N];
    fa_fetch<DH>(rk, p.k, bt0, 0, T, d, hoff);
    fa_fetch<DH>(rv, p.v, bt0, 0, T, d, hoff);
    for (int kt = 0; kt <= qb; ++kt) {
        __syncthreads();
        fa_commit<DH>(Ks, rk, 1.f);
        fa_commit<DH>(Vs, rv, 1.f);
        __syncthreads();
        if (kt < qb) {                                         // next tile in flight during this tile's MFMAs
            fa_fetch<DH>(rk, p.k, bt0, (kt + 1) * FA_BLK, T, d, hoff);
            fa_fetch<DH>(rv, p.v, bt0, (kt + 1) * FA_BLK, T, d, hoff);
        }
        f32x4_t s[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) s[st] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float4 a[4];
#pragma unroll
            for (int st = 0; st < 4; ++st) a[st] = *reinterpret_cast<const float4*>(Ks + (16 * st + li) * LD + 16 * c + 4 * g);
            // four independent accumulation chains keep the MFMA pipe issuing back to back
#pragma unroll
            for (int st = 0; st < 4; ++st) s[st] = MFMA16(a[st].x, qf[c].x, s[st]);
#pragma unroll
            for (int st = 0; st < 4; ++st) s[st] = MFMA16(a[st].y, qf[c].y, s[st]);
#pragma unroll
            for (int st = 0; st < 4; ++st) s[st] = MFMA16(a[st].z, qf[c].z, s[st]);
#pragma unroll
            for (int st = 0; st < 4; ++st) s[st] = MFMA16(a[st].w, qf[c].w, s[st]);
        }
        float mx = -INFINITY;
        if (kt == qb || (kt + 1) * FA_BLK > T) {          // only the diagonal / ragged tile needs masking
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const int key0 = kt * FA_BLK + 16 * st + 4 * g;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (key0 + r > q_abs || key0 + r >= T) s[st][r] = -INFINITY;
            }
        }
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[st][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m, mx);
        const float alpha = (m_new == -INFINITY) ? 1.f : __expf(m - m_new);
        float rs = 0.f;
#pragma unroll
        for (int st = 0; st < 4; ++st)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = (m_new == -INFINITY) ? 0.f : __expf(s[st][r] - m_new);
                s[st][r] = e;
                rs += e;
            }
        rs += __shfl_xor(rs, 16, 64);
        rs += __shfl_xor(rs, 32, 64);
        l = l * alpha + rs;
        m = m_new;
#pragma unroll
        for (int c = 0; c < NC; ++c) o[c] *= alpha;
        if (p.p > 0.f) {
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const uint64_t idx = ((uint64_t)bh * T + (uint64_t)(q_abs < T ? q_abs : 0)) * T + (uint64_t)(kt * FA_BLK + 16 * st + 4 * g);
                const uint2 bits = rng_bits4(p.seed, p.site, idx >> 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) s[st][r] = rng_keep(bits, r, thr) ? s[st][r] * dsc : 0.f;
            }
        }
        // V^T operands of sub-tile st + 1 are read from LDS before the 4 * NC MFMAs of sub-tile st are issued (ping-pong registers;
        // the scheduling barriers keep that order), so the LDS latency sits under a sub-tile's worth of matrix work
        float av[2][4][NC];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < NC; ++c) av[0][r][c] = Vs[(4 * g + r) * LD + 16 * c + li];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            if (st + 1 < 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < NC; ++c) av[(st + 1) & 1][r][c] = Vs[(16 * (st + 1) + 4 * g + r) * LD + 16 * c + li];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < NC; ++c) o[c] = MFMA16(av[st & 1][r][c], s[st][r], o[c]);          // NC independent chains per step
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (q_abs < T) {
        const float inv = 1.0f / l;
#pragma unroll
        for (int c = 0; c < NC; ++c)
            *reinterpret_cast<float4*>(p.o + (bt0 + q_abs) * p.d + hoff + 16 * c + 4 * g) = make_float4(o[c][0] * inv, o[c][1] * inv, o[c][2] * inv, o[c][3] * inv);
        if (g == 0) p.lse[bh * T + q_abs] = m + __logf(l);
    }
}

// delta[b,h,q] = sum_dim dO[q,dim] * O[q,dim].  A row's d/4 float4 are read by d/4 consecutive threads (full cache lines);
// the per-head sums go through LDS.  RPB rows per 256-thread block.
__global__ __launch_bounds__(256) void attn_delta_kernel(const float* __restrict__ dO, const float* __restrict__ O, float* __restrict__ delta, long long BT, int T, int d,
                                                         int h) {
    __shared__ float part[256];
    const int F4 = d >> 2, RPB = 256 / F4, HF = F4 / h;            // float4 per row, rows per block, float4 per head
    const int rl = threadIdx.x / F4, c4 = threadIdx.x - rl * F4;
    const long long row = (long long)blockIdx.x * RPB + rl;
    float s = 0.f;
    if (rl < RPB && row < BT) {
        const float4 x = *reinterpret_cast<const float4*>(dO + row * d + c4 * 4), y = *reinterpret_cast<const float4*>(O + row * d + c4 * 4);
        s = x.x * y.x + x.y * y.y + x.z * y.z + x.w * y.w;
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if ((int)threadIdx.x < RPB * h) {
        const int r2 = threadIdx.x / h, hd = threadIdx.x - r2 * h;
        const long long row2 = (long long)blockIdx.x * RPB + r2;
        if (row2 < BT) {
            float t = 0.f;
            for (int k = 0; k < HF; ++k) t += part[r2 * F4 + hd * HF + k];
            const long long bb = row2 / T, q = row2 % T;
            delta[(bb * h + hd) * T + q] = t;
        }
    }
}

// ------------------------------------------------------------------------------------------- backward: dK, dV
template <int DH>
__global__ __launch_bounds__(256, FA_BWD_WGS) void attn_bwd_kv_kernel(AttnArgs p) {
    constexpr int NC = FaCfg<DH>::NC, LD = FaCfg<DH>::LD;
    __shared__ __attribute__((aligned(16))) float Qs[FA_BLK * LD];
    __shared__ __attribute__((aligned(16))) float Gs[FA_BLK * LD];      // dO
    __shared__ float Ls[FA_BLK], Ds[FA_BLK];
    const int nkb = (p.T + FA_BLK - 1) / FA_BLK;
    int bid = blockIdx.x;                                   // longest first: key block 0 meets every query tile
    const int nbh = gridDim.x / nkb;
    const int kb = FA_LPT ? bid / nbh : fa_block_index(bid % nkb, bid / nkb, nkb);
    bid = FA_LPT ? bid % nbh : bid / nkb;
    const int hd = bid % p.h;
    const long long b = bid / p.h;
    const int T = p.T, d = p.ld, hoff = hd * DH;
    const long long bt0 = b * T;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int k_abs = kb * FA_BLK + wv * 16 + li;                       // this lane's key (column of S)
    const float scale = rsqrtf((float)DH);
    const long long bh = b * p.h + hd;

    float4 kf[NC], vf[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        kf[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        vf[c] = kf[c];
        if (k_abs < T) {
            kf[c] = *reinterpret_cast<const float4*>(p.k + (bt0 + k_abs) * d + hoff + 16 * c + 4 * g);
            vf[c] = *reinterpret_cast<const float4*>(p.v + (bt0 + k_abs) * d + hoff + 16 * c + 4 * g);
        }
    }
    f32x4_t dk[NC], dv[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { dk[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[c] = dk[c]; }
    const uint32_t thr = drop_thresh(p.p);
    const float dsc = p.p > 0.f ? 1.0f / (1.0f - p.p) : 1.f;

    float4 rq[FaRegs<DH>::N], rg[FaRegs<DH>::N];
    float rl = 0.f, rd = 0.f;
    auto fetch = [&](int qt) {
        fa_fetch<DH>(rq, p.q, bt0, qt * FA_BLK, T, d, hoff);
        fa_fetch<DH>(rg, p.dO, bt0, qt * FA_BLK, T, p.d, hoff);
        if (threadIdx.x < FA_BLK) {
            const int qq = qt * FA_BLK + threadIdx.x;
            rl = qq < T ? p.lse[bh * T + qq] : 0.f;
            rd = qq < T ? p.delta[bh * T + qq] : 0.f;
        }
    };
    fetch(kb);
    for (int qt = kb; qt < nkb; ++qt) {
        __syncthreads();
        fa_commit<DH>(Qs, rq, scale);
        fa_commit<DH>(Gs, rg, 1.f);
        if (threadIdx.x < FA_BLK) { Ls[threadIdx.x] = rl; Ds[threadIdx.x] = rd; }
        __syncthreads();
        if (qt + 1 < nkb) fetch(qt + 1);
#pragma unroll
        for (int sq = 0; sq < 4; ++sq) {
            f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const float4 a = *reinterpret_cast<const float4*>(Qs + (16 * sq + li) * LD + 16 * c + 4 * g);
                s = MFMA16(a.x, kf[c].x, s);
                s = MFMA16(a.y, kf[c].y, s);
                s = MFMA16(a.z, kf[c].z, s);
                s = MFMA16(a.w, kf[c].w, s);
                const float4 e = *reinterpret_cast<const float4*>(Gs + (16 * sq + li) * LD + 16 * c + 4 * g);
                dp = MFMA16(e.x, vf[c].x, dp);
                dp = MFMA16(e.y, vf[c].y, dp);
                dp = MFMA16(e.z, vf[c].z, dp);
                dp = MFMA16(e.w, vf[c].w, dp);
            }
            // s[r] = S[query 16*sq+4g+r][key k_abs], dp[r] likewise.  Dropout: the decision for (query, key) lives in the
            // 64-bit draw of (query, key/4); lane (g, li) draws for query 4g + (li&3) and its key group, and the quad
            // exchanges the draws with DPP broadcasts (1 hash per lane per sub-tile instead of 4).
            uint2 hb = make_uint2(0u, 0u);
            if (p.p > 0.f) {
                const int qh = qt * FA_BLK + 16 * sq + 4 * g + (li & 3);
                const uint64_t idx = ((uint64_t)bh * T + (uint64_t)(qh < T ? qh : 0)) * T + (uint64_t)(k_abs < T ? k_abs : 0);
                hb = rng_bits4(p.seed, p.site, idx >> 2);
            }
            uint2 hq[4];
            hq[0] = make_uint2(__builtin_amdgcn_update_dpp(0, (int)hb.x, 0x00, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(0, (int)hb.y, 0x00, 0xF, 0xF, false));
            hq[1] = make_uint2(__builtin_amdgcn_update_dpp(0, (int)hb.x, 0x55, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(0, (int)hb.y, 0x55, 0xF, 0xF, false));
            hq[2] = make_uint2(__builtin_amdgcn_update_dpp(0, (int)hb.x, 0xAA, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(0, (int)hb.y, 0xAA, 0xF, 0xF, false));
            hq[3] = make_uint2(__builtin_amdgcn_update_dpp(0, (int)hb.x, 0xFF, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(0, (int)hb.y, 0xFF, 0xF, 0xF, false));
            f32x4_t pd, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ql = 16 * sq + 4 * g + r;
                const int qa = qt * FA_BLK + ql;
                float pr = 0.f;
                if (qa < T && k_abs <= qa && k_abs < T) pr = __expf(s[r] - Ls[ql]);
                float keep = 1.f;
                if (p.p > 0.f) keep = rng_keep(hq[r], li & 3, thr) ? dsc : 0.f;
                pd[r] = pr * keep;
                ds[r] = pr * (dp[r] * keep - Ds[ql]);
            }
            // all 8 * NC transposed operands of the sub-tile are requested first, then the 8 * NC MFMAs run (2 * NC independent chains)
            float ag[4][NC], aq[4][NC];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    ag[r][c] = Gs[(16 * sq + 4 * g + r) * LD + 16 * c + li];
                    aq[r][c] = Qs[(16 * sq + 4 * g + r) * LD + 16 * c + li];
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    dv[c] = MFMA16(ag[r][c], pd[r], dv[c]);
                    dk[c] = MFMA16(aq[r][c], ds[r], dk[c]);
                }
        }
    }
    if (k_abs < T) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            *reinterpret_cast<float4*>(p.dv + (bt0 + k_abs) * d + hoff + 16 * c + 4 * g) = make_float4(dv[c][0], dv[c][1], dv[c][2], dv[c][3]);
            *reinterpret_cast<float4*>(p.dk + (bt0 + k_abs) * d + hoff + 16 * c + 4 * g) = make_float4(dk[c][0], dk[c][1], dk[c][2], dk[c][3]);
        }
    }
}

// ------------------------------------------------------------------------------------------- backward: dQ
template <int DH>
__global__ __launch_bounds__(256, FA_BWD_WGS) void attn_bwd_q_kernel(AttnArgs p) {
    constexpr int NC = FaCfg<DH>::NC, LD = FaCfg<DH>::LD;
    __shared__ __attribute__((aligned(16))) float Ks[FA_BLK * LD];
    __shared__ __attribute__((aligned(16))) float Vs[FA_BLK * LD];
    const int nqb = (p.T + FA_BLK - 1) / FA_BLK;
    // longest-first dispatch: blockIdx runs over the tile-count classes from the heaviest (last query block: nqb key tiles) to the
    // lightest, all (image, head) pairs of a class together, so the launch ends on one-tile blocks instead of on a 16-tile block
    // that started late (with the rotated (image, head)-major order a third of the wave slots sat empty: SQ_WAVE_CYCLES).
    // Every class has B*h blocks dealt round-robin to the XCDs, so each XCD still gets the same mix.
    int bid = blockIdx.x;
    const int nbh = gridDim.x / nqb;
    const int qb = FA_LPT ? nqb - 1 - bid / nbh : fa_block_index(bid % nqb, bid / nqb, nqb);
    bid = FA_LPT ? bid % nbh : bid / nqb;
    const int hd = bid % p.h;
    const long long b = bid / p.h;
    const int T = p.T, d = p.ld, hoff = hd * DH;
    const long long bt0 = b * T;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int q_abs = qb * FA_BLK + wv * 16 + li;
    const float scale = rsqrtf((float)DH);
    const long long bh = b * p.h + hd;

    float4 qf[NC], gf[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        qf[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        gf[c] = qf[c];
        if (q_abs < T) {
            const float4 v = *reinterpret_cast<const float4*>(p.q + (bt0 + q_abs) * d + hoff + 16 * c + 4 * g);
            qf[c] = make_float4(v.x * scale, v.y * scale, v.z * scale, v.w * scale);
            gf[c] = *reinterpret_cast<const float4*>(p.dO + (bt0 + q_abs) * p.d + hoff + 16 * c + 4 * g);
        }
    }
    const float lse = q_abs < T ? p.lse[bh * T + q_abs] : 0.f;
    const float dlt = q_abs < T ? p.delta[bh * T + q_abs] : 0.f;
    f32x4_t dq[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) dq[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    const uint32_t thr = drop_thresh(p.p);
    const float dsc = p.p > 0.f ? 1.0f / (1.0f - p.p) : 1.f;

    float4 rk[FaRegs<DH>::N], rv[FaRegs<DH>::N];
    fa_fetch<DH>(rk, p.k, bt0, 0, T, d, hoff);
    fa_fetch<DH>(rv, p.v, bt0, 0, T, d, hoff);
    for (int kt = 0; kt <= qb; ++kt) {
        __syncthreads();
        fa_commit<DH>(Ks, rk, 1.f);
        fa_commit<DH>(Vs, rv, 1.f);
        __syncthreads();
        if (kt < qb) {
            fa_fetch<DH>(rk, p.k, bt0, (kt + 1) * FA_BLK, T, d, hoff);
            fa_fetch<DH>(rv, p.v, bt0, (kt + 1) * FA_BLK, T, d, hoff);
        }
        f32x4_t sS[4], sP[4];
#pragma unroll
        for (int st = 0; st < 4; ++st) { sS[st] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; sP[st] = sS[st]; }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float4 a[4], e[4];
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                a[st] = *reinterpret_cast<const float4*>(Ks + (16 * st + li) * LD + 16 * c + 4 * g);
                e[st] = *reinterpret_cast<const float4*>(Vs + (16 * st + li) * LD + 16 * c + 4 * g);
            }
#pragma unroll
            for (int st = 0; st < 4; ++st) { sS[st] = MFMA16(a[st].x, qf[c].x, sS[st]); sP[st] = MFMA16(e[st].x, gf[c].x, sP[st]); }
#pragma unroll
            for (int st = 0; st < 4; ++st) { sS[st] = MFMA16(a[st].y, qf[c].y, sS[st]); sP[st] = MFMA16(e[st].y, gf[c].y, sP[st]); }
#pragma unroll
            for (int st = 0; st < 4; ++st) { sS[st] = MFMA16(a[st].z, qf[c].z, sS[st]); sP[st] = MFMA16(e[st].z, gf[c].z, sP[st]); }
#pragma unroll
            for (int st = 0; st < 4; ++st) { sS[st] = MFMA16(a[st].w, qf[c].w, sS[st]); sP[st] = MFMA16(e[st].w, gf[c].w, sP[st]); }
        }
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int key0 = kt * FA_BLK + 16 * st + 4 * g;
            uint2 bits = make_uint2(0u, 0u);
            if (p.p > 0.f) {
                const uint64_t idx = ((uint64_t)bh * T + (uint64_t)(q_abs < T ? q_abs : 0)) * T + (uint64_t)key0;
                bits = rng_bits4(p.seed, p.site, idx >> 2);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pr = 0.f;
                if (q_abs < T && key0 + r <= q_abs && key0 + r < T) pr = __expf(sS[st][r] - lse);
                float keep = 1.f;
                if (p.p > 0.f) keep = rng_keep(bits, r, thr) ? dsc : 0.f;
                sS[st][r] = pr * (sP[st][r] * keep - dlt);          // dS^T
            }
        }
        float ak[2][4][NC];                              // K^T operands one sub-tile ahead of their MFMAs (see the forward kernel)
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < NC; ++c) ak[0][r][c] = Ks[(4 * g + r) * LD + 16 * c + li];
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            if (st + 1 < 4) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < NC; ++c) ak[(st + 1) & 1][r][c] = Ks[(16 * (st + 1) + 4 * g + r) * LD + 16 * c + li];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < NC; ++c) dq[c] = MFMA16(ak[st & 1][r][c], sS[st][r], dq[c]);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    if (q_abs < T) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
            *reinterpret_cast<float4*>(p.dq + (bt0 + q_abs) * d + hoff + 16 * c + 4 * g) =
                make_float4(dq[c][0] * scale, dq[c][1] * scale, dq[c][2] * scale, dq[c][3] * scale);
    }
}

// ------------------------------------------------------------------------------------------- backward, one pass: dK, dV and dQ
// One workgroup of 8 waves per (image, head) walks the causal triangle: key blocks of 128 (16 keys per wave) outermost, 64-query tiles
// inside.  Per pair it computes S and dP once (the two-kernel form computes them twice: 7 product-equivalents, 5 here), accumulates
// dK / dV of its keys in registers, and adds the pair's dQ contribution to the query tile in global memory: the workgroup is the only
// writer of its (image, head)'s dQ, a thread always touches the same addresses of a tile, and the key blocks arrive in a fixed order, so
// the read-modify-write needs no atomics and is bitwise reproducible.  For the dQ product the score gradients go through LDS once
// ([64 queries][128 keys]); wave (qs, kh) takes 16 queries x 64 keys with its 64 keys' K rows held as B operands in registers, the upper
// key half hands its partial to the lower one through LDS.  Every workgroup does the same work (no causal imbalance, no dispatch order
// tricks); B * h workgroups.  Selected by OCRL_ATTN_BWD=1 (attn_launch).
#define FB_KB 128
template <int DH>
__global__ __launch_bounds__(512, 2) void attn_bwd_fused_kernel(AttnArgs p) {
    constexpr int NC = FaCfg<DH>::NC, LD = FaCfg<DH>::LD, SLD = FB_KB + 4;
    __shared__ __attribute__((aligned(16))) float Qs[FA_BLK * LD];
    __shared__ __attribute__((aligned(16))) float Gs[FA_BLK * LD];      // dO
    __shared__ __attribute__((aligned(16))) float Ss[FA_BLK * SLD];     // dS [query][key of the block]
    __shared__ __attribute__((aligned(16))) float Rs[4 * 16 * DH];      // dQ partials of the upper key half
    __shared__ float Ls[FA_BLK], Ds[FA_BLK];
    const int hd = blockIdx.x % p.h;
    const long long b = blockIdx.x / p.h;
    const int T = p.T, d = p.ld, hoff = hd * DH;
    const long long bt0 = b * T;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int li = lane & 15, g = lane >> 4;
    const int qs = wv & 3, kh = wv >> 2;
    const float scale = rsqrtf((float)DH);
    const long long bh = b * p.h + hd;
    const uint32_t thr = drop_thresh(p.p);
    const float dsc = p.p > 0.f ? 1.0f / (1.0f - p.p) : 1.f;
    const int nqt = (T + FA_BLK - 1) / FA_BLK, nkb = (T + FB_KB - 1) / FB_KB;
    constexpr int F4 = DH / 4;
#pragma unroll 1
    for (int kb = 0; kb < nkb; ++kb) {
        const int k_abs = kb * FB_KB + wv * 16 + li;                       // this lane's key (column of S)
        float4 kf[NC], vf[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            kf[c] = make_float4(0.f, 0.f, 0.f, 0.f);
            vf[c] = kf[c];
            if (k_abs < T) {
                kf[c] = *reinterpret_cast<const float4*>(p.k + (bt0 + k_abs) * d + hoff + 16 * c + 4 * g);
                vf[c] = *reinterpret_cast<const float4*>(p.v + (bt0 + k_abs) * d + hoff + 16 * c + 4 * g);
            }
        }
        // B operands of the dQ product: step (j, e) of lane group g uses key kh*64 + 16 j + 4 g + e, column = head dimension 16 c + li
        float kq[16][NC];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const int key = kb * FB_KB + kh * 64 + 16 * (s >> 2) + 4 * g + (s & 3);
#pragma unroll
            for (int c = 0; c < NC; ++c) kq[s][c] = key < T ? p.k[(bt0 + key) * d + hoff + 16 * c + li] * scale : 0.f;
        }
        f32x4_t dk[NC], dv[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) { dk[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f}; dv[c] = dk[c]; }
#pragma unroll 1
        for (int qt = (kb * FB_KB) / FA_BLK; qt < nqt; ++qt) {
            __syncthreads();                                  // the previous pair is done with Qs / Gs / Ss / Rs
            for (int idx = tid; idx < FA_BLK * F4; idx += 512) {
                const int c4 = idx % F4, r = idx / F4;
                float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f), g4 = q4;
                if (qt * FA_BLK + r < T) {
                    q4 = *reinterpret_cast<const float4*>(p.q + (bt0 + qt * FA_BLK + r) * d + hoff + c4 * 4);
                    g4 = *reinterpret_cast<const float4*>(p.dO + (bt0 + qt * FA_BLK + r) * p.d + hoff + c4 * 4);
                }
                *reinterpret_cast<float4*>(Qs + r * LD + c4 * 4) = make_float4(q4.x * scale, q4.y * scale, q4.z * scale, q4.w * scale);
                *reinterpret_cast<float4*>(Gs + r * LD + c4 * 4) = g4;
            }
            if (tid < FA_BLK) {
                const int qq = qt * FA_BLK + tid;
                Ls[tid] = qq < T ? p.lse[bh * T + qq] : 0.f;
                Ds[tid] = qq < T ? p.delta[bh * T + qq] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int sq = 0; sq < 4; ++sq) {
                f32x4_t s = (f32x4_t){0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    const float4 a = *reinterpret_cast<const float4*>(Qs + (16 * sq + li) * LD + 16 * c + 4 * g);
                    s = MFMA16(a.x, kf[c].x, s);
                    s = MFMA16(a.y, kf[c].y, s);
                    s = MFMA16(a.z, kf[c].z, s);
                    s = MFMA16(a.w, kf[c].w, s);
                    const float4 e = *reinterpret_cast<const float4*>(Gs + (16 * sq + li) * LD + 16 * c + 4 * g);
                    dp = MFMA16(e.x, vf[c].x, dp);
                    dp = MFMA16(e.y, vf[c].y, dp);
                    dp = MFMA16(e.z, vf[c].z, dp);
                    dp = MFMA16(e.w, vf[c].w, dp);
                }
                uint2 hb = make_uint2(0u, 0u);
                if (p.p > 0.f) {
                    const int qh = qt * FA_BLK + 16 * sq + 4 * g + (li & 3);
                    const uint64_t idx = ((uint64_t)bh * T + (uint64_t)(qh < T ? qh : 0)) * T + (uint64_t)(k_abs < T ? k_abs : 0);
                    hb = rng_bits4(p.seed, p.site, idx >> 2);
                }
                uint2 hq[4];
                hq[0] = make_uint2(__builtin_amdgcn_update_dpp(0, (int)hb.x, 0x00, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(0, (int)hb.y, 0x00, 0xF, 0xF, false));
                hq[1] = make_uint2(__builtin_amdgcn_update_dpp(0, (int)hb.x, 0x55, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(0, (int)hb.y, 0x55, 0xF, 0xF, false));
                hq[2] = make_uint2(__builtin_amdgcn_update_dpp(0, (int)hb.x, 0xAA, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(0, (int)hb.y, 0xAA, 0xF, 0xF, false));
                hq[3] = make_uint2(__builtin_amdgcn_update_dpp(0, (int)hb.x, 0xFF, 0xF, 0xF, false), __builtin_amdgcn_update_dpp(0, (int)hb.y, 0xFF, 0xF, 0xF, false));
                f32x4_t pd, ds;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ql = 16 * sq + 4 * g + r;
                    const int qa = qt * FA_BLK + ql;
                    float pr = 0.f;
                    if (qa < T && k_abs <= qa && k_abs < T) pr = __expf(s[r] - Ls[ql]);
                    float keep = 1.f;
                    if (p.p > 0.f) keep = rng_keep(hq[r], li & 3, thr) ? dsc : 0.f;
                    pd[r] = pr * keep;
                    ds[r] = pr * (dp[r] * keep - Ds[ql]);
                    Ss[ql * SLD + wv * 16 + li] = ds[r];
                }
                float ag[4][NC], aq[4][NC];
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        ag[r][c] = Gs[(16 * sq + 4 * g + r) * LD + 16 * c + li];
                        aq[r][c] = Qs[(16 * sq + 4 * g + r) * LD + 16 * c + li];
                    }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int r = 0; r < 4; ++r)
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        dv[c] = MFMA16(ag[r][c], pd[r], dv[c]);
                        dk[c] = MFMA16(aq[r][c], ds[r], dk[c]);
                    }
            }
            __syncthreads();                                  // dS of the pair is complete
            // ---- dQ[16 queries of qs][DH] over this wave's 64 keys: A = dS[q][key] (four consecutive keys per ds_read_b128), B = kq
            f32x4_t dq[NC];
#pragma unroll
            for (int c = 0; c < NC; ++c) dq[c] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 a4 = *reinterpret_cast<const float4*>(Ss + (16 * qs + li) * SLD + kh * 64 + 16 * j + 4 * g);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    dq[c] = MFMA16(a4.x, kq[4 * j + 0][c], dq[c]);
                    dq[c] = MFMA16(a4.y, kq[4 * j + 1][c], dq[c]);
                    dq[c] = MFMA16(a4.z, kq[4 * j + 2][c], dq[c]);
                    dq[c] = MFMA16(a4.w, kq[4 * j + 3][c], dq[c]);
                }
            }
            // dq[c][r] = dQ[query 16 qs + 4 g + r][dimension 16 c + li]
            if (kh == 1) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Rs[(qs * 16 + 4 * g + r) * DH + 16 * c + li] = dq[c][r];
            }
            __syncthreads();
            if (kh == 0) {
                const bool first = kb == 0;                    // key block 0 meets every query tile first: it initialises dQ
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int qa = qt * FA_BLK + 16 * qs + 4 * g + r;
                    if (qa >= T) continue;
                    float* dst = p.dq + (bt0 + qa) * d + hoff + li;
#pragma unroll
                    for (int c = 0; c < NC; ++c) {
                        const float v = dq[c][r] + Rs[(qs * 16 + 4 * g + r) * DH + 16 * c + li];
                        dst[16 * c] = first ? v : dst[16 * c] + v;
                    }
                }
            }
        }
        if (k_abs < T) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                *reinterpret_cast<float4*>(p.dv + (bt0 + k_abs) * d + hoff + 16 * c + 4 * g) = make_float4(dv[c][0], dv[c][1], dv[c][2], dv[c][3]);
                *reinterpret_cast<float4*>(p.dk + (bt0 + k_abs) * d + hoff + 16 * c + 4 * g) = make_float4(dk[c][0], dk[c][1], dk[c][2], dk[c][3]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------- launchers
template <int DH>
static int attn_launch_dh(const AttnArgs& a, int mode, hipStream_t st) {
    const int nblk = cdiv(a.T, FA_BLK);
    const dim3 grid(a.B * a.h * nblk), blk(256);
    if (mode == 0) {
        hipLaunchKernelGGL((attn_fwd_kernel<DH>), grid, blk, 0, st, a);
        OCRL_CHECK_LAUNCH("attn_fwd");
    } else {
        const long long BT = (long long)a.B * a.T;
        OCRL_REQUIRE(a.d % 4 == 0 && a.d <= 1024 && (a.d / 4) % a.h == 0, "attention: d must be a multiple of 4*h, <= 1024");
        hipLaunchKernelGGL(attn_delta_kernel, dim3(cdiv(BT, 256 / (a.d / 4))), dim3(256), 0, st, a.dO, a.o, a.delta, BT, a.T, a.d, a.h);
        OCRL_CHECK_LAUNCH("attn_delta");
        // OCRL_ATTN_BWD=1 selects the one-pass form (5 instead of 7 product-equivalents); default: the two-kernel form
        // (measured at B = 128, T = 1024, 4 heads of 48: 2.21 ms against 2.11 ms -- 222 registers leave one 8-wave workgroup per CU, and
        // what the two products save is lost to the lower occupancy; kept as a tested alternative, not the default)
        const char* fe = getenv("OCRL_ATTN_BWD");
        if (fe && atoi(fe) == 1) {
            hipLaunchKernelGGL((attn_bwd_fused_kernel<DH>), dim3(a.B * a.h), dim3(512), 0, st, a);
            OCRL_CHECK_LAUNCH("attn_bwd_fused");
        } else {
        hipLaunchKernelGGL((attn_bwd_kv_kernel<DH>), grid, blk, 0, st, a);
        OCRL_CHECK_LAUNCH("attn_bwd_kv");
        hipLaunchKernelGGL((attn_bwd_q_kernel<DH>), grid, blk, 0, st, a);
        OCRL_CHECK_LAUNCH("attn_bwd_q");
        }
    }
    return 0;
}

// mode 0: forward (q,k,v -> o, lse);  mode 1: backward (q,k,v,o,lse,dO -> dq,dk,dv; delta is scratch [B,h,T])
int attn_launch(const AttnArgs& a, int mode, hipStream_t st) {
    OCRL_REQUIRE(a.B > 0 && a.T > 0 && a.h > 0 && a.d % a.h == 0 && a.ld >= a.d && a.ld % 4 == 0, "attention: bad shape");
    OCRL_REQUIRE(a.q && a.k && a.v && a.o && a.lse, "attention: missing buffers");
    if (mode) OCRL_REQUIRE(a.dO && a.dq && a.dk && a.dv && a.delta, "attention backward: missing buffers");
    OCRL_REQUIRE((long long)a.T * a.T * a.B * a.h < (1ll << 62), "attention: too large");
    const int pi = prof_begin(mode ? PROF_ATTN_BWD : PROF_ATTN_FWD, st);
    int rc;
    switch (a.d / a.h) {
        case 16: rc = attn_launch_dh<16>(a, mode, st); break;
        case 32: rc = attn_launch_dh<32>(a, mode, st); break;
        case 48: rc = attn_launch_dh<48>(a, mode, st); break;
        case 64: rc = attn_launch_dh<64>(a, mode, st); break;
        default: ocrl_set_error("attention: head dim %d unsupported (16/32/48/64)", a.d / a.h); return 1;
    }
    prof_end(pi, st);
    return rc;
}
