// fp32 GEMM on v_mfma_f32_32x32x2_f32 (exact fp32, gfx950) with fused epilogues.
//
//   C[m,n] = epi( alpha * sum_k A(m,k) * B(k,n) )
//
// Operand storage is selected per operand:
//   AKC : A stored [M,K] (k contiguous, lda)        else A stored [K,M] (m contiguous, lda)
//   BKC : B stored [N,K] (k contiguous, ldb)        else B stored [K,N] (n contiguous, ldb)
// which covers  y = x W^T (AKC,BKC),  dx = dy W (AKC,!BKC),  dW = dy^T x (!AKC,!BKC).
//
// 256 threads = 4 waves in a 2x2 grid; each wave owns (BM/2)x(BN/2) as 32x32 MFMA tiles.
// K is consumed in 32-wide tiles, double-buffered in LDS with register prefetch (one barrier
// per k-tile).  Within a tile, the k index fed to MFMA step j by lane-half h is  8*c + 4*h + j
// for both operands, so a k-contiguous operand is one ds_read_b128 per 32 rows per 8 k.
// This header holds the kernel and its tile-shape dispatch; it is compiled three times (gemm_tt.hip, gemm_tn.hip, gemm_nn.hip -- one per
// operand-layout family) so that the ~110 instantiations build in parallel; gemm.hip keeps the argument checks and the split-k reduction.
#pragma once
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

#ifndef OCRL_GEMM_BK
#define OCRL_GEMM_BK 32      // k extent of a staged tile (development: -DOCRL_GEMM_BK=64 halves the barriers per k at twice the LDS)
#endif
#define BK OCRL_GEMM_BK

template <int BM, bool KC>
struct TileA {
    // KC: [BM][BK+4]   !KC: [BK][BM+4]
    static constexpr int LD = KC ? (BK + 4) : (BM + 4);
    static constexpr int ELEMS = KC ? BM * (BK + 4) : BK * (BM + 4);
    // float4 per thread per k-tile.  !KC: a k-row is BM/4 float4, 256/(BM/4) k-rows per pass (BM = 192: 5 rows, threads >= 240 idle)
    static constexpr int F4 = BM / 4, RPP = 256 / F4;
    static constexpr int KF4 = BK / 4, KRPP = 256 / KF4;       // KC: float4 per row, rows per pass
    static constexpr int NV = KC ? BM / KRPP : (BK + RPP - 1) / RPP;
};

// global -> registers for one operand tile.  rows = extent along m (or n), base points at
// element (row 0, k 0) of this block's tile.  rows_valid / k_valid bound the loads; anything
// outside is zero.
struct ADrop {            // A-operand dropout: element index = lrow * ld + lcol of the logical row-major tensor
    float p; unsigned site; unsigned long long seed; long long base; int ld;
};
__device__ inline float4 adrop_apply(float4 v, const ADrop& d, long long lrow, int lcol) {
    const uint64_t idx = (uint64_t)(d.base + lrow * d.ld + lcol);
    const uint2 bits = rng_bits4(d.seed, d.site, idx >> 2);
    const uint32_t thr = drop_thresh(d.p);
    const float sc = 1.0f / (1.0f - d.p);
    v.x = rng_keep(bits, 0, thr) ? v.x * sc : 0.f;
    v.y = rng_keep(bits, 1, thr) ? v.y * sc : 0.f;
    v.z = rng_keep(bits, 2, thr) ? v.z * sc : 0.f;
    v.w = rng_keep(bits, 3, thr) ? v.w * sc : 0.f;
    return v;
}

// mn0 / k0: logical coordinates of the tile origin (only used for the dropout index)
template <int BMN, bool KC, bool DROP>
__device__ inline void load_tile(const float* __restrict__ g, int ld, int rows_valid, int k_valid,
                                 float4 (&r)[TileA<BMN, KC>::NV], const ADrop& dr, int mn0, int k0) {
    const int t = threadIdx.x;
    if (KC) {
        constexpr int KF4 = TileA<BMN, KC>::KF4, KRPP = TileA<BMN, KC>::KRPP;
        const int c4 = t % KF4, r0 = t / KF4;
#pragma unroll
        for (int i = 0; i < TileA<BMN, KC>::NV; ++i) {
            const int row = r0 + KRPP * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < rows_valid && c4 * 4 < k_valid) {
                v = *reinterpret_cast<const float4*>(g + (size_t)row * ld + c4 * 4);
                if (DROP) v = adrop_apply(v, dr, mn0 + row, k0 + c4 * 4);          // logical [m][k]
            }
            r[i] = v;
        }
    } else {
        constexpr int F4 = BMN / 4;          // float4 per k-row
        constexpr int RPP = 256 / F4;        // k-rows per pass
        const int c4 = t % F4, r0 = t / F4;
#pragma unroll
        for (int i = 0; i < TileA<BMN, KC>::NV; ++i) {
            const int kr = r0 + RPP * i;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r0 < RPP && kr < BK && kr < k_valid && c4 * 4 < rows_valid) {
                v = *reinterpret_cast<const float4*>(g + (size_t)kr * ld + c4 * 4);
                if (DROP) v = adrop_apply(v, dr, k0 + kr, mn0 + c4 * 4);           // stored [k][m]: logical row = k
            }
            r[i] = v;
        }
    }
}

template <int BMN, bool KC>
__device__ inline void store_tile(float* __restrict__ s, const float4 (&r)[TileA<BMN, KC>::NV]) {
    const int t = threadIdx.x;
    constexpr int LD = TileA<BMN, KC>::LD;
    if (KC) {
        constexpr int KF4 = TileA<BMN, KC>::KF4, KRPP = TileA<BMN, KC>::KRPP;
        const int c4 = t % KF4, r0 = t / KF4;
#pragma unroll
        for (int i = 0; i < TileA<BMN, KC>::NV; ++i) *reinterpret_cast<float4*>(s + (r0 + KRPP * i) * LD + c4 * 4) = r[i];
    } else {
        constexpr int F4 = BMN / 4;
        constexpr int RPP = 256 / F4;
        const int c4 = t % F4, r0 = t / F4;
#pragma unroll
        for (int i = 0; i < TileA<BMN, KC>::NV; ++i)
            if (r0 < RPP && r0 + RPP * i < BK) *reinterpret_cast<float4*>(s + (r0 + RPP * i) * LD + c4 * 4) = r[i];
    }
}

// ---- soft-max operand transforms (GemmArgs::a_mode / b_mode): the stored scores become probabilities / cross-entropy gradients while
// the tile goes from registers to LDS.  The per-row constants are fetched together with the tile (load_aux) and consumed a whole
// MFMA phase later (xform_tile), so no load is waited on early.  Padding elements get lse = +inf -> exp(0 - inf) = 0.
#define TINYF_G 1.17549435e-38f
template <int BMN, bool KC>
__device__ inline void load_aux(const float* __restrict__ lse, const int* __restrict__ tok, int rows_valid, int k_valid, int mn0, int k0,
                                float (&al)[TileA<BMN, KC>::NV], int (&at)[TileA<BMN, KC>::NV]) {
    const int t = threadIdx.x;
#pragma unroll
    for (int i = 0; i < TileA<BMN, KC>::NV; ++i) {
        int trow;
        bool ok;
        if (KC) {
            const int c4 = t % TileA<BMN, KC>::KF4, row = t / TileA<BMN, KC>::KF4 + TileA<BMN, KC>::KRPP * i;
            ok = row < rows_valid && c4 * 4 < k_valid;
            trow = mn0 + row;
        } else {
            constexpr int F4 = BMN / 4, RPP = 256 / F4;
            const int c4 = t % F4, r0 = t / F4, kr = r0 + RPP * i;
            ok = r0 < RPP && kr < BK && kr < k_valid && c4 * 4 < rows_valid;
            trow = k0 + kr;
        }
        al[i] = ok ? lse[trow] : INFINITY;
        at[i] = (ok && tok) ? tok[trow] : -0x40000000;
    }
}
template <int BMN, bool KC>
__device__ inline void xform_tile(int mode, float scale, int mn0, int k0, float4 (&r)[TileA<BMN, KC>::NV],
                                  const float (&al)[TileA<BMN, KC>::NV], const int (&at)[TileA<BMN, KC>::NV]) {
    const int t = threadIdx.x;
    const int c4 = KC ? t % TileA<BMN, KC>::KF4 : t % (BMN / 4);
    const int col0 = (KC ? k0 : mn0) + c4 * 4;          // vocabulary index of .x
#pragma unroll
    for (int i = 0; i < TileA<BMN, KC>::NV; ++i) {
        float4 v = r[i];
        const float l = al[i];
        v.x = __expf(v.x - l); v.y = __expf(v.y - l); v.z = __expf(v.z - l); v.w = __expf(v.w - l);
        if (mode == 3) {
            const int tk = at[i] - col0;
            v.x = (v.x - (tk == 0 ? 1.f : 0.f)) * scale; v.y = (v.y - (tk == 1 ? 1.f : 0.f)) * scale;
            v.z = (v.z - (tk == 2 ? 1.f : 0.f)) * scale; v.w = (v.w - (tk == 3 ? 1.f : 0.f)) * scale;
        }
        r[i] = v;
    }
}

// fragment for one 32-row MFMA tile, 8-k chunk c: f[j] is the operand of MFMA step j.
template <int BMN, bool KC>
__device__ inline void load_frag(const float* __restrict__ s, int row0, int c, float (&f)[4]) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 31, h = lane >> 5;
    constexpr int LD = TileA<BMN, KC>::LD;
    if (KC) {
        const float4 v = *reinterpret_cast<const float4*>(s + (row0 + i) * LD + c * 8 + 4 * h);
        f[0] = v.x; f[1] = v.y; f[2] = v.z; f[3] = v.w;
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = s[(c * 8 + 4 * h + j) * LD + row0 + i];
    }
}

// XF: 0 plain operands, 1 A-operand dropout, 2 soft-max operand transforms (a_mode / b_mode);  EPI: GemmArgs::epi_mode (0 = standard epilogue)
template <int BM, int BN, bool AKC, bool BKC, bool SB, int XF, int EPI>
__global__ __launch_bounds__(256, (BM * BN == 128 * 128) ? 3 : ((BM * BN == 128 * 64) ? 4 : 1)) void gemm_kernel(GemmArgs p) {
    constexpr bool ADROP = XF == 1;
    constexpr int TM = BM / 64, TN = BN / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int AE = TileA<BM, AKC>::ELEMS, BE = TileA<BN, BKC>::ELEMS;
    float* const As0 = smem;
    float* const Bs0 = smem + (SB ? 1 : 2) * AE;

    // block -> (tile, z) with an XCD-aware bijective remap of the tile index
    const int nbm = (p.M + BM - 1) / BM, nbn = (p.N + BN - 1) / BN;
    const int nwg = nbm * nbn;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, x = bid & 7, y = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int bn = bid % nbn, bm = bid / nbn;
    const int z = blockIdx.y;
    const int batch = z / p.splitk, split = z % p.splitk;
    const int bo = batch / p.batch_inner, bi = batch % p.batch_inner;
    const int m0 = bm * BM, n0 = bn * BN;

    const int nkt = (p.K + BK - 1) / BK;
    const int kt_per = (nkt + p.splitk - 1) / p.splitk;
    const int kt0 = split * kt_per;
    const int kt1 = min(nkt, kt0 + kt_per);

    const float* A = p.A + (size_t)bo * p.sA + (size_t)bi * p.sAi;
    const float* B = p.B + (size_t)bo * p.sB + (size_t)bi * p.sBi;
    // tile base pointers at k = 0
    const float* Ag = AKC ? A + (size_t)m0 * p.lda : A + m0;
    const float* Bg = BKC ? B + (size_t)n0 * p.ldb : B + n0;
    const int mval = p.M - m0, nval = p.N - n0;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int wm0 = (wave >> 1) * (BM / 2), wn0 = (wave & 1) * (BN / 2);

    ADrop adr;
    adr.p = p.adrop_p; adr.site = p.adrop_site; adr.seed = p.drop_seed; adr.ld = p.adrop_ld; adr.base = 0;
    // fused bias gradient (dW form): column sums of the staged A tile, taken by the first BM threads of the bn == 0 blocks
    const bool do_bias = !AKC && p.bias_out != nullptr && bn == 0;
    float bsum = 0.f;
    auto bias_acc = [&](const float* as) {
        if (!AKC && do_bias && (int)threadIdx.x < BM) {
            constexpr int LD = TileA<BM, AKC>::LD;
#pragma unroll 8
            for (int k = 0; k < BK; ++k) bsum += as[k * LD + threadIdx.x];
        }
    };
    float4 ra[TileA<BM, AKC>::NV], rb[TileA<BN, BKC>::NV];
    // soft-max operand transforms: per-row constants of the tile held in ra / rb
    float ala[XF == 2 ? TileA<BM, AKC>::NV : 1], alb[XF == 2 ? TileA<BN, BKC>::NV : 1];
    int ata[XF == 2 ? TileA<BM, AKC>::NV : 1], atb[XF == 2 ? TileA<BN, BKC>::NV : 1];
    auto aux_load = [&](int k0) {
        if constexpr (XF == 2) {
            if (p.a_mode) load_aux<BM, AKC>(p.x_lse, p.a_mode == 3 ? p.x_tok : nullptr, mval, p.K - k0, m0, k0, ala, ata);
            if (p.b_mode) load_aux<BN, BKC>(p.x_lse, nullptr, nval, p.K - k0, n0, k0, alb, atb);
        }
    };
    auto aux_apply = [&](int k0) {
        if constexpr (XF == 2) {
            if (p.a_mode) xform_tile<BM, AKC>(p.a_mode, p.x_scale, m0, k0, ra, ala, ata);
            if (p.b_mode) xform_tile<BN, BKC>(p.b_mode, 1.f, n0, k0, rb, alb, atb);
        }
    };
    if (SB) {
        // single LDS buffer (half the LDS -> twice the resident workgroups): next tile's global loads fly during compute
        if (kt0 < kt1) {
            const int k0 = kt0 * BK;
            load_tile<BM, AKC, ADROP>(AKC ? Ag + k0 : Ag + (size_t)k0 * p.lda, p.lda, mval, p.K - k0, ra, adr, m0, k0);
            load_tile<BN, BKC, false>(BKC ? Bg + k0 : Bg + (size_t)k0 * p.ldb, p.ldb, nval, p.K - k0, rb, adr, n0, k0);
            aux_load(k0);
        }
        for (int kt = kt0; kt < kt1; ++kt) {
            __syncthreads();
            aux_apply(kt * BK);
            store_tile<BM, AKC>(As0, ra);
            store_tile<BN, BKC>(Bs0, rb);
            __syncthreads();
            if (kt + 1 < kt1) {
                const int k0 = (kt + 1) * BK;
                load_tile<BM, AKC, ADROP>(AKC ? Ag + k0 : Ag + (size_t)k0 * p.lda, p.lda, mval, p.K - k0, ra, adr, m0, k0);
                load_tile<BN, BKC, false>(BKC ? Bg + k0 : Bg + (size_t)k0 * p.ldb, p.ldb, nval, p.K - k0, rb, adr, n0, k0);
                aux_load(k0);
            }
            bias_acc(As0);
#pragma unroll
            for (int c = 0; c < BK / 8; ++c) {
                float fa[TM][4], fb[TN][4];
#pragma unroll
                for (int i = 0; i < TM; ++i) load_frag<BM, AKC>(As0, wm0 + i * 32, c, fa[i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) load_frag<BN, BKC>(Bs0, wn0 + j * 32, c, fb[j]);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
            }
        }
    } else {
    if (kt0 < kt1) {
        const int k0 = kt0 * BK;
        load_tile<BM, AKC, ADROP>(AKC ? Ag + k0 : Ag + (size_t)k0 * p.lda, p.lda, mval, p.K - k0, ra, adr, m0, k0);
        load_tile<BN, BKC, false>(BKC ? Bg + k0 : Bg + (size_t)k0 * p.ldb, p.ldb, nval, p.K - k0, rb, adr, n0, k0);
        aux_load(k0);
        aux_apply(k0);
        store_tile<BM, AKC>(As0, ra);
        store_tile<BN, BKC>(Bs0, rb);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int cur = (kt - kt0) & 1;
        if (kt + 1 < kt1) {
            const int k0 = (kt + 1) * BK;
            load_tile<BM, AKC, ADROP>(AKC ? Ag + k0 : Ag + (size_t)k0 * p.lda, p.lda, mval, p.K - k0, ra, adr, m0, k0);
            load_tile<BN, BKC, false>(BKC ? Bg + k0 : Bg + (size_t)k0 * p.ldb, p.ldb, nval, p.K - k0, rb, adr, n0, k0);
            aux_load(k0);
        }
        const float* as = As0 + cur * AE;
        const float* bs = Bs0 + cur * BE;
        bias_acc(as);
#pragma unroll
        for (int c = 0; c < BK / 8; ++c) {
            float fa[TM][4], fb[TN][4];
#pragma unroll
            for (int i = 0; i < TM; ++i) load_frag<BM, AKC>(as, wm0 + i * 32, c, fa[i]);
#pragma unroll
            for (int j = 0; j < TN; ++j) load_frag<BN, BKC>(bs, wn0 + j * 32, c, fb[j]);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][s], fb[j][s], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < kt1) {
            aux_apply((kt + 1) * BK);
            store_tile<BM, AKC>(As0 + (cur ^ 1) * AE, ra);
            store_tile<BN, BKC>(Bs0 + (cur ^ 1) * BE, rb);
        }
        __syncthreads();
    }
    }

    // ---- epilogue
    const bool partial = p.splitk > 1;
    if (!AKC && do_bias && (int)threadIdx.x < BM && m0 + (int)threadIdx.x < p.M)
        p.bias_out[(partial ? (size_t)split * p.sBias : 0) + m0 + threadIdx.x] = bsum;
    float* C = p.C + (size_t)bo * p.sC + (size_t)bi * p.sCi + (partial ? (size_t)split * p.sCsplit : 0);
    const float* R = p.resid ? p.resid + (size_t)batch * p.sR : nullptr;
    const float* Mk = p.mask ? p.mask + (size_t)batch * p.sMask : nullptr;
    const uint32_t thr = drop_thresh(p.drop_p);
    const float dscale = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.0f;
    if constexpr (EPI != 0) {
        // ---- soft-max head epilogues (GemmArgs::epi_mode).  Same patch walk as the fast path below: in a patch pass the 8 lanes with
        // equal (lane >> 3) hold the 32 columns of one row, so a row reduction is three xor-shuffles; the TN patches of a wave row are
        // merged online, giving one (max, sum-exp) pair per row and 64-column segment.  All reductions have a fixed order.
        __syncthreads();
        float* patch = smem + wave * (32 * 36);
        const int rr0 = lane >> 3, c4 = lane & 7;
        const int nseg = 2 * nbn, seg = 2 * bn + (wave & 1);
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float sm[4], ss[4], hb[4], hs[4], rl[4], rv[4];
            int hi[4];
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                sm[ps] = -INFINITY; ss[ps] = 0.f; hb[ps] = -INFINITY; hs[ps] = 0.f; hi[ps] = 0; rl[ps] = 0.f; rv[ps] = 0.f;
                const int row = m0 + wm0 + i * 32 + ps * 8 + rr0;
                if (EPI == 3 && row < p.M) { rl[ps] = p.e_lse[row]; rv[ps] = p.e_rowvec[row]; }
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 36 + (lane & 31)] = acc[i][j][r] * p.alpha;
                __builtin_amdgcn_wave_barrier();
                const int col = n0 + wn0 + j * 32 + c4 * 4;
                const bool cok = col < p.N;
                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (p.bias && cok) bv = *reinterpret_cast<const float4*>(p.bias + col);
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int rr = ps * 8 + rr0;
                    const int row = m0 + wm0 + i * 32 + rr;
                    const bool ok = cok && row < p.M;
                    float4 v = *reinterpret_cast<const float4*>(patch + rr * 36 + c4 * 4);
                    if (EPI == 3) {
                        if (ok) {
                            const float4 y = *reinterpret_cast<const float4*>(p.mask + (size_t)row * p.ldmask + col);
                            const float l = rl[ps], dv = rv[ps];
                            v.x = __expf(y.x - l) * (v.x - dv) * p.e_scale; v.y = __expf(y.y - l) * (v.y - dv) * p.e_scale;
                            v.z = __expf(y.z - l) * (v.z - dv) * p.e_scale; v.w = __expf(y.w - l) * (v.w - dv) * p.e_scale;
                            *reinterpret_cast<float4*>(C + (size_t)row * p.ldc + col) = v;
                        }
                        continue;
                    }
                    v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                    if (EPI == 2 && p.e1) {
                        // injected Exp(1) noise (parity runs): both Gumbel samples exactly as the reference forms them,
                        // g = -log(E + tiny); the hard sample is the first maximum of l + g2 (the factor 1/tau > 0 does not move it)
                        float4 la = make_float4(0.f, 0.f, 0.f, 0.f), lb = la;
                        if (ok) {
                            const uint64_t idx = (uint64_t)row * (uint64_t)p.N + col;
                            const float4 ea = *reinterpret_cast<const float4*>(p.e1 + idx), eb = *reinterpret_cast<const float4*>(p.e2 + idx);
                            la = make_float4(__logf(ea.x + TINYF_G), __logf(ea.y + TINYF_G), __logf(ea.z + TINYF_G), __logf(ea.w + TINYF_G));
                            lb = make_float4(__logf(eb.x + TINYF_G), __logf(eb.y + TINYF_G), __logf(eb.z + TINYF_G), __logf(eb.w + TINYF_G));
                        }
                        const float h0 = v.x - lb.x, h1 = v.y - lb.y, h2 = v.z - lb.z, h3 = v.w - lb.w;
                        float b = h0; int bi = col;
                        if (h1 > b) { b = h1; bi = col + 1; }
                        if (h2 > b) { b = h2; bi = col + 2; }
                        if (h3 > b) { b = h3; bi = col + 3; }
                        if (!cok) b = -INFINITY;
#pragma unroll
                        for (int o = 1; o < 8; o <<= 1) {
                            const float ob = __shfl_xor(b, o, 64);
                            const int oi = __shfl_xor(bi, o, 64);
                            if (ob > b || (ob == b && oi < bi)) { b = ob; bi = oi; }
                        }
                        if (b > hb[ps]) { hb[ps] = b; hi[ps] = bi; }          // later patches hold higher columns: only a strictly larger value wins
                        v.x = (v.x - la.x) * p.e_scale; v.y = (v.y - la.y) * p.e_scale; v.z = (v.z - la.z) * p.e_scale; v.w = (v.w - la.w) * p.e_scale;
                    } else if (EPI == 2) {
                        // device RNG.  The hard sample argmax(l + g2) is a draw from Categorical(soft-max(l)); it is taken by inverse CDF in
                        // softmax_stat_combine_kernel (two uniforms per ROW) from the per-segment (max l, sum exp(l - max)) pairs gathered here,
                        // instead of a second Gumbel draw per vocabulary entry (a hash, two logarithms and a compare per element less)
                        {
                            float hm = cok ? fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)) : -INFINITY;
#pragma unroll
                            for (int o = 1; o < 8; o <<= 1) hm = fmaxf(hm, __shfl_xor(hm, o, 64));
                            float he = cok ? (__expf(v.x - hm) + __expf(v.y - hm)) + (__expf(v.z - hm) + __expf(v.w - hm)) : 0.f;
#pragma unroll
                            for (int o = 1; o < 8; o <<= 1) he += __shfl_xor(he, o, 64);
                            if (hm > -INFINITY) {
                                const float mn = fmaxf(hb[ps], hm);
                                hs[ps] = hs[ps] * __expf(hb[ps] - mn) + he * __expf(hm - mn);
                                hb[ps] = mn;
                            }
                        }
                        // soft sample scores (l + g1) / tau, g1 = -log(E + tiny), E = -log(u): the draws of gumbel_softmax_kernel (element
                        // index as the counter), the key taken once per four elements.  The + tiny guard is live: the largest 24-bit
                        // uniform rounds to 1.0f, E = -log(1) = 0 (about 64 of the 10^9 draws of a step at B = 128)
                        float4 la = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (ok) {
                            const uint64_t idx = (uint64_t)row * (uint64_t)p.N + col;
                            const uint32_t key = rng_key(p.e_seed, SITE_GUMBEL_Z, (uint32_t)(idx >> 32)), lo = (uint32_t)idx;
                            la = make_float4(__logf(TINYF_G - __logf(u01_24(rng_bits1_keyed(key, lo)))), __logf(TINYF_G - __logf(u01_24(rng_bits1_keyed(key, lo + 1)))),
                                             __logf(TINYF_G - __logf(u01_24(rng_bits1_keyed(key, lo + 2)))), __logf(TINYF_G - __logf(u01_24(rng_bits1_keyed(key, lo + 3)))));
                        }
                        v.x = (v.x - la.x) * p.e_scale; v.y = (v.y - la.y) * p.e_scale; v.z = (v.z - la.z) * p.e_scale; v.w = (v.w - la.w) * p.e_scale;
                    }
                    float lm = cok ? fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w)) : -INFINITY;
#pragma unroll
                    for (int o = 1; o < 8; o <<= 1) lm = fmaxf(lm, __shfl_xor(lm, o, 64));
                    float le = cok ? (__expf(v.x - lm) + __expf(v.y - lm)) + (__expf(v.z - lm) + __expf(v.w - lm)) : 0.f;
#pragma unroll
                    for (int o = 1; o < 8; o <<= 1) le += __shfl_xor(le, o, 64);
                    if (lm > -INFINITY) {
                        const float mn = fmaxf(sm[ps], lm);
                        ss[ps] = ss[ps] * __expf(sm[ps] - mn) + le * __expf(lm - mn);
                        sm[ps] = mn;
                    }
                    if (ok) *reinterpret_cast<float4*>(C + (size_t)row * p.ldc + col) = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (EPI != 3 && c4 == 0) {
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int row = m0 + wm0 + i * 32 + ps * 8 + rr0;
                    if (row >= p.M) continue;
                    const size_t si = (size_t)row * nseg + seg;
                    p.stat[si * 2] = sm[ps];
                    p.stat[si * 2 + 1] = ss[ps];
                    if (EPI == 2) { p.hstat[si] = hb[ps]; p.hidx[si] = p.e1 ? hi[ps] : __builtin_bit_cast(int, hs[ps]); }
                }
            }
        }
        return;
    }
    // Fast path: every 32x32 accumulator tile goes through a wave-private LDS patch and leaves as float4 rows — 4x fewer store
    // (and mask / residual load) instructions, full 128-byte lines, and one dropout draw per 4 outputs instead of one per output.
    const bool vec = (p.N & 3) == 0 && (p.ldc & 3) == 0 && (((uintptr_t)C) & 15) == 0 && (!R || ((p.ldr & 3) == 0 && (((uintptr_t)R) & 15) == 0)) &&
                     (!Mk || ((p.ldmask & 3) == 0 && (((uintptr_t)Mk) & 15) == 0)) && (!p.bias || (((uintptr_t)p.bias) & 15) == 0);
    if (vec) {
        __syncthreads();                                   // the operand tiles are dead: their LDS becomes four staging patches
        float* patch = smem + wave * (32 * 36);
        const int rr0 = lane >> 3, c4 = lane & 7;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) {
#pragma unroll
                for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 36 + (lane & 31)] = acc[i][j][r] * p.alpha;
                __builtin_amdgcn_wave_barrier();
                const int col = n0 + wn0 + j * 32 + c4 * 4;
#pragma unroll
                for (int ps = 0; ps < 4; ++ps) {
                    const int rr = ps * 8 + rr0;
                    const int row = m0 + wm0 + i * 32 + rr;
                    float4 v = *reinterpret_cast<const float4*>(patch + rr * 36 + c4 * 4);
                    if (row >= p.M || col >= p.N) continue;
                    if (!partial) {
                        if (p.bias) {
                            const float4 bv = *reinterpret_cast<const float4*>(p.bias + col);
                            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
                        }
                        if (p.relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                        else if (p.relu == 2) {
                            v.x = v.x > 0.f ? v.x : __expf(v.x) - 1.f; v.y = v.y > 0.f ? v.y : __expf(v.y) - 1.f;
                            v.z = v.z > 0.f ? v.z : __expf(v.z) - 1.f; v.w = v.w > 0.f ? v.w : __expf(v.w) - 1.f;
                        }
                        if (p.drop_p > 0.f) {
                            const uint64_t idx = ((uint64_t)batch * p.M + row) * (uint64_t)p.N + col;      // multiple of 4
                            const uint2 bits = rng_bits4(p.drop_seed, p.drop_site, idx >> 2);
                            v.x = rng_keep(bits, 0, thr) ? v.x * dscale : 0.f; v.y = rng_keep(bits, 1, thr) ? v.y * dscale : 0.f;
                            v.z = rng_keep(bits, 2, thr) ? v.z * dscale : 0.f; v.w = rng_keep(bits, 3, thr) ? v.w * dscale : 0.f;
                        }
                        if (Mk) {
                            const float4 mk = *reinterpret_cast<const float4*>(Mk + (size_t)row * p.ldmask + col);
                            const float e = p.mask_elu ? 1.f : 0.f;
                            v.x = mk.x > 0.f ? v.x : e * v.x * (mk.x + 1.f); v.y = mk.y > 0.f ? v.y : e * v.y * (mk.y + 1.f);
                            v.z = mk.z > 0.f ? v.z : e * v.z * (mk.z + 1.f); v.w = mk.w > 0.f ? v.w : e * v.w * (mk.w + 1.f);
                        }
                        if (R) {
                            const float4 rv = *reinterpret_cast<const float4*>(R + (size_t)row * p.ldr + col);
                            v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
                        }
                    }
                    *reinterpret_cast<float4*>(C + (size_t)row * p.ldc + col) = v;
                }
                __builtin_amdgcn_wave_barrier();
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn0 + j * 32 + (lane & 31);
            if (col >= p.N) continue;
            const float bv = (!partial && p.bias) ? p.bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (row >= p.M) continue;
                float v = acc[i][j][r] * p.alpha;
                if (!partial) {
                    v += bv;
                    if (p.relu == 1) v = fmaxf(v, 0.f);
                    else if (p.relu == 2) v = v > 0.f ? v : __expf(v) - 1.f;      // ELU
                    if (p.drop_p > 0.f) {
                        const uint64_t idx = ((uint64_t)batch * p.M + row) * (uint64_t)p.N + col;
                        const uint2 bits = rng_bits4(p.drop_seed, p.drop_site, idx >> 2);
                        v = rng_keep(bits, (int)(idx & 3), thr) ? v * dscale : 0.f;
                    }
                    if (Mk) {
                        const float mk = Mk[(size_t)row * p.ldmask + col];
                        v = mk > 0.f ? v : (p.mask_elu ? v * (mk + 1.f) : 0.f);     // ReLU' / ELU' of the saved activation
                    }
                    if (R) v += R[(size_t)row * p.ldr + col];
                }
                C[(size_t)row * p.ldc + col] = v;
            }
        }
}

static int sb_mode() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("OCRL_GEMM_SB"); v = e ? atoi(e) : 0; }
    return v;
}

template <int BM, int BN, bool AKC, bool BKC, bool SB, int XF, int EPI = 0>
static int launch_cfg3(const GemmArgs& a, hipStream_t st) {
    constexpr int smem = ((SB ? 1 : 2) * TileA<BM, AKC>::ELEMS + (SB ? 1 : 2) * TileA<BN, BKC>::ELEMS) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        OCRL_HIP(hipFuncSetAttribute((const void*)gemm_kernel<BM, BN, AKC, BKC, SB, XF, EPI>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    dim3 grid(cdiv(a.M, BM) * cdiv(a.N, BN), a.batch * a.splitk);
    const int pi = prof_begin(PROF_GEMM, st);
    hipLaunchKernelGGL((gemm_kernel<BM, BN, AKC, BKC, SB, XF, EPI>), grid, dim3(256), smem, st, a);
    prof_end(pi, st);
    OCRL_CHECK_LAUNCH("gemm_kernel");
    return 0;
}
template <int BM, int BN, bool AKC, bool BKC, bool SB>
static int launch_cfg2(const GemmArgs& a, hipStream_t st) {
    if (a.a_mode || a.b_mode) {
        if constexpr (AKC || !BKC)
            return launch_cfg3<BM, BN, AKC, BKC, SB, 2>(a, st);
        else
            OCRL_REQUIRE(false, "gemm: operand transform not built for tile %dx%d akc=%d bkc=%d", BM, BN, (int)AKC, (int)BKC);
    }
    if (a.adrop_p > 0.f) return launch_cfg3<BM, BN, AKC, BKC, SB, 1>(a, st);
    return launch_cfg3<BM, BN, AKC, BKC, SB, 0>(a, st);
}
template <int BM, int BN, bool AKC, bool BKC>
static int launch_cfg(const GemmArgs& a, hipStream_t st) {
    // measured (tools/bench_gemm.py): a single LDS buffer (twice the resident workgroups) wins for the short-K
    // forward / dX forms (+8..37 %); the long split-K weight-gradient loops keep the double buffer.
    const int mode = sb_mode();           // OCRL_GEMM_SB: 0 = rule above, 1 = always single, 2 = always double
    const bool sb = mode == 1 || (mode == 0 && AKC);
    if (sb) return launch_cfg2<BM, BN, AKC, BKC, true>(a, st);
    return launch_cfg2<BM, BN, AKC, BKC, false>(a, st);
}

// development override: OCRL_GEMM_TILE=BMxBN (e.g. 128x64) forces one tile shape
static int tile_override() {
    static int v = -1;
    if (v < 0) {
        const char* e = getenv("OCRL_GEMM_TILE");
        v = 0;
        if (e) { int bm = 0, bn = 0; if (sscanf(e, "%dx%d", &bm, &bn) == 2) v = bm * 1000 + bn; }
    }
    return v;
}

template <bool AKC, bool BKC>
static int launch_tr(const GemmArgs& a, hipStream_t st) {
    switch (tile_override()) {
        case 128128: return launch_cfg<128, 128, AKC, BKC>(a, st);
        case 128064: return launch_cfg<128, 64, AKC, BKC>(a, st);
        case 64128: return launch_cfg<64, 128, AKC, BKC>(a, st);
        case 64064: return launch_cfg<64, 64, AKC, BKC>(a, st);
        case 128192: if (AKC || !BKC) return launch_cfg<128, 192, AKC, BKC>(a, st); break;
        default: break;
    }
    // measured on MI355X (tools/bench_gemm.py): 128-wide column tiles only pay when N is a multiple of 128;
    // N = 192 / 64 (projections, weight gradients with 192 inputs) run 10-20 % faster on 128x64 tiles
    // long-K activation x weight products with N = 192 (the model width), e.g. the vocabulary-head dX: one 128x192 tile reads A
    // once and moves 38 FLOP per staged byte instead of 21 (+7 % measured at K = 4096; short K is faster on 128x64)
    if (AKC && a.N == 192 && a.K >= 1024 && a.M >= 4096) return launch_cfg<128, 192, AKC, BKC>(a, st);
    // weight gradients with 192 input features (dW = dY^T X over >= 10^5 rows, split-K): a 128x192 tile reads X once per split
    // (PMC: the 128x64 tiling moved 643 MB per launch for 201 MB of operands); single LDS buffer to keep two workgroups per CU
    if (!AKC && !BKC && a.N == 192 && a.K >= 4096) return launch_cfg2<128, 192, AKC, BKC, true>(a, st);
    const bool wide = (a.N % 128 == 0);
    if (wide) {
        if (a.M > 64) return launch_cfg<128, 128, AKC, BKC>(a, st);
        return launch_cfg<64, 128, AKC, BKC>(a, st);
    }
    if (a.M > 64) return launch_cfg<128, 64, AKC, BKC>(a, st);
    return launch_cfg<64, 64, AKC, BKC>(a, st);
}

