// Stride-1 "same" KSxKS convolution, NHWC, fp32 on v_mfma_f32_32x32x2_f32 (gfx950).
//
// conv_fwd_kernel  — implicit GEMM with the input halo tile resident in LDS:
//     M = 4x32 output pixels per workgroup (one image row segment per wave), N = 64 output
//     channels, K = KS*KS*CIN walked tap by tap over the SAME LDS tile (input read once per
//     tile instead of KS*KS times).  Weights come pre-packed as [tap][CIN/8][COUT][8] so a
//     wave's B fragment is one fully coalesced 1 KiB global read (L1/L2 resident, 400 KB total).
//     Also serves as the backward-data kernel: feed dY and the flipped/transposed pack, and
//     give the previous layer's output as `mask` to fuse the ReLU backward.
// conv_wgrad_kernel — dW[tap][co][ci] = sum_pixels dY[p][co] * X[p+tap][ci]: pixels are the
//     MFMA k dimension; grid = (chunks, KS) so one workgroup owns one kernel row (KS taps) and
//     keeps KS 32x32 accumulators per wave across all its tiles; partial slabs are reduced (and
//     permuted to the reference [co][ci][ky][kx] layout) by conv_wgrad_reduce_kernel.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

#define TH 4
#define TW 32

// 3x3 / 64 channels: the tap loop is 2.8x shorter than the 5x5 one, so the halo load and the epilogue weigh more; its halo rows are
// stored unpadded (64 floats) with the 16-byte chunk index XOR-ed by the pixel index (same bank spread as the +4 padding) so that
// the tile needs 51 KB instead of 54.2 KB of LDS and THREE workgroups fit a CU to cover each other's load / store phases.
template <int KS, int CIN> struct ConvCfg { static constexpr bool SWZ = (KS == 3 && CIN == 64); static constexpr int LDH = SWZ ? CIN : CIN + 4, WGS = SWZ ? 3 : 2; };

template <int KS, int CIN, int COUT>
__global__ __launch_bounds__(256, (ConvCfg<KS, CIN>::WGS)) void conv_fwd_kernel(ConvArgs p) {
    static_assert(COUT == 64, "COUT must be 64");
    static_assert(CIN % 8 == 0, "CIN must be a multiple of 8");
    constexpr int P = KS / 2;
    constexpr int HW_ = TW + KS - 1, HH_ = TH + KS - 1;
    constexpr int LDH = ConvCfg<KS, CIN>::LDH;
    constexpr bool SWZ = ConvCfg<KS, CIN>::SWZ;
    constexpr int NCH = CIN / 8;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    int bid = blockIdx.x;
    {
        // workgroups are dealt round-robin to the 8 XCDs: give each XCD a contiguous run of tiles (whole images at the usual sizes) so
        // that the halo rows two vertically adjacent tiles share are served by that XCD's L2 instead of being fetched twice
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7, y = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; bid /= tiles_y;
    const int b = bid;
    const int x0 = tx * TW, y0 = ty * TH;

    // ---- halo tile -> LDS (zero outside the image): all global loads are issued before the first LDS store
    {
        constexpr int F4 = CIN / 4;
        constexpr int TOTAL = HH_ * HW_ * F4;
        constexpr int NLD = (TOTAL + 255) / 256;
        const float* Xb = p.X + (size_t)b * p.H * p.W * CIN;
        float4 hv[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = threadIdx.x + i * 256;
            const int c4 = idx % F4, hp = idx / F4;
            const int hx = hp % HW_, hy = hp / HW_;
            const int y = y0 - P + hy, x = x0 - P + hx;
            hv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < TOTAL && y >= 0 && y < p.H && x >= 0 && x < p.W)
                hv[i] = *reinterpret_cast<const float4*>(Xb + ((size_t)y * p.W + x) * CIN + c4 * 4);
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = threadIdx.x + i * 256;
            if (idx < TOTAL) *reinterpret_cast<float4*>(smem + (idx / F4) * LDH + (SWZ ? (((idx % F4) ^ ((idx / F4) & 15)) * 4) : (idx % F4) * 4)) = hv[i];
        }
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    const float* wl = p.Wp + li * 8 + 4 * lh;   // lane's slice inside one [COUT][8] block
    constexpr int NIT = KS * KS * NCH;
    if constexpr (NCH >= 4) {
        // Operand ring of one tap (NCH slots): slot cc holds A (LDS) and B (packed weights, L2) of k-chunk cc; right after
        // its 8 MFMAs are issued the slot is refilled with the NEXT tap's chunk cc, i.e. NCH-1 chunks (>= 1.5k MFMA
        // cycles) ahead of its use, so neither the L2 nor the LDS latency is exposed.
        float4 ra[NCH], rb0[NCH], rb1[NCH];
        {
            const int prow = wave * HW_ + li;
            const float* arow = smem + prow * LDH + (SWZ ? 0 : 4 * lh);
            const int sv = (4 * lh) ^ ((prow & 15) * 4);            // swizzled layout: float offset of chunk (2cc + lh) = (8cc) ^ sv
#pragma unroll
            for (int cc = 0; cc < NCH; ++cc) {
                rb0[cc] = *reinterpret_cast<const float4*>(wl + (size_t)cc * COUT * 8);
                rb1[cc] = *reinterpret_cast<const float4*>(wl + (size_t)cc * COUT * 8 + 32 * 8);
                ra[cc] = *reinterpret_cast<const float4*>(arow + (SWZ ? ((cc * 8) ^ sv) : cc * 8));
                __builtin_amdgcn_sched_barrier(0);      // keep the ring's issue order: the loop's s_waitcnt counts rely on it
            }
        }
#pragma unroll 1
        for (int tap = 0; tap < KS * KS; ++tap) {
            const int tn = tap + 1;
            const bool more = tn < KS * KS;
            const int ky = tn / KS, kx = tn - ky * KS;
            const int prow = (wave + (more ? ky : 0)) * HW_ + li + (more ? kx : 0);
            const float* arow = smem + prow * LDH + (SWZ ? 0 : 4 * lh);
            const int sv = (4 * lh) ^ ((prow & 15) * 4);
            const float* wn = wl + (!more ? 0 : (size_t)tn * NCH * COUT * 8);
#pragma unroll
            for (int cc = 0; cc < NCH; ++cc) {
                const float4 a = ra[cc], b0 = rb0[cc], b1 = rb1[cc];
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
                // (after the last tap this harmlessly re-reads tap 0: no branch in the loop body)
                rb0[cc] = *reinterpret_cast<const float4*>(wn + (size_t)cc * COUT * 8);
                rb1[cc] = *reinterpret_cast<const float4*>(wn + (size_t)cc * COUT * 8 + 32 * 8);
                ra[cc] = *reinterpret_cast<const float4*>(arow + (SWZ ? ((cc * 8) ^ sv) : cc * 8));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        float4 nb0 = *reinterpret_cast<const float4*>(wl);
        float4 nb1 = *reinterpret_cast<const float4*>(wl + 32 * 8);
#pragma unroll 1
        for (int tap = 0; tap < KS * KS; ++tap) {
            const int ky = tap / KS, kx = tap % KS;
            const float* arow = smem + ((wave + ky) * HW_ + li + kx) * LDH + 4 * lh;
#pragma unroll
            for (int cc = 0; cc < NCH; ++cc) {
                const int it = tap * NCH + cc;
                const float4 b0 = nb0, b1 = nb1;
                if (it + 1 < NIT) {
                    const size_t wo = (size_t)(it + 1) * COUT * 8;
                    nb0 = *reinterpret_cast<const float4*>(wl + wo);
                    nb1 = *reinterpret_cast<const float4*>(wl + wo + 32 * 8);
                }
                const float4 a = *reinterpret_cast<const float4*>(arow + cc * 8);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, acc1, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, acc1, 0, 0, 0);
            }
        }
    }

    // ---- epilogue: bias, relu / elu, + posmap, activation-derivative mask.  The 32-pixel x 64-channel wave tile goes through a
    // wave-private LDS patch (the halo is dead once every wave has left the tap loop) and leaves as float4 runs of a pixel's
    // channels: 8 store instructions of full 256-byte pixels instead of 32 two-pixel scalar stores, float4 posmap / mask reads.
    const int y = y0 + wave;
    if (HH_ * HW_ * LDH >= 4 * 32 * (COUT + 4)) {       // (compile-time) the halo region holds the four patches
        __syncthreads();
        float* patch = smem + wave * 32 * (COUT + 4);
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * (COUT + 4) + tn * 32 + li] = tn == 0 ? acc0[r] : acc1[r];
        __builtin_amdgcn_wave_barrier();
        if (y >= p.H) return;
        const int c4 = lane & 15, px0 = lane >> 4;
        const float4 bv = p.bias ? *reinterpret_cast<const float4*>(p.bias + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ps = 0; ps < 8; ++ps) {
            const int px = ps * 4 + px0, x = x0 + px;
            if (x >= p.W) continue;
            float4 v = *reinterpret_cast<const float4*>(patch + px * (COUT + 4) + c4 * 4);
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
            if (p.relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            else if (p.relu == 2) {
                v.x = v.x > 0.f ? v.x : __expf(v.x) - 1.f; v.y = v.y > 0.f ? v.y : __expf(v.y) - 1.f;
                v.z = v.z > 0.f ? v.z : __expf(v.z) - 1.f; v.w = v.w > 0.f ? v.w : __expf(v.w) - 1.f;
            }
            const size_t pix = ((size_t)b * p.H + y) * p.W + x;
            if (p.posmap) {
                const float4 pm = *reinterpret_cast<const float4*>(p.posmap + ((size_t)y * p.W + x) * COUT + c4 * 4);
                v.x += pm.x; v.y += pm.y; v.z += pm.z; v.w += pm.w;
            }
            if (p.mask) {
                const float4 mk = *reinterpret_cast<const float4*>(p.mask + pix * COUT + c4 * 4);
                const float e = p.mask_elu ? 1.f : 0.f;
                v.x = mk.x > 0.f ? v.x : e * v.x * (mk.x + 1.f); v.y = mk.y > 0.f ? v.y : e * v.y * (mk.y + 1.f);
                v.z = mk.z > 0.f ? v.z : e * v.z * (mk.z + 1.f); v.w = mk.w > 0.f ? v.w : e * v.w * (mk.w + 1.f);
            }
            *reinterpret_cast<float4*>(p.Y + pix * COUT + c4 * 4) = v;
        }
        return;
    }
    if (y >= p.H) return;
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int co = tn * 32 + li;
        const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int x = x0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (x >= p.W) continue;
            float v = (tn == 0 ? acc0[r] : acc1[r]) + bv;
            if (p.relu == 1) v = fmaxf(v, 0.f);
            else if (p.relu == 2) v = v > 0.f ? v : __expf(v) - 1.f;
            const size_t pix = ((size_t)b * p.H + y) * p.W + x;
            if (p.posmap) v += p.posmap[((size_t)y * p.W + x) * COUT + co];
            if (p.mask) {
                const float mk = p.mask[pix * COUT + co];
                v = mk > 0.f ? v : (p.mask_elu ? v * (mk + 1.f) : 0.f);
            }
            p.Y[pix * COUT + co] = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Low-latency forward for grids that do not fill the machine (the RL extractor's encode() at num_envs images: one 64x64 image is 32
// tiles of the kernel above, each a 43 us dependent MFMA chain on one CU while 224 CUs idle).  One workgroup = one row segment of 32
// pixels x 32 output channels; its four waves split the INPUT channels (16 each: k-split) over all KS*KS taps, each wave stages only
// its own channels of the halo in a wave-private LDS region (no workgroup barrier before the MFMAs), and the four partial 32x32 tiles
// are summed in a fixed order in the epilogue.  200 MFMAs per wave instead of 1600: ~8 us per layer at 256 workgroups.
// Forward features only (bias, ReLU, position map).
template <int KS, int CIN>
__global__ __launch_bounds__(256) void conv_lat_kernel(ConvArgs p) {
    static_assert(CIN == 64, "four waves x 16 input channels");
    constexpr int P = KS / 2, HW_ = TW + KS - 1, NPX = KS * HW_;
    constexpr int LDW = 20;                       // 16 channels + 4: conflict-free ds_read_b128 across 16 pixels
    constexpr int REGION = NPX * LDW;             // floats per wave (>= the 32 x 36 partial patch)
    static_assert(REGION >= 32 * 36, "partial patch must fit the wave's halo region");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int tiles_x = (p.W + TW - 1) / TW;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int y = bid % p.H;
    const int b = bid / p.H;
    const int x0 = tx * TW, half = blockIdx.y;
    float* reg = smem + wave * REGION;
    {   // this wave's 16 channels of the KS x (32 + KS - 1) halo: every load issued before the first LDS store
        constexpr int TOTAL = NPX * 4, NLD = (TOTAL + 63) / 64;
        const float* Xb = p.X + (size_t)b * p.H * p.W * CIN + wave * 16;
        float4 hv[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = lane + i * 64, c4 = idx & 3, hp = idx >> 2;
            const int hx = hp % HW_, hy = hp / HW_;
            const int yy = y - P + hy, xx = x0 - P + hx;
            hv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < TOTAL && yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) hv[i] = *reinterpret_cast<const float4*>(Xb + ((size_t)yy * p.W + xx) * CIN + c4 * 4);
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = lane + i * 64;
            if (idx < TOTAL) *reinterpret_cast<float4*>(reg + (idx >> 2) * LDW + (idx & 3) * 4) = hv[i];
        }
    }
    __builtin_amdgcn_wave_barrier();
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // packed weights [tap][CIN/8][64][8]: this wave's chunks 2w, 2w+1; this half's output channels
    const float* wl = p.Wp + ((size_t)(2 * wave) * 64 + half * 32 + li) * 8 + 4 * lh;
    constexpr int NIT = KS * KS * 2, RING = 10;
    float4 rb[RING];
#pragma unroll
    for (int i = 0; i < RING; ++i) rb[i] = *reinterpret_cast<const float4*>(wl + ((size_t)(i >> 1) * (CIN / 8) + (i & 1)) * 64 * 8);
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int tap = it >> 1, cc = it & 1, ky = tap / KS, kx = tap - ky * KS;
        const float4 a = *reinterpret_cast<const float4*>(reg + (ky * HW_ + li + kx) * LDW + cc * 8 + 4 * lh);
        const float4 w = rb[it % RING];
        if (it + RING < NIT) rb[it % RING] = *reinterpret_cast<const float4*>(wl + ((size_t)((it + RING) >> 1) * (CIN / 8) + ((it + RING) & 1)) * 64 * 8);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, w.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, w.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, w.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, w.w, acc, 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();            // the wave is done reading its halo: the region becomes its partial patch [32 px][36]
#pragma unroll
    for (int r = 0; r < 16; ++r) reg[((r & 3) + 8 * (r >> 2) + 4 * lh) * 36 + li] = acc[r];
    __syncthreads();
    const int px = threadIdx.x >> 3, c4 = threadIdx.x & 7, x = x0 + px;
    if (x >= p.W) return;
    float4 v = *reinterpret_cast<const float4*>(smem + px * 36 + c4 * 4);
#pragma unroll
    for (int w = 1; w < 4; ++w) {               // fixed order: waves 0..3
        const float4 u = *reinterpret_cast<const float4*>(smem + w * REGION + px * 36 + c4 * 4);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const int co = half * 32 + c4 * 4;
    if (p.bias) { const float4 bv = *reinterpret_cast<const float4*>(p.bias + co); v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w; }
    if (p.relu == 1) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    if (p.posmap) {
        const float4 pm = *reinterpret_cast<const float4*>(p.posmap + ((size_t)y * p.W + x) * 64 + co);
        v.x += pm.x; v.y += pm.y; v.z += pm.z; v.w += pm.w;
    }
    *reinterpret_cast<float4*>(p.Y + (((size_t)b * p.H + y) * p.W + x) * 64 + co) = v;
}

// ---------------------------------------------------------------------------------------------
template <int KS, int CIN, int COUT>
__global__ __launch_bounds__(256, 2) void conv_wgrad_kernel(WgradArgs p) {
    static_assert(COUT == 64, "COUT must be 64");
    static_assert(CIN == 64 || CIN == 8, "CIN must be 64 or 8");
    constexpr int P = KS / 2;
    constexpr int HW_ = TW + KS - 1;
    constexpr int KSPLIT = (CIN == 64) ? 1 : 2;     // CIN=8: waves split the pixel rows instead of ci
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dYs = smem;                     // [TH*TW][COUT]
    float* Xs = smem + TH * TW * COUT;     // [TH][HW_][CIN]

    // (chunk worker, kernel row) from the linear dispatch index, re-dealt so that the KS workgroups that stream the same pixel tiles
    // for different kernel rows sit on one XCD (the dispatcher deals linear ids round-robin to the 8 XCDs): their X / dY tiles then
    // come out of that XCD's L2 four times out of five (PMC: 5.4 GB per launch for 1.07 GB of operands before)
    int lin = blockIdx.x + gridDim.x * blockIdx.y;
    {
        const int nwg = gridDim.x * gridDim.y, q = nwg >> 3, r = nwg & 7, x = lin & 7, y = lin >> 3;
        lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int cx = lin / (int)gridDim.y, ky = lin % (int)gridDim.y;
    const int tiles_x = (p.W + TW - 1) / TW, tiles_y = (p.H + TH - 1) / TH;
    const int ntiles = tiles_x * tiles_y * p.B;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int coh = (CIN == 64) ? (wave >> 1) : (wave & 1);
    const int cih = (CIN == 64) ? (wave & 1) : 0;
    const int ksp = (CIN == 64) ? 0 : (wave >> 1);
    const int lci = (CIN == 64) ? li : (li < CIN ? li : CIN - 1);   // clamp: columns >= CIN are discarded

    f32x16 acc[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    // global -> register staging of one tile (dY tile + X rows shifted by this workgroup's ky): the loads of tile t+1
    // are in flight while tile t's MFMAs run; the registers are written to LDS after the barrier that retires tile t.
    constexpr int FY = COUT / 4, FX = CIN / 4;
    constexpr int NDY = (TH * TW * FY + 255) / 256, XT = TH * HW_ * FX, NX = (XT + 255) / 256;
    float4 rdy[NDY], rx[NX];
    auto gload = [&](int t) {
        int q = t;
        const int tx = q % tiles_x; q /= tiles_x;
        const int ty = q % tiles_y; q /= tiles_y;
        const int b = q;
        const int x0 = tx * TW, y0 = ty * TH;
        const float* gy = p.dY + (size_t)b * p.H * p.W * COUT;
        const float* gx = p.X + (size_t)b * p.H * p.W * CIN;
#pragma unroll
        for (int i = 0; i < NDY; ++i) {
            const int idx = threadIdx.x + i * 256;
            const int c4 = idx % FY, pp = idx / FY;
            const int x = x0 + (pp % TW), y = y0 + (pp / TW);
            rdy[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y < p.H && x < p.W) rdy[i] = *reinterpret_cast<const float4*>(gy + ((size_t)y * p.W + x) * COUT + c4 * 4);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = threadIdx.x + i * 256;
            const int c4 = idx % FX, hp = idx / FX;
            const int x = x0 - P + (hp % HW_), y = y0 + ky - P + (hp / HW_);
            rx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < XT && y >= 0 && y < p.H && x >= 0 && x < p.W) rx[i] = *reinterpret_cast<const float4*>(gx + ((size_t)y * p.W + x) * CIN + c4 * 4);
        }
    };
    if ((int)cx < ntiles) gload(cx);
    for (int t = cx; t < ntiles; t += gridDim.x) {
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < NDY; ++i) {
            const int idx = threadIdx.x + i * 256;
            *reinterpret_cast<float4*>(dYs + (idx / FY) * COUT + (idx % FY) * 4) = rdy[i];
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = threadIdx.x + i * 256;
            if (idx < XT) *reinterpret_cast<float4*>(Xs + (idx / FX) * CIN + (idx % FX) * 4) = rx[i];
        }
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) gload(t + gridDim.x);
        constexpr int R0N = TH / KSPLIT;
#pragma unroll 1
        for (int rr = 0; rr < R0N; ++rr) {
            const int r = ksp * R0N + rr;
            const float* ay = dYs + (r * TW + lh) * COUT + coh * 32 + li;
            const float* bx = Xs + (r * HW_ + lh) * CIN + cih * 32 + lci;
#pragma unroll 4
            for (int xx = 0; xx < TW; xx += 2) {
                const float a = ay[xx * COUT];
#pragma unroll
                for (int kx = 0; kx < KS; ++kx) {
                    const float bb = bx[(xx + kx) * CIN];
                    acc[kx] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc[kx], 0, 0, 0);
                }
            }
        }
    }
    // ---- partial slab [slab][ky*KS+kx][co][CIN]
    const int slab = cx * KSPLIT + ksp;
    float* out = p.part + ((size_t)slab * KS * KS + ky * KS) * COUT * CIN;
    if (CIN == 64 || li < CIN) {
#pragma unroll
        for (int kx = 0; kx < KS; ++kx)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = coh * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                out[((size_t)kx * COUT + co) * CIN + cih * 32 + li] = acc[kx][r];
            }
    }
}

// dW[co][ci][ky][kx] (reference layout, cin_real channels) = sum_slabs part[slab][tap][co][ci]
__global__ void conv_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW, int nslab,
                                         int KS, int CIN, int COUT, int cin_real, int accumulate) {
    const int n = KS * KS * COUT * CIN;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ci = i % CIN, co = (i / CIN) % COUT, tap = i / (CIN * COUT);
    if (ci >= cin_real) return;
    float s = 0.f;
    for (int k = 0; k < nslab; ++k) s += part[(size_t)k * n + i];
    float* d = dW + ((size_t)co * cin_real + ci) * KS * KS + tap;
    *d = accumulate ? *d + s : s;
}

// W[co][ci][ky][kx] -> forward pack [tap][CIN/8][COUT][8]   (ci >= cin_real -> 0)
//                   -> backward-data pack [tap'][COUT/8][cin][8] with tap' = flipped tap (roles of ci/co swapped)
__global__ void conv_pack_kernel(const float* __restrict__ W, float* __restrict__ fwd, float* __restrict__ bwd,
                                 int KS, int CIN, int COUT, int cin_real) {
    const int n = KS * KS * CIN * COUT;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // i enumerates the forward pack: [tap][cc][co][e]
    const int e = i % 8, co = (i / 8) % COUT, cc = (i / (8 * COUT)) % (CIN / 8), tap = i / (8 * COUT * (CIN / 8));
    const int ci = cc * 8 + e;
    const float v = ci < cin_real ? W[((size_t)co * cin_real + ci) * KS * KS + tap] : 0.f;
    fwd[i] = v;
    if (bwd) {   // only for CIN == cin_real (square layers): bwd[tapf][co/8][ci][co%8]
        const int tapf = KS * KS - 1 - tap;
        bwd[(((size_t)tapf * (COUT / 8) + co / 8) * CIN + ci) * 8 + (co % 8)] = v;
    }
}

template <int KS, int CIN>
static int conv_fwd_cfg(const ConvArgs& a, hipStream_t st) {
    constexpr int smem = (TH + KS - 1) * (TW + KS - 1) * ConvCfg<KS, CIN>::LDH * 4;
    static bool attr_set = false;
    if (!attr_set) {
        OCRL_HIP(hipFuncSetAttribute((const void*)conv_fwd_kernel<KS, CIN, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    const int grid = cdiv(a.W, TW) * cdiv(a.H, TH) * a.B;
    const int pi = prof_begin((KS == 5 && CIN == 64) ? PROF_CONV5 : PROF_CONV_OTHER, st);
    hipLaunchKernelGGL((conv_fwd_kernel<KS, CIN, 64>), dim3(grid), dim3(256), smem, st, a);
    prof_end(pi, st);
    OCRL_CHECK_LAUNCH("conv_fwd_kernel");
    return 0;
}

// the low-latency variant pays when the throughput kernel's grid leaves most CUs idle: fewer than ~0.6 workgroups per CU (measured: 64x64, B = 8 = 256 tiles is already faster on the throughput kernel)
static bool conv_lat_applies(const ConvArgs& a, int KS, int CIN) {
    static int on = -1;
    if (on < 0) { const char* e = getenv("OCRL_CONV_LAT"); on = e ? atoi(e) : 1; }
    return on && KS == 5 && CIN == 64 && !a.mask && a.relu <= 1 && (long long)cdiv(a.W, TW) * cdiv(a.H, TH) * a.B <= 160;
}
int conv_fwd_launch(const ConvArgs& a_in, int KS, int CIN, int COUT, hipStream_t st, int low_latency) {
    ConvArgs a = a_in;
    OCRL_REQUIRE(COUT == 64, "conv: COUT must be 64 (got %d)", COUT);
    OCRL_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0, "conv: empty input");
    OCRL_REQUIRE(((uintptr_t)a.X & 15) == 0 && ((uintptr_t)a.Wp & 15) == 0, "conv: X/Wp must be 16-byte aligned");
    if (low_latency && conv_lat_applies(a, KS, CIN)) {
        constexpr int smem = 4 * 5 * (TW + 4) * 20 * 4;
        const int pi = prof_begin(PROF_CONV_OTHER, st);
        hipLaunchKernelGGL((conv_lat_kernel<5, 64>), dim3(cdiv(a.W, TW) * a.H * a.B, 2), dim3(256), smem, st, a);
        prof_end(pi, st);
        OCRL_CHECK_LAUNCH("conv_lat_kernel");
        return 0;
    }
    if (KS == 5 && CIN == 64) return conv_fwd_cfg<5, 64>(a, st);
    if (KS == 5 && CIN == 8) return conv_fwd_cfg<5, 8>(a, st);
    if (KS == 3 && CIN == 64) return conv_fwd_cfg<3, 64>(a, st);
    ocrl_set_error("conv: unsupported KS=%d CIN=%d", KS, CIN);
    return 1;
}

template <int KS, int CIN>
static int conv_wgrad_cfg(const WgradArgs& a, int nchunk, hipStream_t st) {
    constexpr int smem = (TH * TW * 64 + TH * (TW + KS - 1) * CIN) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        OCRL_HIP(hipFuncSetAttribute((const void*)conv_wgrad_kernel<KS, CIN, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    const int pi = prof_begin(PROF_WGRAD, st);
    hipLaunchKernelGGL((conv_wgrad_kernel<KS, CIN, 64>), dim3(nchunk, KS), dim3(256), smem, st, a);
    prof_end(pi, st);
    OCRL_CHECK_LAUNCH("conv_wgrad_kernel");
    return 0;
}

int conv_wgrad_chunks(int B, int H, int W, int KS) {
    const int ntiles = cdiv(W, TW) * cdiv(H, TH) * B;
    int n = (2 * 256) / KS;          // all workgroups co-resident (2 per CU): no straggler round
    return n < ntiles ? n : ntiles;
}
size_t conv_wgrad_ws_floats(int B, int H, int W, int KS, int CIN) {
    const int ks = CIN == 64 ? 1 : 2;
    return (size_t)conv_wgrad_chunks(B, H, W, KS) * ks * KS * KS * 64 * CIN;
}

int conv_wgrad_launch(const WgradArgs& a, int KS, int CIN, int COUT, int cin_real, float* dW, int accumulate, hipStream_t st, int x3) {
    OCRL_REQUIRE(COUT == 64, "conv wgrad: COUT must be 64 (got %d)", COUT);
    const int nchunk = conv_wgrad_chunks(a.B, a.H, a.W, KS);
    static int x3_env = -1;
    if (x3_env < 0) { const char* e = getenv("OCRL_CONV_X3"); x3_env = e ? atoi(e) : 0; }
    if (x3 < 0) x3 = x3_env;
    int rc;
    if ((KS == 5 || KS == 3) && CIN == 64 && x3 > 0) rc = conv_wgrad_x3_stage(a, nchunk, st, KS);
    else if (KS == 5 && CIN == 64) rc = conv_wgrad_cfg<5, 64>(a, nchunk, st);
    else if (KS == 5 && CIN == 8) rc = conv_wgrad_cfg<5, 8>(a, nchunk, st);
    else if (KS == 3 && CIN == 64) rc = conv_wgrad_cfg<3, 64>(a, nchunk, st);
    else { ocrl_set_error("conv wgrad: unsupported KS=%d CIN=%d", KS, CIN); return 1; }
    if (rc) return rc;
    const int nslab = nchunk * (CIN == 64 ? 1 : 2);
    const int n = KS * KS * COUT * CIN;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, a.part, dW, nslab, KS, CIN, COUT, cin_real, accumulate);
    OCRL_CHECK_LAUNCH("conv_wgrad_reduce");
    return 0;
}

int conv_pack_launch(const float* W, float* fwd, float* bwd, int KS, int CIN, int COUT, int cin_real, hipStream_t st) {
    OCRL_REQUIRE(bwd == nullptr || (CIN == cin_real && CIN == COUT), "conv pack: backward pack needs a square layer");
    const int n = KS * KS * CIN * COUT;
    hipLaunchKernelGGL(conv_pack_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, W, fwd, bwd, KS, CIN, COUT, cin_real);
    OCRL_CHECK_LAUNCH("conv_pack");
    return 0;
}
