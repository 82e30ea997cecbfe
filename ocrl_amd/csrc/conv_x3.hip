// EXPLORATORY (OCRL_CONV_X3=1, never the default, never the headline bench line): the 5x5 and 3x3 / 64-channel convolutions (forward,
// backward data, weight gradient) on the bf16 matrix pipe with fp32-equivalent split precision.  gfx950 has no xf32 / tf32 MFMA and the fp32 MFMA runs at 1/16 of the bf16 rate, so the
// fp32 convolutions (0.87 of that peak) cannot get faster on it.  Here every fp32 operand is written as the EXACT sum of three bf16
// numbers, x = h + m + l (h = bf16(x) rounded to nearest, m = bf16(x - h), l = x - h - m: both subtractions are exact in fp32 and l fits
// bf16), and x*w is accumulated in fp32 from six v_mfma_f32_32x32x16_bf16 products
//      h*h' + h*m' + m*h' + m*m' + h*l' + l*h'
// -- the three dropped terms (m*l', l*m', l*l') are each below 2^-24 of |x*w| (|m| <= 2^-8 |x|, |l| <= 2^-16 |x|), the rounding of one fp32
// product; against fp64 the results measure the same as the fp32-MFMA kernels'.  Six bf16 products of 16 k
// each cost 6 x 32 cycles for what eight fp32 32x32x2 products need 8 x 64 cycles for: 2.67x fewer matrix-pipe cycles.
//
// Kernel shape: 8 rows x 32 pixels per workgroup, a wave owns two rows (two 32-pixel M blocks) x 64 output channels, so one weight
// fragment feeds two MFMAs per product and the packed weights (three bf16 planes, [tap][chunk][n-block][plane][lane][8]) can stream
// from L2 at 32 B/clk/CU.  The halo stays fp32 in LDS, 32 input channels at a time (two phases per tile, 62 KB, two workgroups per
// CU; the first version held all 64 channels, 117.5 KB, one workgroup per CU: 2.05 ms per launch at B = 128 / 128x128), and an A
// fragment (8 channels of a pixel) is split into its three bf16 planes in registers right before its MFMAs (three v_cvt_pk_bf16_f32, two
// shifts / masks and four exact subtractions per pair of values, issued in the shadow of the MFMAs).  Same epilogue features as
// conv_fwd_kernel (bias, ReLU / ELU, position map, activation mask), so it also serves the backward-data pass.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define X3_TH 8
#define X3_TW 32
#define X3_NCH 4          // 16-channel chunks of the 64 input channels

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
// two fp32 values -> their bf16 roundings (round to nearest even: v_cvt_pk_bf16_f32), packed {hi16 = b, lo16 = a}
__device__ __forceinline__ uint32_t x3_pk(float a, float b) {
    const f32x2_t v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ float x3_lo(uint32_t pk) { return __builtin_bit_cast(float, pk << 16); }
__device__ __forceinline__ float x3_up(uint32_t pk) { return __builtin_bit_cast(float, pk & 0xFFFF0000u); }
// eight fp32 values (two float4) -> three planes of eight bf16 (h, m, l), x = h + m + l exactly: h = bf16(x), m = bf16(x - h), l = x - h - m
// (both subtractions are exact in fp32 and l has at most 8 significant bits, so it is a bf16 number).  Rounding to nearest instead of
// truncating halves |m| and |l|: |m| <= 2^-8 |x|, |l| <= 2^-16 |x|, the dropped products m*l', l*m' are below 2^-24 of |x*w|.
__device__ __forceinline__ void x3_split(const float4& a, const float4& b, uint4& h, uint4& m, uint4& l) {
    const float x[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    uint32_t ph[4], pm[4], pl[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const float x0 = x[2 * p], x1 = x[2 * p + 1];
        ph[p] = x3_pk(x0, x1);
        const float r0 = x0 - x3_lo(ph[p]), r1 = x1 - x3_up(ph[p]);
        pm[p] = x3_pk(r0, r1);
        pl[p] = x3_pk(r0 - x3_lo(pm[p]), r1 - x3_up(pm[p]));
    }
    h = make_uint4(ph[0], ph[1], ph[2], ph[3]); m = make_uint4(pm[0], pm[1], pm[2], pm[3]); l = make_uint4(pl[0], pl[1], pl[2], pl[3]);
}
#define X3_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

// Two workgroups per CU: the input channels are processed in two phases of 32 (a 12 x 36 x 36-float halo = 62 KB per workgroup), so
// one workgroup's halo loads and epilogue run under the other's MFMAs and every SIMD holds two waves.
#define X3_CP 32          // input channels per phase
#define X3_LDP 36         // floats per halo pixel of a phase (32 + 4: conflict-free ds_read_b128 over 16 pixels)
template <int KS>
__global__ __launch_bounds__(256, 2) void conv_x3_kernel(ConvArgs p, const uint4* __restrict__ Wp3) {
    constexpr int P = KS / 2, HW_ = X3_TW + KS - 1, HH_ = X3_TH + KS - 1, LDH = X3_LDP, NCL = X3_CP / 16, CIN = 64, COUT = 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tiles_x = (p.W + X3_TW - 1) / X3_TW, tiles_y = (p.H + X3_TH - 1) / X3_TH;
    int bid = blockIdx.x;
    {   // contiguous runs of tiles per XCD (vertically adjacent tiles share halo rows in that XCD's L2), as conv_fwd_kernel
        const int nwg = gridDim.x, q = nwg >> 3, r = nwg & 7, x = bid & 7, y = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; bid /= tiles_y;
    const int b = bid;
    const int x0 = tx * X3_TW, y0 = ty * X3_TH;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    const float* Xb = p.X + (size_t)b * p.H * p.W * CIN;
    const float* abase = smem + ((2 * wave) * HW_ + li) * LDH + 8 * lh;

#pragma unroll 1
    for (int ph = 0; ph < CIN / X3_CP; ++ph) {
        if (ph) __syncthreads();            // every wave has left the previous phase's halo
        {   // this phase's 32 channels of the halo -> LDS, two batches (loads of a batch issued before its stores)
            constexpr int F4 = X3_CP / 4, TOTAL = HH_ * HW_ * F4, NB = 7, NBATCH = (TOTAL + 256 * NB - 1) / (256 * NB);
#pragma unroll 1
            for (int bt = 0; bt < NBATCH; ++bt) {
                float4 hv[NB];
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int idx = threadIdx.x + (bt * NB + i) * 256;
                    const int c4 = idx % F4, hp = idx / F4;
                    const int hx = hp % HW_, hy = hp / HW_;
                    const int y = y0 - P + hy, x = x0 - P + hx;
                    hv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (idx < TOTAL && y >= 0 && y < p.H && x >= 0 && x < p.W)
                        hv[i] = *reinterpret_cast<const float4*>(Xb + ((size_t)y * p.W + x) * CIN + ph * X3_CP + c4 * 4);
                }
#pragma unroll
                for (int i = 0; i < NB; ++i) {
                    const int idx = threadIdx.x + (bt * NB + i) * 256;
                    if (idx < TOTAL) *reinterpret_cast<float4*>(smem + (idx / F4) * LDH + (idx % F4) * 4) = hv[i];
                }
            }
        }
        __syncthreads();
        // Operand rings of one tap (NCL slots): slot c holds the raw A values (two rows) and the six B fragments of the phase's chunk c.
        // A tap is NCL regions, one per chunk, fenced by sched_barriers so the compiler keeps this order (left alone it sinks every load to
        // the top of the next tap and the MFMAs wait for them).  Region c: (1) refill slot c's B fragments with the NEXT tap's chunk c;
        // (2) the 24 MFMAs of chunk c on the A planes that were split one region earlier; (3) split the raw A values of the next chunk
        // into their planes (VALU work in the shadow of (2)); (4) refill that slot's raw A values from LDS.
        const uint4* wl = Wp3 + (size_t)(ph * NCL) * 2 * 3 * 64 + lane;          // chunk index inside a tap: ph * NCL + c
        float4 ra[NCL][2][2];
        uint4 rb[NCL][2][3];
#pragma unroll
        for (int c = 0; c < NCL; ++c) {
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) rb[c][n][pl] = wl[((c * 2 + n) * 3 + pl) * 64];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                ra[c][m][0] = *reinterpret_cast<const float4*>(abase + m * HW_ * LDH + c * 16);
                ra[c][m][1] = *reinterpret_cast<const float4*>(abase + m * HW_ * LDH + c * 16 + 4);
            }
        }
        uint4 ah[2], am[2], al[2];           // planes of the chunk whose MFMAs come next
#pragma unroll
        for (int m = 0; m < 2; ++m) x3_split(ra[0][m][0], ra[0][m][1], ah[m], am[m], al[m]);
#pragma unroll
        for (int m = 0; m < 2; ++m) {        // slot 0's raw values are consumed: refill it with tap 1's chunk 0 (the last region of tap 0 splits it)
            ra[0][m][0] = *reinterpret_cast<const float4*>(abase + 1 * LDH + m * HW_ * LDH);
            ra[0][m][1] = *reinterpret_cast<const float4*>(abase + 1 * LDH + m * HW_ * LDH + 4);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll 1
        for (int tap = 0; tap < KS * KS; ++tap) {
            // next tap (B refills, A refills of the chunks > 0) and the tap after it (A refill of chunk 0, whose split runs one region early)
            const int tn = tap + 1 < KS * KS ? tap + 1 : 0, t2 = tap + 2 < KS * KS ? tap + 2 : 0;
            const float* an = abase + ((tn / KS) * HW_ + tn % KS) * LDH;
            const float* an2 = abase + ((t2 / KS) * HW_ + t2 % KS) * LDH;
            const uint4* wn = wl + (size_t)tn * X3_NCH * 2 * 3 * 64;
#pragma unroll
            for (int c = 0; c < NCL; ++c) {
                uint4 bh[2], bm[2], bl[2];
#pragma unroll
                for (int n = 0; n < 2; ++n) { bh[n] = rb[c][n][0]; bm[n] = rb[c][n][1]; bl[n] = rb[c][n][2]; }
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) rb[c][n][pl] = wn[((c * 2 + n) * 3 + pl) * 64];
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) {
                        f32x16 a_ = acc[m][n];
                        a_ = X3_MFMA(al[m], bh[n], a_);          // small terms first
                        a_ = X3_MFMA(ah[m], bl[n], a_);
                        a_ = X3_MFMA(am[m], bm[n], a_);
                        a_ = X3_MFMA(am[m], bh[n], a_);
                        a_ = X3_MFMA(ah[m], bm[n], a_);
                        a_ = X3_MFMA(ah[m], bh[n], a_);
                        acc[m][n] = a_;
                    }
                const int cn = (c + 1) & (NCL - 1);
                uint4 nh[2], nm[2], nl[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) x3_split(ra[cn][m][0], ra[cn][m][1], nh[m], nm[m], nl[m]);
                {   // slot cn's raw values are consumed: refill it -- chunk cn of the next tap; for cn = 0 of the tap after the next, because
                    // slot 0 already holds the next tap's chunk 0 when the last region splits it
                    const float* src = (cn == 0 ? an2 : an) + cn * 16;
#pragma unroll
                    for (int m = 0; m < 2; ++m) {
                        ra[cn][m][0] = *reinterpret_cast<const float4*>(src + m * HW_ * LDH);
                        ra[cn][m][1] = *reinterpret_cast<const float4*>(src + m * HW_ * LDH + 4);
                    }
                }
#pragma unroll
                for (int m = 0; m < 2; ++m) { ah[m] = nh[m]; am[m] = nm[m]; al[m] = nl[m]; }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- epilogue (conv_fwd_kernel's), one of the wave's two rows at a time: a 32-pixel x 64-channel tile through the wave's LDS
    // patch, float4 rows out
    __syncthreads();
    const int c4 = lane & 15, px0 = lane >> 4;
    const float4 bv = p.bias ? *reinterpret_cast<const float4*>(p.bias + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    float* patch = smem + wave * 32 * (COUT + 4);
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        if (m) __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 16; ++r) patch[((r & 3) + 8 * (r >> 2) + 4 * lh) * (COUT + 4) + tn * 32 + li] = acc[m][tn][r];
        __builtin_amdgcn_wave_barrier();
        const int y = y0 + 2 * wave + m;
        if (y >= p.H) continue;
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
    }
}

// W [64][64][KS][KS] (reference layout) -> the split-precision packs: fwd[s][n][plane][lane] and, flipped / transposed for the
// backward-data pass, bwd[..] (uint4 = eight bf16: k = 8*(lane>>5) + j of chunk c, column (lane & 31) of n-block n; s = tap*4 + c)
__global__ void conv_pack_x3_kernel(const float* __restrict__ W, uint4* __restrict__ fwd, uint4* __restrict__ bwd, int KK) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;          // (s, n, lane); KK = KS * KS taps
    if (i >= KK * 4 * 2 * 64) return;
    const int lane = i & 63, n = (i >> 6) & 1, s = i >> 7, c = s & 3, tap = s >> 2;
    const int r = lane & 31, h = lane >> 5;
    float vf[8], vb[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = c * 16 + 8 * h + j, col = n * 32 + r;
        vf[j] = W[((size_t)col * 64 + k) * KK + tap];                 // forward: B[k = ci][col = co]
        vb[j] = W[((size_t)k * 64 + col) * KK + (KK - 1 - tap)];      // backward data: B[k = co][col = ci] at the flipped tap
    }
    uint4 ph, pm, pl;
    x3_split(make_float4(vf[0], vf[1], vf[2], vf[3]), make_float4(vf[4], vf[5], vf[6], vf[7]), ph, pm, pl);
    const size_t o = ((size_t)(s * 2 + n) * 3) * 64 + lane;
    fwd[o] = ph; fwd[o + 64] = pm; fwd[o + 128] = pl;
    if (bwd) {
        x3_split(make_float4(vb[0], vb[1], vb[2], vb[3]), make_float4(vb[4], vb[5], vb[6], vb[7]), ph, pm, pl);
        bwd[o] = ph; bwd[o + 64] = pm; bwd[o + 128] = pl;
    }
}

// ---------------------------------------------------------------------------------------------
// Weight gradient of the same layer on the split-precision path: dW[tap][co][ci] = sum_pixels dY[p][co] * X[p + tap][ci], pixels are
// the MFMA k dimension (16 per v_mfma_f32_32x32x16_bf16).  Same decomposition as conv_wgrad_kernel (conv.hip): grid = (chunks, KS), a
// workgroup owns kernel row ky and walks 4 x 32-pixel tiles, a wave owns one 32 x 32 quadrant of [co x ci] for the five kx, partial
// slabs [slab][ky*5+kx][co][ci] go to conv_wgrad_reduce_kernel.  A bf16 MFMA wants its 8 k-values per lane contiguous, i.e. 8
// consecutive PIXELS of one channel, so the tiles are stored transposed in LDS ([channel][pixel], fp32) and a fragment is two
// (dY) or three (X: the 12 pixels that cover the five kx windows) ds_read_b128; the values are split into their three bf16 planes in
// registers, the X planes packed once for even and once for odd kx (the windows of neighbouring kx overlap by 7 pixels).
#define WX_TH 4
#define WX_TW 32
#define WX_LDY (WX_TH * WX_TW + 4)        // floats per co row of the transposed dY tile
// transposed X tile: rows of HW = 32 + KS - 1 pixels stored at a stride of HWP (a multiple of 4, so every fragment read is 16-byte
// aligned), LDX floats per ci row (+4: conflict-free reads)
template <int KS> struct WxGeo { static constexpr int HW = WX_TW + KS - 1, HWP = (HW + 3) & ~3, NPX = WX_TH * HWP, LDX = NPX + 4, NV = KS + 7; };
// one value -> its three planes, each as an fp32 bit pattern with the low 16 bits clear (packed into pairs by X3_PK)
__device__ __forceinline__ void x3_planes(float v, uint32_t& h, uint32_t& m, uint32_t& l) {
    h = __builtin_bit_cast(uint32_t, (float)(__bf16)v);
    const float r = v - __builtin_bit_cast(float, h);
    m = __builtin_bit_cast(uint32_t, (float)(__bf16)r);
    l = __builtin_bit_cast(uint32_t, r - __builtin_bit_cast(float, m));
}
#define X3_PK(hi_, lo_) __builtin_amdgcn_perm((hi_), (lo_), 0x07060302u)

template <int KS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_x3_kernel(WgradArgs p) {
    constexpr int P = KS / 2, HW_ = WxGeo<KS>::HW, HWP = WxGeo<KS>::HWP, NPX = WxGeo<KS>::NPX, CIN = 64, COUT = 64, TH_ = WX_TH, TW_ = WX_TW,
                  LDY = WX_LDY, LDX = WxGeo<KS>::LDX, NV = WxGeo<KS>::NV, NE = NV / 2, NO = NV / 2 - 1;
    static_assert(KS == 3 || KS == 5, "window of 8 + KS - 1 pixels inside three float4");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dYt = smem;                    // [COUT][LDY]: dY tile, pixel index = row * 32 + x
    float* Xt = smem + COUT * LDY;        // [CIN][LDX]: X tile rows shifted by ky, pixel index = row * 36 + hx
    int lin = blockIdx.x + gridDim.x * blockIdx.y;
    {   // the KS workgroups that stream the same pixel tiles for different kernel rows sit on one XCD (conv_wgrad_kernel)
        const int nwg = gridDim.x * gridDim.y, q = nwg >> 3, r = nwg & 7, x = lin & 7, y = lin >> 3;
        lin = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int cx = lin / (int)gridDim.y, ky = lin % (int)gridDim.y;
    const int tiles_x = (p.W + TW_ - 1) / TW_, tiles_y = (p.H + TH_ - 1) / TH_;
    const int ntiles = tiles_x * tiles_y * p.B;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
    const int coh = wave >> 1, cih = wave & 1;
    f32x16 acc[KS];
#pragma unroll
    for (int k = 0; k < KS; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[k][r] = 0.f;

    constexpr int FY = COUT / 4, FX = CIN / 4;
    constexpr int NDY = (TH_ * TW_ * FY + 255) / 256, NX = (NPX + 15) / 16;      // one float4 per (pixel, channel quad): 256 per 16 pixels
    float4 rdy[NDY], rx[NX];
    auto gload = [&](int t) {
        int q = t;
        const int tx = q % tiles_x; q /= tiles_x;
        const int ty = q % tiles_y; q /= tiles_y;
        const int b = q;
        const int x0 = tx * TW_, y0 = ty * TH_;
        const float* gy = p.dY + (size_t)b * p.H * p.W * COUT;
        const float* gx = p.X + (size_t)b * p.H * p.W * CIN;
#pragma unroll
        for (int i = 0; i < NDY; ++i) {
            // a wave covers 16 consecutive pixels x 4 channel quads, so that its transposed LDS stores hit 64 different banks
            const int idx = threadIdx.x + i * 256;
            const int c4 = (idx >> 4) & 15, pp = ((idx >> 8) << 4) | (idx & 15);
            const int x = x0 + (pp % TW_), y = y0 + (pp / TW_);
            rdy[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (y < p.H && x < p.W) rdy[i] = *reinterpret_cast<const float4*>(gy + ((size_t)y * p.W + x) * COUT + c4 * 4);
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = threadIdx.x + i * 256;
            const int c4 = (idx >> 4) & 15, hp = ((idx >> 8) << 4) | (idx & 15);
            const int hx = hp % HWP, x = x0 - P + hx, y = y0 + ky - P + (hp / HWP);
            rx[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (hp < NPX && hx < HW_ && y >= 0 && y < p.H && x >= 0 && x < p.W) rx[i] = *reinterpret_cast<const float4*>(gx + ((size_t)y * p.W + x) * CIN + c4 * 4);
        }
    };
    const float* arow = dYt + (coh * 32 + li) * LDY + 8 * lh;
    const float* brow = Xt + (cih * 32 + li) * LDX + 8 * lh;
    if ((int)cx < ntiles) gload(cx);
    for (int t = cx; t < ntiles; t += gridDim.x) {
        __syncthreads();   // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < NDY; ++i) {          // transposed stores: [channel][pixel]
            const int idx = threadIdx.x + i * 256, c = ((idx >> 4) & 15) * 4, pp = ((idx >> 8) << 4) | (idx & 15);
            dYt[(c + 0) * LDY + pp] = rdy[i].x; dYt[(c + 1) * LDY + pp] = rdy[i].y; dYt[(c + 2) * LDY + pp] = rdy[i].z; dYt[(c + 3) * LDY + pp] = rdy[i].w;
        }
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int idx = threadIdx.x + i * 256, c = ((idx >> 4) & 15) * 4, hp = ((idx >> 8) << 4) | (idx & 15);
            if (hp < NPX) { Xt[(c + 0) * LDX + hp] = rx[i].x; Xt[(c + 1) * LDX + hp] = rx[i].y; Xt[(c + 2) * LDX + hp] = rx[i].z; Xt[(c + 3) * LDX + hp] = rx[i].w; }
        }
        __syncthreads();
        float4 na[2], nb[3];                  // raw values of the next (row, 16-pixel group)
        na[0] = *reinterpret_cast<const float4*>(arow); na[1] = *reinterpret_cast<const float4*>(arow + 4);
        nb[0] = *reinterpret_cast<const float4*>(brow); nb[1] = *reinterpret_cast<const float4*>(brow + 4); nb[2] = *reinterpret_cast<const float4*>(brow + 8);
#pragma unroll
        for (int it = 0; it < TH_ * 2; ++it) {
            // ---- planes of this group: dY (8 pixels of channel co) and X (12 pixels of channel ci)
            uint4 ah, am, al;
            x3_split(na[0], na[1], ah, am, al);
            const float v[12] = {nb[0].x, nb[0].y, nb[0].z, nb[0].w, nb[1].x, nb[1].y, nb[1].z, nb[1].w, nb[2].x, nb[2].y, nb[2].z, nb[2].w};
            uint32_t vh[NV], vm[NV], vl[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) x3_planes(v[k], vh[k], vm[k], vl[k]);
            uint32_t eh[NE], em[NE], el[NE], oh[NO], om[NO], ol[NO];       // pairs (2i, 2i+1) and (2i+1, 2i+2)
#pragma unroll
            for (int i = 0; i < NE; ++i) { eh[i] = X3_PK(vh[2 * i + 1], vh[2 * i]); em[i] = X3_PK(vm[2 * i + 1], vm[2 * i]); el[i] = X3_PK(vl[2 * i + 1], vl[2 * i]); }
#pragma unroll
            for (int i = 0; i < NO; ++i) { oh[i] = X3_PK(vh[2 * i + 2], vh[2 * i + 1]); om[i] = X3_PK(vm[2 * i + 2], vm[2 * i + 1]); ol[i] = X3_PK(vl[2 * i + 2], vl[2 * i + 1]); }
            if (it == TH_ && t + (int)gridDim.x < ntiles) gload(t + gridDim.x);      // next tile: global -> registers, under the second half's MFMAs
            // ---- raw values of the next group
            if (it + 1 < TH_ * 2) {
                const int rr = (it + 1) >> 1, g = (it + 1) & 1;
                const float* an = arow + rr * TW_ + g * 16;
                const float* bn = brow + rr * HWP + g * 16;
                na[0] = *reinterpret_cast<const float4*>(an); na[1] = *reinterpret_cast<const float4*>(an + 4);
                nb[0] = *reinterpret_cast<const float4*>(bn); nb[1] = *reinterpret_cast<const float4*>(bn + 4); nb[2] = *reinterpret_cast<const float4*>(bn + 8);
            }
#pragma unroll
            for (int kx = 0; kx < KS; ++kx) {
                const int o = kx >> 1;
                const uint4 bh = (kx & 1) ? make_uint4(oh[o], oh[o + 1], oh[o + 2], oh[o + 3]) : make_uint4(eh[o], eh[o + 1], eh[o + 2], eh[o + 3]);
                const uint4 bm = (kx & 1) ? make_uint4(om[o], om[o + 1], om[o + 2], om[o + 3]) : make_uint4(em[o], em[o + 1], em[o + 2], em[o + 3]);
                const uint4 bl = (kx & 1) ? make_uint4(ol[o], ol[o + 1], ol[o + 2], ol[o + 3]) : make_uint4(el[o], el[o + 1], el[o + 2], el[o + 3]);
                f32x16 a_ = acc[kx];
                a_ = X3_MFMA(al, bh, a_);
                a_ = X3_MFMA(ah, bl, a_);
                a_ = X3_MFMA(am, bm, a_);
                a_ = X3_MFMA(am, bh, a_);
                a_ = X3_MFMA(ah, bm, a_);
                a_ = X3_MFMA(ah, bh, a_);
                acc[kx] = a_;
            }
        }
    }
    // ---- partial slab [slab][ky*KS+kx][co][CIN] (conv_wgrad_kernel's layout: conv_wgrad_reduce_kernel sums the slabs)
    float* out = p.part + ((size_t)cx * KS * KS + ky * KS) * COUT * CIN;
#pragma unroll
    for (int kx = 0; kx < KS; ++kx)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = coh * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            out[((size_t)kx * COUT + co) * CIN + cih * 32 + li] = acc[kx][r];
        }
}
template <int KS>
static int conv_wgrad_x3_cfg(const WgradArgs& a, int nchunk, hipStream_t st) {
    constexpr int smem = (64 * WX_LDY + 64 * WxGeo<KS>::LDX) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        OCRL_HIP(hipFuncSetAttribute((const void*)conv_wgrad_x3_kernel<KS>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    const int pi = prof_begin(PROF_WGRAD, st);
    hipLaunchKernelGGL(conv_wgrad_x3_kernel<KS>, dim3(nchunk, KS), dim3(256), smem, st, a);
    prof_end(pi, st);
    OCRL_CHECK_LAUNCH("conv_wgrad_x3_kernel");
    return 0;
}
int conv_wgrad_x3_stage(const WgradArgs& a, int nchunk, hipStream_t st, int KS) {
    if (KS == 5) return conv_wgrad_x3_cfg<5>(a, nchunk, st);
    if (KS == 3) return conv_wgrad_x3_cfg<3>(a, nchunk, st);
    ocrl_set_error("conv wgrad x3: unsupported KS=%d", KS);
    return 1;
}

size_t conv_x3_pack_floats(int KS) { return (size_t)KS * KS * 4 * 2 * 3 * 64 * 4; }      // floats per pack (614 KB at 5x5)
int conv_pack_x3_launch(const float* W, float* fwd3, float* bwd3, hipStream_t st, int KS) {
    hipLaunchKernelGGL(conv_pack_x3_kernel, dim3(cdiv(KS * KS * 4 * 2 * 64, 256)), dim3(256), 0, st, W, reinterpret_cast<uint4*>(fwd3), reinterpret_cast<uint4*>(bwd3),
                       KS * KS);
    OCRL_CHECK_LAUNCH("conv_pack_x3");
    return 0;
}
template <int KS>
static int conv_x3_cfg(const ConvArgs& a, const float* pack3, hipStream_t st) {
    constexpr int smem = (X3_TH + KS - 1) * (X3_TW + KS - 1) * X3_LDP * 4;
    static_assert((X3_TH + KS - 1) * (X3_TW + KS - 1) * X3_LDP >= 4 * 32 * 68, "the halo region must hold the four epilogue patches");
    static bool attr_set = false;
    if (!attr_set) {
        OCRL_HIP(hipFuncSetAttribute((const void*)conv_x3_kernel<KS>, hipFuncAttributeMaxDynamicSharedMemorySize, smem));
        attr_set = true;
    }
    const int grid = cdiv(a.W, X3_TW) * cdiv(a.H, X3_TH) * a.B;
    const int pi = prof_begin(KS == 5 ? PROF_CONV5 : PROF_CONV_OTHER, st);
    hipLaunchKernelGGL(conv_x3_kernel<KS>, dim3(grid), dim3(256), smem, st, a, reinterpret_cast<const uint4*>(pack3));
    prof_end(pi, st);
    OCRL_CHECK_LAUNCH("conv_x3_kernel");
    return 0;
}
int conv_x3_launch(const ConvArgs& a, const float* pack3, hipStream_t st, int KS) {
    OCRL_REQUIRE(a.B > 0 && a.H > 0 && a.W > 0 && pack3, "conv x3: empty input / missing pack");
    OCRL_REQUIRE(((uintptr_t)a.X & 15) == 0 && ((uintptr_t)pack3 & 15) == 0, "conv x3: X / pack must be 16-byte aligned");
    if (KS == 5) return conv_x3_cfg<5>(a, pack3, st);
    if (KS == 3) return conv_x3_cfg<3>(a, pack3, st);
    ocrl_set_error("conv x3: unsupported KS=%d", KS);
    return 1;
}
