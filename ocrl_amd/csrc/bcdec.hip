// Spatial-broadcast decoder of the Slot-Attention configuration (reference: ocrs/common/models.py:110-141,
// ocrs/slate/slate_module.py:218-225).
//
// First layer without the broadcast tensor: the 5x5 / 192->64 convolution runs on  s[bk,:] + posmap[y,x,:]
// (spatially constant per slot plus a batch-independent map).  A convolution is linear, so
//     c1[bk,y,x,:] = relu( conv(posmap)[y,x,:] + sum_{taps valid at (y,x)} W_tap s[bk,:] + b )
// i.e. one tiny GEMM  M[bk, tap, :] = W_tap s  plus 25 border classes (5 row x 5 column classes of valid taps):
// 82 % of the configuration's forward FLOPs disappear.  The position-map term is folded through
// the 4-channel ramp grid: conv(posmap) = sum_taps (Wc[tap,:,0..3] . grid(y+dy,x+dx) + Wc[tap,:,4]).
// Layers 2-3 reuse conv.hip; the 64->4 output convolution and the slot mixture have their own kernels here.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

#define BT_H 4
#define BT_W 32

__device__ inline int bc_class(int y, int S) { return y < 2 ? y : (y >= S - 2 ? 4 - (S - 1 - y) : 2); }
// tap row ky (0..4) is inside the image for row class rc
__device__ inline bool bc_valid(int rc, int ky) { return rc == 0 ? ky >= 2 : rc == 1 ? ky >= 1 : rc == 3 ? ky <= 3 : rc == 4 ? ky <= 2 : true; }

// Wc[tap][co][0..3] = sum_ci W1[co][ci][tap] Wpos[ci][j];  Wc[tap][co][4] = sum_ci W1[co][ci][tap] bpos[ci]
// W1r[tap][co][ci] = W1[co][ci][tap]
__global__ void bc_compose_kernel(const float* __restrict__ W1, const float* __restrict__ Wpos, const float* __restrict__ bpos,
                                  float* __restrict__ Wc, float* __restrict__ W1r, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // (tap, co)
    if (i >= 25 * 64) return;
    const int co = i % 64, tap = i / 64;
    float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int ci = 0; ci < D; ++ci) {
        const float w = W1[((size_t)co * D + ci) * 25 + tap];
        W1r[(size_t)i * D + ci] = w;
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] += w * Wpos[ci * 4 + j];
        a[4] += w * bpos[ci];
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) Wc[i * 5 + j] = a[j];
}

__device__ inline void bc_grid(int y, int x, int S, float (&g)[5]) {
    const float east = S > 1 ? (float)x / (float)(S - 1) : 0.f, south = S > 1 ? (float)y / (float)(S - 1) : 0.f;
    g[0] = 1.f - south; g[1] = south; g[2] = 1.f - east; g[3] = east; g[4] = 1.f;
}

// P1[y][x][co] = sum over taps inside the image of Wc[tap][co][:] . (grid(y+dy, x+dx), 1)
__global__ void bc_posconv_kernel(const float* __restrict__ Wc, float* __restrict__ P1, int S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * S * 64) return;
    const int co = i % 64, x = (i / 64) % S, y = i / (64 * S);
    float a = 0.f;
    for (int ky = 0; ky < 5; ++ky)
        for (int kx = 0; kx < 5; ++kx) {
            const int yy = y + ky - 2, xx = x + kx - 2;
            if (yy < 0 || yy >= S || xx < 0 || xx >= S) continue;
            float g[5];
            bc_grid(yy, xx, S, g);
            const float* w = Wc + ((ky * 5 + kx) * 64 + co) * 5;
            a += w[0] * g[0] + w[1] * g[1] + w[2] * g[2] + w[3] * g[3] + w[4];
        }
    P1[i] = a;
}

// forward: Tc[bk][cls][co] = sum_{taps valid in cls} M[bk][tap][co];  backward (transpose): dM[bk][tap][co] = sum_{cls valid} dT
__global__ void bc_class_sum_kernel(const float* __restrict__ in, float* __restrict__ out, int BK, int forward) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BK * 25 * 64) return;
    const int co = i % 64, a = (i / 64) % 25;
    const long long bk = i / (64 * 25);
    float s = 0.f;
    for (int b = 0; b < 25; ++b) {
        const int cls = forward ? a : b, tap = forward ? b : a;
        if (bc_valid(cls / 5, tap / 5) && bc_valid(cls % 5, tap % 5)) s += in[(bk * 25 + b) * 64 + co];
    }
    out[i] = s;
}

// c1[bk][y][x][co] = relu(P1[y][x][co] + Tc[bk][cls(y,x)][co] + b1[co])
__global__ void bc_layer1_kernel(const float* __restrict__ P1, const float* __restrict__ Tc, const float* __restrict__ b1,
                                 float* __restrict__ c1, int BK, int S) {
    const long long n4 = (long long)BK * S * S * 16;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    const int c4 = i % 16;
    const long long pix = i / 16;
    const int x = pix % S, y = (pix / S) % S;
    const long long bk = pix / ((long long)S * S);
    const int cls = bc_class(y, S) * 5 + bc_class(x, S);
    const float4 p = *reinterpret_cast<const float4*>(P1 + ((size_t)y * S + x) * 64 + c4 * 4);
    const float4 t = *reinterpret_cast<const float4*>(Tc + (bk * 25 + cls) * 64 + c4 * 4);
    const float4 b = *reinterpret_cast<const float4*>(b1 + c4 * 4);
    *reinterpret_cast<float4*>(c1 + i * 4) = make_float4(fmaxf(p.x + t.x + b.x, 0.f), fmaxf(p.y + t.y + b.y, 0.f),
                                                        fmaxf(p.z + t.z + b.z, 0.f), fmaxf(p.w + t.w + b.w, 0.f));
}

// rowpart[bk][y][k][co] = sum over the pixels of row y in column class k of g[bk][y][x][co]   (one block per (bk, y));
// bc_layer1_reduce_kernel then sums the rows of each row class in row order (no float atomics: bitwise reproducible)
__global__ __launch_bounds__(256) void bc_layer1_bwd_kernel(const float* __restrict__ g, float* __restrict__ rowpart, int S) {
    __shared__ float red[4][5][64];
    const int y = blockIdx.x % S;
    const long long bk = blockIdx.x / S;
    const int co = threadIdx.x & 63, xl = threadIdx.x >> 6;
    float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const float* row = g + ((bk * S + y) * S) * 64;
    for (int x = xl; x < S; x += 4) {
        const float v = row[(size_t)x * 64 + co];
        const int cc = bc_class(x, S);
#pragma unroll
        for (int k = 0; k < 5; ++k) a[k] += (cc == k) ? v : 0.f;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) red[xl][k][co] = a[k];
    __syncthreads();
    if (xl == 0) {
#pragma unroll
        for (int k = 0; k < 5; ++k) rowpart[((bk * S + y) * 5 + k) * 64 + co] = (red[0][k][co] + red[1][k][co]) + (red[2][k][co] + red[3][k][co]);
    }
}
// dT[bk][rc*5+k][co] = sum over the rows y of row class rc of rowpart[bk][y][k][co]
__global__ void bc_layer1_reduce_kernel(const float* __restrict__ rowpart, float* __restrict__ dT, long long n, int S) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over [BK][25][64]
    if (i >= n) return;
    const int co = i & 63, cls = (i >> 6) % 25;
    const long long bk = i / (25 * 64);
    const int rc = cls / 5, k = cls - rc * 5;
    const int y0 = rc < 2 ? rc : (rc == 2 ? 2 : S - 2 + (rc - 3)), y1 = rc == 2 ? S - 2 : y0 + 1;
    float a = 0.f;
    for (int y = y0; y < y1; ++y) a += rowpart[((bk * S + y) * 5 + k) * 64 + co];
    dT[i] = a;
}

// dWc[tap][co][j] = sum_{y,x : tap inside} G[y][x][co] * (grid_j(y+dy, x+dx) | 1)     (one block per tap, G = sum over bk)
__global__ __launch_bounds__(256) void bc_posconv_bwd_kernel(const float* __restrict__ G, float* __restrict__ dWc, int S) {
    __shared__ float red[4][5][64];
    const int tap = blockIdx.x, ky = tap / 5, kx = tap % 5;
    const int co = threadIdx.x & 63, pl = threadIdx.x >> 6;
    float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    for (int pix = pl; pix < S * S; pix += 4) {
        const int x = pix % S, y = pix / S;
        const int yy = y + ky - 2, xx = x + kx - 2;
        if (yy < 0 || yy >= S || xx < 0 || xx >= S) continue;
        float gr[5];
        bc_grid(yy, xx, S, gr);
        const float v = G[(size_t)pix * 64 + co];
#pragma unroll
        for (int j = 0; j < 5; ++j) a[j] += v * gr[j];
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) red[pl][j][co] = a[j];
    __syncthreads();
    if (pl == 0)
#pragma unroll
        for (int j = 0; j < 5; ++j) dWc[(tap * 64 + co) * 5 + j] = red[0][j][co] + red[1][j][co] + red[2][j][co] + red[3][j][co];
}

// dW1[co][ci][tap] = dW1r[tap][co][ci] + sum_j dWc[tap][co][j] Wpos[ci][j] + dWc[tap][co][4] bpos[ci]
// dWpos[ci][j] = sum_{tap,co} dWc[tap][co][j] W1[co][ci][tap] ;  dbpos[ci] = sum_{tap,co} dWc[tap][co][4] W1[co][ci][tap]
__global__ void bc_compose_bwd_kernel(const float* __restrict__ W1, const float* __restrict__ Wpos, const float* __restrict__ bpos,
                                      const float* __restrict__ dWc, const float* __restrict__ dW1r, float* __restrict__ dW1,
                                      float* __restrict__ dWpos, float* __restrict__ dbpos, int D) {
    const int ci = blockIdx.x * blockDim.x + threadIdx.x;
    if (ci >= D) return;
    float dp[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    const float wp[5] = {Wpos[ci * 4], Wpos[ci * 4 + 1], Wpos[ci * 4 + 2], Wpos[ci * 4 + 3], bpos[ci]};
    for (int i = 0; i < 25 * 64; ++i) {
        const int co = i % 64, tap = i / 64;
        const float* d = dWc + i * 5;
        const size_t wi = ((size_t)co * D + ci) * 25 + tap;
        const float w = W1[wi];
        dW1[wi] = dW1r[(size_t)i * D + ci] + d[0] * wp[0] + d[1] * wp[1] + d[2] * wp[2] + d[3] * wp[3] + d[4] * wp[4];
#pragma unroll
        for (int j = 0; j < 5; ++j) dp[j] += d[j] * w;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) dWpos[ci * 4 + j] = dp[j];
    dbpos[ci] = dp[4];
}

// ------------------------------------------------------------------------------------------- 3x3 conv, 64 -> 4
// Wk[tap][ci][4] (forward) and Wb[tap][co][64] (backward-data, flipped taps) come from bc_c4_pack_kernel.
__global__ void bc_c4_pack_kernel(const float* __restrict__ W, float* __restrict__ Wk, float* __restrict__ Wb, int co_n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;      // (tap, ci)
    if (i >= 9 * 64) return;
    const int ci = i % 64, tap = i / 64;
    for (int co = 0; co < 4; ++co) {
        const float w = co < co_n ? W[((size_t)co * 64 + ci) * 9 + tap] : 0.f;
        Wk[i * 4 + co] = w;
        Wb[((8 - tap) * 4 + co) * 64 + ci] = w;
    }
}

template <int MODE>   // 0: forward (X [.,64] -> Y [.,4]);  1: backward data (dY [.,4] -> dX [.,64], masked by act > 0)
__global__ __launch_bounds__(128) void bc_c4_conv_kernel(const float* __restrict__ X, const float* __restrict__ Wt, const float* __restrict__ bias,
                                                         const float* __restrict__ act, float* __restrict__ Y, int Bn, int S, int elu) {
    // MODE 0 stages the 64 input channels in two passes of 32 (LD 36): 29 KB of LDS instead of 55 KB, so five workgroups (10 waves)
    // share a CU and cover each other's LDS latency -- this kernel is VALU / LDS work with one pixel per thread.
    constexpr int CI = MODE == 0 ? 64 : 4;
    constexpr int CP = MODE == 0 ? 32 : 4;              // channels per pass
    constexpr int LD = MODE == 0 ? 36 : 4;
    constexpr int HW_ = BT_W + 2, HH_ = BT_H + 2;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* halo = sm;                                   // [HH_][HW_][LD]
    float* stage = sm + HH_ * HW_ * LD;                 // MODE 1: [128][65] output transpose
    const int tiles_x = (S + BT_W - 1) / BT_W, tiles_y = (S + BT_H - 1) / BT_H;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y; bid /= tiles_y;
    const long long b = bid;
    const int x0 = tx * BT_W, y0 = ty * BT_H;
    auto load_halo = [&](int c0) {          // every global load is issued before the first LDS store: one memory round trip per pass
        constexpr int F4 = CP / 4, TOT = HH_ * HW_ * F4, NLD = (TOT + 127) / 128;
        const float* Xb = X + b * S * S * CI + c0;
        float4 hv[NLD];
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = threadIdx.x + i * 128;
            const int c4 = idx % F4, hp = idx / F4;
            const int x = x0 - 1 + hp % HW_, y = y0 - 1 + hp / HW_;
            hv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < TOT && y >= 0 && y < S && x >= 0 && x < S) hv[i] = *reinterpret_cast<const float4*>(Xb + ((size_t)y * S + x) * CI + c4 * 4);
        }
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = threadIdx.x + i * 128;
            if (idx < TOT) *reinterpret_cast<float4*>(halo + (idx / F4) * LD + (idx % F4) * 4) = hv[i];
        }
    };
    load_halo(0);
    __syncthreads();
    const int px = threadIdx.x % BT_W, py = threadIdx.x / BT_W;
    const int x = x0 + px, y = y0 + py;
    if (MODE == 0) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
#pragma unroll 1
        for (int c0 = 0; c0 < CI; c0 += CP) {
            if (c0) {
                __syncthreads();
                load_halo(c0);
                __syncthreads();
            }
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float* h = halo + ((py + tap / 3) * HW_ + px + tap % 3) * LD;
#pragma unroll
                for (int c4 = 0; c4 < CP / 4; ++c4) {
                    const float4 v = *reinterpret_cast<const float4*>(h + c4 * 4);
                    const float* w = Wt + (tap * 64 + c0 + c4 * 4) * 4;      // wave-uniform: scalar loads
                    a0 += v.x * w[0] + v.y * w[4] + v.z * w[8] + v.w * w[12];
                    a1 += v.x * w[1] + v.y * w[5] + v.z * w[9] + v.w * w[13];
                    a2 += v.x * w[2] + v.y * w[6] + v.z * w[10] + v.w * w[14];
                    a3 += v.x * w[3] + v.y * w[7] + v.z * w[11] + v.w * w[15];
                }
            }
        }
        if (x < S && y < S)
            *reinterpret_cast<float4*>(Y + ((b * S + y) * S + x) * 4) = make_float4(a0 + bias[0], a1 + bias[1], a2 + bias[2], a3 + bias[3]);
    } else {
        float acc[64];
#pragma unroll
        for (int c = 0; c < 64; ++c) acc[c] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float4 d = *reinterpret_cast<const float4*>(halo + ((py + tap / 3) * HW_ + px + tap % 3) * LD);
            const float* w = Wt + tap * 4 * 64;                     // Wb[tap'][co][ci], wave-uniform
#pragma unroll
            for (int c = 0; c < 64; ++c) acc[c] += d.x * w[c] + d.y * w[64 + c] + d.z * w[128 + c] + d.w * w[192 + c];
        }
#pragma unroll
        for (int c = 0; c < 64; ++c) stage[threadIdx.x * 65 + c] = acc[c];
        __syncthreads();
        // coalesced store of the 4x32 tile: 16 float4 per pixel; the 16 activation reads of a thread are issued together
        constexpr int NST = BT_H * BT_W * 16 / 128;
        float4 mv[NST];
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int idx = threadIdx.x + i * 128;
            const int c4 = idx % 16, p = idx / 16;
            const int xx = x0 + p % BT_W, yy = y0 + p / BT_W;
            mv[i] = make_float4(1.f, 1.f, 1.f, 1.f);
            if (xx < S && yy < S) mv[i] = *reinterpret_cast<const float4*>(act + ((b * S + yy) * S + xx) * 64 + c4 * 4);
        }
        const float e = elu ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            const int idx = threadIdx.x + i * 128;
            const int c4 = idx % 16, p = idx / 16;
            const int xx = x0 + p % BT_W, yy = y0 + p / BT_W;
            if (xx >= S || yy >= S) continue;
            const size_t o = ((b * S + yy) * S + xx) * 64 + c4 * 4;
            const float4 m = mv[i];
            const float* sp = stage + p * 65 + c4 * 4;
            // gate by the derivative of the saved activation: ReLU' (m > 0) or ELU' (m > 0 ? 1 : m + 1)
            *reinterpret_cast<float4*>(Y + o) = make_float4(m.x > 0.f ? sp[0] : e * (m.x + 1.f) * sp[0], m.y > 0.f ? sp[1] : e * (m.y + 1.f) * sp[1],
                                                            m.z > 0.f ? sp[2] : e * (m.z + 1.f) * sp[2], m.w > 0.f ? sp[3] : e * (m.w + 1.f) * sp[3]);
        }
    }
}

// dW[co][ci][tap] partials: part[blk][co][ci][tap]; each block loops over its tiles
__global__ __launch_bounds__(256) void bc_c4_wgrad_kernel(const float* __restrict__ X, const float* __restrict__ dY, float* __restrict__ part, int Bn, int S) {
    constexpr int HW_ = BT_W + 2, HH_ = BT_H + 2;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* halo = sm;                              // [HH_][HW_][64]
    float* ds = sm + HH_ * HW_ * 64;               // [128] float4
    const int tiles_x = (S + BT_W - 1) / BT_W, tiles_y = (S + BT_H - 1) / BT_H;
    const int ntiles = tiles_x * tiles_y * Bn;
    const int ci = threadIdx.x & 63, tq = threadIdx.x >> 6;
    float acc[3][4];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[t][c] = 0.f;
    // the next tile's halo (13 float4 per thread) and dY are fetched into registers while the current tile is being reduced
    constexpr int TOT = HH_ * HW_ * 16, NLD = (TOT + 255) / 256;
    float4 hv[NLD], dv;
    auto fetch = [&](int t) {
        int q = t;
        const int tx = q % tiles_x; q /= tiles_x;
        const int ty = q % tiles_y; q /= tiles_y;
        const long long b = q;
        const int x0 = tx * BT_W, y0 = ty * BT_H;
        const float* Xb = X + b * S * S * 64;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = threadIdx.x + i * 256;
            const int c4 = idx % 16, hp = idx / 16;
            const int x = x0 - 1 + hp % HW_, y = y0 - 1 + hp / HW_;
            hv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < TOT && y >= 0 && y < S && x >= 0 && x < S) hv[i] = *reinterpret_cast<const float4*>(Xb + ((size_t)y * S + x) * 64 + c4 * 4);
        }
        dv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (threadIdx.x < BT_H * BT_W) {
            const int x = x0 + threadIdx.x % BT_W, y = y0 + threadIdx.x / BT_W;
            if (x < S && y < S) dv = *reinterpret_cast<const float4*>(dY + ((b * S + y) * S + x) * 4);
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = threadIdx.x + i * 256;
            if (idx < TOT) *reinterpret_cast<float4*>(halo + (idx / 16) * 64 + (idx % 16) * 4) = hv[i];
        }
        if (threadIdx.x < BT_H * BT_W) *reinterpret_cast<float4*>(ds + threadIdx.x * 4) = dv;
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
#pragma unroll 8
        for (int p = 0; p < BT_H * BT_W; ++p) {          // unrolled: eight pixels' LDS reads in flight per thread
            const float4 d = *reinterpret_cast<const float4*>(ds + p * 4);
            const int px = p % BT_W, py = p / BT_W;
#pragma unroll
            for (int t3 = 0; t3 < 3; ++t3) {
                const int tap = tq + 4 * t3;
                if (tap < 9) {
                    const float v = halo[((py + tap / 3) * HW_ + px + tap % 3) * 64 + ci];
                    acc[t3][0] += d.x * v; acc[t3][1] += d.y * v; acc[t3][2] += d.z * v; acc[t3][3] += d.w * v;
                }
            }
        }
    }
    float* out = part + (size_t)blockIdx.x * 4 * 64 * 9;
#pragma unroll
    for (int t3 = 0; t3 < 3; ++t3) {
        const int tap = tq + 4 * t3;
        if (tap < 9)
#pragma unroll
            for (int co = 0; co < 4; ++co) out[(co * 64 + ci) * 9 + tap] = acc[t3][co];
    }
}

// ------------------------------------------------------------------------------------------- slot mixture + loss
// out4 [B*K][S][S][4] (rgb, mask logit) -> recon [B][S][S][4], loss partials, d out4 (gradient of sum (obs-recon)^2 / B)
__global__ __launch_bounds__(256) void bc_mix_kernel(const float* __restrict__ out4, const float* __restrict__ obs, float* __restrict__ recon,
                                                     float* __restrict__ dout4, float* __restrict__ part, int B, int K, int S, int C, float inv_b) {
    __shared__ float red[4];
    const long long n = (long long)B * S * S, hw = (long long)S * S;
    float loss = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / hw, r = i % hw;
        float4 o[16];
        float mx = -INFINITY;
        for (int k = 0; k < K; ++k) { o[k] = *reinterpret_cast<const float4*>(out4 + ((b * K + k) * hw + r) * 4); mx = fmaxf(mx, o[k].w); }
        float m[16], se = 0.f;
        for (int k = 0; k < K; ++k) { m[k] = __expf(o[k].w - mx); se += m[k]; }
        float rc[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < K; ++k) { m[k] /= se; rc[0] += o[k].x * m[k]; rc[1] += o[k].y * m[k]; rc[2] += o[k].z * m[k]; }
        float dr[3];
        for (int c = 0; c < 3; ++c) {
            const float e = rc[c] - (c < C ? obs[(b * C + c) * hw + r] : rc[c]);
            loss += e * e;
            dr[c] = 2.f * e * inv_b;
        }
        if (recon) *reinterpret_cast<float4*>(recon + i * 4) = make_float4(rc[0], rc[1], rc[2], 0.f);
        if (dout4) {
            float dm[16], dot = 0.f;
            for (int k = 0; k < K; ++k) { dm[k] = dr[0] * o[k].x + dr[1] * o[k].y + dr[2] * o[k].z; dot += m[k] * dm[k]; }
            for (int k = 0; k < K; ++k)
                *reinterpret_cast<float4*>(dout4 + ((b * K + k) * hw + r) * 4) = make_float4(dr[0] * m[k], dr[1] * m[k], dr[2] * m[k], m[k] * (dm[k] - dot));
        }
    }
    loss = wave_sum(loss);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = loss;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ================================================================== launchers
#define GRID1D(n) dim3(cdiv((n), 256)), dim3(256)

int bc_compose_launch(const float* W1, const float* Wpos, const float* bpos, float* Wc, float* W1r, int D, hipStream_t st) {
    hipLaunchKernelGGL(bc_compose_kernel, GRID1D(25 * 64), 0, st, W1, Wpos, bpos, Wc, W1r, D);
    OCRL_CHECK_LAUNCH("bc_compose");
    return 0;
}
int bc_posconv_launch(const float* Wc, float* P1, int S, hipStream_t st) {
    OCRL_REQUIRE(S >= 5, "broadcast decoder needs obs_size >= 5");
    hipLaunchKernelGGL(bc_posconv_kernel, GRID1D(S * S * 64), 0, st, Wc, P1, S);
    OCRL_CHECK_LAUNCH("bc_posconv");
    return 0;
}
int bc_class_sum_launch(const float* in, float* out, int BK, int forward, hipStream_t st) {
    hipLaunchKernelGGL(bc_class_sum_kernel, GRID1D((long long)BK * 25 * 64), 0, st, in, out, BK, forward);
    OCRL_CHECK_LAUNCH("bc_class_sum");
    return 0;
}
int bc_layer1_launch(const float* P1, const float* Tc, const float* b1, float* c1, int BK, int S, hipStream_t st) {
    hipLaunchKernelGGL(bc_layer1_kernel, GRID1D((long long)BK * S * S * 16), 0, st, P1, Tc, b1, c1, BK, S);
    OCRL_CHECK_LAUNCH("bc_layer1");
    return 0;
}
size_t bc_layer1_bwd_ws_floats(int BK, int S) { return (size_t)BK * S * 5 * 64; }
// dT [BK][25][64] is written (not accumulated); ws: bc_layer1_bwd_ws_floats() floats of scratch
int bc_layer1_bwd_launch(const float* g, float* dT, int BK, int S, float* ws, size_t ws_floats, hipStream_t st) {
    OCRL_REQUIRE(S >= 5 && ws && ws_floats >= bc_layer1_bwd_ws_floats(BK, S), "bc_layer1_bwd: S < 5 or scratch too small");
    hipLaunchKernelGGL(bc_layer1_bwd_kernel, dim3(BK * S), dim3(256), 0, st, g, ws, S);
    OCRL_CHECK_LAUNCH("bc_layer1_bwd");
    const long long n = (long long)BK * 25 * 64;
    hipLaunchKernelGGL(bc_layer1_reduce_kernel, dim3(cdiv(n, 256)), dim3(256), 0, st, ws, dT, n, S);
    OCRL_CHECK_LAUNCH("bc_layer1_reduce");
    return 0;
}
int bc_posconv_bwd_launch(const float* G, float* dWc, int S, hipStream_t st) {
    hipLaunchKernelGGL(bc_posconv_bwd_kernel, dim3(25), dim3(256), 0, st, G, dWc, S);
    OCRL_CHECK_LAUNCH("bc_posconv_bwd");
    return 0;
}
int bc_compose_bwd_launch(const float* W1, const float* Wpos, const float* bpos, const float* dWc, const float* dW1r, float* dW1,
                          float* dWpos, float* dbpos, int D, hipStream_t st) {
    hipLaunchKernelGGL(bc_compose_bwd_kernel, GRID1D(D), 0, st, W1, Wpos, bpos, dWc, dW1r, dW1, dWpos, dbpos, D);
    OCRL_CHECK_LAUNCH("bc_compose_bwd");
    return 0;
}
int bc_c4_pack_launch(const float* W, float* Wk, float* Wb, int co_n, hipStream_t st) {
    OCRL_REQUIRE(co_n <= 4, "output conv: at most 4 output channels");
    hipLaunchKernelGGL(bc_c4_pack_kernel, GRID1D(9 * 64), 0, st, W, Wk, Wb, co_n);
    OCRL_CHECK_LAUNCH("bc_c4_pack");
    return 0;
}
int bc_c4_fwd_launch(const float* X, const float* Wk, const float* bias4, float* Y, int Bn, int S, hipStream_t st) {
    const int grid = cdiv(S, BT_W) * cdiv(S, BT_H) * Bn;
    const int smem = (BT_H + 2) * (BT_W + 2) * 36 * 4;
    static bool set = false;
    if (!set) { OCRL_HIP(hipFuncSetAttribute((const void*)bc_c4_conv_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, smem)); set = true; }
    hipLaunchKernelGGL((bc_c4_conv_kernel<0>), dim3(grid), dim3(128), smem, st, X, Wk, bias4, nullptr, Y, Bn, S, 0);
    OCRL_CHECK_LAUNCH("bc_c4_fwd");
    return 0;
}
int bc_c4_bwd_data_launch(const float* dY, const float* Wb, const float* act, float* dX, int Bn, int S, hipStream_t st, int elu) {
    const int grid = cdiv(S, BT_W) * cdiv(S, BT_H) * Bn;
    const int smem = ((BT_H + 2) * (BT_W + 2) * 4 + 128 * 65) * 4;
    hipLaunchKernelGGL((bc_c4_conv_kernel<1>), dim3(grid), dim3(128), smem, st, dY, Wb, nullptr, act, dX, Bn, S, elu);
    OCRL_CHECK_LAUNCH("bc_c4_bwd_data");
    return 0;
}
// The same weight gradient on the matrix cores (round 3): re-indexed over the INPUT pixel, dW[co][ci][ky][kx] = sum_p X[p][ci] *
// dY[p - (ky-1, kx-1)][co] (dY zero outside the image), so an X tile needs no halo and every input element is read exactly once; the
// 36 (tap, co) pairs are the rows of three 16-row MFMA blocks, the 64 input channels four 16-column blocks, pixels the k dimension
// (v_mfma_f32_16x16x4_f32: a lane supplies dY of one (tap, co) at one of four pixels, and X of one channel at that pixel).  The VALU
// form above spends 6144 issue cycles per 128-pixel tile on dependent LDS reads and FMAs at two waves per SIMD (0.75 TB/s);
// this one 96 MFMAs per wave and tile.
__global__ __launch_bounds__(256, 4) void bc_c4_wgrad_mfma_kernel(const float* __restrict__ X, const float* __restrict__ dY, float* __restrict__ part, int Bn, int S) {
    constexpr int HW_ = BT_W + 2, HH_ = BT_H + 2, NPX = BT_H * BT_W, LDXS = 65;        // X tile [pixel][64 + 1]
    __shared__ float xs[NPX * LDXS];
    __shared__ __attribute__((aligned(16))) float dh[HH_ * HW_ * 4];                 // dY halo [hy][hx][4]
    const int tiles_x = (S + BT_W - 1) / BT_W, tiles_y = (S + BT_H - 1) / BT_H;
    const int ntiles = tiles_x * tiles_y * Bn;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r16 = lane & 15, g = lane >> 4;
    f32x4 acc[3][4];
#pragma unroll
    for (int mb = 0; mb < 3; ++mb)
#pragma unroll
        for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // per M block: this lane's (tap, co) row and its offset inside the dY halo relative to the pixel's (ly, lx)
    int aoff[3]; bool aval[3];
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) {
        const int i = mb * 16 + r16, tap = i >> 2, co = i & 3, ky = tap / 3, kx = tap - ky * 3;
        aval[mb] = i < 36;
        aoff[mb] = aval[mb] ? ((2 - ky) * HW_ + (2 - kx)) * 4 + co : 0;
    }
    constexpr int NLD = NPX * 16 / 256;            // float4 of the X tile per thread
    constexpr int NDH = (HH_ * HW_ + 255) / 256;
    float4 xv[NLD], dv[NDH];
    auto fetch = [&](int t) {
        int q = t;
        const int tx = q % tiles_x; q /= tiles_x;
        const int ty = q % tiles_y; q /= tiles_y;
        const long long b = q;
        const int x0 = tx * BT_W, y0 = ty * BT_H;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = threadIdx.x + i * 256, c4 = idx & 15, pp = idx >> 4;
            const int x = x0 + pp % BT_W, y = y0 + pp / BT_W;
            xv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (x < S && y < S) xv[i] = *reinterpret_cast<const float4*>(X + ((b * S + y) * S + x) * 64 + c4 * 4);
        }
#pragma unroll
        for (int i = 0; i < NDH; ++i) {
            const int hp = threadIdx.x + i * 256;
            const int x = x0 - 1 + hp % HW_, y = y0 - 1 + hp / HW_;
            dv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (hp < HH_ * HW_ && x >= 0 && x < S && y >= 0 && y < S) dv[i] = *reinterpret_cast<const float4*>(dY + ((b * S + y) * S + x) * 4);
        }
    };
    if ((int)blockIdx.x < ntiles) fetch(blockIdx.x);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int idx = threadIdx.x + i * 256, c = (idx & 15) * 4, pp = idx >> 4;
            xs[pp * LDXS + c] = xv[i].x; xs[pp * LDXS + c + 1] = xv[i].y; xs[pp * LDXS + c + 2] = xv[i].z; xs[pp * LDXS + c + 3] = xv[i].w;
        }
#pragma unroll
        for (int i = 0; i < NDH; ++i) {
            const int hp = threadIdx.x + i * 256;
            if (hp < HH_ * HW_) *reinterpret_cast<float4*>(dh + hp * 4) = dv[i];
        }
        __syncthreads();
        if (t + (int)gridDim.x < ntiles) fetch(t + gridDim.x);
        // this wave's 32 pixels (one tile row), four at a time
#pragma unroll
        for (int s = 0; s < BT_W / 4; ++s) {
            const int lx = s * 4 + g, pp = wave * BT_W + lx;
            const float* dp = dh + (wave * HW_ + lx) * 4;
            float av[3], bv[4];
#pragma unroll
            for (int mb = 0; mb < 3; ++mb) av[mb] = aval[mb] ? dp[aoff[mb]] : 0.f;
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) bv[nb] = xs[pp * LDXS + nb * 16 + r16];
#pragma unroll
            for (int mb = 0; mb < 3; ++mb)
#pragma unroll
                for (int nb = 0; nb < 4; ++nb) acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[mb], bv[nb], acc[mb][nb], 0, 0, 0);
        }
    }
    // ---- sum the four waves' accumulators in a fixed order and write this block's partial [co][ci][tap]
    __syncthreads();
    float* red = xs;                               // [4 waves][16 rows][64 cols], one M block at a time
    static_assert(4 * 16 * 64 <= NPX * LDXS, "reduction scratch must fit the X tile region");
#pragma unroll
    for (int mb = 0; mb < 3; ++mb) {
        if (mb) __syncthreads();
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[(wave * 16 + 4 * g + r) * 64 + nb * 16 + r16] = acc[mb][nb][r];
        __syncthreads();
        for (int e = threadIdx.x; e < 16 * 64; e += 256) {
            const int row = e >> 6, ci = e & 63, i = mb * 16 + row;
            if (i < 36) {
                const float v = ((red[(0 * 16 + row) * 64 + ci] + red[(1 * 16 + row) * 64 + ci]) + red[(2 * 16 + row) * 64 + ci]) + red[(3 * 16 + row) * 64 + ci];
                part[(size_t)blockIdx.x * 4 * 64 * 9 + ((i & 3) * 64 + ci) * 9 + (i >> 2)] = v;
            }
        }
    }
}

int bc_c4_wgrad_blocks(int Bn, int S) {
    const int nt = cdiv(S, BT_W) * cdiv(S, BT_H) * Bn;
    return nt < 1024 ? nt : 1024;
}
int bc_c4_wgrad_launch(const float* X, const float* dY, float* part, int Bn, int S, hipStream_t st) {
    static int form = -1;
    if (form < 0) { const char* e = getenv("OCRL_BC_WGRAD"); form = e ? atoi(e) : 2; }       // 1: the VALU form
    if (form != 1) {
        hipLaunchKernelGGL(bc_c4_wgrad_mfma_kernel, dim3(bc_c4_wgrad_blocks(Bn, S)), dim3(256), 0, st, X, dY, part, Bn, S);
        OCRL_CHECK_LAUNCH("bc_c4_wgrad_mfma");
        return 0;
    }
    const int smem = ((BT_H + 2) * (BT_W + 2) * 64 + 128 * 4) * 4;
    static bool set = false;
    if (!set) { OCRL_HIP(hipFuncSetAttribute((const void*)bc_c4_wgrad_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem)); set = true; }
    hipLaunchKernelGGL(bc_c4_wgrad_kernel, dim3(bc_c4_wgrad_blocks(Bn, S)), dim3(256), smem, st, X, dY, part, Bn, S);
    OCRL_CHECK_LAUNCH("bc_c4_wgrad");
    return 0;
}
int bc_mix_launch(const float* out4, const float* obs, float* recon, float* dout4, float* loss_out, int B, int K, int S, int C, float* ws,
                  size_t ws_floats, hipStream_t st) {
    OCRL_REQUIRE(K <= 16 && C <= 3 && ws_floats >= 1024, "mixture: K <= 16, C <= 3");
    hipLaunchKernelGGL(bc_mix_kernel, dim3(1024), dim3(256), 0, st, out4, obs, recon, dout4, ws, B, K, S, C, 1.0f / B);
    OCRL_CHECK_LAUNCH("bc_mix");
    return reduce_partials_launch(ws, 1024, loss_out, 1.0f / B, 0, st);
}
