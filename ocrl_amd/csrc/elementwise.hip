// Bandwidth-bound helper kernels of the SLATE step (layout changes, LayerNorm, reductions,
// Gumbel-softmax, cross-entropy, embedding, dropout, causal softmax, cross-attention).
// All fp32; rows are processed by whole waves / workgroups with coalesced float4 access.
#include "common.h"
#include "kernels.h"

#define TINYF 1.17549435e-38f

__device__ inline float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < nw; ++i) s += red[i];
    return s;
}
__device__ inline float block_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = red[0];
    for (int i = 1; i < nw; ++i) s = fmaxf(s, red[i]);
    return s;
}

// ------------------------------------------------------------------ layout kernels
// obs [B,C,H,W] -> out [B,H,W,8] (channels >= C are zero)
__global__ void nchw_to_nhwc8_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int H, int W) {
    const long long n = (long long)B * H * W;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const long long hw = (long long)H * W;
    const long long b = i / hw, r = i % hw;
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = c < C ? in[(b * C + c) * hw + r] : 0.f;
    float4* o = reinterpret_cast<float4*>(out + i * 8);
    o[0] = make_float4(v[0], v[1], v[2], v[3]);
    o[1] = make_float4(v[4], v[5], v[6], v[7]);
}

// obs [B,C,S,S] -> out [B*(S/4)^2, C*16], k = c*16 + ky*4 + kx  (matches W[64,C,4,4] flattened)
__global__ void patchify4_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int C, int S) {
    const int E = S / 4;
    const long long n = (long long)B * E * E * C * 4;   // one thread per (row, c, ky): 4 contiguous kx
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ky = i & 3;
    const int c = (i >> 2) % C;
    const long long row = (i >> 2) / C;
    const int ex = row % E, ey = (row / E) % E;
    const long long b = row / ((long long)E * E);
    const float4 v = *reinterpret_cast<const float4*>(in + ((b * C + c) * S + (ey * 4 + ky)) * S + ex * 4);
    *reinterpret_cast<float4*>(out + row * (C * 16) + c * 16 + ky * 4) = v;
}

// PixelShuffle(2) in NHWC: out[b, 2h+i, 2w+j, c] = in[b, h, w, 4c + 2i + j]; forward==0 runs the inverse map.
// One thread per (b, h, w, c): the four sub-pixels of a channel are one float4 of the unshuffled tensor, and consecutive
// threads (channels) touch consecutive floats of each shuffled pixel, so both sides move in full cache lines.
__global__ void pixel_shuffle_kernel(const float* __restrict__ in, float* __restrict__ out, int B, int h, int w, int Cout, int forward,
                                     const float* __restrict__ mask) {
    const long long n = (long long)B * h * w * Cout;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = i % Cout;
    long long r = i / Cout;
    const int x = r % w; r /= w;
    const int y = r % h;
    const long long b = r / h;
    const long long u = i * 4;                                                      // unshuffled [b,h,w,4c..4c+3]
    const long long s00 = ((b * 2 * h + 2 * y) * (2 * w) + 2 * x) * Cout + c;       // shuffled (2y, 2x)
    const long long s01 = s00 + Cout, s10 = s00 + (long long)2 * w * Cout, s11 = s10 + Cout;
    if (forward) {
        const float4 v = *reinterpret_cast<const float4*>(in + u);
        out[s00] = v.x; out[s01] = v.y; out[s10] = v.z; out[s11] = v.w;
    } else {          // backward: optional ReLU mask on the unshuffled tensor
        float4 v = make_float4(in[s00], in[s01], in[s10], in[s11]);
        if (mask) {
            const float4 m = *reinterpret_cast<const float4*>(mask + u);
            v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        }
        *reinterpret_cast<float4*>(out + u) = v;
    }
}

// ------------------------------------------------------------------ LayerNorm (eps 1e-5, biased variance)
template <int NPL>   // F = 64 * NPL, one wave per row
__global__ void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ bta,
                                     float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, long long R) {
    constexpr int F = 64 * NPL;
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= R) return;
    float v[NPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) { v[i] = x[row * F + i * 64 + lane]; s += v[i]; }
    const float mu = wave_sum(s) * (1.0f / F);
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NPL; ++i) { const float d = v[i] - mu; q += d * d; }
    const float rs = rsqrtf(wave_sum(q) * (1.0f / F) + 1e-5f);
#pragma unroll
    for (int i = 0; i < NPL; ++i) {
        const int c = i * 64 + lane;
        y[row * F + c] = (v[i] - mu) * rs * g[c] + bta[c];
    }
    if (lane == 0) {
        if (mean) mean[row] = mu;
        if (rstd) rstd[row] = rs;
    }
}

// dx = rstd * (dy*g - mean(dy*g) - xhat * mean(dy*g*xhat)); block partials of dgamma/dbeta -> part[blk][2F]
template <int NPL>
__global__ void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                     const float* __restrict__ rstd, const float* __restrict__ g, float* __restrict__ dx,
                                     float* __restrict__ part, long long R, int accumulate) {
    constexpr int F = 64 * NPL;
    __shared__ float red[4][2 * F];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float dg[NPL], db[NPL], gg[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) { dg[i] = 0.f; db[i] = 0.f; gg[i] = g[i * 64 + lane]; }
    for (long long row = (long long)blockIdx.x * 4 + wv; row < R; row += (long long)gridDim.x * 4) {
        const float mu = mean[row], rs = rstd[row];
        float xh[NPL], d[NPL];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int c = i * 64 + lane;
            xh[i] = (x[row * F + c] - mu) * rs;
            const float dyv = dy[row * F + c];
            dg[i] += dyv * xh[i];
            db[i] += dyv;
            d[i] = dyv * gg[i];
            s1 += d[i];
            s2 += d[i] * xh[i];
        }
        s1 = wave_sum(s1) * (1.0f / F);
        s2 = wave_sum(s2) * (1.0f / F);
#pragma unroll
        for (int i = 0; i < NPL; ++i) {
            const int c = i * 64 + lane;
            const float v = rs * (d[i] - s1 - xh[i] * s2);
            dx[row * F + c] = accumulate ? dx[row * F + c] + v : v;
        }
    }
#pragma unroll
    for (int i = 0; i < NPL; ++i) { red[wv][i * 64 + lane] = dg[i]; red[wv][F + i * 64 + lane] = db[i]; }
    __syncthreads();
    for (int c = threadIdx.x; c < 2 * F; c += blockDim.x)
        part[(size_t)blockIdx.x * 2 * F + c] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
}

// 64-wide rows (the slot-attention input norm over B*H*W positions): one wave per row meant one 4-byte load per lane and two full wave
// reductions per 256 bytes -- latency-bound at 2.4 TB/s.  Here 16 lanes own a row (float4 each), a wave works on 4 rows per pass and
// has 4 passes (16 rows) in flight; the reductions are four xor-shuffles inside the 16-lane group.
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__global__ __launch_bounds__(256) void layernorm_fwd64_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ bta,
                                                              float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, long long R) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c4 = lane & 15, rsub = lane >> 4;
    const long long row0 = ((long long)blockIdx.x * 4 + wv) * 16 + rsub;
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long row = row0 + 4 * i;
        v[i] = row < R ? *reinterpret_cast<const float4*>(x + row * 64 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float4 gg = *reinterpret_cast<const float4*>(g + c4 * 4), bb = *reinterpret_cast<const float4*>(bta + c4 * 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const long long row = row0 + 4 * i;
        const float mu = group16_sum((v[i].x + v[i].y) + (v[i].z + v[i].w)) * (1.0f / 64);
        const float4 dd = make_float4(v[i].x - mu, v[i].y - mu, v[i].z - mu, v[i].w - mu);
        const float rs = rsqrtf(group16_sum((dd.x * dd.x + dd.y * dd.y) + (dd.z * dd.z + dd.w * dd.w)) * (1.0f / 64) + 1e-5f);
        if (row < R) {
            *reinterpret_cast<float4*>(y + row * 64 + c4 * 4) =
                make_float4(dd.x * rs * gg.x + bb.x, dd.y * rs * gg.y + bb.y, dd.z * rs * gg.z + bb.z, dd.w * rs * gg.w + bb.w);
            if (c4 == 0) {
                if (mean) mean[row] = mu;
                if (rstd) rstd[row] = rs;
            }
        }
    }
}
// backward, same layout; a block walks 64-row groups with stride gridDim.x and leaves its dgamma | dbeta partial in part[blk][128]
__global__ __launch_bounds__(256) void layernorm_bwd64_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, const float* __restrict__ g, float* __restrict__ dx,
                                                              float* __restrict__ part, long long R, int accumulate) {
    __shared__ float4 red[2][16][16];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c4 = lane & 15, rsub = lane >> 4;
    const float4 gg = *reinterpret_cast<const float4*>(g + c4 * 4);
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
    for (long long grp = blockIdx.x; grp * 64 < R; grp += gridDim.x) {
        const long long row0 = (grp * 4 + wv) * 16 + rsub;
        float4 xv[4], yv[4];
        float mu[4], rs[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long row = row0 + 4 * i;
            const bool ok = row < R;
            xv[i] = ok ? *reinterpret_cast<const float4*>(x + row * 64 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            yv[i] = ok ? *reinterpret_cast<const float4*>(dy + row * 64 + c4 * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            mu[i] = ok ? mean[row] : 0.f;
            rs[i] = ok ? rstd[row] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long row = row0 + 4 * i;
            const float4 xh = make_float4((xv[i].x - mu[i]) * rs[i], (xv[i].y - mu[i]) * rs[i], (xv[i].z - mu[i]) * rs[i], (xv[i].w - mu[i]) * rs[i]);
            dg.x += yv[i].x * xh.x; dg.y += yv[i].y * xh.y; dg.z += yv[i].z * xh.z; dg.w += yv[i].w * xh.w;
            db.x += yv[i].x; db.y += yv[i].y; db.z += yv[i].z; db.w += yv[i].w;
            const float4 d4 = make_float4(yv[i].x * gg.x, yv[i].y * gg.y, yv[i].z * gg.z, yv[i].w * gg.w);
            const float s1 = group16_sum((d4.x + d4.y) + (d4.z + d4.w)) * (1.0f / 64);
            const float s2 = group16_sum((d4.x * xh.x + d4.y * xh.y) + (d4.z * xh.z + d4.w * xh.w)) * (1.0f / 64);
            if (row < R) {
                float4 o = make_float4(rs[i] * (d4.x - s1 - xh.x * s2), rs[i] * (d4.y - s1 - xh.y * s2), rs[i] * (d4.z - s1 - xh.z * s2),
                                       rs[i] * (d4.w - s1 - xh.w * s2));
                float* dp = dx + row * 64 + c4 * 4;
                if (accumulate) { const float4 a = *reinterpret_cast<const float4*>(dp); o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; }
                *reinterpret_cast<float4*>(dp) = o;
            }
        }
    }
    red[0][wv * 4 + rsub][c4] = dg;
    red[1][wv * 4 + rsub][c4] = db;
    __syncthreads();
    if (threadIdx.x < 32) {
        const int which = threadIdx.x >> 4, c = threadIdx.x & 15;
        float4 a = red[which][0][c];
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 t = red[which][k][c]; a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
        *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * 128 + which * 64 + c * 4) = a;
    }
}

// 192-wide rows (the transformer decoder's d_model): the one-wave-per-row kernels keep one row of 4-byte loads in flight per wave and
// need every wave slot of the machine to reach HBM speed -- alone they do (4.7-5.3 TB/s), but on the step's main stream they share the
// CUs with the weight-gradient and dVAE streams and ran 2-4x longer (layernorm_bwd 243 vs 64 us, 13 launches per step on the critical
// path).  Here 16 lanes own a row (three float4 each), a wave works on 4 rows per pass and keeps 2 passes (8 rows, 6 KB) in flight.
__global__ __launch_bounds__(256) void layernorm_fwd192_kernel(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ bta,
                                                               float* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd, long long R) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c4 = lane & 15, rsub = lane >> 4;
    const long long row0 = ((long long)blockIdx.x * 4 + wv) * 8 + rsub;
    float4 v[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const long long row = row0 + 4 * i;
            v[i][k] = row < R ? *reinterpret_cast<const float4*>(x + row * 192 + (k * 16 + c4) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    float4 gg[3], bb[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { gg[k] = *reinterpret_cast<const float4*>(g + (k * 16 + c4) * 4); bb[k] = *reinterpret_cast<const float4*>(bta + (k * 16 + c4) * 4); }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const long long row = row0 + 4 * i;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) s += (v[i][k].x + v[i][k].y) + (v[i][k].z + v[i][k].w);
        const float mu = group16_sum(s) * (1.0f / 192);
        float q = 0.f;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            v[i][k].x -= mu; v[i][k].y -= mu; v[i][k].z -= mu; v[i][k].w -= mu;
            q += (v[i][k].x * v[i][k].x + v[i][k].y * v[i][k].y) + (v[i][k].z * v[i][k].z + v[i][k].w * v[i][k].w);
        }
        const float rs = rsqrtf(group16_sum(q) * (1.0f / 192) + 1e-5f);
        if (row < R) {
#pragma unroll
            for (int k = 0; k < 3; ++k)
                *reinterpret_cast<float4*>(y + row * 192 + (k * 16 + c4) * 4) = make_float4(v[i][k].x * rs * gg[k].x + bb[k].x, v[i][k].y * rs * gg[k].y + bb[k].y,
                                                                                            v[i][k].z * rs * gg[k].z + bb[k].z, v[i][k].w * rs * gg[k].w + bb[k].w);
            if (c4 == 0) { if (mean) mean[row] = mu; if (rstd) rstd[row] = rs; }
        }
    }
}
__global__ __launch_bounds__(256) void layernorm_bwd192_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                                               const float* __restrict__ rstd, const float* __restrict__ g, float* __restrict__ dx,
                                                               float* __restrict__ part, long long R, int accumulate) {
    __shared__ float4 red[2][16][48];            // [dgamma | dbeta][wave * 4 + row group][float4 column]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c4 = lane & 15, rsub = lane >> 4;
    float4 gg[3], dg[3], db[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { gg[k] = *reinterpret_cast<const float4*>(g + (k * 16 + c4) * 4); dg[k] = make_float4(0.f, 0.f, 0.f, 0.f); db[k] = dg[k]; }
    for (long long grp = blockIdx.x; grp * 32 < R; grp += gridDim.x) {         // 32 rows per workgroup and pass
        const long long row0 = (grp * 4 + wv) * 8 + rsub;
        float4 xv[2][3], yv[2][3];
        float mu[2], rs[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long row = row0 + 4 * i;
            const bool ok = row < R;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                xv[i][k] = ok ? *reinterpret_cast<const float4*>(x + row * 192 + (k * 16 + c4) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                yv[i][k] = ok ? *reinterpret_cast<const float4*>(dy + row * 192 + (k * 16 + c4) * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            mu[i] = ok ? mean[row] : 0.f;
            rs[i] = ok ? rstd[row] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long long row = row0 + 4 * i;
            float4 xh[3], d4[3];
            float s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                xh[k] = make_float4((xv[i][k].x - mu[i]) * rs[i], (xv[i][k].y - mu[i]) * rs[i], (xv[i][k].z - mu[i]) * rs[i], (xv[i][k].w - mu[i]) * rs[i]);
                const float4 yy = yv[i][k];
                dg[k].x += yy.x * xh[k].x; dg[k].y += yy.y * xh[k].y; dg[k].z += yy.z * xh[k].z; dg[k].w += yy.w * xh[k].w;
                db[k].x += yy.x; db[k].y += yy.y; db[k].z += yy.z; db[k].w += yy.w;
                d4[k] = make_float4(yy.x * gg[k].x, yy.y * gg[k].y, yy.z * gg[k].z, yy.w * gg[k].w);
                s1 += (d4[k].x + d4[k].y) + (d4[k].z + d4[k].w);
                s2 += (d4[k].x * xh[k].x + d4[k].y * xh[k].y) + (d4[k].z * xh[k].z + d4[k].w * xh[k].w);
            }
            s1 = group16_sum(s1) * (1.0f / 192);
            s2 = group16_sum(s2) * (1.0f / 192);
            if (row < R) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float4 o = make_float4(rs[i] * (d4[k].x - s1 - xh[k].x * s2), rs[i] * (d4[k].y - s1 - xh[k].y * s2), rs[i] * (d4[k].z - s1 - xh[k].z * s2),
                                           rs[i] * (d4[k].w - s1 - xh[k].w * s2));
                    float* dp = dx + row * 192 + (k * 16 + c4) * 4;
                    if (accumulate) { const float4 a = *reinterpret_cast<const float4*>(dp); o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; }
                    *reinterpret_cast<float4*>(dp) = o;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { red[0][wv * 4 + rsub][k * 16 + c4] = dg[k]; red[1][wv * 4 + rsub][k * 16 + c4] = db[k]; }
    __syncthreads();
    if (threadIdx.x < 96) {                      // fixed order over the 16 (wave, row group) partials
        const int which = threadIdx.x / 48, c = threadIdx.x % 48;
        float4 a = red[which][0][c];
#pragma unroll
        for (int k = 1; k < 16; ++k) { const float4 t = red[which][k][c]; a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
        *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * 384 + which * 192 + c * 4) = a;
    }
}

// ------------------------------------------------------------------ column sums (bias gradients etc.)
// part[chunk][F] = sum over the chunk's rows of X[r][f]
__global__ void colsum_kernel(const float* __restrict__ X, long long ld, float* __restrict__ part, long long R, int F, long long rows_per_chunk) {
    __shared__ float red[4][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const long long r0 = (long long)blockIdx.y * rows_per_chunk;
    const long long r1 = min(R, r0 + rows_per_chunk);
    float s = 0.f;
    if (col < F)
        for (long long r = r0 + rl; r < r1; r += 4) s += X[r * ld + col];
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && col < F) part[(size_t)blockIdx.y * F + col] = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}
// float4 variant (F, ld multiples of 4; F <= 1024): a block owns whole rows of its chunk — thread = (row lane, float4 column),
// four independent row streams per thread keep enough loads in flight to run at HBM speed.
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ X, long long ld, float* __restrict__ part, long long R, int F,
                                                      long long rows_per_chunk) {
    __shared__ float4 red[256];
    const int F4 = F >> 2, RL = 256 / F4;            // row lanes per block
    const int c4 = threadIdx.x % F4, rl = threadIdx.x / F4;
    const long long r0 = (long long)blockIdx.x * rows_per_chunk;
    const long long r1 = min(R, r0 + rows_per_chunk);
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0, s2 = s0, s3 = s0;
    if (rl < RL) {
        const float* base = X + (size_t)c4 * 4;
        long long r = r0 + rl;
        for (; r + 3 * RL < r1; r += 4 * RL) {
            const float4 a = *reinterpret_cast<const float4*>(base + r * ld), b = *reinterpret_cast<const float4*>(base + (r + RL) * ld);
            const float4 c = *reinterpret_cast<const float4*>(base + (r + 2 * RL) * ld), d = *reinterpret_cast<const float4*>(base + (r + 3 * RL) * ld);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
            s1.x += b.x; s1.y += b.y; s1.z += b.z; s1.w += b.w;
            s2.x += c.x; s2.y += c.y; s2.z += c.z; s2.w += c.w;
            s3.x += d.x; s3.y += d.y; s3.z += d.z; s3.w += d.w;
        }
        for (; r < r1; r += RL) {
            const float4 a = *reinterpret_cast<const float4*>(base + r * ld);
            s0.x += a.x; s0.y += a.y; s0.z += a.z; s0.w += a.w;
        }
    }
    red[threadIdx.x] = make_float4(s0.x + s1.x + s2.x + s3.x, s0.y + s1.y + s2.y + s3.y, s0.z + s1.z + s2.z + s3.z, s0.w + s1.w + s2.w + s3.w);
    __syncthreads();
    if (threadIdx.x < F4) {
        float4 t = red[threadIdx.x];
        for (int k = 1; k < RL; ++k) { const float4 v = red[k * F4 + threadIdx.x]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
        *reinterpret_cast<float4*>(part + (size_t)blockIdx.x * F + threadIdx.x * 4) = t;
    }
}
// out[c] (+)= scale * sum_k part[k][c]: 16 columns x 16 partial-row lanes per 256-thread block (the partials are read in parallel, not
// as one serial chain per column; a 1024-thread block of 64 columns needed sixteen free wave slots on one CU and waited for them 3x its
// own run time when other streams' kernels held the machine)
__global__ __launch_bounds__(256) void colsum_final_kernel(const float* __restrict__ part, float* __restrict__ out, int nchunk, int F, int accumulate, float scale) {
    __shared__ float red[16][16];
    const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    float s0 = 0.f, s1 = 0.f;
    if (c < F) {
        int k = rl;
        for (; k + 16 < nchunk; k += 32) { s0 += part[(size_t)k * F + c]; s1 += part[(size_t)(k + 16) * F + c]; }
        if (k < nchunk) s0 += part[(size_t)k * F + c];
    }
    red[rl][cl] = s0 + s1;
    __syncthreads();
    if (rl == 0 && c < F) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        t *= scale;
        out[c] = accumulate ? out[c] + t : t;
    }
}

// out[0] (+)= scale * sum(part[0..n))   — single block, deterministic
__global__ void reduce_partials_kernel(const float* __restrict__ part, int n, float* __restrict__ out, float scale, int accumulate) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) s += part[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = (accumulate ? out[0] : 0.f) + s * scale;
}

// ------------------------------------------------------------------ dVAE reconstruction loss
// obs [B,C,H,W] vs recon [B,H,W,4]: part[blk] = sum (obs-recon)^2 ; drecon = 2*(recon-obs)*inv_b (pad channel 0)
__global__ void mse_kernel(const float* __restrict__ obs, const float* __restrict__ recon, float* __restrict__ drecon,
                           float* __restrict__ part, int B, int C, int H, int W, float inv_b) {
    __shared__ float red[4];
    const long long n = (long long)B * H * W, hw = (long long)H * W;
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long b = i / hw, r = i % hw;
        const float4 rc = *reinterpret_cast<const float4*>(recon + i * 4);
        const float rv[4] = {rc.x, rc.y, rc.z, rc.w};
        float d[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < C; ++c) {
            const float e = rv[c] - obs[(b * C + c) * hw + r];
            s += e * e;
            d[c] = 2.f * e * inv_b;
        }
        if (drecon) *reinterpret_cast<float4*>(drecon + i * 4) = make_float4(d[0], d[1], d[2], d[3]);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

// ------------------------------------------------------------------ Gumbel softmax over the vocabulary
// One workgroup per row of V logits.  logp = log_softmax(raw); z = softmax((logp + g1)/tau);
// tok = argmax(logp + g2);  g = -log(e + tiny), e ~ Exp(1) injected (e1/e2) or drawn on device.
#define GS_MAXPT 16
template <int NPT>      // values per thread = V / 256 (compile-time for the common vocabulary sizes: no per-value guards); 0 = runtime
__global__ __launch_bounds__(256) void gumbel_softmax_kernel(const float* __restrict__ raw, const float* __restrict__ e1,
                                                             const float* __restrict__ e2, float* __restrict__ z,
                                                             int* __restrict__ tokens, int V, float inv_tau,
                                                             unsigned long long seed, float* __restrict__ zst) {
    __shared__ float red[4];
    __shared__ float redv[4];
    __shared__ int redi[4];
    const long long row = blockIdx.x;
    const float* r = raw + row * V;
    const int npt = NPT ? NPT : V / 256;     // V % 256 == 0, V <= 4096
    float x[GS_MAXPT];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < GS_MAXPT; ++i)
        if (i < npt) { x[i] = r[i * 256 + threadIdx.x]; mx = fmaxf(mx, x[i]); }
    mx = block_max(mx, red);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < GS_MAXPT; ++i)
        if (i < npt) s += __expf(x[i] - mx);
    s = block_sum(s, red);
    const float lse = mx + __logf(s);
    // soft sample
    float u[GS_MAXPT];
    float best = -INFINITY;
    int besti = 0;
    float m2 = -INFINITY;
#pragma unroll
    for (int i = 0; i < GS_MAXPT; ++i)
        if (i < npt) {
            const int v = i * 256 + threadIdx.x;
            const float lp = x[i] - lse;
            float ea, eb;
            if (e1) { ea = e1[row * V + v]; eb = e2[row * V + v]; }
            else {
                const uint64_t idx = (uint64_t)row * V + v;
                const uint2 ba = rng_bits4(seed, SITE_GUMBEL_Z, idx);
                ea = -__logf(u01_24(ba.x));
                eb = -__logf(u01_24(ba.y));
            }
            const float g1 = -__logf(ea + TINYF), g2 = -__logf(eb + TINYF);
            u[i] = (lp + g1) * inv_tau;
            m2 = fmaxf(m2, u[i]);
            const float h = (lp + g2) * inv_tau;
            if (h > best) { best = h; besti = v; }
        }
    m2 = block_max(m2, red);
    float s2 = 0.f;
#pragma unroll
    for (int i = 0; i < GS_MAXPT; ++i)
        if (i < npt) { u[i] = __expf(u[i] - m2); s2 += u[i]; }
    s2 = block_sum(s2, red);
    const float inv = 1.0f / s2;
#pragma unroll
    for (int i = 0; i < GS_MAXPT; ++i)
        if (i < npt) z[row * V + i * 256 + threadIdx.x] = u[i] * inv;
    if (zst) {
        // hard=True (ocrs/common/utils.py:81-83): the dVAE decoder sees y_hard - y_soft.detach() + y_soft, evaluated in the
        // reference's order; index = first maximum of y_soft (exp is monotone, so of u)
        float b1 = -1.f;
        int b1i = 0;
#pragma unroll
        for (int i = 0; i < GS_MAXPT; ++i)
            if (i < npt && u[i] > b1) { b1 = u[i]; b1i = i * 256 + threadIdx.x; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const float ob = __shfl_xor(b1, o, 64);
            const int oi = __shfl_xor(b1i, o, 64);
            if (ob > b1 || (ob == b1 && oi < b1i)) { b1 = ob; b1i = oi; }
        }
        __shared__ float hv[4];
        __shared__ int hi[4];
        if ((threadIdx.x & 63) == 0) { hv[threadIdx.x >> 6] = b1; hi[threadIdx.x >> 6] = b1i; }
        __syncthreads();
        for (int k = 0; k < 4; ++k)
            if (hv[k] > b1 || (hv[k] == b1 && hi[k] < b1i)) { b1 = hv[k]; b1i = hi[k]; }
#pragma unroll
        for (int i = 0; i < GS_MAXPT; ++i)
            if (i < npt) {
                const int v = i * 256 + threadIdx.x;
                const float ys = u[i] * inv;
                zst[row * V + v] = ((v == b1i ? 1.f : 0.f) - ys) + ys;
            }
    }
    // argmax (first index wins ties, like torch.argmax on CPU)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(besti, o, 64);
        if (ob > best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { redv[w] = best; redi[w] = besti; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k)
            if (redv[k] > best || (redv[k] == best && redi[k] < besti)) { best = redv[k]; besti = redi[k]; }
        tokens[row] = besti;
    }
}

// in place: d[v] = z[v] * (d[v] - sum_v z*d) * scale     (softmax backward for one row of V)
template <int NPT>
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ z, float* __restrict__ d, int V, float scale) {
    __shared__ float red[4];
    const long long row = blockIdx.x;
    const int npt = NPT ? NPT : V / 256;                 // V % 256 == 0, V <= 256 * GS_MAXPT: one read of z and d, one write
    float zz[GS_MAXPT], dd[GS_MAXPT];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < GS_MAXPT; ++i)
        if (i < npt) {
            zz[i] = z[row * V + i * 256 + threadIdx.x];
            dd[i] = d[row * V + i * 256 + threadIdx.x];
            s += zz[i] * dd[i];
        }
    s = block_sum(s, red);
#pragma unroll
    for (int i = 0; i < GS_MAXPT; ++i)
        if (i < npt) d[row * V + i * 256 + threadIdx.x] = zz[i] * (dd[i] - s) * scale;
}

// cross entropy with hard targets, one workgroup per row: part[row] = lse - pred[tok];
// pred <- (softmax(pred) - onehot(tok)) * inv_b   (in place gradient)
template <int NPT>
__global__ __launch_bounds__(256) void ce_kernel(float* __restrict__ pred, const int* __restrict__ tokens, float* __restrict__ part,
                                                 int V, float inv_b, int write_grad) {
    __shared__ float red[4];
    const long long row = blockIdx.x;
    float* r = pred + row * V;
    const int tok = tokens[row];
    const int npt = NPT ? NPT : V / 256;                 // V % 256 == 0, V <= 256 * GS_MAXPT: the row lives in registers (one read, one exp, one write)
    float x[GS_MAXPT];
    float mx = -INFINITY, rt = 0.f;
#pragma unroll
    for (int i = 0; i < GS_MAXPT; ++i)
        if (i < npt) {
            x[i] = r[i * 256 + threadIdx.x];
            mx = fmaxf(mx, x[i]);
            if (i * 256 + (int)threadIdx.x == tok) rt = x[i];
        }
    mx = block_max(mx, red);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < GS_MAXPT; ++i)
        if (i < npt) { x[i] = __expf(x[i] - mx); s += x[i]; }
    s = block_sum(s, red);
    rt = block_sum(rt, red);                 // exactly one thread holds the target logit
    if (threadIdx.x == 0) part[row] = (mx + __logf(s)) - rt;
    if (write_grad) {
        const float inv = 1.0f / s;
#pragma unroll
        for (int i = 0; i < GS_MAXPT; ++i)
            if (i < npt) {
                const int v = i * 256 + threadIdx.x;
                r[v] = (x[i] * inv - (v == tok ? 1.f : 0.f)) * inv_b;
            }
    }
}

// ------------------------------------------------------------------ soft-max heads fused into the vocabulary GEMMs (gemm.hip, epi_mode 1-3)
// One wave per row, lane = 64-column segment: combines the per-segment (max, sum-exp) pairs the GEMM epilogue left in `stat` into
// lse[row]; optionally picks the hard Gumbel sample (first maximum over the segment maxima -> tokens[row]) and the cross-entropy term
// lse - pred[row, tok[row]], summed per block (16 rows, fixed order) into part[block].
// hard sample by inverse CDF (device-RNG mode of the Gumbel head, gemm_impl.h epi_mode 2): hstat / hidx hold the per-segment
// (max l, sum exp(l - max)) of the logits l; the token is a draw from Categorical(soft-max(l)) -- first the segment (cumulative segment
// masses against one uniform of the row), then the entry inside it (a second uniform against the cumulative exp(l - max) of its 64
// entries, l recovered from the stored scores: l = tau * y - g1 with g1 regenerated from the counter RNG exactly as the epilogue drew it)
struct CdfArgs { const float* scores; int V; float tau; unsigned long long seed; };
__device__ __forceinline__ float wave_incl_scan(float v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(v, o, 64);
        if (lane >= o) v += t;
    }
    return v;
}
__global__ __launch_bounds__(1024) void softmax_stat_combine_kernel(const float* __restrict__ stat, int nseg, long long R, float* __restrict__ lse,
                                                                    const float* __restrict__ hstat, const int* __restrict__ hidx,
                                                                    int* __restrict__ tokens, const float* __restrict__ pred, int ldp,
                                                                    const int* __restrict__ tok, float* __restrict__ part, CdfArgs cdf) {
    __shared__ float red[16];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 16 + wave;
    float loss = 0.f;
    if (row < R) {
        float m = -INFINITY, sv = 0.f;
        if (lane < nseg) { m = stat[((size_t)row * nseg + lane) * 2]; sv = stat[((size_t)row * nseg + lane) * 2 + 1]; }
        float mx = m;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float t = m > -INFINITY ? sv * __expf(m - mx) : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        const float l = mx + __logf(t);
        if (lane == 0) lse[row] = l;
        if (hstat && cdf.scores) {
            float hm = -INFINITY, hsum = 0.f;
            if (lane < nseg) { hm = hstat[(size_t)row * nseg + lane]; hsum = __builtin_bit_cast(float, hidx[(size_t)row * nseg + lane]); }
            float M = hm;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) M = fmaxf(M, __shfl_xor(M, o, 64));
            const float w = hm > -INFINITY ? hsum * __expf(hm - M) : 0.f;
            const float cw = wave_incl_scan(w, lane);
            const float tot = __shfl(cw, 63, 64);
            const uint2 ub = rng_bits4(cdf.seed, SITE_GUMBEL_ZH, (uint64_t)row);
            const float target = u01_24(ub.x) * tot;
            int js = __popcll(__ballot(cw < target));              // segments whose cumulative mass stays below the target
            if (js > nseg - 1) js = nseg - 1;
            const float hmj = __shfl(hm, js, 64);
            // inside segment js: entry v = lane
            const int col = js * 64 + lane;
            float q = 0.f;
            if (col < cdf.V) {
                const uint64_t idx = (uint64_t)row * (uint64_t)cdf.V + col;
                const uint32_t key = rng_key(cdf.seed, SITE_GUMBEL_Z, (uint32_t)(idx >> 32));
                const float la = __logf(1.17549435e-38f - __logf(u01_24(rng_bits1_keyed(key, (uint32_t)idx))));
                const float l = cdf.scores[idx] * cdf.tau + la;
                q = __expf(l - hmj);
            }
            const float cq = wave_incl_scan(q, lane);
            const float tq = __shfl(cq, 63, 64);
            const float t2 = u01_24(ub.y) * tq;
            int vs = __popcll(__ballot(cq < t2));
            const int vmax = (cdf.V - js * 64) < 64 ? (cdf.V - js * 64) - 1 : 63;
            if (vs > vmax) vs = vmax;
            if (lane == 0) tokens[row] = js * 64 + vs;
        } else if (hstat) {
            float b = -INFINITY;
            int bi = 0x7fffffff;
            if (lane < nseg) { b = hstat[(size_t)row * nseg + lane]; bi = hidx[(size_t)row * nseg + lane]; }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ob = __shfl_xor(b, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ob > b || (ob == b && oi < bi)) { b = ob; bi = oi; }
            }
            if (lane == 0) tokens[row] = bi;
        }
        if (part && lane == 0) loss = l - pred[(size_t)row * ldp + tok[row]];
    }
    if (part) {
        if (lane == 0) red[wave] = loss;
        __syncthreads();
        if (threadIdx.x == 0) {
            float a = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) a += red[k];
            part[blockIdx.x] = a;
        }
    }
}
// z[row, v] = exp(y[row, v] - lse[row]): the soft-max the fused heads never write, for callers that ask for it (with_rep, tests)
__global__ __launch_bounds__(256) void exp_rows_kernel(const float* __restrict__ y, const float* __restrict__ lse, float* __restrict__ z, long long n4, int V4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float l = lse[i / V4];
    const float4 v = *reinterpret_cast<const float4*>(y + i * 4);
    *reinterpret_cast<float4*>(z + i * 4) = make_float4(__expf(v.x - l), __expf(v.y - l), __expf(v.z - l), __expf(v.w - l));
}
// out[row] = sum_c g[row,c] * (act[row,c] - bias[c]),  64 channels, 16 lanes x float4 per row.  With act = relu(z W^T + bias) and g the
// gradient behind that ReLU this is sum_v z_v (dL/dz_v): the row term of the soft-max backward, without a pass over the vocabulary.
__global__ __launch_bounds__(256) void rowdot_bias64_kernel(const float* __restrict__ g, const float* __restrict__ act, const float* __restrict__ bias,
                                                            long long R, float* __restrict__ out) {
    const int c4 = threadIdx.x & 15;
    const long long row = (long long)blockIdx.x * 16 + (threadIdx.x >> 4);
    float a = 0.f;
    if (row < R) {
        const float4 gv = *reinterpret_cast<const float4*>(g + row * 64 + c4 * 4);
        const float4 av = *reinterpret_cast<const float4*>(act + row * 64 + c4 * 4);
        const float4 bv = *reinterpret_cast<const float4*>(bias + c4 * 4);
        a = (gv.x * (av.x - bv.x) + gv.y * (av.y - bv.y)) + (gv.z * (av.z - bv.z) + gv.w * (av.w - bv.w));
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) a += __shfl_xor(a, o, 64);
    if (row < R && c4 == 0) out[row] = a;
}

// ------------------------------------------------------------------ token embedding (+BOS, +pos, dropout)
// out[b,t,:] = drop( (t==0 ? bos : dict[tok[b,t-1]]) + pe[t] ),  dropout index over the reference's [B,T+1,d] tensor
// four rows of the same column group per thread: the token ids, then the four dictionary rows, are in flight together (the gather is two
// dependent round trips; with one element per thread the kernel ran 7x longer beside other streams' kernels than alone)
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int* __restrict__ tokens, const float* __restrict__ dict, const float* __restrict__ bos,
                                                        const float* __restrict__ pe, float* __restrict__ out, int B, int T, int d, float p,
                                                        unsigned long long seed) {
    const int d4 = d / 4;
    const long long n = (long long)B * T * d4;
    const long long i0 = (long long)blockIdx.x * 1024 + threadIdx.x;
    const uint32_t thr = drop_thresh(p);
    const float sc = 1.0f / (1.0f - p);
    int tok[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = i0 + k * 256, bt = i / d4;
        const int t = (int)(bt % T);
        tok[k] = (i < n && t > 0) ? tokens[bt - 1] : -1;          // row (b, t) embeds token (b, t - 1); bt - 1 = b * T + t - 1
    }
    float4 v[4], pv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = i0 + k * 256, bt = i / d4;
        const int c4 = (int)(i % d4), t = (int)(bt % T);
        v[k] = make_float4(0.f, 0.f, 0.f, 0.f); pv[k] = v[k];
        if (i < n) {
            v[k] = *reinterpret_cast<const float4*>((tok[k] < 0 ? bos : dict + (size_t)tok[k] * d) + c4 * 4);
            pv[k] = *reinterpret_cast<const float4*>(pe + (size_t)t * d + c4 * 4);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const long long i = i0 + k * 256;
        if (i >= n) continue;
        const long long bt = i / d4;
        const int c4 = (int)(i % d4), t = (int)(bt % T);
        const long long b = bt / T;
        float4 o = make_float4(v[k].x + pv[k].x, v[k].y + pv[k].y, v[k].z + pv[k].z, v[k].w + pv[k].w);
        if (p > 0.f) {
            const uint64_t idx4 = ((uint64_t)(b * (T + 1) + t) * d) / 4 + c4;
            const uint2 bits = rng_bits4(seed, SITE_ZPOS, idx4);
            o.x = rng_keep(bits, 0, thr) ? o.x * sc : 0.f;
            o.y = rng_keep(bits, 1, thr) ? o.y * sc : 0.f;
            o.z = rng_keep(bits, 2, thr) ? o.z * sc : 0.f;
            o.w = rng_keep(bits, 3, thr) ? o.w * sc : 0.f;
        }
        *reinterpret_cast<float4*>(out + i * 4) = o;
    }
}

// ---- gradient of the token dictionary: ddict[tok] = sum of the rows of g whose input token is tok (slate_module.py:141-146: row (b, t)
// of the decoder input is Embedding(token[b, t-1]) for t > 0, the BOS parameter for t = 0).  No atomics on the floats: the rows are
// sorted by token with a stable counting sort (integer histograms and prefix sums only), and every token's rows are then summed in
// ascending row order, so the result is bitwise reproducible.  A token may own any share of the rows (a background token of real scenes
// owns most of them), so the sorted list is cut into fixed units of EB_UNIT rows: a unit writes the sums of the segments that lie inside
// it and one partial for a segment that enters from / continues into a neighbour; a last pass adds the partials of the tokens that
// span units, again in a fixed order.  Round 2 used 64-bit fixed-point atomics on the [V,d] table (order-free and exact, but 25 M
// device-scope atomics took 0.59 ms at B = 128); this form reads g once (0.1 ms).
#define EB_CHUNK 512          // rows per histogram / placement workgroup
#define EB_UNIT 128           // sorted rows per summation workgroup
// token of row bt (V = "none": the BOS rows)
__device__ __forceinline__ int eb_row_token(const int* __restrict__ tokens, long long bt, int T, int V) {
    return (bt % T) ? tokens[bt - 1] : V;
}
__global__ __launch_bounds__(256) void eb_hist_kernel(const int* __restrict__ tokens, int* __restrict__ hist, long long BT, int T, int V) {
    extern __shared__ int eb_h[];           // [V + 1]
    for (int i = threadIdx.x; i <= V; i += 256) eb_h[i] = 0;
    __syncthreads();
    const long long r0 = (long long)blockIdx.x * EB_CHUNK;
    for (int i = threadIdx.x; i < EB_CHUNK; i += 256)
        if (r0 + i < BT) atomicAdd(&eb_h[eb_row_token(tokens, r0 + i, T, V)], 1);        // integer counts: exact in any order
    __syncthreads();
    for (int i = threadIdx.x; i <= V; i += 256) hist[(size_t)blockIdx.x * (V + 1) + i] = eb_h[i];
}
// per token: exclusive running count over the chunks (in place) and the total
__global__ void eb_scan_chunks_kernel(int* __restrict__ hist, int* __restrict__ total, int nchunk, int V) {
    const int tk = blockIdx.x * blockDim.x + threadIdx.x;
    if (tk > V) return;
    int run = 0;
    for (int c = 0; c < nchunk; ++c) {
        const int h = hist[(size_t)c * (V + 1) + tk];
        hist[(size_t)c * (V + 1) + tk] = run;
        run += h;
    }
    total[tk] = run;
}
// base[tk] = first sorted position of token tk (one workgroup, V + 1 <= 1024 * 16 entries)
__global__ __launch_bounds__(1024) void eb_base_kernel(const int* __restrict__ total, int* __restrict__ base, int V) {
    __shared__ int part[1024];
    const int per = (V + 1 + 1023) / 1024;
    const int i0 = threadIdx.x * per;
    int s = 0;
    for (int i = i0; i < i0 + per && i <= V; ++i) s += total[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = threadIdx.x >= o ? part[threadIdx.x - o] : 0;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    int run = part[threadIdx.x] - s;
    for (int i = i0; i < i0 + per && i <= V; ++i) { base[i] = run; run += total[i]; }
}
// stable placement: perm[base[tk] + (rows of tk in earlier chunks) + (earlier rows of tk in this chunk)] = row
__global__ __launch_bounds__(256) void eb_place_kernel(const int* __restrict__ tokens, const int* __restrict__ hist, const int* __restrict__ base,
                                                       int* __restrict__ perm, long long BT, int T, int V) {
    __shared__ int tk[EB_CHUNK];
    const long long r0 = (long long)blockIdx.x * EB_CHUNK;
    for (int i = threadIdx.x; i < EB_CHUNK; i += 256) tk[i] = r0 + i < BT ? eb_row_token(tokens, r0 + i, T, V) : -1;
    __syncthreads();
    for (int i = threadIdx.x; i < EB_CHUNK; i += 256) {
        const int t = tk[i];
        if (t < 0) continue;
        int rank = 0;
        for (int j = 0; j < i; ++j) rank += tk[j] == t;
        perm[base[t] + hist[(size_t)blockIdx.x * (V + 1) + t] + rank] = (int)(r0 + i);
    }
}
// one unit of EB_UNIT sorted rows; thread = column.  part: [units][2][d] (0: segment entering from the previous unit, 1: segment that
// starts here and continues into the next)
__global__ void eb_segsum_kernel(const float* __restrict__ g, const int* __restrict__ tokens, const int* __restrict__ perm, const int* __restrict__ base,
                                 const int* __restrict__ total, float* __restrict__ ddict, float* __restrict__ part, long long BT, int T, int V, int d) {
    __shared__ int rows[EB_UNIT], tks[EB_UNIT];
    const long long p0 = (long long)blockIdx.x * EB_UNIT;
    const int nrow = (int)((BT - p0) < EB_UNIT ? (BT - p0) : EB_UNIT);
    for (int i = threadIdx.x; i < EB_UNIT; i += blockDim.x) {
        const int r = i < nrow ? perm[p0 + i] : 0;
        rows[i] = r;
        tks[i] = i < nrow ? eb_row_token(tokens, r, T, V) : V;
    }
    __syncthreads();
    const int c = threadIdx.x;
    if (c >= d) return;
    float acc = 0.f;
    int cur = tks[0];
    auto flush = [&](int tk, float v) {
        if (tk >= V) return;                                              // BOS rows / padding
        const long long b0 = base[tk], b1 = b0 + total[tk];
        if (b0 >= p0 && b1 <= p0 + EB_UNIT) ddict[(size_t)tk * d + c] = v;                              // the whole segment lies in this unit
        else part[((size_t)blockIdx.x * 2 + (b0 < p0 ? 0 : 1)) * d + c] = v;
    };
    for (int i0 = 0; i0 < nrow; i0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = (i0 + u < nrow && tks[i0 + u] < V) ? g[(size_t)rows[i0 + u] * d + c] : 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (i0 + u >= nrow) break;
            const int tk = tks[i0 + u];
            if (tk != cur) { flush(cur, acc); acc = 0.f; cur = tk; }
            acc += v[u];
        }
    }
    flush(cur, acc);
}
// tokens whose rows span several units: tail partial of the first unit + head partials of the following ones, in order; absent tokens: 0
__global__ void eb_combine_kernel(const int* __restrict__ base, const int* __restrict__ total, const float* __restrict__ part, float* __restrict__ ddict, int V, int d) {
    const int tk = blockIdx.x, c = threadIdx.x;
    if (c >= d) return;
    const int n = total[tk];
    if (n == 0) { ddict[(size_t)tk * d + c] = 0.f; return; }
    const long long b0 = base[tk];
    const long long u0 = b0 / EB_UNIT, u1 = (b0 + n - 1) / EB_UNIT;
    if (u0 == u1) return;
    float acc = part[((size_t)u0 * 2 + 1) * d + c];
    for (long long u = u0 + 1; u <= u1; ++u) acc += part[((size_t)u * 2) * d + c];
    ddict[(size_t)tk * d + c] = acc;
}
// in-place dropout-backward of the decoder-input gradient (site SITE_ZPOS: the mask of embed_fwd_kernel, indexed over the [B, T+1, d] tensor)
__global__ void embed_drop_bwd_kernel(float* __restrict__ g, int B, int T, int d, float p, unsigned long long seed) {
    const int d4 = d / 4;
    const long long n = (long long)B * T * d4;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c4 = i % d4;
    const long long bt = i / d4;
    const int t = bt % T;
    const long long b = bt / T;
    float4 v = *reinterpret_cast<float4*>(g + i * 4);
    const uint64_t idx4 = ((uint64_t)(b * (T + 1) + t) * d) / 4 + c4;
    const uint2 bits = rng_bits4(seed, SITE_ZPOS, idx4);
    const uint32_t thr = drop_thresh(p);
    const float sc = 1.0f / (1.0f - p);
    v.x = rng_keep(bits, 0, thr) ? v.x * sc : 0.f;
    v.y = rng_keep(bits, 1, thr) ? v.y * sc : 0.f;
    v.z = rng_keep(bits, 2, thr) ? v.z * sc : 0.f;
    v.w = rng_keep(bits, 3, thr) ? v.w * sc : 0.f;
    *reinterpret_cast<float4*>(g + i * 4) = v;
}

// y = x * keep/(1-p) for the dropout site (n % 4 == 0); index = element offset
__global__ void dropout_apply_kernel(const float* __restrict__ x, float* __restrict__ y, long long n4, float p,
                                     unsigned long long seed, unsigned site) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 v = *reinterpret_cast<const float4*>(x + i * 4);
    const uint2 bits = rng_bits4(seed, site, (uint64_t)i);
    const uint32_t thr = drop_thresh(p);
    const float sc = 1.0f / (1.0f - p);
    v.x = rng_keep(bits, 0, thr) ? v.x * sc : 0.f;
    v.y = rng_keep(bits, 1, thr) ? v.y * sc : 0.f;
    v.z = rng_keep(bits, 2, thr) ? v.z * sc : 0.f;
    v.w = rng_keep(bits, 3, thr) ? v.w * sc : 0.f;
    *reinterpret_cast<float4*>(y + i * 4) = v;
}
// keep-mask dump (float 0/1) for the parity tests: exactly the decisions the kernels take
__global__ void dropout_mask_kernel(float* __restrict__ y, long long n, float p, unsigned long long seed, unsigned site) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint2 bits = rng_bits4(seed, site, (uint64_t)(i >> 2));
    y[i] = rng_keep(bits, (int)(i & 3), drop_thresh(p)) ? 1.f : 0.f;
}

// ------------------------------------------------------------------ cross attention to K (<= 16) slots
// Q [B,T,d] (projected, unscaled), Km/Vm [B,K,d], heads h, dh = d/h (<= 64).  One thread per (b,head,q).
// P [B,h,T,K] = softmax (pre-dropout) is saved for the backward.
// MAXK (8 or 16) is the compile-time slot capacity: per-slot values live in registers.
#define RC_(x) do { int rc__ = (x); if (rc__) return rc__; } while (0)
#define CA_MAXDH 64
template <int CA_MAXK>
__global__ __launch_bounds__(256) void cross_attn_fwd_kernel(const float* __restrict__ Q, const float* __restrict__ Km,
                                                             const float* __restrict__ Vm, float* __restrict__ O,
                                                             float* __restrict__ P, int T, int K, int d, int h, float p,
                                                             unsigned long long seed, unsigned site) {
    extern __shared__ float sm[];   // Ks [K][dh], Vs [K][dh]
    const int dh = d / h;
    const int nqb = (T + 255) / 256;
    int bid = blockIdx.x;
    const int qb = bid % nqb; bid /= nqb;
    const int hd = bid % h;
    const long long b = bid / h;
    float* Ks = sm;
    float* Vs = sm + K * dh;
    for (int i = threadIdx.x; i < K * dh; i += 256) {
        const int k = i / dh, c = i % dh;
        Ks[i] = Km[(b * K + k) * d + hd * dh + c];
        Vs[i] = Vm[(b * K + k) * d + hd * dh + c];
    }
    __syncthreads();
    const int q = qb * 256 + threadIdx.x;
    if (q >= T) return;
    const float scale = rsqrtf((float)dh);
    float qv[CA_MAXDH];
    const float* qp = Q + (b * T + q) * d + hd * dh;
#pragma unroll
    for (int c = 0; c < CA_MAXDH; c += 4)
        if (c < dh) {
            const float4 v = *reinterpret_cast<const float4*>(qp + c);
            qv[c] = v.x * scale; qv[c + 1] = v.y * scale; qv[c + 2] = v.z * scale; qv[c + 3] = v.w * scale;
        }
    float s[CA_MAXK];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < CA_MAXK; ++k)
        if (k < K) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < CA_MAXDH; c += 4)              // broadcast ds_read_b128: a quarter of the LDS instructions
                if (c < dh) {
                    const float4 k4 = *reinterpret_cast<const float4*>(Ks + k * dh + c);
                    a += (qv[c] * k4.x + qv[c + 1] * k4.y) + (qv[c + 2] * k4.z + qv[c + 3] * k4.w);
                }
            s[k] = a;
            mx = fmaxf(mx, a);
        }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < CA_MAXK; ++k)
        if (k < K) { s[k] = __expf(s[k] - mx); sum += s[k]; }
    const float inv = 1.0f / sum;
    const uint32_t thr = drop_thresh(p);
    const float sc = p > 0.f ? 1.0f / (1.0f - p) : 1.f;
    const long long prow = ((b * h + hd) * T + q) * K;
    float o[CA_MAXDH];
#pragma unroll
    for (int c = 0; c < CA_MAXDH; ++c) o[c] = 0.f;
#pragma unroll
    for (int k = 0; k < CA_MAXK; ++k)
        if (k < K) {
            const float pr = s[k] * inv;
            P[prow + k] = pr;
            float pd = pr;
            if (p > 0.f) {
                const uint64_t idx = (uint64_t)prow + k;
                pd = rng_keep(rng_bits4(seed, site, idx >> 2), (int)(idx & 3), thr) ? pr * sc : 0.f;
            }
#pragma unroll
            for (int c = 0; c < CA_MAXDH; c += 4)
                if (c < dh) {
                    const float4 v4 = *reinterpret_cast<const float4*>(Vs + k * dh + c);
                    o[c] += pd * v4.x; o[c + 1] += pd * v4.y; o[c + 2] += pd * v4.z; o[c + 3] += pd * v4.w;
                }
        }
    float* op = O + (b * T + q) * d + hd * dh;
#pragma unroll
    for (int c = 0; c < CA_MAXDH; c += 4)
        if (c < dh) *reinterpret_cast<float4*>(op + c) = make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]);
}

// backward: dQ [B,T,d] written; the slot-side gradients leave each workgroup as one partial [2][K*dh] (dV, dK) in `part`,
// summed over the query blocks in a fixed order by cross_attn_reduce_kernel (no float atomics: bitwise reproducible).  The
// per-query outer products are reduced over the 64 queries of a wave through a wave-private LDS staging tile (queries x (K + dh)),
// then over the four waves through LDS.
template <int CA_MAXK>
__global__ __launch_bounds__(256) void cross_attn_bwd_kernel(const float* __restrict__ dO, const float* __restrict__ Q,
                                                             const float* __restrict__ Km, const float* __restrict__ Vm,
                                                             const float* __restrict__ P, float* __restrict__ dQ,
                                                             float* __restrict__ part, int T, int K, int d,
                                                             int h, float p, unsigned long long seed, unsigned site) {
    extern __shared__ float sm[];   // Ks [K][dh], Vs [K][dh], stage [4][64][CA_SW]
    constexpr int CA_SW = CA_MAXK + CA_MAXDH + 1;
    float* mypart = part + (size_t)blockIdx.x * 2 * K * (d / h);
    const int dh = d / h;
    const int nqb = (T + 255) / 256;
    int bid = blockIdx.x;
    const int qb = bid % nqb; bid /= nqb;
    const int hd = bid % h;
    const long long b = bid / h;
    float* Ks = sm;
    float* Vs = sm + K * dh;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float* stg = sm + 2 * K * dh + wv * 64 * CA_SW;      // this wave's staging tile
    for (int i = threadIdx.x; i < K * dh; i += 256) {
        const int k = i / dh, c = i % dh;
        Ks[i] = Km[(b * K + k) * d + hd * dh + c];
        Vs[i] = Vm[(b * K + k) * d + hd * dh + c];
    }
    __syncthreads();
    const int q = qb * 256 + threadIdx.x;
    const bool act = q < T;
    const float scale = rsqrtf((float)dh);
    // The wave's 64 query rows of dO (and later of Q) are fetched as one coalesced tile (consecutive lanes = consecutive 16 bytes of a
    // row) straight into the staging tile the matrix products read; a thread then takes its own row from LDS.  Thread-per-row global
    // loads (64 different cache lines per instruction) were what bound this kernel.
    const int q0 = qb * 256 + wv * 64, dh4 = dh >> 2;
    // All dh/4 loads of a lane are issued before the first LDS store (a rolled load -> store loop paid one memory round trip per
    // 16 bytes: 2 x 12 round trips per workgroup, with two workgroups per CU to hide them).  (row, chunk) advance without divisions.
    const int r_step = 64 / dh4, c_step = 64 - r_step * dh4;
    auto load_tile = [&](const float* src, float mul) {
        float4 tmp[CA_MAXDH / 4];
        int r = lane / dh4, c4 = lane - r * dh4;
#pragma unroll
        for (int i = 0; i < CA_MAXDH / 4; ++i) {
            tmp[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < dh4 && q0 + r < T) tmp[i] = *reinterpret_cast<const float4*>(src + (b * T + q0 + r) * d + hd * dh + 4 * c4);
            r += r_step; c4 += c_step;
            if (c4 >= dh4) { c4 -= dh4; ++r; }
        }
        r = lane / dh4; c4 = lane - r * dh4;
#pragma unroll
        for (int i = 0; i < CA_MAXDH / 4; ++i) {
            if (i < dh4) {
                float* dst = stg + r * CA_SW + CA_MAXK + 4 * c4;
                dst[0] = tmp[i].x * mul; dst[1] = tmp[i].y * mul; dst[2] = tmp[i].z * mul; dst[3] = tmp[i].w * mul;
            }
            r += r_step; c4 += c_step;
            if (c4 >= dh4) { c4 -= dh4; ++r; }
        }
        __builtin_amdgcn_wave_barrier();
    };
    load_tile(dO, 1.f);
    float go[CA_MAXDH];
    float pr[CA_MAXK], dp[CA_MAXK], keepv[CA_MAXK];
#pragma unroll
    for (int c = 0; c < CA_MAXDH; ++c) go[c] = c < dh ? stg[lane * CA_SW + CA_MAXK + c] : 0.f;
#pragma unroll
    for (int k = 0; k < CA_MAXK; ++k) { pr[k] = 0.f; dp[k] = 0.f; keepv[k] = 0.f; }
    float delta = 0.f;
    if (act) {
        const uint32_t thr = drop_thresh(p);
        const float sc = p > 0.f ? 1.0f / (1.0f - p) : 1.f;
        const long long prow = ((b * h + hd) * T + q) * K;
#pragma unroll
        for (int k = 0; k < CA_MAXK; ++k)
            if (k < K) {
                pr[k] = P[prow + k];
                float keep = 1.f;
                if (p > 0.f) {
                    const uint64_t idx = (uint64_t)prow + k;
                    keep = rng_keep(rng_bits4(seed, site, idx >> 2), (int)(idx & 3), thr) ? sc : 0.f;
                }
                keepv[k] = keep;
                float a = 0.f;
#pragma unroll
                for (int c = 0; c < CA_MAXDH; c += 4)          // one broadcast ds_read_b128 per four channels (the per-channel form waited on 576 LDS reads per query)
                    if (c < dh) {
                        const float4 v4 = *reinterpret_cast<const float4*>(Vs + k * dh + c);
                        a += (go[c] * v4.x + go[c + 1] * v4.y) + (go[c + 2] * v4.z + go[c + 3] * v4.w);
                    }
                dp[k] = a * keep;
                delta += pr[k] * dp[k];
            }
    }
    // ---- dV[k][c] += sum_q pd[q][k] * go[q][c]      (go is already in the staging tile)
#pragma unroll
    for (int k = 0; k < CA_MAXK; ++k) stg[lane * CA_SW + k] = pr[k] * keepv[k];
    __syncthreads();
    // [slots x 64 queries] . [64 queries x dh] on the matrix cores: slots on the 16 MFMA rows, 16-channel tiles on the columns, four
    // queries per v_mfma_f32_16x16x4_f32 (lane (li, g): A = stage[query 4s+g][slot li], B = stage[query 4s+g][channel 16m+li])
    const int li = lane & 15, g4 = lane >> 4, nmt = (dh + 15) >> 4;
    f32x4 acc[CA_MAXDH / 16];
#pragma unroll
    for (int m = 0; m < CA_MAXDH / 16; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int s16 = 0; s16 < 16; ++s16) {
        const float* rowp = stg + (4 * s16 + g4) * CA_SW;
        const float a = li < CA_MAXK ? rowp[li] : 0.f;
#pragma unroll
        for (int m = 0; m < CA_MAXDH / 16; ++m)
            if (m < nmt) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, rowp[CA_MAXK + 16 * m + li], acc[m], 0, 0, 0);
    }
    __syncthreads();                                       // every wave is done reading its staging tile
#pragma unroll
    for (int m = 0; m < CA_MAXDH / 16; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 4 * g4 + r, c = 16 * m + li;
            if (m < nmt && k < K && c < dh) stg[k * dh + c] = acc[m][r];
        }
    __syncthreads();
    for (int o = threadIdx.x; o < K * dh; o += 256) {
        const float* r = sm + 2 * K * dh + o;
        mypart[o] = (r[0] + r[64 * CA_SW]) + (r[2 * 64 * CA_SW] + r[3 * 64 * CA_SW]);
    }
    __syncthreads();
    // ---- dS, dQ, and dK[k][c] += sum_q ds[q][k] * q'[q][c]
    float dq[CA_MAXDH];
#pragma unroll
    for (int c = 0; c < CA_MAXDH; ++c) dq[c] = 0.f;
#pragma unroll
    for (int k = 0; k < CA_MAXK; ++k) {
        const float ds = (k < K) ? pr[k] * (dp[k] - delta) : 0.f;
        stg[lane * CA_SW + k] = ds;
        if (k < K) {
#pragma unroll
            for (int c = 0; c < CA_MAXDH; c += 4)
                if (c < dh) {
                    const float4 k4 = *reinterpret_cast<const float4*>(Ks + k * dh + c);
                    dq[c] += ds * k4.x; dq[c + 1] += ds * k4.y; dq[c + 2] += ds * k4.z; dq[c + 3] += ds * k4.w;
                }
        }
    }
    load_tile(Q, scale);                                   // q' = scale * q of the wave's rows, straight into the staging tile
    __syncthreads();
#pragma unroll
    for (int m = 0; m < CA_MAXDH / 16; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int s16 = 0; s16 < 16; ++s16) {
        const float* rowp = stg + (4 * s16 + g4) * CA_SW;
        const float a = li < CA_MAXK ? rowp[li] : 0.f;
#pragma unroll
        for (int m = 0; m < CA_MAXDH / 16; ++m)
            if (m < nmt) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, rowp[CA_MAXK + 16 * m + li], acc[m], 0, 0, 0);
    }
    __syncthreads();
#pragma unroll
    for (int m = 0; m < CA_MAXDH / 16; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = 4 * g4 + r, c = 16 * m + li;
            if (m < nmt && k < K && c < dh) stg[k * dh + c] = acc[m][r];
        }
    __syncthreads();
    for (int o = threadIdx.x; o < K * dh; o += 256) {
        const float* r = sm + 2 * K * dh + o;
        mypart[K * dh + o] = (r[0] + r[64 * CA_SW]) + (r[2 * 64 * CA_SW] + r[3 * 64 * CA_SW]);
    }
    __syncthreads();                                       // the partials are out: the staging tile carries dQ to a coalesced store
#pragma unroll
    for (int c = 0; c < CA_MAXDH; ++c)
        if (c < dh) stg[lane * CA_SW + CA_MAXK + c] = dq[c] * scale;
    __builtin_amdgcn_wave_barrier();
    for (int idx = lane; idx < 64 * dh4; idx += 64) {
        const int r = idx / dh4, c4 = idx - r * dh4;
        if (q0 + r < T) {
            const float* src = stg + r * CA_SW + CA_MAXK + 4 * c4;
            *reinterpret_cast<float4*>(dQ + (b * T + q0 + r) * d + hd * dh + 4 * c4) = make_float4(src[0], src[1], src[2], src[3]);
        }
    }
}

// ------------------------------------------------------------------ first conv layer's weight gradient as a GEMM
// col[(b,y,x)][tap*C + c] = x8[b][y+ky-2][x+kx-2][c] (0 outside; columns >= 25*C are zero); x8 is NHWC with 8-float pixels
__global__ void im2col5_kernel(const float* __restrict__ x8, float* __restrict__ col, long long npix, int H, int W, int C, int ldc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix * ldc) return;
    const int j = i % ldc;
    const long long pix = i / ldc;
    float v = 0.f;
    if (j < 25 * C) {
        const int c = j % C, tap = j / C;
        const int x = pix % W, y = (pix / W) % H;
        const long long b = pix / ((long long)W * H);
        const int yy = y + tap / 5 - 2, xx = x + tap % 5 - 2;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) v = x8[((b * H + yy) * W + xx) * 8 + c];
    }
    col[i] = v;
}
// The common case (3 channels, 76-float rows) with compile-time divisors and a (row chunk, y, image) grid: four consecutive columns
// per thread, one float4 store; the general kernel above spends its time in 64-bit divisions (0.56 ms for 637 MB at B = 128).
__global__ __launch_bounds__(256) void im2col5_c3_kernel(const float* __restrict__ x8, float* __restrict__ col, int H, int W) {
    constexpr int C = 3, LDC4 = 19;                     // 76 floats per pixel
    const int e = blockIdx.x * 256 + threadIdx.x;       // float4 index inside the image row: pixel x, column group j4
    if (e >= W * LDC4) return;
    const int x = e / LDC4, j4 = e - x * LDC4;
    const int y = blockIdx.y;
    const long long b = blockIdx.z;
    float v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = j4 * 4 + k;
        const int tap = j / C, c = j - tap * C;
        const int yy = y + tap / 5 - 2, xx = x + tap % 5 - 2;
        v[k] = (j < 25 * C && yy >= 0 && yy < H && xx >= 0 && xx < W) ? x8[((b * H + yy) * W + xx) * 8 + c] : 0.f;
    }
    *reinterpret_cast<float4*>(col + ((b * H + y) * (long long)W) * (LDC4 * 4) + (long long)e * 4) = make_float4(v[0], v[1], v[2], v[3]);
}
// dW[co][c][tap] = dWp[co][tap*C + c]
__global__ void unpack5_kernel(const float* __restrict__ dWp, float* __restrict__ dW, int C, int ldc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * 25 * C) return;
    const int tap = i % 25, c = (i / 25) % C, co = i / (25 * C);
    dW[i] = dWp[co * ldc + tap * C + c];
}

// ------------------------------------------------------------------ misc
__global__ void fill_kernel(float* __restrict__ x, long long n, float v) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] = v;
}
__global__ void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, float a) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] += a * x[i];
}
// posmap[y,x,c] = sum_j grid[j,y,x] * Wpos[c,j] + bpos[c]   (grid = [north, south, west, east] ramps)
__global__ void posmap_kernel(const float* __restrict__ Wpos, const float* __restrict__ bpos, float* __restrict__ out, int S, int C) {
    const long long n = (long long)S * S * C;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = i % C;
    const int x = (i / C) % S, y = i / ((long long)C * S);
    const float east = S > 1 ? (float)x / (float)(S - 1) : 0.f, south = S > 1 ? (float)y / (float)(S - 1) : 0.f;
    const float west = 1.f - east, north = 1.f - south;
    out[i] = north * Wpos[c * 4 + 0] + south * Wpos[c * 4 + 1] + west * Wpos[c * 4 + 2] + east * Wpos[c * 4 + 3] + bpos[c];
}
// gridT[(y*S+x)*4 + j]: the position grid as an [S*S,4] matrix (for the pos-embedding weight gradient GEMM)
__global__ void posgrid_kernel(float* __restrict__ out, int S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * S) return;
    const int x = i % S, y = i / S;
    const float east = S > 1 ? (float)x / (float)(S - 1) : 0.f, south = S > 1 ? (float)y / (float)(S - 1) : 0.f;
    *reinterpret_cast<float4*>(out + (size_t)i * 4) = make_float4(1.f - south, south, 1.f - east, east);
}
// copy a [R, C] matrix into a [R, ldo] one (ldo >= C), zero padding; and back (pad==0)
__global__ void pad_cols_kernel(const float* __restrict__ in, int ldi, float* __restrict__ out, int ldo, long long R, int C, int Cout) {
    const long long n = R * Cout;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = i % Cout;
    const long long r = i / Cout;
    out[r * ldo + c] = c < C ? in[r * ldi + c] : 0.f;
}

// slots0[r][c] = mu[c] + exp(logsig[c]) * eps[r][c]; eps injected (noise) or N(0,1) drawn on device (Box-Muller)
__device__ inline float slot_eps(const float* noise, long long i, unsigned long long seed) {
    if (noise) return noise[i];
    const uint2 b = rng_bits4(seed, SITE_SLOT_NOISE, (uint64_t)i);
    return sqrtf(-2.0f * __logf(u01_24(b.x))) * __cosf(6.2831853f * u01_24(b.y));
}
__global__ void store_u64_kernel(unsigned long long* dst, unsigned long long v) { *dst = v; }
// seed_dev (optional): the seed is read from device memory, so that a captured launch (hipGraph) can be replayed with a new one
__global__ void slot_init_kernel(const float* __restrict__ mu, const float* __restrict__ logsig, const float* __restrict__ noise,
                                 float* __restrict__ slots0, long long n, int D, unsigned long long seed, const unsigned long long* __restrict__ seed_dev) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (seed_dev) seed = *seed_dev;
    const int c = i % D;
    slots0[i] = mu[c] + __expf(logsig[c]) * slot_eps(noise, i, seed);
}
// dmu[c] = sum_r d[r][c]; dlogsig[c] = sum_r d[r][c] * exp(logsig[c]) * eps[r][c].  One block per 64 columns: 16 row lanes per
// column run in parallel and are combined through LDS (deterministic order).
__global__ __launch_bounds__(1024) void slot_init_bwd_kernel(const float* __restrict__ d, const float* __restrict__ logsig, const float* __restrict__ noise,
                                                             float* __restrict__ dmu, float* __restrict__ dlogsig, int R, int D, unsigned long long seed) {
    __shared__ float ra[16][64], rb[16][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float a = 0.f, b = 0.f;
    if (c < D)
        for (int r = rl; r < R; r += 16) {
            const float g = d[(size_t)r * D + c];
            a += g;
            b += g * slot_eps(noise, (long long)r * D + c, seed);
        }
    ra[rl][cl] = a; rb[rl][cl] = b;
    __syncthreads();
    if (rl == 0 && c < D) {
        float sa = 0.f, sb = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { sa += ra[k][cl]; sb += rb[k][cl]; }
        dmu[c] = sa;
        dlogsig[c] = sb * __expf(logsig[c]);
    }
}
__global__ void copy_kernel(const float* __restrict__ src, float* __restrict__ dst, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

// tokens[b*T + t] = argmax_v of image b's logits at position t (first index wins ties); the logits are row b*T + t of a [B*T, V]
// matrix, or row b of a [B, V] matrix holding only position t (dense_rows: the KV-cached decode)
__global__ __launch_bounds__(256) void argmax_pos_kernel(const float* __restrict__ pred, int* __restrict__ tokens, int T, int V, int t, int dense_rows) {
    __shared__ float rv[4];
    __shared__ int ri[4];
    const long long row = (long long)blockIdx.x * T + t;
    const float* r = pred + (dense_rows ? (long long)blockIdx.x : row) * V;
    float best = -INFINITY;
    int bi = 0;
    for (int v = threadIdx.x; v < V; v += 256) { const float x = r[v]; if (x > best) { best = x; bi = v; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    if ((threadIdx.x & 63) == 0) { rv[threadIdx.x >> 6] = best; ri[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) if (rv[k] > best || (rv[k] == best && ri[k] < bi)) { best = rv[k]; bi = ri[k]; }
        tokens[row] = bi;
    }
}
// ---- KV-cached greedy decode (SLATE_Module._gen_imgs, ocrs/slate/slate_module.py:163-179): one new token per image and step
// x[b] = (t == 0 ? BOS : dictionary[tokens[b][t-1]]) + pe[t]          (slate_module.py:167-176, transformer.py:66; no dropout in eval)
__global__ void embed_step_kernel(const int* __restrict__ tokens, const float* __restrict__ dict, const float* __restrict__ bos,
                                  const float* __restrict__ pe, float* __restrict__ out, int B, int T, int d, int t) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * d) return;
    const int b = i / d, c = i - b * d;
    const float* src = t == 0 ? bos : dict + (size_t)tokens[b * T + t - 1] * d;
    out[i] = src[c] + pe[(size_t)t * d + c];
}
// Causal self-attention of the newest token against the cached keys / values (transformer.py:31-47 restricted to query row t):
// qkv is the fused projection buffer [B*T, ld] (q | k | v side by side), rows b*T + 0..t of image b are filled; out [B, d].
// One workgroup per (image, head): scores in LDS, softmax, then channels x key-slices accumulate P.V.
__global__ __launch_bounds__(256) void decode_attn_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T, int t, int d, int h, int ld) {
    extern __shared__ float sm[];
    const int dh = d / h, nk = t + 1, tid = threadIdx.x;
    float* sc = sm;                  // [nk] scores -> probabilities
    float* qs = sc + ((T + 3) & ~3); // [dh]
    float* red = qs + 64;            // [256]
    const int b = blockIdx.x / h, hd = blockIdx.x % h;
    const float* base = qkv + (size_t)b * T * ld + hd * dh;
    const float scale = rsqrtf((float)dh);
    if (tid < dh) qs[tid] = base[(size_t)t * ld + tid] * scale;
    __syncthreads();
    float mx = -INFINITY;
    for (int j = tid; j < nk; j += 256) {
        const float* kr = base + (size_t)j * ld + d;
        float a = 0.f;
        for (int c = 0; c < dh; c += 4) {
            const float4 kv = *reinterpret_cast<const float4*>(kr + c);
            a += qs[c] * kv.x + qs[c + 1] * kv.y + qs[c + 2] * kv.z + qs[c + 3] * kv.w;
        }
        sc[j] = a;
        mx = fmaxf(mx, a);
    }
    red[tid] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int j = tid; j < nk; j += 256) { const float e = __expf(sc[j] - mx); sc[j] = e; sum += e; }
    red[tid] = sum;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) red[tid] += red[tid + o]; __syncthreads(); }
    const float inv = 1.0f / red[0];
    __syncthreads();
    const int nsl = 256 / dh, c = tid % dh, sl = tid / dh;
    float acc = 0.f;
    if (sl < nsl)
        for (int j = sl; j < nk; j += nsl) acc += sc[j] * base[(size_t)j * ld + 2 * d + c];
    red[tid] = sl < nsl ? acc : 0.f;
    __syncthreads();
    if (tid < dh) {
        float a = 0.f;
        for (int k = 0; k < nsl; ++k) a += red[k * dh + tid];
        out[(size_t)b * d + hd * dh + tid] = a * inv;
    }
}

// z[row][v] = (v == tokens[row])
__global__ void onehot_kernel(const int* __restrict__ tokens, float* __restrict__ z, long long n, int V) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    z[i] = (int)(i % V) == tokens[i / V] ? 1.f : 0.f;
}

// ================================================================== launchers
#define GRID1D(n) dim3(cdiv((n), 256)), dim3(256)

int nchw_to_nhwc8_launch(const float* in, float* out, int B, int C, int H, int W, hipStream_t st) {
    OCRL_REQUIRE(C <= 8, "nchw_to_nhwc8: C must be <= 8");
    const long long n = (long long)B * H * W;
    hipLaunchKernelGGL(nchw_to_nhwc8_kernel, GRID1D(n), 0, st, in, out, B, C, H, W);
    OCRL_CHECK_LAUNCH("nchw_to_nhwc8");
    return 0;
}
int patchify4_launch(const float* in, float* out, int B, int C, int S, hipStream_t st) {
    OCRL_REQUIRE(S % 4 == 0, "patchify4: S %% 4 != 0");
    const long long n = (long long)B * (S / 4) * (S / 4) * C * 4;
    hipLaunchKernelGGL(patchify4_kernel, GRID1D(n), 0, st, in, out, B, C, S);
    OCRL_CHECK_LAUNCH("patchify4");
    return 0;
}
int pixel_shuffle_launch(const float* in, float* out, int B, int h, int w, int Cout, int forward, const float* mask, hipStream_t st) {
    const long long n = (long long)B * h * w * Cout;
    hipLaunchKernelGGL(pixel_shuffle_kernel, GRID1D(n), 0, st, in, out, B, h, w, Cout, forward, mask);
    OCRL_CHECK_LAUNCH("pixel_shuffle");
    return 0;
}
int layernorm_fwd_launch(const float* x, const float* g, const float* b, float* y, float* mean, float* rstd, long long R, int F, hipStream_t st) {
    OCRL_REQUIRE(F % 64 == 0 && F >= 64 && F <= 256, "layernorm: F must be 64..256, multiple of 64 (got %d)", F);
    dim3 grid(cdiv(R, 4)), blk(256);
    const bool al16 = (((uintptr_t)x | (uintptr_t)y | (uintptr_t)g | (uintptr_t)b) & 15) == 0;
    switch (F / 64) {
        case 1:
            if (al16) hipLaunchKernelGGL(layernorm_fwd64_kernel, dim3((unsigned)cdiv(R, 64)), blk, 0, st, x, g, b, y, mean, rstd, R);
            else hipLaunchKernelGGL(layernorm_fwd_kernel<1>, grid, blk, 0, st, x, g, b, y, mean, rstd, R);
            break;
        case 2: hipLaunchKernelGGL(layernorm_fwd_kernel<2>, grid, blk, 0, st, x, g, b, y, mean, rstd, R); break;
        case 3:
            if (al16) hipLaunchKernelGGL(layernorm_fwd192_kernel, dim3((unsigned)cdiv(R, 32)), blk, 0, st, x, g, b, y, mean, rstd, R);
            else hipLaunchKernelGGL(layernorm_fwd_kernel<3>, grid, blk, 0, st, x, g, b, y, mean, rstd, R);
            break;
        default: hipLaunchKernelGGL(layernorm_fwd_kernel<4>, grid, blk, 0, st, x, g, b, y, mean, rstd, R); break;
    }
    OCRL_CHECK_LAUNCH("layernorm_fwd");
    return 0;
}
int colsum_launch(const float* X, long long ld, float* out, long long R, int F, int accumulate, float scale, float* ws, size_t ws_floats, hipStream_t st) {
    if (F % 4 == 0 && ld % 4 == 0 && F <= 1024 && F >= 4 && (((uintptr_t)X) & 15) == 0 && R >= 1024) {
        long long nchunk = 2048;
        if (nchunk > R / 8) nchunk = R / 8;          // >= 8 rows per chunk: a 2048-row partial table (LayerNorm gamma / beta) spreads over 256 workgroups, not 32
        if ((size_t)nchunk * F > ws_floats) nchunk = (long long)(ws_floats / F);
        if (nchunk >= 16) {
            const long long rpc = (R + nchunk - 1) / nchunk;
            nchunk = (R + rpc - 1) / rpc;
            hipLaunchKernelGGL(colsum4_kernel, dim3((unsigned)nchunk), dim3(256), 0, st, X, ld, ws, R, F, rpc);
            OCRL_CHECK_LAUNCH("colsum4");
            hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(F, 16)), dim3(256), 0, st, ws, out, (int)nchunk, F, accumulate, scale);
            OCRL_CHECK_LAUNCH("colsum_final");
            return 0;
        }
    }
    const int cb = cdiv(F, 64);
    long long nchunk = 1024 / cb;
    if (nchunk < 1) nchunk = 1;
    if (nchunk > (R + 63) / 64) nchunk = (R + 63) / 64;
    if (nchunk < 1) nchunk = 1;
    OCRL_REQUIRE((size_t)nchunk * F <= ws_floats, "colsum: workspace too small (%lld x %d)", nchunk, F);
    const long long rpc = (R + nchunk - 1) / nchunk;
    hipLaunchKernelGGL(colsum_kernel, dim3(cb, (int)nchunk), dim3(256), 0, st, X, ld, ws, R, F, rpc);
    OCRL_CHECK_LAUNCH("colsum");
    hipLaunchKernelGGL(colsum_final_kernel, dim3(cdiv(F, 16)), dim3(256), 0, st, ws, out, (int)nchunk, F, accumulate, scale);
    OCRL_CHECK_LAUNCH("colsum_final");
    return 0;
}
int layernorm_bwd_launch(const float* dy, const float* x, const float* mean, const float* rstd, const float* g, float* dx,
                         float* dgb, long long R, int F, int accumulate_dx, int accumulate_dgb, float* ws, size_t ws_floats, hipStream_t st) {
    OCRL_REQUIRE(F % 64 == 0 && F >= 64 && F <= 256, "layernorm bwd: F must be 64..256, multiple of 64 (got %d)", F);
    int nblk = (int)((R + 3) / 4);
    if (nblk > 2048) nblk = 2048;         // 8 waves per SIMD (1024 / 4096 workgroups measured the same in the step)
    while (nblk > 64 && (size_t)nblk * 2 * F * 2 + (size_t)8 * 2 * F > ws_floats) nblk /= 2;
    const size_t need = (size_t)nblk * 2 * F;
    OCRL_REQUIRE(need + (size_t)8 * 2 * F <= ws_floats, "layernorm bwd: workspace too small");
    dim3 grid(nblk), blk(256);
    switch (F / 64) {
        case 1:
            if (((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)g) & 15) == 0) && (((uintptr_t)ws) & 15) == 0)
                hipLaunchKernelGGL(layernorm_bwd64_kernel, grid, blk, 0, st, dy, x, mean, rstd, g, dx, ws, R, accumulate_dx);
            else hipLaunchKernelGGL(layernorm_bwd_kernel<1>, grid, blk, 0, st, dy, x, mean, rstd, g, dx, ws, R, accumulate_dx);
            break;
        case 2: hipLaunchKernelGGL(layernorm_bwd_kernel<2>, grid, blk, 0, st, dy, x, mean, rstd, g, dx, ws, R, accumulate_dx); break;
        case 3:
            if (((((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dx | (uintptr_t)g) & 15) == 0) && (((uintptr_t)ws) & 15) == 0)
                hipLaunchKernelGGL(layernorm_bwd192_kernel, grid, blk, 0, st, dy, x, mean, rstd, g, dx, ws, R, accumulate_dx);
            else hipLaunchKernelGGL(layernorm_bwd_kernel<3>, grid, blk, 0, st, dy, x, mean, rstd, g, dx, ws, R, accumulate_dx);
            break;
        default: hipLaunchKernelGGL(layernorm_bwd_kernel<4>, grid, blk, 0, st, dy, x, mean, rstd, g, dx, ws, R, accumulate_dx); break;
    }
    OCRL_CHECK_LAUNCH("layernorm_bwd");
    return colsum_launch(ws, 2 * F, dgb, nblk, 2 * F, accumulate_dgb, 1.f, ws + need, ws_floats - need, st);
}
int reduce_partials_launch(const float* part, int n, float* out, float scale, int accumulate, hipStream_t st) {
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(256), 0, st, part, n, out, scale, accumulate);
    OCRL_CHECK_LAUNCH("reduce_partials");
    return 0;
}
int mse_launch(const float* obs, const float* recon, float* drecon, float* out, int B, int C, int H, int W, float* ws, size_t ws_floats, hipStream_t st) {
    OCRL_REQUIRE(C <= 4, "mse: C must be <= 4");
    const int nblk = 1024;
    OCRL_REQUIRE(ws_floats >= (size_t)nblk, "mse: workspace too small");
    hipLaunchKernelGGL(mse_kernel, dim3(nblk), dim3(256), 0, st, obs, recon, drecon, ws, B, C, H, W, 1.0f / B);
    OCRL_CHECK_LAUNCH("mse");
    return reduce_partials_launch(ws, nblk, out, 1.0f / B, 0, st);
}
int gumbel_softmax_launch(const float* raw, const float* e1, const float* e2, float* z, int* tokens, long long R, int V, float tau,
                          unsigned long long seed, hipStream_t st, float* zst) {
    OCRL_REQUIRE(V % 256 == 0 && V <= 256 * GS_MAXPT, "gumbel_softmax: V must be a multiple of 256, <= %d", 256 * GS_MAXPT);
    OCRL_REQUIRE((e1 == nullptr) == (e2 == nullptr), "gumbel_softmax: give both noise tensors or none");
    switch (V / 256) {
        case 16: hipLaunchKernelGGL(gumbel_softmax_kernel<16>, dim3((unsigned)R), dim3(256), 0, st, raw, e1, e2, z, tokens, V, 1.0f / tau, seed, zst); break;
        case 8: hipLaunchKernelGGL(gumbel_softmax_kernel<8>, dim3((unsigned)R), dim3(256), 0, st, raw, e1, e2, z, tokens, V, 1.0f / tau, seed, zst); break;
        default: hipLaunchKernelGGL(gumbel_softmax_kernel<0>, dim3((unsigned)R), dim3(256), 0, st, raw, e1, e2, z, tokens, V, 1.0f / tau, seed, zst); break;
    }
    OCRL_CHECK_LAUNCH("gumbel_softmax");
    return 0;
}
int softmax_bwd_rows_launch(const float* z, float* d, long long R, int V, float scale, hipStream_t st) {
    OCRL_REQUIRE(V % 256 == 0 && V <= 256 * GS_MAXPT, "softmax_bwd_rows: V must be a multiple of 256, <= %d", 256 * GS_MAXPT);
    if (V == 4096) hipLaunchKernelGGL(softmax_bwd_rows_kernel<16>, dim3((unsigned)R), dim3(256), 0, st, z, d, V, scale);
    else hipLaunchKernelGGL(softmax_bwd_rows_kernel<0>, dim3((unsigned)R), dim3(256), 0, st, z, d, V, scale);
    OCRL_CHECK_LAUNCH("softmax_bwd_rows");
    return 0;
}
int ce_launch(float* pred, const int* tokens, float* out, long long R, int V, int B, int write_grad, float* ws, size_t ws_floats, hipStream_t st) {
    OCRL_REQUIRE(ws_floats >= (size_t)R, "ce: workspace too small");
    OCRL_REQUIRE(V % 256 == 0 && V <= 256 * GS_MAXPT, "ce: V must be a multiple of 256, <= %d", 256 * GS_MAXPT);
    if (V == 4096) hipLaunchKernelGGL(ce_kernel<16>, dim3((unsigned)R), dim3(256), 0, st, pred, tokens, ws, V, 1.0f / B, write_grad);
    else hipLaunchKernelGGL(ce_kernel<0>, dim3((unsigned)R), dim3(256), 0, st, pred, tokens, ws, V, 1.0f / B, write_grad);
    OCRL_CHECK_LAUNCH("ce");
    return reduce_partials_launch(ws, (int)R, out, 1.0f / B, 0, st);
}
// lse [R]; tokens (with hstat / hidx) and the cross-entropy sum (with pred / tok; out[0] = scale * sum, ws >= ceil(R/16) floats) are optional
int softmax_stat_combine_launch(const float* stat, int nseg, long long R, float* lse, const float* hstat, const int* hidx, int* tokens,
                                const float* pred, int ldp, const int* tok, float* out, float scale, float* ws, size_t ws_floats, hipStream_t st,
                                const float* cdf_scores, int cdf_V, float cdf_tau, unsigned long long cdf_seed) {
    OCRL_REQUIRE(nseg >= 1 && nseg <= 64, "softmax_stat_combine: 1..64 segments (got %d)", nseg);
    const long long nblk = (R + 15) / 16;
    OCRL_REQUIRE(!pred || (tok && out && ws && ws_floats >= (size_t)nblk), "softmax_stat_combine: cross-entropy needs tokens, an output and %lld workspace floats", nblk);
    CdfArgs cdf;
    cdf.scores = cdf_scores; cdf.V = cdf_V; cdf.tau = cdf_tau; cdf.seed = cdf_seed;
    OCRL_REQUIRE(!cdf_scores || (hstat && hidx && tokens && cdf_V > 0 && cdf_V <= nseg * 64), "softmax_stat_combine: inverse-CDF sampling arguments");
    hipLaunchKernelGGL(softmax_stat_combine_kernel, dim3((unsigned)nblk), dim3(1024), 0, st, stat, nseg, R, lse, hstat, hidx, tokens, pred, ldp, tok,
                       pred ? ws : nullptr, cdf);
    OCRL_CHECK_LAUNCH("softmax_stat_combine");
    if (pred) return reduce_partials_launch(ws, (int)nblk, out, scale, 0, st);
    return 0;
}
int exp_rows_launch(const float* y, const float* lse, float* z, long long R, int V, hipStream_t st) {
    OCRL_REQUIRE(V % 4 == 0, "exp_rows: V %% 4 != 0");
    const long long n4 = R * (V / 4);
    hipLaunchKernelGGL(exp_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, st, y, lse, z, n4, V / 4);
    OCRL_CHECK_LAUNCH("exp_rows");
    return 0;
}
int rowdot_bias64_launch(const float* g, const float* act, const float* bias, long long R, float* out, hipStream_t st) {
    OCRL_REQUIRE(((((uintptr_t)g) | ((uintptr_t)act) | ((uintptr_t)bias)) & 15) == 0, "rowdot: operands must be 16-byte aligned");
    hipLaunchKernelGGL(rowdot_bias64_kernel, dim3((unsigned)((R + 15) / 16)), dim3(256), 0, st, g, act, bias, R, out);
    OCRL_CHECK_LAUNCH("rowdot_bias64");
    return 0;
}
int embed_fwd_launch(const int* tokens, const float* dict, const float* bos, const float* pe, float* out, int B, int T, int d, float p,
                     unsigned long long seed, hipStream_t st) {
    OCRL_REQUIRE(d % 4 == 0, "embed: d %% 4 != 0");
    const long long n = (long long)B * T * (d / 4);
    hipLaunchKernelGGL(embed_fwd_kernel, dim3((unsigned)cdiv(n, 1024)), dim3(256), 0, st, tokens, dict, bos, pe, out, B, T, d, p, seed);
    OCRL_CHECK_LAUNCH("embed_fwd");
    return 0;
}
// scratch of embed_bwd_launch, in floats: chunk histograms | totals | bases | sorted row list | unit partials
static size_t eb_align(size_t n) { return (n + 63) & ~(size_t)63; }
size_t embed_bwd_ws_floats(long long BT, int V, int d) {
    const size_t nchunk = (size_t)cdiv(BT, EB_CHUNK), nunit = (size_t)cdiv(BT, EB_UNIT);
    return eb_align(nchunk * (V + 1)) + 2 * eb_align((size_t)V + 2) + eb_align((size_t)BT) + eb_align(nunit * 2 * d);
}
// g <- dropout-backward(g) in place (p > 0); ddict [V,d] is written (no pre-zeroing needed)
int embed_bwd_launch(float* g, const int* tokens, float* ddict, int B, int T, int V, int d, float p, unsigned long long seed, float* ws,
                     size_t ws_floats, hipStream_t st) {
    const long long BT = (long long)B * T;
    OCRL_REQUIRE(d % 4 == 0 && d <= 1024 && V + 1 <= 16384 && BT < (1ll << 31), "embed_bwd: unsupported shape (d %d, vocabulary %d, %lld rows)", d, V, BT);
    OCRL_REQUIRE(ws && ws_floats >= embed_bwd_ws_floats(BT, V, d), "embed_bwd: scratch too small");
    const int nchunk = cdiv(BT, EB_CHUNK), nunit = cdiv(BT, EB_UNIT);
    int* hist = reinterpret_cast<int*>(ws);
    int* total = hist + eb_align((size_t)nchunk * (V + 1));
    int* base = total + eb_align((size_t)V + 2);
    int* perm = base + eb_align((size_t)V + 2);
    float* part = reinterpret_cast<float*>(perm + eb_align((size_t)BT));
    if (p > 0.f) {
        hipLaunchKernelGGL(embed_drop_bwd_kernel, GRID1D(BT * (d / 4)), 0, st, g, B, T, d, p, seed);
        OCRL_CHECK_LAUNCH("embed_drop_bwd");
    }
    hipLaunchKernelGGL(eb_hist_kernel, dim3(nchunk), dim3(256), (V + 1) * sizeof(int), st, tokens, hist, BT, T, V);
    hipLaunchKernelGGL(eb_scan_chunks_kernel, dim3(cdiv(V + 1, 256)), dim3(256), 0, st, hist, total, nchunk, V);
    hipLaunchKernelGGL(eb_base_kernel, dim3(1), dim3(1024), 0, st, total, base, V);
    hipLaunchKernelGGL(eb_place_kernel, dim3(nchunk), dim3(256), 0, st, tokens, hist, base, perm, BT, T, V);
    const int nthr = (d + 63) & ~63;
    hipLaunchKernelGGL(eb_segsum_kernel, dim3(nunit), dim3(nthr), 0, st, g, tokens, perm, base, total, ddict, part, BT, T, V, d);
    hipLaunchKernelGGL(eb_combine_kernel, dim3(V), dim3(nthr), 0, st, base, total, part, ddict, V, d);
    OCRL_CHECK_LAUNCH("embed_bwd");
    return 0;
}
int dropout_apply_launch(const float* x, float* y, long long n, float p, unsigned long long seed, unsigned site, hipStream_t st) {
    OCRL_REQUIRE(n % 4 == 0, "dropout_apply: n %% 4 != 0");
    hipLaunchKernelGGL(dropout_apply_kernel, GRID1D(n / 4), 0, st, x, y, n / 4, p, seed, site);
    OCRL_CHECK_LAUNCH("dropout_apply");
    return 0;
}
int dropout_mask_launch(float* y, long long n, float p, unsigned long long seed, unsigned site, hipStream_t st) {
    hipLaunchKernelGGL(dropout_mask_kernel, GRID1D(n), 0, st, y, n, p, seed, site);
    OCRL_CHECK_LAUNCH("dropout_mask");
    return 0;
}
int cross_attn_fwd_launch(const float* Q, const float* Km, const float* Vm, float* O, float* P, int B, int T, int K, int d, int h, float p,
                          unsigned long long seed, unsigned site, hipStream_t st) {
    OCRL_REQUIRE(K >= 1 && K <= 16 && d % h == 0 && (d / h) <= CA_MAXDH && (d / h) % 4 == 0, "cross_attn: unsupported K=%d d=%d h=%d", K, d, h);
    const int grid = B * h * cdiv(T, 256);
    if (K <= 8) hipLaunchKernelGGL(cross_attn_fwd_kernel<8>, dim3(grid), dim3(256), 2 * K * (d / h) * 4, st, Q, Km, Vm, O, P, T, K, d, h, p, seed, site);
    else hipLaunchKernelGGL(cross_attn_fwd_kernel<16>, dim3(grid), dim3(256), 2 * K * (d / h) * 4, st, Q, Km, Vm, O, P, T, K, d, h, p, seed, site);
    OCRL_CHECK_LAUNCH("cross_attn_fwd");
    return 0;
}
// dKm / dVm [B,K,d] = sum over the query blocks of the per-workgroup partials, in block order
__global__ void cross_attn_reduce_kernel(const float* __restrict__ part, float* __restrict__ dKm, float* __restrict__ dVm, long long n, int nqb,
                                         int K, int d, int h) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over [B][K][d]
    if (i >= n) return;
    const int dh = d / h;
    const int col = i % d, k = (i / d) % K;
    const long long b = i / ((long long)d * K);
    const int hd = col / dh, c = col - hd * dh;
    const float* src = part + ((b * h + hd) * nqb) * 2 * K * dh + k * dh + c;
    float av = 0.f, ak = 0.f;
    for (int q = 0; q < nqb; ++q) { av += src[(size_t)q * 2 * K * dh]; ak += src[(size_t)q * 2 * K * dh + K * dh]; }
    dVm[i] = av;
    dKm[i] = ak;
}

size_t cross_attn_bwd_ws_floats(int B, int T, int K, int d, int h) { return (size_t)B * h * cdiv(T, 256) * 2 * K * (d / h); }

template <int MAXK>
static int cross_attn_bwd_launch_k(const float* dO, const float* Q, const float* Km, const float* Vm, const float* P, float* dQ, float* dKm, float* dVm,
                                   int B, int T, int K, int d, int h, float p, unsigned long long seed, unsigned site, float* part, hipStream_t st) {
    constexpr int SW = MAXK + CA_MAXDH + 1;
    const int grid = B * h * cdiv(T, 256);
    static bool attr_set = false;
    if (!attr_set) {
        OCRL_HIP(hipFuncSetAttribute((const void*)cross_attn_bwd_kernel<MAXK>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (2 * MAXK * CA_MAXDH + 4 * 64 * SW) * 4));
        attr_set = true;
    }
    hipLaunchKernelGGL(cross_attn_bwd_kernel<MAXK>, dim3(grid), dim3(256), (2 * K * (d / h) + 4 * 64 * SW) * 4, st, dO, Q, Km, Vm, P, dQ, part, T, K, d, h, p, seed, site);
    OCRL_CHECK_LAUNCH("cross_attn_bwd");
    const long long n = (long long)B * K * d;
    hipLaunchKernelGGL(cross_attn_reduce_kernel, GRID1D(n), 0, st, part, dKm, dVm, n, cdiv(T, 256), K, d, h);
    OCRL_CHECK_LAUNCH("cross_attn_reduce");
    return 0;
}
// dKm / dVm are written (not accumulated); part: cross_attn_bwd_ws_floats() floats of scratch
int cross_attn_bwd_launch(const float* dO, const float* Q, const float* Km, const float* Vm, const float* P, float* dQ, float* dKm, float* dVm,
                          int B, int T, int K, int d, int h, float p, unsigned long long seed, unsigned site, float* part, size_t part_floats,
                          hipStream_t st) {
    OCRL_REQUIRE(K >= 1 && K <= 16 && d % h == 0 && (d / h) <= CA_MAXDH && (d / h) % 4 == 0, "cross_attn: unsupported K=%d d=%d h=%d", K, d, h);
    OCRL_REQUIRE(part && part_floats >= cross_attn_bwd_ws_floats(B, T, K, d, h), "cross_attn_bwd: scratch too small");
    if (K <= 8) return cross_attn_bwd_launch_k<8>(dO, Q, Km, Vm, P, dQ, dKm, dVm, B, T, K, d, h, p, seed, site, part, st);
    return cross_attn_bwd_launch_k<16>(dO, Q, Km, Vm, P, dQ, dKm, dVm, B, T, K, d, h, p, seed, site, part, st);
}
// Input pipeline on the device (reference: utils/datasets.py:13-24, `torch.Tensor(obss[i]).permute(2, 0, 1) / 255.0`):
// uint8 [B,H,W,C] -> fp32 [B,C,H,W], a correctly rounded fp32 division by 255 (bit-identical to the reference's).  One thread per
// pixel reads its C bytes and writes C planes; consecutive threads = consecutive pixels, so the plane writes are coalesced.
__global__ void obs_u8_to_f32_kernel(const unsigned char* __restrict__ in, float* __restrict__ out, long long npix_total, long long HW, int C) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix_total) return;
    const long long b = i / HW, pix = i - b * HW;
    const unsigned char* src = in + i * C;
    float* dst = out + b * C * HW + pix;
    for (int c = 0; c < C; ++c) dst[c * HW] = (float)src[c] / 255.0f;
}
int obs_u8_to_f32_launch(const unsigned char* in, float* out, int B, int H, int W, int C, hipStream_t st) {
    const long long n = (long long)B * H * W;
    hipLaunchKernelGGL(obs_u8_to_f32_kernel, GRID1D(n), 0, st, in, out, n, (long long)H * W, C);
    OCRL_CHECK_LAUNCH("obs_u8_to_f32");
    return 0;
}
int fill_launch(float* x, long long n, float v, hipStream_t st) {
    hipLaunchKernelGGL(fill_kernel, GRID1D(n), 0, st, x, n, v);
    OCRL_CHECK_LAUNCH("fill");
    return 0;
}
int axpy_launch(const float* x, float* y, long long n, float a, hipStream_t st) {
    hipLaunchKernelGGL(axpy_kernel, GRID1D(n), 0, st, x, y, n, a);
    OCRL_CHECK_LAUNCH("axpy");
    return 0;
}
int posmap_launch(const float* Wpos, const float* bpos, float* out, int S, int C, hipStream_t st) {
    hipLaunchKernelGGL(posmap_kernel, GRID1D((long long)S * S * C), 0, st, Wpos, bpos, out, S, C);
    OCRL_CHECK_LAUNCH("posmap");
    return 0;
}
int posgrid_launch(float* out, int S, hipStream_t st) {
    hipLaunchKernelGGL(posgrid_kernel, GRID1D(S * S), 0, st, out, S);
    OCRL_CHECK_LAUNCH("posgrid");
    return 0;
}
int pad_cols_launch(const float* in, int ldi, float* out, int ldo, long long R, int C, int Cout, hipStream_t st) {
    hipLaunchKernelGGL(pad_cols_kernel, GRID1D(R * Cout), 0, st, in, ldi, out, ldo, R, C, Cout);
    OCRL_CHECK_LAUNCH("pad_cols");
    return 0;
}
int store_u64_launch(unsigned long long* dst, unsigned long long v, hipStream_t st) {
    hipLaunchKernelGGL(store_u64_kernel, dim3(1), dim3(1), 0, st, dst, v);
    OCRL_CHECK_LAUNCH("store_u64");
    return 0;
}
int slot_init_launch(const float* mu, const float* logsig, const float* noise, float* slots0, int BK, int D, unsigned long long seed, hipStream_t st,
                     const unsigned long long* seed_dev) {
    const long long n = (long long)BK * D;
    hipLaunchKernelGGL(slot_init_kernel, GRID1D(n), 0, st, mu, logsig, noise, slots0, n, D, seed, seed_dev);
    OCRL_CHECK_LAUNCH("slot_init");
    return 0;
}
int slot_init_bwd_launch(const float* dslots0, const float* logsig, const float* noise, float* dmu, float* dlogsig, int BK, int D, unsigned long long seed, hipStream_t st) {
    hipLaunchKernelGGL(slot_init_bwd_kernel, dim3(cdiv(D, 64)), dim3(1024), 0, st, dslots0, logsig, noise, dmu, dlogsig, BK, D, seed);
    OCRL_CHECK_LAUNCH("slot_init_bwd");
    return 0;
}
int copy_launch(const float* src, float* dst, long long n, hipStream_t st) {
    hipLaunchKernelGGL(copy_kernel, GRID1D(n), 0, st, src, dst, n);
    OCRL_CHECK_LAUNCH("copy");
    return 0;
}
int argmax_pos_launch(const float* pred, int* tokens, int B, int T, int V, int t, hipStream_t st, int dense_rows) {
    hipLaunchKernelGGL(argmax_pos_kernel, dim3(B), dim3(256), 0, st, pred, tokens, T, V, t, dense_rows);
    OCRL_CHECK_LAUNCH("argmax_pos");
    return 0;
}
int embed_step_launch(const int* tokens, const float* dict, const float* bos, const float* pe, float* out, int B, int T, int d, int t, hipStream_t st) {
    hipLaunchKernelGGL(embed_step_kernel, GRID1D((long long)B * d), 0, st, tokens, dict, bos, pe, out, B, T, d, t);
    OCRL_CHECK_LAUNCH("embed_step");
    return 0;
}
int decode_attn_launch(const float* qkv, float* out, int B, int T, int t, int d, int h, int ld, hipStream_t st) {
    OCRL_REQUIRE(d % h == 0 && (d / h) % 4 == 0 && (d / h) <= 64 && t >= 0 && t < T && ld % 4 == 0, "decode_attn: unsupported shape");
    const size_t smem = (size_t)(((T + 3) & ~3) + 64 + 256) * 4;
    OCRL_REQUIRE(smem <= 64 * 1024, "decode_attn: sequence too long for the LDS score buffer (T = %d)", T);
    hipLaunchKernelGGL(decode_attn_kernel, dim3(B * h), dim3(256), smem, st, qkv, out, T, t, d, h, ld);
    OCRL_CHECK_LAUNCH("decode_attn");
    return 0;
}
int onehot_launch(const int* tokens, float* z, long long rows, int V, hipStream_t st) {
    hipLaunchKernelGGL(onehot_kernel, GRID1D(rows * V), 0, st, tokens, z, rows * V, V);
    OCRL_CHECK_LAUNCH("onehot");
    return 0;
}

int im2col5_launch(const float* x8, float* col, long long npix, int H, int W, int C, int ldc, hipStream_t st) {
    if (C == 3 && ldc == 76 && npix % ((long long)H * W) == 0 && (((uintptr_t)col) & 15) == 0 && H <= 65535 && npix / ((long long)H * W) <= 65535)
        hipLaunchKernelGGL(im2col5_c3_kernel, dim3((unsigned)((W * 19 + 255) / 256), (unsigned)H, (unsigned)(npix / ((long long)H * W))), dim3(256), 0, st,
                           x8, col, H, W);
    else
        hipLaunchKernelGGL(im2col5_kernel, GRID1D(npix * ldc), 0, st, x8, col, npix, H, W, C, ldc);
    OCRL_CHECK_LAUNCH("im2col5");
    return 0;
}
int unpack5_launch(const float* dWp, float* dW, int C, int ldc, hipStream_t st) {
    hipLaunchKernelGGL(unpack5_kernel, GRID1D(64 * 25 * C), 0, st, dWp, dW, C, ldc);
    OCRL_CHECK_LAUNCH("unpack5");
    return 0;
}
