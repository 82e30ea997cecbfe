// IODINE-specific kernels (reference: ocrs/iodine/iodine_module.py).  The 3x3 / 64-channel decoder convolutions and every
// dense contraction (linears, LSTM, stride-2 refinement convolutions as im2col GEMMs) run on conv.hip / gemm.hip; this
// file holds what is particular to the model:
//   * the decoder's first layer on the spatial broadcast (iodine_module.py:438-470): a convolution of a per-slot
//     constant plus two coordinate channels is  P1[y,x,:] + T[bk, class(y,x), :]  — one [BK,64]x[64,576] GEMM and 9 border
//     classes instead of a 66-channel convolution over every pixel;
//   * the mixture likelihood / ELBO, its in-forward gradients and the 17-channel refinement encoding (:92-229) with the
//     reference's non-affine layer norms (:307-330), forward and backward;
//   * im2col / col2im for the stride-2 refinement convolutions, pooling, the double ELU, the LSTM cell, the latent vector.
// Layouts: activations NHWC / [rows, features]; obs stays in the caller's NCHW.
#include "common.h"
#include "kernels.h"

#define IO_ENC 17

__device__ inline float elu_(float v) { return v > 0.f ? v : expf(v) - 1.f; }
__device__ inline float io_lin(int i, int S) { return S > 1 ? -1.f + 2.f * (float)i / (float)(S - 1) : -1.f; }
// border class of a coordinate for a 3x3 "same" convolution: 0 first row/column, 2 last, 1 interior
__device__ inline int io_cls(int i, int S) { return i == 0 ? 0 : (i == S - 1 ? 2 : 1); }
// tap offset k (0..2) stays inside the image for class c
__device__ inline bool io_valid(int c, int k) { return !(c == 0 && k == 0) && !(c == 2 && k == 2); }

// ------------------------------------------------------------------------------------------- sampling + KL
// slots = mu + exp(ls) * eps (eps injected or drawn: Box-Muller on the counter RNG, stored for the backward);
// kl_out += sum(-ls + (exp(2 ls) + mu^2)/2 - 1/2)
__global__ __launch_bounds__(256) void io_sample_kernel(const float* __restrict__ mu, const float* __restrict__ ls, const float* __restrict__ noise,
                                                        float* __restrict__ eps_out, float* __restrict__ slots, float* __restrict__ kl_elem, long long n,
                                                        unsigned long long seed, unsigned site) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    float kl = 0.f;
    if (i < n) {
        float e;
        if (noise) e = noise[i];
        else {
            const uint2 b = rng_bits4(seed, site, (uint64_t)i);
            e = sqrtf(-2.0f * logf(u01_24(b.x))) * __cosf(6.2831853f * u01_24(b.y));
        }
        const float m = mu[i], l = ls[i], s = expf(l);
        eps_out[i] = e;
        slots[i] = m + s * e;
        kl = -l + 0.5f * (s * s + m * m) - 0.5f;
    }
    if (i < n) kl_elem[i] = kl;
}
// out[0] = sum of x[0..n) in a fixed order (one workgroup: strided partial sums, then a tree): bitwise reproducible
__global__ __launch_bounds__(1024) void io_ordered_sum_kernel(const float* __restrict__ x, long long n, long long stride, float* __restrict__ out) {
    __shared__ float red[1024];
    float a = 0.f;
    for (long long i = threadIdx.x; i < n; i += 1024) a += x[i * stride];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 512; o > 0; o >>= 1) { if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) out[0] = red[0];
}

// ------------------------------------------------------------------------------------------- decoder layer 1
// W1 [64][L+2][9] -> W1r[tap][co][ci < L],  Wxy[tap][co][2]
__global__ void io_w1_pack_kernel(const float* __restrict__ W1, float* __restrict__ W1r, float* __restrict__ Wxy, int L) {
    const int n = 9 * 64 * (L + 2);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int ci = i % (L + 2), co = (i / (L + 2)) % 64, tap = i / ((L + 2) * 64);
    const float v = W1[((size_t)co * (L + 2) + ci) * 9 + tap];
    if (ci < L) W1r[((size_t)tap * 64 + co) * L + ci] = v;
    else Wxy[(tap * 64 + co) * 2 + (ci - L)] = v;
}
// P1[y][x][co] = b1[co] + sum over taps inside the image of Wxy[tap][co][0] * xx(x+dx) + Wxy[tap][co][1] * yy(y+dy)
__global__ void io_p1_kernel(const float* __restrict__ Wxy, const float* __restrict__ b1, float* __restrict__ P1, int S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= S * S * 64) return;
    const int co = i & 63, x = (i >> 6) % S, y = (i >> 6) / S;
    float a = b1[co];
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
            const int yy = y + ky - 1, xx = x + kx - 1;
            if (yy < 0 || yy >= S || xx < 0 || xx >= S) continue;
            const float* w = Wxy + ((ky * 3 + kx) * 64 + co) * 2;
            a += w[0] * io_lin(xx, S) + w[1] * io_lin(yy, S);
        }
    P1[i] = a;
}
// forward: T[bk][cls][co] = sum_{taps valid in cls} M[bk][tap][co];  backward: dM[bk][tap][co] = sum_{cls valid for tap} dT[bk][cls][co]
__global__ void io_class_sum_kernel(const float* __restrict__ in, float* __restrict__ out, long long BK, int forward) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BK * 9 * 64) return;
    const int co = i & 63, j = (i >> 6) % 9;
    const long long bk = i / (9 * 64);
    const int j0 = j / 3, j1 = j % 3;
    float a = 0.f;
    for (int q = 0; q < 9; ++q) {
        const int q0 = q / 3, q1 = q % 3;
        // forward: j = class, q = tap; backward: j = tap, q = class
        const bool ok = forward ? (io_valid(j0, q0) && io_valid(j1, q1)) : (io_valid(q0, j0) && io_valid(q1, j1));
        if (ok) a += in[(bk * 9 + q) * 64 + co];
    }
    out[i] = a;
}
// c1[bk][y][x][co] = elu(P1[y][x][co] + T[bk][cls(y,x)][co])
__global__ void io_layer1_kernel(const float* __restrict__ P1, const float* __restrict__ T, float* __restrict__ c1, long long BK, int S) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // float4 index
    const long long n4 = BK * S * S * 16;
    if (i >= n4) return;
    const int c4 = i & 15;
    const long long pix = (i >> 4) % ((long long)S * S), bk = (i >> 4) / ((long long)S * S);
    const int x = pix % S, y = pix / S;
    const float4 p = *reinterpret_cast<const float4*>(P1 + pix * 64 + c4 * 4);
    const float4 t = *reinterpret_cast<const float4*>(T + (bk * 9 + io_cls(y, S) * 3 + io_cls(x, S)) * 64 + c4 * 4);
    *reinterpret_cast<float4*>(c1 + i * 4) = make_float4(elu_(p.x + t.x), elu_(p.y + t.y), elu_(p.z + t.z), elu_(p.w + t.w));
}
// rowpart[bk][y][c][co] = sum over the pixels of row y in column class c of g[bk][y][x][co]   (one block per (bk, y));
// io_layer1_reduce_kernel sums the rows of each row class in row order (no float atomics: bitwise reproducible)
__global__ __launch_bounds__(256) void io_layer1_bwd_kernel(const float* __restrict__ g, float* __restrict__ rowpart, int S) {
    __shared__ float red[4][3][64];
    const long long bk = blockIdx.x / S;
    const int y = blockIdx.x % S;
    const int co = threadIdx.x & 63, part = threadIdx.x >> 6;
    float a[3] = {0.f, 0.f, 0.f};
    const float* row = g + ((bk * S + y) * S) * 64 + co;
    for (int x = part; x < S; x += 4) a[io_cls(x, S)] += row[(size_t)x * 64];
    for (int c = 0; c < 3; ++c) red[part][c][co] = a[c];
    __syncthreads();
    if (part == 0) {
        for (int c = 0; c < 3; ++c) rowpart[((bk * S + y) * 3 + c) * 64 + co] = (red[0][c][co] + red[1][c][co]) + (red[2][c][co] + red[3][c][co]);
    }
}
// dT[bk][rc*3+c][co] = sum over the rows y of row class rc of rowpart[bk][y][c][co]
__global__ void io_layer1_reduce_kernel(const float* __restrict__ rowpart, float* __restrict__ dT, long long n, int S) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over [BK][9][64]
    if (i >= n) return;
    const int co = i & 63, cls = (i >> 6) % 9;
    const long long bk = i / (9 * 64);
    const int rc = cls / 3, c = cls - rc * 3;
    const int y0 = rc == 0 ? 0 : (rc == 1 ? 1 : S - 1), y1 = rc == 1 ? S - 1 : y0 + 1;
    float a = 0.f;
    for (int y = y0; y < y1; ++y) a += rowpart[((bk * S + y) * 3 + c) * 64 + co];
    dT[i] = a;
}
// Weight gradient of layer 1 from the accumulated pieces: dW1[co][ci<L][tap] = dW1r[tap][co][ci];
// dW1[co][L+j][tap] = sum_{pixels where tap is inside} G[y][x][co] * coord_j;  db1[co] = sum G   (one block per tap)
__global__ __launch_bounds__(256) void io_w1_grad_kernel(const float* __restrict__ dW1r, const float* __restrict__ G, float* __restrict__ dW1,
                                                         float* __restrict__ db1, int S, int L) {
    __shared__ float red[4][3][64];
    const int tap = blockIdx.x, ky = tap / 3, kx = tap % 3;
    const int co = threadIdx.x & 63, part = threadIdx.x >> 6;
    float ax = 0.f, ay = 0.f, ab = 0.f;
    for (int pix = part; pix < S * S; pix += 4) {
        const int x = pix % S, y = pix / S;
        const int yy = y + ky - 1, xx = x + kx - 1;
        if (yy < 0 || yy >= S || xx < 0 || xx >= S) continue;
        const float v = G[(size_t)pix * 64 + co];
        ax += v * io_lin(xx, S);
        ay += v * io_lin(yy, S);
        ab += v;
    }
    red[part][0][co] = ax; red[part][1][co] = ay; red[part][2][co] = ab;
    __syncthreads();
    if (part == 0) {
        float s[3];
        for (int c = 0; c < 3; ++c) s[c] = red[0][c][co] + red[1][c][co] + red[2][c][co] + red[3][c][co];
        dW1[((size_t)co * (L + 2) + L) * 9 + tap] = s[0];
        dW1[((size_t)co * (L + 2) + L + 1) * 9 + tap] = s[1];
        if (tap == 4) db1[co] = s[2];
    }
    for (int i = threadIdx.x; i < 64 * L; i += 256) {
        const int ci = i % L, c2 = i / L;
        dW1[((size_t)c2 * (L + 2) + ci) * 9 + tap] = dW1r[((size_t)tap * 64 + c2) * L + ci];
    }
}

// ------------------------------------------------------------------------------------------- mixture likelihood / encoding
// One thread per (image, pixel).  out4 [B*K][N][4] = (rgb means, mask logit); obs NCHW.
//   part[block][0] += sum_c pixel log-likelihood, part[block][1] += sum_c (obs - recon)^2
//   enc  [B*K][N][17] raw refinement encoding (channels 9..14 are normalised by io_enc_norm afterwards)
//   st1  [B*K][4]    += per-(image,slot) sums of the four normalised groups (grad_means, grad_mask, likelihood, leave-one-out)
//   dout4 [B*K][N][4] = d(B*elbo)/d(out4) (the in-forward gradient that is pushed through the decoder)
template <int MAXK>
__global__ __launch_bounds__(256) void io_elbo_kernel(const float* __restrict__ out4, const float* __restrict__ obs, int B, int K, int S, float sigma,
                                                      float* __restrict__ enc, float* __restrict__ blkpart, float* __restrict__ dout4,
                                                      float* __restrict__ masks_out, float* __restrict__ recon_out,
                                                      float* __restrict__ rmasked_out) {
    __shared__ float red[4][4 * MAXK + 2];
    for (int i = threadIdx.x; i < 4 * (4 * MAXK + 2); i += 256) (&red[0][0])[i] = 0.f;
    __syncthreads();
    const int N = S * S;
    const long long gi = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = gi / N, pix = gi % N;          // N % 256 == 0: a block never straddles two images
    float x[3], r[MAXK][3], m[MAXK], a[MAXK];
#pragma unroll
    for (int c = 0; c < 3; ++c) x[c] = obs[((size_t)b * 3 + c) * N + pix];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) {
            const float4 o = *reinterpret_cast<const float4*>(out4 + (((size_t)b * K + k) * N + pix) * 4);
            r[k][0] = o.x; r[k][1] = o.y; r[k][2] = o.z; a[k] = o.w;
            mx = fmaxf(mx, o.w);
        }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) { m[k] = expf(a[k] - mx); se += m[k]; }
    const float inv = 1.f / se;
    const float i2s2 = 0.5f / (sigma * sigma), is2 = 1.f / (sigma * sigma), cst = -logf(sigma) - 0.9189385332f;
    float post[MAXK][3], A[MAXK], ll = 0.f, mse = 0.f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) { m[k] = k < K ? m[k] * inv : 0.f; A[k] = 0.f; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float t[MAXK], tm = -INFINITY, rc = 0.f;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) {
                const float d = x[c] - r[k][c];
                const float clp = -d * d * i2s2 + cst;
                A[k] += clp;
                t[k] = logf(m[k] + 1e-12f) + clp;
                tm = fmaxf(tm, t[k]);
                rc += m[k] * r[k][c];
            }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) { post[k][c] = expf(t[k] - tm); s += post[k][c]; }
        const float is = 1.f / s;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) post[k][c] *= is;
        ll += tm + logf(s);
        mse += (x[c] - rc) * (x[c] - rc);
        if (recon_out) recon_out[((size_t)b * 3 + c) * N + pix] = fminf(fmaxf(rc, 0.f), 1.f);
    }
    if (masks_out) {
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) masks_out[((size_t)b * K + k) * N + pix] = m[k];
    }
    if (rmasked_out) {
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K)
                for (int c = 0; c < 3; ++c) rmasked_out[(((size_t)b * K + k) * 3 + c) * N + pix] = fminf(fmaxf(m[k] * r[k][c], 0.f), 1.f);
    }
    // ---- in-forward gradients of B*elbo:  d/dr = post (x - r) / sigma^2 ;  d/dm = sum_c post / (m + 1e-12)
    float mg[MAXK], mgm = 0.f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k) {
        mg[k] = 0.f;
        if (k < K) { mg[k] = (post[k][0] + post[k][1] + post[k][2]) / (m[k] + 1e-12f); mgm += m[k] * mg[k]; }
    }
    if (dout4) {
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K)
                *reinterpret_cast<float4*>(dout4 + (((size_t)b * K + k) * N + pix) * 4) =
                    make_float4(post[k][0] * (x[0] - r[k][0]) * is2, post[k][1] * (x[1] - r[k][1]) * is2, post[k][2] * (x[2] - r[k][2]) * is2,
                                m[k] * (mg[k] - mgm));
    }
    if (enc) {
        float am = -INFINITY;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) am = fmaxf(am, A[k]);
        float as = 0.f, smap = 0.f;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) { as += expf(A[k] - am); smap += m[k] * expf(A[k]); }
        const float lsA = am + logf(as), like = expf(ll);
        const float cx = io_lin(pix % S, S), cy = io_lin(pix / S, S);
        const int lane = threadIdx.x & 63;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) {
                float* e = enc + (((size_t)b * K + k) * N + pix) * IO_ENC;
                const float g0 = post[k][0] * (x[0] - r[k][0]) * is2, g1 = post[k][1] * (x[1] - r[k][1]) * is2, g2 = post[k][2] * (x[2] - r[k][2]) * is2;
                const float loo = (smap - m[k] * expf(A[k])) / (1.f - m[k] + 1e-5f);
                e[0] = x[0]; e[1] = x[1]; e[2] = x[2]; e[3] = r[k][0]; e[4] = r[k][1]; e[5] = r[k][2]; e[6] = m[k]; e[7] = a[k]; e[8] = A[k] - lsA;
                e[9] = g0; e[10] = g1; e[11] = g2; e[12] = mg[k]; e[13] = like; e[14] = loo; e[15] = cx; e[16] = cy;
                const float s0 = wave_sum(g0 + g1 + g2), s1 = wave_sum(mg[k]), s2 = wave_sum(like), s3 = wave_sum(loo);
                if (lane == 0) { float* r4 = &red[threadIdx.x >> 6][4 * k]; r4[0] = s0; r4[1] = s1; r4[2] = s2; r4[3] = s3; }
            }
    }
    ll = wave_sum(ll);
    mse = wave_sum(mse);
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6][4 * MAXK] = ll; red[threadIdx.x >> 6][4 * MAXK + 1] = mse; }
    __syncthreads();
    // this block's partial [4 MAXK + 2]: the four waves in order (io_elbo_reduce_kernel sums the blocks in order)
    if ((int)threadIdx.x < 4 * MAXK + 2) blkpart[(size_t)blockIdx.x * (4 * MAXK + 2) + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// st1[b*K + k][g] = sum over the N/256 blocks of image b (in order) of blkpart[blk][4k + g];  part[0..1] += sum over all blocks of (ll, mse)
template <int MAXK>
__global__ __launch_bounds__(256) void io_elbo_reduce_kernel(const float* __restrict__ blkpart, float* __restrict__ st1, float* __restrict__ part, int B, int K,
                                                            int nb_img, int with_st1) {
    constexpr int W = 4 * MAXK + 2;
    __shared__ float red[2][256];
    const int tid = threadIdx.x;
    if (with_st1)
        for (int i = tid; i < B * K * 4; i += 256) {
            const int g = i & 3, k = (i >> 2) % K, b = i / (4 * K);
            float a = 0.f;
            for (int j = 0; j < nb_img; ++j) a += blkpart[((size_t)b * nb_img + j) * W + 4 * k + g];
            st1[i] = a;
        }
    float l = 0.f, m = 0.f;
    for (int j = tid; j < B * nb_img; j += 256) { l += blkpart[(size_t)j * W + 4 * MAXK]; m += blkpart[(size_t)j * W + 4 * MAXK + 1]; }
    red[0][tid] = l; red[1][tid] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (tid < o) { red[0][tid] += red[0][tid + o]; red[1][tid] += red[1][tid + o]; } __syncthreads(); }
    if (tid == 0) { part[0] += red[0][0]; part[1] += red[1][0]; }
}
// second pass of the 5-D layer norm: blkpart[block][g] = sum over the block's pixels of (v - mean)^2, mean = st1 / count   (one thread
// per (bk, pixel); io_enc_var_reduce_kernel sums the N/256 blocks of a (bk) in order)
__global__ __launch_bounds__(256) void io_enc_var_kernel(const float* __restrict__ enc, const float* __restrict__ st1, float* __restrict__ blkpart, int N) {
    __shared__ float red[4][4];
    const long long gi = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long bk = gi / N;
    const float* e = enc + gi * IO_ENC;
    const float* s = st1 + bk * 4;
    const float m0 = s[0] / (3.f * N), m1 = s[1] / N, m2 = s[2] / N, m3 = s[3] / N;
    const float d0 = e[9] - m0, d1 = e[10] - m0, d2 = e[11] - m0, d3 = e[12] - m1, d4 = e[13] - m2, d5 = e[14] - m3;
    const float v0 = wave_sum(d0 * d0 + d1 * d1 + d2 * d2), v1 = wave_sum(d3 * d3), v2 = wave_sum(d4 * d4), v3 = wave_sum(d5 * d5);
    if ((threadIdx.x & 63) == 0) { float* t = red[threadIdx.x >> 6]; t[0] = v0; t[1] = v1; t[2] = v2; t[3] = v3; }
    __syncthreads();
    if (threadIdx.x < 4) blkpart[(size_t)blockIdx.x * 4 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void io_enc_var_reduce_kernel(const float* __restrict__ blkpart, float* __restrict__ st2, long long n, int nb) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // over [BK][4]
    if (i >= n) return;
    const long long bk = i >> 2;
    const int g = i & 3;
    float a = 0.f;
    for (int j = 0; j < nb; ++j) a += blkpart[((size_t)bk * nb + j) * 4 + g];
    st2[i] = a;
}
// (v - mean) / (sqrt(var) + 1e-5), population variance over (C,H,W)  (iodine_module.py:316-329); layer_norm == 0 leaves the raw values
__global__ __launch_bounds__(256) void io_enc_norm_kernel(float* __restrict__ enc, const float* __restrict__ st1, const float* __restrict__ st2, int N) {
    const long long gi = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long bk = gi / N;
    float* e = enc + gi * IO_ENC;
    const float* s = st1 + bk * 4;
    const float* t = st2 + bk * 4;
    const float m0 = s[0] / (3.f * N), m1 = s[1] / N, m2 = s[2] / N, m3 = s[3] / N;
    const float r0 = 1.f / (sqrtf(t[0] / (3.f * N)) + 1e-5f), r1 = 1.f / (sqrtf(t[1] / N) + 1e-5f), r2 = 1.f / (sqrtf(t[2] / N) + 1e-5f),
                r3 = 1.f / (sqrtf(t[3] / N) + 1e-5f);
    e[9] = (e[9] - m0) * r0; e[10] = (e[10] - m0) * r0; e[11] = (e[11] - m0) * r0;
    e[12] = (e[12] - m1) * r1; e[13] = (e[13] - m2) * r2; e[14] = (e[14] - m3) * r3;
}

// Backward of one iteration's ELBO and encoding wrt out4:  cw = -(i+1)/(I*B) weights the log-likelihood; denc (nullable)
// is the gradient of the refinement encoding: channels 3-5 (means), 6 (mask), 7 (mask logits), 8 (mask posterior) carry gradient.
template <int MAXK>
__global__ __launch_bounds__(256) void io_elbo_bwd_kernel(const float* __restrict__ out4, const float* __restrict__ obs, const float* __restrict__ denc,
                                                          int B, int K, int S, float sigma, float cw, float* __restrict__ dout4) {
    const int N = S * S;
    const long long gi = (long long)blockIdx.x * 256 + threadIdx.x;
    const int b = gi / N, pix = gi % N;
    float x[3], r[MAXK][3], m[MAXK], a[MAXK];
#pragma unroll
    for (int c = 0; c < 3; ++c) x[c] = obs[((size_t)b * 3 + c) * N + pix];
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) {
            const float4 o = *reinterpret_cast<const float4*>(out4 + (((size_t)b * K + k) * N + pix) * 4);
            r[k][0] = o.x; r[k][1] = o.y; r[k][2] = o.z; a[k] = o.w;
            mx = fmaxf(mx, o.w);
        }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) { m[k] = expf(a[k] - mx); se += m[k]; }
    const float inv = 1.f / se;
    const float i2s2 = 0.5f / (sigma * sigma), is2 = 1.f / (sigma * sigma), cst = -logf(sigma) - 0.9189385332f;
    float dr[MAXK][3], dm[MAXK], A[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) { m[k] = k < K ? m[k] * inv : 0.f; A[k] = 0.f; dm[k] = 0.f; }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float t[MAXK], tm = -INFINITY;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) {
                const float d = x[c] - r[k][c];
                const float clp = -d * d * i2s2 + cst;
                A[k] += clp;
                t[k] = logf(m[k] + 1e-12f) + clp;
                tm = fmaxf(tm, t[k]);
            }
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) { t[k] = expf(t[k] - tm); s += t[k]; }
        const float is = 1.f / s;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) {
                const float post = t[k] * is;
                dr[k][c] = cw * post * (x[c] - r[k][c]) * is2;
                dm[k] += cw * post / (m[k] + 1e-12f);
            }
    }
    float da[MAXK];
#pragma unroll
    for (int k = 0; k < MAXK; ++k) da[k] = 0.f;
    if (denc) {
        float am = -INFINITY;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) am = fmaxf(am, A[k]);
        float as = 0.f, dls = 0.f, dlp[MAXK];
#pragma unroll
        for (int k = 0; k < MAXK; ++k) {
            dlp[k] = 0.f;
            if (k < K) {
                const float* e = denc + (((size_t)b * K + k) * N + pix) * IO_ENC;
                dr[k][0] += e[3]; dr[k][1] += e[4]; dr[k][2] += e[5];
                dm[k] += e[6];
                da[k] = e[7];
                dlp[k] = e[8];
                dls += e[8];
                A[k] = expf(A[k] - am);
                as += A[k];
            }
        }
        const float ias = 1.f / as;
#pragma unroll
        for (int k = 0; k < MAXK; ++k)
            if (k < K) {
                const float dA = dlp[k] - A[k] * ias * dls;          // through log_softmax over the slots
#pragma unroll
                for (int c = 0; c < 3; ++c) dr[k][c] += dA * (x[c] - r[k][c]) * is2;
            }
    }
    float dmm = 0.f;
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K) dmm += m[k] * dm[k];
#pragma unroll
    for (int k = 0; k < MAXK; ++k)
        if (k < K)
            *reinterpret_cast<float4*>(dout4 + (((size_t)b * K + k) * N + pix) * 4) = make_float4(dr[k][0], dr[k][1], dr[k][2], da[k] + m[k] * (dm[k] - dmm));
}

// ------------------------------------------------------------------------------------------- latent vector
// One wave per (image, slot) row: latent = [mu, ls, LN(gmu), LN(gls)], gmu = ds - beta mu, gls = ds sigma eps - beta (sigma^2 - 1)
// with the reference's 3-D layer norm: unbiased std, eps added to the std (iodine_module.py:313-315,329).
__global__ __launch_bounds__(256) void io_latent_kernel(const float* __restrict__ mu, const float* __restrict__ ls, const float* __restrict__ eps,
                                                        const float* __restrict__ ds, float* __restrict__ latent, long long BK, int L, float beta,
                                                        int layer_norm, int ld) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= BK) return;
    float s1 = 0.f, s2 = 0.f;
    for (int c = lane; c < L; c += 64) {
        const float m = mu[row * L + c], l = ls[row * L + c], sg = expf(l), d = ds[row * L + c];
        s1 += d - beta * m;
        s2 += d * sg * eps[row * L + c] - beta * (sg * sg - 1.f);
    }
    const float m1 = wave_sum(s1) / L, m2 = wave_sum(s2) / L;
    float v1 = 0.f, v2 = 0.f;
    for (int c = lane; c < L; c += 64) {
        const float m = mu[row * L + c], l = ls[row * L + c], sg = expf(l), d = ds[row * L + c];
        const float a = d - beta * m - m1, b = d * sg * eps[row * L + c] - beta * (sg * sg - 1.f) - m2;
        v1 += a * a; v2 += b * b;
    }
    const float r1 = 1.f / (sqrtf(wave_sum(v1) / (L - 1)) + 1e-5f), r2 = 1.f / (sqrtf(wave_sum(v2) / (L - 1)) + 1e-5f);
    for (int c = lane; c < L; c += 64) {
        const float m = mu[row * L + c], l = ls[row * L + c], sg = expf(l), d = ds[row * L + c];
        const float a = d - beta * m, b = d * sg * eps[row * L + c] - beta * (sg * sg - 1.f);
        float* o = latent + row * ld;
        o[c] = m; o[L + c] = l;
        o[2 * L + c] = layer_norm ? (a - m1) * r1 : a;
        o[3 * L + c] = layer_norm ? (b - m2) * r2 : b;
    }
}

// ------------------------------------------------------------------------------------------- stride-2 3x3 convolutions as GEMMs
// col[(b,oy,ox)][tap*C + c] = x[b][2oy-1+ky][2ox-1+kx][c] (0 outside; columns >= 9C are zero padding)
__global__ void io_im2col_kernel(const float* __restrict__ x, float* __restrict__ col, long long rows, int C, int Hi, int Wi, int Ho, int Wo, int ldc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * ldc) return;
    const int j = i % ldc;
    const long long row = i / ldc;
    float v = 0.f;
    if (j < 9 * C) {
        const int c = j % C, tap = j / C;
        const int ox = row % Wo, oy = (row / Wo) % Ho;
        const long long b = row / ((long long)Wo * Ho);
        const int y = 2 * oy - 1 + tap / 3, xx = 2 * ox - 1 + tap % 3;
        if (y >= 0 && y < Hi && xx >= 0 && xx < Wi) v = x[((b * Hi + y) * Wi + xx) * C + c];
    }
    col[i] = v;
}
// same, on a (row chunk, oy, image) grid with 32-bit index arithmetic and one float4 store per thread (ldc % 4 == 0); the kernel above
// spends its time in five 64-bit divisions per element
__global__ __launch_bounds__(256) void io_im2col4_kernel(const float* __restrict__ x, float* __restrict__ col, int C, int Hi, int Wi, int Ho, int Wo, int ldc) {
    const int L4 = ldc >> 2;
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Wo * L4) return;
    const int ox = e / L4, j4 = e - ox * L4;
    const int oy = blockIdx.y;
    const long long b = blockIdx.z;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    const float* xb = x + b * Hi * Wi * C;
    if ((C & 3) == 0) {                                   // four consecutive columns = four channels of one tap
        const int j = j4 * 4, tap = j / C, c = j - tap * C;
        const int y = 2 * oy - 1 + tap / 3, xx = 2 * ox - 1 + tap % 3;
        if (tap < 9 && y >= 0 && y < Hi && xx >= 0 && xx < Wi) {
            const float4 t = *reinterpret_cast<const float4*>(xb + ((size_t)y * Wi + xx) * C + c);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int j = j4 * 4 + k, tap = j / C, c = j - tap * C;
            const int y = 2 * oy - 1 + tap / 3, xx = 2 * ox - 1 + tap % 3;
            if (tap < 9 && y >= 0 && y < Hi && xx >= 0 && xx < Wi) v[k] = xb[((size_t)y * Wi + xx) * C + c];
        }
    }
    *reinterpret_cast<float4*>(col + ((b * Ho + oy) * (long long)Wo) * ldc + (long long)e * 4) = make_float4(v[0], v[1], v[2], v[3]);
}
// dx[b][y][x][c] = (sum over the windows that contain the pixel of dcol) * elu'(act)   (act = this tensor's ELU output, nullable)
__global__ void io_col2im_kernel(const float* __restrict__ dcol, const float* __restrict__ act, float* __restrict__ dx, long long n, int C, int Hi,
                                 int Wi, int Ho, int Wo, int ldc) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int c = i % C;
    const int x = (i / C) % Wi, y = (i / ((long long)C * Wi)) % Hi;
    const long long b = i / ((long long)C * Wi * Hi);
    float a = 0.f;
    for (int ky = 0; ky < 3; ++ky) {
        const int ty = y + 1 - ky;
        if (ty < 0 || (ty & 1) || (ty >> 1) >= Ho) continue;
        for (int kx = 0; kx < 3; ++kx) {
            const int tx = x + 1 - kx;
            if (tx < 0 || (tx & 1) || (tx >> 1) >= Wo) continue;
            a += dcol[((b * Ho + (ty >> 1)) * Wo + (tx >> 1)) * ldc + (ky * 3 + kx) * C + c];
        }
    }
    if (act) { const float m = act[i]; a = m > 0.f ? a : a * (m + 1.f); }
    dx[i] = a;
}
// same on a (row chunk, y, image) grid with 32-bit index arithmetic
__global__ __launch_bounds__(256) void io_col2im3_kernel(const float* __restrict__ dcol, const float* __restrict__ act, float* __restrict__ dx, int C,
                                                         int Hi, int Wi, int Ho, int Wo, int ldc) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= Wi * C) return;
    const int x = e / C, c = e - x * C;
    const int y = blockIdx.y;
    const long long b = blockIdx.z;
    float a = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int ty = y + 1 - ky;
        if (ty < 0 || (ty & 1) || (ty >> 1) >= Ho) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int tx = x + 1 - kx;
            if (tx < 0 || (tx & 1) || (tx >> 1) >= Wo) continue;
            a += dcol[((b * Ho + (ty >> 1)) * Wo + (tx >> 1)) * ldc + (ky * 3 + kx) * C + c];
        }
    }
    const long long i = ((b * Hi + y) * (long long)Wi) * C + e;
    if (act) { const float m = act[i]; a = m > 0.f ? a : a * (m + 1.f); }
    dx[i] = a;
}
// W[co][C][3][3] -> Wp[co][tap*C + c] (row stride ldc, zero padded);  backward: dW[co][c][tap] (+)= dWp[co][tap*C + c]
__global__ void io_refw_pack_kernel(const float* __restrict__ W, float* __restrict__ Wp, int C, int ldc, int backward, float* __restrict__ dW,
                                    const float* __restrict__ dWp) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 64 * ldc) return;
    const int j = i % ldc, co = i / ldc;
    if (!backward) {
        float v = 0.f;
        if (j < 9 * C) v = W[((size_t)co * C + j % C) * 9 + j / C];
        Wp[i] = v;
    } else if (j < 9 * C) {
        dW[((size_t)co * C + j % C) * 9 + j / C] = dWp[i];
    }
}
// pool[bk][c] = mean over the n pixels of r[bk][n][64]
__global__ __launch_bounds__(256) void io_pool_kernel(const float* __restrict__ r, float* __restrict__ pool, int n) {
    __shared__ float red[4][64];
    const long long bk = blockIdx.x;
    const int c = threadIdx.x & 63, part = threadIdx.x >> 6;
    float a = 0.f;
    for (int p = part; p < n; p += 4) a += r[(bk * n + p) * 64 + c];
    red[part][c] = a;
    __syncthreads();
    if (part == 0) pool[bk * 64 + c] = (red[0][c] + red[1][c] + red[2][c] + red[3][c]) / n;
}
// d(pre-activation of the last refinement conv) = dpool / n * elu'(r)
__global__ void io_pool_bwd_kernel(const float* __restrict__ dpool, const float* __restrict__ r, float* __restrict__ dpre, long long total, int n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const long long bk = i / ((long long)n * 64);
    const float m = r[i], g = dpool[bk * 64 + (i & 63)] / n;
    dpre[i] = m > 0.f ? g : g * (m + 1.f);
}
// y = elu(elu(a)) (iodine_module.py:415 + :493);  backward: da = dy * elu'(elu(a)) * elu'(a)
__global__ void io_elu2_kernel(const float* __restrict__ a, int lda, float* __restrict__ y, int ldy, long long rows, int F, int backward,
                               const float* __restrict__ dy, int lddy) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * F) return;
    const long long row = i / F;
    const int c = i % F;
    const float v = a[row * lda + c];
    if (!backward) y[row * ldy + c] = elu_(elu_(v));
    else {
        const float e1 = elu_(v);
        const float d1 = v > 0.f ? 1.f : e1 + 1.f, d2 = e1 > 0.f ? 1.f : elu_(e1) + 1.f;
        y[row * ldy + c] = dy[row * lddy + c] * d1 * d2;
    }
}

// ------------------------------------------------------------------------------------------- LSTM cell (torch gate order i, f, g, o)
__global__ void io_lstm_fwd_kernel(const float* __restrict__ gates, const float* __restrict__ c0, float* __restrict__ acts, float* __restrict__ c1,
                                   float* __restrict__ h1, long long rows, int H) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * H) return;
    const long long row = i / H;
    const int c = i % H;
    const float* g = gates + row * 4 * H;
    const float gi = 1.f / (1.f + expf(-g[c])), gf = 1.f / (1.f + expf(-g[H + c])), gg = tanhf(g[2 * H + c]), go = 1.f / (1.f + expf(-g[3 * H + c]));
    const float cn = gf * c0[i] + gi * gg;
    float* a = acts + row * 4 * H;
    a[c] = gi; a[H + c] = gf; a[2 * H + c] = gg; a[3 * H + c] = go;
    c1[i] = cn;
    h1[i] = go * tanhf(cn);
}
// dgates (pre-activation) and dc0 from dh1, dc1 (either may be null = zero)
__global__ void io_lstm_bwd_kernel(const float* __restrict__ acts, const float* __restrict__ c0, const float* __restrict__ c1, const float* __restrict__ dh1,
                                   const float* __restrict__ dc1, float* __restrict__ dgates, float* __restrict__ dc0, long long rows, int H) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * H) return;
    const long long row = i / H;
    const int c = i % H;
    const float* a = acts + row * 4 * H;
    const float gi = a[c], gf = a[H + c], gg = a[2 * H + c], go = a[3 * H + c];
    const float tc = tanhf(c1[i]);
    const float dh = dh1 ? dh1[i] : 0.f;
    const float dc = (dc1 ? dc1[i] : 0.f) + dh * go * (1.f - tc * tc);
    float* d = dgates + row * 4 * H;
    d[c] = dc * gg * gi * (1.f - gi);
    d[H + c] = dc * c0[i] * gf * (1.f - gf);
    d[2 * H + c] = dc * gi * (1.f - gg * gg);
    d[3 * H + c] = dh * tc * go * (1.f - go);
    dc0[i] = dc * gf;
}

// ------------------------------------------------------------------------------------------- posterior gradients
// gmu (+)= ds + kw * mu ;  gls (+)= ds * sigma * eps + kw * (sigma^2 - 1)      (kw = w_i * beta / B;  dlat (nullable) adds the
// latent-vector gradient: columns [0,L) -> mu, [L,2L) -> ls)
__global__ void io_post_grad_kernel(const float* __restrict__ mu, const float* __restrict__ ls, const float* __restrict__ eps, const float* __restrict__ ds,
                                    const float* __restrict__ dlat, int ldl, float kw, float* __restrict__ gmu, float* __restrict__ gls, long long BK, int L) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BK * L) return;
    const long long row = i / L;
    const int c = i % L;
    const float sg = expf(ls[i]);
    float a = ds[i] + kw * mu[i], b = ds[i] * sg * eps[i] + kw * (sg * sg - 1.f);
    if (dlat) { a += dlat[row * ldl + c]; b += dlat[row * ldl + L + c]; }
    gmu[i] += a;
    gls[i] += b;
}
// sqrt in place (L2 norm from the sum of squares)
__global__ void io_sqrt_kernel(float* x) { x[0] = sqrtf(x[0]); }
// sum of squares: blkpart[block] = this block's share (io_ordered_sum_kernel adds the blocks in a fixed order)
__global__ __launch_bounds__(256) void io_sumsq_kernel(const float* __restrict__ g, long long n, float* __restrict__ blkpart) {
    __shared__ float red[4];
    float a = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) a += g[i] * g[i];
    a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = a;
    __syncthreads();
    if (threadIdx.x == 0) blkpart[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// loss = -sum_i w_i (ll_i / B - beta kl_i / B); metrics: [0] loss, [1] mse (last iteration), [2] kld (last iteration)
__global__ void io_loss_kernel(const float* __restrict__ parts, float* __restrict__ metrics, int I, int B, float beta) {
    if (threadIdx.x || blockIdx.x) return;
    float loss = 0.f;
    for (int i = 0; i < I; ++i) {
        const float ll = parts[i * 4 + 0] / B, kl = parts[i * 4 + 2] / B;
        loss -= (float)(i + 1) / I * (ll - beta * kl);
    }
    metrics[0] = loss;
    metrics[1] = parts[(I - 1) * 4 + 1] / B;
    metrics[2] = parts[(I - 1) * 4 + 2] / B;
}

// =========================================================================================== launchers
#define IO_GRID(n) dim3(cdiv((n), 256)), dim3(256)
// kl_out[0] = sum of the elementwise KL terms (written, not accumulated); ws: n floats of scratch
int io_sample_launch(const float* mu, const float* ls, const float* noise, float* eps_out, float* slots, float* kl_out, long long n,
                     unsigned long long seed, unsigned site, float* ws, size_t ws_floats, hipStream_t st) {
    OCRL_REQUIRE(ws && ws_floats >= (size_t)n, "io_sample: scratch too small");
    hipLaunchKernelGGL(io_sample_kernel, IO_GRID(n), 0, st, mu, ls, noise, eps_out, slots, ws, n, seed, site);
    hipLaunchKernelGGL(io_ordered_sum_kernel, dim3(1), dim3(1024), 0, st, ws, n, 1ll, kl_out);
    OCRL_CHECK_LAUNCH("io_sample");
    return 0;
}
int io_w1_pack_launch(const float* W1, float* W1r, float* Wxy, int L, hipStream_t st) {
    hipLaunchKernelGGL(io_w1_pack_kernel, IO_GRID(9 * 64 * (L + 2)), 0, st, W1, W1r, Wxy, L);
    OCRL_CHECK_LAUNCH("io_w1_pack");
    return 0;
}
int io_p1_launch(const float* Wxy, const float* b1, float* P1, int S, hipStream_t st) {
    hipLaunchKernelGGL(io_p1_kernel, IO_GRID((long long)S * S * 64), 0, st, Wxy, b1, P1, S);
    OCRL_CHECK_LAUNCH("io_p1");
    return 0;
}
int io_class_sum_launch(const float* in, float* out, long long BK, int forward, hipStream_t st) {
    hipLaunchKernelGGL(io_class_sum_kernel, IO_GRID(BK * 9 * 64), 0, st, in, out, BK, forward);
    OCRL_CHECK_LAUNCH("io_class_sum");
    return 0;
}
int io_layer1_launch(const float* P1, const float* T, float* c1, long long BK, int S, hipStream_t st) {
    hipLaunchKernelGGL(io_layer1_kernel, IO_GRID(BK * S * S * 16), 0, st, P1, T, c1, BK, S);
    OCRL_CHECK_LAUNCH("io_layer1");
    return 0;
}
// dT [BK][9][64] is written (not accumulated); ws: BK*S*3*64 floats of scratch
int io_layer1_bwd_launch(const float* g, float* dT, long long BK, int S, float* ws, size_t ws_floats, hipStream_t st) {
    OCRL_REQUIRE(S >= 3 && ws && ws_floats >= (size_t)BK * S * 3 * 64, "io_layer1_bwd: S < 3 or scratch too small");
    hipLaunchKernelGGL(io_layer1_bwd_kernel, dim3((unsigned)(BK * S)), dim3(256), 0, st, g, ws, S);
    const long long n = BK * 9 * 64;
    hipLaunchKernelGGL(io_layer1_reduce_kernel, IO_GRID(n), 0, st, ws, dT, n, S);
    OCRL_CHECK_LAUNCH("io_layer1_bwd");
    return 0;
}
int io_w1_grad_launch(const float* dW1r, const float* G, float* dW1, float* db1, int S, int L, hipStream_t st) {
    hipLaunchKernelGGL(io_w1_grad_kernel, dim3(9), dim3(256), 0, st, dW1r, G, dW1, db1, S, L);
    OCRL_CHECK_LAUNCH("io_w1_grad");
    return 0;
}
// st1 [B*K][4] is written when enc is given; part[0..1] += (log-likelihood, squared error) totals; ws: (B*S*S/256) * 66 floats of scratch
int io_elbo_launch(const float* out4, const float* obs, int B, int K, int S, float sigma, float* enc, float* st1, float* dout4, float* part,
                   float* masks_out, float* recon_out, float* rmasked_out, float* ws, size_t ws_floats, hipStream_t st) {
    OCRL_REQUIRE(K >= 1 && K <= 16 && (S * S) % 256 == 0, "iodine elbo: 1 <= K <= 16 and S %% 16 == 0 required (K=%d S=%d)", K, S);
    const int nb_img = S * S / 256;
    OCRL_REQUIRE(ws && ws_floats >= (size_t)B * nb_img * 66, "iodine elbo: scratch too small");
    const dim3 grid((unsigned)((long long)B * nb_img));
    if (K <= 8) {
        hipLaunchKernelGGL(io_elbo_kernel<8>, grid, dim3(256), 0, st, out4, obs, B, K, S, sigma, enc, ws, dout4, masks_out, recon_out, rmasked_out);
        hipLaunchKernelGGL(io_elbo_reduce_kernel<8>, dim3(1), dim3(256), 0, st, ws, st1, part, B, K, nb_img, enc ? 1 : 0);
    } else {
        hipLaunchKernelGGL(io_elbo_kernel<16>, grid, dim3(256), 0, st, out4, obs, B, K, S, sigma, enc, ws, dout4, masks_out, recon_out, rmasked_out);
        hipLaunchKernelGGL(io_elbo_reduce_kernel<16>, dim3(1), dim3(256), 0, st, ws, st1, part, B, K, nb_img, enc ? 1 : 0);
    }
    OCRL_CHECK_LAUNCH("io_elbo");
    return 0;
}
// st2 [BK][4] is written; ws: BK * N/256 * 4 floats of scratch
int io_enc_norm_launch(float* enc, const float* st1, float* st2, long long BK, int N, float* ws, size_t ws_floats, hipStream_t st) {
    OCRL_REQUIRE(N % 256 == 0, "iodine encoding: S %% 16 == 0 required");
    OCRL_REQUIRE(ws && ws_floats >= (size_t)BK * (N / 256) * 4, "iodine encoding: scratch too small");
    hipLaunchKernelGGL(io_enc_var_kernel, dim3((unsigned)(BK * N / 256)), dim3(256), 0, st, enc, st1, ws, N);
    hipLaunchKernelGGL(io_enc_var_reduce_kernel, IO_GRID(BK * 4), 0, st, ws, st2, BK * 4, N / 256);
    hipLaunchKernelGGL(io_enc_norm_kernel, dim3((unsigned)(BK * N / 256)), dim3(256), 0, st, enc, st1, st2, N);
    OCRL_CHECK_LAUNCH("io_enc_norm");
    return 0;
}
int io_elbo_bwd_launch(const float* out4, const float* obs, const float* denc, int B, int K, int S, float sigma, float cw, float* dout4, hipStream_t st) {
    OCRL_REQUIRE(K >= 1 && K <= 16 && (S * S) % 256 == 0, "iodine elbo bwd: 1 <= K <= 16 and S %% 16 == 0 required");
    const dim3 grid((unsigned)((long long)B * S * S / 256));
    if (K <= 8) hipLaunchKernelGGL(io_elbo_bwd_kernel<8>, grid, dim3(256), 0, st, out4, obs, denc, B, K, S, sigma, cw, dout4);
    else hipLaunchKernelGGL(io_elbo_bwd_kernel<16>, grid, dim3(256), 0, st, out4, obs, denc, B, K, S, sigma, cw, dout4);
    OCRL_CHECK_LAUNCH("io_elbo_bwd");
    return 0;
}
int io_latent_launch(const float* mu, const float* ls, const float* eps, const float* ds, float* latent, long long BK, int L, float beta, int layer_norm,
                     int ld, hipStream_t st) {
    hipLaunchKernelGGL(io_latent_kernel, dim3(cdiv(BK, 4)), dim3(256), 0, st, mu, ls, eps, ds, latent, BK, L, beta, layer_norm, ld);
    OCRL_CHECK_LAUNCH("io_latent");
    return 0;
}
int io_im2col_launch(const float* x, float* col, long long Bn, int C, int Hi, int Wi, int ldc, hipStream_t st) {
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    const long long rows = Bn * Ho * Wo;
    if ((ldc & 3) == 0 && (((uintptr_t)col) & 15) == 0 && (((uintptr_t)x) & 15) == 0 && Ho <= 65535 && Bn <= 65535)
        hipLaunchKernelGGL(io_im2col4_kernel, dim3((unsigned)((Wo * (ldc >> 2) + 255) / 256), (unsigned)Ho, (unsigned)Bn), dim3(256), 0, st, x, col, C, Hi, Wi, Ho, Wo, ldc);
    else
        hipLaunchKernelGGL(io_im2col_kernel, IO_GRID(rows * ldc), 0, st, x, col, rows, C, Hi, Wi, Ho, Wo, ldc);
    OCRL_CHECK_LAUNCH("io_im2col");
    return 0;
}
int io_col2im_launch(const float* dcol, const float* act, float* dx, long long Bn, int C, int Hi, int Wi, int ldc, hipStream_t st) {
    const int Ho = (Hi - 1) / 2 + 1, Wo = (Wi - 1) / 2 + 1;
    const long long n = Bn * Hi * Wi * C;
    if (Hi <= 65535 && Bn <= 65535)
        hipLaunchKernelGGL(io_col2im3_kernel, dim3((unsigned)((Wi * C + 255) / 256), (unsigned)Hi, (unsigned)Bn), dim3(256), 0, st, dcol, act, dx, C, Hi, Wi, Ho, Wo, ldc);
    else
        hipLaunchKernelGGL(io_col2im_kernel, IO_GRID(n), 0, st, dcol, act, dx, n, C, Hi, Wi, Ho, Wo, ldc);
    OCRL_CHECK_LAUNCH("io_col2im");
    return 0;
}
int io_refw_pack_launch(const float* W, float* Wp, int C, int ldc, hipStream_t st) {
    hipLaunchKernelGGL(io_refw_pack_kernel, IO_GRID(64 * ldc), 0, st, W, Wp, C, ldc, 0, nullptr, nullptr);
    OCRL_CHECK_LAUNCH("io_refw_pack");
    return 0;
}
int io_refw_unpack_launch(const float* dWp, float* dW, int C, int ldc, hipStream_t st) {
    hipLaunchKernelGGL(io_refw_pack_kernel, IO_GRID(64 * ldc), 0, st, nullptr, nullptr, C, ldc, 1, dW, dWp);
    OCRL_CHECK_LAUNCH("io_refw_unpack");
    return 0;
}
int io_pool_launch(const float* r, float* pool, long long BK, int n, hipStream_t st) {
    hipLaunchKernelGGL(io_pool_kernel, dim3((unsigned)BK), dim3(256), 0, st, r, pool, n);
    OCRL_CHECK_LAUNCH("io_pool");
    return 0;
}
int io_pool_bwd_launch(const float* dpool, const float* r, float* dpre, long long BK, int n, hipStream_t st) {
    hipLaunchKernelGGL(io_pool_bwd_kernel, IO_GRID(BK * n * 64), 0, st, dpool, r, dpre, BK * n * 64, n);
    OCRL_CHECK_LAUNCH("io_pool_bwd");
    return 0;
}
int io_elu2_launch(const float* a, int lda, float* y, int ldy, long long rows, int F, const float* dy, int lddy, hipStream_t st) {
    hipLaunchKernelGGL(io_elu2_kernel, IO_GRID(rows * F), 0, st, a, lda, y, ldy, rows, F, dy ? 1 : 0, dy, lddy);
    OCRL_CHECK_LAUNCH("io_elu2");
    return 0;
}
int io_lstm_fwd_launch(const float* gates, const float* c0, float* acts, float* c1, float* h1, long long rows, int H, hipStream_t st) {
    hipLaunchKernelGGL(io_lstm_fwd_kernel, IO_GRID(rows * H), 0, st, gates, c0, acts, c1, h1, rows, H);
    OCRL_CHECK_LAUNCH("io_lstm_fwd");
    return 0;
}
int io_lstm_bwd_launch(const float* acts, const float* c0, const float* c1, const float* dh1, const float* dc1, float* dgates, float* dc0,
                       long long rows, int H, hipStream_t st) {
    hipLaunchKernelGGL(io_lstm_bwd_kernel, IO_GRID(rows * H), 0, st, acts, c0, c1, dh1, dc1, dgates, dc0, rows, H);
    OCRL_CHECK_LAUNCH("io_lstm_bwd");
    return 0;
}
int io_post_grad_launch(const float* mu, const float* ls, const float* eps, const float* ds, const float* dlat, int ldl, float kw, float* gmu,
                        float* gls, long long BK, int L, hipStream_t st) {
    hipLaunchKernelGGL(io_post_grad_kernel, IO_GRID(BK * L), 0, st, mu, ls, eps, ds, dlat, ldl, kw, gmu, gls, BK, L);
    OCRL_CHECK_LAUNCH("io_post_grad");
    return 0;
}
// ws: 1024 floats of scratch
int io_l2norm_launch(const float* g, long long n, float* out, float* ws, size_t ws_floats, hipStream_t st) {
    OCRL_REQUIRE(ws && ws_floats >= 1024, "io_l2norm: scratch too small");
    hipLaunchKernelGGL(io_sumsq_kernel, dim3(1024), dim3(256), 0, st, g, n, ws);
    hipLaunchKernelGGL(io_ordered_sum_kernel, dim3(1), dim3(1024), 0, st, ws, 1024ll, 1ll, out);
    hipLaunchKernelGGL(io_sqrt_kernel, dim3(1), dim3(1), 0, st, out);
    OCRL_CHECK_LAUNCH("io_l2norm");
    return 0;
}
int io_loss_launch(const float* parts, float* metrics, int I, int B, float beta, hipStream_t st) {
    hipLaunchKernelGGL(io_loss_kernel, dim3(1), dim3(1), 0, st, parts, metrics, I, B, beta);
    OCRL_CHECK_LAUNCH("io_loss");
    return 0;
}
