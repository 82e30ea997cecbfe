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
// The kernel and its tile dispatch live in gemm_impl.h, compiled per operand-layout family (gemm_tt.hip: A [M,K] x B [N,K];
// gemm_tn.hip: A [M,K] x B [K,N]; gemm_nn.hip: A [K,M]); this file checks the arguments, picks the family and owns the split-k reduction.
#include "common.h"
#include "kernels.h"
#include <stdlib.h>

int gemm_launch_tt(const GemmArgs& a, hipStream_t st);
int gemm_launch_tn(const GemmArgs& a, hipStream_t st);
int gemm_launch_nn(const GemmArgs& a, hipStream_t st);

// out[i] = (accumulate ? out[i] : 0) + sum_s part[s*stride + i]   (n % 4 == 0)
// block = 16 float4 columns x 16 split lanes: the slabs are read in parallel and combined through LDS.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, long long n4,
                                                            int splits, long long stride, int accumulate) {
    __shared__ float4 red[16][16];
    const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long long i = (long long)blockIdx.x * 16 + cl;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < n4) {
        int k = sl;
        for (; k + 48 < splits; k += 64) {          // four slabs in flight per thread
            const float4 v0 = *reinterpret_cast<const float4*>(part + (size_t)k * stride + i * 4);
            const float4 v1 = *reinterpret_cast<const float4*>(part + (size_t)(k + 16) * stride + i * 4);
            const float4 v2 = *reinterpret_cast<const float4*>(part + (size_t)(k + 32) * stride + i * 4);
            const float4 v3 = *reinterpret_cast<const float4*>(part + (size_t)(k + 48) * stride + i * 4);
            s.x += (v0.x + v1.x) + (v2.x + v3.x); s.y += (v0.y + v1.y) + (v2.y + v3.y);
            s.z += (v0.z + v1.z) + (v2.z + v3.z); s.w += (v0.w + v1.w) + (v2.w + v3.w);
        }
        for (; k < splits; k += 16) {
            const float4 v = *reinterpret_cast<const float4*>(part + (size_t)k * stride + i * 4);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
    }
    red[sl][cl] = s;
    __syncthreads();
    if (sl == 0 && i < n4) {
        float4 t = accumulate ? *reinterpret_cast<const float4*>(out + i * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 16; ++k) { t.x += red[k][cl].x; t.y += red[k][cl].y; t.z += red[k][cl].z; t.w += red[k][cl].w; }
        *reinterpret_cast<float4*>(out + i * 4) = t;
    }
}

int gemm_launch(const GemmArgs& a_in, hipStream_t st) {
    GemmArgs a = a_in;
    OCRL_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "gemm: empty problem %d %d %d", a.M, a.N, a.K);
    OCRL_REQUIRE(a.batch >= 1 && a.splitk >= 1, "gemm: bad batch/splitk");
    if (a.akc) OCRL_REQUIRE(a.K % 4 == 0 && a.lda % 4 == 0, "gemm: A k-contiguous needs K,lda %% 4 == 0 (K=%d lda=%d)", a.K, a.lda);
    else OCRL_REQUIRE(a.M % 4 == 0 && a.lda % 4 == 0, "gemm: A m-contiguous needs M,lda %% 4 == 0 (M=%d lda=%d)", a.M, a.lda);
    if (a.bkc) OCRL_REQUIRE(a.K % 4 == 0 && a.ldb % 4 == 0, "gemm: B k-contiguous needs K,ldb %% 4 == 0 (K=%d ldb=%d)", a.K, a.ldb);
    else OCRL_REQUIRE(a.N % 4 == 0 && a.ldb % 4 == 0, "gemm: B n-contiguous needs N,ldb %% 4 == 0 (N=%d ldb=%d)", a.N, a.ldb);
    OCRL_REQUIRE(((uintptr_t)a.A & 15) == 0 && ((uintptr_t)a.B & 15) == 0, "gemm: operands must be 16-byte aligned");
    OCRL_REQUIRE((a.sA % 4) == 0 && (a.sB % 4) == 0 && (a.sAi % 4) == 0 && (a.sBi % 4) == 0, "gemm: batch strides must be multiples of 4");
    OCRL_REQUIRE(a.batch_inner >= 1 && a.batch % a.batch_inner == 0, "gemm: batch must be a multiple of batch_inner");
    OCRL_REQUIRE(a.splitk == 1 || a.batch == 1, "gemm: split-k with batches is not supported");
    if (a.splitk > 1) OCRL_REQUIRE(a.sCsplit >= (long long)(a.M - 1) * a.ldc + a.N, "gemm: split-k slab stride too small");
    if (a.adrop_p > 0.f) OCRL_REQUIRE(a.adrop_ld > 0 && a.adrop_ld % 4 == 0 && a.batch == 1, "gemm: A-dropout needs adrop_ld %% 4 == 0 and no batching");
    if (a.bias_out) OCRL_REQUIRE(!a.akc && (a.splitk == 1 || a.sBias >= a.M), "gemm: fused bias gradient needs the dW form");
    OCRL_REQUIRE(!(a.a_mode || a.b_mode) || (a.batch == 1 && a.adrop_p == 0.f && a.x_lse && (a.a_mode != 3 || a.x_tok) &&
                 (a.a_mode == 0 || a.a_mode == 2 || a.a_mode == 3) && (a.b_mode == 0 || a.b_mode == 2)), "gemm: bad operand transform arguments");
    if (a.epi_mode) {
        OCRL_REQUIRE(a.akc && a.batch == 1 && a.splitk == 1 && !a.a_mode && !a.b_mode && a.adrop_p == 0.f && a.relu == 0 && a.drop_p == 0.f && !a.resid,
                     "gemm: soft-max epilogue needs a plain k-contiguous A, no split-k / batches / activation");
        OCRL_REQUIRE((a.N & 3) == 0 && (a.ldc & 3) == 0 && (((uintptr_t)a.C) & 15) == 0 && (!a.bias || (((uintptr_t)a.bias) & 15) == 0),
                     "gemm: soft-max epilogue needs N, ldc multiples of 4 and 16-byte aligned C / bias");
        if (a.epi_mode == 1 || a.epi_mode == 2) {
            OCRL_REQUIRE(a.bkc && a.stat && !a.mask && gemm_stat_segments(a.N) <= 64, "gemm: soft-max statistics need a k-contiguous B, a stat buffer and N <= 4096");
            if (a.epi_mode == 2) {
                OCRL_REQUIRE(a.hstat && a.hidx && (a.e1 == nullptr) == (a.e2 == nullptr) && a.ldc == a.N, "gemm: Gumbel head arguments");
                OCRL_REQUIRE(!a.e1 || ((((uintptr_t)a.e1) | ((uintptr_t)a.e2)) & 15) == 0, "gemm: Gumbel noise must be 16-byte aligned");
                return gemm_launch_tt(a, st);
            }
            return gemm_launch_tt(a, st);
        }
        OCRL_REQUIRE(a.epi_mode == 3 && !a.bkc && a.mask && a.e_lse && a.e_rowvec && !a.bias && (a.ldmask & 3) == 0 && (((uintptr_t)a.mask) & 15) == 0,
                     "gemm: soft-max backward epilogue arguments");
        return gemm_launch_tn(a, st);
    }
    if (a.akc && a.bkc) return gemm_launch_tt(a, st);
    if (a.akc && !a.bkc) return gemm_launch_tn(a, st);
    return gemm_launch_nn(a, st);
}

int splitk_reduce_launch(const float* part, float* out, long long n, int splits, long long stride,
                         int accumulate, hipStream_t st) {
    OCRL_REQUIRE(n % 4 == 0 && stride % 4 == 0, "splitk_reduce: n and stride must be multiples of 4");
    const long long n4 = n / 4;
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(cdiv(n4, 16)), dim3(256), 0, st, part, out, n4, splits, stride, accumulate);
    OCRL_CHECK_LAUNCH("splitk_reduce");
    return 0;
}
