// Global inf-norm gradient clip + Adam on flat fp32 buffers (reference: ocrs/base.py:65-72,
// torch.nn.utils.clip_grad_norm_(…, "inf") and torch.optim.Adam defaults).
#include "common.h"
#include "kernels.h"

// part[blk] = max |g| over the block's grid-stride range
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ g, long long n4, float* __restrict__ part) {
    __shared__ float red[4];
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(g)[i];
        m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}
__global__ void absmax_final_kernel(const float* __restrict__ part, int n, float* __restrict__ out) {
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 64) m = fmaxf(m, part[i]);
    m = wave_max(m);
    if (threadIdx.x == 0) out[0] = m;
}

// g *= min(1, clip/(norm+1e-6)) (clip <= 0: no clipping); then the Adam update, all in one pass.
__global__ __launch_bounds__(256) void clip_adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long long n4, const float* __restrict__ norm, float clip,
                                                       float lr, float b1, float b2, float eps, float bc1, float bc2_sqrt, float gscale) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float coef = gscale;
    if (clip > 0.f) coef *= fminf(1.0f, clip / (norm[0] * gscale + 1e-6f));
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i], pv = reinterpret_cast<float4*>(p)[i];
    const float step = lr / bc1;
    auto upd = [&](float gg, float& mm, float& vq, float& pp) {
        gg *= coef;
        mm = mm + (gg - mm) * (1.f - b1);                  // exp_avg.lerp_(grad, 1 - beta1)
        vq = vq * b2 + (1.f - b2) * gg * gg;
        const float denom = sqrtf(vq) / bc2_sqrt + eps;
        pp -= step * (mm / denom);
    };
    upd(gv.x, mv.x, vv.x, pv.x); upd(gv.y, mv.y, vv.y, pv.y); upd(gv.z, mv.z, vv.z, pv.z); upd(gv.w, mv.w, vv.w, pv.w);
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
    reinterpret_cast<float4*>(p)[i] = pv;
}

int absmax_launch(const float* g, long long n, float* out, float* ws, size_t ws_floats, hipStream_t st) {
    OCRL_REQUIRE(n % 4 == 0 && ws_floats >= 1024, "absmax: n %% 4 != 0 or workspace too small");
    int nblk = cdiv(n / 4, 256);
    if (nblk > 1024) nblk = 1024;
    hipLaunchKernelGGL(absmax_kernel, dim3(nblk), dim3(256), 0, st, g, n / 4, ws);
    OCRL_CHECK_LAUNCH("absmax");
    hipLaunchKernelGGL(absmax_final_kernel, dim3(1), dim3(64), 0, st, ws, nblk, out);
    OCRL_CHECK_LAUNCH("absmax_final");
    return 0;
}
int clip_adam_launch(float* p, const float* g, float* m, float* v, long long n, const float* norm, float clip, float lr, float b1,
                     float b2, float eps, int step, float gscale, hipStream_t st) {
    OCRL_REQUIRE(n % 4 == 0 && step >= 1, "clip_adam: n %% 4 != 0 or step < 1");
    const double bc1 = 1.0 - pow((double)b1, step), bc2 = 1.0 - pow((double)b2, step);
    hipLaunchKernelGGL(clip_adam_kernel, dim3(cdiv(n / 4, 256)), dim3(256), 0, st, p, g, m, v, n / 4, norm, clip, lr, b1, b2, eps,
                       (float)bc1, (float)sqrt(bc2), gscale);
    OCRL_CHECK_LAUNCH("clip_adam");
    return 0;
}
