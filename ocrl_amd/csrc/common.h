// Shared helpers for the ocrl_hip kernels (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- error convention: every C-ABI entry returns 0 on success; message via ocrl_last_error()
void ocrl_set_error(const char* fmt, ...);

#define OCRL_REQUIRE(cond, ...)                         \
    do {                                                \
        if (!(cond)) {                                  \
            ocrl_set_error(__VA_ARGS__);                \
            return 1;                                   \
        }                                               \
    } while (0)

#define OCRL_CHECK_LAUNCH(name)                                              \
    do {                                                                     \
        hipError_t e__ = hipGetLastError();                                  \
        if (e__ != hipSuccess) {                                             \
            ocrl_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
            return 1;                                                        \
        }                                                                    \
    } while (0)

#define OCRL_HIP(call)                                                       \
    do {                                                                     \
        hipError_t e__ = (call);                                             \
        if (e__ != hipSuccess) {                                             \
            ocrl_set_error("%s failed: %s", #call, hipGetErrorString(e__));  \
            return 1;                                                        \
        }                                                                    \
    } while (0)

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// ---- counter-based RNG (Philox-2x32, 7 rounds): stateless, so forward and backward
//      regenerate the same dropout decisions without storing masks.
__host__ __device__ inline uint2 philox2x32(uint32_t c0, uint32_t c1, uint32_t key) {
#pragma unroll
    for (int r = 0; r < 7; ++r) {
        const uint64_t p = (uint64_t)c0 * 0xD256D193u;
        const uint32_t hi = (uint32_t)(p >> 32), lo = (uint32_t)p;
        c0 = hi ^ key ^ c1;
        c1 = lo;
        key += 0x9E3779B9u;
    }
    return make_uint2(c0, c1);
}

// 64 random bits for the group of four consecutive elements idx4 = element_index / 4 at `site`.
__host__ __device__ inline uint2 rng_bits4(uint64_t seed, uint32_t site, uint64_t idx4) {
    const uint32_t key = (uint32_t)seed ^ (site * 0x85EBCA6Bu) ^ (uint32_t)(idx4 >> 32) * 0xC2B2AE35u;
    return philox2x32((uint32_t)idx4, (uint32_t)(seed >> 32) ^ site, key);
}

// keep decision for element e (0..3) of a group: 16-bit uniform >= thresh, thresh = round(p * 65536)
__host__ __device__ inline bool rng_keep(uint2 bits, int e, uint32_t thresh) {
    const uint32_t w = (e & 2) ? bits.y : bits.x;
    const uint32_t u = (e & 1) ? (w >> 16) : (w & 0xFFFFu);
    return u >= thresh;
}
__host__ __device__ inline uint32_t drop_thresh(float p) { return (uint32_t)(p * 65536.0f + 0.5f); }

// two 24-bit uniforms in (0,1) from a 64-bit draw
__host__ __device__ inline float u01_24(uint32_t w) { return ((float)(w >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// dropout site ids (shared by forward, backward and the mask dump used by the parity tests)
enum : uint32_t {
    SITE_ZPOS = 1,
    SITE_BLK_BASE = 16,   // + 8 * block + {0: self.attn, 1: self.out, 2: cross.attn, 3: cross.out, 4: ffn}
    SITE_GUMBEL_Z = 200,
    SITE_GUMBEL_ZH = 201,
    SITE_SLOT_NOISE = 202,
};

__device__ inline float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ inline float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
