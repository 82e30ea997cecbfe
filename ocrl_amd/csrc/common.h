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

// ---- counter-based RNG: stateless, so forward and backward regenerate the same dropout decisions without storing masks.
// 64 bits per call from two passes of a 32-bit avalanche mixer (two multiplies and three xor-shifts each; the "lowbias32" constants,
// bias 0.17 %) over the element counter xor a per-(seed, site) key, with the key added once more inside the mixer (rng_mix32k).  Round 1 used Philox-2x32-7 (14 quarter-rate integer multiplies per
// call); this form needs 4 (measured: +0.4 % images/s — the Gumbel kernel turned out to be bound by its logs and exp, not the draws).
// tests/test_gpu_determinism.py::test_device_rng_quality checks rates, serial and cross-site / cross-seed correlations.
__host__ __device__ inline uint32_t rng_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}

// 64 random bits for the group of four consecutive elements idx4 = element_index / 4 at `site`.
// key: uniform over a launch (scalar work); the high counter bits enter here so that >2^32 groups do not repeat
__host__ __device__ inline uint32_t rng_key(uint64_t seed, uint32_t site, uint32_t idx4_hi) {
    return rng_mix32((uint32_t)seed ^ (site * 0x9E3779B9u)) ^ rng_mix32((uint32_t)(seed >> 32) + 0x85EBCA6Bu * (idx4_hi + 1u));
}
// The key enters twice: xor-ed into the counter and, through a second word derived from it, ADDED between the mixer's two multiplies.
// With the xor alone every (seed, site) stream was the same fixed permutation of the counter, translated: stream A at group i equalled
// stream B at group i ^ (kA ^ kB), so two steps or sites could share a whole noise field as a block permutation.  The addition after the
// first multiply does not commute with the xor-translate (tests/test_gpu_determinism.py::test_device_rng_streams_are_not_translates).
__host__ __device__ inline uint32_t rng_mix32k(uint32_t x, uint32_t k2) {
    x ^= x >> 16; x *= 0x7FEB352Du;
    x += k2;
    x ^= x >> 15; x *= 0x846CA68Bu;
    x ^= x >> 16;
    return x;
}
__host__ __device__ inline uint2 rng_bits4_keyed(uint32_t key, uint32_t idx4_lo) {
    const uint32_t c = idx4_lo ^ key;
    const uint32_t k2 = key * 0x9E3779B9u + 0x7F4A7C15u;      // uniform over a launch: scalar work
    return make_uint2(rng_mix32k(c, k2), rng_mix32k(c ^ 0x68E31DA4u, k2) + key);
}
// the first word alone (callers that need one uniform per counter)
__host__ __device__ inline uint32_t rng_bits1_keyed(uint32_t key, uint32_t idx_lo) {
    return rng_mix32k(idx_lo ^ key, key * 0x9E3779B9u + 0x7F4A7C15u);
}
__host__ __device__ inline uint2 rng_bits4(uint64_t seed, uint32_t site, uint64_t idx4) {
    return rng_bits4_keyed(rng_key(seed, site, (uint32_t)(idx4 >> 32)), (uint32_t)idx4);
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
