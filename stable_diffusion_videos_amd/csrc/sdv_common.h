// Internal helpers shared by the gfx950 kernels of libsdv_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/sdv_hip.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;   // 8 bf16 = 4 VGPRs (MFMA A/B fragment)
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;   // 4 bf16 = 8 bytes
typedef __attribute__((ext_vector_type(16))) float f32x16_t;  // 32x32 MFMA accumulator
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

#define SDV_DEVICE __device__ __forceinline__

// ---- bf16 <-> fp32 (round-to-nearest-even, NaN preserved) ------------------------------------
SDV_DEVICE float bf16_to_f32(uint16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
SDV_DEVICE uint16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw_t;
// two fp32 -> packed bf16x2 with the gfx950 hardware converter (v_cvt_pk_bf16_f32, round-to-nearest-even)
SDV_DEVICE uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_hw_t));
}
SDV_DEVICE float silu_f(float x) { return x / (1.0f + __expf(-x)); }
SDV_DEVICE float quick_gelu_f(float x) { return x / (1.0f + __expf(-1.702f * x)); }   // CLIP 'quick_gelu'
SDV_DEVICE float lrelu02_f(float x) { return x > 0.f ? x : 0.2f * x; }   // nn.LeakyReLU(negative_slope=0.2), RRDBNet
// exact-erf GELU (F.gelu default, what diffusers' GEGLU uses) with a branch-free erf: Abramowitz-Stegun 7.1.26,
// |abs error| < 1.5e-7 - three orders of magnitude below the bf16 output rounding - one v_rcp + one v_exp
// instead of libm erff's divergent branches (the GEGLU epilogue evaluates it 168 M times per UNet forward).
SDV_DEVICE float erf_as_f(float x) {
    const float ax = fabsf(x);
    const float t = __frcp_rn(1.0f + 0.3275911f * ax);
    float y = 1.061405429f;
    y = y * t - 1.453152027f;
    y = y * t + 1.421413741f;
    y = y * t - 0.284496736f;
    y = y * t + 0.254829592f;
    y = 1.0f - y * t * __expf(-ax * ax);
    return copysignf(y, x);
}
SDV_DEVICE float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erf_as_f(x * 0.70710678118654752440f)); }
// The same function (same Abramowitz-Stegun 7.1.26 erfc, |abs error of Phi| < 0.8e-7) arranged for the GEGLU epilogue, which
// evaluates it 80 times per lane and tile and is VALU-bound: with h = 0.5 * erfc(|x| / sqrt 2) = P'(t) * exp2(-(k x)^2),
// t = 1 / (1 + p' |x|) (the 1/sqrt 2, the 0.5 and log2 e folded into p', the polynomial and k), gelu(x) = max(x, 0) - |x| h:
// 11 plain VALU + v_rcp + v_exp instead of 18 + 2 (no copysign, no 1 + erf, no separate scaling of the argument).
SDV_DEVICE float gelu_erf_fast_f(float x) {
    // (|x| capped at a finite value: at x = +inf the last FMA would otherwise be fma(-inf, 0, inf) = NaN where gelu(+inf) = +inf -
    //  an overflowed GEGLU gate has to saturate, not poison the residual stream; one v_min with the |.| source modifier)
    const float ax = fminf(fabsf(x), 1e18f);
    const float t = __frcp_rn(__builtin_fmaf(0.3275911f * 0.70710678118654752440f, ax, 1.0f));
    const float u = x * 0.84932180028801904272f;                  // sqrt(0.5 * log2(e))
    const float e = __builtin_amdgcn_exp2f(-(u * u));
    float y = 0.5f * 1.061405429f;
    y = __builtin_fmaf(y, t, 0.5f * -1.453152027f);
    y = __builtin_fmaf(y, t, 0.5f * 1.421413741f);
    y = __builtin_fmaf(y, t, 0.5f * -0.284496736f);
    y = __builtin_fmaf(y, t, 0.5f * 0.254829592f);
    const float h = y * t * e;
    return __builtin_fmaf(-ax, h, fmaxf(x, 0.0f));
}

// 16-byte vector of 8 bf16 <-> 8 floats
struct alignas(16) bf16x8_raw { uint32_t w[4]; };
SDV_DEVICE void unpack8(const bf16x8_raw& r, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f[2 * i] = __uint_as_float(r.w[i] << 16);
        f[2 * i + 1] = __uint_as_float(r.w[i] & 0xffff0000u);
    }
}
SDV_DEVICE bf16x8_raw pack8(const float* f) {
    bf16x8_raw r;
#pragma unroll
    for (int i = 0; i < 4; ++i) r.w[i] = pack_bf16x2(f[2 * i], f[2 * i + 1]);
    return r;
}

// ---- async global -> LDS copy (16 B per lane, LDS destination = wave-uniform base + lane*16) ---
SDV_DEVICE void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

SDV_DEVICE float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
SDV_DEVICE float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// ---- host side error plumbing ------------------------------------------------------------------
__attribute__((visibility("hidden"))) void sdv_set_error(const char* fmt, ...);   // (internal: not part of the C ABI)
#define SDV_REQUIRE(cond, ...)            \
    do {                                  \
        if (!(cond)) {                    \
            sdv_set_error(__VA_ARGS__);   \
            return SDV_ERR_ARG;           \
        }                                 \
    } while (0)
#define SDV_CHECK_LAUNCH(name)                                                        \
    do {                                                                              \
        hipError_t e__ = hipGetLastError();                                           \
        if (e__ != hipSuccess) {                                                      \
            sdv_set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
            return SDV_ERR_LAUNCH;                                                    \
        }                                                                             \
    } while (0)
