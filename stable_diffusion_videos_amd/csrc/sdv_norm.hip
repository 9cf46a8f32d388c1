// GroupNorm (+SiLU, + skip-connection concat) and LayerNorm over NHWC / token-major bf16 activations.
// Replaces nn.GroupNorm / F.silu / torch.cat / nn.LayerNorm inside unet(...) and vae.decode(...)
// (/root/reference/.../stable_diffusion_pipeline.py:418, :433).  HBM-bound: every access is a 16-byte
// (8 x bf16) vector, each thread keeps ONE channel chunk so the affine parameters live in registers,
// statistics accumulate in fp32.
#include "sdv_common.h"

namespace {

// Thread layout shared by stats/apply: a block walks `npix` pixels of one image; thread t owns channel
// chunk (t % CT) [8 channels] and pixel lane (t / CT); chunks beyond 256 are looped.
struct GnGeom {
    int C, nchunk, CT, PT;
};
__device__ __forceinline__ GnGeom gn_geom(int C) {
    GnGeom g;
    g.C = C;
    g.nchunk = C >> 3;
    g.CT = g.nchunk < 256 ? g.nchunk : 256;
    g.PT = 256 / g.CT;
    return g;
}

__device__ __forceinline__ const uint16_t* gn_src(const uint16_t* X, const uint16_t* X2, int C1, int C2, long long pix,
                                                  int c0) {
    return c0 < C1 ? X + pix * C1 + c0 : X2 + pix * C2 + (c0 - C1);
}

// partials[img][split][group][2] = (sum, sumsq) over this block's pixel range.
// Deterministic: every thread parks its per-channel partial sums in LDS ([pixel-lane][channel][2]) and one
// thread per group folds them in a fixed order - no atomics, so frames are bit-reproducible run to run.
__global__ __launch_bounds__(256) void gn_stats_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ X2,
                                                       int C1, int C2, int HW, int groups, int splits,
                                                       float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [PT][C][2]
    const int C = C1 + C2;
    const GnGeom g = gn_geom(C);
    const int cpg = C / groups;
    const int img = blockIdx.y, split = blockIdx.x;
    const int tid = threadIdx.x;
    const int per = (HW + splits - 1) / splits;
    const int p_begin = split * per;
    const int p_end = p_begin + per < HW ? p_begin + per : HW;
    const int tc = tid % g.CT, tp = tid / g.CT;
    if (tp < g.PT) {
        for (int ch = tc; ch < g.nchunk; ch += g.CT) {
            const int c0 = ch << 3;
            float s[8], q[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] = q[e] = 0.f;
            // 4 independent 16-byte loads in flight per thread (HBM-bound: memory-level parallelism is the lever)
            int p = p_begin + tp;
            for (; p + 3 * g.PT < p_end; p += 4 * g.PT) {
                bf16x8_raw r[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    r[u] = *(const bf16x8_raw*)gn_src(X, X2, C1, C2, (long long)img * HW + p + u * g.PT, c0);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    float f[8];
                    unpack8(r[u], f);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        s[e] += f[e];
                        q[e] += f[e] * f[e];
                    }
                }
            }
            for (; p < p_end; p += g.PT) {
                const bf16x8_raw r = *(const bf16x8_raw*)gn_src(X, X2, C1, C2, (long long)img * HW + p, c0);
                float f[8];
                unpack8(r, f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    s[e] += f[e];
                    q[e] += f[e] * f[e];
                }
            }
            float* dst = sm + ((long long)tp * C + c0) * 2;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                dst[2 * e] = s[e];
                dst[2 * e + 1] = q[e];
            }
        }
    }
    __syncthreads();
    if (tid < groups) {
        float as = 0.f, aq = 0.f;
        for (int t = 0; t < g.PT; ++t) {
            const float* src = sm + ((long long)t * C + tid * cpg) * 2;
            for (int c = 0; c < cpg; ++c) {
                as += src[2 * c];
                aq += src[2 * c + 1];
            }
        }
        float* out = partials + (((long long)img * splits + split) * groups + tid) * 2;
        out[0] = as;
        out[1] = aq;
    }
}

// 8 floats -> 8 OCP e4m3 bytes, saturating at +-448 (v_cvt_pk_fp8_f32 rounds to nearest even)
__device__ __forceinline__ uint2 pack8_fp8(const float* f, float q) {
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = fminf(fmaxf(f[e] * q, -448.f), 448.f);
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[4], v[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(v[6], v[7], hi, true);
    return make_uint2((unsigned)lo, (unsigned)hi);
}

// FP8OUT: Y is bytes (e4m3) = sat(y * q_scale) instead of bf16 - the activation operand of the fp8 conv
template <bool FP8OUT>
__global__ __launch_bounds__(256) void gn_apply_kernel(const uint16_t* __restrict__ X, const uint16_t* __restrict__ X2,
                                                       int C1, int C2, int HW, int groups, int splits,
                                                       const float* __restrict__ partials,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, int silu, int pix_per_block,
                                                       uint16_t* __restrict__ Y, float q_scale, int* __restrict__ sat) {
    // `sat` (debug, sdv_groupnorm_fp8_set_saturation_counter): number of elements the e4m3 conversion had to clamp - a calibration
    // that is too tight for the prompts actually run shows up here instead of as silently clipped activations
    int nsat = 0;
    auto put = [&](long long pix, int c0, const float* f) {
        if constexpr (FP8OUT) {
            if (sat) {
#pragma unroll
                for (int e = 0; e < 8; ++e) nsat += fabsf(f[e] * q_scale) > 448.f ? 1 : 0;
            }
            *(uint2*)((uint8_t*)Y + pix * C1 + pix * C2 + c0) = pack8_fp8(f, q_scale);
        } else {
            *(bf16x8_raw*)(Y + pix * (C1 + C2) + c0) = pack8(f);
        }
    };
    __shared__ float gmean[64], grstd[64];
    const int C = C1 + C2;
    const GnGeom g = gn_geom(C);
    const int cpg = C / groups;
    const int img = blockIdx.y;
    const int tid = threadIdx.x;
    if (tid < groups) {
        float s = 0.f, q = 0.f;
        for (int sp = 0; sp < splits; ++sp) {
            const float* in = partials + (((long long)img * splits + sp) * groups + tid) * 2;
            s += in[0];
            q += in[1];
        }
        const float cnt = (float)HW * (float)cpg;
        const float mean = s / cnt;
        float var = q / cnt - mean * mean;
        var = var > 0.f ? var : 0.f;
        gmean[tid] = mean;
        grstd[tid] = rsqrtf(var + eps);
    }
    __syncthreads();
    const int p_begin = blockIdx.x * pix_per_block;
    const int p_end = p_begin + pix_per_block < HW ? p_begin + pix_per_block : HW;
    const int tc = tid % g.CT, tp = tid / g.CT;
    if (tp >= g.PT) return;
    for (int ch = tc; ch < g.nchunk; ch += g.CT) {
        const int c0 = ch << 3;
        float sc[8], sh[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            const int ge = c / cpg;
            const float a = grstd[ge] * gamma[c];
            sc[e] = a;
            sh[e] = beta[c] - gmean[ge] * a;
        }
        int p = p_begin + tp;
        // 4 independent 16-byte loads in flight per thread before the first use (as in the statistics pass)
        for (; p + 3 * g.PT < p_end; p += 4 * g.PT) {
            bf16x8_raw r[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                r[u] = *(const bf16x8_raw*)gn_src(X, X2, C1, C2, (long long)img * HW + p + u * g.PT, c0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                float f[8];
                unpack8(r[u], f);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    float v = f[e] * sc[e] + sh[e];
                    f[e] = silu ? silu_f(v) : v;
                }
                put((long long)img * HW + p + u * g.PT, c0, f);
            }
        }
        for (; p < p_end; p += g.PT) {
            const long long pix = (long long)img * HW + p;
            const bf16x8_raw r = *(const bf16x8_raw*)gn_src(X, X2, C1, C2, pix, c0);
            float f[8];
            unpack8(r, f);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = f[e] * sc[e] + sh[e];
                f[e] = silu ? silu_f(v) : v;
            }
            put(pix, c0, f);
        }
    }
    if constexpr (FP8OUT) {
        if (sat && nsat) atomicAdd(sat, nsat);   // (never taken on a calibrated model: no atomic traffic in the normal case)
    }
}

// One wave normalises RPW rows per pass (RPW = 1 is what ships: measured on MI355X, 4 rows per wave HALVES the achieved
// bandwidth - 3.8 -> 2.1 TB/s - because a quarter as many waves are in flight).  C <= 8 * 64 * MAXC.
template <int MAXC, int RPW>
__global__ __launch_bounds__(256) void layernorm_kernel(const uint16_t* __restrict__ X, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float eps, long long rows, int C,
                                                        uint16_t* __restrict__ Y) {
    const int lane = threadIdx.x & 63;
    const long long row0 = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW;
    if (row0 >= rows) return;
    const int nchunk = C >> 3;
    float gm[MAXC][8], bt[MAXC][8];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int ch = lane + i * 64;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            gm[i][e] = ch < nchunk ? gamma[ch * 8 + e] : 0.f;
            bt[i][e] = ch < nchunk ? beta[ch * 8 + e] : 0.f;
        }
    }
    bf16x8_raw raw[RPW][MAXC];
#pragma unroll
    for (int r = 0; r < RPW; ++r)
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = lane + i * 64;
            if (ch < nchunk && row0 + r < rows) raw[r][i] = *(const bf16x8_raw*)(X + (row0 + r) * C + ch * 8);
        }
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        if (row0 + r >= rows) break;
        float f[MAXC][8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = lane + i * 64;
            if (ch < nchunk) {
                unpack8(raw[r][i], f[i]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += f[i][e];
            }
        }
        const float mean = wave_sum(s) / (float)C;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = lane + i * 64;
            if (ch < nchunk) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float d = f[i][e] - mean;
                    q += d * d;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / (float)C + eps);
        uint16_t* y = Y + (row0 + r) * C;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int ch = lane + i * 64;
            if (ch < nchunk) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (f[i][e] - mean) * rstd * gm[i][e] + bt[i][e];
                *(bf16x8_raw*)(y + ch * 8) = pack8(o);
            }
        }
    }
}

// LayerNorm for the SD widths C = 320 * k (k = 1, 2, 4): LPR = 8k lanes share a row, 5 x 16-byte chunks per lane,
// 64 / LPR rows per wave - every lane is busy (the one-wave-per-row kernel idles 24 of 64 lanes at C = 320) and has five
// independent loads in flight; row statistics reduce over LPR lanes with log2(LPR) shuffles.
template <int LPR>
__global__ __launch_bounds__(256) void layernorm320_kernel(const uint16_t* __restrict__ X, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, long long rows,
                                                           uint16_t* __restrict__ Y) {
    constexpr int C = LPR * 40;
    constexpr int RPWV = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const int sub = lane % LPR;
    const long long row = ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * RPWV + lane / LPR;
    const bool live = row < rows;
    const uint16_t* x = X + (live ? row : 0) * C;
    bf16x8_raw raw[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) raw[i] = *(const bf16x8_raw*)(x + (sub + LPR * i) * 8);
    float f[5][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        unpack8(raw[i], f[i]);
#pragma unroll
        for (int e = 0; e < 8; ++e) s += f[i][e];
    }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float d = f[i][e] - mean;
            q += d * d;
        }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = rsqrtf(q / (float)C + eps);
    if (!live) return;
    uint16_t* y = Y + row * C;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int c0 = (sub + LPR * i) * 8;
        const float4 g0 = *(const float4*)(gamma + c0), g1 = *(const float4*)(gamma + c0 + 4);
        const float4 b0 = *(const float4*)(beta + c0), b1 = *(const float4*)(beta + c0 + 4);
        const float gg[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
        const float bb[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (f[i][e] - mean) * rstd * gg[e] + bb[e];
        *(bf16x8_raw*)(y + c0) = pack8(o);
    }
}

}  // namespace

// ---- GroupNorm statistics from the producing igemm's epilogue (sdv_gemm_args.gn_out) ------------------------------------------
// P [blocks][2][ld]: (sum, sumsq) per 32-row block and channel.  One workgroup per image: thread t sums its channels over the
// image's blocks (fixed order), one thread per group folds the group's channels (fixed order) -> partials[img][0][group][2],
// the layout sdv_groupnorm_apply reads with splits = 1.  Two sources = the channel concat of the up blocks (cat([x, skip])):
// groups may straddle the seam, which is why the producers emit per-CHANNEL sums.
namespace {
struct GnSrc {
    const float* P;
    int C, ld, bpi, nrep;      // channels, row stride of P, 32-row blocks per image (and repetition), repetitions (mode 4: 4 phases)
    long long rep_stride;      // blocks between two repetitions
};
__global__ __launch_bounds__(256) void gn_finalize_kernel(GnSrc a, GnSrc b, int groups, int splits, float* __restrict__ partials) {
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [C][2]
    const int C = a.C + b.C, cpg = C / groups;
    const int img = blockIdx.x, split = blockIdx.y;
    for (int c = threadIdx.x; c < C; c += 256) {
        const GnSrc& s = c < a.C ? a : b;
        const int cc = c < a.C ? c : c - a.C;
        const int per = (s.bpi + splits - 1) / splits;
        const int k0 = split * per, k1 = k0 + per < s.bpi ? k0 + per : s.bpi;
        float s1 = 0.f, s2 = 0.f;
        for (int r = 0; r < s.nrep; ++r) {
            const float* base = s.P + ((r * s.rep_stride + (long long)img * s.bpi) * 2) * s.ld + cc;
            int k = k0;
            for (; k + 3 < k1; k += 4) {               // 8 independent loads in flight; the adds keep the block order
                float t[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) t[j] = base[(long long)(2 * k + j) * s.ld];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s1 += t[2 * j];
                    s2 += t[2 * j + 1];
                }
            }
            for (; k < k1; ++k) {
                s1 += base[(long long)(2 * k) * s.ld];
                s2 += base[(long long)(2 * k + 1) * s.ld];
            }
        }
        sm[2 * c] = s1;
        sm[2 * c + 1] = s2;
    }
    __syncthreads();
    if (threadIdx.x < groups) {
        float as = 0.f, aq = 0.f;
        for (int c = 0; c < cpg; ++c) {
            as += sm[2 * (threadIdx.x * cpg + c)];
            aq += sm[2 * (threadIdx.x * cpg + c) + 1];
        }
        float* out = partials + (((long long)img * splits + split) * groups + threadIdx.x) * 2;
        out[0] = as;
        out[1] = aq;
    }
}
}  // namespace

extern "C" int sdv_groupnorm_finalize(const float* P1, int32_t C1, int32_t ld1, int32_t bpi1, int32_t nrep1, int64_t rep_stride1,
                                      const float* P2, int32_t C2, int32_t ld2, int32_t bpi2, int32_t nrep2, int64_t rep_stride2,
                                      int32_t nimg, int32_t groups, int32_t splits, float* partials, void* stream) {
    SDV_REQUIRE(P1 && partials && C1 > 0 && nimg > 0, "sdv_groupnorm_finalize: bad args");
    SDV_REQUIRE(C2 == 0 || P2, "sdv_groupnorm_finalize: C2 > 0 needs P2");
    const int C = C1 + C2;
    SDV_REQUIRE(groups > 0 && groups <= 64 && C % groups == 0, "sdv_groupnorm_finalize: bad groups %d for C=%d", groups, C);
    SDV_REQUIRE(bpi1 > 0 && nrep1 > 0 && (C2 == 0 || (bpi2 > 0 && nrep2 > 0)) && splits > 0 && splits <= 64, "sdv_groupnorm_finalize: bad block geometry");
    SDV_REQUIRE((size_t)C * 8 <= 64 * 1024, "sdv_groupnorm_finalize: C=%d too large", C);
    GnSrc a{P1, C1, ld1, bpi1, nrep1, rep_stride1};
    GnSrc b{C2 ? P2 : P1, C2, C2 ? ld2 : ld1, C2 ? bpi2 : 1, C2 ? nrep2 : 1, C2 ? rep_stride2 : 0};
    hipLaunchKernelGGL(gn_finalize_kernel, dim3(nimg, splits), dim3(256), (size_t)C * 8, (hipStream_t)stream, a, b, groups, splits, partials);
    SDV_CHECK_LAUNCH("sdv_groupnorm_finalize");
    return SDV_OK;
}

extern "C" int sdv_groupnorm_stats(const sdv_bf16* X, const sdv_bf16* X2, int32_t C1, int32_t C2, int32_t nimg,
                                   int32_t HW, int32_t groups, int32_t splits, float* partials, void* stream) {
    SDV_REQUIRE(X && partials, "sdv_groupnorm_stats: null pointer");
    SDV_REQUIRE(C2 == 0 || X2, "sdv_groupnorm_stats: C2 > 0 needs X2");
    const int C = C1 + C2;
    SDV_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0 && C > 0, "sdv_groupnorm_stats: channels must be multiples of 8");
    SDV_REQUIRE(groups > 0 && groups <= 64 && C % groups == 0, "sdv_groupnorm_stats: bad groups %d for C=%d", groups, C);
    SDV_REQUIRE(splits > 0 && nimg > 0 && HW > 0, "sdv_groupnorm_stats: bad sizes");
    const int nchunk = C / 8;
    const int CT = nchunk < 256 ? nchunk : 256;
    const size_t lds = (size_t)(256 / CT) * C * 2 * sizeof(float);
    SDV_REQUIRE(lds <= 64 * 1024, "sdv_groupnorm_stats: C=%d too large", C);
    hipLaunchKernelGGL(gn_stats_kernel, dim3(splits, nimg), dim3(256), lds, (hipStream_t)stream, X, X2 ? X2 : X, C1, C2, HW,
                       groups, splits, partials);
    SDV_CHECK_LAUNCH("sdv_groupnorm_stats");
    return SDV_OK;
}

// debug counter of clamped e4m3 conversions (one process drives one GPU; the pointer is baked into captured graphs like every other
// kernel argument, so it has to be set before a step is captured)
static int* g_fp8_sat_counter = nullptr;
extern "C" int sdv_groupnorm_fp8_set_saturation_counter(int32_t* counter) {
    g_fp8_sat_counter = counter;
    return SDV_OK;
}

static int gn_apply_launch(const sdv_bf16* X, const sdv_bf16* X2, int32_t C1, int32_t C2, int32_t nimg, int32_t HW,
                           int32_t groups, int32_t splits, const float* partials, const float* gamma, const float* beta,
                           float eps, int32_t silu, void* Y, bool fp8, float q_scale, void* stream) {
    SDV_REQUIRE(X && partials && gamma && beta && Y, "sdv_groupnorm_apply: null pointer");
    SDV_REQUIRE(C2 == 0 || X2, "sdv_groupnorm_apply: C2 > 0 needs X2");
    const int C = C1 + C2;
    SDV_REQUIRE(C1 % 8 == 0 && C2 % 8 == 0 && C > 0, "sdv_groupnorm_apply: channels must be multiples of 8");
    SDV_REQUIRE(groups > 0 && groups <= 64 && C % groups == 0, "sdv_groupnorm_apply: bad groups");
    // Elements per block: ~4k blocks in total, between 16 K and 128 K elements each.  Measured at 128 samples
    // (tools/gn_bench.py): 64^2 x 320 channels 3.4 TB/s with 16 K-element blocks (10 k blocks) vs 5.0 TB/s with 64 K; the
    // 32^2 / 16^2 levels are best at 16 K - 32 K.
    long long ppb_elems = (long long)nimg * HW * C / 4096;
    ppb_elems = ppb_elems < 16384 ? 16384 : (ppb_elems > 131072 ? 131072 : ppb_elems);
    int ppb = (int)((ppb_elems + C - 1) / C);
    if (ppb < 1) ppb = 1;
    const int nblk = (HW + ppb - 1) / ppb;
    if (fp8)
        hipLaunchKernelGGL(gn_apply_kernel<true>, dim3(nblk, nimg), dim3(256), 0, (hipStream_t)stream, X, X2 ? X2 : X, C1, C2, HW,
                           groups, splits, partials, gamma, beta, eps, silu, ppb, (uint16_t*)Y, q_scale, g_fp8_sat_counter);
    else
        hipLaunchKernelGGL(gn_apply_kernel<false>, dim3(nblk, nimg), dim3(256), 0, (hipStream_t)stream, X, X2 ? X2 : X, C1, C2, HW,
                           groups, splits, partials, gamma, beta, eps, silu, ppb, (uint16_t*)Y, 1.0f, (int*)nullptr);
    SDV_CHECK_LAUNCH("sdv_groupnorm_apply");
    return SDV_OK;
}

extern "C" int sdv_groupnorm_apply(const sdv_bf16* X, const sdv_bf16* X2, int32_t C1, int32_t C2, int32_t nimg, int32_t HW,
                                   int32_t groups, int32_t splits, const float* partials, const float* gamma,
                                   const float* beta, float eps, int32_t silu, sdv_bf16* Y, void* stream) {
    return gn_apply_launch(X, X2, C1, C2, nimg, HW, groups, splits, partials, gamma, beta, eps, silu, Y, false, 1.0f, stream);
}

extern "C" int sdv_groupnorm_apply_fp8(const sdv_bf16* X, const sdv_bf16* X2, int32_t C1, int32_t C2, int32_t nimg, int32_t HW,
                                       int32_t groups, int32_t splits, const float* partials, const float* gamma,
                                       const float* beta, float eps, int32_t silu, uint8_t* Y8, float q_scale, void* stream) {
    SDV_REQUIRE(q_scale > 0.f, "sdv_groupnorm_apply_fp8: q_scale must be positive");
    return gn_apply_launch(X, X2, C1, C2, nimg, HW, groups, splits, partials, gamma, beta, eps, silu, Y8, true, q_scale, stream);
}

extern "C" int sdv_layernorm_bf16(const sdv_bf16* X, const float* gamma, const float* beta, float eps, int64_t rows,
                                  int32_t C, sdv_bf16* Y, void* stream) {
    SDV_REQUIRE(X && gamma && beta && Y, "sdv_layernorm_bf16: null pointer");
    SDV_REQUIRE(C % 8 == 0 && C > 0 && C <= 8 * 64 * 4, "sdv_layernorm_bf16: C=%d unsupported (multiple of 8, <= 2048)", C);
    SDV_REQUIRE(rows > 0, "sdv_layernorm_bf16: rows");
    hipStream_t s = (hipStream_t)stream;
    const int nchunk = C / 8;
    const bool al16 = ((((uintptr_t)gamma | (uintptr_t)beta) & 15) == 0);
    if (al16 && (C == 320 || C == 640 || C == 1280)) {
        const int lpr = C / 40, rpb = 4 * (64 / lpr);
        const dim3 grid((unsigned)((rows + rpb - 1) / rpb));
        if (lpr == 8)
            hipLaunchKernelGGL(layernorm320_kernel<8>, grid, dim3(256), 0, s, X, gamma, beta, eps, (long long)rows, Y);
        else if (lpr == 16)
            hipLaunchKernelGGL(layernorm320_kernel<16>, grid, dim3(256), 0, s, X, gamma, beta, eps, (long long)rows, Y);
        else
            hipLaunchKernelGGL(layernorm320_kernel<32>, grid, dim3(256), 0, s, X, gamma, beta, eps, (long long)rows, Y);
    } else if (nchunk <= 64)
        hipLaunchKernelGGL((layernorm_kernel<1, 1>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, X, gamma, beta, eps,
                           (long long)rows, C, Y);
    else if (nchunk <= 128)
        hipLaunchKernelGGL((layernorm_kernel<2, 1>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, X, gamma, beta, eps,
                           (long long)rows, C, Y);
    else
        hipLaunchKernelGGL((layernorm_kernel<4, 1>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, X, gamma, beta, eps,
                           (long long)rows, C, Y);
    SDV_CHECK_LAUNCH("sdv_layernorm_bf16");
    return SDV_OK;
}


// ---- LayerNorm statistics from the producer GEMM's per-row partials (LayerNorm folded into the consumer GEMM) ---------
namespace {
__global__ __launch_bounds__(256) void rowstats_finalize_kernel(const float* __restrict__ part, long long rows, int slots, float inv_c,
                                                                float eps, float* __restrict__ out) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float2* pr = (const float2*)part + r * slots;
    float s1 = 0.f, s2 = 0.f;
    for (int i = 0; i < slots; ++i) {        // fixed order: deterministic
        const float2 v = pr[i];
        s1 += v.x;
        s2 += v.y;
    }
    const float mean = s1 * inv_c;
    const float var = fmaxf(s2 * inv_c - mean * mean, 0.f);
    ((float2*)out)[r] = make_float2(mean, rsqrtf(var + eps));
}
}  // namespace

extern "C" int sdv_rowstats_finalize(const float* partials, int64_t rows, int32_t slots, int32_t C, float eps, float* out,
                                     void* stream) {
    SDV_REQUIRE(partials && out && rows > 0 && slots > 0 && C > 0, "sdv_rowstats_finalize: bad args");
    hipLaunchKernelGGL(rowstats_finalize_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, partials,
                       (long long)rows, slots, 1.0f / C, eps, out);
    SDV_CHECK_LAUNCH("sdv_rowstats_finalize");
    return SDV_OK;
}
