// Wave-level elementwise / small kernels of the walk hot path: endpoint interpolation (lerp / slerp),
// the fused classifier-free-guidance + DDIM update, timestep embedding, the tiny fp32 linear used for
// the per-walk time-embedding tables, layout helpers, and the small-channel direct convolutions
// (UNet conv_in / conv_out, VAE conv_in / conv_out with the image epilogue).
#include <stdarg.h>

#include "sdv_common.h"

// ------------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";
void sdv_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* sdv_last_error(void) { return g_err; }
extern "C" int sdv_abi_version(void) { return 12; }

namespace {

constexpr int kThreads = 256;
inline unsigned grid_for(long long n, int per_block = kThreads, long long cap = 16384) {
    long long g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (unsigned)g;
}

// ---- slerp: whole-tensor reductions (utils.py:51) -----------------------------------------------
__global__ __launch_bounds__(1024) void slerp_stats_kernel(const float* __restrict__ v0, const float* __restrict__ v1,
                                                           long long n, double* __restrict__ stats) {
    __shared__ double red[3][16];
    double d = 0, a = 0, b = 0;
    for (long long i = threadIdx.x; i < n; i += 1024) {
        const double x = v0[i], y = v1[i];
        d += x * y;
        a += x * x;
        b += y * y;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        d += __shfl_xor(d, o);
        a += __shfl_xor(a, o);
        b += __shfl_xor(b, o);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        red[0][w] = d;
        red[1][w] = a;
        red[2][w] = b;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        double s = 0;
        for (int i = 0; i < 16; ++i) s += red[threadIdx.x][i];
        stats[threadIdx.x] = s;
    }
}

// out[f] = s0(f) * v0 + s1(f) * v1   (utils.py:52-61); grid.y = frame
__global__ __launch_bounds__(kThreads) void slerp_batch_kernel(const float* __restrict__ v0, const float* __restrict__ v1,
                                                               const double* __restrict__ stats,
                                                               const float* __restrict__ T, int C, int HW, int to_hwc,
                                                               float dot_threshold, float* __restrict__ out) {
    const int f = blockIdx.y;
    const float t = T[f];
    // the reductions and the two coefficients are evaluated in fp64 (once per block), then rounded to the
    // fp32 the reference computes in; sin(x)/sin(x) == 1 exactly, so slerp(0) == v0 and slerp(1) == v1 hold.
    const double dot = stats[0] / (sqrt(stats[1]) * sqrt(stats[2]));
    float s0, s1;
    if (fabs(dot) > (double)dot_threshold) {
        s0 = 1.0f - t;
        s1 = t;
    } else {
        const double theta0 = acos(dot);
        const double sin0 = sin(theta0);
        const double theta_t = theta0 * (double)t;
        s0 = (float)(sin(theta0 - theta_t) / sin0);
        s1 = (float)(sin(theta_t) / sin0);
    }
    const long long n = (long long)C * HW;
    float* o = out + (long long)f * n;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const float v = s0 * v0[i] + s1 * v1[i];
        if (to_hwc) {
            const int c = (int)(i / HW);
            const int p = (int)(i - (long long)c * HW);
            o[(long long)p * C + c] = v;
        } else {
            o[i] = v;
        }
    }
}

// torch.lerp: weight < 0.5 ? a + w (b - a) : b - (b - a)(1 - w)
__global__ __launch_bounds__(kThreads) void lerp_batch_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              const float* __restrict__ T, long long n,
                                                              float* __restrict__ out_f32, uint16_t* __restrict__ out_bf16) {
    const int f = blockIdx.y;
    const float w = T[f];
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const float x = a[i], y = b[i];
        const float d = y - x;
        const float v = w < 0.5f ? x + w * d : y - d * (1.0f - w);
        if (out_f32) out_f32[(long long)f * n + i] = v;
        if (out_bf16) out_bf16[(long long)f * n + i] = f32_to_bf16(v);
    }
}

// ---- CFG + DDIM ------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void cfg_ddim_kernel(const float* __restrict__ eps, float* __restrict__ latents,
                                                            uint16_t* __restrict__ x2, const float* __restrict__ coefs,
                                                            const int* __restrict__ step_ptr,
                                                            const float* __restrict__ noise, float guidance, int cfg,
                                                            long long n) {
    const int step = step_ptr ? *step_ptr : 0;
    const float cx = coefs[step * 4 + 0], ce = coefs[step * 4 + 1], sg = coefs[step * 4 + 2];
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        float e;
        if (cfg) {
            const float eu = eps[i], ec = eps[n + i];
            e = eu + guidance * (ec - eu);
        } else {
            e = eps[i];
        }
        float x = cx * latents[i] + ce * e;
        if (noise) x += sg * noise[(long long)step * n + i];
        latents[i] = x;
        const uint16_t xb = f32_to_bf16(x);
        x2[i] = xb;
        if (cfg) x2[n + i] = xb;
    }
}

// ---- CFG + linear multistep update (PNDM/PLMS, LMS, Euler, Euler-ancestral, DPM-Solver++ 2M) -----------------------------
// Every scheduler the reference accepts (stable_diffusion_pipeline.py:71-78) advances the latents by a LINEAR combination of
// the current sample, the current model output and up to three earlier ones; the host (scheduler.py) precomputes one row of
// coefficients per UNet evaluation (layout in sdv_hip.h) and this kernel is the whole of scheduler.step() + the next
// scale_model_input() + torch.cat([latents] * 2):
//     g    = eps_u + guidance (eps_c - eps_u)                      classifier-free guidance (:422-423)
//     m    = u x + v g                                            model output in the solver's variable (eps, x0 or dx/dsigma)
//     comb = w0 m + w1 H[-1] + w2 H[-2] + w3 H[-3]                H = ring of the m of earlier evaluations
//     x'   = a x_base + c comb (+ sn noise)                       x_base = x, or the sample saved at PLMS's first evaluation
//     x2   = bf16(s_in x')                                        the next UNet input, both CFG halves
__global__ __launch_bounds__(kThreads) void cfg_multistep_kernel(const float* __restrict__ eps, float* __restrict__ latents,
                                                                 uint16_t* __restrict__ x2, float* __restrict__ hist,
                                                                 float* __restrict__ xsave, const float* __restrict__ table,
                                                                 const int* __restrict__ step_ptr, const float* __restrict__ noise,
                                                                 float guidance, int cfg, long long n) {
    const int step = step_ptr ? *step_ptr : 0;
    const float* r = table + (long long)step * 16;
    const float a = r[0], c = r[1], w0 = r[2], w1 = r[3], w2 = r[4], w3 = r[5], u = r[6], v = r[7], s_in = r[8], sn = r[9];
    const int flags = (int)r[10], head = (int)r[11];
    const bool push = flags & 1, save = flags & 2, use_saved = flags & 4;
    const float* h1 = hist + (long long)((head + 3) & 3) * n;
    const float* h2 = hist + (long long)((head + 2) & 3) * n;
    const float* h3 = hist + (long long)((head + 1) & 3) * n;
    float* hp = hist + (long long)(head & 3) * n;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        float g;
        if (cfg) {
            const float eu = eps[i], ec = eps[n + i];
            g = eu + guidance * (ec - eu);
        } else {
            g = eps[i];
        }
        const float xc = latents[i];
        const float m = u * xc + v * g;
        float comb = w0 * m;
        if (w1 != 0.f) comb += w1 * h1[i];
        if (w2 != 0.f) comb += w2 * h2[i];
        if (w3 != 0.f) comb += w3 * h3[i];
        const float xb = use_saved ? xsave[i] : xc;
        if (save) xsave[i] = xc;
        float x = a * xb + c * comb;
        if (noise && sn != 0.f) x += sn * noise[(long long)step * n + i];
        latents[i] = x;
        if (push) hp[i] = m;
        const uint16_t xq = f32_to_bf16(x * s_in);
        x2[i] = xq;
        if (cfg) x2[n + i] = xq;
    }
}

__global__ __launch_bounds__(kThreads) void latents_to_input_kernel(const float* __restrict__ latents,
                                                                    uint16_t* __restrict__ x2, int cfg, long long n) {
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const uint16_t xb = f32_to_bf16(latents[i]);
        x2[i] = xb;
        if (cfg) x2[n + i] = xb;
    }
}

__global__ void step_add_kernel(int* p, int inc) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *p += inc;
}

// ---- timestep embedding ---------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void timestep_embedding_kernel(const float* __restrict__ ts, int n, int dim,
                                                                      int flip, float freq_shift,
                                                                      float* __restrict__ out) {
    const int half = dim / 2;
    const int i = blockIdx.x * kThreads + threadIdx.x;
    if (i >= n * half) return;
    const int r = i / half, k = i - r * half;
    // once-per-walk kernel: the frequency is evaluated in fp64, the product is rounded to fp32 exactly as
    // diffusers does (t.float() * exp(exponent).float()) and sin/cos are taken in fp64 of that fp32 argument
    const float freq = (float)exp(-9.210340371976184 * (double)k / ((double)half - (double)freq_shift));  // ln(10000)
    const float a = ts[r] * freq;
    const float s = (float)sin((double)a), c = (float)cos((double)a);
    float* o = out + (long long)r * dim;
    if (flip) {
        o[k] = c;
        o[half + k] = s;
    } else {
        o[k] = s;
        o[half + k] = c;
    }
}

// out[m][n] = sum_k act(x[m][k]) w[n][k] + b[n] + add[n]; one wave per output element row-pair
__global__ __launch_bounds__(kThreads) void linear_small_kernel(const float* __restrict__ x, const uint16_t* __restrict__ w,
                                                                const float* __restrict__ b, const float* __restrict__ add,
                                                                float* __restrict__ out, int M, int N, int K, int silu_in) {
    const int lane = threadIdx.x & 63;
    const long long o = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= (long long)M * N) return;
    const int m = (int)(o / N), n = (int)(o - (long long)m * N);
    const float* xr = x + (long long)m * K;
    const uint16_t* wr = w + (long long)n * K;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) {
        float xv = xr[k];
        if (silu_in) xv = silu_f(xv);
        acc += xv * bf16_to_f32(wr[k]);
    }
    acc = wave_sum(acc);
    if (lane == 0) out[o] = acc + (b ? b[n] : 0.f) + (add ? add[n] : 0.f);
}

// ---- layout helpers -------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void permute_f32_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                               int n, int C, int HW, int to_nhwc) {
    const long long tot = (long long)n * C * HW;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < tot; i += (long long)gridDim.x * kThreads) {
        // i indexes the OUTPUT
        const long long per = (long long)C * HW;
        const long long img = i / per, r = i - img * per;
        if (to_nhwc) {
            const int p = (int)(r / C), c = (int)(r - (long long)p * C);
            out[i] = in[img * per + (long long)c * HW + p];
        } else {
            const int c = (int)(r / HW), p = (int)(r - (long long)c * HW);
            out[i] = in[img * per + (long long)p * C + c];
        }
    }
}

__global__ __launch_bounds__(kThreads) void f32_to_bf16_kernel(const float* __restrict__ in, uint16_t* __restrict__ out,
                                                               long long n) {
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads)
        out[i] = f32_to_bf16(in[i]);
}

// CLIP text embeddings: out[t][:] = tok[ids[t]][:] + pos[t % L][:]  (fp32 tables, bf16 out); D % 4 == 0
__global__ __launch_bounds__(kThreads) void embed_tokens_kernel(const int64_t* __restrict__ ids, const float* __restrict__ tok,
                                                                const float* __restrict__ pos, uint16_t* __restrict__ out,
                                                                long long n_tokens, int L, int D, int vocab) {
    const int q = D >> 2;
    const long long n = n_tokens * q;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const long long t = i / q;
        const int c = (int)(i - t * q) * 4;
        long long id = ids[t];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);   // the binding rejects out-of-range ids; never read out of bounds
        const float4 a = *(const float4*)(tok + id * D + c);
        const float4 b = *(const float4*)(pos + (t % L) * D + c);
        uint2 o;
        o.x = pack_bf16x2(a.x + b.x, a.y + b.y);
        o.y = pack_bf16x2(a.z + b.z, a.w + b.w);
        *(uint2*)(out + t * D + c) = o;
    }
}

// uint8 RGB pixels -> 4-channel bf16 rows {r, g, b, 0} * scale  (RealESRGANer.pre_process: img / 255)
__global__ __launch_bounds__(kThreads) void rgb_u8_to_bf16_c4_kernel(const uint8_t* __restrict__ in, uint16_t* __restrict__ out,
                                                                     long long npix, float scale) {
    for (long long p = (long long)blockIdx.x * kThreads + threadIdx.x; p < npix; p += (long long)gridDim.x * kThreads) {
        const uint8_t* px = in + p * 3;
        uint2 o;
        o.x = pack_bf16x2(px[0] * scale, px[1] * scale);
        o.y = pack_bf16x2(px[2] * scale, 0.f);
        *(uint2*)(out + p * 4) = o;
    }
}

// out[r][c] = alpha * a[r][c] + beta * b[r][c] over strided bf16 rows (the 0.2-scaled residual sums of RRDBNet)
__global__ __launch_bounds__(kThreads) void axpby_bf16_kernel(const uint16_t* __restrict__ a, int lda, const uint16_t* __restrict__ b,
                                                              int ldb, uint16_t* __restrict__ out, int ldo, long long rows,
                                                              int cols, float alpha, float beta) {
    const int cpr = cols >> 3;
    const long long n = rows * cpr;
    for (long long i = (long long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long long)gridDim.x * kThreads) {
        const long long r = i / cpr;
        const int c = (int)(i - r * cpr) * 8;
        float fa[8], fb[8];
        unpack8(*(const bf16x8_raw*)(a + r * lda + c), fa);
        unpack8(*(const bf16x8_raw*)(b + r * ldb + c), fb);
#pragma unroll
        for (int e = 0; e < 8; ++e) fa[e] = alpha * fa[e] + beta * fb[e];
        *(bf16x8_raw*)(out + r * ldo + c) = pack8(fa);
    }
}

// z[p][o] = sum_c Wpq[o][c] * (x[p][c] * in_scale) + b[o]      (C <= 8)
__global__ __launch_bounds__(kThreads) void latent_affine_kernel(const float* __restrict__ X, const float* __restrict__ Wpq,
                                                                 const float* __restrict__ bias, float in_scale,
                                                                 uint16_t* __restrict__ Y, long long npix, int C) {
    for (long long p = (long long)blockIdx.x * kThreads + threadIdx.x; p < npix; p += (long long)gridDim.x * kThreads) {
        float xin[8];
        for (int c = 0; c < C; ++c) xin[c] = X[p * C + c] * in_scale;
        for (int o = 0; o < C; ++o) {
            float a = bias ? bias[o] : 0.f;
            for (int c = 0; c < C; ++c) a += Wpq[o * C + c] * xin[c];
            Y[p * C + o] = f32_to_bf16(a);
        }
    }
}


// ---- im2col for the 4-channel latent convs (UNet conv_in, VAE decoder.conv_in): every output pixel gets one 128-byte
//      row [tap0 c0..c3 | tap1 c0..c3 | ... | tap8 | zeros up to 64] so that the conv becomes a K = 64 dense GEMM on
//      the matrix cores (weights zero-padded the same way).  HBM-bound: 8-byte gathers in, 16-byte stores out. ----
__global__ __launch_bounds__(kThreads) void im2col3x3_c4_kernel(const uint16_t* __restrict__ X, uint16_t* __restrict__ Y,
                                                                int nimg, int H, int Wd, int circular) {
    const long long npix = (long long)nimg * H * Wd;
    for (long long pix = (long long)blockIdx.x * kThreads + threadIdx.x; pix < npix; pix += (long long)gridDim.x * kThreads) {
        const int img = (int)(pix / ((long long)H * Wd));
        const int rem = (int)(pix - (long long)img * H * Wd);
        const int y = rem / Wd, x = rem - y * Wd;
        uint2 v[10];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
            bool ok = true;
            if (circular) {
                iy = (iy + H) % H;
                ix = (ix + Wd) % Wd;
            } else {
                ok = iy >= 0 && iy < H && ix >= 0 && ix < Wd;
            }
            v[tap] = ok ? *(const uint2*)(X + (((long long)img * H + iy) * Wd + ix) * 4) : make_uint2(0u, 0u);
        }
        v[9] = make_uint2(0u, 0u);
        uint4* out = (uint4*)(Y + pix * 64);
#pragma unroll
        for (int i = 0; i < 5; ++i) out[i] = make_uint4(v[2 * i].x, v[2 * i].y, v[2 * i + 1].x, v[2 * i + 1].y);
#pragma unroll
        for (int i = 5; i < 8; ++i) out[i] = make_uint4(0u, 0u, 0u, 0u);
    }
}

// ---- direct conv3x3, tiny Cin (4 -> 320 / 512): thread = (pixel, 8 output channels) ------------
// weights [Cout][9][Cin] bf16 are transposed into LDS as fp32 [9*Cin][Cout] once per block.
template <int CIN>
__global__ __launch_bounds__(kThreads) void conv3x3_cin_small_kernel(const uint16_t* __restrict__ X,
                                                                     const uint16_t* __restrict__ Wt,
                                                                     const float* __restrict__ bias,
                                                                     uint16_t* __restrict__ Y, int nimg, int H, int Wd,
                                                                     int Cout, int circular) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* wl = (float*)smem_raw;  // [9*CIN][Cout]
    constexpr int KK = 9 * CIN;
    for (int i = threadIdx.x; i < KK * Cout; i += kThreads) {   // contiguous (conflict-free) LDS writes
        const int k = i / Cout, co = i - k * Cout;
        wl[i] = bf16_to_f32(Wt[co * KK + k]);
    }
    __syncthreads();
    const int cgroups = Cout >> 3;                  // 8-channel groups per pixel
    const long long npix = (long long)nimg * H * Wd;
    const long long total = npix * cgroups;
    for (long long idx = (long long)blockIdx.x * kThreads + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * kThreads) {
        const long long pix = idx / cgroups;
        const int cg = (int)(idx - pix * cgroups);
        const int img = (int)(pix / ((long long)H * Wd));
        const int rem = (int)(pix - (long long)img * H * Wd);
        const int y = rem / Wd, x = rem - y * Wd;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = bias ? bias[cg * 8 + e] : 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
            bool ok = true;
            if (circular) {
                iy = (iy + H) % H;
                ix = (ix + Wd) % Wd;
            } else {
                ok = iy >= 0 && iy < H && ix >= 0 && ix < Wd;
            }
            if (!ok) continue;
            const uint16_t* xp = X + (((long long)img * H + iy) * Wd + ix) * CIN;
#pragma unroll
            for (int c = 0; c < CIN; ++c) {
                const float xv = bf16_to_f32(xp[c]);
                const float* wrow = wl + (tap * CIN + c) * Cout + cg * 8;
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] += xv * wrow[e];
            }
        }
        *(bf16x8_raw*)(Y + pix * Cout + cg * 8) = pack8(acc);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" int sdv_slerp_stats(const float* v0, const float* v1, int64_t n, double* stats, void* stream) {
    SDV_REQUIRE(v0 && v1 && stats && n > 0, "sdv_slerp_stats: bad args");
    hipLaunchKernelGGL(slerp_stats_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, v0, v1, (long long)n, stats);
    SDV_CHECK_LAUNCH("sdv_slerp_stats");
    return SDV_OK;
}

extern "C" int sdv_slerp_batch(const float* v0, const float* v1, const double* stats, const float* T, int32_t nframes,
                               int32_t C, int32_t HW, int32_t to_hwc, float dot_threshold, float* out, void* stream) {
    SDV_REQUIRE(v0 && v1 && stats && T && out, "sdv_slerp_batch: null pointer");
    SDV_REQUIRE(nframes > 0 && C > 0 && HW > 0, "sdv_slerp_batch: bad sizes");
    const long long n = (long long)C * HW;
    hipLaunchKernelGGL(slerp_batch_kernel, dim3(grid_for(n, kThreads, 256), nframes), dim3(kThreads), 0,
                       (hipStream_t)stream, v0, v1, stats, T, C, HW, to_hwc, dot_threshold, out);
    SDV_CHECK_LAUNCH("sdv_slerp_batch");
    return SDV_OK;
}

extern "C" int sdv_lerp_batch(const float* a, const float* b, const float* T, int32_t nframes, int64_t n, float* out_f32,
                              sdv_bf16* out_bf16, void* stream) {
    SDV_REQUIRE(a && b && T && (out_f32 || out_bf16), "sdv_lerp_batch: null pointer");
    SDV_REQUIRE(nframes > 0 && n > 0, "sdv_lerp_batch: bad sizes");
    hipLaunchKernelGGL(lerp_batch_kernel, dim3(grid_for(n, kThreads, 256), nframes), dim3(kThreads), 0,
                       (hipStream_t)stream, a, b, T, (long long)n, out_f32, out_bf16);
    SDV_CHECK_LAUNCH("sdv_lerp_batch");
    return SDV_OK;
}

extern "C" int sdv_cfg_ddim_step(const float* eps, float* latents, sdv_bf16* x2, const float* coefs,
                                 const int32_t* step_ptr, const float* noise, float guidance, int32_t cfg,
                                 int64_t n_per_batch, void* stream) {
    SDV_REQUIRE(eps && latents && x2 && coefs && n_per_batch > 0, "sdv_cfg_ddim_step: bad args");
    hipLaunchKernelGGL(cfg_ddim_kernel, dim3(grid_for(n_per_batch, kThreads, 2048)), dim3(kThreads), 0,
                       (hipStream_t)stream, eps, latents, x2, coefs, step_ptr, noise, guidance, cfg, (long long)n_per_batch);
    SDV_CHECK_LAUNCH("sdv_cfg_ddim_step");
    return SDV_OK;
}

extern "C" int sdv_cfg_multistep_step(const float* eps, float* latents, sdv_bf16* x2, float* hist, float* xsave,
                                      const float* table, const int32_t* step_ptr, const float* noise, float guidance,
                                      int32_t cfg, int64_t n_per_batch, void* stream) {
    SDV_REQUIRE(eps && latents && x2 && hist && xsave && table && n_per_batch > 0, "sdv_cfg_multistep_step: bad args");
    hipLaunchKernelGGL(cfg_multistep_kernel, dim3(grid_for(n_per_batch, kThreads, 2048)), dim3(kThreads), 0, (hipStream_t)stream,
                       eps, latents, x2, hist, xsave, table, step_ptr, noise, guidance, cfg, (long long)n_per_batch);
    SDV_CHECK_LAUNCH("sdv_cfg_multistep_step");
    return SDV_OK;
}

extern "C" int sdv_latents_to_unet_input(const float* latents, sdv_bf16* x2, int32_t cfg, int64_t n, void* stream) {
    SDV_REQUIRE(latents && x2 && n > 0, "sdv_latents_to_unet_input: bad args");
    hipLaunchKernelGGL(latents_to_input_kernel, dim3(grid_for(n, kThreads, 2048)), dim3(kThreads), 0, (hipStream_t)stream,
                       latents, x2, cfg, (long long)n);
    SDV_CHECK_LAUNCH("sdv_latents_to_unet_input");
    return SDV_OK;
}

extern "C" int sdv_step_counter_add(int32_t* step_ptr, int32_t inc, void* stream) {
    SDV_REQUIRE(step_ptr, "sdv_step_counter_add: null pointer");
    hipLaunchKernelGGL(step_add_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, step_ptr, inc);
    SDV_CHECK_LAUNCH("sdv_step_counter_add");
    return SDV_OK;
}

extern "C" int sdv_timestep_embedding(const float* timesteps, int32_t n, int32_t dim, int32_t flip_sin_to_cos,
                                      float freq_shift, float* out, void* stream) {
    SDV_REQUIRE(timesteps && out && n > 0 && dim > 0 && dim % 2 == 0, "sdv_timestep_embedding: bad args");
    hipLaunchKernelGGL(timestep_embedding_kernel, dim3(grid_for((long long)n * dim / 2)), dim3(kThreads), 0,
                       (hipStream_t)stream, timesteps, n, dim, flip_sin_to_cos, freq_shift, out);
    SDV_CHECK_LAUNCH("sdv_timestep_embedding");
    return SDV_OK;
}

extern "C" int sdv_linear_small(const float* x, const sdv_bf16* w, const float* b, const float* add, float* out, int32_t M,
                                int32_t N, int32_t K, int32_t silu_in, void* stream) {
    SDV_REQUIRE(x && w && out && M > 0 && N > 0 && K > 0, "sdv_linear_small: bad args");
    const long long outs = (long long)M * N;
    hipLaunchKernelGGL(linear_small_kernel, dim3((unsigned)((outs + 3) / 4)), dim3(kThreads), 0, (hipStream_t)stream, x, w,
                       b, add, out, M, N, K, silu_in);
    SDV_CHECK_LAUNCH("sdv_linear_small");
    return SDV_OK;
}

extern "C" int sdv_nchw_to_nhwc_f32(const float* in, float* out, int32_t n, int32_t C, int32_t HW, void* stream) {
    SDV_REQUIRE(in && out && n > 0 && C > 0 && HW > 0, "sdv_nchw_to_nhwc_f32: bad args");
    hipLaunchKernelGGL(permute_f32_kernel, dim3(grid_for((long long)n * C * HW)), dim3(kThreads), 0, (hipStream_t)stream,
                       in, out, n, C, HW, 1);
    SDV_CHECK_LAUNCH("sdv_nchw_to_nhwc_f32");
    return SDV_OK;
}

extern "C" int sdv_nhwc_to_nchw_f32(const float* in, float* out, int32_t n, int32_t C, int32_t HW, void* stream) {
    SDV_REQUIRE(in && out && n > 0 && C > 0 && HW > 0, "sdv_nhwc_to_nchw_f32: bad args");
    hipLaunchKernelGGL(permute_f32_kernel, dim3(grid_for((long long)n * C * HW)), dim3(kThreads), 0, (hipStream_t)stream,
                       in, out, n, C, HW, 0);
    SDV_CHECK_LAUNCH("sdv_nhwc_to_nchw_f32");
    return SDV_OK;
}

extern "C" int sdv_f32_to_bf16(const float* in, sdv_bf16* out, int64_t n, void* stream) {
    SDV_REQUIRE(in && out && n > 0, "sdv_f32_to_bf16: bad args");
    hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, in, out, (long long)n);
    SDV_CHECK_LAUNCH("sdv_f32_to_bf16");
    return SDV_OK;
}

extern "C" int sdv_embed_tokens(const int64_t* ids, const float* tok, const float* pos, sdv_bf16* out, int64_t n_tokens,
                                int32_t L, int32_t D, int32_t vocab, void* stream) {
    SDV_REQUIRE(ids && tok && pos && out && n_tokens > 0 && L > 0 && vocab > 0, "sdv_embed_tokens: bad args");
    SDV_REQUIRE(D > 0 && D % 4 == 0, "sdv_embed_tokens: D=%d must be a multiple of 4", D);
    hipLaunchKernelGGL(embed_tokens_kernel, dim3(grid_for(n_tokens * (D / 4))), dim3(kThreads), 0, (hipStream_t)stream, ids, tok,
                       pos, out, (long long)n_tokens, L, D, vocab);
    SDV_CHECK_LAUNCH("sdv_embed_tokens");
    return SDV_OK;
}

extern "C" int sdv_rgb_u8_to_bf16_c4(const uint8_t* in, sdv_bf16* out, int64_t npix, float scale, void* stream) {
    SDV_REQUIRE(in && out && npix > 0, "sdv_rgb_u8_to_bf16_c4: bad args");
    hipLaunchKernelGGL(rgb_u8_to_bf16_c4_kernel, dim3(grid_for(npix)), dim3(kThreads), 0, (hipStream_t)stream, in, out,
                       (long long)npix, scale);
    SDV_CHECK_LAUNCH("sdv_rgb_u8_to_bf16_c4");
    return SDV_OK;
}

extern "C" int sdv_axpby_bf16(const sdv_bf16* a, int32_t lda, const sdv_bf16* b, int32_t ldb, sdv_bf16* out, int32_t ldo,
                              int64_t rows, int32_t cols, float alpha, float beta, void* stream) {
    SDV_REQUIRE(a && b && out && rows > 0 && cols > 0, "sdv_axpby_bf16: bad args");
    SDV_REQUIRE(cols % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldo % 8 == 0, "sdv_axpby_bf16: cols / strides must be multiples of 8");
    SDV_REQUIRE((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0, "sdv_axpby_bf16: pointers must be 16-byte aligned");
    hipLaunchKernelGGL(axpby_bf16_kernel, dim3(grid_for(rows * (cols / 8))), dim3(kThreads), 0, (hipStream_t)stream, a, lda, b, ldb,
                       out, ldo, (long long)rows, cols, alpha, beta);
    SDV_CHECK_LAUNCH("sdv_axpby_bf16");
    return SDV_OK;
}

extern "C" int sdv_latent_affine(const float* X, const float* Wpq, const float* bias, float in_scale, sdv_bf16* Y,
                                 int64_t npix, int32_t C, void* stream) {
    SDV_REQUIRE(X && Wpq && Y && npix > 0 && C > 0 && C <= 8, "sdv_latent_affine: bad args (C <= 8)");
    hipLaunchKernelGGL(latent_affine_kernel, dim3(grid_for(npix)), dim3(kThreads), 0, (hipStream_t)stream, X, Wpq, bias,
                       in_scale, Y, (long long)npix, C);
    SDV_CHECK_LAUNCH("sdv_latent_affine");
    return SDV_OK;
}

extern "C" int sdv_im2col3x3_c4(const sdv_bf16* X, sdv_bf16* Y, int32_t nimg, int32_t H, int32_t Wd, int32_t circular,
                                void* stream) {
    SDV_REQUIRE(X && Y && nimg > 0 && H > 0 && Wd > 0, "sdv_im2col3x3_c4: bad args");
    const long long npix = (long long)nimg * H * Wd;
    hipLaunchKernelGGL(im2col3x3_c4_kernel, dim3(grid_for(npix, kThreads, 8192)), dim3(kThreads), 0, (hipStream_t)stream, X, Y,
                       nimg, H, Wd, circular);
    SDV_CHECK_LAUNCH("sdv_im2col3x3_c4");
    return SDV_OK;
}

extern "C" int sdv_conv3x3_cin_small(const sdv_bf16* X, const sdv_bf16* W, const float* bias, sdv_bf16* Y, int32_t nimg,
                                     int32_t H, int32_t Wd, int32_t Cin, int32_t Cout, int32_t circular, void* stream) {
    SDV_REQUIRE(X && W && Y, "sdv_conv3x3_cin_small: null pointer");
    SDV_REQUIRE(Cin == 4 || Cin == 8, "sdv_conv3x3_cin_small: Cin must be 4 or 8 (got %d)", Cin);
    SDV_REQUIRE(Cout % 8 == 0 && Cout > 0, "sdv_conv3x3_cin_small: Cout must be a multiple of 8");
    const size_t lds = (size_t)9 * Cin * Cout * sizeof(float);
    SDV_REQUIRE(lds <= 160 * 1024, "sdv_conv3x3_cin_small: weights do not fit LDS");
    const long long total = (long long)nimg * H * Wd * (Cout / 8);
    const unsigned grid = grid_for(total, kThreads, 256 * 3);   // ~3 resident workgroups per CU, weights staged once each
    hipStream_t s = (hipStream_t)stream;
    if (Cin == 4) {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)conv3x3_cin_small_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr = true;
        }
        hipLaunchKernelGGL(conv3x3_cin_small_kernel<4>, dim3(grid), dim3(kThreads), lds, s, X, W, bias, Y, nimg, H, Wd, Cout,
                           circular);
    } else {
        static bool attr = false;
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)conv3x3_cin_small_kernel<8>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            attr = true;
        }
        hipLaunchKernelGGL(conv3x3_cin_small_kernel<8>, dim3(grid), dim3(kThreads), lds, s, X, W, bias, Y, nimg, H, Wd, Cout,
                           circular);
    }
    SDV_CHECK_LAUNCH("sdv_conv3x3_cin_small");
    return SDV_OK;
}

