// Implicit-GEMM on the gfx950 bf16 matrix cores: dense GEMM, conv3x3 (stride 1 / stride 2 /
// nearest-2x-upsample-then-conv) over NHWC activations, with the skip-connection concat, bias /
// time-embedding bias table, residual add, SiLU and GEGLU fused into the kernel.
//
// Replaces the ATen->cuDNN/cuBLAS work dispatched by
//   unet(x, t, encoder_hidden_states=ctx)      /root/reference/.../stable_diffusion_pipeline.py:418
//   vae.decode(latents)                        /root/reference/.../stable_diffusion_pipeline.py:433
//
// Design (CDNA4):
//   * C[m][n] = sum_k X[m][k] W[n][k].  Both operands are K-contiguous rows, so both are staged the
//     same way: 64-wide K tiles (128 B per row) go HBM -> LDS with global_load_lds_dwordx4 (LDS-DMA,
//     no VGPR round trip), double buffered, ONE barrier per K tile, next tile in flight during MFMA.
//   * LDS image is lane-linear (DMA constraint), so the bank-conflict swizzle is applied on the
//     per-lane SOURCE address and again on the ds_read_b128: physical 16-B chunk = logical chunk ^
//     ((row >> 1) & 7).  With 128-B rows two rows share a 256-B bank row, and this spreads any 16
//     rows that are distinct mod 16 over all 16 slots -> conflict-free ds_read_b128 fragments.
//   * v_mfma_f32_32x32x16_bf16 with the WEIGHT rows as the A operand and the ACTIVATION rows as the
//     B operand: lane l then owns output row m = l & 31 and, per accumulator quad, 4 CONSECUTIVE
//     output channels n -> 8-byte bf16 stores / residual loads and per-register bias.
//   * conv3x3: the K loop walks (tap, channel-tile); the per-lane source address is the shifted /
//     strided / upsampled pixel, and padding pixels are redirected to a 256-B zero page in HBM.
#include <type_traits>

#include "sdv_common.h"

namespace {

constexpr int kBK = 64;  // K tile (bf16 elements) = 128 bytes per row

template <int WM, int WN, int TM, int TN, bool CONV>
__global__ __launch_bounds__(WM * WN * 64) void igemm_kernel(const sdv_gemm_args p) {
    constexpr int NWV = WM * WN;  // waves per workgroup (4 or 8)
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int ROWS = BM + BN;
    constexpr int TILE_BYTES = ROWS * 128;
    constexpr int NX = BM / (8 * NWV);  // X rows staged per lane per K tile
    constexpr int NW = BN / (8 * NWV);  // W rows staged per lane per K tile
    static_assert(BM % (8 * NWV) == 0 && BN % (8 * NWV) == 0, "tile rows must split evenly over the waves");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;

    // XCD-aware tile order: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs; remap
    // (bijectively) so that each XCD - each private L2 - works on one contiguous run of tiles: neighbouring
    // M tiles of a conv share their halo rows, and all tiles of a run share the same W panel.
    const int nblk = gridDim.x;
    const int xq = nblk >> 3, xr = nblk & 7;
    const int xcd = blockIdx.x & 7, xi = blockIdx.x >> 3;
    const int tile_id = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
    const int tiles_m = (p.M + BM - 1) / BM;
    const int bm = tile_id % tiles_m;
    const int bn = tile_id / tiles_m;
    const int m0 = bm * BM;
    const int n0 = bn * BN;
    const long long bz = blockIdx.z;

    const int K = p.K;

    // ---- operand addressing: buffer descriptors + 32-bit lane offsets + scalar K offset -----------------
    // Every operand row is fetched with `buffer_load_dwordx4 ... offen lds` (LDS-DMA).  The descriptor base is
    // anchored at this workgroup's first row, so a 31-bit lane offset always reaches the window it touches;
    // the K position travels in the scalar offset (no per-tile VALU address math at all), and padding pixels /
    // rows beyond M or N use an offset past num_records: the hardware range check returns zeros for them.
    constexpr unsigned kOOB = 0x80000000u;
    constexpr int kRecords = 0x7ffffff0;
    const int rg = lane >> 3;  // row inside the 8-row group one wave-instruction moves
    const int pc = lane & 7;   // physical 16-B chunk inside the 128-B LDS row
    long long xbase1, xbase2;  // element offsets of the window start in X / X2 (wave-uniform)
    int pix0 = 0;
    if constexpr (CONV) {
        pix0 = (m0 / (p.Hout * p.Wout)) * p.Hin * p.Win;
        xbase1 = (long long)pix0 * p.ldx;
        xbase2 = (long long)pix0 * p.ldx2;
    } else {
        xbase1 = bz * p.sX + (long long)m0 * p.ldx;
        xbase2 = (long long)m0 * p.ldx2;
    }
    const __amdgpu_buffer_rsrc_t rs_x1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.X + xbase1), 0, kRecords, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.X2 + xbase2), 0, kRecords, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + bz * p.sW + (long long)n0 * p.ldw), 0, kRecords, 0x00020000);

    int xr_[NX];               // dense: row inside the tile (or -1 beyond M); conv: pixel index of the image origin - pix0
    int xay[NX], xax[NX];      // conv: anchor coordinates (oy*stride, ox*stride) or (oy, ox) for upsample
    int xlc[NX];               // byte offset of this lane's logical chunk inside a 128-B K tile
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int r = (wave + NWV * i) * 8 + rg;
        const int m = m0 + r;
        xlc[i] = (pc ^ ((r >> 1) & 7)) * 16;
        if constexpr (CONV) {
            const int hw = p.Hout * p.Wout;
            const int mc = m < p.M ? m : p.M - 1;
            const int img = mc / hw;
            const int rem = mc - img * hw;
            const int oy = rem / p.Wout;
            const int ox = rem - oy * p.Wout;
            const int st = p.mode == 2 ? 2 : 1;
            xr_[i] = m < p.M ? img * p.Hin * p.Win - pix0 : -1;
            xay[i] = oy * st;
            xax[i] = ox * st;
        } else {
            xr_[i] = m < p.M ? r : -1;
            xay[i] = 0;
            xax[i] = 0;
        }
    }
    unsigned wvo[NW];  // lane byte offsets of the W rows relative to rs_w
#pragma unroll
    for (int i = 0; i < NW; ++i) {
        const int r = (wave + NWV * (NX + i)) * 8 + rg;  // tile row (>= BM)
        const int rw = r - BM;
        wvo[i] = (n0 + rw < p.N) ? (unsigned)(rw * p.ldw * 2 + (pc ^ ((r >> 1) & 7)) * 16) : kOOB;
    }

    // The K loop walks segments = (tap, source) pairs; inside a segment only the scalar offset advances.
    unsigned xvo[NX];
    int tap = 0, srcsel = 0, seg_left = 0;
    int kx = 0;   // scalar byte offset inside the current X source
    int kwb = 0;  // scalar byte offset along the W rows (all taps and sources are contiguous in K)
    const bool two_src = p.C1 < K;
    const int up_shift = p.mode == 3 ? 1 : 0;
    const int ext_y = p.mode == 3 ? p.Hout : p.Hin;  // extent the tap offset is applied in
    const int ext_x = p.mode == 3 ? p.Wout : p.Win;

    auto new_segment = [&]() {
        const int ld2 = 2 * (srcsel ? p.ldx2 : p.ldx);
        if constexpr (CONV) {
            const int dy = tap / 3 - 1;
            const int dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
            for (int i = 0; i < NX; ++i) {
                int vy = xay[i] + dy, vx = xax[i] + dx;
                if (p.circular) {
                    vy = vy < 0 ? vy + ext_y : (vy >= ext_y ? vy - ext_y : vy);
                    vx = vx < 0 ? vx + ext_x : (vx >= ext_x ? vx - ext_x : vx);
                }
                const int iy = vy >> up_shift, ix = vx >> up_shift;
                const bool ok = ((unsigned)iy < (unsigned)p.Hin) & ((unsigned)ix < (unsigned)p.Win) & (xr_[i] >= 0);
                const int pix = xr_[i] + iy * p.Win + ix;
                xvo[i] = ok ? (unsigned)(pix * ld2 + xlc[i]) : kOOB;
            }
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i) xvo[i] = xr_[i] >= 0 ? (unsigned)(xr_[i] * ld2 + xlc[i]) : kOOB;
        }
        kx = 0;
        seg_left = (srcsel ? K - p.C1 : p.C1) / kBK;
    };

    auto stage = [&](int buf) {
        char* base = smem + buf * TILE_BYTES;
        if (seg_left == 0) new_segment();
        const __amdgpu_buffer_rsrc_t rs_x = srcsel ? rs_x2 : rs_x1;
#pragma unroll
        for (int i = 0; i < NX; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(base + (wave + NWV * i) * 1024),
                                                     16, (int)xvo[i], kx, 0, 0);
#pragma unroll
        for (int i = 0; i < NW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(base + (wave + NWV * (NX + i)) * 1024),
                                                     16, (int)wvo[i], kwb, 0, 0);
        kx += 128;
        kwb += 128;
        if (--seg_left == 0) {
            if (two_src && srcsel == 0) {
                srcsel = 1;
            } else {
                srcsel = 0;
                ++tap;
            }
        }
    };

    f32x16_t acc[TN][TM];
#pragma unroll
    for (int a = 0; a < TN; ++a)
#pragma unroll
        for (int b = 0; b < TM; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    const int l31 = lane & 31;
    const int lhi = lane >> 5;

    // LDS byte offsets of this lane's fragment rows (swizzle term folded per k-step below)
    int xrow_off[TM], xrow_sw[TM], wrow_off[TN], wrow_sw[TN];
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
        const int r = wm * TM * 32 + mt * 32 + l31;
        xrow_off[mt] = r * 128;
        xrow_sw[mt] = (r >> 1) & 7;
    }
#pragma unroll
    for (int nt = 0; nt < TN; ++nt) {
        const int r = BM + wn * TN * 32 + nt * 32 + l31;
        wrow_off[nt] = r * 128;
        wrow_sw[nt] = (r >> 1) & 7;
    }

    // Software-pipelined K tile: the fragments of k-step ks+1 are read into a second register set while the
    // MFMAs of k-step ks issue (1 ds_read_b128 slotted behind each MFMA), so the matrix pipe does not wait for
    // LDS latency inside the tile.
    auto compute = [&](int buf) {
        const char* base = smem + buf * TILE_BYTES;
        bf16x8_t xf[2][TM], wf[2][TN];
        auto load_frags = [&](int ks, bf16x8_t* xd, bf16x8_t* wd) {
            const int lc = ks * 2 + lhi;
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) xd[mt] = *(const bf16x8_t*)(base + xrow_off[mt] + ((lc ^ xrow_sw[mt]) << 4));
#pragma unroll
            for (int nt = 0; nt < TN; ++nt) wd[nt] = *(const bf16x8_t*)(base + wrow_off[nt] + ((lc ^ wrow_sw[nt]) << 4));
        };
        load_frags(0, xf[0], wf[0]);
#pragma unroll
        for (int ks = 0; ks < kBK / 16; ++ks) {
            if (ks + 1 < kBK / 16) load_frags(ks + 1, xf[(ks + 1) & 1], wf[(ks + 1) & 1]);
#pragma unroll
            for (int nt = 0; nt < TN; ++nt)
#pragma unroll
                for (int mt = 0; mt < TM; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ks & 1][nt], xf[ks & 1][mt], acc[nt][mt], 0, 0, 0);
            if (ks + 1 < kBK / 16) {
#pragma unroll
                for (int i = 0; i < TM + TN; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // 1 MFMA
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // 1 DS read
                }
            }
        }
    };

    // ---- main loop: one barrier per K tile, tile t+1 in flight (LDS-DMA) while tile t computes ----
    const int ntaps = CONV ? 9 : 1;
    const int nkt = (K / kBK) * ntaps;
    stage(0);
    for (int kt = 0; kt < nkt; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nkt) stage((kt + 1) & 1);
        compute(kt & 1);
    }

    // ---- epilogue ------------------------------------------------------------------------------
    const float alpha = p.alpha;
    const float* bias = p.bias;
    if (bias && p.step_ptr) bias += (long long)(*p.step_ptr) * p.bias_step_stride;
    uint16_t* __restrict__ C = p.C + bz * p.sC;
    const uint16_t* __restrict__ R = p.R ? p.R + bz * p.sR : nullptr;

    // ---- staged epilogue (the normal case): the wave parks its 32 x (TN*32) fp32 sub-tile in LDS, then writes /
    //      reads HBM in whole rows - 16-byte coalesced stores and residual loads instead of the 8-byte,
    //      32-lines-per-instruction pattern the MFMA C layout would give.  Per-wave private region, XOR-swizzled
    //      16-byte chunks (conflict-free for both the b128 writes and the row reads). ----
    {
        const bool geglu = p.epi == 1;
        const int ncols_out = geglu ? (p.N >> 1) : p.N;
        const bool aligned = ((p.ldc & 7) == 0) && ((ncols_out & 7) == 0) && (!R || (p.ldr & 7) == 0) &&
                             ((((uintptr_t)C | (uintptr_t)R | (uintptr_t)bias) & 15) == 0) &&
                             (((p.sC | p.sR) & 7) == 0) && (!geglu || (TN % 2 == 0));
        if (aligned) {
            constexpr int WCOLS = TN * 32;
            float* stg = (float*)smem + wave * (32 * WCOLS);
            __syncthreads();  // every wave has left the K loop: the tile buffers may be overwritten
            const int wcol0 = n0 + wn * WCOLS;  // first (permuted, for GEGLU) weight row of this wave
            auto drain = [&](auto oc_tag, int mt) {
                constexpr int OC = decltype(oc_tag)::value;  // output columns of the wave per pass
                constexpr int CPR = OC / 8;                  // 16-byte bf16 chunks per output row
                const int ocol0 = geglu ? (wcol0 >> 1) : wcol0;
#pragma unroll 2
                for (int idx = lane; idx < 32 * CPR; idx += 64) {
                    const int r = idx / CPR, cj = idx - r * CPR;
                    const int m = m0 + wm * TM * 32 + mt * 32 + r;
                    const int n = ocol0 + cj * 8;
                    if (m < p.M && n < ncols_out) {
                        const float4 a = *(const float4*)(stg + r * WCOLS + (((2 * cj) ^ (r & 7)) << 2));
                        const float4 b = *(const float4*)(stg + r * WCOLS + (((2 * cj + 1) ^ (r & 7)) << 2));
                        float f[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
                        if (R) {
                            const bf16x8_raw rr = *(const bf16x8_raw*)(R + (long long)m * p.ldr + n);
                            float g[8];
                            unpack8(rr, g);
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] += g[e];
                        }
                        if (p.epi == 2) {
#pragma unroll
                            for (int e = 0; e < 8; ++e) f[e] = silu_f(f[e]);
                        }
                        *(bf16x8_raw*)(C + (long long)m * p.ldc + n) = pack8(f);
                    }
                }
            };
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const int mrow = m0 + wm * TM * 32 + mt * 32 + l31;
                const float bm_ = (bias && p.bias_mode == 2 && mrow < p.M) ? bias[mrow] : 0.f;
                if (geglu) {
                    if constexpr (TN % 2 == 0) {
#pragma unroll
                        for (int nt = 0; nt < TN; nt += 2)
#pragma unroll
                            for (int g4 = 0; g4 < 4; ++g4) {
                                const int nv = wcol0 + nt * 32 + 8 * g4 + 4 * lhi;
                                float4 bv = make_float4(0.f, 0.f, 0.f, 0.f), bg = bv;
                                if (bias && nv + 35 < p.N) {
                                    bv = *(const float4*)(bias + nv);
                                    bg = *(const float4*)(bias + nv + 32);
                                }
                                float4 o;
                                o.x = (acc[nt][mt][4 * g4 + 0] * alpha + bv.x) * gelu_erf_f(acc[nt + 1][mt][4 * g4 + 0] * alpha + bg.x);
                                o.y = (acc[nt][mt][4 * g4 + 1] * alpha + bv.y) * gelu_erf_f(acc[nt + 1][mt][4 * g4 + 1] * alpha + bg.y);
                                o.z = (acc[nt][mt][4 * g4 + 2] * alpha + bv.z) * gelu_erf_f(acc[nt + 1][mt][4 * g4 + 2] * alpha + bg.z);
                                o.w = (acc[nt][mt][4 * g4 + 3] * alpha + bv.w) * gelu_erf_f(acc[nt + 1][mt][4 * g4 + 3] * alpha + bg.w);
                                const int chunk = (nt >> 1) * 8 + 2 * g4 + lhi;
                                *(float4*)(stg + l31 * WCOLS + ((chunk ^ (l31 & 7)) << 2)) = o;
                            }
                        drain(std::integral_constant<int, (TN / 2) * 32 + (TN < 2 ? 32 : 0)>{}, mt);
                    }
                } else {
#pragma unroll
                    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const int nb = wcol0 + nt * 32 + 8 * g4 + 4 * lhi;
                            float4 bv = make_float4(bm_, bm_, bm_, bm_);
                            if (bias && p.bias_mode == 1 && nb + 3 < p.N) {
                                const float4 t4 = *(const float4*)(bias + nb);
                                bv.x += t4.x;
                                bv.y += t4.y;
                                bv.z += t4.z;
                                bv.w += t4.w;
                            }
                            float4 o;
                            o.x = acc[nt][mt][4 * g4 + 0] * alpha + bv.x;
                            o.y = acc[nt][mt][4 * g4 + 1] * alpha + bv.y;
                            o.z = acc[nt][mt][4 * g4 + 2] * alpha + bv.z;
                            o.w = acc[nt][mt][4 * g4 + 3] * alpha + bv.w;
                            const int chunk = nt * 8 + 2 * g4 + lhi;
                            *(float4*)(stg + l31 * WCOLS + ((chunk ^ (l31 & 7)) << 2)) = o;
                        }
                    drain(std::integral_constant<int, WCOLS>{}, mt);
                }
            }
            return;
        }
    }

    // ---- fallback epilogue straight from the MFMA registers (odd leading dims / N, e.g. the 77-token V^T) ----
    if (p.epi == 1) {  // GEGLU: even n-tile = value rows, odd n-tile = gate rows of the same channels
        if constexpr (TN % 2 == 0) {
            const int nout = p.N >> 1;
#pragma unroll
            for (int nt = 0; nt < TN; nt += 2)
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
                    const int m = m0 + wm * TM * 32 + mt * 32 + l31;
                    if (m >= p.M) continue;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int nv = n0 + wn * TN * 32 + nt * 32 + 8 * g4 + 4 * lhi;  // value row (permuted index)
                        const int oc = ((n0 + wn * TN * 32 + nt * 32) >> 1) + 8 * g4 + 4 * lhi;
                        if (oc >= nout) continue;
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float a = acc[nt][mt][4 * g4 + e] * alpha;
                            float g = acc[nt + 1][mt][4 * g4 + e] * alpha;
                            if (bias) {
                                a += bias[nv + e];
                                g += bias[nv + 32 + e];
                            }
                            v[e] = a * gelu_erf_f(g);
                        }
                        uint2 o;
                        o.x = pack_bf16x2(v[0], v[1]);
                        o.y = pack_bf16x2(v[2], v[3]);
                        *(uint2*)(C + (long long)m * p.ldc + oc) = o;
                    }
                }
        }
        return;
    }

    const bool vec_ok = ((p.ldc & 3) == 0) && (!R || (p.ldr & 3) == 0);
#pragma unroll
    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const int m = m0 + wm * TM * 32 + mt * 32 + l31;
            if (m >= p.M) continue;
            const float bm_ = (bias && p.bias_mode == 2) ? bias[m] : 0.f;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int nb = n0 + wn * TN * 32 + nt * 32 + 8 * g4 + 4 * lhi;
                if (nb >= p.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[nt][mt][4 * g4 + e] * alpha + bm_;
                    if (bias && p.bias_mode == 1 && nb + e < p.N) v[e] += bias[nb + e];
                }
                if (vec_ok && nb + 3 < p.N) {
                    if (R) {
                        const uint2 r = *(const uint2*)(R + (long long)m * p.ldr + nb);
                        v[0] += __uint_as_float(r.x << 16);
                        v[1] += __uint_as_float(r.x & 0xffff0000u);
                        v[2] += __uint_as_float(r.y << 16);
                        v[3] += __uint_as_float(r.y & 0xffff0000u);
                    }
                    if (p.epi == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = silu_f(v[e]);
                    }
                    uint2 o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
                    *(uint2*)(C + (long long)m * p.ldc + nb) = o;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (nb + e >= p.N) continue;
                        float t = v[e];
                        if (R) t += bf16_to_f32(R[(long long)m * p.ldr + nb + e]);
                        if (p.epi == 2) t = silu_f(t);
                        C[(long long)m * p.ldc + nb + e] = f32_to_bf16(t);
                    }
                }
            }
        }
}

template <int WM, int WN, int TM, int TN, bool CONV>
int launch_igemm_t(const sdv_gemm_args& a, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int STG = WM * WN * 32 * TN * 32 * 4;              // fp32 staging of the epilogue
    constexpr int LDS = 2 * (BM + BN) * 128 > STG ? 2 * (BM + BN) * 128 : STG;
    static_assert(LDS <= 160 * 1024, "tile does not fit the 160 KiB LDS");
    static bool attr_set = false;
    if (LDS > 64 * 1024 && !attr_set) {
        (void)hipFuncSetAttribute((const void*)igemm_kernel<WM, WN, TM, TN, CONV>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
        attr_set = true;
    }
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    dim3 grid(tiles_m * tiles_n, 1, a.batch > 0 ? a.batch : 1);
    hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN, CONV>), grid, dim3(WM * WN * 64), LDS, stream, a);
    SDV_CHECK_LAUNCH("sdv_gemm_bf16");
    return SDV_OK;
}

template <int WM, int WN, int TM, int TN>
int launch_igemm(const sdv_gemm_args& a, hipStream_t stream) {
    return a.mode == 0 ? launch_igemm_t<WM, WN, TM, TN, false>(a, stream) : launch_igemm_t<WM, WN, TM, TN, true>(a, stream);
}

}  // namespace

extern "C" int sdv_gemm_bf16(const sdv_gemm_args* args, void* stream) {
    SDV_REQUIRE(args != nullptr, "sdv_gemm_bf16: null args");
    sdv_gemm_args a = *args;
    SDV_REQUIRE(a.X && a.W && a.C, "sdv_gemm_bf16: null operand");
    SDV_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "sdv_gemm_bf16: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    SDV_REQUIRE(a.K % kBK == 0, "sdv_gemm_bf16: K=%d must be a multiple of %d", a.K, kBK);
    SDV_REQUIRE(a.mode >= 0 && a.mode <= 3, "sdv_gemm_bf16: bad mode %d", a.mode);
    if (!a.X2) {
        a.C1 = a.K;
        a.ldx2 = a.ldx;
        a.X2 = a.X;
    }
    SDV_REQUIRE(a.C1 % kBK == 0 && a.C1 > 0 && a.C1 <= a.K, "sdv_gemm_bf16: C1=%d must be a multiple of %d in (0,K]",
                a.C1, kBK);
    SDV_REQUIRE(a.ldx % 8 == 0 && a.ldx2 % 8 == 0 && a.ldw % 8 == 0, "sdv_gemm_bf16: ldx/ldx2/ldw must be multiples of 8");
    if (a.mode != 0) {
        SDV_REQUIRE(a.zero_page != nullptr, "sdv_gemm_bf16: conv modes need zero_page");
        SDV_REQUIRE(a.Hin > 0 && a.Win > 0 && a.Hout > 0 && a.Wout > 0, "sdv_gemm_bf16: bad conv geometry");
        SDV_REQUIRE(a.M % (a.Hout * a.Wout) == 0, "sdv_gemm_bf16: M must be nimg*Hout*Wout");
        SDV_REQUIRE(a.ldw >= 9 * a.K, "sdv_gemm_bf16: conv weights must be [N][3][3][K]");
        SDV_REQUIRE(a.batch <= 1, "sdv_gemm_bf16: conv modes are not batched");
        if (a.mode == 1) SDV_REQUIRE(a.Hin == a.Hout && a.Win == a.Wout, "conv s1 geometry");
        if (a.mode == 2) SDV_REQUIRE(a.Hout == (a.Hin + 1) / 2 && a.Wout == (a.Win + 1) / 2, "conv s2 geometry");
        if (a.mode == 3) SDV_REQUIRE(a.Hout == 2 * a.Hin && a.Wout == 2 * a.Win, "upsample-conv geometry");
    }
    if (a.epi == 1) {
        SDV_REQUIRE(a.N % 64 == 0, "sdv_gemm_bf16: GEGLU needs N %% 64 == 0");
        SDV_REQUIRE(a.ldc % 4 == 0, "sdv_gemm_bf16: GEGLU needs ldc %% 4 == 0");
    }
    if (a.bias_mode == 0 && a.bias) a.bias_mode = 1;
    if (a.alpha == 0.f) a.alpha = 1.f;
    hipStream_t s = (hipStream_t)stream;
    int tile = a.tile;
    const long long nb = a.batch > 0 ? a.batch : 1;
    auto blocks = [&](int bm, int bn) { return (long long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn) * nb; };
    if (tile == 0) {
        // Cost model over the compiled tiles: rounds of resident workgroups x work per round / relative MFMA
        // efficiency of the tile shape.  Big 8-wave tiles move the fewest L2->LDS bytes per MFMA (the 128x128
        // tile is L2-bandwidth bound near 0.9 PF/s) but quantise badly on the low-resolution levels.
        // rate = measured MFMA throughput per busy CU (TFLOP/s, tools/tile_sweep.py on MI355X, K >= 1280)
        struct Cand { int id, bm, bn; float rate; };
        static const Cand cands[] = {{6, 256, 320, 4.9f}, {7, 256, 256, 4.6f}, {9, 128, 320, 4.2f}, {8, 256, 128, 3.2f},
                                     {1, 128, 128, 3.15f}, {2, 128, 64, 2.2f}, {3, 64, 64, 2.5f}};
        double best = 1e300;
        for (const Cand& c : cands) {
            if (a.epi == 1 && (c.id == 6 || c.id == 9 || c.id == 3)) continue;   // GEGLU pairs n-tiles: even TN only
            const long long per_cu = (blocks(c.bm, c.bn) + 255) / 256;            // workgroups the busiest CU runs
            const double cost = (double)per_cu * c.bm * c.bn / c.rate;            // padded tiles are counted
            if (cost < best) {
                best = cost;
                tile = c.id;
            }
        }
    }
    if (a.epi == 1 && tile == 3) tile = 2;
    if (a.epi == 1 && (tile == 6 || tile == 9)) tile = (a.N % 256 == 0) ? 7 : 1;   // GEGLU pairs n-tiles: even TN only
    switch (tile) {
        case 1: return launch_igemm<2, 2, 2, 2>(a, s);   // 128 x 128, 4 waves
        case 2: return launch_igemm<4, 1, 1, 2>(a, s);   // 128 x  64
        case 3: return launch_igemm<2, 2, 1, 1>(a, s);   //  64 x  64
        case 4: return launch_igemm<2, 2, 4, 2>(a, s);   // 256 x 128, 4 waves
        case 6: return launch_igemm<4, 2, 2, 5>(a, s);   // 256 x 320, 8 waves (UNet widths are multiples of 320)
        case 7: return launch_igemm<4, 2, 2, 4>(a, s);   // 256 x 256, 8 waves
        case 8: return launch_igemm<4, 2, 2, 2>(a, s);   // 256 x 128, 8 waves
        case 9: return launch_igemm<4, 2, 1, 5>(a, s);   // 128 x 320, 8 waves
        default: SDV_REQUIRE(false, "sdv_gemm_bf16: bad tile %d", tile);
    }
    return SDV_OK;
}
