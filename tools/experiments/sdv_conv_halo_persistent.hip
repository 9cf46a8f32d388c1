// TILE-WALK VARIANT of sdv_conv_halo.hip (one loop over tiles per workgroup; -DHALO_PERSISTENT launches one workgroup per CU, the next
// tile's X window streams in behind the epilogue).  Correct (tools/conv_halo_ab.py, HALO_SRC=sdv_conv_halo_persistent.hip), but the
// variant is 0-6 % SLOWER than the one-tile-per-workgroup file, persistent or not (profiles/round3_conv_halo_prototype.txt) - also with
// the register discipline of the shipped igemm applied (lane-derived state re-derived per tile from an opaque copy: 120 -> 36 B of scratch).
//
// conv3x3 (stride 1, pad 1, NHWC bf16) with the input window staged ONCE per channel slab - the "halo tile" form of the
// implicit GEMM in sdv_gemm.hip, for the ResBlock convolutions of UNet2DConditionModel / AutoencoderKL
// (reference: unet(...) stable_diffusion_pipeline.py:418, vae.decode :433).
//
// Why a second conv kernel: DESIGN.md (d) "Staged bytes per FLOP" - the igemm tiles run near the CU's LDS-fill
// rate (18-20 B/clk/CU for cache-resident operands), and the tap-major implicit GEMM stages (256 + 320) x 64 x 2 B for EVERY
// (tap, 64-channel slab): 663 KB per channel slab of a 256 x 320 tile, 295 KB of it the same pixels nine times.  Here the
// (rows + 2) x (W + 2) pixel window of the tile's 256 output pixels is staged once per channel slab (<= 400 halo pixels x 128 B =
// 51 KB; image borders and the padding columns are zero-filled by the buffer range check) and the nine taps read their X
// fragments from it at shifted rows; only the nine 40 KB W slabs stream, double-buffered: 411 KB per channel slab, 1.6x fewer
// staged bytes per FLOP.
//
// Tile: 256 output pixels (whole image rows; several whole images when H*W < 256) x 320 output channels, 8 waves = 4 (pixels) x 2
// (channels), wave tile 64 x 160 = TM 2 x TN 5 accumulators of v_mfma_f32_32x32x16_bf16 (A = W rows, B = pixels - the layout of
// the igemm, so the results differ from it only by the order the (channel slab, tap) partial sums are accumulated in).
// LDS rows are 128 B (64 channels) with the igemm's XOR swizzle: position (row, chunk') holds channel chunk chunk' ^ ((row >> 1) & 7).
#include <type_traits>

#include "sdv_common.h"

namespace {

constexpr unsigned kOOB = 0x80000000u;
constexpr int kRecords = 0x7ffffff0;
#ifdef HALO_WAVES4   // sandbox: ONE wave per SIMD, wave tile 128 x 160 (TM 4 x TN 5 = 320 accumulator registers -> AccVGPRs)
constexpr int NWAVE = 4, TM = 4;
#else
constexpr int NWAVE = 8, TM = 2;
#endif
#ifndef HALO_TN
#define HALO_TN 5
#endif
constexpr int TN = HALO_TN;                        // n-tiles per wave (sandbox: 4 -> 256-column tiles, 16 accumulator tiles = 256 AccVGPRs)
constexpr int BM = 256, BN = 2 * TN * 32, ROWB = 128, KSTEPS = 4;
constexpr int NXP = 56 / NWAVE, NWP = BN / 8 / NWAVE;  // LDS-DMA pieces (1 KB = 8 rows) per wave: X window (56 >= 50), W slab (40)
constexpr int XROWS = 8 * 56;                      // 448 rows staged (the layouts use <= 400; surplus pieces read zeros)
constexpr int X_BYTES = XROWS * ROWB;             // 57344
constexpr int W_BYTES = BN * ROWB;                // 40960
constexpr int LDS_BYTES = X_BYTES + 2 * W_BYTES;  // 139264

struct HaloArgs {
    const uint16_t* X;
    const uint16_t* X2;     // second source of a channel concat (up blocks), or null
    const uint16_t* W;      // [Cout][9 * (C1 + C2)], OHWI: column = tap * K + channel, tap = ky * 3 + kx
    const float* bias;      // [Cout] (already offset by the step table row)
    const uint16_t* R;      // residual [M][ldr] or null
    uint16_t* C;            // [M][ldc]
    int nimg, H, Wd, C1, C2, Cout, ldx, ldx2, ldw, ldr, ldc;
};

__device__ __forceinline__ int swz(int r) { return (r >> 1) & 7; }

__global__ __launch_bounds__(NWAVE * 64) void conv3x3_halo_kernel(const HaloArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const ldsX = smem;
    char* const ldsW = smem + X_BYTES;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int K = p.C1 + p.C2;
    const int tiles_n = p.Cout / BN;
    const int Wd = p.Wd, H = p.H, HW = H * Wd;
    const long long M = (long long)p.nimg * HW;

    // ---- tile geometry: SEG whole images of RS rows each (SEG = 1: RS consecutive rows of one image) ----
    const int SEG = HW >= BM ? 1 : BM / HW;
    const int RS = HW >= BM ? BM / Wd : H;
    const int PW = Wd + 2;
    const int SEGSZ = (RS + 2) * PW;
    const int SEGPX = RS * Wd;                       // output pixels per segment
    const int ntiles = (int)((M + BM - 1) / BM) * tiles_n;

    // ---- staging addresses (tile-invariant over the K loop) ----
    // X window piece g = wave + 8 i covers halo rows g*8 .. g*8+7; lane -> (row g*8 + lane/8, LDS chunk lane%8)
    int xpix[NXP];        // pixel index relative to image img0 (or -1: zero row)
    int xchk[NXP];        // byte offset of the channel chunk this lane fetches (swizzled)
    int m0 = 0, n0 = 0;   // the tile the ADDRESSING points at
    __amdgpu_buffer_rsrc_t rs_x1, rs_x2, rs_w;
    // (everything derived from the lane id is re-derived per tile from an OPAQUE copy of it, so that none of it stays alive across
    //  the epilogue, where the 160 accumulators + the staging state fill the register file - the igemm's discipline)
    auto tile_window = [&](int tile, int lk, int* xp, int* xc, __amdgpu_buffer_rsrc_t& r1, __amdgpu_buffer_rsrc_t& r2, int& tm0, int& tn0) {
        const int bn = tile % tiles_n, bm = tile / tiles_n;
        tm0 = bm * BM;
        tn0 = bn * BN;
        const int img0 = tm0 / HW;
        const int y0 = (tm0 - img0 * HW) / Wd;            // first image row of the tile (0 when SEG > 1)
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
            const int hr = (wave + NWAVE * i) * 8 + (lk >> 3);
            const int seg = hr / SEGSZ, rem = hr - seg * SEGSZ;
            const int hy = rem / PW, hx = rem - hy * PW;
            const int y = y0 + hy - 1, x = hx - 1;
            const bool ok = seg < SEG && img0 + seg < p.nimg && (unsigned)y < (unsigned)H && (unsigned)x < (unsigned)Wd;
            xp[i] = ok ? seg * HW + y * Wd + x : -1;
            xc[i] = ((lk & 7) ^ swz(hr)) * 16;
        }
        r1 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.X + (long long)img0 * HW * p.ldx), 0, kRecords, 0x00020000);
        r2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.X2 ? p.X2 + (long long)img0 * HW * p.ldx2 : p.X), 0, kRecords, 0x00020000);
    };
    unsigned wvo[NWP];
    auto stage_x_with = [&](int cs, const int* xp, const int* xc, const __amdgpu_buffer_rsrc_t& r1, const __amdgpu_buffer_rsrc_t& r2) {
        // channel slab cs of the concatenated input: source 1 holds channels [0, C1), source 2 the rest
        const bool s2 = cs * 64 >= p.C1;
        const int ld2 = 2 * (s2 ? p.ldx2 : p.ldx);
        const int soff = 2 * (s2 ? cs * 64 - p.C1 : cs * 64);
        const __amdgpu_buffer_rsrc_t rs = s2 ? r2 : r1;
#pragma unroll
        for (int i = 0; i < NXP; ++i) {
            const unsigned vo = xp[i] >= 0 ? (unsigned)(xp[i] * ld2 + xc[i]) : kOOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(ldsX + (wave + NWAVE * i) * 1024), 16,
                                                     (int)vo, soff, 0, 0);
        }
    };
    auto stage_x = [&](int cs) { stage_x_with(cs, xpix, xchk, rs_x1, rs_x2); };
    auto stage_w = [&](int buf, int cs, int tap) {
        const int soff = 2 * (tap * K + cs * 64);
        char* const base = ldsW + buf * W_BYTES;
#pragma unroll
        for (int i = 0; i < NWP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(base + (wave + NWAVE * i) * 1024), 16,
                                                     (int)wvo[i], soff, 0, 0);
    };

    // ---- fragment addresses ----
    int hrow0[TM];          // halo row of this lane's pixel of m-tile mt, centre tap
    int wfo[KSTEPS];        // W fragment: row wn*160 + nt*32 + l31 (the n-tile part is an immediate offset)
    auto lane_state = [&](int lk) {
        const int k31 = lk & 31, khi = lk >> 5;
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const int t = wm * (TM * 32) + mt * 32 + k31;
            const int seg = t / SEGPX, r = t - seg * SEGPX;
            const int ry = r / Wd, x = r - ry * Wd;
            hrow0[mt] = seg * SEGSZ + (ry + 1) * PW + (x + 1);
        }
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) wfo[ks] = (wn * (TN * 32) + k31) * ROWB + (((ks * 2 + khi) ^ swz(k31)) << 4);
#pragma unroll
        for (int i = 0; i < NWP; ++i) {
            const int rw = (wave + NWAVE * i) * 8 + (lk >> 3);
            wvo[i] = (unsigned)(rw * p.ldw * 2 + (((lk & 7) ^ swz(rw)) * 16));
        }
    };

    f32x16_t acc[TN][TM];

    auto compute = [&](int tap, int buf) {
        const int toff = (tap / 3 - 1) * PW + (tap - (tap / 3) * 3 - 1);
        int xbase[TM], xs[TM];
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const int hr = hrow0[mt] + toff;
            xbase[mt] = hr * ROWB;
            xs[mt] = swz(hr);
        }
        const char* const wb = ldsW + buf * W_BYTES;
        auto xfrag = [&](int ks, int mt) __attribute__((always_inline)) {
            return *(const bf16x8_t*)(ldsX + xbase[mt] + (((ks * 2 + lhi) ^ xs[mt]) << 4));
        };
        auto wfrag = [&](int st) __attribute__((always_inline)) {
            return *(const bf16x8_t*)(wb + wfo[st / TN] + (st % TN) * 32 * ROWB);
        };
#ifdef HALO_SETPRIO
        __builtin_amdgcn_s_setprio(1);
#endif
        // rotating W fragments, as the bf16 K loop of the igemm (sdv_gemm.hip, SDV_BF16_ROT_AH)
        constexpr int AH = 2, STEPS = KSTEPS * TN;
        bf16x8_t xa[2][TM], wq[AH + 1];
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) xa[0][mt] = xfrag(0, mt);
#pragma unroll
        for (int a = 0; a < AH; ++a) wq[a] = wfrag(a);
#pragma unroll
        for (int st = 0; st < STEPS; ++st) {
            const int ks = st / TN, nt = st % TN;
            if (st + AH < STEPS) wq[(st + AH) % (AH + 1)] = wfrag(st + AH);
            if (nt == TN - 1 - AH && ks + 1 < KSTEPS) {
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) xa[(ks + 1) & 1][mt] = xfrag(ks + 1, mt);
            }
#pragma unroll
            for (int mt = 0; mt < TM; ++mt)
                acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[st % (AH + 1)], xa[ks & 1][mt], acc[nt][mt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef HALO_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
    };

    // ---- tile walk (gridDim.x == number of tiles: one tile per workgroup; fewer workgroups: persistent) ----
    const int ncs = K / 64;
    int tile = blockIdx.x;
    bool first = true;
    while (true) {
    {
        int lk = lane;
        asm volatile("" : "+v"(lk));
        lane_state(lk);
        tile_window(tile, lk, xpix, xchk, rs_x1, rs_x2, m0, n0);
        rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)(p.W + (long long)n0 * p.ldw), 0, kRecords, 0x00020000);
    }
    if (first) stage_x(0);          // (later tiles: their first X window was fetched behind the previous tile's epilogue)
    first = false;
    stage_w(0, 0, 0);
#pragma unroll
    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
        for (int mt = 0; mt < TM; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[nt][mt][e] = 0.f;
    // ---- K loop: channel slabs outside, taps inside ----
    int buf = 0;
    for (int cs = 0; cs < ncs; ++cs) {
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();        // this tap's W slab (and, tap 0, the X window) landed; every wave is past the previous tap
            int ntap = tap + 1, ncsl = cs;
            if (ntap == 9) {
                ntap = 0;
                ++ncsl;
            }
            if (ncsl < ncs) stage_w(buf ^ 1, ncsl, ntap);
            compute(tap, buf);
            buf ^= 1;
        }
#ifndef HALO_WHATIF_NO_X_RESTAGE   // (timing-only what-if build: wrong results)
        if (cs + 1 < ncs) {
            __syncthreads();        // every wave has read the X window of this slab for the last time
            stage_x(cs + 1);
        }
#endif
    }
    // the tile the epilogue writes; the addressing moves on to the next tile, whose X window streams in behind the epilogue
    const int em0 = m0, en0 = n0;
    const int next = tile + (int)gridDim.x;
    const bool has_next = next < ntiles;
    __syncthreads();                // every wave is past its last fragment read: X window and W buffers are free
    if (has_next) {                 // the next tile's first X window, from addressing state that dies right here
        int lk = lane;
        asm volatile("" : "+v"(lk));
        int xp2[NXP], xc2[NXP], dm0, dn0;
        __amdgpu_buffer_rsrc_t r1, r2;
        tile_window(next, lk, xp2, xc2, r1, r2, dm0, dn0);
        stage_x_with(0, xp2, xc2, r1, r2);
    }

#ifdef HALO_ROWMAJOR_EPILOGUE
    // ---- row-major epilogue (the igemm's idea, in its plainest form): a pass = 32 pixels x 64 channels (bf16 slab; 32 channels as fp32
    //      when a residual has to be added before the single rounding) goes through a wave-private LDS slab (rows of 128 B + 16 B
    //      pad) and leaves as 16-byte stores - 8 adjacent lanes write one pixel's 128 contiguous bytes.  Two slabs per wave. ----
    {
        typedef unsigned int __attribute__((ext_vector_type(4), may_alias)) slab_u4;
        typedef unsigned int __attribute__((ext_vector_type(2), may_alias)) slab_u2;
        char* const slab0 = ldsW + wave * 2 * 4608;
        // straight-line code per case (every pass index is a compile-time constant: the accumulators are never indexed dynamically)
        auto run = [&](auto hasr_tag) {
            constexpr bool has_r = decltype(hasr_tag)::value;
            constexpr int NPP = has_r ? TN : (TN + 1) / 2;     // passes per m-tile
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const long long mrow0 = (long long)em0 + wm * (TM * 32) + mt * 32;
#pragma unroll
                for (int pp = 0; pp < NPP; ++pp) {
                    char* const slab = slab0 + ((mt * NPP + pp) & 1) * 4608;
                    const int nt0 = has_r ? pp : 2 * pp;
                    const int cnt = has_r ? 1 : (TN - nt0 < 2 ? TN - nt0 : 2);
                    // phase 1: bias in the MFMA layout, park
#pragma unroll
                    for (int k = 0; k < cnt; ++k) {
                        const int nt = nt0 + k;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int c0 = en0 + wn * (TN * 32) + nt * 32 + 8 * q + 4 * lhi;
                            const float4 b = *(const float4*)(p.bias + c0);
                            const float v[4] = {acc[nt][mt][4 * q] + b.x, acc[nt][mt][4 * q + 1] + b.y, acc[nt][mt][4 * q + 2] + b.z,
                                                acc[nt][mt][4 * q + 3] + b.w};
                            const int cl = k * 32 + 8 * q + 4 * lhi;
                            if constexpr (has_r)
                                *(slab_u4*)(slab + l31 * 144 + cl * 4) =
                                    slab_u4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                            else
                                *(slab_u2*)(slab + l31 * 144 + cl * 2) = slab_u2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                        }
                    }
                    asm volatile("" ::: "memory");
                    // phase 2: rows back out, 16 bytes per lane
                    const int col0 = en0 + wn * (TN * 32) + nt0 * 32;
                    if constexpr (has_r) {
#pragma unroll
                        for (int it = 0; it < 2; ++it) {
                            const int idx = lane + 64 * it, r = idx >> 2, cj = idx & 3;
                            const long long m = mrow0 + r;
                            const slab_u4 lo = *(const slab_u4*)(slab + r * 144 + cj * 32), hi = *(const slab_u4*)(slab + r * 144 + cj * 32 + 16);
                            if (m < M) {
                                const u32x4_t rr = *(const u32x4_t*)(p.R + m * p.ldr + col0 + cj * 8);
                                float f[8] = {__uint_as_float(lo[0]), __uint_as_float(lo[1]), __uint_as_float(lo[2]), __uint_as_float(lo[3]),
                                              __uint_as_float(hi[0]), __uint_as_float(hi[1]), __uint_as_float(hi[2]), __uint_as_float(hi[3])};
#pragma unroll
                                for (int e = 0; e < 4; ++e) {
                                    f[2 * e] += __uint_as_float(rr[e] << 16);
                                    f[2 * e + 1] += __uint_as_float(rr[e] & 0xffff0000u);
                                }
                                *(u32x4_t*)(p.C + m * p.ldc + col0 + cj * 8) =
                                    u32x4_t{pack_bf16x2(f[0], f[1]), pack_bf16x2(f[2], f[3]), pack_bf16x2(f[4], f[5]), pack_bf16x2(f[6], f[7])};
                            }
                        }
                    } else {
                        const int cpo = cnt * 4;            // 8-column groups per row: 8 or 4
#pragma unroll
                        for (int it = 0; it < 4; ++it) {
                            if (it * 64 >= 32 * cpo) continue;
                            const int idx = lane + 64 * it, r = idx / cpo, cj = idx - r * cpo;
                            const long long m = mrow0 + r;
                            const slab_u4 d = *(const slab_u4*)(slab + r * 144 + cj * 16);
                            if (m < M) *(u32x4_t*)(p.C + m * p.ldc + col0 + cj * 8) = u32x4_t{d[0], d[1], d[2], d[3]};
                        }
                    }
                    asm volatile("" ::: "memory");
                }
            }
        };
        if (p.R) run(std::true_type{});
        else run(std::false_type{});
    }
#else
    // ---- epilogue straight from the MFMA layout: lane (l31, lhi) holds, per (nt, mt), pixel l31 x channels (r&3) + 8 (r>>2) + 4 lhi ----
#pragma unroll
    for (int mt = 0; mt < TM; ++mt) {
        const long long m = (long long)em0 + wm * (TM * 32) + mt * 32 + l31;
#ifdef HALO_WHATIF_NO_EPILOGUE      // (timing-only what-if build: nothing is stored unless an accumulator is exactly 12345)
        if (acc[0][mt][0] != 12345.f) continue;
#endif
        if (m >= M) continue;
        uint16_t* const crow = p.C + m * p.ldc;
        const uint16_t* const rrow = p.R ? p.R + m * p.ldr : nullptr;
#pragma unroll
        for (int nt = 0; nt < TN; ++nt) {
            u32x2_t rr[4];
            if (rrow) {
#pragma unroll
                for (int q = 0; q < 4; ++q) rr[q] = *(const u32x2_t*)(rrow + en0 + wn * (TN * 32) + nt * 32 + 8 * q + 4 * lhi);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c0 = en0 + wn * (TN * 32) + nt * 32 + 8 * q + 4 * lhi;
                const float4 b = *(const float4*)(p.bias + c0);
                float v[4] = {acc[nt][mt][4 * q] + b.x, acc[nt][mt][4 * q + 1] + b.y, acc[nt][mt][4 * q + 2] + b.z,
                              acc[nt][mt][4 * q + 3] + b.w};
                if (rrow) {
                    v[0] += __uint_as_float(rr[q][0] << 16);
                    v[1] += __uint_as_float(rr[q][0] & 0xffff0000u);
                    v[2] += __uint_as_float(rr[q][1] << 16);
                    v[3] += __uint_as_float(rr[q][1] & 0xffff0000u);
                }
                *(u32x2_t*)(crow + c0) = u32x2_t{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            }
        }
    }
#endif
    if (!has_next) break;
    __syncthreads();                // every wave has read its staging slabs back: the W buffers are free again
    tile = next;
    }   // tile walk
}

}  // namespace

extern "C" int sdv_conv3x3_halo_bf16(const sdv_bf16* X, const sdv_bf16* X2, const sdv_bf16* W, const float* bias, const sdv_bf16* R,
                                     sdv_bf16* C, int32_t nimg, int32_t H, int32_t Wd, int32_t C1, int32_t C2, int32_t Cout,
                                     int32_t ldx, int32_t ldx2, int32_t ldw, int32_t ldr, int32_t ldc, const int32_t* step_ptr,
                                     int32_t bias_step_stride, void* stream) {
    SDV_REQUIRE(X && W && bias && C && nimg > 0 && H > 0 && Wd > 0, "sdv_conv3x3_halo_bf16: bad args");
    SDV_REQUIRE(C1 > 0 && C1 % 64 == 0 && C2 >= 0 && C2 % 64 == 0 && (C2 == 0 || X2), "sdv_conv3x3_halo_bf16: channel counts must be multiples of 64");
    SDV_REQUIRE(Cout % BN == 0, "sdv_conv3x3_halo_bf16: Cout=%d must be a multiple of %d", Cout, BN);
    const int HW = H * Wd;
    SDV_REQUIRE(BM % Wd == 0 && (HW % BM == 0 || BM % HW == 0), "sdv_conv3x3_halo_bf16: %d x %d images do not tile into %d-pixel row blocks", H, Wd, BM);
    {   // the halo window must fit the staged rows
        const int seg = HW >= BM ? 1 : BM / HW, rs = HW >= BM ? BM / Wd : H;
        SDV_REQUIRE(seg * (rs + 2) * (Wd + 2) <= XROWS, "sdv_conv3x3_halo_bf16: window of %d x %d images exceeds %d halo rows", H, Wd, XROWS);
    }
    SDV_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0 && (!R || ldr % 4 == 0) && (C2 == 0 || ldx2 % 8 == 0), "sdv_conv3x3_halo_bf16: unaligned leading dims");
    SDV_REQUIRE((long long)HW * 4 * (ldx > ldx2 ? ldx : ldx2) * 2 < 0x7fffffffLL, "sdv_conv3x3_halo_bf16: tile window exceeds 31-bit offsets");
    static bool attr_set[16] = {};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 16 && !attr_set[dev]) {
        (void)hipFuncSetAttribute((const void*)conv3x3_halo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        attr_set[dev] = true;
    }
    HaloArgs a;
    a.X = X;
    a.X2 = X2;
    a.W = W;
    a.bias = bias;   // the step-table row is selected on the device side below (graph-replayable)
    a.R = R;
    a.C = C;
    a.nimg = nimg, a.H = H, a.Wd = Wd, a.C1 = C1, a.C2 = C2, a.Cout = Cout;
    a.ldx = ldx, a.ldx2 = ldx2, a.ldw = ldw, a.ldr = ldr, a.ldc = ldc;
    SDV_REQUIRE(step_ptr == nullptr || bias_step_stride == 0, "sdv_conv3x3_halo_bf16: the per-step bias table is not wired yet (prototype)");
    const long long M = (long long)nimg * HW;
    const long long tiles = ((M + BM - 1) / BM) * (Cout / BN);
#ifdef HALO_PERSISTENT
    const long long grid = tiles < 256 ? tiles : 256;   // one workgroup per CU (sandbox: MI355X)
#else
    const long long grid = tiles;
#endif
    hipLaunchKernelGGL(conv3x3_halo_kernel, dim3((unsigned)grid), dim3(NWAVE * 64), LDS_BYTES, (hipStream_t)stream, a);
    SDV_CHECK_LAUNCH("sdv_conv3x3_halo_bf16");
    return SDV_OK;
}
