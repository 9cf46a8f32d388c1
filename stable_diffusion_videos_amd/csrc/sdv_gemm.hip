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
//     same way: BK-wide K tiles go HBM -> LDS with `buffer_load_dwordx4 ... lds` (LDS-DMA, no VGPR
//     round trip), double buffered, ONE barrier per K tile, next tile in flight during the MFMAs.
//   * Addressing = buffer descriptor anchored at the workgroup's first row + a 32-bit lane offset + the K
//     position in the SCALAR offset: no per-tile VALU address math.  conv padding pixels and rows beyond
//     M / N use an offset past num_records, so the descriptor range check writes zeros for them.
//   * LDS image is lane-linear (DMA constraint), so the bank-conflict swizzle is applied on the per-lane
//     SOURCE offset and again on the ds_read_b128 (conflict-free fragment reads for 128-B and 64-B rows).
//   * v_mfma_f32_32x32x16_bf16 with the WEIGHT rows as the A operand and the ACTIVATION rows as the
//     B operand: lane l owns output row m = l & 31 and, per accumulator quad, 4 consecutive channels.
//   * conv3x3: the K loop walks (tap, source, channel-tile) segments; only segment changes touch VALU.
//   * Epilogue: fp32 sub-tiles are parked in LDS in 64-column passes and written as whole 128-byte row
//     segments (16-B coalesced stores, residual loads prefetched before the pass).
//   * Tiles: 8-wave 256x320 / 256x256 / 128x320 (UNet widths are multiples of 320), one workgroup per CU; small
//     4-wave tiles for the low-resolution levels; chosen per launch by a measured cost model.  XCD-aware order.
#include <type_traits>

#include "sdv_common.h"

// (The A/B knobs of rounds 2-4 - W fragments ahead, GELU form, 32-column residual passes, panel walk, channel-major K order, the
//  LDS-ring tiles 12 / 13, the transposed 320 x 256 tile 14 with the column-side LayerNorm fold - lived here as #if branches and
//  extra template instances.  Their measurements are under profiles/; the source that can rebuild any of them is the round-4
//  tree (git 2b6591d) through tools/ubench/build_variant.py.  This file keeps what the cost model can pick.)
constexpr int kRotAhead = 2;   // W fragments in flight ahead of the MFMAs in the bf16 K loop of the big tiles (profiles/round3_rot_w_fragments_ab.txt)

namespace {

SDV_DEVICE float geglu_gate_f(float x) { return gelu_erf_fast_f(x); }   // (rearranged exact-erf GELU: profiles/round4_geglu_gelu_ab.txt)

// waves per SIMD the register allocator must leave room for: the 32-wide-K tiles are meant to run two workgroups
// per CU (16 waves -> 4 per SIMD -> <= 128 VGPRs)
// NST = K-slab buffers in LDS: always 2 (double buffer, one `vmcnt(0)` + barrier per K slab; the 4-slot ring form of rounds 2-3
// measured 2-14 % slower on every transformer shape - profiles/round3_ring_ab_nimg256.txt - and is gone; the parameter stays so that
// kernel names in earlier profiles still read the same).
// FEAT compiles ONE of the LayerNorm-fold features into the epilogue (dense GEMMs of the transformer blocks only - each costs
// registers the 256-row tiles do not have to spare): 1 = ln_side 1 consumer, 3 = stats_out producer,
// 4 = gn_out producer (GroupNorm statistics of the stored tile; the 4-wave tiles carry that inside FEAT 0 - they have the registers).
template <int WM, int WN, int TM, int TN, int BK, bool CONV, int NST = 2, int FEAT = 0>
__global__ __launch_bounds__(WM * WN * 64, (((BK == 32 && NST == 2) ? 2 : 1) * WM * WN / 4)) void igemm_kernel(const sdv_gemm_args p) {
    constexpr int NWV = WM * WN;            // waves per workgroup (4 or 8)
    // The extra activations (epi 3 LeakyReLU, 4 quick_gelu, 5 GELU: RRDBNet / CLIP text encoder) are compiled into the
    // 4-wave tiles only: the 8-wave 256x320 tile has no VGPRs to spare (adding them to its epilogue spilled 768 B of
    // scratch and cost 5x on every UNet GEMM), and those networks' shapes use the small tiles anyway.
    constexpr bool XACT = NWV == 4;
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    // FEAT 8: fp8 (OCP e4m3) operands - X and W are bytes, the MFMA is v_mfma_f32_32x32x16_fp8_fp8 (bf16 rate, half the
    // LDS-DMA and LDS-read bytes per MFMA), accumulation fp32, output bf16; the per-tensor dequantisation scale is `alpha`.
    // FEAT 9: the same operands on the block-scaled form v_mfma_scale_f32_32x32x64_f8f6f4 with every block scale = 1 (E8M0 127):
    // 64 K values per instruction at TWICE the bf16 / plain-fp8 MFMA rate.  A K slab is a PAIR of the fp8 tile's 64-byte-row
    // images (one per k-step; the second one of the K loop's last slab may be "dead": fetched through a zero-length buffer
    // descriptor, i.e. zeros that cost no memory traffic), so LDS-DMA pieces, fragment reads and barriers per MFMA cycle are
    // those of the bf16 tile, at twice the FLOPs.
    constexpr bool MX = FEAT == 9;
    constexpr int ES = (FEAT == 8 || MX) ? 1 : 2;   // operand element size in bytes
    constexpr int ROWB = BK * ES;           // bytes per LDS row (128 or 64)
    constexpr int CPRW = ROWB / 16;         // 16-byte chunks per row (8 or 4)
    constexpr int RPI = 1024 / ROWB;        // rows one wave-instruction (1 KiB) moves (8 or 16)
    constexpr int HALF_BYTES = (BM + BN) * ROWB;          // one 64-wide K image of both panels
    constexpr int TILE_BYTES = (MX ? 2 : 1) * HALF_BYTES;
    constexpr int GX = BM / RPI, GW = BN / RPI;                    // 1-KiB row groups of the X / W panels
    constexpr int NX = (GX + NWV - 1) / NWV, NW = (GW + NWV - 1) / NWV;
    constexpr int KSTEPS = BK / 16;
    static_assert(BM % RPI == 0 && BN % RPI == 0, "panels must be whole 1-KiB groups");
    static_assert(BK == 64 && NST == 2, "64-wide K slabs, double buffered");
    static_assert(FEAT != 2, "(FEAT 2 was the column-side LayerNorm fold of the transposed V^T projections: gone with them)");
    static_assert(FEAT != 8 || (BK == 64 && NST == 2), "fp8 tiles: 64 elements (64 bytes) per K tile, double buffered");
    static_assert(!MX || (BK == 64 && NST == 2), "MX fp8 tiles: two 64-element images per K tile, double buffered");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;
    const int wn = wave % WN;

    // ---- tile walk ------------------------------------------------------------------------------------------------
    // The 8-wave double-buffered tiles (one workgroup per CU) are PERSISTENT: the host launches min(tiles, CUs) workgroups and
    // workgroup b walks tiles b, b + grid, b + 2 grid, ...  A tile's first K slab is fetched (LDS-DMA) during the LAST K slab
    // of the tile before it and stays in flight across that tile's epilogue, and the epilogue's stores drain behind the next
    // tile's first MFMAs: with one workgroup per CU nothing else can cover those two latencies (they were 8-9 us of a 19 us
    // K = 320 tile, profiles/round2_gemm_overhead.txt), and the CUs stop moving through load / compute / store phases in
    // lockstep.  Every other tile runs this loop exactly once (grid = tiles).
    constexpr bool PERSIST = NWV == 8;
    // LDS: NST K-slab buffers of SLOT bytes, then the epilogue's column vectors (3 x BN floats) and row accumulators.  The
    // epilogue's staging slabs (SLAB bytes per wave) alias the K-slab buffers - in a persistent workgroup the ONE slot the
    // tile's last K slab was read from, the other slot already receives the next tile.
    constexpr int SLAB = 32 * 144;   // 32 rows x (128 + 16 pad) bytes
    constexpr int STG = 2 * NWV * SLAB;
    constexpr int SLOT = PERSIST ? (TILE_BYTES > STG ? TILE_BYTES : STG) : TILE_BYTES;
    // two slabs per wave (pass p+1 is parked while pass p is read back) wherever the K-slab buffers have the room
    constexpr bool DBL = (PERSIST ? SLOT : NST * SLOT) >= 2 * NWV * SLAB;
    static_assert(NST * SLOT >= NWV * SLAB, "the staging slabs must fit the K-slab buffers");
    const int tiles_m = (p.M + BM - 1) / BM;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int nblk = tiles_m * tiles_n;
    const int total = nblk * (p.batch > 0 ? p.batch : 1);
    const int K = p.K;
    // SPLIT-K (sdv_hip.h "split_k"; the plain variants of the 4-wave tiles only, so that no 8-wave kernel sees a line of it): with few
    // frames per call the low-resolution layers have M = 128 ... 2048 rows against K = 11 520 ... 23 040 - a handful of tiles on 256
    // CUs, each walking hundreds of K slabs.  S workgroups per tile then take S contiguous K-slab ranges (blockIdx = split * tiles +
    // tile) and leave their fp32 accumulators in a workspace [S][M][N]; splitk_reduce_kernel adds the S partials in split order
    // (deterministic), applies alpha / bias / residual and rounds once.
    constexpr bool SPLITK = NWV == 4 && FEAT == 0;
    int ks_ = 0, seek_r = 0;   // this workgroup's split; K slabs to skip inside the first segment it stages

    // ---- operand addressing: buffer descriptors + 32-bit lane offsets + scalar K offset -----------------
    constexpr unsigned kOOB = 0x80000000u;
    constexpr int kRecords = 0x7ffffff0;
    // Everything derived from the lane id is re-derived per tile from an opaque copy (lane_setup): the persistent walk must
    // not keep ~30 addressing registers alive across the epilogue, where the 160 accumulators + the staging state already
    // fill the 256-VGPR budget (they spilled 100-300 B of scratch per lane when simply hoisted out of the tile loop).
    int rg, pc;    // row inside the group one wave-instruction moves / physical 16-B chunk inside the LDS row
    int l31, lhi;
    // Fragment reads: every 32-row MFMA tile of either panel starts on a multiple of 32 rows, so the row part of this lane's
    // read address - l31 * ROWB - and its swizzle term - swz(l31) - are the SAME for all of them: one VGPR per k-step
    // (frag_off[ks]) + a wave-uniform tile base, which ends up in the ds_read's immediate offset.  (An array of per-tile
    // offsets + swizzles cost 2 x (TM + TN) VGPRs and 2-3 VALU per read.)
    int frag_off[KSTEPS];
    const int xrow0 = wm * TM * 32, wrow0 = BM + wn * TN * 32;   // first panel row of this wave's fragments (wave-uniform)
    auto swz = [](int r) { return ROWB == 128 ? ((r >> 1) & 7) : ((r >> 2) & 3); };

    // addressing state of ONE tile (during a tile's last K slab it is re-pointed at the next tile)
    int a_m0 = 0, a_n0 = 0, a_bn = 0, a_bz = 0;
    __amdgpu_buffer_rsrc_t rs_x1, rs_x2, rs_w;
    int xr_[NX];               // dense: row inside the tile (or -1 beyond M); conv: pixel index of the image origin - pix0
    int xay[NX], xax[NX];      // conv: anchor coordinates (oy*stride, ox*stride) or (oy, ox) for upsample
    int xlc[NX];               // byte offset of this lane's logical chunk inside a K tile
    unsigned wvo[NW];          // lane byte offsets of the W rows relative to rs_w
    unsigned xvo[NX];
    int tap = 0, srcsel = 0, seg_left = 0;
    int kx = 0;   // scalar byte offset inside the current X source
    int halves_left = 0;   // MX: 64-wide K images of the current tile not yet staged (0 -> the slab's second image is dead)
    int kwb = 0;  // scalar byte offset along the W rows (all taps and sources are contiguous in K)

    auto setup_tile = [&](int vb) {
        int bm, bn, bz;
        {
        bz = vb / nblk;           // batch index (mode 4: the phase)
        const int lb = vb - bz * nblk;
        if constexpr (SPLITK) {
            if (p.split_k > 1) {      // (not batched: the block index above the tiles is the split)
                ks_ = bz;
                bz = 0;
            }
        }
        // XCD-aware tile order: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs; remap
        // (bijectively) so that each XCD - each private L2 - works on one contiguous run of tiles: neighbouring
        // M tiles of a conv share their halo rows, and all tiles of a run share the same W panel.
        const int xq = nblk >> 3, xr = nblk & 7;
        const int xcd = lb & 7, xi = lb >> 3;
        const int tile_id = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + xi;
        // Raster order inside an XCD's run: N is cut into strips of `gn` tiles and a strip is walked N-fastest, so the ~32
        // workgroups an XCD runs at a time form a (32/gn) x gn block of output tiles - they stream 32/gn activation panels and
        // gn weight panels through the XCD's 4 MiB L2 instead of 32 + 1 (M-fastest).  Measured with rocprofv3 FETCH_SIZE: the
        // M-fastest order re-fetched every activation panel once per N tile from the fabric (profiles/round2_*).
        const int gn = p.tile > 0 ? (p.tile < tiles_n ? p.tile : tiles_n) : 1;
        const int strip_sz = tiles_m * gn;
        const int full = tiles_n / gn;
        int strip = tile_id / strip_sz;
        int gcols = gn;
        if (strip >= full) {
            strip = full;
            gcols = tiles_n - full * gn;
        }
        const int rr = tile_id - strip * strip_sz;
        bm = rr / gcols;
        bn = strip * gn + (rr - bm * gcols);
        }
        const int m0 = bm * BM;
        const int n0 = bn * BN;
        a_m0 = m0;
        a_n0 = n0;
        a_bn = bn;
        a_bz = bz;
        long long xbase1, xbase2;     // element offsets of the window start in X / X2 (wave-uniform)
        int pix0 = 0;
        if constexpr (CONV) {
            pix0 = (m0 / (p.Hout * p.Wout)) * p.Hin * p.Win;
            xbase1 = (long long)pix0 * p.ldx;
            xbase2 = (long long)pix0 * p.ldx2;
        } else {
            xbase1 = bz * p.sX + (long long)m0 * p.ldx;
            xbase2 = (long long)m0 * p.ldx2;
        }
        rs_x1 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.X + xbase1 * ES), 0, kRecords, 0x00020000);
        rs_x2 = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.X2 + xbase2 * ES), 0, kRecords, 0x00020000);
        rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.W + (bz * p.sW + (long long)n0 * p.ldw) * ES), 0, kRecords,
                                                 0x00020000);
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int r = (wave + NWV * i) * RPI + rg;
            const int m = m0 + r;
            xlc[i] = (pc ^ swz(r)) * 16;
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
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int rw = (wave + NWV * i) * RPI + rg;  // row inside the W panel
            wvo[i] = (n0 + rw < p.N) ? (unsigned)(rw * p.ldw * ES + (pc ^ swz(BM + rw)) * 16) : kOOB;
        }
        tap = 0;
        srcsel = 0;
        seg_left = 0;
        kx = 0;
        kwb = 0;
        halves_left = (K / BK) * (CONV ? (p.mode == 4 ? 4 : 9) : 1);
        if constexpr (SPLITK) {
            if (p.split_k > 1) {
                // first K slab of this split: which (tap, source) segment it lies in and how far into it
                const int all = (K / BK) * (CONV ? 9 : 1);
                const int k0 = (int)((long long)ks_ * all / p.split_k);
                const int seg1 = p.C1 / BK, per_tap = K / BK;
                tap = k0 / per_tap;
                int r = k0 - tap * per_tap;
                srcsel = r >= seg1 ? 1 : 0;
                seek_r = srcsel ? r - seg1 : r;
                kwb = k0 * ROWB;
            }
        }
    };

    // The K loop walks segments = (tap, source) pairs; inside a segment only the scalar offset advances.
    const bool two_src = p.C1 < K;
    const int up_shift = p.mode == 3 ? 1 : 0;
    const int ext_y = p.mode == 3 ? p.Hout : p.Hin;  // extent the tap offset is applied in
    const int ext_x = p.mode == 3 ? p.Wout : p.Win;

    auto new_segment = [&]() {
        const int ld2 = ES * (srcsel ? p.ldx2 : p.ldx);
        if constexpr (CONV) {
            // mode 4 (one phase (py, px) = blockIdx.z of a nearest-2x-upsample + conv3x3, see sdv_hip.h): 2 x 2 taps on the
            // low-resolution grid, rows {y-1+py, y+py}, columns {x-1+px, x+px}
            const int dy = p.mode == 4 ? (tap >> 1) - 1 + (a_bz >> 1) : tap / 3 - 1;
            const int dx = p.mode == 4 ? (tap & 1) - 1 + (a_bz & 1) : tap - (tap / 3) * 3 - 1;
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
        seg_left = (srcsel ? K - p.C1 : p.C1) / BK;
    };

    // `issue` false: only the bookkeeping of a slab that is already in LDS (the prefetched first slab of a persistent tile)
    auto stage = [&](int buf, bool issue = true) {
        char* base = smem + buf * SLOT;
#pragma unroll
        for (int hf = 0; hf < (MX ? 2 : 1); ++hf, base += HALF_BYTES) {
        // MX: an odd number of 64-wide K images leaves the last slab's SECOND image dead - every piece of it gets an offset
        // past the descriptor's range (the range check answers with zeros, nothing is fetched), no bookkeeping advances
        const bool dead = MX && hf == 1 && halves_left == 0;
        const unsigned dmask = dead ? kOOB : 0u;
        if (!dead && seg_left == 0) {
            new_segment();
            if constexpr (SPLITK) {          // (a split that starts inside a segment: seek_r is 0 everywhere else)
                kx += seek_r * ROWB;
                seg_left -= seek_r;
                seek_r = 0;
            }
        }
        const __amdgpu_buffer_rsrc_t rs_x = srcsel ? rs_x2 : rs_x1;
        if (issue) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int g = wave + NWV * i;
            if (GX % NWV == 0 || g < GX)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(base + g * 1024), 16,
                                                         (int)(MX && hf == 1 ? xvo[i] | dmask : xvo[i]), kx, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int g = wave + NWV * i;
            if (GW % NWV == 0 || g < GW)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (__attribute__((address_space(3))) void*)(base + (GX + g) * 1024),
                                                         16, (int)(MX && hf == 1 ? wvo[i] | dmask : wvo[i]), kwb, 0, 0);
        }
        }
        if (!dead) {
        --halves_left;
        kx += ROWB;
        kwb += ROWB;
        if (--seg_left == 0) {
            if (two_src && srcsel == 0) {
                srcsel = 1;
            } else {
                srcsel = 0;
                ++tap;
            }
        }
        }
        }
    };

    f32x16_t acc[TN][TM];

    auto lane_setup = [&]() {
        int lk = lane;
        if constexpr (PERSIST) asm volatile("" : "+v"(lk));
        rg = lk / CPRW;
        pc = lk % CPRW;
        l31 = lk & 31;
        lhi = lk >> 5;
        static_assert(BM % 32 == 0, "fragment tiles start on multiples of 32 rows (the swizzle term depends on l31 only)");
        // logical chunk of k-step ks: bf16 tiles 2 ks + lhi (16 bytes = 8 elements); fp8 tiles read chunk 2 j + lhi for the
        // k-step PAIR j (see compute) - the same expression with ks = j
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) frag_off[ks] = l31 * ROWB + (((ks * 2 + lhi) ^ swz(l31)) << 4);
    };

    // Software-pipelined K tile: the fragments of k-step ks+1 are read into a second register set while the
    // MFMAs of k-step ks issue (1 ds_read_b128 slotted behind each MFMA).
    constexpr bool kPipeFrags = TM * TN >= 8;   // big tiles: 1 workgroup / CU, hide LDS latency inside the wave
    auto compute = [&](int buf) {
        const char* base = smem + buf * SLOT;
        if constexpr (MX) {
            // K image s of the slab feeds ONE 64-deep MFMA per (n, m) tile: lane (l31, lhi) supplies the 16-byte chunks lhi and
            // 2 + lhi of its row (the two fragment reads of the fp8 tile).  Both operands walk K in that same order, so the
            // contraction is unchanged whatever order the instruction assigns to a lane's 32 bytes.
            typedef int __attribute__((ext_vector_type(8))) i32x8_t;
            constexpr int kOne = 0x7f7f7f7f;   // E8M0 127 = 2^0 in every scale byte
            auto frag = [&](const char* hb, int row0) __attribute__((always_inline)) -> i32x8_t {
                const u32x4_t lo = *(const u32x4_t*)(hb + row0 * ROWB + frag_off[0]);
                const u32x4_t hi = *(const u32x4_t*)(hb + row0 * ROWB + frag_off[1]);
                return i32x8_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
            };
            // Low-pressure order (the 256 x 320 tile has ~90 VGPRs besides its 160 accumulators, and an MX operand is an aligned
            // 8-register tuple): the X fragments of a K image (TM x 8 registers, the second image's are read AH steps before
            // their first use), ONE W fragment in use and AH more in flight behind the TM MFMAs of each step.
            constexpr int AH = 2;   // W fragments in flight ahead of the one the MFMAs use (1 / 2 / 3 measured equal on MI355X)
            i32x8_t xa[2][TM];
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) xa[0][mt] = frag(base, xrow0 + mt * 32);
            auto wfrag = [&](int st) __attribute__((always_inline)) { return frag(base + (st / TN) * HALF_BYTES, wrow0 + (st % TN) * 32); };
            i32x8_t wq[AH + 1];
#pragma unroll
            for (int a = 0; a < AH; ++a) wq[a] = wfrag(a);
#pragma unroll
            for (int st = 0; st < 2 * TN; ++st) {
                const int s = st / TN, nt = st % TN;
                if (st + AH < 2 * TN) wq[(st + AH) % (AH + 1)] = wfrag(st + AH);
                if (st == (TN - 1 - AH >= 0 ? TN - 1 - AH : 0)) {      // the second image's X fragments, AH steps before their first use
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt) xa[1][mt] = frag(base + HALF_BYTES, xrow0 + mt * 32);
                }
#pragma unroll
                for (int mt = 0; mt < TM; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wq[st % (AH + 1)], xa[s][mt], acc[nt][mt], 0, 0, 0, kOne, 0, kOne);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
        if constexpr (ES == 1) {
            // fp8: a K tile is 64 bytes per row = four 16-byte chunks.  One ds_read_b128 of chunk 2j + lhi yields two
            // 8-byte MFMA operands, used for k-steps 2j and 2j+1 (both operands walk K in the same permuted order, so the
            // contraction is unchanged): 2 fragment reads per row set and K tile feed 4 k-steps.
            typedef __attribute__((ext_vector_type(2))) long long i64x2_t;
            i64x2_t xq[2][TM], wq[2][TN];
            auto load_q = [&](int j, i64x2_t* xd, i64x2_t* wd) {
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) xd[mt] = *(const i64x2_t*)(base + (xrow0 + mt * 32) * ROWB + frag_off[j]);
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) wd[nt] = *(const i64x2_t*)(base + (wrow0 + nt * 32) * ROWB + frag_off[j]);
            };
            load_q(0, xq[0], wq[0]);
            load_q(1, xq[1], wq[1]);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
                        for (int mt = 0; mt < TM; ++mt)
                            acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(wq[j][nt][h], xq[j][mt][h], acc[nt][mt], 0, 0, 0);
            return;
        }
        if constexpr (!kPipeFrags) {
            // small wave tiles run 2+ workgroups per CU: other waves cover the LDS latency, registers matter more
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                bf16x8_t xs[TM], ws[TN];
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) xs[mt] = *(const bf16x8_t*)(base + (xrow0 + mt * 32) * ROWB + frag_off[ks]);
#pragma unroll
                for (int nt = 0; nt < TN; ++nt) ws[nt] = *(const bf16x8_t*)(base + (wrow0 + nt * 32) * ROWB + frag_off[ks]);
#pragma unroll
                for (int nt = 0; nt < TN; ++nt)
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ws[nt], xs[mt], acc[nt][mt], 0, 0, 0);
            }
            return;
        }
        {
            // Rotating W fragments (the MX order above, for the bf16 tiles): a step = ONE W fragment against the TM X fragments of
            // its k-step.  Live fragment registers: X of this k-step and - from AH steps before its first use - of the next one
            // (2 x TM x 4), one W fragment in use and AH in flight behind the TM MFMAs of each step ((AH + 1) x 4): 28 registers
            // at AH = 2 where two whole fragment sets took 56.  With those the compiler had run out of registers in the last
            // k-steps of a slab and fallen back to read -> s_waitcnt lgkmcnt(0) -> 2 MFMAs per W fragment, the LDS latency of every
            // read exposed (ISA of round 3); here the issue order is pinned by the sched_barriers and every read has AH x TM MFMAs
            // (64 matrix-pipe cycles each pair) to land.  Same accumulation order per output tile: bit-identical results.
            constexpr int AH = kRotAhead, STEPS = KSTEPS * TN;
            bf16x8_t xa[2][TM], wq[AH + 1];
            auto wfrag = [&](int st) __attribute__((always_inline)) {
                return *(const bf16x8_t*)(base + (wrow0 + (st % TN) * 32) * ROWB + frag_off[st / TN]);
            };
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) xa[0][mt] = *(const bf16x8_t*)(base + (xrow0 + mt * 32) * ROWB + frag_off[0]);
#pragma unroll
            for (int a = 0; a < AH; ++a) wq[a] = wfrag(a);
#pragma unroll
            for (int st = 0; st < STEPS; ++st) {
                const int ks = st / TN, nt = st % TN;
                if (st + AH < STEPS) wq[(st + AH) % (AH + 1)] = wfrag(st + AH);
                if (nt == (TN - 1 - AH >= 0 ? TN - 1 - AH : 0) && ks + 1 < KSTEPS) {
#pragma unroll
                    for (int mt = 0; mt < TM; ++mt)
                        xa[(ks + 1) & 1][mt] = *(const bf16x8_t*)(base + (xrow0 + mt * 32) * ROWB + frag_off[ks + 1]);
                }
#pragma unroll
                for (int mt = 0; mt < TM; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wq[st % (AH + 1)], xa[ks & 1][mt], acc[nt][mt], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            return;
        }
    };

    const int ntaps = CONV ? (p.mode == 4 ? 4 : 9) : 1;
    int nkt = MX ? ((K / BK) * ntaps + 1) / 2 : (K / BK) * ntaps;   // (MX: two 64-wide K images per slab; split-K: this split's share, below)
    // tile walk state: `vb` = the tile being computed, `slot0` = the LDS slot its first K slab is in, `landed` = that slab
    // was fetched (and waited for) during the tile before
    int vb = blockIdx.x;
    int slot0 = 0;
    bool landed = false;
    bool has_next = false;
    auto kloop = [&]() {
    {
    // ---- main loop: one barrier per K tile, tile t+1 in flight (LDS-DMA) while tile t computes ----
    // The barriers here are RAW (own LDS reads retired + s_barrier): __syncthreads() is lowered to s_waitcnt vmcnt(0) +
    // s_barrier, and in a persistent workgroup that would make the first slab of every tile wait for the PREVIOUS tile's
    // stores - every load this loop depends on has its own explicit wait below.
    auto lds_barrier = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    stage(slot0, !landed);
    // (a prefetched first slab was waited for before the previous tile's epilogue barrier; waiting again here would
    //  also wait for that epilogue's STORES, which may drain behind this tile's first MFMAs instead)
    if (!(PERSIST && landed)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int kt = 0; kt + 1 < nkt; ++kt) {
        lds_barrier();
        stage((slot0 + kt + 1) & 1);
        compute((slot0 + kt) & 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // last K slab (kept out of the loop: the re-pointing below must not add register pressure to the steady state)
    lds_barrier();
    if (PERSIST && has_next) {
        // every wave is past the MFMAs of slab nkt-2, so its slot is free - point the addressing at the next tile and
        // start its first slab
        setup_tile(vb + (int)gridDim.x);
        stage((slot0 + nkt) & 1);
    }
    compute((slot0 + nkt - 1) & 1);
    }
    };

    // ---- epilogue ------------------------------------------------------------------------------
    int e_m0, e_n0, e_bn, e_bzi;
    auto epilogue = [&]() {
    const int m0 = e_m0, n0 = e_n0, bn = e_bn;
    const long long bz = e_bzi;
    if constexpr (SPLITK) {
        if (p.split_k > 1) {          // this split's partial sums, unscaled fp32, straight from the accumulators (N % 4 == 0: host)
            float* __restrict__ ws = p.out_f32 + (long long)ks_ * p.M * p.N;
#pragma unroll
            for (int nt = 0; nt < TN; ++nt)
#pragma unroll
                for (int mt = 0; mt < TM; ++mt) {
                    const int m = m0 + wm * TM * 32 + mt * 32 + l31;
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int nb = n0 + wn * TN * 32 + nt * 32 + 8 * g4 + 4 * lhi;
                        if (m < p.M && nb < p.N)
                            *(float4*)(ws + (long long)m * p.N + nb) = make_float4(acc[nt][mt][4 * g4], acc[nt][mt][4 * g4 + 1],
                                                                                  acc[nt][mt][4 * g4 + 2], acc[nt][mt][4 * g4 + 3]);
                    }
                }
            return;
        }
    }
    const float alpha = p.alpha;
    const int acols = p.alpha_cols > 0 ? p.alpha_cols : 0x7fffffff;
    const float* bias = p.bias;
    if (bias && p.step_ptr) bias += (long long)(*p.step_ptr) * p.bias_step_stride;
    uint16_t* __restrict__ C = p.C + bz * p.sC;
    const uint16_t* __restrict__ R = p.R ? p.R + bz * p.sR : nullptr;
    // output row of GEMM row m: m itself, except for mode 4 where low-resolution pixel (img, y, x) of phase (py, px) lands on
    // pixel (2y+py, 2x+px) of the 2x image (divisions by host-made magic numbers: exact for m < 2^31)
    auto out_row = [&](int m) -> long long {
        if constexpr (!CONV) return m;
        if (p.mode != 4) return m;
        const unsigned um = (unsigned)m;
        const unsigned img = p.div_hw_mul ? __umulhi(um, p.div_hw_mul) >> p.div_hw_shr : um >> p.div_hw_shr;
        const unsigned rem = um - img * (unsigned)(p.Hout * p.Wout);
        const unsigned y = p.div_w_mul ? __umulhi(rem, p.div_w_mul) >> p.div_w_shr : rem >> p.div_w_shr;
        const unsigned x = rem - y * (unsigned)p.Wout;
        return ((long long)img * (2 * p.Hout) + (2 * y + (unsigned)(bz >> 1))) * (2 * p.Wout) + (2 * x + (unsigned)(bz & 1));
    };
    const bool geglu = p.epi == 1;
    const int wcol0 = n0 + wn * TN * 32;  // first (permuted, for GEGLU) weight row of this wave

    // ---- the normal case: row-major 16-byte stores through a per-wave LDS slab ----
    {
        const int ncols_out = geglu ? (p.N >> 1) : p.N;
        const bool aligned = ((p.ldc & 7) == 0) && ((ncols_out & 7) == 0) && (!R || (p.ldr & 7) == 0) &&
                             ((((uintptr_t)C | (uintptr_t)R | (uintptr_t)bias) & 15) == 0) &&
                             (((p.sC | p.sR) & 7) == 0);
        // (the 8-wave tiles compile the three common store sequences only: SiLU - the one extra activation they offer, used
        //  by no network here - takes the register-direct path below; the 4-wave tiles carry the generic sequence as well)
        if (aligned && !p.out_mode && (XACT || p.epi == 0 || (geglu && !R))) {
            // Staged stores: the MFMA leaves lane (l31, lhi) with row l31 and, per accumulator quad, 4 consecutive columns -
            // stored as they are, every 16-byte piece of a wave's store instruction lies in a different row, and the texture
            // addresser takes them one by one (measured: 9.5k clocks for the 160 stores of a 256 x 320 tile, the whole K loop
            // of a K = 320 tile is 13k MFMA clocks).  So each wave parks a PASS - 32 rows x 64 bf16 (or 32 fp32 when a
            // residual has to be added before the single rounding) - in its private 4.5 KB LDS slab and reads it back
            // row-major: 8 (4) adjacent lanes then hold one row's 128 (64) contiguous bytes.  The passes are software
            // pipelined - values of pass p+1 are converted and written while the rows of pass p are still in flight from LDS
            // (two slabs per wave where LDS allows, fenced hand-overs) - the code is straight-line per
            // case (no runtime flags in the common ones), and rows beyond M / columns beyond N are dropped by the buffer
            // descriptor's range check instead of exec-mask branches.
            // (an opaque copy of the lane id: everything derived from it below is tile-invariant, and hoisted out of the
            //  persistent tile loop it would sit in ~17 registers across the K loop - they spilled)
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));
            // (LDS slab accesses go through may_alias types: the slab is written as 8-byte / 16-byte pieces in one layout and
            //  read back as 16-byte pieces in another, and type-based alias analysis must not reorder the two)
            typedef unsigned int __attribute__((ext_vector_type(4), may_alias)) slab_u4;
            typedef unsigned int __attribute__((ext_vector_type(2), may_alias)) slab_u2;
            char* const stg_region = !PERSIST ? smem : smem + ((slot0 + nkt - 1) & 1) * SLOT;
            char* const slab0 = stg_region + wave * (DBL ? 2 : 1) * SLAB;
            // Per-column epilogue vectors of this tile's BN columns, staged in LDS ONCE per tile (their own region behind the
            // K-slab buffers): bias, the LayerNorm-fold row sums s (ln_side 1) or the per-column (mean, rstd) (ln_side 2).
            float* vbias = (float*)(smem + NST * SLOT);
            float* vaux = vbias + BN;            // [BN] (ln_side 1: the row sums s of the gamma-scaled weights)
            float* rowacc = vbias + 3 * BN + wave * 64;   // FEAT 3: (sum, sumsq) of this wave's 32 rows of the current m-tile
            const int mfirst = m0 < p.M ? m0 : p.M - 1;
            const long long orow0 = out_row(mfirst);
            const __amdgpu_buffer_rsrc_t rs_c = __builtin_amdgcn_make_buffer_rsrc((void*)(C + orow0 * p.ldc), 0, kRecords, 0x00020000);
            const __amdgpu_buffer_rsrc_t rs_r =
                __builtin_amdgcn_make_buffer_rsrc((void*)(R ? R + orow0 * p.ldr : C), 0, kRecords, 0x00020000);
            // per-ROW operands of this lane's TM accumulator rows, fetched before the barrier (their latency hides behind it):
            // LayerNorm fold (ln_side 1): (mean, rstd) of the row; bias_mode 2: the row's bias
            float ln_row[TM][2], bm_[TM];
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                ln_row[mt][0] = 0.f;
                ln_row[mt][1] = 1.f;
                const int mrow = m0 + wm * TM * 32 + mt * 32 + l31;
                const bool ok = mrow < p.M;
                bm_[mt] = (bias && p.bias_mode == 2 && ok) ? bias[mrow] : 0.f;
                if constexpr (FEAT == 1) {
                    if (ok) {
                        const float2 st = *(const float2*)(p.ln_stats + 2 * (bz * p.M + mrow));
                        ln_row[mt][0] = st.x;
                        ln_row[mt][1] = st.y;
                    }
                }
            }
            {
                // (one element per thread - BN <= 512 - so that the global loads are all issued BEFORE the wait below and
                //  their latency overlaps the tail of the prefetch instead of following it)
                constexpr int NV = (BN + NWV * 64 - 1) / (NWV * 64);   // elements per thread (1; 2 for the 4-wave 320-column tile)
                constexpr int lnsd = FEAT == 1 ? 1 : 0;
                const bool col_bias = bias && p.bias_mode == 1;
                float vb_[NV], vs_[NV];
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int i = tid + j * NWV * 64, n = n0 + i;
                    const bool live = i < BN && n < p.N;
                    vb_[j] = vs_[j] = 0.f;
                    if (col_bias && live) vb_[j] = bias[n];
                    if constexpr (lnsd == 1) {
                        if (live) vs_[j] = p.ln_s[n];
                    }
                }
                // the next tile's first slab (issued during the last K slab) has had a whole slab of MFMAs to land: wait for it
                // HERE, so that the next K loop does not have to wait on anything this epilogue is about to store
                if (PERSIST && has_next) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int j = 0; j < NV; ++j) {
                    const int i = tid + j * NWV * 64;
                    if (i < BN) {
                        vbias[i] = vb_[j];
                        if constexpr (lnsd == 1) vaux[i] = vs_[j];
                    }
                }
            }
            // (the epilogue barrier - the vectors are staged AND every wave has left the K loop, whose buffers the slabs alias -
            //  sits inside run(), behind the first residual loads: their latency overlaps the barrier wait)
            // LayerNorm folded into this GEMM (sdv_hip.h "ln_side"): the weights were pre-multiplied by gamma, so
            //   LN(x) W^T = rstd * (x (gamma o W)^T - mean * s) + (W beta + b),  s = row sums of gamma o W
            // (mean, rstd) belong to the output ROW (this lane's m), s to the output column.  (A column-side form - statistics per
            //  output column, for a transposed V^T projection - existed in rounds 2-4; the fused QKV projection + row-major V in the
            //  attention kernel replaced it, DESIGN.md "the round-4 open item".)
            constexpr int ln_side = FEAT == 1 ? 1 : 0;

            // the 4 values of accumulator quad q of tile (nt, mt) with scale / LayerNorm fold / bias applied (GEGLU: W rows
            // are interleaved [16 value | 16 gate] per 32-row MFMA tile, so quads g and g+2 of a lane hold the value and the
            // gate of the SAME 4 channels -> quad g in {0, 1} yields 4 of the n-tile's 16 output columns)
            auto quad_vals = [&](bool gg, int nt, int mt, int q, int z, float* v) {   // z: an opaque 0 (see park)
                if (gg) {
                    const float ln_mu = ln_row[mt][0], ar = alpha * ln_row[mt][1];
                    const int nv = wcol0 + nt * 32 + 8 * q + 4 * lhi - n0 + z;
                    const float4 bv = *(const float4*)(vbias + nv), bg = *(const float4*)(vbias + nv + 16);
                    float4 sv = make_float4(0.f, 0.f, 0.f, 0.f), sg = sv;
                    if constexpr (ln_side == 1) {
                        sv = *(const float4*)(vaux + nv);
                        sg = *(const float4*)(vaux + nv + 16);
                    }
                    const float bvv[4] = {bv.x, bv.y, bv.z, bv.w}, bgv[4] = {bg.x, bg.y, bg.z, bg.w};
                    const float svv[4] = {sv.x, sv.y, sv.z, sv.w}, sgv[4] = {sg.x, sg.y, sg.z, sg.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] = ((acc[nt][mt][4 * q + e] - ln_mu * svv[e]) * ar + bvv[e]) *
                               geglu_gate_f((acc[nt][mt][4 * (q + 2) + e] - ln_mu * sgv[e]) * ar + bgv[e]);
                    return;
                }
                const int nb = wcol0 + nt * 32 + 8 * q + 4 * lhi;
                const float al = nb < acols ? alpha : 1.f;    // alpha_cols: scale only the leading output columns
                const float4 bq = *(const float4*)(vbias + (nb - n0) + z);
                const float brow = bm_[mt];
                const float bvv[4] = {bq.x + brow, bq.y + brow, bq.z + brow, bq.w + brow};
                if constexpr (ln_side == 1) {
                    const float4 s4 = *(const float4*)(vaux + (nb - n0) + z);
                    const float sv4[4] = {s4.x, s4.y, s4.z, s4.w};
                    const float ar = al * ln_row[mt][1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = (acc[nt][mt][4 * q + e] - ln_row[mt][0] * sv4[e]) * ar + bvv[e];
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = acc[nt][mt][4 * q + e] * al + bvv[e];
                }
            };

            // One straight-line pass sequence per case:
            // 0 = bias only (bf16 slab, 2 n-tiles per pass; 4 = the same for the phase up-conv's scattered output rows), 1 = + residual (fp32 slab, 1 n-tile per pass), 2 = GEGLU (bf16
            // slab, 4 n-tiles = 64 output columns per pass), 3 = everything else (activations, GEGLU + residual) on runtime
            // flags (fp32 slab, 1 n-tile per pass).
            auto run = [&](auto mode_tag) {
                constexpr int MODE = decltype(mode_tag)::value;
                constexpr bool F32 = MODE == 1 || MODE == 3;
                const bool gg = MODE == 2 || (MODE == 3 && geglu);
                const bool has_r = MODE == 1 || (MODE == 3 && R != nullptr);
                // WIDE: the +residual case of the tiles that have two slabs per wave runs 64-column fp32 passes through both of them
                // as ONE slab (row stride 272 B): 8 lanes then hold one row's 64 columns, and every residual load / store instruction
                // covers 8 rows x 128 contiguous bytes instead of 16 rows x 64 - whole cache lines on both streams.
                // (dense GEMMs only: the conv variants keep ~25 more addressing registers alive across the tile boundary and spilled
                //  380 - 416 B per lane with the wide pass's 32 + 32 read-back / residual registers)
                constexpr bool WIDE = MODE == 1 && DBL && TN >= 2 && !CONV;
                constexpr int RS = WIDE ? 272 : 144;                         // slab row stride in bytes
                constexpr bool TWO_SLABS = DBL && !WIDE;
                constexpr int NPT = (MODE == 0 || MODE == 4 || WIDE) ? 2 : (MODE == 2 ? 4 : 1);   // n-tiles per pass
                constexpr int PP = (TN + NPT - 1) / NPT;                    // passes per m-tile
                constexpr int NP = TM * PP;
                constexpr int MAXIT = (F32 && !WIDE) ? 2 : 4;               // 64-lane iterations of the row-major phase
                // residual rows are fetched DEPTH passes ahead (the fold / statistics variants and the wide passes - twice the rows per
                // pass - have no registers for a second one)
                constexpr int DEPTH = ((FEAT == 0 || FEAT == 8 || FEAT == 9) && !WIDE) ? 2 : 1, RRING = DEPTH + 1;
                static_assert(!WIDE || 32 * RS <= 2 * SLAB, "the wide fp32 pass must fit the wave's two slabs");
                auto slab_of = [&](int pi) { return slab0 + (TWO_SLABS ? (pi & 1) * SLAB : 0); };
                // geometry of pass pi
                auto p_mt = [&](int pi) { return pi / PP; };
                auto p_nt0 = [&](int pi) { return (pi % PP) * NPT; };
                auto p_cnt = [&](int pi) { return TN - p_nt0(pi) < NPT ? TN - p_nt0(pi) : NPT; };
                auto p_oc = [&](int pi) { return p_cnt(pi) * (gg ? 16 : 32); };                     // output columns
                auto p_col0 = [&](int pi) { return gg ? (wcol0 >> 1) + p_nt0(pi) * 16 : wcol0 + p_nt0(pi) * 32; };
                // item `it` of the row-major phase of pass pi: lane -> (row r of the m-tile, 8-column group cj).  The lane part of
                // its byte offset inside C / R goes into the VGPR offset (past num_records when the row or the column group
                // is outside the matrix), the pass part - m-tile row, first column - is wave-uniform and rides in the SGPR offset.
                auto item = [&](int pi, int it, int& r, int& cj, unsigned& vo_c, unsigned& vo_r) {
                    const int cpo = p_oc(pi) >> 3;          // 8-column groups per row: 8, 4 or 2
                    const int idx = lane_e + 64 * it;
                    r = idx / cpo;
                    cj = idx - r * cpo;
                    const int rows_left = p.M - (m0 + wm * TM * 32 + p_mt(pi) * 32);
                    const int cols_left = ncols_out - p_col0(pi);
                    const bool ok = r < rows_left && cj * 8 < cols_left;
                    if constexpr (MODE == 4) {   // phase up-conv: the output row is not linear in m
                        const long long d = out_row(ok ? m0 + wm * TM * 32 + p_mt(pi) * 32 + r : mfirst) - orow0;
                        vo_c = ok ? (unsigned)(d * p.ldc + p_col0(pi) + cj * 8) * 2u : kOOB;
                        vo_r = kOOB;
                    } else {
                        vo_c = ok ? (unsigned)(r * p.ldc + cj * 8) * 2u : kOOB;
                        vo_r = ok ? (unsigned)(r * p.ldr + cj * 8) * 2u : kOOB;
                    }
                };
                auto soff = [&](int pi, int ld) {   // wave-uniform part of the byte offset (not range-checked by the hardware)
                    if constexpr (MODE == 4) return 0;
                    return ((m0 - mfirst + wm * TM * 32 + p_mt(pi) * 32) * ld + p_col0(pi)) * 2;
                };
                auto n_iters = [&](int pi) { return (4 * p_oc(pi) + 63) / 64; };   // 32 rows x OC/8 items over 64 lanes

                // phase 1 of pass pi: scale / fold / bias (/ GEGLU) in the MFMA layout, park the values in the slab
                auto park = [&](int pi) {
                    const int mt = p_mt(pi);
                    char* const slab = slab_of(pi);
                    // (an opaque 0 in the vector indices: the two m-tiles read the SAME bias / fold vectors, and the compiler
                    //  would otherwise keep the first m-tile's 8 registers per quad alive for the second - they spilled)
                    int z = 0;
                    asm volatile("" : "+v"(z));
#pragma unroll
                    for (int k = 0; k < NPT; ++k) {
                        if (k >= p_cnt(pi)) continue;
                        const int nt = p_nt0(pi) + k;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            if (gg && q >= 2) continue;
                            float v[4];
                            quad_vals(gg, nt, mt, q, z, v);
                            const int cl = (gg ? k * 16 : k * 32) + 8 * q + 4 * lhi;   // column inside the pass
                            if constexpr (F32) {
                                *(slab_u4*)(slab + l31 * RS + cl * 4) =
                                    slab_u4{__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                            } else {
                                const slab_u2 pv = slab_u2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                                *(slab_u2*)(slab + l31 * RS + cl * 2) = pv;
                            }
                        }
                    }
                };
                // phase 2a: issue the row-major reads of pass pi
                u32x4_t rd[MAXIT][F32 ? 2 : 1];
                auto fetch = [&](int pi) {
                    const char* const slab = slab_of(pi);
#pragma unroll
                    for (int it = 0; it < MAXIT; ++it) {
                        if (it >= n_iters(pi)) continue;
                        int r, cj;
                        unsigned a, b;
                        item(pi, it, r, cj, a, b);
                        if constexpr (F32) {
                            rd[it][0] = *(const slab_u4*)(slab + r * RS + cj * 32);
                            rd[it][1] = *(const slab_u4*)(slab + r * RS + cj * 32 + 16);
                        } else {
                            rd[it][0] = *(const slab_u4*)(slab + r * RS + cj * 16);
                        }
                    }
                };
                // residual rows of pass pi, straight in the row-major layout
                u32x4_t rres[F32 ? RRING : 1][MAXIT];
                auto load_res = [&](int pi) {
                    if constexpr (F32) {
#pragma unroll
                        for (int it = 0; it < MAXIT; ++it) {
                            if (it >= n_iters(pi)) continue;
                            int r, cj;
                            unsigned a, b;
                            item(pi, it, r, cj, a, b);
                            rres[pi % RRING][it] = __builtin_amdgcn_raw_buffer_load_b128(rs_r, (int)b, soff(pi, p.ldr), 0);
                        }
                    }
                };
                // GroupNorm statistics of the tile as STORED (sdv_hip.h "gn_out"): per 32-row block and output column the (sum, sumsq)
                // of the bf16 values, so that the consumer's GroupNorm needs no statistics pass over the tensor.  In the row-major
                // phase a lane holds 8 columns of one row per item; the rows of a pass sit in the OTHER lanes and items, so the 16
                // per-lane values (8 columns x 2 statistics) are reduced across the lanes that share a column group with a transposing
                // butterfly: v_permlane32_swap / v_permlane16_swap fold lane bits 5 and 4 and halve the number of live values each
                // (both results of a swap are used), row_ror DPP adds fold the lane bits left inside a row of 16.  Afterwards lane
                // (rho0, rho1, column group cj) holds 4 consecutive columns of ONE statistic: one 16-byte store.  (Reducing one
                // statistic at a time - 8 -> 4 -> 2 values - was tried for register pressure: 192 B of scratch instead of 164 B.)
                // (8-wave tiles: compiled into the FEAT 4 variant only, so that the plain kernels - every launch whose consumer is not
                //  a GroupNorm - keep their registers: with the statistics code inside FEAT 0 the 256 x 320 conv went from 88 to 164 B
                //  of scratch per lane whether gn_out was set or not)
                constexpr bool GN_OK = (FEAT == 4 || (FEAT == 0 && NWV == 4)) && (MODE == 0 || MODE == 1 || MODE == 4);
                const bool gn_on = GN_OK && p.gn_out != nullptr;
                // The summation tree of a (32-row block, column) entry is the SAME whatever tile and pass width produced it - row bits
                // 4, 3 (the items: rows r, r+16 first, then r+8), then 2, 1, 0 (the lanes) - so the statistics of a tensor do not
                // depend on the tile the cost model picked for its producer (batch size, CFG-shared prefix ...).
                auto gn_reduce_store = [&](int pi, const u32x4_t* pk) {
                    // The file is built with -ffast-math: without these two lines every template instance is free to re-associate the
                    // sums below its own way, and the "same tree whatever the tile" of the comment above held by luck of codegen - the
                    // phase-form up-conv's statistics differed in the last fp32 bit in 0.3 % of the entries between the 256 x 320 tile and
                    // a 4-wave tile (tools/vae_tile_probe.py), enough to flip a uint8 of a frame decoded from large latents.
#pragma clang fp reassociate(off)
#pragma clang fp contract(off)
                    const int cpo = p_oc(pi) >> 3;                                  // lanes per row of the pass: 8 or 4
                    float cs[8], cq[8];
#pragma unroll
                    for (int h = 0; h < 4; ++h)
#pragma unroll
                        for (int o = 0; o < 2; ++o) {
                            auto el = [&](int it) { return o ? __uint_as_float(pk[it][h] & 0xffff0000u) : __uint_as_float(pk[it][h] << 16); };
                            if (n_iters(pi) >= 4) {            // rows r, r + 8, r + 16, r + 24 of this lane's column
                                const float g0 = el(0), g1 = el(1), g2 = el(2), g3 = el(3);
                                cs[2 * h + o] = (g0 + g2) + (g1 + g3);
                                cq[2 * h + o] = __builtin_fmaf(g2, g2, g0 * g0) + __builtin_fmaf(g3, g3, g1 * g1);
                            } else {                            // rows r, r + 16
                                const float g0 = el(0), g1 = el(1);
                                cs[2 * h + o] = g0 + g1;
                                cq[2 * h + o] = __builtin_fmaf(g1, g1, g0 * g0);
                            }
                        }
                    // v[4 k + 2 st + ch] = statistic st of column 4 ch + k: after the two swaps lane (rho0 = st, rho1 = ch) holds columns
                    // 4 ch .. 4 ch + 3 of statistic st in u[0..3]
                    float v[16];
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int st = 0; st < 2; ++st)
#pragma unroll
                            for (int ch = 0; ch < 2; ++ch) v[4 * k + 2 * st + ch] = (st ? cq : cs)[4 * ch + k];
                    float w[8], u[4];
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * i]), __float_as_uint(v[2 * i + 1]), false, false);
                        w[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(w[2 * k]), __float_as_uint(w[2 * k + 1]), false, false);
                        u[k] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        u[k] += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(u[k]), 0x128, 0xf, 0xf, false));   // row_ror:8
                        if (cpo == 4)
                            u[k] += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(u[k]), 0x124, 0xf, 0xf, false));   // row_ror:4
                    }
                    // lane (rho1 = lane >> 5, rho0 = (lane >> 4) & 1, cj = lane & 15 < cpo): statistic rho0 of columns cj*8 + rho1*4 .. +3
                    const int row_first = m0 + wm * TM * 32 + p_mt(pi) * 32;
                    const int cj = lane_e & 15, st = (lane_e >> 4) & 1, ch = lane_e >> 5;
                    const int col = p_col0(pi) + cj * 8 + ch * 4;
                    if (row_first < p.M && cj < cpo && col < ncols_out) {
                        const long long rb = bz * (long long)(p.M >> 5) + (row_first >> 5);
                        *(float4*)(p.gn_out + (rb * 2 + st) * p.gn_ld + col) = make_float4(u[0], u[1], u[2], u[3]);
                    }
                };
                // phase 2b: residual, activation, bf16, one 16-byte store per item (+ the row statistics of the stored values)
                auto finish = [&](int pi) {
                    u32x4_t gpk[MAXIT];      // the packed bf16 values of this pass's items (the store data, kept for the statistics)
#pragma unroll
                    for (int it = 0; it < MAXIT; ++it) {
                        if (it >= n_iters(pi)) continue;
                        int r, cj;
                        unsigned vo_c, vo_r;
                        item(pi, it, r, cj, vo_c, vo_r);
                        u32x4_t packed;
                        if constexpr (F32) {
                            float f[8];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                f[e] = __uint_as_float(rd[it][0][e]);
                                f[4 + e] = __uint_as_float(rd[it][1][e]);
                            }
                            if (has_r) {
                                float g[8];
                                unpack8(__builtin_bit_cast(bf16x8_raw, rres[pi % RRING][it]), g);
#pragma unroll
                                for (int e = 0; e < 8; ++e) f[e] += g[e];
                            }
                            if constexpr (MODE == 3) {
                                if (p.epi == 2) {
#pragma unroll
                                    for (int e = 0; e < 8; ++e) f[e] = silu_f(f[e]);
                                }
                                if constexpr (XACT) {
                                    if (p.epi == 3) {
#pragma unroll
                                        for (int e = 0; e < 8; ++e) f[e] = lrelu02_f(f[e]);
                                    } else if (p.epi == 4) {
#pragma unroll
                                        for (int e = 0; e < 8; ++e) f[e] = quick_gelu_f(f[e]);
                                    } else if (p.epi == 5) {
#pragma unroll
                                        for (int e = 0; e < 8; ++e) f[e] = gelu_erf_f(f[e]);
                                    }
                                }
                            }
                            packed = __builtin_bit_cast(u32x4_t, pack8(f));
                        } else {
                            packed = rd[it][0];
                        }
                        __builtin_amdgcn_raw_buffer_store_b128(packed, rs_c, (int)vo_c, soff(pi, p.ldc), 0);
                        // The 16 bytes of store data are NOT all sampled when the instruction issues: a VALU write to the first
                        // data register in the very next slot reached memory in lanes 12..15 of every 16 (measured, tools/
                        // epi_race_diag.py: the stored dword held the next item's column index).  The compiler's hazard table
                        // has no wait state for a store with an SGPR offset, so: keep the registers live across a few idle slots.
                        asm volatile("s_nop 7" ::"v"(packed), "v"(vo_c) : "memory");
                        if constexpr (GN_OK) gpk[it] = packed;   // (M is a multiple of 32 whenever gn_out is set: every row of a live block is live)
                        if constexpr (FEAT == 3) {   // LayerNorm statistics of the values as STORED (bf16), for the consumer's fold
                            float g[8], s1 = 0.f, s2 = 0.f;
                            unpack8(__builtin_bit_cast(bf16x8_raw, packed), g);
                            const bool live = (int)vo_c >= 0;
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const float ge = live ? g[e] : 0.f;
                                s1 += ge;
                                s2 = __builtin_fmaf(ge, ge, s2);
                            }
                            // the lanes that share a row are neighbours: butterfly over them, lane cj == 0 adds the pass's
                            // partial to the wave's per-row accumulators in LDS (same wave -> ordered)
                            const int cpo = p_oc(pi) >> 3;
                            if (cpo >= 2) {
                                s1 += __shfl_xor(s1, 1);
                                s2 += __shfl_xor(s2, 1);
                            }
                            if (cpo >= 4) {
                                s1 += __shfl_xor(s1, 2);
                                s2 += __shfl_xor(s2, 2);
                            }
                            if (cpo >= 8) {
                                s1 += __shfl_xor(s1, 4);
                                s2 += __shfl_xor(s2, 4);
                            }
                            if (cj == 0) {
                                atomicAdd(rowacc + 2 * r, s1);
                                atomicAdd(rowacc + 2 * r + 1, s2);
                            }
                        }
                    }
                    if constexpr (GN_OK) {
                        if (gn_on) gn_reduce_store(pi, gpk);
                    }
                };

                if constexpr (F32) {
                    if (has_r) {
#pragma unroll
                        for (int pi = 0; pi < DEPTH && pi < NP; ++pi) load_res(pi);
                    }
                }
                // exactly one run() executes per tile, and every wave takes the same one.  RAW barrier (own LDS ops retired +
                // s_barrier): __syncthreads() would add s_waitcnt vmcnt(0) and wait for the residual rows issued just above;
                // the loads this barrier publishes (the last K slab, the next tile's first slab) were waited for explicitly.
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if constexpr (FEAT == 3) rowacc[lane_e] = 0.f;
                // Slab hand-overs (write -> read of the same slab, read -> re-write with a single slab) are fenced with
                // s_waitcnt lgkmcnt(0): a wave's DS ops execute in order, the fence only keeps the COMPILER from moving the slab
                // accesses (differently typed views of the same bytes) across each other; with two slabs per wave it sits where
                // the older ops have long completed.
                // (experiment: compiler-only fence - a wave's DS ops execute in order, no hardware wait is needed)
                auto lds_fence = [&]() { asm volatile("" ::: "memory"); };
                park(0);
                lds_fence();
                fetch(0);
#pragma unroll
                for (int pi = 0; pi < NP; ++pi) {
                    if constexpr (F32) {
                        if (has_r && pi + DEPTH < NP) load_res(pi + DEPTH);
                    }
                    if (pi + 1 < NP) {
                        if constexpr (!TWO_SLABS) lds_fence();   // one slab: the rows of pass pi must have been read before it is re-used
                        park(pi + 1);
                    }
                    finish(pi);
                    if constexpr (FEAT == 3) {
                        if (pi % PP == PP - 1) {
                            // one (sum, sumsq) slot per (row, N tile, wave column): deterministic partials, no global atomics
                            const int mrow = m0 + wm * TM * 32 + p_mt(pi) * 32 + (lane_e >> 1);
                            if (mrow < p.M)
                                p.stats_out[((bz * p.M + mrow) * (long long)p.stats_p + (bn * WN + wn)) * 2 + (lane_e & 1)] = rowacc[lane_e];
                            rowacc[lane_e] = 0.f;
                        }
                    }
                    if (pi + 1 < NP) {
                        lds_fence();                       // (the values parked before finish(pi) have landed by now)
                        fetch(pi + 1);
                    }
                    // (keep the scheduler from hoisting later passes' conversions up here: with 160 live accumulators that spills)
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            if (CONV && p.mode == 4) run(std::integral_constant<int, CONV ? 4 : 0>{});   // (host: no residual / GEGLU with mode 4)
            else if (p.epi == 0 && !R) run(std::integral_constant<int, 0>{});
            else if (p.epi == 0) run(std::integral_constant<int, 1>{});
            else if (geglu && !R) run(std::integral_constant<int, 2>{});
            else if constexpr (XACT) run(std::integral_constant<int, 3>{});
            return;
        }
    }

    // ---- fallback epilogue straight from the MFMA registers (odd leading dims / N, e.g. the 77-token V^T) ----
    if (PERSIST && has_next) {   // (as above: the next tile's first slab is waited for here)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (geglu) {
        const int nout = p.N >> 1;
#pragma unroll
        for (int nt = 0; nt < TN; ++nt)
#pragma unroll
            for (int mt = 0; mt < TM; ++mt) {
                const int m = m0 + wm * TM * 32 + mt * 32 + l31;
                if (m >= p.M) continue;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int nv = wcol0 + nt * 32 + 8 * g + 4 * lhi;  // value row (interleaved index), gate = +16
                    const int oc = ((wcol0 + nt * 32) >> 1) + 8 * g + 4 * lhi;
                    if (oc >= nout) continue;
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a = acc[nt][mt][4 * g + e] * alpha;
                        float gt = acc[nt][mt][4 * (g + 2) + e] * alpha;
                        if (bias) {
                            a += bias[nv + e];
                            gt += bias[nv + 16 + e];
                        }
                        v[e] = a * geglu_gate_f(gt);
                    }
                    uint2 o;
                    o.x = pack_bf16x2(v[0], v[1]);
                    o.y = pack_bf16x2(v[2], v[3]);
                    *(uint2*)(C + (long long)m * p.ldc + oc) = o;
                }
            }
        return;
    }

    const bool vec_ok = ((p.ldc & 3) == 0) && (!R || (p.ldr & 3) == 0);
    const bool f32_vec_ok = ((p.ldc & 3) == 0) && ((p.sC & 3) == 0) && ((((uintptr_t)p.out_f32) & 15) == 0);
#pragma unroll
    for (int nt = 0; nt < TN; ++nt)
#pragma unroll
        for (int mt = 0; mt < TM; ++mt) {
            const int m = m0 + wm * TM * 32 + mt * 32 + l31;
            if (m >= p.M) continue;
            const float bm_ = (bias && p.bias_mode == 2) ? bias[m] : 0.f;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int nb = wcol0 + nt * 32 + 8 * g4 + 4 * lhi;
                if (nb >= p.N) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[nt][mt][4 * g4 + e] * (nb < acols ? alpha : 1.f) + bm_;
                    if (bias && p.bias_mode == 1 && nb + e < p.N) v[e] += bias[nb + e];
                }
                if constexpr (XACT) {
                    if (p.out_mode) {   // fp32 / image output (sdv_hip.h "out_mode"): the Cout <= 4 convolutions; fp32 scores of the VAE attention
                        if (p.out_mode == 1 && f32_vec_ok && nb + 3 < p.N) {
                            *(float4*)(p.out_f32 + bz * p.sC + (long long)m * p.ldc + nb) = make_float4(v[0], v[1], v[2], v[3]);
                            continue;
                        }
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (nb + e >= p.N) continue;
                            const long long idx = bz * p.sC + (long long)m * p.ldc + nb + e;
                            float t = v[e];
                            if (p.out_mode == 1) {
                                p.out_f32[idx] = t;
                            } else {
                                t = fminf(fmaxf(p.out_mode == 2 ? t * 0.5f + 0.5f : t, 0.f), 1.f);
                                if (p.out_f32) p.out_f32[idx] = t;
                                if (p.out_u8) p.out_u8[idx] = (uint8_t)rintf(t * 255.0f);
                            }
                        }
                        continue;
                    }
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
                    if constexpr (XACT) {
                        if (p.epi == 3) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = lrelu02_f(v[e]);
                        } else if (p.epi == 4) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = quick_gelu_f(v[e]);
                        } else if (p.epi == 5) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) v[e] = gelu_erf_f(v[e]);
                        }
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
                        if constexpr (XACT) {
                            if (p.epi == 3) t = lrelu02_f(t);
                            if (p.epi == 4) t = quick_gelu_f(t);
                            if (p.epi == 5) t = gelu_erf_f(t);
                        }
                        C[(long long)m * p.ldc + nb + e] = f32_to_bf16(t);
                    }
                }
            }
        }
    };

    // ---- the tile walk -------------------------------------------------------------------------------------------
    for (;;) {
        lane_setup();
        {
            int v = vb;
            if constexpr (PERSIST) asm volatile("" : "+s"(v));   // (recomputed, not carried across the previous epilogue)
            setup_tile(v);
        }
        e_m0 = a_m0;
        e_n0 = a_n0;
        e_bn = a_bn;
        e_bzi = a_bz;
        has_next = PERSIST && vb + (int)gridDim.x < total;
        if constexpr (SPLITK) {
            if (p.split_k > 1) {
                const int all = (K / BK) * ntaps;
                nkt = (int)((long long)(ks_ + 1) * all / p.split_k) - (int)((long long)ks_ * all / p.split_k);
            }
        }
#pragma unroll
        for (int a = 0; a < TN; ++a)
#pragma unroll
            for (int b = 0; b < TM; ++b)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
        kloop();
        epilogue();
        if (!has_next) break;
        vb += (int)gridDim.x;
        slot0 = (slot0 + nkt) & 1;
        landed = true;
    }
}

// Second pass of a split-K launch: C[m][n] = bf16( alpha * sum_s ws[s][m][n] + bias (+ R[m][n]) ), the S partials added in split order.
// GN: a workgroup owns a 32-row x 64-column block of the result and also emits the block's per-column (sum, sumsq) of the STORED bf16
// values into gn_out (sdv_hip.h "gn_out" layout) - the consumer's GroupNorm then needs no statistics pass of its own, as after an
// unsplit launch (rows summed in row order: deterministic; not the unsplit epilogue's tree, but a split result is another fp32
// summation order anyway).
template <bool GN>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S, int M, int N, float alpha,
                                                            const float* __restrict__ bias, int bias_mode, const int32_t* step_ptr,
                                                            int bias_step_stride, const uint16_t* __restrict__ R, int ldr,
                                                            uint16_t* __restrict__ C, int ldc, float* __restrict__ gn_out, int gn_ld) {
    if (bias && step_ptr) bias += (long long)(*step_ptr) * bias_step_stride;
    auto one = [&](int m, int n, float* kept) {      // one float4 of the result: reduce, finish, round, store
        float4 a = *(const float4*)(ws + (long long)m * N + n);
        for (int s = 1; s < S; ++s) {
            const float4 b = *(const float4*)(ws + ((long long)s * M + m) * N + n);
            a.x += b.x;
            a.y += b.y;
            a.z += b.z;
            a.w += b.w;
        }
        float v[4] = {a.x * alpha, a.y * alpha, a.z * alpha, a.w * alpha};
        if (bias) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] += bias_mode == 2 ? bias[m] : bias[n + e];
        }
        if (R) {
            const uint2 r = *(const uint2*)(R + (long long)m * ldr + n);
            v[0] += __uint_as_float(r.x << 16);
            v[1] += __uint_as_float(r.x & 0xffff0000u);
            v[2] += __uint_as_float(r.y << 16);
            v[3] += __uint_as_float(r.y & 0xffff0000u);
        }
        uint2 o;
        o.x = pack_bf16x2(v[0], v[1]);
        o.y = pack_bf16x2(v[2], v[3]);
        *(uint2*)(C + (long long)m * ldc + n) = o;
        if (kept) {
            kept[0] = __uint_as_float(o.x << 16);
            kept[1] = __uint_as_float(o.x & 0xffff0000u);
            kept[2] = __uint_as_float(o.y << 16);
            kept[3] = __uint_as_float(o.y & 0xffff0000u);
        }
    };
    if constexpr (!GN) {
        const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
        const int nq = N >> 2;
        if (q >= (long long)M * nq) return;
        const int m = (int)(q / nq);
        one(m, (int)(q - (long long)m * nq) * 4, nullptr);
    } else {
        __shared__ float blk[32][65];
        const int r = threadIdx.x >> 3, cg = threadIdx.x & 7;
        const int m = blockIdx.x * 32 + r, n0 = blockIdx.y * 64;     // (M % 32 == 0 whenever gn_out is set)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = h * 32 + cg * 4;
            float kept[4] = {0.f, 0.f, 0.f, 0.f};
            if (n0 + c < N) one(m, n0 + c, kept);
#pragma unroll
            for (int e = 0; e < 4; ++e) blk[r][c + e] = kept[e];
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            const int c = threadIdx.x & 63, st = threadIdx.x >> 6;
            float acc = 0.f;
            for (int rr = 0; rr < 32; ++rr) {
                const float x = blk[rr][c];
                acc = st ? __builtin_fmaf(x, x, acc) : acc + x;
            }
            if (n0 + c < N) gn_out[((long long)blockIdx.x * 2 + st) * gn_ld + n0 + c] = acc;
        }
    }
}

int g_persistent = 1;   // sdv_gemm_set_persistent(): A/B switch for tools/ (0 = one workgroup per tile, as in round 1)
int g_grid_limit = 0;   // sdv_gemm_set_grid_limit(): > 0 caps the persistent grid (tests: make small problems WALK tiles)

int current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64 ? dev : 0;
}

// CU count of the CURRENT device (per device: a process may drive several GPUs - SDV_FORCE_DEVICE tests, in-process fan-out)
int num_cus() {
    static int n[64] = {0};
    const int dev = current_device();
    if (!n[dev]) {
        hipDeviceProp_t prop;
        int c = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 0;
        n[dev] = c > 0 ? c : 256;
    }
    return n[dev];
}

template <int WM, int WN, int TM, int TN, int BK, bool CONV, int NST, int FEAT = 0>
int launch_igemm_t(const sdv_gemm_args& a, hipStream_t stream) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int TILE_BYTES = (BM + BN) * BK * (FEAT == 8 ? 1 : 2);   // (FEAT 9: two 64-byte-row images = the bf16 tile's bytes)
    static_assert(NST == 2, "double-buffered K slabs");
    constexpr bool PERSIST = WM * WN == 8;                             // (see the kernel)
    constexpr int SLABS = 2 * WM * WN * 32 * 144;                      // the epilogue's staging slabs (alias ONE K-slab buffer)
    constexpr int SLOT = PERSIST && SLABS > TILE_BYTES ? SLABS : TILE_BYTES;
    // K-slab buffers + column vectors + FEAT 3's row-stat accumulators.  + 48 KiB that nothing uses for the LayerNorm-fold / row-statistics
    // variants of the 4-wave 128 x 128 tile: that tile normally runs TWO workgroups per CU (68 KiB each), and its fold variants are
    // not safe in company - round 4's column-side fold raced against its own second workgroup (11 of 11 repeated launches differed),
    // and the row-side GEGLU consumer gave 12 of 300 forwards a different result while ANOTHER PROCESS kept the GPU busy (two ranks on
    // one GPU: tools/contention_probe.py, profiles/round5_contention_probe.txt); alone on its CU - this padding, or the 8-wave tiles,
    // which own their CU's LDS anyway - both were clean in 300 of 300.  What exactly two co-resident workgroups do to each other
    // there is NOT understood (idle slots, full drains and full barriers around the LDS vector reads changed nothing: DESIGN.md);
    // one workgroup per CU is the measured cure, and it costs 0 - 2 % at 1 - 4 frames per call, nothing above.
    constexpr int LDS = NST * SLOT + 3 * BN * 4 + WM * WN * 256 + ((WM * WN == 4 && (FEAT == 1 || FEAT == 3)) ? 48 * 1024 : 0);
    static_assert(LDS <= 160 * 1024, "tile does not fit the 160 KiB LDS");
    static unsigned long long attr_set = 0;   // one bit per device: the attribute belongs to the device's copy of the function
    const unsigned long long dev_bit = 1ull << current_device();
    if (LDS > 64 * 1024 && !(attr_set & dev_bit)) {
        (void)hipFuncSetAttribute((const void*)igemm_kernel<WM, WN, TM, TN, BK, CONV, NST, FEAT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  LDS);
        attr_set |= dev_bit;
    }
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const long long total = (long long)tiles_m * tiles_n * (a.split_k > 1 ? a.split_k : (a.batch > 0 ? a.batch : 1));
    SDV_REQUIRE(total < 0x7fffffffLL, "sdv_gemm_bf16: too many tiles");
    SDV_REQUIRE(a.split_k <= 1 || (WM * WN == 4 && FEAT == 0), "sdv_gemm_bf16: split-K exists in the plain 4-wave tiles only");
    const int cus = g_grid_limit > 0 && g_grid_limit < num_cus() ? g_grid_limit : num_cus();
    const bool walk = PERSIST && g_persistent && total > cus;
    dim3 grid((unsigned)(walk ? cus : total), 1, 1);
    hipLaunchKernelGGL((igemm_kernel<WM, WN, TM, TN, BK, CONV, NST, FEAT>), grid, dim3(WM * WN * 64), LDS, stream, a);
    SDV_CHECK_LAUNCH("sdv_gemm_bf16");
    return SDV_OK;
}

// LN_OK: the tile carries the LayerNorm-fold / row-statistics / fp8 variants
template <int WM, int WN, int TM, int TN, int BK, int NST = 2, bool LN_OK = false>
int launch_igemm(const sdv_gemm_args& a, hipStream_t stream) {
    if (a.fp8) {
        if constexpr (LN_OK && BK == 64 && NST == 2) {   // the same four 8-wave / 4-wave tiles carry the fp8 variants
            SDV_REQUIRE(!a.ln_side && !a.stats_out, "sdv_gemm_bf16: fp8 operands do not combine with the LayerNorm fold");
            if (a.fp8 == 2)
                return a.mode == 0 ? launch_igemm_t<WM, WN, TM, TN, BK, false, NST, 9>(a, stream)
                                   : launch_igemm_t<WM, WN, TM, TN, BK, true, NST, 9>(a, stream);
            return a.mode == 0 ? launch_igemm_t<WM, WN, TM, TN, BK, false, NST, 8>(a, stream)
                               : launch_igemm_t<WM, WN, TM, TN, BK, true, NST, 8>(a, stream);
        }
        SDV_REQUIRE(false, "sdv_gemm_bf16: fp8 operands exist on tiles 1, 6, 7, 9 only");
    }
    if (a.ln_side || a.stats_out) {
        if constexpr (LN_OK) {
            if (a.mode == 0 && a.ln_side == 1 && !a.stats_out) return launch_igemm_t<WM, WN, TM, TN, BK, false, NST, 1>(a, stream);
            if (a.mode == 0 && a.ln_side == 0) return launch_igemm_t<WM, WN, TM, TN, BK, false, NST, 3>(a, stream);
        }
        SDV_REQUIRE(false, "sdv_gemm_bf16: the LayerNorm fold / row statistics exist for dense GEMMs on tiles 1, 6, 7, 9 only (not combined)");
    }
    if constexpr (WM * WN == 8) {
        if (a.gn_out) {   // the 8-wave tiles carry the GroupNorm-statistics epilogue as a variant of its own (FEAT 4)
            return a.mode == 0 ? launch_igemm_t<WM, WN, TM, TN, BK, false, NST, 4>(a, stream)
                               : launch_igemm_t<WM, WN, TM, TN, BK, true, NST, 4>(a, stream);
        }
    }
    return a.mode == 0 ? launch_igemm_t<WM, WN, TM, TN, BK, false, NST>(a, stream)
                       : launch_igemm_t<WM, WN, TM, TN, BK, true, NST>(a, stream);
}

}  // namespace

static int sdv_gemm_impl(const sdv_gemm_args* args, void* stream, int plan);   // plan: 0 launch, 1 -> stats slots, 2 -> split-K factor

extern "C" int sdv_gemm_bf16(const sdv_gemm_args* args, void* stream) { return sdv_gemm_impl(args, stream, 0); }

// Split-K factor sdv_gemm_bf16 would use for these arguments if it were given a workspace (>= 2), or 1: the caller sizes the
// workspace out_f32 as [S][M][N] floats, sets split_k = S and launches.
extern "C" int sdv_gemm_split_k(const sdv_gemm_args* args) { return sdv_gemm_impl(args, nullptr, 2); }

// Number of (sum, sumsq) slots per output row that sdv_gemm_bf16 would write to `stats_out` for these arguments
// (= N tiles x wave columns of the tile the launch would pick); the caller sizes stats_out as [batch][M][slots][2] floats.
extern "C" int sdv_gemm_stats_slots(const sdv_gemm_args* args) { return sdv_gemm_impl(args, nullptr, 1); }

// 1 (default): the 8-wave tiles run as persistent workgroups (one per CU, walking tiles); 0: one workgroup per tile.  Returns
// the previous setting.  Results are identical either way; this exists so tools/ can time both on the same box.
extern "C" int sdv_gemm_set_persistent(int on) {
    const int prev = g_persistent;
    g_persistent = on ? 1 : 0;
    return prev;
}

extern "C" int sdv_gemm_set_grid_limit(int n) {
    const int prev = g_grid_limit;
    g_grid_limit = n > 0 ? n : 0;
    return prev;
}

static int sdv_gemm_impl(const sdv_gemm_args* args, void* stream, int plan) {
    SDV_REQUIRE(args != nullptr, "sdv_gemm_bf16: null args");
    sdv_gemm_args a = *args;
    SDV_REQUIRE(a.X && a.W && (a.C || a.out_mode), "sdv_gemm_bf16: null operand");
    // split-K: `split_k` > 1 = the caller holds a workspace out_f32 [split_k][M][N] (out_mode 0) and allows up to that many splits;
    // plan 2 asks how many this launch would take
    const int split_cap = plan == 2 ? 8 : ((a.split_k > 1 && a.out_f32 && !a.out_mode) ? (a.split_k < 8 ? a.split_k : 8) : 1);
    a.split_k = 1;
    SDV_REQUIRE(a.out_mode >= 0 && a.out_mode <= 3, "sdv_gemm_bf16: bad out_mode %d", a.out_mode);
    if (a.out_mode) {
        // out_mode 1 (fp32 output) takes any N and a batch (stride sC, in fp32 elements): the VAE attention's scores leave the
        // accumulators unrounded; the image forms (2, 3) are the N <= 32 output convolutions
        SDV_REQUIRE(a.ldc >= a.N && !a.R && a.epi == 0 && !a.ln_side && !a.stats_out && a.mode != 4 && !a.gn_out,
                    "sdv_gemm_bf16: out_mode is a plain output form (no residual / activation / fold / statistics)");
        SDV_REQUIRE(a.out_mode == 1 || (a.N <= 32 && a.batch <= 1), "sdv_gemm_bf16: the image output forms take N <= 32 and no batch");
        SDV_REQUIRE(a.out_mode == 1 ? a.out_f32 != nullptr : (a.out_f32 || a.out_u8), "sdv_gemm_bf16: out_mode %d without an output", a.out_mode);
        SDV_REQUIRE(a.tile == 0 || (a.tile >= 1 && a.tile <= 4) || a.tile == 10 || a.tile == 11, "sdv_gemm_bf16: out_mode exists in the 4-wave tiles only");
        if (a.tile == 0) a.tile = a.N <= 32 ? 10 : (a.N <= 64 ? 11 : 4);   // 256 x 32: one 32-column MFMA tile holds all the outputs; wide: 256 x 128
    }
    SDV_REQUIRE(a.M > 0 && a.N > 0 && a.K > 0, "sdv_gemm_bf16: bad shape M=%d N=%d K=%d", a.M, a.N, a.K);
    SDV_REQUIRE(a.K % 64 == 0, "sdv_gemm_bf16: K=%d must be a multiple of 64", a.K);
    SDV_REQUIRE(a.mode >= 0 && a.mode <= 4, "sdv_gemm_bf16: bad mode %d", a.mode);
    SDV_REQUIRE(a.epi >= 0 && a.epi <= 5, "sdv_gemm_bf16: bad epi %d", a.epi);
    if (!a.X2) {
        a.C1 = a.K;
        a.ldx2 = a.ldx;
        a.X2 = a.X;
    }
    SDV_REQUIRE(a.C1 % 64 == 0 && a.C1 > 0 && a.C1 <= a.K, "sdv_gemm_bf16: C1=%d must be a multiple of 64 in (0,K]", a.C1);
    SDV_REQUIRE(a.ldx % 8 == 0 && a.ldx2 % 8 == 0 && a.ldw % 8 == 0, "sdv_gemm_bf16: ldx/ldx2/ldw must be multiples of 8");
    SDV_REQUIRE(a.fp8 >= 0 && a.fp8 <= 2, "sdv_gemm_bf16: bad fp8 flag %d", a.fp8);
    if (a.fp8) SDV_REQUIRE(a.ldx % 16 == 0 && a.ldx2 % 16 == 0 && a.ldw % 16 == 0, "sdv_gemm_bf16: fp8 rows must be 16-byte multiples");
    if (a.mode != 0) {
        SDV_REQUIRE(a.Hin > 0 && a.Win > 0 && a.Hout > 0 && a.Wout > 0, "sdv_gemm_bf16: bad conv geometry");
        SDV_REQUIRE(a.M % (a.Hout * a.Wout) == 0, "sdv_gemm_bf16: M must be nimg*Hout*Wout");
        SDV_REQUIRE(a.ldw >= (a.mode == 4 ? 4 : 9) * a.K, "sdv_gemm_bf16: conv weights must be [N][3][3][K]");
        SDV_REQUIRE(a.batch <= 1, "sdv_gemm_bf16: conv modes are not batched");
        if (a.mode == 1) SDV_REQUIRE(a.Hin == a.Hout && a.Win == a.Wout, "conv s1 geometry");
        if (a.mode == 2) SDV_REQUIRE(a.Hout == (a.Hin + 1) / 2 && a.Wout == (a.Win + 1) / 2, "conv s2 geometry");
        if (a.mode == 3) SDV_REQUIRE(a.Hout == 2 * a.Hin && a.Wout == 2 * a.Win, "upsample-conv geometry");
        if (a.mode == 4) {
            SDV_REQUIRE(a.Hin == a.Hout && a.Win == a.Wout, "phase upsample-conv: Hout/Wout are the LOW-resolution grid");
            SDV_REQUIRE(a.ldw >= 4 * a.K, "sdv_gemm_bf16: phase upsample-conv weights must be [4][N][2][2][K]");
            SDV_REQUIRE(!a.R && a.epi != 1 && (a.ldc & 7) == 0 && (a.N & 7) == 0 && (((uintptr_t)a.C) & 15) == 0,
                        "sdv_gemm_bf16: phase upsample-conv needs the aligned epilogue (no residual / GEGLU)");
            a.batch = 4;                                  // blockIdx.z = phase (py, px)
            a.sX = 0;
            a.sC = 0;
            a.sR = 0;
            a.sW = (long long)a.N * a.ldw;
            auto magic = [](unsigned d, unsigned* mul, unsigned* shr) {   // n / d == umulhi(n, mul) >> shr for n < 2^31
                unsigned s = 0;
                while ((2u << s) <= d) ++s;               // s = floor(log2 d)
                if ((1u << s) == d) {
                    *mul = 0;
                    *shr = s;
                } else {
                    *mul = (unsigned)((((unsigned long long)1 << (32 + s)) + d - 1) / d);
                    *shr = s;
                }
            };
            magic((unsigned)(a.Hout * a.Wout), &a.div_hw_mul, &a.div_hw_shr);
            magic((unsigned)a.Wout, &a.div_w_mul, &a.div_w_shr);
        }
        // 31-bit lane offsets inside the workgroup's window (a 320-row tile spans at most 320/HWout + 2 images)
        const long long win = ((long long)(320 / (a.Hout * a.Wout)) + 2) * a.Hin * a.Win * (a.ldx > a.ldx2 ? a.ldx : a.ldx2) * 2;
        SDV_REQUIRE(win < 0x7fffffffLL, "sdv_gemm_bf16: conv window too large for 31-bit offsets");
    } else {
        SDV_REQUIRE(320LL * (a.ldx > a.ldx2 ? a.ldx : a.ldx2) * 2 < 0x7fffffffLL, "sdv_gemm_bf16: ldx too large");
    }
    SDV_REQUIRE(320LL * a.ldw * 2 < 0x7fffffffLL, "sdv_gemm_bf16: ldw too large");
    if (a.epi == 1) {
        SDV_REQUIRE(a.N % 32 == 0, "sdv_gemm_bf16: GEGLU needs N %% 32 == 0");
        SDV_REQUIRE(a.ldc % 4 == 0, "sdv_gemm_bf16: GEGLU needs ldc %% 4 == 0");
    }
    if (a.bias_mode == 0 && a.bias) a.bias_mode = 1;
    if (a.alpha == 0.f) a.alpha = 1.f;
    SDV_REQUIRE(a.ln_side == 0 || a.ln_side == 1, "sdv_gemm_bf16: bad ln_side %d (1 = the row-side LayerNorm fold; the column-side form 2 of ABI <= 9 is gone)", a.ln_side);
    if (a.ln_side) SDV_REQUIRE(a.ln_stats && a.ln_s, "sdv_gemm_bf16: ln_side needs ln_stats + ln_s");
    if (a.ln_side || a.stats_out)
        SDV_REQUIRE((a.ldc & 7) == 0 && ((a.epi == 1 ? a.N >> 1 : a.N) & 7) == 0 && (((uintptr_t)a.C | (uintptr_t)a.bias | (uintptr_t)a.ln_s | (uintptr_t)a.ln_stats) & 15) == 0 &&
                        (!a.R || (a.ldr & 7) == 0) && ((a.sC | a.sR) & 7) == 0 && (((uintptr_t)a.R) & 15) == 0 && a.mode != 4,
                    "sdv_gemm_bf16: the LayerNorm fold / row statistics need the aligned epilogue");
    SDV_REQUIRE(a.alpha_cols >= 0 && a.alpha_cols % 8 == 0 && (a.alpha_cols == 0 || a.epi != 1),
                "sdv_gemm_bf16: alpha_cols=%d must be a multiple of 8 (and is not available with GEGLU)", a.alpha_cols);
    if (a.gn_out) {
        const int nout = a.N;
        SDV_REQUIRE(a.epi == 0 && !a.out_mode && !a.ln_side && !a.stats_out && !a.fp8 && a.M % 32 == 0 && (a.ldc & 7) == 0 && (nout & 7) == 0 &&
                        (((uintptr_t)a.C | (uintptr_t)a.R | (uintptr_t)a.bias) & 15) == 0 && (!a.R || (a.ldr & 7) == 0) && ((a.sC | a.sR) & 7) == 0 &&
                        a.gn_ld >= nout && (a.gn_ld & 3) == 0 && (((uintptr_t)a.gn_out) & 15) == 0,
                    "sdv_gemm_bf16: gn_out needs the plain bf16 epilogue (epi 0, no fold / fp8 / out_mode), M %% 32 == 0 and aligned operands");
    }
    hipStream_t s = (hipStream_t)stream;
    int tile = a.tile;
    const long long nb = a.batch > 0 ? a.batch : 1;
    auto blocks = [&](int bm, int bn) { return (long long)((a.M + bm - 1) / bm) * ((a.N + bn - 1) / bn) * nb; };
    // Split-K factor of a small-tile launch with `nblocks` output tiles: only plain launches (the second pass applies alpha / bias /
    // residual - no activation, fold, statistics, typed output, batch or phase form), only when the tiles leave most CUs idle, never
    // fewer than 16 K slabs (1024 K values) per split, at most what the caller's workspace holds.
    const long long kslabs = (long long)(a.K / 64) * (a.mode ? 9 : 1);
    // (alpha_cols: the second pass scales EVERY column by alpha - a launch that scales only its leading columns is never split;
    //  C / R 8-byte aligned: the second pass moves them as uint2)
    const bool split_ok = split_cap > 1 && !a.fp8 && !a.ln_side && !a.stats_out && !a.out_mode && a.epi == 0 && a.batch <= 1 &&
                          a.mode != 4 && (a.N & 3) == 0 && (a.ldc & 3) == 0 && (!a.R || (a.ldr & 3) == 0) && !a.alpha_cols &&
                          ((((uintptr_t)a.C | (uintptr_t)a.R) & 7) == 0) && ((((uintptr_t)a.out_f32) & 15) == 0);
    auto splits_for = [&](long long nblocks) -> int {
        if (!split_ok || nblocks > 128 || kslabs < 32) return 1;
        long long sk = 256 / nblocks;
        if (sk > kslabs / 16) sk = kslabs / 16;
        if (sk > split_cap) sk = split_cap;
        return sk >= 2 ? (int)sk : 1;
    };
    if (tile == 0) {
        // Cost model over the compiled tiles, calibrated with tools/tile_sweep.py on MI355X: time ~ workgroups the
        // busiest CU runs x tile area / rate, rate = MFMA throughput per busy CU (TFLOP/s) of that tile's K loop.
        // The 8-wave 256-row tiles move the fewest L2->LDS bytes per MFMA (the 128x128 tile saturates L2 bandwidth
        // near 0.9 PF/s) but quantise badly on the low-resolution levels, where the small tiles win.
        // (Measured and rejected: 32-wide K tiles with two workgroups per CU - slower than one big workgroup on
        //  every UNet shape, short K included; profiles/round1_tile_sweep_nimg64.txt.  Re-measured in round 4 with the persistent
        //  256 x 320 tile as the baseline - 128 x 320 x 32 tiles, two workgroups per CU, 4 or 8 waves each: 0.35 - 0.75x on all 15
        //  transformer shapes; profiles/round4_two_workgroups_per_cu.txt.)
        struct Cand { int id, bm, bn; float rate; };
        static const Cand cands[] = {{6, 256, 320, 5.0f}, {7, 256, 256, 4.7f}, {9, 128, 320, 4.2f}, {8, 256, 128, 3.3f},
                                     {1, 128, 128, 3.4f}, {2, 128, 64, 2.4f}, {3, 64, 64, 2.0f}};
        // (the 4-wave 256x32 / 256x64 tiles 10 / 11 stay selectable but are not candidates: on the RRDBNet convs the
        //  128x64 tile wins or ties everywhere - tools/esrgan_tile_sweep.py, profiles/round1_esrgan.txt)
        double best = 1e300;
        for (const Cand& c : cands) {
            if (a.epi >= 3 && c.id >= 6) continue;   // extended activations exist in the 4-wave tiles only
            if ((a.ln_side || a.stats_out || a.fp8) && !(c.id == 1 || c.id == 6 || c.id == 7 || c.id == 9)) continue;   // LN fold / fp8 tiles
            // The LayerNorm-fold + GEGLU epilogue never runs on the 4-wave 128 x 128 tile: that one kernel variant - igemm_kernel<2, 2, 2, 2,
            // 64, false, 2, 1> in its GEGLU pass - gave 24 - 29 of 1500 forwards a different result while ANOTHER PROCESS kept the same GPU
            // busy (tools/contention_probe.py; alone it is bit-reproducible, and so is every other fold / statistics variant in company:
            // profiles/round6_contention_bisect.txt).  The 8-wave tiles run the same epilogue source clean in 1500 of 1500.
            if (c.id == 1 && a.ln_side && a.epi == 1) continue;
            const int sk = c.id <= 3 ? splits_for(blocks(c.bm, c.bn)) : 1;
            const long long per_cu = (blocks(c.bm, c.bn) * sk + 255) / 256;       // workgroups on the busiest CU
            const double cost = (double)per_cu * c.bm * c.bn / c.rate / sk;       // padded tiles are counted
            if (cost < best) {
                best = cost;
                tile = c.id;
                a.split_k = sk;
            }
        }
    } else if (tile >= 1 && tile <= 3) {
        a.split_k = splits_for(blocks(tile == 3 ? 64 : 128, tile == 1 ? 128 : 64));
    }
    a.tile = 4;   // the kernel reads `tile` as the raster strip width: 8 x 4 blocks of output tiles per XCD wave (1 / 2 / 4 / 8
                  // measured on the UNet: 120.2 / 118.5 / 118.1 / 118.0 ms per forward, profiles/round2_raster_order.txt)
    {
        static const int kBN[] = {0, 128, 64, 64, 128, 0, 320, 256, 128, 320, 32, 64};
        static const int kWN[] = {0, 2, 1, 2, 2, 0, 2, 2, 2, 2, 1, 1};
        SDV_REQUIRE(tile >= 1 && tile <= 11 && kBN[tile], "sdv_gemm_bf16: bad tile %d (1-4, 6-11)", tile);
        a.stats_p = ((a.N + kBN[tile] - 1) / kBN[tile]) * kWN[tile];
    }
    if (plan == 1) return a.stats_p;
    if (plan == 2) return a.split_k;
    SDV_REQUIRE(!(a.epi >= 3 && tile >= 6 && tile <= 9), "sdv_gemm_bf16: epi %d is not available in the 8-wave tile %d", a.epi, tile);
    SDV_REQUIRE(!(a.epi == 1 && a.R), "sdv_gemm_bf16: GEGLU does not take a residual");
    auto run_tile = [&]() -> int {
    switch (tile) {
#ifdef SDV_GEMM_ONLY_TILE6   // (tools: compile the 256 x 320 tile alone for resource / ISA inspection)
        case 6: return launch_igemm<4, 2, 2, 5, 64, 2, true>(a, s);
#else
        case 1: return launch_igemm<2, 2, 2, 2, 64, 2, true>(a, s);    // 128 x 128, 4 waves
        case 2: return launch_igemm<4, 1, 1, 2, 64>(a, s);    // 128 x  64
        case 3: return launch_igemm<2, 2, 1, 1, 64>(a, s);    //  64 x  64
        case 4: return launch_igemm<2, 2, 4, 2, 64>(a, s);    // 256 x 128, 4 waves
        case 6: return launch_igemm<4, 2, 2, 5, 64, 2, true>(a, s);    // 256 x 320, 8 waves (UNet widths are multiples of 320)
        case 7: return launch_igemm<4, 2, 2, 4, 64, 2, true>(a, s);    // 256 x 256, 8 waves
        case 8: return launch_igemm<4, 2, 2, 2, 64>(a, s);    // 256 x 128, 8 waves
        case 9: return launch_igemm<4, 2, 1, 5, 64, 2, true>(a, s);    // 128 x 320, 8 waves
        case 10: return launch_igemm<4, 1, 2, 1, 64>(a, s);   // 256 x  32, 4 waves (RRDB growth convs, Cout = 32)
        case 11: return launch_igemm<4, 1, 2, 2, 64>(a, s);   // 256 x  64, 4 waves
#endif
        default: SDV_REQUIRE(false, "sdv_gemm_bf16: bad tile %d", tile);
    }
    return SDV_OK;
    };
    const int rc = run_tile();
    if (rc != SDV_OK || a.split_k <= 1) return rc;
    // second pass of a split-K launch (same stream: ordered behind the partial sums)
    const long long quads = (long long)a.M * (a.N >> 2);
    if (a.gn_out)     // (the GroupNorm statistics of a split launch come out of the second pass - the first one only holds partial sums)
        hipLaunchKernelGGL(splitk_reduce_kernel<true>, dim3((unsigned)(a.M / 32), (unsigned)((a.N + 63) / 64)), dim3(256), 0, s, a.out_f32, a.split_k,
                           a.M, a.N, a.alpha, a.bias, a.bias_mode, a.step_ptr, a.bias_step_stride, a.R, a.ldr, a.C, a.ldc, a.gn_out, a.gn_ld);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<false>, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, a.out_f32, a.split_k, a.M, a.N,
                           a.alpha, a.bias, a.bias_mode, a.step_ptr, a.bias_step_stride, a.R, a.ldr, a.C, a.ldc, (float*)nullptr, 0);
    SDV_CHECK_LAUNCH("sdv_gemm_bf16 (split-K reduce)");
    return SDV_OK;
}
