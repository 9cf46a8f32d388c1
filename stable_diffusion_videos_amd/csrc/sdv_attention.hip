// Flash-style attention for the UNet transformer blocks (self-attention, Lk = Lq in {4096,1024,256,64},
// and text cross-attention, Lk = 77) on the gfx950 matrix cores.  Never materialises the Lq x Lk
// score matrix - the reason the reference has to expose attention slicing
// (/root/reference/.../stable_diffusion_pipeline.py:161-189).
// Replaces CrossAttention.forward inside unet(...) (stable_diffusion_pipeline.py:418).
//
// Formulation ("swapped", so that everything per-query is lane-local):
//   S^T[key][q] = K . Q^T      v_mfma_f32_32x32x16_bf16, A = K rows (LDS), B = Q rows (registers)
//       -> lane l owns query q = l & 31 and 16 keys per 32-key subtile in its accumulator registers;
//          the row max / row sum are register reductions plus ONE exchange with lane l^32.
//   O^T[d][q]  = V^T . P^T     A = V^T rows, B = P^T straight from the softmax registers (no cross-lane movement: the
//       PV contraction may visit keys in any order as long as A and B agree).  Two forms of the A operand:
//         * V supplied TRANSPOSED ([head dim][keys], the text context's V^T, made once per walk): the tile is written to
//           LDS in the MFMA C-layout key order and one ds_read_b128 yields a lane's 8 keys;
//         * V supplied ROW-MAJOR (VRM: [keys][head dim] - the V columns of the fused QKV projection, so the UNet's
//           self-attention needs no transposed V^T projection launch): the tile is staged exactly like K and the
//           transpose happens in the LDS read - ds_read_b64_tr_b16 hands lane (d, 4-key group) the four keys of ITS
//           d from four different rows (two reads per 8-key fragment).
//   -> the running rescale O *= exp2(m_old - m_new) is also lane-local.
// One workgroup = 4 waves x 32 queries = 128 queries of one head; K / V^T tiles of 64 keys are
// register-staged (global -> VGPR issued before the MFMA phase, VGPR -> LDS after the barrier).
// Head sizes: dh = 40 (K padded to 48 for QK^T, to 64 rows for PV), 64, 80 (96 rows for PV), 160.
#include "sdv_common.h"

namespace {

constexpr f32x16_t kZero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#ifndef SDV_ATTN_SLAB64
#define SDV_ATTN_SLAB64 0
#endif
#ifndef SDV_ATTN_PP_DBUF
#define SDV_ATTN_PP_DBUF 0
#endif
constexpr float kDefer = 5.0f;  // log2 units: skip the O rescale while the running max moves by < 2^5

template <int DH, bool VRM = false>
struct AttnCfg {
    static constexpr int DKS = (DH + 15) / 16;        // 16-wide k-steps of Q.K^T
    static constexpr int DKP = DKS * 16;              // padded head dim for Q.K^T
    static constexpr int DVT = (DH + 31) / 32;        // 32-row tiles of O^T
    static constexpr int DVP = DVT * 32;
    static constexpr int KROW = DKP * 2 + 16;         // LDS row stride of the K tile (odd # of 16-B slots)
    static constexpr int VROW = 64 * 2 + 16;          // LDS row stride of the V^T tile (64 keys)
    // VRM: row stride of the row-major V tile (one row per key, DVP columns).  A transposing read covers 4 rows x 64 bytes
    // per 32 lanes: conflict-free when the four rows start 64 bytes apart modulo the 256-byte bank row, i.e. the stride is
    // an odd multiple of 64 bytes: 192 (dh 40 / 64: 128 bytes of columns), 192 (dh 80: 192), 320 (dh 160: 320)
    static constexpr int VRS = (DVP * 2) % 128 == 64 ? DVP * 2 : DVP * 2 + 64;
    static constexpr int K_BYTES = 64 * KROW;
    static constexpr int V_BYTES = VRM ? 64 * VRS : DVP * VROW;
    // one K / V tile pair, or the four waves' Q / O slabs (32 rows x (DH * 2 + 16) bytes each) that alias the same region before
    // the first and after the last tile - whichever is larger (dh 160 with a row-major V: the slabs)
    static constexpr int LDS_BYTES = K_BYTES + V_BYTES > 4 * 32 * (DH * 2 + 16) ? K_BYTES + V_BYTES : 4 * 32 * (DH * 2 + 16);
    static constexpr int CPR = DH / 8;                // 16-B chunks per K row
    static constexpr int KCH = 64 * CPR;              // chunks per K tile
    static constexpr int VCH = DH * 8;                // chunks per V^T tile
    static constexpr int KPT = (KCH + 255) / 256;     // chunks per thread (4-wave workgroup)
    static constexpr int VPT = (VCH + 255) / 256;
};

// RES ("resident K / V^T": the text cross-attention - Lk <= 128 keys, the same K / V^T for every query of a (sample, head)): the
// workgroup stages BOTH key tiles once and walks kResBlocks query blocks over them; the Q / O slabs get LDS of their own, so a query
// block needs no barrier at all, and the next block's Q rows are requested while this one computes.  One query block per workgroup
// (the self-attention form) is a chain of four dependent memory latencies - Q, key tile 0, key tile 1, the stores - for two key
// tiles of work: 8.7 us per workgroup at four workgroups per CU.
#ifndef SDV_ATTN_RES_BLOCKS
#define SDV_ATTN_RES_BLOCKS 8
#endif
constexpr int kResBlocks = SDV_ATTN_RES_BLOCKS;
template <int DH, int QT, bool PRIO, bool LEAN, bool DBUF, bool PP = false, int NW = 4, bool RES = false, bool VRM = false>
__global__ __launch_bounds__(NW * 64, (DH == 64 && VRM && !RES) ? 4 : 1) void attention_kernel(const uint16_t* __restrict__ Q, const uint16_t* __restrict__ Kp,
                                                        const uint16_t* __restrict__ Vt, uint16_t* __restrict__ O,
                                                        int H, int Lq, int Lk, int ldq, int ldk, int ldv, int ldo,
                                                        float scale_log2e, int flags, int BH) {
    const int causal = flags & 1;
    // QT = 32-query tiles per wave: the K / V^T fragments read from LDS are reused for QT MFMAs each.
    // NW = waves per workgroup: one staged K / V^T tile serves NW * QT * 32 queries (staging a tile costs ~17 % of the
    //      4-wave kernel's time in VMEM / LDS-write issue, tools/ubench/build_whatif.py variants 4-6).
    using Cfg = AttnCfg<DH, VRM>;
    static_assert(!(VRM && RES), "the resident (text cross-attention) form takes the walk's precomputed V^T");
    constexpr int NT = NW * 64;
    constexpr int KPT = (Cfg::KCH + NT - 1) / NT, VPT = (Cfg::VCH + NT - 1) / NT;
    static_assert(Cfg::KCH % 64 == 0 && Cfg::VCH % 64 == 0, "staging predicates must be wave-uniform");
    constexpr int DKS = Cfg::DKS, DVT = Cfg::DVT, KROW = Cfg::KROW, VROW = Cfg::VROW, VRS = Cfg::VRS;
    // LEAN softmax - the dh = 40 / 80 kernels are VALU-issue-bound, not MFMA-bound (~150 VALU per 14 MFMAs at dh = 40):
    //  * Q is pre-multiplied by scale*log2(e) when its fragments are loaded, so scores come out of the MFMA in log2 units;
    //  * ONES (dh % 32 != 0, the V^T tile has padding rows): rows DH and DH+4 of the V^T tile hold 1.0, so the PV MFMA
    //    itself accumulates the row sum l = sum_k P into accumulator register ONES_R of the last d-tile of every lane -
    //    no per-score add, and l is rescaled together with O;
    //  * PADM (dh % 16 == 8, the K tile has padding columns): K column DH holds 1.0 and Q' column DH holds -m_ref (the
    //    running max, kept bf16-representable), so the MFMA delivers S' = K.Q'^T - m_ref and P = exp2(S') needs no
    //    per-score subtract either.
    constexpr bool ONES = LEAN && (DH % 32 != 0);
    constexpr int ONES_R = 4 * ((DH % 32) / 8);
    constexpr bool PADM = LEAN && (DH % 16 == 8);
    constexpr int PADM_KS = DH / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // DBUF: two K / V^T tile buffers, tile t+1 is written while tile t is consumed -> ONE barrier per tile
    constexpr int KV_BYTES = Cfg::K_BYTES + Cfg::V_BYTES;
    constexpr int NBUF = (DBUF || RES) ? 2 : 1;
    static_assert(!RES || (!DBUF && !PP && QT == 1), "resident K / V^T: the plain one-query-tile body");
    char* ldsK = smem;
    char* ldsV = smem + Cfg::K_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform (scalar branches on it)
    const int l31 = lane & 31;
    const int lhi = lane >> 5;
    // XCD-aware order: the dispatcher deals consecutive workgroups round-robin to the 8 XCDs (8 private L2s).  All
    // query blocks of one (batch, head) stream the SAME K / V^T (655 KB at 64^2), so they are mapped to ONE XCD, back to
    // back: workgroup w -> XCD w & 7, position i = w >> 3 inside it -> head group i / nqb, query block i % nqb.
    // QUERY-MAJOR order (flags bit 1; the text cross-attention, whose 77 keys are no working set at all): an XCD takes whole
    // (sample, query block) units and runs their H heads back to back.  A token's row interleaves the heads - DH * 2 = 80 bytes each
    // at dh 40 - so with the head-major order every 128-byte line of Q and of O was pulled through (and partially written from)
    // up to eight different L2s.
    const int nqb_all = (Lq + 32 * NW * QT - 1) / (32 * NW * QT);                 // query blocks of one (sample, head)
    const int nqb = RES ? (nqb_all + kResBlocks - 1) / kResBlocks : nqb_all;      // workgroups per (sample, head)
    const int wg_i = blockIdx.x >> 3;
    int h, b, qw;                                                                 // qw: this workgroup's slot among the nqb
    if (flags & 2) {
        const int u = (wg_i / H) * 8 + (blockIdx.x & 7);      // unit = (sample, query block [group])
        if (u * H >= BH * nqb) return;                        // (BH * nqb = units * H; the grid is padded to 8 unit slots)
        h = wg_i % H;
        b = u / nqb;
        qw = u - b * nqb;
    } else {
        const int bh = (wg_i / nqb) * 8 + (blockIdx.x & 7);
        if (bh >= BH) return;                 // BH = batch * heads; the grid is padded to 8 head slots per group
        h = bh % H;
        b = bh / H;
        qw = wg_i % nqb;
    }
    const int qb_first = RES ? qw * kResBlocks : qw;
    const int qb_last = RES ? (qb_first + kResBlocks < nqb_all ? qb_first + kResBlocks : nqb_all) : qb_first + 1;

    // ---- register staging of the next K / V^T tile --------------------------------------------
    u32x4_t kreg[KPT], vreg[VPT];
    const uint16_t* Kb = Kp + (long long)b * Lk * ldk + h * DH;
    // (VRM: `Vt` points at the V columns of the row-major [keys][ldv] buffer - a sample's rows follow each other as K's do)
    const uint16_t* Vb = VRM ? Vt + (long long)b * Lk * ldv + h * DH : Vt + ((long long)b * H + h) * DH * ldv;
    // (chunk indices are clamped instead of predicated: surplus threads re-load / re-store the last
    //  chunk with identical data, which keeps the staging registers free of divergent control flow)
    // Buffer loads: the per-lane byte offsets are tile-invariant (computed once, here) and the tile position travels in
    // the SCALAR offset, so staging a tile costs no VALU address arithmetic (the what-if build without staging ran 19 %
    // faster - most of that was 64-bit address math, clamps and waits, not bandwidth).  Keys beyond Lk fall outside
    // num_records and read as zero.
    const __amdgpu_buffer_rsrc_t rs_k =
        __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (int)(((long long)(Lk - 1) * ldk + DH) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_v = __builtin_amdgcn_make_buffer_rsrc(
        (void*)Vb, 0, VRM ? (int)(((long long)(Lk - 1) * ldv + DH) * 2) : (int)((long long)DH * ldv * 2), 0x00020000);
    int kvo[KPT], vvo[VPT];
    // chunk c of a tile belongs to thread c % NT; KCH and VCH are multiples of 64, so "c < KCH" is the same for all
    // lanes of a wave: surplus WAVES skip their loads / stores altogether (no divergence, no redundant traffic)
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const int c = tid + i * NT;
        const int row = c / Cfg::CPR, cc = c - row * Cfg::CPR;
        kvo[i] = (row * ldk + cc * 8) * 2;
    }
#pragma unroll
    for (int i = 0; i < VPT; ++i) {
        const int c = tid + i * NT;
        if constexpr (VRM) {
            const int row = c / Cfg::CPR, cc = c - row * Cfg::CPR;   // (key row, 16-byte piece of its DH columns) - as K
            vvo[i] = (row * ldv + cc * 8) * 2;
        } else {
            const int row = c >> 3, cc = c & 7;
            vvo[i] = (row * ldv + cc * 8) * 2;
        }
    }
    auto load_tile = [&](int kv0) {
        const int ks_off = kv0 * ldk * 2, vs_off = VRM ? kv0 * ldv * 2 : kv0 * 2;
#pragma unroll
        for (int i = 0; i < KPT; ++i)
            if (wave * 64 + i * NT < Cfg::KCH)
                kreg[i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_k, kvo[i], ks_off, 0));
        // VRM, ragged last tile: the rows past this sample's last key belong to the NEXT sample (or lie past the tensor) and the
        // scalar tile offset is not part of the range check - their pieces get a lane offset beyond num_records and read as zeros
        // (P is 0 for those keys, but 0 x an Inf / NaN bit pattern would not be).  Wave-uniform branch, last tile only.
        const bool ragged = VRM && kv0 + 64 > Lk;
#pragma unroll
        for (int i = 0; i < VPT; ++i)
            if (wave * 64 + i * NT < Cfg::VCH) {
                int off = vvo[i];
                if (ragged) off = kv0 + (tid + i * NT) / Cfg::CPR < Lk ? off : (int)0x80000000;
                vreg[i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs_v, off, vs_off, 0));
            }
    };
    auto store_tile = [&](int boff) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const int c = tid + i * NT;
            const int row = c / Cfg::CPR, cc = c - row * Cfg::CPR;
            if (wave * 64 + i * NT < Cfg::KCH) *(u32x4_t*)(ldsK + boff + row * KROW + cc * 16) = kreg[i];
        }
        // V^T row d holds 64 keys; within each 16-key block the 4-key groups are stored in the order
        // [0-3][8-11][4-7][12-15] so that one ds_read_b128 at (block*16 + lhi*8) keys yields exactly
        // the keys this lane's P registers hold (MFMA 32x32 C-layout: key = (r&3) + 8*(r>>2) + 4*lhi).
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int c = tid + i * NT;
            if constexpr (VRM) {     // row-major V: the rows as they come, one 16-byte piece per chunk (the read transposes)
                const int row = c / Cfg::CPR, cc = c - row * Cfg::CPR;
                if (wave * 64 + i * NT < Cfg::VCH) *(u32x4_t*)(ldsV + boff + row * VRS + cc * 16) = vreg[i];
                continue;
            }
            const int row = c >> 3, cc = c & 7;  // cc: 8-key chunk; block = cc>>1, half = cc&1
            if (wave * 64 + i * NT < Cfg::VCH) {
                char* dst = ldsV + boff + row * VROW + (cc >> 1) * 32 + (cc & 1) * 8;
                *(u32x2_t*)(dst) = u32x2_t{vreg[i][0], vreg[i][1]};        // keys +0..3
                *(u32x2_t*)(dst + 16) = u32x2_t{vreg[i][2], vreg[i][3]};   // keys +4..7
            }
        }
    };

    auto init_pads = [&]() {
        // zero the LDS pads once: K columns [DH, DKP) and V^T rows [DH, DVP) are never rewritten
        for (int i = tid; i < NBUF * KV_BYTES / 16; i += NT) ((uint4*)smem)[i] = make_uint4(0, 0, 0, 0);
        if constexpr (ONES) {
            __syncthreads();
            for (int bf = 0; bf < NBUF; ++bf) {
                if constexpr (VRM) {   // the same two rows of V^T are COLUMNS DH and DH + 4 of every key row here
                    if (tid < 128) *(uint16_t*)(ldsV + bf * KV_BYTES + (tid >> 1) * VRS + (DH + 4 * (tid & 1)) * 2) = 0x3F80;
                } else
                if (tid < 16)   // rows DH (lanes 0-31 hold it in acc register ONES_R) and DH + 4 (lanes 32-63), 64 keys each
                    *(uint4*)(ldsV + bf * KV_BYTES + (DH + 4 * (tid >> 3)) * VROW + (tid & 7) * 16) =
                        make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u);
                if constexpr (PADM)
                    if (tid < 64) *(uint16_t*)(ldsK + bf * KV_BYTES + tid * KROW + DH * 2) = 0x3F80;   // K[key][DH] = 1.0
            }
        }

    };
    // A operand of one PV MFMA: d-tile dt (32 rows of O^T), key group ju = (32-key subtile j, register half u) - this lane's 8
    // keys are j*32 + 16u + 4 lhi + {0..3} and the same + 8 (the MFMA C layout of its P registers).
    // VRM: lane (g = lane >> 4, i = lane & 15) of a transposing read addresses 4 columns [16 (g & 1) + 4 (i & 3), +4) of key row
    // 4 lhi + (i >> 2) of the block and RECEIVES, for its own column d = 16 (g & 1) + i, the four keys of the block's rows.
    const int vtr_lane = (4 * lhi + ((lane & 15) >> 2)) * VRS + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    auto vfrag_at = [&](int boff, int dt, int ju) __attribute__((always_inline)) -> bf16x8_t {
        if constexpr (VRM) {
            typedef short __attribute__((ext_vector_type(4))) s16x4_t;
            typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_p;
            const char* a = ldsV + boff + vtr_lane + ((ju >> 1) * 32 + (ju & 1) * 16) * VRS + dt * 64;
            const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(a));
            const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(a + 8 * VRS));
            typedef short __attribute__((ext_vector_type(8))) s16x8_t;
            return __builtin_bit_cast(bf16x8_t, s16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]});
        } else {
            return *(const bf16x8_t*)(ldsV + boff + (dt * 32 + l31) * VROW + ju * 32 + lhi * 16);
        }
    };
    const int ntiles = (Lk + 63) / 64;
    if constexpr (RES) {
        // both key tiles -> LDS, ONCE per workgroup (ntiles <= 2: the launcher sends Lk <= 128 here)
        init_pads();
        __syncthreads();
        for (int t = 0; t < ntiles; ++t) {
            load_tile(t * 64);
            store_tile(t * KV_BYTES);
        }
        __syncthreads();
    }
    constexpr int QNI = (32 * Cfg::CPR + 63) / 64;   // 16-byte pieces per lane of a 32-query tile
    u32x4_t qraw[RES ? QNI : 1];
    auto load_q_raw = [&](int qblk) {
        if constexpr (RES) {
            const int qbase = (qblk * NW + wave) * 32;
#pragma unroll
            for (int i = 0; i < QNI; ++i) {
                const int c = lane + 64 * i;
                if (c < 32 * Cfg::CPR) {
                    const int row = c / Cfg::CPR, cc = c - row * Cfg::CPR;
                    int q = qbase + row;
                    q = q < Lq ? q : Lq - 1;
                    qraw[i] = *(const u32x4_t*)(Q + ((long long)b * Lq + q) * ldq + h * DH + cc * 8);
                }
            }
        }
    };
    for (int qb = qb_first; qb < qb_last; ++qb) {
    const int q0 = (qb * NW + wave) * (32 * QT);
    // ---- Q fragments (B operand): lane owns query rows q0 + qt*32 + l31, d = ks*16 + lhi*8 .. +7 ----
    // The rows travel through a WAVE-PRIVATE slab of the (not yet initialised) K / V^T region: loaded so that adjacent lanes hold
    // adjacent 16-byte pieces of a row (DH / 8 lanes per row, ~13 rows per instruction), read back in the fragment layout.  Loaded
    // straight in the fragment layout every instruction touched 32 different rows - 16 bytes from each - and so did the stores of the
    // epilogue: for the text cross-attention (two key tiles per query block) that was HALF the kernel's time (timing-only builds
    // without the stores / with one row per wave instruction: 0.63 -> 0.45 / 0.48 ms at the 64^2 level, tools/ubench/build_whatif.py 7, 8).
    constexpr int QROW = DH * 2 + 16;                 // slab row stride (bytes)
    constexpr int QSLAB = 32 * QROW;                  // one 32-query tile
    constexpr int QCH = 32 * Cfg::CPR;                // 16-byte pieces of a 32-query tile
    static_assert(RES || NW * QSLAB <= (NBUF == 1 ? Cfg::LDS_BYTES : NBUF * KV_BYTES), "the Q / O slabs alias the K / V^T buffers");
    // (dh = 64 keeps the direct accesses: the slab code cost it 6 VGPRs - 124 -> 130 - and with them the fourth wave per SIMD;
    //  -DSDV_ATTN_SLAB64=1 builds it the other way for A/B)
    constexpr bool SLAB_IO = DH != 64 || SDV_ATTN_SLAB64 || RES;
    char* const qslab = smem + (RES ? NBUF * KV_BYTES : 0) + wave * QSLAB;     // (RES: the key tiles stay - the slabs sit behind them)
    bf16x8_t qf[QT][DKS];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        if constexpr (!SLAB_IO) {
            int q = q0 + qt * 32 + l31;
            q = q < Lq ? q : Lq - 1;
            const uint16_t* qrow = Q + ((long long)b * Lq + q) * ldq + h * DH;
#pragma unroll
            for (int ks = 0; ks < DKS; ++ks) {
                const int d0 = ks * 16 + lhi * 8;
                if (d0 < DH) {
                    const u32x4_t raw = *(const u32x4_t*)(qrow + d0);
                    if constexpr (LEAN) {
                        u32x4_t sc;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            sc[e] = pack_bf16x2(__builtin_bit_cast(float, raw[e] << 16) * scale_log2e,
                                                __builtin_bit_cast(float, raw[e] & 0xFFFF0000u) * scale_log2e);
                        qf[qt][ks] = __builtin_bit_cast(bf16x8_t, sc);
                    } else {
                        qf[qt][ks] = __builtin_bit_cast(bf16x8_t, raw);
                    }
                } else {
                    qf[qt][ks] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
                }
            }
            continue;
        }
        if constexpr (RES) {
            if (qb == qb_first) load_q_raw(qb);
#pragma unroll
            for (int i = 0; i < QNI; ++i) {
                const int c = lane + 64 * i;
                if (c < QCH) *(u32x4_t*)(qslab + (c / Cfg::CPR) * QROW + (c % Cfg::CPR) * 16) = qraw[i];
            }
            if (qb + 1 < qb_last) load_q_raw(qb + 1);     // in flight beside this block's MFMAs and exponentials
        } else {
#pragma unroll 2
        for (int i = 0; i < (QCH + 63) / 64; ++i) {
            const int c = lane + 64 * i;
            if (c < QCH) {
                const int row = c / Cfg::CPR, cc = c - row * Cfg::CPR;
                int q = q0 + qt * 32 + row;
                q = q < Lq ? q : Lq - 1;
                *(u32x4_t*)(qslab + row * QROW + cc * 16) = *(const u32x4_t*)(Q + ((long long)b * Lq + q) * ldq + h * DH + cc * 8);
            }
        }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (a wave's DS ops execute in order; this keeps the compiler in order too)
#pragma unroll
        for (int ks = 0; ks < DKS; ++ks) {
            const int d0 = ks * 16 + lhi * 8;
            if (d0 < DH) {
                const u32x4_t raw = *(const u32x4_t*)(qslab + l31 * QROW + d0 * 2);
                if constexpr (LEAN) {
                    u32x4_t sc;
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        sc[e] = pack_bf16x2(__builtin_bit_cast(float, raw[e] << 16) * scale_log2e,
                                            __builtin_bit_cast(float, raw[e] & 0xFFFF0000u) * scale_log2e);
                    qf[qt][ks] = __builtin_bit_cast(bf16x8_t, sc);
                } else {
                    qf[qt][ks] = __builtin_bit_cast(bf16x8_t, raw);
                }
            } else {
                qf[qt][ks] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the next query tile re-uses the slab
    }
    if constexpr (!RES) {
        if constexpr (SLAB_IO) __syncthreads();   // every wave has its Q fragments: the region can be initialised for K / V^T
        init_pads();
    }
    f32x16_t o[QT][DVT];
    float m_run[QT], l_run[QT];  // l_run: this lane's partial row sum (its 32 of the 64 keys per tile)
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        m_run[qt] = PADM ? 0.f : -INFINITY;   // PADM: m_ref, what Q' column DH currently subtracts
        l_run[qt] = 0.f;
#pragma unroll
        for (int t = 0; t < DVT; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) o[qt][t][e] = 0.f;
    }

    if constexpr (!RES) {
    load_tile(0);
    if constexpr (DBUF) {
        __syncthreads();   // pad initialisation visible / ordered before the first tile lands
        store_tile(0);
        if (ntiles > 1) load_tile(64);
        __syncthreads();
    }
    }
    for (int t = 0; t < ntiles; ++t) {
        int boff = 0;
        if constexpr (RES) {
            boff = t * KV_BYTES;                      // both tiles are resident: nothing to stage, no barrier
        } else if constexpr (DBUF) {
            boff = (t & 1) * KV_BYTES;
            if (t + 1 < ntiles) {
                store_tile(KV_BYTES - boff);          // tile t+1 -> the buffer tile t-1 was read from (barrier below)
                if (t + 2 < ntiles) load_tile((t + 2) * 64);
            }
        } else {
            __syncthreads();  // previous tile fully consumed (and the pad zeroing is visible)
            store_tile(0);
            __syncthreads();
            if (t + 1 < ntiles) load_tile((t + 1) * 64);
        }

        if constexpr (PP) {
            // ---- software-pipelined body (dh = 40, two 32-query tiles per wave, full key tiles, no mask) -------------
            // The two query tiles are skewed by half a step so that every VALU phase of one has MFMAs of the other to
            // hide behind (the softmax here is ~95 VALU per 14 MFMAs - far above what one wave can hide per MFMA):
            //   S0 = K.Q0 | S1 = K.Q1 || max(S0) | { P0 quarter = exp2(S0 quarter); O0 += V.P0 quarter } x 4 | max(S1) |
            //   { P1 quarter = exp2(S1 quarter); O1 += V.P1 quarter } x 4
            static_assert(QT == 2 && PADM && ONES, "PP: dh = 40 LEAN kernel with two query tiles");
            bf16x8_t kfr[2][DKS];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int ks = 0; ks < DKS; ++ks)
                    kfr[j][ks] = *(const bf16x8_t*)(ldsK + boff + (j * 32 + l31) * KROW + (ks * 16 + lhi * 8) * 2);
            auto qk = [&](int qt, int j) {
                f32x16_t acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[j][0], qf[qt][0], kZero16, 0, 0, 0);
#pragma unroll
                for (int ks = 1; ks < DKS; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[j][ks], qf[qt][ks], acc, 0, 0, 0);
                return acc;
            };
            auto tile_max = [&](const f32x16_t& a, const f32x16_t& b) {
                float mx = a[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) mx = fmaxf(mx, a[r]);
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, b[r]);
                return fmaxf(mx, __shfl_xor(mx, 32));
            };
            auto rescale = [&](int qt, float mx, f32x16_t& a, f32x16_t& b) {   // rare: the running max moved by > 2^kDefer
                const float want = m_run[qt] + (t == 0 ? mx : fmaxf(mx, 0.f));
                const uint32_t mbits = pack_bf16x2(want, 0.f) << 16;
                const float m_new = __builtin_bit_cast(float, mbits);
                const float delta = m_new - m_run[qt];
                const float alpha = __builtin_amdgcn_exp2f(-delta);
                m_run[qt] = m_new;
                u32x4_t qw = __builtin_bit_cast(u32x4_t, qf[qt][PADM_KS]);
                qw[0] = lhi ? ((mbits ^ 0x80000000u) >> 16) : qw[0];
                qf[qt][PADM_KS] = __builtin_bit_cast(bf16x8_t, qw);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    a[r] -= delta;
                    b[r] -= delta;
                }
#pragma unroll
                for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[qt][dt][e] *= alpha;
            };
            // V fragments: with a row-major V every fragment costs TWO transposing reads, and the loop is instruction-issue bound -
            // read each fragment ONCE per key tile and use it for both query tiles (32 more live registers: 198 -> ~230, still two
            // waves per SIMD) instead of once per query tile (tools/attn_vrm_ab.py: the per-query-tile reads cost the 64 x 64 level 7 %)
            bf16x8_t vfr[VRM ? DVT : 1][VRM ? 4 : 1];
            if constexpr (VRM) {
#pragma unroll
                for (int ju = 0; ju < 4; ++ju)
#pragma unroll
                    for (int dt = 0; dt < DVT; ++dt) vfr[dt][ju] = vfrag_at(boff, dt, ju);
            }
            auto vfrag = [&](int dt, int ju) __attribute__((always_inline)) {
                if constexpr (VRM) return vfr[dt][ju];
                else return vfrag_at(boff, dt, ju);
            };
            auto exp_quarter = [&](const f32x16_t& a, int u) {                  // 8 scores -> one B fragment
                u32x4_t pr;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    pr[e] = pack_bf16x2(__builtin_amdgcn_exp2f(a[8 * u + 2 * e]), __builtin_amdgcn_exp2f(a[8 * u + 2 * e + 1]));
                return __builtin_bit_cast(bf16x8_t, pr);
            };
            // P quarter ju = (key half j, register half u) feeds both d-tiles straight away: the v_exp_f32 stream (the
            // transcendental unit costs ~11 cycles per wave instruction, tools/ubench/valu_rate.hip) runs beside the
            // PV MFMAs of the same query tile instead of in front of them
            auto softmax_pv = [&](int qt, const f32x16_t& a, const f32x16_t& b) {
#pragma unroll
                for (int ju = 0; ju < 4; ++ju) {
                    const bf16x8_t pq = exp_quarter(ju < 2 ? a : b, ju & 1);
#pragma unroll
                    for (int dt = 0; dt < DVT; ++dt)
                        o[qt][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfrag(dt, ju), pq, o[qt][dt], 0, 0, 0);
                }
            };
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
            f32x16_t s0a = qk(0, 0), s0b = qk(0, 1);
            f32x16_t s1a = qk(1, 0), s1b = qk(1, 1);                           // in flight beside max(S0)
            float mx = tile_max(s0a, s0b);
            if (t == 0 || !__all(mx <= kDefer)) rescale(0, mx, s0a, s0b);
            softmax_pv(0, s0a, s0b);
            mx = tile_max(s1a, s1b);
            if (t == 0 || !__all(mx <= kDefer)) rescale(1, mx, s1a, s1b);
            softmax_pv(1, s1a, s1b);
            if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
            if constexpr (DBUF) __syncthreads();   // tile t+1 visible; everyone is done reading tile t's buffer
            continue;
        }
        // ---- S^T = K . Q^T for two 32-key subtiles (each K fragment feeds QT MFMAs) ----
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);   // matrix-pipe clusters win issue arbitration (guide T5)
        f32x16_t s[QT][2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int ks = 0; ks < DKS; ++ks) {
                const bf16x8_t kf = *(const bf16x8_t*)(ldsK + boff + (j * 32 + l31) * KROW + (ks * 16 + lhi * 8) * 2);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) {
                    if (ks == 0)
                        s[qt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qt][0], kZero16, 0, 0, 0);   // C = inline 0
                    else
                        s[qt][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qt][ks], s[qt][j], 0, 0, 0);
                }
            }
        }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        // ---- online softmax (base-2), lane-local.  Raw scores stay unscaled; the softmax scale is folded into
        //      the exp2 argument with one FMA.  Keys beyond Lk only exist in the last tile (uniform branch). ----
        const int kv0 = t * 64;
        bf16x8_t pf[QT][4];
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) {
            if (kv0 + 64 > Lk || causal) {
                // causal (CLIP text encoder): key <= query.  Key 0 is visible to every query, so the first tile always
                // sets a finite running max; later tiles may be fully masked for a lane (all P = 0).
                const int klim = causal ? min(Lk, q0 + qt * 32 + l31 + 1) : Lk;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int key = kv0 + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                        s[qt][j][r] = key < klim ? s[qt][j][r] : -INFINITY;
                    }
            }
            float mx = s[qt][0][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[qt][0][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qt][1][r]);
            if constexpr (PADM) {
                // scores arrive relative to m_ref (0 before the first tile, which is forced to set it): mx is how far
                // this tile rises above it.  m_ref stays bf16-representable so that Q' column DH holds it exactly.
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                if (t == 0 || !__all(mx <= kDefer)) {
                    const float want = m_run[qt] + (t == 0 ? mx : fmaxf(mx, 0.f));
                    const uint32_t mbits = pack_bf16x2(want, 0.f) << 16;
                    const float m_new = __builtin_bit_cast(float, mbits);
                    const float delta = m_new - m_run[qt];
                    const float alpha = __builtin_amdgcn_exp2f(-delta);
                    m_run[qt] = m_new;
                    u32x4_t qw = __builtin_bit_cast(u32x4_t, qf[qt][PADM_KS]);
                    qw[0] = lhi ? ((mbits ^ 0x80000000u) >> 16) : qw[0];           // element DH of Q' = -m_ref
                    qf[qt][PADM_KS] = __builtin_bit_cast(bf16x8_t, qw);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) s[qt][j][r] -= delta;
#pragma unroll
                    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) o[qt][dt][e] *= alpha;
                }
            } else if constexpr (LEAN) {
                mx = fmaxf(mx, __shfl_xor(mx, 32));
                if (!__all(mx - m_run[qt] <= kDefer)) {
                    const float m_new = fmaxf(m_run[qt], mx);
                    const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
                    m_run[qt] = m_new;
                    if constexpr (!ONES) l_run[qt] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) o[qt][dt][e] *= alpha;
                }
            } else {
                mx = fmaxf(mx, __shfl_xor(mx, 32)) * scale_log2e;      // scale > 0: max commutes with it
                // deferred rescale: keep the old running max while the tile max grows by < kDefer (P stays <= 2^kDefer,
                // exact in fp32 accumulation); when it fires, O and l are rescaled BEFORE this tile's P exists.
                if (!__all(mx - m_run[qt] <= kDefer)) {
                    const float m_new = fmaxf(m_run[qt], mx);
                    const float alpha = __builtin_amdgcn_exp2f(m_run[qt] - m_new);
                    m_run[qt] = m_new;
                    l_run[qt] *= alpha;
#pragma unroll
                    for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
                        for (int e = 0; e < 16; ++e) o[qt][dt][e] *= alpha;
                }
            }
            float psum = 0.f;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float pv[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        if constexpr (PADM)
                            pv[e] = __builtin_amdgcn_exp2f(s[qt][j][8 * u + e]);
                        else if constexpr (LEAN)
                            pv[e] = __builtin_amdgcn_exp2f(s[qt][j][8 * u + e] - m_run[qt]);
                        else
                            pv[e] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[qt][j][8 * u + e], scale_log2e, -m_run[qt]));
                        if constexpr (!ONES) psum += pv[e];
                    }
                    u32x4_t pr;
                    pr[0] = pack_bf16x2(pv[0], pv[1]);
                    pr[1] = pack_bf16x2(pv[2], pv[3]);
                    pr[2] = pack_bf16x2(pv[4], pv[5]);
                    pr[3] = pack_bf16x2(pv[6], pv[7]);
                    pf[qt][j * 2 + u] = __builtin_bit_cast(bf16x8_t, pr);
                }
            if constexpr (!ONES) l_run[qt] += psum;
        }
        // ---- O^T += V^T . P^T (each V^T fragment feeds QT MFMAs) ----
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
            for (int ju = 0; ju < 4; ++ju) {
                const bf16x8_t vf = vfrag_at(boff, dt, ju);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt)
                    o[qt][dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[qt][ju], o[qt][dt], 0, 0, 0);
            }
        if constexpr (PRIO) __builtin_amdgcn_s_setprio(0);
        if constexpr (DBUF) __syncthreads();   // tile t+1 visible; everyone is done reading tile t's buffer
    }

    // ---- normalise and store O[q][h*DH + d]: through the wave's slab, 16 bytes per lane, adjacent lanes in one row ----
    if constexpr (SLAB_IO && !RES) __syncthreads();   // every wave is done with the last K / V^T tile: the region is the slabs' again
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l_tot;
        if constexpr (ONES)
            l_tot = o[qt][DVT - 1][ONES_R];
        else
            l_tot = l_run[qt] + __shfl_xor(l_run[qt], 32);
        const float inv = 1.0f / l_tot;
        if constexpr (!SLAB_IO) {
            const int q = q0 + qt * 32 + l31;
            if (q < Lq) {
                uint16_t* orow = O + ((long long)b * Lq + q) * ldo + h * DH;
#pragma unroll
                for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        const int d = dt * 32 + 8 * g4 + 4 * lhi;
                        if (d < DH) {
                            uint2 w;
                            w.x = pack_bf16x2(o[qt][dt][4 * g4 + 0] * inv, o[qt][dt][4 * g4 + 1] * inv);
                            w.y = pack_bf16x2(o[qt][dt][4 * g4 + 2] * inv, o[qt][dt][4 * g4 + 3] * inv);
                            *(uint2*)(orow + d) = w;
                        }
                    }
            }
            continue;
        }
#pragma unroll
        for (int dt = 0; dt < DVT; ++dt)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d = dt * 32 + 8 * g4 + 4 * lhi;
                if (d < DH)
                    *(u32x2_t*)(qslab + l31 * QROW + d * 2) = u32x2_t{pack_bf16x2(o[qt][dt][4 * g4 + 0] * inv, o[qt][dt][4 * g4 + 1] * inv),
                                                                     pack_bf16x2(o[qt][dt][4 * g4 + 2] * inv, o[qt][dt][4 * g4 + 3] * inv)};
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll 2
        for (int i = 0; i < (QCH + 63) / 64; ++i) {
            const int c = lane + 64 * i;
            const int row = c / Cfg::CPR, cc = c - row * Cfg::CPR;
            const int q = q0 + qt * 32 + row;
            if (c < QCH && q < Lq)
                *(u32x4_t*)(O + ((long long)b * Lq + q) * ldo + h * DH + cc * 8) = *(const u32x4_t*)(qslab + row * QROW + cc * 16);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the next query tile re-uses the slab
    }
    }   // query blocks
}

template <int DH, int QT, bool VRM>
int launch_attention_q(const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, uint16_t* O, int B, int H, int Lq, int Lk,
                       int ldq, int ldk, int ldv, int ldo, float scale, int causal, hipStream_t s) {
    using Cfg = AttnCfg<DH, VRM>;
    const int nqb = (Lq + 128 * QT - 1) / (128 * QT);
    // (see the kernel: query-major order + resident key tiles for the text cross-attention, whose V^T is made once per walk)
    const bool qmajor = !VRM && !causal && Lk <= 128;
    dim3 grid(qmajor ? (unsigned)(((B * nqb + 7) / 8) * 8 * H) : (unsigned)(nqb * (((B * H + 7) / 8) * 8)));   // 1-D over (head group, query block, XCD slot)
    const int flags = (causal ? 1 : 0) | (qmajor ? 2 : 0);
    // Variants measured and not shipped (the template flags remain so that a tools build can instantiate them): one barrier per
    // tile with two K / V^T buffers (DBUF) -4 % at dh = 40 (8 more VGPRs -> 3 waves / SIMD), +-0 at dh = 80; no s_setprio
    // around the MFMA blocks (PRIO = false) -1..3 %; 8 waves per workgroup (NW = 8, one staged tile serves 512 queries) +-0.
    const int lds = Cfg::LDS_BYTES;
    constexpr bool lean = DH % 32 != 0;   // dh = 64 / 160 have no padding rows or columns to move softmax work into
    const float sl = scale * 1.4426950408889634f;   // (the caller passes 1 / log2(e) for a pre-scaled Q: sl == 1, the
                                                    //  kernels' own Q scaling then reproduces the bf16 values bit for bit)
#define SDV_ATTN_LAUNCH(P, L, D) \
    hipLaunchKernelGGL((attention_kernel<DH, QT, P, L, D, false, 4, false, VRM>), grid, dim3(256), lds, s, Q, K, Vt, O, H, Lq, Lk, ldq, ldk, \
                       ldv, ldo, sl, flags, B * H)
    if constexpr (QT == 1 && DH <= 80 && !VRM) {   // (dh 160: the resident form needs 276 registers - one wave per SIMD - and is not built)
    if (qmajor) {
        // text cross-attention: both key tiles resident, kResBlocks query blocks per workgroup, query-major order
        const int nwg = (nqb + kResBlocks - 1) / kResBlocks;
        dim3 grid_r((unsigned)(((B * nwg + 7) / 8) * 8 * H));
        constexpr int lds_r = 2 * (Cfg::K_BYTES + Cfg::V_BYTES) + 4 * 32 * (DH * 2 + 16);
        static_assert(lds_r <= 160 * 1024, "resident K / V^T + slabs must fit the LDS");
        auto kern = attention_kernel<DH, 1, true, lean, false, false, 4, true>;
        if (lds_r > 64 * 1024) {
            static unsigned long long attr_set = 0;            // one bit per device
            int dev = 0;
            (void)hipGetDevice(&dev);
            if (!(attr_set & (1ull << dev))) {
                (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_r);
                attr_set |= 1ull << dev;
            }
        }
        hipLaunchKernelGGL(kern, grid_r, dim3(256), lds_r, s, Q, K, Vt, O, H, Lq, Lk, ldq, ldk, ldv, ldo, sl, flags, B * H);
        SDV_CHECK_LAUNCH("sdv_attention_bf16");
        return SDV_OK;
    }
    }
    if constexpr (QT == 2) {
        // software-pipelined two-query-tile kernel: dh = 40, full key tiles only, no mask (the 64^2 self-attention; the caller
        // checks).  Two tiles per wave WITHOUT the pipelined body measured -3 % .. +1 % and are not compiled.
        static_assert(DH == 40, "two query tiles per wave exist for dh = 40 only");
        // (SDV_ATTN_PP_DBUF: two K / V tile buffers, one barrier per key tile instead of two - an A/B switch for tools builds)
        constexpr bool dbuf = SDV_ATTN_PP_DBUF != 0;
        constexpr int lds2 = dbuf ? 2 * (Cfg::K_BYTES + Cfg::V_BYTES) : Cfg::LDS_BYTES;
        hipLaunchKernelGGL((attention_kernel<DH, QT, true, true, dbuf, true, 4, false, VRM>), grid, dim3(256), lds2, s, Q, K, Vt, O, H, Lq,
                           Lk, ldq, ldk, ldv, ldo, sl, flags, B * H);
    } else {
        SDV_ATTN_LAUNCH(true, lean, false);
    }
#undef SDV_ATTN_LAUNCH
    SDV_CHECK_LAUNCH("sdv_attention_bf16");
    return SDV_OK;
}

template <int DH, bool VRM>
int launch_attention(const uint16_t* Q, const uint16_t* K, const uint16_t* Vt, uint16_t* O, int B, int H, int Lq, int Lk,
                     int ldq, int ldk, int ldv, int ldo, float scale, int causal, hipStream_t s) {
    // Two 32-query tiles per wave (halves the LDS fragment traffic and barriers per MFMA, 2 waves / SIMD) with the
    // software-pipelined body: +5.5 % on the 64^2 self-attention (dh = 40, full key tiles, no mask).
    if constexpr (DH == 40) {
        const bool pp_ok = !causal && Lk % 64 == 0;
        if (pp_ok && Lq >= 1024) return launch_attention_q<DH, 2, VRM>(Q, K, Vt, O, B, H, Lq, Lk, ldq, ldk, ldv, ldo, scale, causal, s);
    }
    return launch_attention_q<DH, 1, VRM>(Q, K, Vt, O, B, H, Lq, Lk, ldq, ldk, ldv, ldo, scale, causal, s);
}

// ---- in-place row softmax over bf16 (VAE mid-block attention scores) -------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(uint16_t* __restrict__ S, int cols, int ld) {
    __shared__ float red[8];
    uint16_t* row = S + (long long)blockIdx.x * ld;
    const int tid = threadIdx.x;
    float mx = -INFINITY;
    for (int c = tid * 8; c < cols; c += 256 * 8) {
        const bf16x8_raw r = *(const bf16x8_raw*)(row + c);
        float f[8];
        unpack8(r, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, f[e]);
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int c = tid * 8; c < cols; c += 256 * 8) {
        const bf16x8_raw r = *(const bf16x8_raw*)(row + c);
        float f[8];
        unpack8(r, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += __expf(f[e] - mx);
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    for (int c = tid * 8; c < cols; c += 256 * 8) {
        const bf16x8_raw r = *(const bf16x8_raw*)(row + c);
        float f[8];
        unpack8(r, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = __expf(f[e] - mx) * inv;
        *(bf16x8_raw*)(row + c) = pack8(f);
    }
}

// ---- row softmax of fp32 scores into bf16 probabilities (VAE mid-block attention: the scores leave the Q K^T GEMM unrounded) ----
__global__ __launch_bounds__(256) void softmax_rows_f32_kernel(const float* __restrict__ S, uint16_t* __restrict__ P, int cols, int lds, int ldp) {
    __shared__ float red[8];
    const float* row = S + (long long)blockIdx.x * lds;
    uint16_t* prow = P + (long long)blockIdx.x * ldp;
    const int tid = threadIdx.x;
    auto load8 = [&](int c, float* f) {
        const float4 a = *(const float4*)(row + c), b = *(const float4*)(row + c + 4);
        f[0] = a.x, f[1] = a.y, f[2] = a.z, f[3] = a.w, f[4] = b.x, f[5] = b.y, f[6] = b.z, f[7] = b.w;
    };
    float mx = -INFINITY;
    for (int c = tid * 8; c < cols; c += 256 * 8) {
        float f[8];
        load8(c, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) mx = fmaxf(mx, f[e]);
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int c = tid * 8; c < cols; c += 256 * 8) {
        float f[8];
        load8(c, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) sum += __expf(f[e] - mx);
    }
    sum = wave_sum(sum);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
    __syncthreads();
    const float inv = 1.0f / (red[4] + red[5] + red[6] + red[7]);
    for (int c = tid * 8; c < cols; c += 256 * 8) {
        float f[8];
        load8(c, f);
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = __expf(f[e] - mx) * inv;
        *(bf16x8_raw*)(prow + c) = pack8(f);
    }
}

}  // namespace

extern "C" int sdv_attention_bf16(const sdv_bf16* Q, const sdv_bf16* K, const sdv_bf16* Vt, sdv_bf16* O, int32_t B,
                                  int32_t H, int32_t Lq, int32_t Lk, int32_t dh, int32_t ldq, int32_t ldk, int32_t ldv,
                                  int32_t ldo, float scale, int32_t causal, int32_t q_prescaled, int32_t v_rowmajor, void* stream) {
    if (q_prescaled) scale = 0.6931471805599453f;   // * log2(e) == 1
    SDV_REQUIRE(Q && K && Vt && O, "sdv_attention_bf16: null pointer");
    SDV_REQUIRE(B > 0 && H > 0 && Lq > 0 && Lk > 0, "sdv_attention_bf16: bad shape");
    SDV_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0, "sdv_attention_bf16: unaligned leading dims");
    SDV_REQUIRE(dh == 40 || dh == 64 || dh == 80 || dh == 160, "sdv_attention_bf16: unsupported head dim %d (40/64/80/160)", dh);
    SDV_REQUIRE(((((uintptr_t)Q) | ((uintptr_t)O)) & 15) == 0, "sdv_attention_bf16: Q / O must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    if (v_rowmajor) {
        // V row-major [B*Lk][ldv] (the V columns of a fused QKV projection): ldv is the row stride, like ldk
        SDV_REQUIRE(ldv >= H * dh && (((uintptr_t)Vt) & 15) == 0, "sdv_attention_bf16: row-major V needs ldv=%d >= H*dh and a 16-byte aligned pointer", ldv);
        SDV_REQUIRE(((long long)(Lk - 1) * ldv + dh) * 2 < 0x7fffffffLL && ((long long)(Lk - 1) * ldk + dh) * 2 < 0x7fffffffLL,
                    "sdv_attention_bf16: one sample's K / V rows must span less than 2 GiB");
        switch (dh) {
            case 40: return launch_attention<40, true>(Q, K, Vt, O, B, H, Lq, Lk, ldq, ldk, ldv, ldo, scale, causal, s);
            case 64: return launch_attention<64, true>(Q, K, Vt, O, B, H, Lq, Lk, ldq, ldk, ldv, ldo, scale, causal, s);
            case 80: return launch_attention<80, true>(Q, K, Vt, O, B, H, Lq, Lk, ldq, ldk, ldv, ldo, scale, causal, s);
            case 160: return launch_attention<160, true>(Q, K, Vt, O, B, H, Lq, Lk, ldq, ldk, ldv, ldo, scale, causal, s);
            default: SDV_REQUIRE(false, "sdv_attention_bf16: unsupported head dim %d (40/64/80/160)", dh);
        }
    }
    SDV_REQUIRE(ldv >= ((Lk + 63) / 64) * 64, "sdv_attention_bf16: ldv=%d must cover roundup(Lk=%d, 64)", ldv, Lk);
    switch (dh) {
        case 40: return launch_attention<40, false>(Q, K, Vt, O, B, H, Lq, Lk, ldq, ldk, ldv, ldo, scale, causal, s);
        case 64: return launch_attention<64, false>(Q, K, Vt, O, B, H, Lq, Lk, ldq, ldk, ldv, ldo, scale, causal, s);
        case 80: return launch_attention<80, false>(Q, K, Vt, O, B, H, Lq, Lk, ldq, ldk, ldv, ldo, scale, causal, s);
        case 160: return launch_attention<160, false>(Q, K, Vt, O, B, H, Lq, Lk, ldq, ldk, ldv, ldo, scale, causal, s);
        default: SDV_REQUIRE(false, "sdv_attention_bf16: unsupported head dim %d (40/64/80/160)", dh);
    }
    return SDV_OK;
}

extern "C" int sdv_softmax_rows_bf16(sdv_bf16* S, int64_t rows, int32_t cols, int32_t ld, void* stream) {
    SDV_REQUIRE(S && rows > 0 && cols > 0, "sdv_softmax_rows_bf16: bad args");
    SDV_REQUIRE(cols % 8 == 0 && ld % 8 == 0, "sdv_softmax_rows_bf16: cols/ld must be multiples of 8");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, S, cols, ld);
    SDV_CHECK_LAUNCH("sdv_softmax_rows_bf16");
    return SDV_OK;
}

extern "C" int sdv_softmax_rows_f32(const float* S, sdv_bf16* P, int64_t rows, int32_t cols, int32_t lds, int32_t ldp, void* stream) {
    SDV_REQUIRE(S && P && rows > 0 && cols > 0, "sdv_softmax_rows_f32: bad args");
    SDV_REQUIRE(cols % 8 == 0 && lds % 4 == 0 && ldp % 8 == 0 && lds >= cols && ldp >= cols, "sdv_softmax_rows_f32: cols / ldp must be multiples of 8, lds of 4");
    SDV_REQUIRE(((((uintptr_t)S) | ((uintptr_t)P)) & 15) == 0, "sdv_softmax_rows_f32: unaligned pointers");
    hipLaunchKernelGGL(softmax_rows_f32_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, S, P, cols, lds, ldp);
    SDV_CHECK_LAUNCH("sdv_softmax_rows_f32");
    return SDV_OK;
}
