// Fused GEGLU feed-forward of a BasicTransformerBlock on the gfx950 matrix cores, for the C = 320 level of the UNet:
//
//     out = x + b2 + W2 . ( v * gelu(g) ),     [v | g] = LayerNorm(x) W1^T + b1          (norm3 folded into W1: sdv_hip.h "ln_side")
//
// Replaces, inside unet(...) (/root/reference/stable_diffusion_videos/stable_diffusion_pipeline.py:418), diffusers'
// BasicTransformerBlock.norm3 -> ff.net.0 (GEGLU: proj 320 -> 2560, value * gelu(gate)) -> ff.net.2 (1280 -> 320) -> + residual.
// As two igemm launches that pair moved the [tokens, 1280] GEGLU output through HBM (2.7 GB written + read back per block at the
// 64 x 64 level of a 128-frame call) and ran ff.net.0 at 0.24 of the MFMA peak: a K = 320 tile spends more time in its prologue and
// its GEGLU epilogue than in its 5 K slabs.  Here a workgroup owns a PANEL of 128 tokens and keeps everything of it on chip:
//
//   * 4 waves, ONE per SIMD, the whole 512-register file each (launch_bounds(256, 1) -> AGPR-form MFMAs): a wave owns 32 tokens -
//     their x rows as 20 MFMA B fragments (80 registers), the ff.net.2 accumulators of all 320 output channels (10 tiles, 160
//     AGPRs) and the ff.net.0 accumulators of the current 64-channel hidden CHUNK (4 tiles [16 value | 16 gate], 64 AGPRs).
//   * the hidden dimension is walked in 20 chunks of 64 channels, software-pipelined over three stages in ONE instruction stream:
//         C(c-1): ff.net.2 MFMAs of the chunk before   |   A(c+1): ff.net.0 MFMAs of the NEXT chunk   |
//         B(c):   rstd scaling + GEGLU of chunk c on the VALU, from a VGPR copy of its accumulators -> packed bf16.
//     The MFMA C layout (lane = token, 4 consecutive weight rows per accumulator quad) IS a valid B operand layout for the next
//     contraction if the other operand walks K in the same order - W2's K axis is permuted on the host accordingly - so the hidden
//     activations never leave the registers.  With one wave per SIMD nothing else fills the matrix pipe's shadow: the stream is
//     laid out by hand in MICRO-GROUPS of { one W-fragment read three MFMAs ahead, one MFMA, ONE instruction of each of the ~5 GEGLU
//     elements in flight } pinned by sched_barriers - independent fillers in the MFMA's shadow, the guide's one-wave-per-SIMD budget.
//   * the LayerNorm fold's per-column terms ride in the MATRIX product: LN(x) W^T + b = rstd (x W'^T - mean s + b / rstd), and
//     "- mean s + b / rstd" is one more k-step - the token side holds (-mean, 1 / rstd), the weight side (s, b), each split into
//     three bf16 pieces whose six leading cross products carry 24 bits (fp32 accumulate: the sum is what an fp32 FMA pair would
//     give).  Stage B is left with ONE multiply per value, no per-column vector reads and no vector registers.
//   * the weights stream L2 -> LDS by LDS-DMA (`buffer_load ... lds`, 1 KiB per wave instruction) into two rings - five slots for
//     the [128 x 64] K slabs of W1's chunk (the last one also takes the chunk's [128 x 16] fold columns), two 20 KiB slots for the
//     [160 x 64] halves of W2's chunk - in ONE fixed order that is the same for every panel, so the stream never drains: every step
//     refills the slot the step before consumed, one s_barrier per slab, taken one slab EARLY (the barrier of step j certifies slab
//     j + 1, so a step's first fragments are read before its barrier).  W1 + W2 = 2.5 MB: L2-resident on every XCD.
//   * x + b2 initialises the ff.net.2 accumulators (the residual comes out of the x fragments with 40 v_permlane32_swap - the B
//     layout and the C layout differ by one exchange between a token's two lanes); the result leaves as 16-byte stores straight from
//     the accumulator layout (the same exchange the other way round).
#include <type_traits>

#include "sdv_common.h"

namespace {

constexpr int FC = 320;            // channels of the level
constexpr int FH = 4 * FC;         // hidden width of the GEGLU feed-forward
constexpr int NCH = FH / 64;       // hidden chunks
constexpr int SLOT_A = 16384, SLOT_B = 20480, FOLD_BYTES = 128 * 32;
constexpr int FOLD_OFF = 5 * SLOT_A;                 // the fold columns of the chunk sit behind ring A (they live and die with its slot 4)
constexpr int RING_B_OFF = FOLD_OFF + FOLD_BYTES;
constexpr int B2_OFF = RING_B_OFF + 2 * SLOT_B;
constexpr int FFN_LDS = B2_OFF + FC * 4;
static_assert(FFN_LDS <= 160 * 1024, "rings + vectors must fit the 160 KiB LDS");

SDV_DEVICE bf16x8_t kZ16x8() { return bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0}; }
constexpr f32x16_t kZ16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

struct ffn_args {
    const uint16_t* X;
    const uint16_t* W1;
    const uint16_t* W1x;
    const uint16_t* W2p;
    const float* bias2;
    const float* ln_stats;
    uint16_t* out;
    long long M;
    int ldx, ldo;
};

template <int I, int N, typename F>
SDV_DEVICE void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int V>
using ic = std::integral_constant<int, V>;

#define SDV_VMCNT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(n) : "memory")
// TIMING-ONLY what-if builds (tools/ubench/build_ffn_whatif.py; wrong results by construction, never shipped): bit 0 = no stage B (the
// GEGLU VALU work; the copies stay), bit 1 = no weight refills (the rings keep their first contents), bit 2 = no s_barrier, bit 4 = no vmcnt waits
#ifndef SDV_FFN_WHATIF
#define SDV_FFN_WHATIF 0
#endif

__global__ __launch_bounds__(256, 1) void ffn_geglu_kernel(const ffn_args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int npanels = (int)((p.M + 127) >> 7);

    // b2 -> LDS in the order a lane reads it: b2p[(t * 2 + lhi) * 16 + q * 4 + e] = b2[32 t + 8 q + 4 lhi + e]
    for (int i = tid; i < FC; i += 256) {
        const int e = i & 3, q = (i >> 2) & 3, lh = (i >> 4) & 1, t = i >> 5;
        ((float*)(smem + B2_OFF))[i] = p.bias2[32 * t + 8 * q + 4 * lh + e];
    }
    const float* b2p = (const float*)(smem + B2_OFF);

    // ---- weight stream: buffer descriptors + per-lane offsets of this wave's 1-KiB pieces, chunk / slab position in the scalar offset ----
    // LDS image of a slab: rows of 128 bytes (64 K values), the 16-byte chunks of row r XOR-swizzled by (r >> 1) & 7 - applied on the
    // SOURCE offset (the DMA writes lane-linear) and again on the fragment reads: conflict-free ds_read_b128 (the igemm's image).
    // The fold columns are 32-byte rows, unswizzled (4 fragment reads per chunk).
    auto swz = [](int r) { return (r >> 1) & 7; };
    const int rg = lane >> 3, pc = lane & 7;
    unsigned voA[4], voB[5];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (wave + 4 * i) * 8 + rg;
        voA[i] = (unsigned)(r * (FC * 2) + ((pc ^ swz(r)) << 4));
    }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int r = (wave + 4 * i) * 8 + rg;
        voB[i] = (unsigned)(r * (FH * 2) + ((pc ^ swz(r)) << 4));
    }
    const unsigned voF = (unsigned)((wave * 32 + (lane >> 1)) * 32 + (lane & 1) * 16);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)p.W1, 0, 2 * FH * FC * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc((void*)p.W1x, 0, 2 * FH * 32, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)p.W2p, 0, FC * FH * 2, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // piece i of: K slab k of W1's chunk -> slot k of ring A (piece 4 of slot 4: this wave's 32 rows of the chunk's fold columns)
    //             output rows [160 h, +160) of W2's chunk -> slot h of ring B
    auto pieceA = [&](auto K, int chunk, auto I) __attribute__((always_inline)) {
        constexpr int k = decltype(K)::value, i = decltype(I)::value;
        if constexpr (i < 4)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(smem + k * SLOT_A + (wave + 4 * i) * 1024), 16, (int)voA[i],
                                                     chunk * (128 * FC * 2) + k * 128, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsF, (lds_ptr)(smem + FOLD_OFF + wave * 1024), 16, (int)voF, chunk * FOLD_BYTES, 0, 0);
    };
    auto pieceB = [&](auto H, int chunk, auto I) __attribute__((always_inline)) {
        constexpr int h = decltype(H)::value, i = decltype(I)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (lds_ptr)(smem + RING_B_OFF + h * SLOT_B + (wave + 4 * i) * 1024), 16, (int)voB[i],
                                                 h * (160 * FH * 2) + chunk * 128, 0, 0);
    };
    int pa = 0, pb = NCH - 1;   // chunk the next refill of ring A's slot 4 / ring B's slot 0 takes (both wrap at NCH: the stream is periodic)

    // fragment reads: lane (row l31 of a 32-row tile, K half lhi) reads logical chunk 2 kk + lhi of its row
    int fo[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fo[kk] = l31 * 128 + (((2 * kk + lhi) ^ swz(l31)) << 4);
    const int fof = l31 * 32 + lhi * 16;
    // fragment i of a step's slab (i = kk * tiles + t): KIND 0 = slot SL of ring A (4 tiles; slot 4: fragments 16 .. 19 = the fold
    // k-step's four tiles), 1 = slot SL of ring B (5 tiles)
    auto frag = [&](auto KIND, auto SL, auto I) __attribute__((always_inline)) -> bf16x8_t {
        constexpr int T = decltype(KIND)::value == 0 ? 4 : 5, i = decltype(I)::value;
        if constexpr (decltype(KIND)::value == 0 && i >= 16) {
            return *(const bf16x8_t*)(smem + FOLD_OFF + (i - 16) * 1024 + fof);
        } else {
            constexpr int base = decltype(KIND)::value == 0 ? decltype(SL)::value * SLOT_A : RING_B_OFF + decltype(SL)::value * SLOT_B;
            return *(const bf16x8_t*)(smem + base + (i % T) * 4096 + fo[(i / T) & 3]);
        }
    };

    // ---- panel state ---------------------------------------------------------------------------------------------------------
    bf16x8_t xf[21];                 // this wave's 32 tokens x 320 channels as B fragments (k-step s: channels 16 s + 8 lhi .. + 7);
                                     // [20] = the fold k-step: the three-way bf16 splits of (-mean, 1 / rstd) of the lane's token
    float rs = 1.f;                  // rstd of this lane's token (the panel stage B works on)
    float nrs = 1.f;                 // ... of the panel whose x is in xf
    f32x16_t acc1[4], acc2[10];
    f32x16_t g[4];                   // VGPR copy of chunk c's ff.net.0 accumulators (stage B's input)
    u32x4_t hidP[4], hidC[4];        // GEGLU outputs (bf16 pairs) of the previous / current chunk = B fragments of ff.net.2
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        acc1[t] = kZ16;
        g[t] = kZ16;
        hidP[t] = u32x4_t{0, 0, 0, 0};
        hidC[t] = u32x4_t{0, 0, 0, 0};
    }
#pragma unroll
    for (int t = 0; t < 10; ++t) acc2[t] = kZ16;

    // x = h + m + l to 24 bits, each piece a bf16 (round to nearest: |m| <= 2^-9 |h|, |l| <= 2^-18 |h|)
    auto split3 = [](float x, unsigned& h, unsigned& m, unsigned& l) {
        h = pack_bf16x2(x, 0.f) & 0xffffu;
        const float r1 = x - __uint_as_float(h << 16);
        m = pack_bf16x2(r1, 0.f) & 0xffffu;
        l = pack_bf16x2(r1 - __uint_as_float(m << 16), 0.f) & 0xffffu;
    };
    auto load_x = [&](int panel) __attribute__((always_inline)) {
        long long tok = (long long)panel * 128 + wave * 32 + l31;
        tok = tok < p.M ? tok : p.M - 1;
        const uint16_t* row = p.X + tok * p.ldx + 8 * lhi;
#pragma unroll
        for (int s = 0; s < 20; ++s) xf[s] = *(const bf16x8_t*)(row + 16 * s);
        const float2 st = *(const float2*)(p.ln_stats + 2 * tok);
        nrs = st.y;
        // token side of the fold k-step (the weight side - sdv_hip.h W1x - pairs them off so that the 12 live products are the
        // six leading cross terms of (-mean) s and of (1 / rstd) t):
        //   K position:   0    1    2    3    4    5    6    7  |  8    9   10   11   12..15
        //   token:       m_h  m_m  m_h  m_l  m_h  m_m  r_h  r_m | r_h  r_l  r_h  r_m    0        m = -mean, r = 1 / rstd
        //   weight row:  s_h  s_h  s_m  s_h  s_l  s_m  t_h  t_h | t_m  t_h  t_l  t_m    0
        unsigned mh, mm, ml, rh, rm, rl;
        split3(-st.x, mh, mm, ml);
        split3(1.0f / st.y, rh, rm, rl);
        const u32x4_t lo = u32x4_t{mh | (mm << 16), mh | (ml << 16), mh | (mm << 16), rh | (rm << 16)};
        const u32x4_t hi = u32x4_t{rh | (rl << 16), rh | (rm << 16), 0u, 0u};
        xf[20] = __builtin_bit_cast(bf16x8_t, lhi ? hi : lo);
    };
    // acc2 <- x + b2 in the accumulator layout: quad q of tile t = channels 32 t + 8 q + 4 lhi + {0..3}; the fragment of k-step
    // s = 2 t + (q >> 1) holds channels 16 s + 8 lhi + {0..7}: lanes l and l + 32 exchange one register pair
    auto init_acc2 = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < 10; ++t)
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                const u32x4_t d = __builtin_bit_cast(u32x4_t, xf[2 * t + h2]);
                const auto r02 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
                const auto r13 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
#pragma unroll
                for (int o = 0; o < 2; ++o) {          // o = 0: quad 2 h2 (r[0]), o = 1: quad 2 h2 + 1 (r[1])
                    const int q = 2 * h2 + o;
                    const float4 b = *(const float4*)(b2p + (t * 2 + lhi) * 16 + q * 4);
                    const unsigned w0 = r02[o], w1 = r13[o];
                    acc2[t][4 * q + 0] = __uint_as_float(w0 << 16) + b.x;
                    acc2[t][4 * q + 1] = __uint_as_float(w0 & 0xffff0000u) + b.y;
                    acc2[t][4 * q + 2] = __uint_as_float(w1 << 16) + b.z;
                    acc2[t][4 * q + 3] = __uint_as_float(w1 & 0xffff0000u) + b.w;
                }
            }
    };
    auto store_panel = [&](int panel) __attribute__((always_inline)) {
        const long long tok = (long long)panel * 128 + wave * 32 + l31;
        uint16_t* row = p.out + tok * p.ldo + 8 * lhi;
        const bool live = tok < p.M;
#pragma unroll
        for (int t = 0; t < 10; ++t)
#pragma unroll
            for (int qp = 0; qp < 2; ++qp) {
                const unsigned a0 = pack_bf16x2(acc2[t][8 * qp + 0], acc2[t][8 * qp + 1]), a1 = pack_bf16x2(acc2[t][8 * qp + 2], acc2[t][8 * qp + 3]);
                const unsigned b0 = pack_bf16x2(acc2[t][8 * qp + 4], acc2[t][8 * qp + 5]), b1 = pack_bf16x2(acc2[t][8 * qp + 6], acc2[t][8 * qp + 7]);
                const auto r0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                const auto r1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                const u32x4_t v = u32x4_t{r0[0], r1[0], r0[1], r1[1]};
                if (live) *(u32x4_t*)(row + 32 * t + 16 * qp) = v;
            }
    };

    // ---- stage B: rstd scaling + GEGLU of chunk c, ONE VALU instruction per element and micro-group ------------------------------
    // A lone wave issues an instruction every ~4 clocks but a DEPENDENT one only every 7 - 30 (plain VALU ... transcendental): a GEGLU
    // element is a chain of 17 dependent instructions.  The 32 elements of a lane are STAGGERED over the iteration's 124 micro-groups -
    // element e runs its step k in micro-group start(e) + k - so that a micro-group carries one instruction from each of ~5 different
    // elements: all independent, every latency covered by a whole group.
    // element e: pair e >> 1 (pair = (tile t, quad q, element pair ep)), half e & 1; state slot e & 7 (at most 6 elements are in flight)
    constexpr int ES = 18;                                  // steps of one element (the last one - odd elements only - packs the pair)
    constexpr int NG = 4 * 16 + 3 * 20;                     // micro-groups of an iteration
    auto e_start = [](int e) constexpr { return e * (NG - ES) / 31; };
    float ev[8], eg[8], eax[8], et[8], ee[8], ey[8], eh[8];
    // step K of element E (sdv_common.h gelu_erf_fast_f, one instruction at a time)
    auto elem_step = [&](auto E, auto K) __attribute__((always_inline)) {
        constexpr int e = decltype(E)::value, k = decltype(K)::value, sl = e & 7, pi = e >> 1, hf = e & 1;
        constexpr int t = pi >> 2, q = (pi >> 1) & 1, ep = pi & 1;
        if constexpr (k == 0) ev[sl] = rs * g[t][4 * q + 2 * ep + hf];
        else if constexpr (k == 1) eg[sl] = rs * g[t][8 + 4 * q + 2 * ep + hf];
        else if constexpr (k == 2) eax[sl] = fminf(fabsf(eg[sl]), 1e18f);
        else if constexpr (k == 3) et[sl] = __builtin_fmaf(0.3275911f * 0.70710678118654752440f, eax[sl], 1.0f);
        else if constexpr (k == 4) et[sl] = __frcp_rn(et[sl]);
        else if constexpr (k == 5) ee[sl] = eg[sl] * eg[sl];
        else if constexpr (k == 6) ee[sl] = ee[sl] * -(0.84932180028801904272f * 0.84932180028801904272f);
        else if constexpr (k == 7) ee[sl] = __builtin_amdgcn_exp2f(ee[sl]);
        else if constexpr (k == 8) ey[sl] = __builtin_fmaf(0.5f * 1.061405429f, et[sl], 0.5f * -1.453152027f);
        else if constexpr (k == 9) ey[sl] = __builtin_fmaf(ey[sl], et[sl], 0.5f * 1.421413741f);
        else if constexpr (k == 10) ey[sl] = __builtin_fmaf(ey[sl], et[sl], 0.5f * -0.284496736f);
        else if constexpr (k == 11) ey[sl] = __builtin_fmaf(ey[sl], et[sl], 0.5f * 0.254829592f);
        else if constexpr (k == 12) ee[sl] = et[sl] * ee[sl];
        else if constexpr (k == 13) ee[sl] = ee[sl] * ey[sl];
        else if constexpr (k == 14) eg[sl] = fmaxf(eg[sl], 0.0f);
        else if constexpr (k == 15) eg[sl] = __builtin_fmaf(-eax[sl], ee[sl], eg[sl]);
        else if constexpr (k == 16) eh[sl] = eg[sl] * ev[sl];
        else if constexpr (hf == 1) hidC[t][2 * q + ep] = pack_bf16x2(eh[(e - 1) & 7], eh[sl]);
    };

    // ---- one step = one slab ------------------------------------------------------------------------------------------------
    // KIND 0: K slab SL of W1 (stage A, 16 MFMAs into acc1; slab 4: + the 4 MFMAs of the fold k-step); KIND 1: half SL of W2's chunk
    // (stage C, 20 MFMAs into acc2).  PK / PS: kind / slot of the step before (its slot is refilled here); NK / NS: of the step
    // after (whose slab this step's barrier certifies and whose first fragments it reads).  VM: vector-memory operations issued
    // after the NEXT step's slab that may still be in flight at this step's wait.  GB: index of the step's first micro-group within
    // the iteration.  GT: tile of acc1 whose VGPR copy is made here (-1: none).
    constexpr int AH = 3;     // W fragments in flight ahead of the MFMA that uses them
    bf16x8_t wq[AH];          // the step's first AH W fragments (read by the step before)
    auto step = [&](auto KIND, auto SL, auto PK, auto PS, auto NK, auto NS, auto VM, auto GB, auto GT0, auto GT1) __attribute__((always_inline)) {
        constexpr int kind = decltype(KIND)::value, sl = decltype(SL)::value, pk = decltype(PK)::value, ps = decltype(PS)::value;
        constexpr int n = (kind == 0 && sl < 4) ? 16 : 20, T = kind == 0 ? 4 : 5, gb = decltype(GB)::value;
        constexpr int npieces = (pk == 0 && ps < 4) ? 4 : 5;
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (!(SDV_FFN_WHATIF & 16)) SDV_VMCNT(decltype(VM)::value);
        if constexpr (!(SDV_FFN_WHATIF & 4)) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const int rchunk = pk == 0 ? pa : pb;
        bf16x8_t w[AH + 1], nw[AH];
#pragma unroll
        for (int a = 0; a < AH; ++a) w[a] = wq[a];
        static_for<0, n>([&](auto I) {
            constexpr int i = decltype(I)::value, G = gb + i;
            // W fragment AH MFMAs ahead (the last AH belong to the next step's slab)
            if constexpr (i + AH < n) w[(i + AH) % (AH + 1)] = frag(KIND, SL, ic<i + AH>{});
            else nw[i + AH - n] = frag(NK, NS, ic<i + AH - n>{});
            // refill of the slot the previous step consumed, one piece per micro-group
            if constexpr (i < npieces && !(SDV_FFN_WHATIF & 2)) {
                if constexpr (pk == 0) pieceA(PS, rchunk, I);
                else pieceB(PS, rchunk, I);
            }
            // VGPR copies of acc1's tiles (the accumulators of chunk c leave the AGPRs before stage A re-uses them)
            if constexpr (i == 0 && decltype(GT0)::value >= 0) {
                asm volatile("" : "+a"(acc1[decltype(GT0)::value]));
                g[decltype(GT0)::value] = acc1[decltype(GT0)::value];
                asm volatile("" : "+v"(g[decltype(GT0)::value]));
            }
            if constexpr (i == (kind == 0 ? n - 1 : n / 2) && decltype(GT1)::value >= 0) {
                asm volatile("" : "+a"(acc1[decltype(GT1)::value]));
                g[decltype(GT1)::value] = acc1[decltype(GT1)::value];
                asm volatile("" : "+v"(g[decltype(GT1)::value]));
            }
            // the MFMA
            if constexpr (kind == 0)
                acc1[i % T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[i % (AH + 1)], xf[i < 16 ? 4 * sl + i / T : 20],
                                                                      (sl == 0 && i / T == 0) ? kZ16 : acc1[i % T], 0, 0, 0);
            else
                acc2[5 * sl + i % T] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[i % (AH + 1)], __builtin_bit_cast(bf16x8_t, hidP[i / T]), acc2[5 * sl + i % T], 0, 0, 0);
            // stage B: one instruction of every element in flight
            if constexpr (!(SDV_FFN_WHATIF & 1))
                static_for<0, 32>([&](auto E) {
                    constexpr int k = G - e_start(decltype(E)::value);
                    if constexpr (k >= 0 && k < ES) elem_step(E, ic<(k >= 0 && k < ES) ? k : 0>{});
                });
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int a = 0; a < AH; ++a) wq[a] = nw[a];
        if constexpr (pk == 0 && ps == 4) pa = pa == NCH - 1 ? 0 : pa + 1;
        if constexpr (pk == 1 && ps == 1) pb = pb == NCH - 1 ? 0 : pb + 1;
    };

    // one iteration of the pipeline, chunk c:  C(c - 1) on hidP  |  A(c + 1)  |  B(c) on g = acc1 of A(c)
    auto iteration = [&]() __attribute__((always_inline)) {
        //   kind slot | prev   | next   | vmcnt  | first group | g tiles copied
        step(ic<1>{}, ic<0>{}, ic<0>{}, ic<4>{}, ic<1>{}, ic<1>{}, ic<16>{}, ic<0>{}, ic<-1>{}, ic<1>{});
        step(ic<1>{}, ic<1>{}, ic<1>{}, ic<0>{}, ic<0>{}, ic<0>{}, ic<16>{}, ic<20>{}, ic<2>{}, ic<3>{});
        step(ic<0>{}, ic<0>{}, ic<1>{}, ic<1>{}, ic<0>{}, ic<1>{}, ic<17>{}, ic<40>{}, ic<-1>{}, ic<-1>{});
        step(ic<0>{}, ic<1>{}, ic<0>{}, ic<0>{}, ic<0>{}, ic<2>{}, ic<18>{}, ic<56>{}, ic<-1>{}, ic<-1>{});
        step(ic<0>{}, ic<2>{}, ic<0>{}, ic<1>{}, ic<0>{}, ic<3>{}, ic<18>{}, ic<72>{}, ic<-1>{}, ic<-1>{});
        step(ic<0>{}, ic<3>{}, ic<0>{}, ic<2>{}, ic<0>{}, ic<4>{}, ic<18>{}, ic<88>{}, ic<-1>{}, ic<-1>{});
        step(ic<0>{}, ic<4>{}, ic<0>{}, ic<3>{}, ic<1>{}, ic<0>{}, ic<17>{}, ic<104>{}, ic<-1>{}, ic<0>{});
#pragma unroll
        for (int t = 0; t < 4; ++t) hidP[t] = hidC[t];
    };

    // ---- the walk ------------------------------------------------------------------------------------------------------------
    int cur = blockIdx.x;
    if (cur >= npanels) return;
    load_x(cur);
    // fill the rings: K slabs 0 .. 3 of W1's chunk 0 (slab 4 + the fold columns are the first step's refill), both halves of W2's
    // chunk 18 (what the prologue iteration - "chunk 19 of the panel before" - pretends to consume)
    static_for<0, 4>([&](auto K) { static_for<0, 4>([&](auto I) { pieceA(K, 0, I); }); });
    static_for<0, 2>([&](auto H) { static_for<0, 5>([&](auto I) { pieceB(H, NCH - 2, I); }); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    wq[0] = frag(ic<1>{}, ic<0>{}, ic<0>{});
    wq[1] = frag(ic<1>{}, ic<0>{}, ic<1>{});
    wq[2] = frag(ic<1>{}, ic<0>{}, ic<2>{});
    iteration();                   // prologue: stage A of chunk 0 (stages B / C run on zeros)
    int prev = -1;
    for (;;) {
        rs = nrs;
        iteration();               // chunk 0: ... and C(19) of the panel before completes its accumulators
        if (prev >= 0) store_panel(prev);
        if (cur >= npanels) break;
        init_acc2();
        const int nxt = cur + (int)gridDim.x;
        for (int c = 1; c < NCH - 1; ++c) iteration();
        if (nxt < npanels) load_x(nxt);      // (this panel's x rows are no longer needed: stage A of its last chunk is done)
        iteration();
        prev = cur;
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the stream runs a few slabs ahead: nothing may land in a released LDS)
}


// =====================================================================================================================================
// out[M][N] = LayerNorm-folded / biased projection of a C = 320 activation, N = 320 (proj_in, attn.to_out, attn2.to_q) or 960 (the fused
// [Q | K | V] projection), on the same panel skeleton: the (M, 320, 320) projections of the 64 x 64 level are HBM-streaming kernels
// (107 FLOP per byte) that the 256 x 320 igemm tile ran at 40 - 58 % of the achievable HBM rate - each tile re-staged its 205 KB of
// weights next to 164 KB of activations and kept neither stream going through its epilogue.  Here a wave keeps its 32 tokens' x rows
// as B fragments, the weights stream through a 5-slot LDS ring (the same [160 x 64] half slabs as ff.net.2's above) in one fixed order
// that never drains, bias and LayerNorm fold ride in the fold k-step (a plain bias is the fold with mean 0, rstd 1), the residual is the
// accumulators' initial value, and the LayerNorm statistics of the stored rows - a whole row lives in one wave - leave as (mean, rstd),
// no partial sums and no finalize launch.  The x rows of the NEXT panel are requested k-slab by k-slab as this panel's last use of
// each passes.
struct lin_args {
    const uint16_t* X;
    const uint16_t* W;
    const uint16_t* Wx;
    const float* ln_stats;
    const float* alpha;
    const uint16_t* R;
    uint16_t* out;
    float* stats_out;
    uint16_t* Vt;
    long long M;
    int ldx, ldr, ldo, ldvt, hw;
    float eps;
};

constexpr int LFOLD_BYTES = 384 * 32;                    // 12 pieces: the block's 320 rows of fold columns + padding to 3 pieces per wave
constexpr int LIN_FOLD_OFF = 5 * SLOT_B;
constexpr int LIN_STAGE_OFF = LIN_FOLD_OFF + LFOLD_BYTES;  // three 16 KiB slots: k slabs of the NEXT panel's x rows on their way into registers
constexpr int LIN_LDS = LIN_STAGE_OFF + 3 * 16384;
static_assert(LIN_LDS <= 160 * 1024, "ring + fold columns + x staging must fit the 160 KiB LDS");

// KSL = 64-wide k slabs of the input rows: 5 (C = 320) or 10 (C = 640; the x rows are then 40 fragments = 160 registers, which is
// why there is no residual form: its 20 more fragments do not fit one wave per SIMD)
template <int NB, bool RES, int KSL = 5>
__global__ __launch_bounds__(256, 1) void panel_linear_kernel(const lin_args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int npanels = (int)((p.M + 127) >> 7);
    constexpr int KC = 64 * KSL, NS = 2 * KSL;        // input channels; half-slab steps per block (step NS = the fold k-step)
    static_assert(KSL == 5 || (KSL == 10 && !RES), "k slabs per row: 5, or 10 without the residual form");

    auto swz = [](int r) { return (r >> 1) & 7; };
    const int rg = lane >> 3, pc = lane & 7;
    unsigned voW[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int r = (wave + 4 * i) * 8 + rg;
        voW[i] = (unsigned)(r * (KC * 2) + ((pc ^ swz(r)) << 4));
    }
    unsigned voF[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) voF[i] = (unsigned)(((wave + 4 * i) * 32 + (lane >> 1)) * 32 + (lane & 1) * 16);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, NB * FC * KC * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsF = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wx, 0, NB * FC * 32, 0x00020000);
    typedef __attribute__((address_space(3))) void* lds_ptr;
    // piece i of half slab (block blk, k slab ks, half hf) -> ring slot `slot`
    auto pieceW = [&](auto SLOT, int blk, auto KS, auto HF, auto I) __attribute__((always_inline)) {
        constexpr int slot = decltype(SLOT)::value, ks = decltype(KS)::value, hf = decltype(HF)::value, i = decltype(I)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(smem + slot * SLOT_B + (wave + 4 * i) * 1024), 16, (int)voW[i],
                                                 (blk * FC + hf * 160) * (KC * 2) + ks * 128, 0, 0);
    };
    auto pieceF = [&](int blk, auto I) __attribute__((always_inline)) {
        constexpr int i = decltype(I)::value;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsF, (lds_ptr)(smem + LIN_FOLD_OFF + (wave + 4 * i) * 1024), 16, (int)voF[i], blk * FC * 32, 0, 0);
    };
    int fo[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) fo[kk] = l31 * 128 + (((2 * kk + lhi) ^ swz(l31)) << 4);
    const int fof = l31 * 32 + lhi * 16;
    // fragment i (= kk * 5 + t) of the half slab in ring slot SLOT; SLOT 5 = the fold columns (fragment i = tile i)
    auto frag = [&](auto SLOT, auto I) __attribute__((always_inline)) -> bf16x8_t {
        constexpr int slot = decltype(SLOT)::value, i = decltype(I)::value;
        if constexpr (slot == 5) return *(const bf16x8_t*)(smem + LIN_FOLD_OFF + i * 1024 + fof);
        else return *(const bf16x8_t*)(smem + slot * SLOT_B + (i % 5) * 4096 + fo[(i / 5) & 3]);
    };

    bf16x8_t xf[4 * KSL + 1];
    bf16x8_t rf[RES ? 20 : 1];
    float rs = 1.f, nrs = 1.f;
    float2 nst = make_float2(0.f, 1.f);
    float alf[NB];                   // (read ONCE: a load inside the walk is a vmcnt(0) wait in front of its first use - the stream drains)
#pragma unroll
    for (int b = 0; b < NB; ++b) alf[b] = p.alpha ? p.alpha[b] : 1.f;
    f32x16_t acc[10];
#pragma unroll
    for (int t = 0; t < 10; ++t) acc[t] = kZ16;
    auto split3 = [](float x, unsigned& h, unsigned& m, unsigned& l) {
        h = pack_bf16x2(x, 0.f) & 0xffffu;
        const float r1 = x - __uint_as_float(h << 16);
        m = pack_bf16x2(r1, 0.f) & 0xffffu;
        l = pack_bf16x2(r1 - __uint_as_float(m << 16), 0.f) & 0xffffu;
    };
    auto tok_of = [&](int panel) {
        long long tok = (long long)panel * 128 + wave * 32 + l31;
        return tok < p.M ? tok : p.M - 1;
    };
    // The x rows reach the registers through LDS: loaded straight in the fragment layout every instruction touches 32 rows x 32 bytes,
    // and that shape streams from HBM at 3 - 5 bytes per clock and CU (profiles/round4_ldsdma_fill.txt) - a panel's 80 KB took as long
    // as the whole panel.  stage_x: this wave's 32 rows of k slab ks as four LDS-DMA pieces of 8 rows x 128 bytes (whole lines, the
    // weight slabs' swizzled image) into staging slot ks % 3; read_x: the four fragments out of it, once a vmcnt window has closed
    // behind the pieces (no barrier: a wave reads only rows it fetched itself).
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc((void*)p.X, 0, 0x7ffffff0, 0x00020000);
    auto stage_x = [&](int panel, auto KS) __attribute__((always_inline)) {
        constexpr int ks = decltype(KS)::value;
        static_for<0, 4>([&](auto I) {
            constexpr int i = decltype(I)::value;
            long long row = (long long)panel * 128 + wave * 32 + 8 * i + rg;
            row = row < p.M ? row : p.M - 1;
            const int vo = (int)(row * p.ldx * 2) + ((pc ^ swz(8 * i + rg)) << 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, (lds_ptr)(smem + LIN_STAGE_OFF + (ks % 3) * 16384 + wave * 4096 + i * 1024), 16, vo, ks * 128, 0, 0);
        });
    };
    auto read_x = [&](auto KS) __attribute__((always_inline)) {
        constexpr int ks = decltype(KS)::value;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) xf[4 * ks + kk] = *(const bf16x8_t*)(smem + LIN_STAGE_OFF + (ks % 3) * 16384 + wave * 4096 + fo[kk]);
    };
    // ks 6: the statistics of the rows are REQUESTED (early); ks 5: they are split into the fold k-step's token side (late)
    auto load_x = [&](int panel, auto KS) __attribute__((always_inline)) {
        constexpr int ks = decltype(KS)::value;
        static_assert(ks >= 5, "the k slabs go through stage_x / read_x; 5 / 6 name the two halves of the statistics load");
        const long long tok = tok_of(panel);
        if constexpr (ks == 6) {      // (the statistics are REQUESTED early - ks 6 - and split into the fold fragment late - ks 5)
            nst = make_float2(0.f, 1.f);
            if (p.ln_stats) nst = *(const float2*)(p.ln_stats + 2 * tok);
        } else {
            const float mean = nst.x, rstd = nst.y;
            nrs = rstd;
            unsigned mh, mm, ml, rh, rm, rl;
            split3(-mean, mh, mm, ml);
            split3(1.0f / rstd, rh, rm, rl);
            const u32x4_t lo = u32x4_t{mh | (mm << 16), mh | (ml << 16), mh | (mm << 16), rh | (rm << 16)};
            const u32x4_t hi = u32x4_t{rh | (rl << 16), rh | (rm << 16), 0u, 0u};
            xf[4 * KSL] = __builtin_bit_cast(bf16x8_t, lhi ? hi : lo);
        }
    };
    auto load_r = [&](int panel) __attribute__((always_inline)) {
        if constexpr (RES) {
            const uint16_t* row = p.R + tok_of(panel) * p.ldr + 8 * lhi;
#pragma unroll
            for (int s = 0; s < 20; ++s) rf[s] = *(const bf16x8_t*)(row + 16 * s);
        }
    };
    // acc tiles [T0, T1) <- residual rows in the accumulator layout (one register-pair exchange between a token's two lanes)
    auto init_acc = [&](auto T0, auto T1) __attribute__((always_inline)) {
        if constexpr (RES) {
            static_for<decltype(T0)::value, decltype(T1)::value>([&](auto TT) {
                constexpr int t = decltype(TT)::value;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2) {
                    const u32x4_t d = __builtin_bit_cast(u32x4_t, rf[2 * t + h2]);
                    const auto r02 = __builtin_amdgcn_permlane32_swap(d[0], d[2], false, false);
                    const auto r13 = __builtin_amdgcn_permlane32_swap(d[1], d[3], false, false);
#pragma unroll
                    for (int o = 0; o < 2; ++o) {
                        const int q = 2 * h2 + o;
                        const unsigned w0 = r02[o], w1 = r13[o];
                        acc[t][4 * q + 0] = __uint_as_float(w0 << 16);
                        acc[t][4 * q + 1] = __uint_as_float(w0 & 0xffff0000u);
                        acc[t][4 * q + 2] = __uint_as_float(w1 << 16);
                        acc[t][4 * q + 3] = __uint_as_float(w1 & 0xffff0000u);
                    }
                }
            });
        }
    };
    // block blk of panel `panel`: scale, round, store; (sum, sumsq) of the stored values into s1 / s2.
    // The values leave through LDS, 64 channels (two tiles) at a time: straight from the accumulator layout a store instruction
    // writes 32 rows x 32 bytes - quarter lines - and the [tokens, 960] output of the fused Q / K / V projection ran at 3.2 TB/s.
    // Each wave parks the slice in its 4 KiB of staging slot 2 (free whenever an epilogue runs: k slab 2 of the next panel was read
    // at step 9, and nothing but the last block of a panel stages) as [32 tokens][128 bytes], 16-byte chunks XOR-swizzled by the row,
    // and reads it back
    //   * row-major - 8 lanes per token row: every store instruction writes 8 whole 128-byte lines; or
    //   * TRANSPOSED (the V third when p.Vt is given) - ds_read_b64_tr_b16 hands lane (token group g, channel i) the 8 tokens
    //     8 g .. 8 g + 7 of ITS channel: one 16-byte store into V^T[sample][channel][token], 64 contiguous bytes per channel and
    //     instruction - the layout the self-attention kernel's one-read-per-fragment form wants (sdv_attention_bf16 v_rowmajor = 0).
    float s1 = 0.f, s2 = 0.f;
    typedef unsigned int __attribute__((ext_vector_type(4), may_alias)) slab_u4;
    typedef unsigned int __attribute__((ext_vector_type(2), may_alias)) slab_u2;
    auto store_block = [&](int panel, int blk, float scale) __attribute__((always_inline)) {
        // buffer stores through a descriptor that ends with this wave's last live row: rows past M drop out in the range check, and
        // EVERY wave issues all 20 stores of a block whatever M is - the counted vmcnt waits of the steps behind rely on that
        const long long row0 = (long long)panel * 128 + wave * 32;
        const long long left = p.M - row0;
        const bool tr = (NB == 3 && KSL == 5) && p.Vt != nullptr && blk == 2;       // (the other forms compile no transposed path)
        char* const slab = smem + LIN_STAGE_OFF + ((KSL - 3) % 3) * 16384 + wave * 4096;      // (the staging slot no pending k slab lives in)
        const unsigned ones = 0x3f803f80u;
        __amdgpu_buffer_rsrc_t rs_o;
        int vo[4];
        if (!tr) {
            rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)(p.out + (left > 0 ? row0 : 0) * p.ldo), 0,
                                                     left > 0 ? (int)((left < 32 ? left : 32) * p.ldo * 2) : 0, 0x00020000);
#pragma unroll
            for (int j = 0; j < 4; ++j) vo[j] = ((8 * j + (lane >> 3)) * p.ldo + 8 * (lane & 7)) * 2;      // (the block's column offset rides in the scalar offset)
        } else {
            // V^T[sample][channel][token]: this wave's 32 tokens are tokens t0 .. t0 + 31 of sample b (hw % 128 == 0: host)
            const long long b = row0 / p.hw, t0 = row0 - b * p.hw;
            rs_o = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Vt + (b * FC * p.ldvt + t0)), 0, left > 0 ? (int)((FC - 1) * p.ldvt + 32) * 2 : 0, 0x00020000);
#pragma unroll
            for (int j = 0; j < 4; ++j) vo[j] = ((16 * j + (lane & 15)) * p.ldvt + 8 * (lane >> 4)) * 2;
        }
#pragma unroll
        for (int sl = 0; sl < 5; ++sl) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int qp = 0; qp < 2; ++qp) {
                    const int t = 2 * sl + tt;
                    float v[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] = RES ? acc[t][8 * qp + e] : acc[t][8 * qp + e] * scale;
                    const unsigned a0 = pack_bf16x2(v[0], v[1]), a1 = pack_bf16x2(v[2], v[3]);
                    const unsigned b0 = pack_bf16x2(v[4], v[5]), b1 = pack_bf16x2(v[6], v[7]);
                    if (p.stats_out) {
                        asm volatile("v_dot2c_f32_bf16 %0, %2, %6\n\tv_dot2c_f32_bf16 %1, %2, %2\n\tv_dot2c_f32_bf16 %0, %3, %6\n\tv_dot2c_f32_bf16 %1, %3, %3\n\t"
                                     "v_dot2c_f32_bf16 %0, %4, %6\n\tv_dot2c_f32_bf16 %1, %4, %4\n\tv_dot2c_f32_bf16 %0, %5, %6\n\tv_dot2c_f32_bf16 %1, %5, %5\n\ts_nop 2"
                                     : "+v"(s1), "+v"(s2)
                                     : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(ones));
                    }
                    // quads 2 qp / 2 qp + 1 = channels 32 tt + 16 qp + {0, 8} + 4 lhi + {0..3} of the slice: 16-byte chunk 4 tt + 2 qp + o, half lhi
                    *(slab_u2*)(slab + l31 * 128 + (((4 * tt + 2 * qp) ^ (l31 & 7)) << 4) + lhi * 8) = slab_u2{a0, a1};
                    *(slab_u2*)(slab + l31 * 128 + (((4 * tt + 2 * qp + 1) ^ (l31 & 7)) << 4) + lhi * 8) = slab_u2{b0, b1};
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (a wave's DS operations execute in order; this keeps the compiler in order too)
            u32x4_t o[4];
            if (!tr) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = 8 * j + (lane >> 3);
                    o[j] = *(const slab_u4*)(slab + row * 128 + (((lane & 7) ^ (row & 7)) << 4));
                }
            } else {
                typedef short __attribute__((ext_vector_type(4))) s16x4_t;
                typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4_p;
                // lane (g = lane >> 4, i = lane & 15) addresses row 8 g + 4 h + (i >> 2), channels 16 j + 4 (i & 3) .. + 3, and receives
                // the four rows of channel 16 j + i
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    s16x4_t part[2];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int row = 8 * (lane >> 4) + 4 * h + ((lane & 15) >> 2);
                        const int c16 = 2 * j + ((lane & 3) >> 1);
                        part[h] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_p)(slab + row * 128 + ((c16 ^ (row & 7)) << 4) + (lane & 1) * 8));
                    }
                    typedef short __attribute__((ext_vector_type(8))) s16x8_t;
                    o[j] = __builtin_bit_cast(u32x4_t, s16x8_t{part[0][0], part[0][1], part[0][2], part[0][3], part[1][0], part[1][1], part[1][2], part[1][3]});
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(o[j], rs_o, vo[j], tr ? sl * 64 * p.ldvt * 2 : sl * 128 + blk * (FC * 2), 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the slice's reads are done before the next one is parked
        }
    };

    // ---- one step = one half slab: J = its position in the block's 10 (slot J % 5, k slab J / 2, rows half J % 2); J = 10: the fold
    // k-step (10 MFMAs, one per tile).  The barrier of step J certifies the slab of step J + 1; the slot of step J - 1 is refilled
    // with the slab five steps after it.  `blk`: the block being computed, `nblk`: the block that follows it in the stream.
    // vmcnt bookkeeping.  The counter is IN ORDER over every vector-memory operation of the wave - the weight pieces, the x / residual
    // loads of the next panel (HBM latency) and the stores - so a wait for a weight slab also waits for whatever was issued before
    // it, and a window that is too small makes the weight stream inherit the HBM latency of the activation loads (the first form of
    // this kernel: 100 clocks per MFMA).  Hence EVERY operation is issued unconditionally (clamped rows, range-checked stores): the
    // sequence is static, and the window of a step is the exact number of operations issued behind the slab it needs.
    //   ops of step s behind its 5 weight pieces: step 0: + 3 fold pieces (+ 20 residual loads of the next panel); odd steps of a
    //   panel's last block: + 4 x loads; the fold step: + 20 stores
#ifndef SDV_LIN_AH
#define SDV_LIN_AH 3
#endif
    constexpr int AH = SDV_LIN_AH;
    bf16x8_t wq[AH];
    auto step = [&](auto JJ, auto LASTB, auto PREVLAST, auto FIRSTB, int blk, int nblk, int xnext) __attribute__((always_inline)) {
        constexpr int J = decltype(JJ)::value;
        constexpr bool lastb = decltype(LASTB)::value != 0, prevlast = decltype(PREVLAST)::value != 0;
        constexpr auto tail_ops = [](int s, bool last) constexpr {
            int nn = 0;
            if (s == 0) nn += 3 + ((RES && last) ? 20 : 0);
            if (last && s < NS && (s & 1)) nn += 4;
            if (s == NS) nn += 20;
            return nn;
        };
        constexpr auto window = [tail_ops](int j, bool last, bool plast) constexpr {
            const int need = j < NS ? j + 1 : NS + 1;             // stream position (this block's steps 0 .. NS - 1; NS + 1 = the next block's step 0) of the slab needed
            const int is = (need == NS + 1 ? NS : need) - 4;      // step that issued it (relative to this block; < 0: the previous block's step is + NS + 1)
            int nn = 0;
            for (int q = is; q < j; ++q) {
                const bool lq = q < 0 ? plast : last;
                const int sq = q < 0 ? q + NS + 1 : q;
                nn += (q == is ? 0 : (sq < NS ? 5 : 0)) + tail_ops(sq, lq);
            }
            return nn;
        };
        constexpr int n = J < NS ? 20 : 10, slot = J < NS ? J % 5 : 5, ks = J / 2, hf = J % 2;
        constexpr int nslot = J < NS - 1 ? (J + 1) % 5 : (J == NS - 1 ? 5 : 0);                 // where the NEXT step's fragments are
        __builtin_amdgcn_sched_barrier(0);
        constexpr int VM0 = window(J, lastb, prevlast);
        static_assert(VM0 >= 10 && VM0 <= 63, "vmcnt window");
#ifdef SDV_LIN_WHATIF      // TIMING-ONLY: windows wider than what was issued (weights may be read before they land)
        constexpr int VM = VM0 + SDV_LIN_WHATIF > 63 ? 63 : VM0 + SDV_LIN_WHATIF;
#else
        constexpr int VM = VM0;
#endif
        SDV_VMCNT(VM);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // the next panel's x rows: k slab ks was staged behind step 2 ks + 1 of the panel's last block; the first window that has
        // closed behind it is the one of step 2 ks + 5 - steps 5 / 7 / 9 of that block, steps 1 / 3 of the next panel's first block
        if constexpr (lastb && J >= 5 && J < NS && (J & 1)) read_x(ic<(J - 5) / 2>{});
        if constexpr (decltype(FIRSTB)::value != 0 && (J == 1 || J == 3)) read_x(ic<KSL - 2 + (J - 1) / 2>{});
        bf16x8_t w[AH + 1], nw[AH];
#pragma unroll
        for (int a = 0; a < AH; ++a) w[a] = wq[a];
        static_for<0, n>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr (i + AH < n) w[(i + AH) % (AH + 1)] = frag(ic<slot>{}, ic<i + AH>{});
            else nw[i + AH - n] = frag(ic<nslot>{}, ic<i + AH - n>{});
            // refills: steps 1 .. 9 refill the slot of step J - 1 with that step's successor five steps on (J + 4: this block's
            // for J + 4 < 10, else the next block's); step 0 refills step 9's slot; the fold step refills nothing, step 0 of the next
            // block re-fetches the fold columns behind its own refill
#if defined(SDV_LIN_NODMA)       // TIMING-ONLY: no weight refills
            if constexpr (false) {
#else
            if constexpr (J < NS && i < 5) {
#endif
                constexpr int pj = (J + NS - 1) % NS, tj = (pj + 5) % NS;              // step whose slot is free / step whose slab goes there
                // J >= 1: target step tj = J + 4 belongs to this block if J + 4 < 10, else to the next; J == 0: tj = 4 of THIS block
                pieceW(ic<pj % 5>{}, (J >= 1 && J + 4 >= NS) ? nblk : blk, ic<tj / 2>{}, ic<tj % 2>{}, I);
            }
            if constexpr (J == 0 && i >= 5 && i < 8) pieceF(blk, ic<i - 5>{});
            if constexpr (J < NS)
                acc[5 * hf + i % 5] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[i % (AH + 1)], xf[4 * ks + i / 5],
                                                                              (!RES && ks == 0 && i / 5 == 0) ? kZ16 : acc[5 * hf + i % 5], 0, 0, 0);
            else
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w[i % (AH + 1)], xf[4 * KSL], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
#pragma unroll
        for (int a = 0; a < AH; ++a) wq[a] = nw[a];
        // the x rows of the next panel, k slab by k slab, once this panel's last block is through with them
        if constexpr (lastb && J < NS && hf == 1) stage_x(xnext, ic<ks>{});
        if constexpr (lastb && J == 1) load_x(xnext, ic<6>{});
        if constexpr (lastb && J == NS) load_x(xnext, ic<5>{});
    };

    // ---- the walk ------------------------------------------------------------------------------------------------------------
    int cur = blockIdx.x;
    if (cur >= npanels) return;
    load_x(cur, ic<6>{});
    // the first panel's x rows, through the three staging slots, at most three k slabs per round; the last two stay in their slots:
    // the first panel's steps 1 / 3 read them (like every later panel's)
    static_for<0, (KSL - 2 + 2) / 3>([&](auto R) {
        constexpr int k0 = 3 * decltype(R)::value, k1 = k0 + 3 < KSL - 2 ? k0 + 3 : KSL - 2;
        static_for<k0, k1>([&](auto KS) { stage_x(cur, KS); });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        static_for<k0, k1>([&](auto KS) { read_x(KS); });
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    });
    static_for<KSL - 2, KSL>([&](auto KS) { stage_x(cur, KS); });
    load_x(cur, ic<5>{});
    load_r(cur);
    // fill the ring: the first five half slabs of block 0 + its fold columns
    static_for<0, 5>([&](auto S) { static_for<0, 5>([&](auto I) { pieceW(S, 0, ic<decltype(S)::value / 2>{}, ic<decltype(S)::value % 2>{}, I); }); });
    static_for<0, 3>([&](auto I) { pieceF(0, I); });
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int a = 0; a < AH; ++a) wq[a] = kZ16x8();
    static_for<0, AH>([&](auto A) { wq[decltype(A)::value] = frag(ic<0>{}, A); });
    for (;;) {
        const int nxt = cur + (int)gridDim.x;
        rs = nrs;
        s1 = s2 = 0.f;
        const int xn = nxt < npanels ? nxt : cur;     // (past the end: the loads are issued all the same - see the vmcnt bookkeeping - and dropped)
        static_for<0, NB>([&](auto BLK) {
            constexpr int blk = decltype(BLK)::value;
            constexpr int nblk = blk + 1 < NB ? blk + 1 : 0;
            using LB = ic<blk == NB - 1>;
            using PL = ic<(NB == 1 || blk == 0)>;
            using FB = ic<blk == 0>;
            if constexpr (RES) init_acc(ic<0>{}, ic<5>{});
            step(ic<0>{}, LB{}, PL{}, FB{}, blk, nblk, xn);
            if constexpr (RES) {
                init_acc(ic<5>{}, ic<10>{});
                load_r(xn);                       // (the residual registers are free again: the next panel's rows)
            }
            static_for<1, NS + 1>([&](auto JJ) { step(JJ, LB{}, PL{}, FB{}, blk, nblk, xn); });
            store_block(cur, blk, rs * alf[blk]);
        });
        if (p.stats_out) {
            const float t1 = s1 + __shfl_xor(s1, 32), t2 = s2 + __shfl_xor(s2, 32);
            const float mean = t1 * (1.0f / (NB * FC));
            const float var = fmaxf(t2 * (1.0f / (NB * FC)) - mean * mean, 0.f);
            const long long tok = (long long)cur * 128 + wave * 32 + l31;
            if (lhi == 0 && tok < p.M) *(float2*)(p.stats_out + 2 * tok) = make_float2(mean, rsqrtf(var + p.eps));
        }
        if (nxt >= npanels) break;
        cur = nxt;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace

// C ABI: sdv_hip.h
extern "C" int sdv_ffn_geglu_bf16(const sdv_bf16* X, const float* ln_stats, int64_t M, int32_t C, int32_t ldx, const sdv_bf16* W1, const sdv_bf16* W1x,
                                  const sdv_bf16* W2p, const float* bias2, sdv_bf16* out, int32_t ldo, void* stream) {
    SDV_REQUIRE(X && ln_stats && W1 && W1x && W2p && bias2 && out, "sdv_ffn_geglu_bf16: null pointer");
    SDV_REQUIRE(C == FC, "sdv_ffn_geglu_bf16: built for C = %d channels (the 64 x 64 level of the UNet), got %d", FC, C);
    SDV_REQUIRE(M > 0 && M < (1LL << 31) - 128, "sdv_ffn_geglu_bf16: bad M");
    SDV_REQUIRE(ldx >= C && ldo >= C && ldx % 8 == 0 && ldo % 8 == 0, "sdv_ffn_geglu_bf16: ldx / ldo must be multiples of 8 and >= C");
    SDV_REQUIRE(((((uintptr_t)X) | ((uintptr_t)out) | ((uintptr_t)W1) | ((uintptr_t)W1x) | ((uintptr_t)W2p)) & 15) == 0 && (((uintptr_t)ln_stats) & 7) == 0,
                "sdv_ffn_geglu_bf16: unaligned pointers");
    static unsigned long long attr_set = 0;     // one bit per device
    static int cus[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (!(attr_set & (1ull << dev))) {
        (void)hipFuncSetAttribute((const void*)ffn_geglu_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, FFN_LDS);
        attr_set |= 1ull << dev;
    }
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const int npanels = (int)((M + 127) >> 7);
    ffn_args a{X, W1, W1x, W2p, bias2, ln_stats, out, M, ldx, ldo};
    hipLaunchKernelGGL(ffn_geglu_kernel, dim3((unsigned)(npanels < cus[dev] ? npanels : cus[dev])), dim3(256), FFN_LDS, (hipStream_t)stream, a);
    SDV_CHECK_LAUNCH("sdv_ffn_geglu_bf16");
    return SDV_OK;
}

// K = 320 (N = 320 / 960) and K = 640 (N = 640 / 1920, no residual, no V^T) share the checks and the launch
static int panel_linear_launch(const char* who, int K, const sdv_bf16* X, int64_t M, int32_t ldx, const sdv_bf16* W, const sdv_bf16* Wx, int32_t N,
                               const float* ln_stats, const float* alpha, const sdv_bf16* R, int32_t ldr, sdv_bf16* out, int32_t ldo, float* stats_out,
                               float eps, sdv_bf16* Vt, int32_t ldvt, int32_t hw, void* stream) {
    SDV_REQUIRE(X && W && Wx && out, "%s: null pointer", who);
    SDV_REQUIRE(N == K || N == 3 * K, "%s: N must be %d or %d, got %d", who, K, 3 * K, N);
    SDV_REQUIRE(M > 0 && M < (1LL << 31) - 128, "%s: bad M", who);
    SDV_REQUIRE(ldx >= K && ldo >= (Vt ? 2 * K : N) && ldx % 8 == 0 && ldo % 8 == 0 && (!R || (ldr >= K && ldr % 8 == 0)), "%s: leading dimensions must be multiples of 8 and cover the rows", who);
    SDV_REQUIRE(!Vt || (K == FC && N == 3 * FC && hw > 0 && hw % 128 == 0 && M % hw == 0 && ldvt >= hw && ldvt % 8 == 0 && (((uintptr_t)Vt) & 15) == 0),
                "%s: a transposed V^T output goes with K = 320, N = 960, whole samples of hw %% 128 == 0 tokens and ldvt >= hw", who);
    SDV_REQUIRE(M * (long long)ldx * 2 < 0x7fffffffLL, "%s: the x rows must span less than 2 GiB", who);
    SDV_REQUIRE(!R || (K == FC && N == FC && !ln_stats && !alpha), "%s: a residual goes with K = N = 320 and no fold / alpha (it is the accumulators' initial value)", who);
    SDV_REQUIRE(!stats_out || N == K, "%s: row statistics exist for N = K", who);
    SDV_REQUIRE(((((uintptr_t)X) | ((uintptr_t)out) | ((uintptr_t)W) | ((uintptr_t)Wx) | ((uintptr_t)R)) & 15) == 0 &&
                    ((((uintptr_t)ln_stats) | ((uintptr_t)stats_out)) & 7) == 0,
                "%s: unaligned pointers", who);
    static int cus[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    dev &= 63;
    if (!cus[dev]) {
        hipDeviceProp_t prop;
        cus[dev] = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    auto launch = [&](auto kern) {
        static unsigned long long attr_set = 0;
        if (!(attr_set & (1ull << dev))) {
            (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LIN_LDS);
            attr_set |= 1ull << dev;
        }
        const int npanels = (int)((M + 127) >> 7);
        lin_args a{X, W, Wx, ln_stats, alpha, R, out, stats_out, Vt, M, ldx, ldr, ldo, ldvt, hw, eps};
        hipLaunchKernelGGL(kern, dim3((unsigned)(npanels < cus[dev] ? npanels : cus[dev])), dim3(256), LIN_LDS, (hipStream_t)stream, a);
    };
    if (K == FC) {
        if (N == FC) {
            if (R) launch(panel_linear_kernel<1, true>);
            else launch(panel_linear_kernel<1, false>);
        } else {
            launch(panel_linear_kernel<3, false>);
        }
    } else {
        if (N == K) launch(panel_linear_kernel<2, false, 10>);
        else launch(panel_linear_kernel<6, false, 10>);
    }
    SDV_CHECK_LAUNCH(who);
    return SDV_OK;
}

extern "C" int sdv_linear320_bf16(const sdv_bf16* X, int64_t M, int32_t ldx, const sdv_bf16* W, const sdv_bf16* Wx, int32_t N, const float* ln_stats,
                                  const float* alpha, const sdv_bf16* R, int32_t ldr, sdv_bf16* out, int32_t ldo, float* stats_out, float eps,
                                  sdv_bf16* Vt, int32_t ldvt, int32_t hw, void* stream) {
    return panel_linear_launch("sdv_linear320_bf16", FC, X, M, ldx, W, Wx, N, ln_stats, alpha, R, ldr, out, ldo, stats_out, eps, Vt, ldvt, hw, stream);
}

extern "C" int sdv_linear640_bf16(const sdv_bf16* X, int64_t M, int32_t ldx, const sdv_bf16* W, const sdv_bf16* Wx, int32_t N, const float* ln_stats,
                                  const float* alpha, sdv_bf16* out, int32_t ldo, float* stats_out, float eps, void* stream) {
    return panel_linear_launch("sdv_linear640_bf16", 2 * FC, X, M, ldx, W, Wx, N, ln_stats, alpha, nullptr, 0, out, ldo, stats_out, eps, nullptr, 0, 0, stream);
}
