// ldsdma_fill: what limits the operand fill of the igemm K loop?  (DESIGN.md (d) "Staged bytes per FLOP": every GEMM / conv
// shape of the 64^2 level moves 12 - 15 bytes per clock and CU through the vector memory path, the L2-resident convs 18 - 20.)
//
// One persistent 8-wave workgroup per CU issues nothing but operand fetches, `depth` slabs of `pieces` wave-instructions per
// wave in flight, counted vmcnt, no LDS reads, no MFMA.  Variants:
//   path   0: buffer_load_dwordx4 ... lds (LDS-DMA, 1 KiB per wave-instruction)      1: buffer_load_dwordx4 -> VGPRs
//   shape  rows x bytes of ONE wave-instruction: 8 x 128 (the igemm's piece: 8 rows of a 64-wide bf16 K slab), 1 x 1024 (the same
//          bytes of a slab-major, pre-tiled operand), 32 x 32 (the MFMA fragment itself: lane l reads 16 B of row l & 31 at k-half l >> 5),
//          16 x 64 (fp8 piece)
//   stride row stride in bytes (640 = K 320, 2560 = K 1280 ...; ignored by 1 x 1024)
//   foot   bytes of the region a workgroup cycles through;  share 0: one region per workgroup (X-like; HBM / Infinity Cache once
//          256 x foot exceeds the caches), 1: all workgroups read the SAME region (W-like, L2-hot), 2: one region per XCD
// Prints bytes per clock and CU (s_memtime over the kernel body, max over workgroups) and TB/s by the host clock.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x)                                                                          \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

constexpr int NWV = 8;
constexpr int MAXP = 9;            // pieces per wave and slab (9 x 8 waves x 1 KiB = the 256 x 320 tile's 72 KiB slab)

struct Args {
    const char* src;
    unsigned long long* cycles;    // per workgroup
    unsigned* sink;
    long long foot;                // bytes per region
    int share, rows, rowbytes, stride, pieces, nslab, depth, path;
};

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int PATH>
__global__ __launch_bounds__(NWV * 64, 1) void fill_kernel(Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.x;
    const long long region = p.share == 1 ? 0 : (p.share == 2 ? (b & 7) : b);
    const char* base = p.src + region * p.foot;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7ffffff0, 0x00020000);
    // lane -> (row, 16-byte chunk) inside a piece
    const int cpr = p.rowbytes / 16;                  // chunks per row: 8, 64, 2, 4
    const int lrow = lane / cpr, lchunk = lane % cpr;
    const int lane_off = lrow * (p.rows == 1 ? 0 : p.stride) + lchunk * 16;
    // a piece covers rows x rowbytes; consecutive pieces of a slab walk down the rows, consecutive slabs walk along the row
    // (the K direction) and wrap inside the region
    const long long piece_rows = p.rows;
    const long long slab_rows = piece_rows * NWV * p.pieces;          // rows one slab covers
    const long long region_rows = p.rows == 1 ? 0 : p.foot / p.stride;
    u32x4 acc = {0, 0, 0, 0};
    u32x4 r[PATH == 1 ? MAXP * 2 : 1];
    const long long t0 = __builtin_readcyclecounter();
    long long koff = 0, rowblk = 0;
    auto issue = [&](int s, int slot) {
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            if (i >= p.pieces) break;
            const int g = wave + NWV * i;
            long long off;
            if (p.rows == 1) {
                off = ((long long)s * NWV * p.pieces + g) * 1024 % p.foot;
            } else {
                off = (rowblk + (long long)g * piece_rows) * p.stride + koff;
            }
            const int voff = (int)off + lane_off;
            if constexpr (PATH == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + (slot * MAXP * NWV + g) * 1024), 16,
                                                         voff, 0, 0, 0);
            } else {
                r[(slot & 1) * MAXP + i] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 0);
            }
        }
        if (p.rows != 1) {      // next slab: along K, then the next block of rows
            koff += p.rowbytes;
            if (koff + p.rowbytes > p.stride) {
                koff = 0;
                rowblk += slab_rows;
                if (rowblk + slab_rows > region_rows) rowblk = 0;
            }
        }
    };
    int issued = 0;
    for (; issued < p.depth && issued < p.nslab; ++issued) issue(issued, issued % 2);
    for (int s = 0; s < p.nslab; ++s) {
        const int younger = issued - s - 1;
        // counted wait: `younger` slabs of p.pieces instructions may stay in flight (pieces is 9 or less; the counts below are for
        // the compiled piece counts 9 / 5 / 4)
        const int n = younger * p.pieces;
        if (n >= 27) asm volatile("s_waitcnt vmcnt(27)" ::: "memory");
        else if (n >= 18) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else if (n >= 15) asm volatile("s_waitcnt vmcnt(15)" ::: "memory");
        else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (n >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (n >= 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (n >= 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (PATH == 1) {
            if (younger == 0) {
#pragma unroll
                for (int i = 0; i < MAXP * 2; ++i) acc ^= r[i];
            }
        }
        __builtin_amdgcn_s_barrier();
        if (issued < p.nslab) {
            issue(issued, issued % 2);
            ++issued;
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    if (tid == 0) p.cycles[b] = (unsigned long long)(t1 - t0);
    if constexpr (PATH == 0) {
        const u32x4 v = *(const u32x4*)(smem + tid * 16);
        acc ^= v;
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) atomicAdd(p.sink, 1u);
}

int main(int argc, char** argv) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    // two slots of 72 KiB (depth <= 2)
    const int lds_bytes = 2 * MAXP * NWV * 1024;
    CHECK(hipFuncSetAttribute((const void*)fill_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    CHECK(hipFuncSetAttribute((const void*)fill_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    const size_t total = (size_t)3 << 30;       // 3 GiB source
    char* src;
    unsigned long long* cyc;
    unsigned* sink;
    CHECK(hipMalloc(&src, total));
    CHECK(hipMalloc(&cyc, 8 * 1024));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(src, 1, total));
    CHECK(hipMemset(sink, 0, 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("%d CUs, one persistent 8-wave workgroup per CU, nothing but operand fetches\n", cus);
    printf("%-5s %-9s %6s %9s %5s %6s %5s | %9s %11s %9s\n", "path", "piece", "stride", "foot", "share", "pieces", "depth", "ms", "B/clk/CU", "TB/s");
    struct Case { int path, rows, rowbytes, stride; long long foot; int share, pieces, depth; };
    std::vector<Case> cases;
    const long long W320 = 320LL * 640, W1280 = 320LL * 2560, XP320 = 256LL * 640;   // W tile (K 320 / 1280), X panel
    for (int depth : {1, 2}) {
        // W-like: every workgroup the same 320-row tile, L2-hot
        cases.push_back({0, 8, 128, 640, W320, 1, 5, depth});
        cases.push_back({0, 1, 1024, 0, W320, 1, 5, depth});
        cases.push_back({0, 8, 128, 2560, W1280, 1, 5, depth});
        cases.push_back({0, 1, 1024, 0, W1280, 1, 5, depth});
        // the whole slab from one L2-hot region (X + W = 9 pieces per wave)
        cases.push_back({0, 8, 128, 640, 576LL * 640, 1, 9, depth});
        cases.push_back({0, 1, 1024, 0, 576LL * 640, 1, 9, depth});
        cases.push_back({0, 8, 128, 2560, 576LL * 2560, 1, 9, depth});
        // per-XCD regions (L2-hot, no cross-XCD hot spot)
        cases.push_back({0, 8, 128, 640, 576LL * 640, 2, 9, depth});
        cases.push_back({0, 1, 1024, 0, 576LL * 640, 2, 9, depth});
        // X-like: one region per workgroup; 256 x 8 MiB = 2 GiB (HBM stream), 256 x 0.5 MiB = 128 MiB (Infinity Cache)
        cases.push_back({0, 8, 128, 640, 8LL << 20, 0, 4, depth});
        cases.push_back({0, 1, 1024, 0, 8LL << 20, 0, 4, depth});
        cases.push_back({0, 8, 128, 640, 512LL << 10, 0, 4, depth});
        cases.push_back({0, 1, 1024, 0, 512LL << 10, 0, 4, depth});
        cases.push_back({0, 8, 128, 640, 8LL << 20, 0, 9, depth});
        cases.push_back({0, 1, 1024, 0, 8LL << 20, 0, 9, depth});
    }
    // register path (depth 2 = two slabs of VGPRs): the igemm piece, the contiguous piece, the MFMA fragment pattern
    for (int pieces : {4, 9}) {
        cases.push_back({1, 8, 128, 640, W320, 1, pieces, 2});
        cases.push_back({1, 1, 1024, 0, W320, 1, pieces, 2});
        cases.push_back({1, 32, 32, 640, W320, 1, pieces, 2});
        cases.push_back({1, 8, 128, 640, 8LL << 20, 0, pieces, 2});
        cases.push_back({1, 32, 32, 640, 8LL << 20, 0, pieces, 2});
        cases.push_back({1, 32, 32, 640, 512LL << 10, 0, pieces, 2});
    }
    (void)XP320;
    std::vector<unsigned long long> h(cus);
    for (const Case& c : cases) {
        const int nslab = 2000;
        Args a{src, cyc, sink, c.foot, c.share, c.rows, c.rowbytes, c.stride, c.pieces, nslab, c.depth, c.path};
        if ((c.share == 0 ? (size_t)cus : 8) * (size_t)c.foot > total) continue;
        for (int rep = 0; rep < 2; ++rep) {
            if (rep == 1) CHECK(hipEventRecord(e0, 0));
            if (c.path == 0) hipLaunchKernelGGL(fill_kernel<0>, dim3(cus), dim3(NWV * 64), lds_bytes, 0, a);
            else hipLaunchKernelGGL(fill_kernel<1>, dim3(cus), dim3(NWV * 64), lds_bytes, 0, a);
            if (rep == 1) CHECK(hipEventRecord(e1, 0));
        }
        CHECK(hipDeviceSynchronize());
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        CHECK(hipMemcpy(h.data(), cyc, cus * 8, hipMemcpyDeviceToHost));
        unsigned long long mx = 0;
        for (int i = 0; i < cus; ++i) mx = h[i] > mx ? h[i] : mx;
        const double bytes_cu = (double)nslab * c.pieces * NWV * 1024.0;
        char piece[32];
        snprintf(piece, sizeof piece, "%dx%d", c.rows, c.rowbytes);
        printf("%-5s %-9s %6d %9lld %5d %6d %5d | %9.3f %11.2f %9.2f\n", c.path ? "vgpr" : "lds", piece, c.stride, c.foot, c.share, c.pieces, c.depth, ms,
               bytes_cu / (double)mx, bytes_cu * cus / (ms * 1e-3) / 1e12);
        fflush(stdout);
    }
    return 0;
}
