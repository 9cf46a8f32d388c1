// stream_tile: what can the memory system deliver for the K <= 640 GEMMs' access pattern, as a function of the bytes a CU
// keeps in flight?  (DESIGN.md "next (2)": is a persistent RING worth building?)
//
// One persistent workgroup per CU (8 waves) walks 256-row tiles of a row-major bf16 matrix X [M][K] exactly as the igemm
// does: K/64 slabs of 128 bytes per row, each slab moved HBM -> LDS by `buffer_load_dwordx4 ... lds` (1 KB per wave
// instruction), DEPTH slabs in flight (1 = the shipped double buffer, 3 = the 4-slot ring), an optional busy loop of
// `spin` clocks per slab standing in for the MFMA time, then the tile's 256 x N bf16 outputs stored as whole 128-byte row
// segments (the row-major store sequence).  No arithmetic: the bytes that land are summed so that nothing is optimised away.
// Build + run: tools/stream_tile.py (hipcc --offload-arch=gfx950, plain HIP runtime, no torch).
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

constexpr int BM = 256;           // rows per tile
constexpr int SLAB = BM * 128;    // bytes of one 64-column slab of the tile (32 KB)
constexpr int NWV = 8;
constexpr int MAXD = 4;           // LDS ring slots

struct Args {
    const uint16_t* X;    // [M][K] bf16
    uint16_t* C;          // [M][N] bf16
    unsigned long long* sink;
    int M, K, N, depth, spin, stores;
};

template <int DEPTH>
__global__ __launch_bounds__(NWV * 64, 1) void stream_kernel(Args p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles = p.M / BM, nslab = p.K / 64;
    const int rg = lane >> 3, pc = lane & 7;       // 8 rows per wave instruction, 8 x 16-byte chunks per 128-byte row
    unsigned acc = 0;
    for (int t = blockIdx.x; t < tiles; t += gridDim.x) {
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc((void*)(p.X + (long long)t * BM * p.K), 0, 0x7ffffff0, 0x00020000);
        auto issue = [&](int s) {   // slab s -> ring slot s % MAXD: 32 wave instructions of 1 KB, 4 per wave
            char* base = smem + (s % MAXD) * SLAB;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int g = wave + NWV * i;                 // piece index 0..31, 8 rows each
                const unsigned vo = (unsigned)((g * 8 + rg) * p.K * 2 + pc * 16);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(base + g * 1024), 16, (int)vo,
                                                         s * 128, 0, 0);
            }
        };
        int issued = 0;
        for (; issued < DEPTH && issued < nslab; ++issued) issue(issued);
        for (int s = 0; s < nslab; ++s) {
            // wait until slab s has landed: at most (issued - s - 1) younger slabs (4 instructions each) stay in flight
            const int younger = issued - s - 1;
            if (younger >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (younger == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // (a RAW barrier: __syncthreads() waits for vmcnt(0) first, which would drain the ring on every slab - the
            //  first run of this probe did exactly that and showed no effect of `depth`)
            __builtin_amdgcn_s_barrier();
            // "compute": touch the slab once (one ds_read_b128 per lane) and burn `spin` clocks
            const uint4 v = *(const uint4*)(smem + (s % MAXD) * SLAB + tid * 16);
            acc += v.x ^ v.y ^ v.z ^ v.w;
            if (p.spin > 0) {
                const long long t0 = __builtin_readcyclecounter();
                while (__builtin_readcyclecounter() - t0 < p.spin) {}
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                      // everyone is done with slot s % MAXD
            if (issued < nslab) issue(issued++);
        }
        if (p.stores) {
            // the tile's outputs, row-major: 8 adjacent lanes store one row's 128 contiguous bytes, N / 64 segments per row
            const __amdgpu_buffer_rsrc_t rc =
                __builtin_amdgcn_make_buffer_rsrc((void*)(p.C + (long long)t * BM * p.N), 0, 0x7ffffff0, 0x00020000);
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 val = {acc, (unsigned)t, (unsigned)lane, (unsigned)wave};
            for (int seg = 0; seg < p.N / 64; ++seg) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = (wave + NWV * i) * 8 + rg;
                    __builtin_amdgcn_raw_buffer_store_b128(val, rc, (row * p.N + seg * 64) * 2 + pc * 16, 0, 0);
                    asm volatile("s_nop 7" ::"v"(val) : "memory");     // (the store hazard, see sdv_gemm.hip)
                }
                asm volatile("s_waitcnt vmcnt(40)" ::: "memory");       // (the 6-bit counter: never more than 44 stores in flight)
            }
        }
    }
    if (acc == 0x12345678u) atomicAdd(p.sink, 1ull);
}

static void launch(const Args& a, int depth, int grid, hipStream_t s) {
    const int lds = MAXD * SLAB;
    switch (depth) {
        case 1: hipLaunchKernelGGL(stream_kernel<1>, dim3(grid), dim3(NWV * 64), lds, s, a); break;
        case 2: hipLaunchKernelGGL(stream_kernel<2>, dim3(grid), dim3(NWV * 64), lds, s, a); break;
        case 3: hipLaunchKernelGGL(stream_kernel<3>, dim3(grid), dim3(NWV * 64), lds, s, a); break;
        default: hipLaunchKernelGGL(stream_kernel<4>, dim3(grid), dim3(NWV * 64), lds, s, a); break;
    }
}

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 1048576;      // 256 samples x 4096 tokens
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int lds = MAXD * SLAB;
    CHECK(hipFuncSetAttribute((const void*)stream_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute((const void*)stream_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute((const void*)stream_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute((const void*)stream_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    uint16_t *X, *C;
    unsigned long long* sink;
    const int Kmax = 1280, Nmax = 1280;
    CHECK(hipMalloc(&X, (size_t)M * Kmax * 2));
    CHECK(hipMalloc(&C, (size_t)M * Nmax * 2));
    CHECK(hipMalloc(&sink, 8));
    CHECK(hipMemset(X, 1, (size_t)M * Kmax * 2));
    CHECK(hipMemset(sink, 0, 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    printf("M = %d rows, %d CUs, one persistent workgroup per CU; spin = clocks of stand-in compute per 64-wide slab\n", M, cus);
    printf("%6s %6s %6s %6s %7s | %9s %9s %9s\n", "K", "N", "depth", "spin", "stores", "ms", "read TB/s", "r+w TB/s");
    const int shapes[][2] = {{320, 320}, {640, 640}, {1280, 320}, {320, 1280}};
    for (auto& sh : shapes)
        for (int stores = 0; stores < 2; ++stores)
            for (int spin : {0, 2560, 3800})
                for (int depth = 1; depth <= 4; ++depth) {
                    Args a{X, C, sink, M, sh[0], sh[1], depth, spin, stores};
                    launch(a, depth, cus, 0);           // warm-up
                    CHECK(hipEventRecord(e0, 0));
                    for (int r = 0; r < 3; ++r) launch(a, depth, cus, 0);
                    CHECK(hipEventRecord(e1, 0));
                    CHECK(hipEventSynchronize(e1));
                    float ms = 0;
                    CHECK(hipEventElapsedTime(&ms, e0, e1));
                    ms /= 3;
                    const double rd = (double)M * sh[0] * 2, wr = stores ? (double)M * sh[1] * 2 : 0;
                    printf("%6d %6d %6d %6d %7d | %9.3f %9.2f %9.2f\n", sh[0], sh[1], depth, spin, stores, ms, rd / ms * 1e-9,
                           (rd + wr) / ms * 1e-9);
                }
    return 0;
}
