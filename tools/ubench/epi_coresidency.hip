// Reduced reproducer attempt for the round-5 co-residency finding: the LayerNorm-fold epilogue of the 4-wave 128 x 128 igemm
// tile (FEAT 1 / 3) gave a dozen lanes of one wave a wrong per-COLUMN operand when a second workgroup shared its CU.  This kernel puts
// the suspected ingredients side by side on one CU, outside the igemm:
//   victim workgroups (even blockIdx): the epilogue's LDS sequence - per-column vectors staged with ds_write_b32, a barrier, then per
//     "quad" a 64-bit row-operand read + four 128-bit BROADCAST column reads (all 32 lanes of a half read one address) feeding VALU
//     in the MFMA layout, with the destination registers re-used immediately (the "operand sampled late" hypothesis: a VALU write to a
//     register of an in-flight ds_read_b128) - every loaded dword checked against the value its address must hold;
//   aggressor workgroups (odd blockIdx): the K loop's traffic - LDS-DMA (buffer_load ... lds) into a double buffer, ds_read_b128
//     fragment reads, MFMAs - in a tight loop, never touching the victim's LDS.
// Both take 68 KiB of LDS (two workgroups per CU, as the tile did).  Output: mismatching dwords per lane group.
// build + run:  hipcc --offload-arch=gfx950 -O3 -o epi_coresidency epi_coresidency.hip && ./epi_coresidency [seconds]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
constexpr int LDS_BYTES = 68 * 1024, BN = 128;

__global__ __launch_bounds__(256) void probe(const float* gvec, const unsigned short* gw, unsigned* bad, unsigned* lanes, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    if (blockIdx.x & 1) {                               // ---- aggressor: DMA + fragment reads + MFMA ----
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)gw, 0, 1 << 20, 0x00020000);
        f32x16_t acc = {0};
        for (int it = 0; it < iters * 8; ++it) {
            char* buf = smem + (it & 1) * 32768;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(buf + (wave * 8 + i) * 1024), 16, lane * 16,
                                                         ((it * 8 + i) & 63) * 1024, 0, 0);
            const char* rd = smem + ((it + 1) & 1) * 32768;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const bf16x8_t a = *(const bf16x8_t*)(rd + ((wave * 8 + k) & 31) * 1024 + lane * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, a, acc, 0, 0, 0);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (acc[0] == 12345.f) bad[1] = 1;
        return;
    }
    // ---- victim: the fold epilogue's LDS traffic ----
    float* vbias = (float*)(smem + 2 * 32768);         // the tile's layout: two K-slab buffers, then the vectors
    float* vaux = vbias + BN;
    float* lnrow = vbias + 3 * BN;                      // per-row (mean, rstd), as round 4's variant staged them
    unsigned nbad = 0, lanemask_lo = 0;
    for (int it = 0; it < iters; ++it) {
        const float salt = (float)(it & 1023);
        if (tid < BN) {
            vbias[tid] = gvec[tid] + salt;
            vaux[tid] = gvec[BN + tid] - salt;
        }
        if (tid < 128) *(float2*)(lnrow + 2 * tid) = make_float2(1000.f + tid + salt, 2000.f + tid - salt);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll 4
        for (int q = 0; q < 16; ++q) {
            const int nv = ((wave & 1) * 64 + (q & 3) * 8 + 4 * lhi + (q >> 2) * 32) & (BN - 4);
            const float2 row = *(const float2*)(lnrow + 2 * ((wave >> 1) * 64 + (q & 1) * 32 + l31));
            float4 bv = *(const float4*)(vbias + nv), sv = *(const float4*)(vaux + nv);
            float4 bg = *(const float4*)(vbias + ((nv + 16) & (BN - 4))), sg = *(const float4*)(vaux + ((nv + 16) & (BN - 4)));
            // consume + immediately recycle the destination registers (what the epilogue's VALU does in the MFMA layout)
            const float e0 = gvec[nv] + salt, e1 = gvec[BN + nv] - salt;
            const float e2 = gvec[(nv + 16) & (BN - 4)] + salt, e3 = gvec[BN + ((nv + 16) & (BN - 4))] - salt;
            const int rr = (wave >> 1) * 64 + (q & 1) * 32 + l31;
            bool ok = bv.x == e0 && sv.x == e1 && bg.x == e2 && sg.x == e3 && row.x == 1000.f + rr + salt && row.y == 2000.f + rr - salt;
            ok = ok && bv.y == gvec[nv + 1] + salt && bv.z == gvec[nv + 2] + salt && bv.w == gvec[nv + 3] + salt;
            ok = ok && sv.y == gvec[BN + nv + 1] - salt && sv.z == gvec[BN + nv + 2] - salt && sv.w == gvec[BN + nv + 3] - salt;
            bv.x = row.x * sv.x;                        // VALU write into the just-read tuples
            sv.y = bv.x + row.y;
            asm volatile("" ::"v"(bv.x), "v"(sv.y));
            if (!ok) {
                ++nbad;
                lanemask_lo |= 1u << (lane & 15);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
    if (nbad) {
        atomicAdd(bad, nbad);
        atomicOr(lanes, lanemask_lo);
    }
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 5.0;
    float* gvec;
    unsigned short* gw;
    unsigned *bad, *lanes;
    hipMalloc(&gvec, 2 * BN * 4);
    hipMalloc(&gw, 1 << 21);
    hipMalloc(&bad, 8);
    hipMalloc(&lanes, 4);
    float h[2 * BN];
    for (int i = 0; i < 2 * BN; ++i) h[i] = 0.25f * i + 7.f;
    hipMemcpy(gvec, h, sizeof(h), hipMemcpyHostToDevice);
    hipMemset(gw, 0x3c, 1 << 21);
    hipMemset(bad, 0, 8);
    hipMemset(lanes, 0, 4);
    hipFuncSetAttribute((const void*)probe, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    long launches = 0;
    const auto t0 = std::chrono::steady_clock::now();
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
        hipLaunchKernelGGL(probe, dim3(512), dim3(256), LDS_BYTES, 0, gvec, gw, bad, lanes, 200);     // 2 workgroups per CU: one of each kind
        hipDeviceSynchronize();
        ++launches;
    }
    unsigned hb[2], hl;
    hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(&hl, lanes, 4, hipMemcpyDeviceToHost);
    printf("%ld launches x 256 victim workgroups x 200 iterations x 16 quads: %u mismatching lane-quads (lane-in-16 mask 0x%04x)%s\n", launches, hb[0], hl,
           hb[0] ? "" : "  -> not reproduced by this reduction");
    return 0;
}
