// Does VALU / transcendental work hide behind MFMAs issued by the SAME wave or by ANOTHER wave of the same SIMD?
// hipcc --offload-arch=gfx950 -O3 mfma_overlap.hip -o mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// MODE 0: 4 MFMAs per iteration; 1: 4 MFMAs + NE exps interleaved after each; 2: only the exps; 3: MFMAs + NF fmas; 4: only fmas
template <int MODE, int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f32x16_t acc[4];
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    bf16x8_t a, b;
    for (int e = 0; e < 8; ++e) {
        a[e] = (__bf16)(threadIdx.x * 0.001f);
        b[e] = (__bf16)(1.0f);
    }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            if (MODE == 0 || MODE == 1 || MODE == 3) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                if (MODE == 1 || MODE == 2) v[(m * NV + i) & 7] = __builtin_amdgcn_exp2f(v[(m * NV + i) & 7]);
                if (MODE == 3 || MODE == 4) v[(m * NV + i) & 7] = __builtin_fmaf(v[(m * NV + i) & 7], 0.999f, 0.001f);
            }
            if (MODE == 1 || MODE == 3) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, NV, 0);
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE, int NV>
float run(int bpc, int iters, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(256 * bpc), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV>), dim3(256 * bpc), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 8 * 256 * 4);
    const int iters = 20000;
    for (int bpc : {1, 2}) {
        const double per = 1e6 / ((double)iters * 4 * bpc);   // ns per MFMA slot per SIMD
        printf("waves/SIMD=%d  ns per (MFMA + fillers) slot per SIMD:\n", bpc);
        printf("  mfma only            %6.2f\n", run<0, 0>(bpc, iters, d) * per);
        printf("  2 exp only           %6.2f   mfma + 2 exp   %6.2f\n", run<2, 2>(bpc, iters, d) * per, run<1, 2>(bpc, iters, d) * per);
        printf("  3 exp only           %6.2f   mfma + 3 exp   %6.2f\n", run<2, 3>(bpc, iters, d) * per, run<1, 3>(bpc, iters, d) * per);
        printf("  4 fma only           %6.2f   mfma + 4 fma   %6.2f\n", run<4, 4>(bpc, iters, d) * per, run<3, 4>(bpc, iters, d) * per);
        printf("  8 fma only           %6.2f   mfma + 8 fma   %6.2f\n", run<4, 8>(bpc, iters, d) * per, run<3, 8>(bpc, iters, d) * per);
    }
    return 0;
}
