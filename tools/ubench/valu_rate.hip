// Micro-benchmark: issue cost (cycles per wave-instruction per SIMD) of v_fma_f32, v_exp_f32, v_rcp_f32, v_cvt_pk_bf16_f32,
// v_max3_f32 with 1, 2 and 4 waves per SIMD.  hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) a[i] = __builtin_fmaf(a[i], 0.999f, 0.001f);
                if (OP == 1) a[i] = __builtin_amdgcn_exp2f(a[i] * 0.0f - 0.5f) + a[i] * 0.0f;   // exp + 2 cheap ops (subtracted below)
                if (OP == 2) a[i] = __builtin_amdgcn_rcpf(a[i] + 2.0f);
                if (OP == 3) a[i] = fmaxf(fmaxf(a[i], a[(i + 1) & 7]), 0.5f);
                if (OP == 4) a[i] = a[i] * 0.0f - 0.5f + a[i] * 0.0f;                              // the 2 cheap ops of OP 1 alone
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OP>
double run(int blocks_per_cu, int iters, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * blocks_per_cu;
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(grid), dim3(256), 0, 0, d, iters);
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
    const char* names[] = {"v_fma_f32", "exp2+2ops", "v_rcp(+add)", "v_max3/max", "2ops"};
    for (int bpc : {1, 2, 4}) {   // 256-thread block = 4 waves = 1 wave per SIMD; bpc blocks per CU -> bpc waves per SIMD
        double ms[5] = {run<0>(bpc, iters, d), run<1>(bpc, iters, d), run<2>(bpc, iters, d), run<3>(bpc, iters, d), run<4>(bpc, iters, d)};
        for (int op = 0; op < 5; ++op) {
            const double instr_per_wave = (double)iters * 32;                       // statements per wave
            const double ns_per = ms[op] * 1e6 / (instr_per_wave * bpc);            // per statement per SIMD
            printf("waves/SIMD=%d %-12s %8.2f ms  %6.2f ns per statement per SIMD (x clock GHz = cycles)\n", bpc, names[op], ms[op], ns_per);
        }
    }
    return 0;
}
