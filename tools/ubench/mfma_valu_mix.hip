// How do an MFMA stream and a VALU stream of two different waves on the same SIMD share the issue port?
// One block of 512 threads per CU: waves 0-3 run MFMAs, waves 4-7 (second wave of each SIMD) run independent VALU adds.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE, int PRIO, int OWN>   // OWN: VALU ops the MFMA wave itself issues after each MFMA; MODE bit0: MFMA waves active, bit1: VALU waves active
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* cyc, int iters, float a0) {
    const int wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_readcyclecounter();
    float s = 0;
    if (wave < 4) {
        if (MODE & 1) {
            f32x4 acc[8];
            float own[8] = {1, 2, 3, 4, 5, 6, 7, 8};
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0 + i, a0 - i, acc[i], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < OWN; ++q) asm volatile("v_add_f32 %0, %0, %1" : "+v"(own[q]) : "v"(a0));
                }
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) s += acc[i][0] + own[i];
        }
    } else {
        if (MODE & 2) {
            if (PRIO) __builtin_amdgcn_s_setprio(3);
            float r[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = a0 + i + threadIdx.x;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a0));
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) s += r[i];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int PRIO, int OWN> int run(const char* label, float* out, unsigned long long* cyc, int iters) {
    hipLaunchKernelGGL((k<MODE, PRIO, OWN>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1.0f);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k<MODE, PRIO, OWN>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1.0f);
    CK(hipDeviceSynchronize());
    unsigned long long h[8];
    CK(hipMemcpy(h, cyc + 8 * 100, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-40s MFMA wave: %.1f cycles/MFMA   VALU wave: %.1f cycles/v_add\n", label, (double)h[0] / (iters * 8.0), (double)h[4] / (iters * 8.0));
    return 0;
}
int main() {
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 256 * 8 * 8));
    const int iters = 4000;
    run<1, 0, 0>("MFMA alone", out, cyc, iters);
    run<2, 0, 0>("VALU alone", out, cyc, iters);
    run<3, 0, 0>("both", out, cyc, iters);
    run<3, 1, 0>("both, VALU wave at s_setprio 3", out, cyc, iters);
    run<1, 0, 1>("MFMA + 1 own VALU per MFMA, alone", out, cyc, iters);
    run<1, 0, 2>("MFMA + 2 own VALU per MFMA, alone", out, cyc, iters);
    run<1, 0, 4>("MFMA + 4 own VALU per MFMA, alone", out, cyc, iters);
    run<1, 0, 6>("MFMA + 6 own VALU per MFMA, alone", out, cyc, iters);
    run<3, 0, 2>("MFMA + 2 own VALU, VALU wave beside", out, cyc, iters);
    return 0;
}
