// How do an MFMA stream and a VALU stream of two different waves on the same SIMD share the issue port?
// One block of 512 threads per CU: waves 0-3 run MFMAs, waves 4-7 (second wave of each SIMD) run independent VALU adds.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE, int PRIO, int OWN, int KIND>   // KIND of the sibling stream: 0 v_add, 1 s_add, 2 ds_read_b32, 3 s_nop; OWN: VALU ops the MFMA wave itself issues after each MFMA; MODE bit0: MFMA waves active, bit1: VALU waves active
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* cyc, int iters, float a0) {
    __shared__ float ldsbuf[256];
    ldsbuf[threadIdx.x & 255] = 1.0f;
    __syncthreads();
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
            unsigned sr[8] = {0, 1, 2, 3, 4, 5, 6, 7};
            const unsigned ldsaddr = (threadIdx.x & 63) * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i) r[i] = a0 + i + threadIdx.x;
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a0));
                    else if (KIND == 1) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sr[i]));
                    else if (KIND == 2) asm volatile("ds_read_b32 %0, %1" : "=v"(r[i]) : "v"(ldsaddr) : "memory");
                    else asm volatile("s_nop 0");
                }
                if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) s += r[i] + sr[i];
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int MODE, int PRIO, int OWN, int KIND = 0> int run(const char* label, float* out, unsigned long long* cyc, int iters) {
    hipLaunchKernelGGL((k<MODE, PRIO, OWN, KIND>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1.0f);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL((k<MODE, PRIO, OWN, KIND>), dim3(256), dim3(512), 0, 0, out, cyc, iters, 1.0f);
    CK(hipDeviceSynchronize());
    unsigned long long h[8];
    CK(hipMemcpy(h, cyc + 8 * 100, sizeof(h), hipMemcpyDeviceToHost));
    printf("%-40s MFMA wave: %.1f cycles/MFMA   sibling wave: %.1f cycles/instr\n", label, (double)h[0] / (iters * 8.0), (double)h[4] / (iters * 8.0));
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
    run<2, 0, 0, 1>("SALU alone", out, cyc, iters);
    run<3, 0, 0, 1>("MFMA wave + SALU wave", out, cyc, iters);
    run<2, 0, 0, 2>("LDS reads alone", out, cyc, iters);
    run<3, 0, 0, 2>("MFMA wave + LDS-read wave", out, cyc, iters);
    run<2, 0, 0, 3>("s_nop alone", out, cyc, iters);
    run<3, 0, 0, 3>("MFMA wave + s_nop wave", out, cyc, iters);
    return 0;
}
