// fp32 MFMA issue-rate probe: how many cycles per v_mfma_f32_16x16x4_f32 / 32x32x2 in the accumulator patterns the
// Winograd kernel uses, at 1 and 2 wavefronts per SIMD.   hipcc --offload-arch=gfx950 -O3 -o /tmp/mr tools/ubench/mfma_f32_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NACC, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256, WAVES_PER_SIMD) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; i += 2) {
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
            acc[i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i + 1], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[i], 0, 0, 0);
            acc[i + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b, a, acc[i + 1], 0, 0, 0);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256, 1) void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc[i], 0, 0, 0);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][5];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F> float run(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    float* out; CK(hipMalloc(&out, 1024 * 256 * 4));
    const int iters = 2000;
    {   // 1 wave per SIMD: 256 blocks of 256 threads; 32 accumulators of 16x16 (128 regs)
        float ms = run([&] { hipLaunchKernelGGL((k16<32, 1>), dim3(256), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); });
        double mf = (double)iters * 64;  // MFMAs per wave
        printf("16x16x4 32acc 1 wave/SIMD: %.1f cycles/MFMA (at 2.4 GHz)\n", ms * 1e-3 * 2.4e9 / mf);
    }
    {   // 2 waves per SIMD
        float ms = run([&] { hipLaunchKernelGGL((k16<32, 2>), dim3(512), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); });
        double mf = (double)iters * 64 * 2;
        printf("16x16x4 32acc 2 waves/SIMD: %.1f cycles/MFMA per SIMD\n", ms * 1e-3 * 2.4e9 / mf);
    }
    {
        float ms = run([&] { hipLaunchKernelGGL((k16<2, 2>), dim3(512), dim3(256), 0, 0, out, iters * 16, 1.0f, 2.0f); });
        double mf = (double)iters * 16 * 4 * 2;
        printf("16x16x4 2acc 2 waves/SIMD: %.1f cycles/MFMA per SIMD\n", ms * 1e-3 * 2.4e9 / mf);
    }
    {
        float ms = run([&] { hipLaunchKernelGGL((k32<16>), dim3(256), dim3(256), 0, 0, out, iters, 1.0f, 2.0f); });
        double mf = (double)iters * 32;
        printf("32x32x2 16acc 1 wave/SIMD: %.1f cycles/MFMA\n", ms * 1e-3 * 2.4e9 / mf);
    }
    return 0;
}
