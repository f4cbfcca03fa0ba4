// What slows the Winograd MFMA loop below the 32-cycle issue rate?  One wave per SIMD (256 blocks x 256 threads),
// 32 accumulators, per "xi": optional ds_read_b128 of the B operands, 4 MFMAs (acc0, acc1, acc0, acc1).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int LDSB, int PF, int DISTINCT_A>
__global__ __launch_bounds__(256, 2) void k(float* out, int iters, float a0) {
    __shared__ __attribute__((aligned(16))) float lds[16 * 2 * 64 * 4];
    for (int i = threadIdx.x; i < 16 * 2 * 64 * 4; i += 256) lds[i] = 1.0f + i;
    __syncthreads();
    f32x4 acc[16][2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i][0] = (f32x4){0, 0, 0, 0}; acc[i][1] = (f32x4){0, 0, 0, 0}; }
    float2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = make_float2(a0 + i + threadIdx.x, a0 - i);
    const float* ub = lds + ((threadIdx.x >> 6 & 1) * 64 + (threadIdx.x & 63)) * 4;
    for (int it = 0; it < iters; ++it) {
        float4 bq[16];
        if (LDSB) {
#pragma unroll
            for (int xi = 0; xi < PF; ++xi) bq[xi] = *reinterpret_cast<const float4*>(ub + xi * 512);
            __builtin_amdgcn_sched_group_barrier(0x100, PF, 0);
        }
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            if (LDSB && xi + PF < 16) bq[xi + PF] = *reinterpret_cast<const float4*>(ub + (xi + PF) * 512);
            const float4 b = LDSB ? bq[xi] : make_float4(a0, a0 + 1, a0 + 2, a0 + 3);
            const float2 a = DISTINCT_A ? v[xi] : v[0];
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.z, acc[xi][1], 0, 0, 0);
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.w, acc[xi][1], 0, 0, 0);
            if (LDSB) { if (xi + PF < 16) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0); __builtin_amdgcn_sched_group_barrier(0x008, 4, 0); }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0][0] + acc[i][1][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename F> float run(F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(); hipDeviceSynchronize();
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
#define RUN(LDSB, PF, DA, NBLK, LABEL) { float ms = run([&] { hipLaunchKernelGGL((k<LDSB, PF, DA>), dim3(NBLK), dim3(256), 0, 0, out, iters, 1.0f); }); \
    printf("%-46s %.1f cycles/MFMA per SIMD\n", LABEL, ms * 1e-3 * 2.4e9 / ((double)iters * 64 * (NBLK / 256))); }
int main() {
    float* out; CK(hipMalloc(&out, 1024 * 256 * 4));
    const int iters = 2000;
    RUN(0, 1, 0, 256, "regs only, same A, 1 wave/SIMD");
    RUN(0, 1, 1, 256, "regs only, distinct A per xi, 1 wave/SIMD");
    RUN(1, 1, 1, 256, "LDS B (prefetch 1), distinct A, 1 wave/SIMD");
    RUN(1, 3, 1, 256, "LDS B (prefetch 3), distinct A, 1 wave/SIMD");
    RUN(1, 1, 1, 512, "LDS B (prefetch 1), distinct A, 2 waves/SIMD");
    RUN(1, 3, 1, 512, "LDS B (prefetch 3), distinct A, 2 waves/SIMD");
    return 0;
}
