// Two waves of one SIMD each run the Winograd MFMA phase (per xi: one ds_read_b128 + 4 MFMAs, 64 MFMAs per phase) after a
// common barrier: do their MFMA streams interleave or does one wave go first?  Prints, per mode, when each wave finished its
// phase relative to the barrier (cycles, median over phases).  Modes: plain; the second wave at higher priority; both waves
// yielding (s_sleep 0 / s_nop) after every xi.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <algorithm>
#include <vector>
#pragma clang diagnostic ignored "-Wunused-value"
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(512, 1) void k(float* out, unsigned long long* stamps, int phases, float a0) {
    __shared__ __attribute__((aligned(16))) float lds[16 * 2 * 64 * 4];
    for (int i = threadIdx.x; i < 16 * 2 * 64 * 4; i += 512) lds[i] = 1.0f + i;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x4 acc[16][2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i][0] = (f32x4){0, 0, 0, 0}; acc[i][1] = (f32x4){0, 0, 0, 0}; }
    float2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = make_float2(a0 + i + threadIdx.x, a0 - i);
    const float* ub = lds + ((wave & 1) * 64 + lane) * 4;
    if (MODE == 1 && wave >= 4) __builtin_amdgcn_s_setprio(3);
    for (int p = 0; p < phases; ++p) {
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        const unsigned long long t0 = __builtin_readcyclecounter();
        float4 bq[16];
#pragma unroll
        for (int xi = 0; xi < 2; ++xi) bq[xi] = *reinterpret_cast<const float4*>(ub + xi * 512);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            if (xi + 2 < 16) bq[xi + 2] = *reinterpret_cast<const float4*>(ub + (xi + 2) * 512);
            const float4 b = bq[xi];
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, v[xi].x, acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, v[xi].x, acc[xi][1], 0, 0, 0);
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, v[xi].y, acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.w, v[xi].y, acc[xi][1], 0, 0, 0);
            if (MODE == 2) __builtin_amdgcn_s_sleep(0);
            if (MODE == 3) asm volatile("s_nop 7\n\ts_nop 7");
            __builtin_amdgcn_sched_barrier(0);
        }
        const unsigned long long t1 = __builtin_readcyclecounter();
        if (lane == 0 && blockIdx.x == 100) stamps[(size_t)p * 8 + wave] = t1 - t0;
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0][0] + acc[i][1][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE> int run(const char* label, float* out, unsigned long long* st, int phases) {
    hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(512), 0, 0, out, st, phases, 1.0f);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(phases * 8);
    CK(hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost));
    printf("%-46s", label);
    for (int w = 0; w < 8; ++w) {
        std::vector<unsigned long long> c;
        for (int p = 4; p < phases; ++p) c.push_back(h[(size_t)p * 8 + w]);
        std::sort(c.begin(), c.end());
        printf(" w%d %5llu", w, c[c.size() / 2]);
    }
    printf("\n");
    return 0;
}
int main() {
    float* out; unsigned long long* st;
    const int phases = 64;
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&st, phases * 8 * 8));
    run<0>("plain", out, st, phases);
    run<1>("waves 4-7 at s_setprio 3", out, st, phases);
    run<2>("s_sleep 0 after every xi", out, st, phases);
    run<3>("16 nop cycles after every xi", out, st, phases);
    return 0;
}
