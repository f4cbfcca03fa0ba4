// The MFMA phase of the Winograd kernel in isolation: 16 transform positions x 4 MFMAs (two accumulators each), the A operands
// streamed from LDS by one ds_read_b128 per position (PF positions ahead), optional patch reads (ds_read_b64) and optional
// LDS-DMA pieces, in several instruction orders.  One or two waves per SIMD.  Prints cycles per MFMA.
//   ORDER 0: reads + wait clustered before the four MFMAs (round-1 order)     ORDER 1: one instruction per MFMA gap
//   ORDER 4 / 5: the operand reads issued in pairs / quads (every 8 / 16 MFMAs)
//   ORDER 2: no LDS traffic at all (A operands constant)                      ORDER 3: as 1 but the wait right after the b128 of 2 positions ago
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define SB() __builtin_amdgcn_sched_barrier(0)
template <int ORDER, int PATCH, int WAVES, int PF>
__global__ __launch_bounds__(256 * WAVES, 1) void k(float* out, unsigned long long* cyc, int iters, float a0) {
    __shared__ __attribute__((aligned(16))) float U[16 * 64 * WAVES * 4 * 4 / 4 + 4096];
    __shared__ __attribute__((aligned(16))) float R[4096];
    for (int i = threadIdx.x; i < 16 * 64 * WAVES * 4 + 4096; i += blockDim.x) U[i] = 0.001f * (i & 63);
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) R[i] = 0.002f * (i & 31);
    __syncthreads();
    f32x4 acc[16][2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i][0] = (f32x4){0, 0, 0, 0}; acc[i][1] = (f32x4){0, 0, 0, 0}; }
    f32x2 v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = (f32x2){a0 + i, a0 - i};
    const float* ub = U + (threadIdx.x & 63) * 4 + (threadIdx.x >> 6) * 256;
    const float* pp = R + (threadIdx.x & 63) * 2;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        float4 bq[PF + 4];
        if (ORDER != 2) {
#pragma unroll
            for (int i = 0; i < PF; ++i) bq[i] = *reinterpret_cast<const float4*>(ub + i * 1024);
        } else {
#pragma unroll
            for (int i = 0; i <= PF; ++i) bq[i] = make_float4(a0, a0 + 1, a0 + 2, a0 + 3);
        }
        SB();
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            if (ORDER == 4 && (xi & 1) == 0) {                       // two positions' operands at once, PF positions ahead
                if (xi + PF < 16) bq[(xi + PF) % (PF + 2)] = *reinterpret_cast<const float4*>(ub + (xi + PF) * 1024);
                if (xi + PF + 1 < 16) bq[(xi + PF + 1) % (PF + 2)] = *reinterpret_cast<const float4*>(ub + (xi + PF + 1) * 1024);
            }
            if (ORDER == 5 && (xi & 3) == 0) {                       // four at once
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if (xi + PF + u < 16) bq[(xi + PF + u) % (PF + 4)] = *reinterpret_cast<const float4*>(ub + (xi + PF + u) * 1024);
            }
            if (ORDER == 0) {
                if (xi + PF < 16) bq[(xi + PF) % (PF + 1)] = *reinterpret_cast<const float4*>(ub + (xi + PF) * 1024);
                if (PATCH && (xi & 1) == 0) { v[(xi + 14) & 15] = *reinterpret_cast<const f32x2*>(pp + xi * 128); v[(xi + 15) & 15] = *reinterpret_cast<const f32x2*>(pp + xi * 128 + 400); }
            }
            const float4 b = bq[xi % (ORDER == 4 ? PF + 2 : ORDER == 5 ? PF + 4 : PF + 1)];
            SB();
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, v[xi].x, acc[xi][0], 0, 0, 0);
            SB();
            if (ORDER == 1 || ORDER == 3) { if (xi + PF < 16) bq[(xi + PF) % (PF + 1)] = *reinterpret_cast<const float4*>(ub + (xi + PF) * 1024); }
            SB();
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, v[xi].x, acc[xi][1], 0, 0, 0);
            SB();
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, v[xi].y, acc[xi][0], 0, 0, 0);
            SB();
            if ((ORDER == 1 || ORDER == 3) && PATCH && xi >= 1) v[xi - 1] = *reinterpret_cast<const f32x2*>(pp + xi * 128);
            SB();
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.w, v[xi].y, acc[xi][1], 0, 0, 0);
            SB();
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0][0] + acc[i][1][1] + v[i].x;
    out[blockIdx.x * 256 * WAVES + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
static float* g_out; static unsigned long long* g_cyc;
template <int ORDER, int PATCH, int WAVES, int PF> int run() {
    const int iters = 400;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<ORDER, PATCH, WAVES, PF>), dim3(256), dim3(256 * WAVES), 0, 0, g_out, g_cyc, iters, 1.0f); CK(hipDeviceSynchronize()); }
    unsigned long long h[8];
    CK(hipMemcpy(h, g_cyc + 8 * 100, sizeof(h), hipMemcpyDeviceToHost));
    printf("{\"order\": %d, \"patch_reads\": %d, \"waves_per_simd\": %d, \"pf\": %d, \"cycles_per_mfma_wave0\": %.2f, \"cycles_per_mfma_lastwave\": %.2f}\n", ORDER, PATCH, WAVES, PF,
           (double)h[0] / (iters * 64.0), (double)h[4 * (WAVES - 1)] / (iters * 64.0));
    return 0;
}
int main() {
    CK(hipMalloc(&g_out, 256 * 512 * 4)); CK(hipMalloc(&g_cyc, 256 * 8 * 8));
    run<2, 0, 1, 2>(); run<0, 0, 1, 2>(); run<0, 1, 1, 2>(); run<1, 0, 1, 2>(); run<1, 1, 1, 2>(); run<1, 1, 1, 3>(); run<0, 1, 1, 4>();
    run<4, 0, 1, 2>(); run<5, 0, 1, 4>(); run<4, 0, 2, 2>(); run<5, 0, 2, 4>();
    run<2, 0, 2, 2>(); run<0, 0, 2, 2>(); run<0, 1, 2, 2>(); run<1, 0, 2, 2>(); run<1, 1, 2, 2>(); run<1, 1, 2, 3>(); run<0, 1, 2, 4>();
    return 0;
}
