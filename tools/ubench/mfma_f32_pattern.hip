// Solo-wave issue rate of v_mfma_f32_16x16x4_f32 in the Winograd kernel's accumulation pattern: per transform position two
// accumulators, four MFMAs (acc0, acc1, acc0, acc1 - the 3rd depends on the 1st, the 4th on the 2nd), 16 positions in a row.
//   PAT 0: all 32 accumulators independent order (a0 a1 a0' a1' with a0' = another position)  -> no dependency within 4
//   PAT 1: the kernel's order (a0 a1 a0 a1)
//   PAT 2: kernel's order with an s_nop 0 after every MFMA
//   PAT 3: order a0 a1 b0 b1 a0 a1 b0 b1 (two positions interleaved: dependent MFMAs are 4 apart)
//   PAT 4: as 3 with s_nop 0 after every MFMA
// WAVES = 1 or 2 per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
#define MF(acc, a, b) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0)
#define NOP() asm volatile("s_nop 0")
template <int PAT, int WAVES>
__global__ __launch_bounds__(256 * WAVES, 1) void k(float* out, unsigned long long* cyc, int iters, float a0) {
    f32x4 acc[16][2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i][0] = (f32x4){0, 0, 0, 0}; acc[i][1] = (f32x4){0, 0, 0, 0}; }
    float a[4] = {a0, a0 + 1, a0 + 2, a0 + 3}, b[2] = {a0 - 1, a0 - 2};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (PAT == 0) {
#pragma unroll
            for (int xi = 0; xi < 16; xi += 2) {
                MF(acc[xi][0], a[0], b[0]); MF(acc[xi][1], a[1], b[0]); MF(acc[xi + 1][0], a[2], b[1]); MF(acc[xi + 1][1], a[3], b[1]);
                MF(acc[xi + 1][0], a[0], b[0]); MF(acc[xi + 1][1], a[1], b[0]); MF(acc[xi][0], a[2], b[1]); MF(acc[xi][1], a[3], b[1]);
            }
        } else if (PAT == 1 || PAT == 2) {
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) {
                MF(acc[xi][0], a[0], b[0]); if (PAT == 2) NOP();
                MF(acc[xi][1], a[1], b[0]); if (PAT == 2) NOP();
                MF(acc[xi][0], a[2], b[1]); if (PAT == 2) NOP();
                MF(acc[xi][1], a[3], b[1]); if (PAT == 2) NOP();
            }
        } else {
#pragma unroll
            for (int xi = 0; xi < 16; xi += 2) {
                MF(acc[xi][0], a[0], b[0]); if (PAT == 4) NOP();
                MF(acc[xi][1], a[1], b[0]); if (PAT == 4) NOP();
                MF(acc[xi + 1][0], a[0], b[0]); if (PAT == 4) NOP();
                MF(acc[xi + 1][1], a[1], b[0]); if (PAT == 4) NOP();
                MF(acc[xi][0], a[2], b[1]); if (PAT == 4) NOP();
                MF(acc[xi][1], a[3], b[1]); if (PAT == 4) NOP();
                MF(acc[xi + 1][0], a[2], b[1]); if (PAT == 4) NOP();
                MF(acc[xi + 1][1], a[3], b[1]); if (PAT == 4) NOP();
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0][0] + acc[i][1][1];
    out[blockIdx.x * 256 * WAVES + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}
static float* g_out; static unsigned long long* g_cyc;
template <int PAT, int WAVES> int run() {
    const int iters = 500;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<PAT, WAVES>), dim3(256), dim3(256 * WAVES), 0, 0, g_out, g_cyc, iters, 1.0f); CK(hipDeviceSynchronize()); }
    unsigned long long h[8];
    CK(hipMemcpy(h, g_cyc + 8 * 100, sizeof(h), hipMemcpyDeviceToHost));
    printf("{\"pattern\": %d, \"waves_per_simd\": %d, \"cycles_per_mfma_wave0\": %.2f, \"cycles_per_mfma_lastwave\": %.2f}\n", PAT, WAVES,
           (double)h[0] / (iters * 64.0), (double)h[4 * (WAVES - 1)] / (iters * 64.0));
    return 0;
}
int main() {
    CK(hipMalloc(&g_out, 256 * 512 * 4)); CK(hipMalloc(&g_cyc, 256 * 8 * 8));
    run<0, 1>(); run<1, 1>(); run<2, 1>(); run<3, 1>(); run<4, 1>();
    run<0, 2>(); run<1, 2>(); run<2, 2>(); run<3, 2>(); run<4, 2>();
    return 0;
}
