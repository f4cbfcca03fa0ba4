// What does a non-MFMA instruction cost next to fp32 MFMAs on gfx950?  One workgroup per CU, WAVES wavefronts per SIMD
// (1 or 2), every wave running the SAME stream: MFMAs (16x16x4 f32, 32 cycles each, or 32x32x2 f32, 64 cycles each) on
// rotating accumulators with FILL filler instructions of one KIND after each MFMA.  Prints cycles per MFMA of a wave and
// the matrix-pipe utilisation of the SIMD (waves * ideal / measured).  Everything is independent of everything else:
// no dependency stalls, so the numbers are pure issue / pipe-sharing costs.
//   KIND 0 v_add_f32   1 v_pk_add_f32   2 ds_read_b128   3 ds_read_b64   4 s_add_u32   5 s_waitcnt lgkmcnt(15) (no-op wait)
//        6 v_fma_f32    7 v_pk_fma_f32   8 ds_write_b64   9 v_mov_b32   10 s_nop 0   11 s_nop 3
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND>
__device__ __forceinline__ void filler(float& r, f32x2& r2, f32x4& r4, unsigned& sr, unsigned ldsaddr, float a0) {
    if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r) : "v"(a0));
    else if (KIND == 1) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(r2) : "v"(r2));
    else if (KIND == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(r4) : "v"(ldsaddr) : "memory");
    else if (KIND == 3) asm volatile("ds_read_b64 %0, %1" : "=v"(r2) : "v"(ldsaddr) : "memory");
    else if (KIND == 4) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sr));
    else if (KIND == 5) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
    else if (KIND == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r) : "v"(a0));
    else if (KIND == 7) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(r2) : "v"(r2));
    else if (KIND == 8) asm volatile("ds_write_b64 %0, %1" ::"v"(ldsaddr), "v"(r2) : "memory");
    else if (KIND == 10) asm volatile("s_nop 0");
    else if (KIND == 11) asm volatile("s_nop 3");
    else asm volatile("v_mov_b32 %0, %1" : "=v"(r) : "v"(a0));
}

template <int BIG, int FILL, int KIND, int WAVES>
__global__ __launch_bounds__(256 * WAVES, 1) void k(float* out, unsigned long long* cyc, int iters, float a0) {
    __shared__ __attribute__((aligned(16))) float ldsbuf[64 * WAVES * 4 * 4 + 64];
    for (int i = threadIdx.x; i < 64 * WAVES * 4 * 4; i += blockDim.x) ldsbuf[i] = 1.0f;
    __syncthreads();
    const int wave = threadIdx.x >> 6;
    float r[4] = {1, 2, 3, 4};
    f32x2 r2[4] = {{1, 2}, {3, 4}, {5, 6}, {7, 8}};
    f32x4 r4[4];
    unsigned sr[4] = {0, 1, 2, 3};
    const unsigned ldsaddr = threadIdx.x * 16;
    float s = 0;
    unsigned long long t0, t1;
    if (BIG) {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[i][j] = 0;
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0 + i, a0 - i, acc[i], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < FILL; ++q) filler<KIND>(r[q & 3], r2[q & 3], r4[q & 3], sr[q & 3], ldsaddr, a0);
            }
        }
        if (KIND == 2 || KIND == 3 || KIND == 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t1 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    } else {
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0, 0, 0, 0};
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0 + i, a0 - i, acc[i], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < FILL; ++q) filler<KIND>(r[q & 3], r2[q & 3], r4[q & 3], sr[q & 3], ldsaddr, a0);
            }
        }
        if (KIND == 2 || KIND == 3 || KIND == 8) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        t1 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0];
    }
    if (KIND == 2) for (int i = 0; i < 4; ++i) s += r4[i][0];
    for (int i = 0; i < 4; ++i) s += r[i] + r2[i][0] + sr[i];
    out[blockIdx.x * 256 * WAVES + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

static float* g_out;
static unsigned long long* g_cyc;

template <int BIG, int FILL, int KIND, int WAVES> int run() {
    const int iters = 2000;
    const int per_iter = BIG ? 4 : 8;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<BIG, FILL, KIND, WAVES>), dim3(256), dim3(256 * WAVES), 0, 0, g_out, g_cyc, iters, 1.0f);
        CK(hipDeviceSynchronize());
    }
    unsigned long long h[8];
    CK(hipMemcpy(h, g_cyc + 8 * 100, sizeof(h), hipMemcpyDeviceToHost));
    const double c0 = (double)h[0] / (iters * (double)per_iter);
    const double cl = (double)h[4 * (WAVES - 1)] / (iters * (double)per_iter);
    const double ideal = BIG ? 64.0 : 32.0;
    static const char* names[] = {"v_add_f32", "v_pk_add_f32", "ds_read_b128", "ds_read_b64", "s_add_u32", "s_waitcnt", "v_fma_f32", "v_pk_fma_f32", "ds_write_b64", "v_mov_b32", "s_nop 0", "s_nop 3"};
    printf("{\"mfma\": \"%s\", \"waves_per_simd\": %d, \"filler\": \"%s\", \"fillers_per_mfma\": %d, \"cycles_per_mfma_wave0\": %.1f, \"cycles_per_mfma_lastwave\": %.1f, "
           "\"pipe_util\": %.3f, \"extra_cycles_per_filler\": %.2f}\n",
           BIG ? "32x32x2" : "16x16x4", WAVES, names[KIND], FILL, c0, cl, WAVES * ideal / ((c0 + cl) / 2), FILL ? ((c0 + cl) / 2 / WAVES - ideal) / FILL : 0.0);
    return 0;
}

template <int BIG, int WAVES> int sweep() {
    run<BIG, 0, 0, WAVES>();
#define ROW(KIND) run<BIG, 1, KIND, WAVES>(); run<BIG, 2, KIND, WAVES>(); run<BIG, 4, KIND, WAVES>();
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(5) ROW(6) ROW(8) ROW(10) ROW(11)
#undef ROW
    run<BIG, 8, 0, WAVES>();
    return 0;
}

int main() {
    CK(hipMalloc(&g_out, 256 * 512 * 4));
    CK(hipMalloc(&g_cyc, 256 * 8 * 8));
    sweep<0, 1>();
    sweep<0, 2>();
    sweep<1, 1>();
    sweep<1, 2>();
    return 0;
}
