// How many vector-ALU / LDS instructions of the kind a Winograd input transform needs (v_fma_mix_f32, v_sub_f32, v_cvt_pk_f16_f32,
// ds_read_b128) hide under one v_mfma_f32_32x32x16_f16 of the SAME wave - with one wave per SIMD (256-thread workgroup, 512 registers) and
// with two (512-thread workgroup)?  Prints shader cycles per MFMA for K fillers per MFMA.  Decides the wave geometry of csrc/conv_w16.hip.
// Build + run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_f16_fillers tools/ubench/mfma_f16_fillers.hip && /tmp/mfma_f16_fillers
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// KIND: 0 v_fma_mix_f32 (f16 + f16 -> f32), 1 v_sub_f32, 2 v_cvt_pk_f16_f32, 3 the transform's mix (per 8: 3 fma_mix, 2 sub, 1 cvt_pk, 2 fma_mixlo),
//       4 v_pk_add_f32
// LDS: one ds_read_b128 every LDS-th MFMA (0: none)
template <int THREADS, int K, int KIND, int LDS>
__global__ __launch_bounds__(THREADS, THREADS / 256) void k(float* out, unsigned long long* cyc, int iters, float a0) {
    __shared__ __attribute__((aligned(16))) float ldsbuf[4096];
    for (int i = threadIdx.x; i < 4096; i += THREADS) ldsbuf[i] = 1.0f + i;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[i][j] = 0.0f;
    h8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(a0 + j + lane); b[j] = (_Float16)(a0 - j); }
    float f[8];
    unsigned p[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { f[q] = a0 + q + lane; p[q] = 0x3c003c00u + q; }
    f32x4 lr = {0, 0, 0, 0};
    const unsigned ldsaddr = (unsigned)(lane * 16);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (LDS && (i % LDS) == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(lr) : "v"(ldsaddr) : "memory");
#pragma unroll
            for (int q = 0; q < K; ++q) {
                const int kind = KIND == 3 ? (q % 8 < 3 ? 0 : q % 8 < 5 ? 1 : q % 8 < 6 ? 2 : 5) : KIND;
                if (kind == 0) asm volatile("v_fma_mix_f32 %0, %1, 1.0, %1 op_sel:[0,0,1] op_sel_hi:[1,0,1]" : "=v"(f[q % 8]) : "v"(p[q % 8]));
                else if (kind == 1) asm volatile("v_sub_f32 %0, %1, %2" : "=v"(f[q % 8]) : "v"(f[(q + 1) % 8]), "v"(f[(q + 2) % 8]));
                else if (kind == 2) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(p[q % 8]) : "v"(f[(q + 1) % 8]), "v"(f[(q + 2) % 8]));
                else if (kind == 4) asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(*reinterpret_cast<double*>(&f[2 * (q % 4)])) : "v"(*reinterpret_cast<double*>(&f[2 * ((q + 1) % 4)])), "v"(*reinterpret_cast<double*>(&f[2 * ((q + 2) % 4)])));
                else asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(p[q % 8]) : "v"(p[(q + 1) % 8]), "v"(f[q % 8]));
            }
        }
        if (LDS) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = lr.x;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + f[i] + (float)p[i];
    out[blockIdx.x * THREADS + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int THREADS, int K, int KIND, int LDS> int run(float* out, unsigned long long* cyc, int iters) {
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<THREADS, K, KIND, LDS>), dim3(256), dim3(THREADS), 0, 0, out, cyc, iters, 1.0f);
        CK(hipDeviceSynchronize());
    }
    unsigned long long h[8];
    CK(hipMemcpy(h, cyc + 8 * 100, sizeof(h), hipMemcpyDeviceToHost));
    static const char* names[] = {"fma_mix", "sub", "cvt_pk", "mix", "pk_add_f32"};
    // per-SIMD figure: with two waves per SIMD each wave's MFMAs take turns, so cycles per (wave) MFMA / 2 = cycles per MFMA of the pipe
    const double per = (double)h[0] / (iters * 8.0) / (THREADS / 256);
    printf("{\"waves_per_simd\": %d, \"fillers_per_mfma\": %d, \"kind\": \"%s\", \"ds_read_b128_every\": %d, \"cycles_per_mfma_per_simd\": %.1f}\n", THREADS / 256, K, names[KIND], LDS, per);
    return 0;
}
#define ROW(T, KIND, LDS) run<T, 0, KIND, LDS>(out, cyc, iters); run<T, 1, KIND, LDS>(out, cyc, iters); run<T, 2, KIND, LDS>(out, cyc, iters); run<T, 3, KIND, LDS>(out, cyc, iters); \
    run<T, 4, KIND, LDS>(out, cyc, iters); run<T, 5, KIND, LDS>(out, cyc, iters); run<T, 6, KIND, LDS>(out, cyc, iters); run<T, 8, KIND, LDS>(out, cyc, iters); run<T, 12, KIND, LDS>(out, cyc, iters)
int main() {
    float* out; unsigned long long* cyc;
    CK(hipMalloc(&out, 256 * 512 * 4)); CK(hipMalloc(&cyc, 256 * 8 * 8));
    const int iters = 2000;
    ROW(256, 0, 0); ROW(256, 1, 0); ROW(256, 2, 0); ROW(256, 3, 0); ROW(256, 4, 0); ROW(256, 3, 2); ROW(256, 3, 1);
    ROW(512, 3, 0); ROW(512, 3, 2); ROW(512, 1, 0);
    return 0;
}
