// What does an LDS-DMA instruction cost a workgroup that is busy with MFMAs?  8 waves (2 per SIMD), each iteration = 36 fp32 MFMAs
// per wave with their A operands streamed from LDS (one ds_read_b128 per 4 MFMAs, as in the Winograd kernels) + NDMA LDS-DMA
// instructions per wave of one of four kinds + s_waitcnt vmcnt(0) + barrier.  Prints cycles per iteration (wave 0).
//   kind 0: none   1: global_load_lds_dwordx4, lanes contiguous (1 KB)   2: buffer_load_dwordx4 ... lds, lanes contiguous
//   kind 3: buffer_load ... lds, lane PAIRS contiguous (32 B), pairs 256 B apart (a channels_last pixel row)
//   kind 4: as 3, the per-lane offsets read from LDS right before the instruction   5: as 3 with the two halves of a pair swapped
//   kind 6: contiguous with the halves of every pair swapped   7: two 512-byte runs + two stray pixels per instruction (blk32 input)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND, int NDMA>
__global__ __launch_bounds__(512, 2) void k(const float* __restrict__ src, float* out, unsigned long long* cyc, int iters, float a0) {
    __shared__ __attribute__((aligned(16))) float U[16384];        // 64 KB of operands
    __shared__ __attribute__((aligned(16))) float D[2 * 6144];     // DMA target, 2 x 24 KB
    __shared__ uint32_t Voff[512 * 3];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 16384; i += 512) U[i] = 0.001f * (i & 63);
    for (int i = tid; i < 1536; i += 512) Voff[i] = 0;
    __syncthreads();
    f32x4 acc[9][2];
#pragma unroll
    for (int i = 0; i < 9; ++i) { acc[i][0] = (f32x4){0, 0, 0, 0}; acc[i][1] = (f32x4){0, 0, 0, 0}; }
    const float* ub = U + lane * 4 + wave * 256;
    const uint32_t d_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)D;
    i32x4 rsrc;
    rsrc.x = (int)(uint32_t)(uint64_t)src; rsrc.y = (int)(uint32_t)((uint64_t)src >> 32); rsrc.z = 1 << 26; rsrc.w = 0x00020000;
    rsrc.x = __builtin_amdgcn_readfirstlane(rsrc.x); rsrc.y = __builtin_amdgcn_readfirstlane(rsrc.y);
    uint32_t vo[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int u = (wave * 3 + j) * 64 + lane;
        vo[j] = (KIND == 3 || KIND == 4) ? (uint32_t)((u >> 1) * 256 + (u & 1) * 16)
              : KIND == 5 ? (uint32_t)((u >> 1) * 256 + ((u & 1) ^ 1) * 16)                      // pairs 256 B apart, halves swapped
              : KIND == 6 ? (uint32_t)((u ^ 1) * 16)                                             // contiguous, halves of every pair swapped
              : KIND == 7 ? (uint32_t)((lane < 2 || (lane >= 34 && lane < 36)) ? (1 << 20) + u * 256 : u * 16)   // two 512-byte runs + two stray pixels
              : (uint32_t)(u * 16);
        Voff[j * 512 + tid] = vo[j];
    }
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if (KIND != 0) {
#pragma unroll
            for (int j = 0; j < NDMA; ++j) {
                const uint32_t m0v = __builtin_amdgcn_readfirstlane(d_lds + (uint32_t)((it & 1) * 24576 + ((wave * 3 + (j % 3)) * 1024)));
                uint32_t v = vo[j % 3];
                if (KIND == 4) v = Voff[(j % 3) * 512 + tid];
                if (KIND == 1)
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(v), "s"(src) : "memory", "m0");
                else
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(v), "s"(rsrc) : "memory", "m0");
            }
        }
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            const float4 b = *reinterpret_cast<const float4*>(ub + s * 2048 - (s >= 8 ? 2048 : 0));
            acc[s][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, a0, acc[s][0], 0, 0, 0);
            acc[s][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, a0, acc[s][1], 0, 0, 0);
            acc[s][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, a0, acc[s][0], 0, 0, 0);
            acc[s][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.w, a0, acc[s][1], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float sum = D[tid];
#pragma unroll
    for (int i = 0; i < 9; ++i) sum += acc[i][0][0] + acc[i][1][1];
    out[blockIdx.x * 512 + tid] = sum;
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

static float* g_src; static float* g_out; static unsigned long long* g_cyc;
template <int KIND, int NDMA> int run(const char* name) {
    const int iters = 400;
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL((k<KIND, NDMA>), dim3(256), dim3(512), 0, 0, g_src, g_out, g_cyc, iters, 1.0f); CK(hipDeviceSynchronize()); }
    unsigned long long h[8];
    CK(hipMemcpy(h, g_cyc + 8 * 100, sizeof(h), hipMemcpyDeviceToHost));
    printf("{\"kind\": \"%s\", \"dma_per_wave_and_iteration\": %d, \"cycles_per_iteration\": %.0f, \"mfma_bound\": %d}\n", name, NDMA, (double)h[0] / iters, 2 * 36 * 32);
    return 0;
}
int main() {
    CK(hipMalloc(&g_src, 64 << 20)); CK(hipMemset(g_src, 0, 64 << 20)); CK(hipMalloc(&g_out, 256 * 512 * 4)); CK(hipMalloc(&g_cyc, 256 * 8 * 8));
    run<0, 0>("none");
    run<1, 3>("global_load_lds contiguous"); run<1, 9>("global_load_lds contiguous");
    run<2, 3>("buffer_load lds contiguous"); run<2, 9>("buffer_load lds contiguous");
    run<3, 3>("buffer_load lds 32-byte pairs, 256 B apart"); run<3, 9>("buffer_load lds 32-byte pairs, 256 B apart");
    run<4, 3>("same, offsets read from LDS");
    run<5, 3>("32-byte pairs 256 B apart, halves swapped"); run<6, 3>("contiguous, halves of every pair swapped"); run<6, 9>("contiguous, halves of every pair swapped");
    run<7, 3>("two 512-byte runs + two stray pixels"); run<7, 9>("two 512-byte runs + two stray pixels");
    return 0;
}
