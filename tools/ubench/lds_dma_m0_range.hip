// Does the LDS-DMA destination (M0) reach beyond 64 KiB on gfx950 (160 KiB of LDS per CU)?  Writes 1 KiB by
// buffer_load_dwordx4 ... lds and by global_load_lds_dwordx4 to LDS byte offsets 1024, 70000-ish and 150000-ish.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma clang diagnostic ignored "-Winline-asm"
constexpr int LDS_FLOATS = 160 * 1024 / 4 - 64;
__global__ __launch_bounds__(64, 1) void k(const float* x, int* out, uint32_t off_bytes, int use_global) {
    extern __shared__ __attribute__((aligned(16))) float buf[];
    for (int i = threadIdx.x; i < LDS_FLOATS; i += 64) buf[i] = -7.0f;
    __syncthreads();
    const uint64_t base = (uint64_t)x;
    i32x4 rsrc;
    rsrc.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)base);
    rsrc.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
    rsrc.z = 1024;
    rsrc.w = 0x00020000;
    const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)buf;
    const uint32_t voff = threadIdx.x * 16u;
    const uint32_t m0v = __builtin_amdgcn_readfirstlane(lds + off_bytes);
    if (use_global) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(voff), "s"(base) : "memory", "m0");
    else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(m0v), "v"(voff), "s"(rsrc) : "memory", "m0");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // where did the 256 floats 0..255 land?  report the first LDS float index holding 0.0 followed by 1.0, and how many non-marker floats exist
    int first = -1, count = 0;
    if (threadIdx.x == 0) {
        for (int i = 0; i < LDS_FLOATS; ++i) {
            if (buf[i] != -7.0f) { ++count; if (first < 0) first = i; }
        }
        out[0] = first; out[1] = count; out[2] = (first >= 0) ? (int)buf[first + 5] : -1;
    }
}
int main() {
    std::vector<float> h(256);
    for (int i = 0; i < 256; ++i) h[i] = (float)i;
    float* x; int* out;
    hipMalloc(&x, 1024); hipMalloc(&out, 16);
    hipMemcpy(x, h.data(), 1024, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_FLOATS * 4);
    for (int g = 0; g < 2; ++g)
        for (uint32_t off : {1024u, 65536u + 4096u, 131072u + 16384u}) {
            hipLaunchKernelGGL(k, dim3(1), dim3(64), LDS_FLOATS * 4, 0, x, out, off, g);
            int o[3];
            hipMemcpy(o, out, 12, hipMemcpyDeviceToHost);
            printf("{\"form\": \"%s\", \"m0_offset_bytes\": %u, \"landed_at_byte\": %d, \"floats_written\": %d, \"sixth_value\": %d}\n",
                   g ? "global_load_lds_dwordx4" : "buffer_load_dwordx4_lds", off, o[0] * 4, o[1], o[2]);
        }
    return 0;
}
