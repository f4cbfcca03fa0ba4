// Semantics check for the LDS-DMA form of a buffer load on gfx950:  buffer_load_dwordx4 voff, rsrc, soff offen lds
//   * LDS destination = M0 + 16 * lane (lane-linear), independent of the per-lane global offset
//   * a lane whose offset is outside the buffer (>= num_records) writes ZEROS to its LDS slot
//   * the scalar offset is added to the address but not to the range check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
#pragma clang diagnostic ignored "-Winline-asm"
__global__ void k(const float* x, float* out, int nbytes, int soff_bytes) {
    __shared__ __attribute__((aligned(16))) float buf[64 * 4 * 2];
    for (int i = threadIdx.x; i < 512; i += 64) buf[i] = -7.0f;               // stale marker
    __syncthreads();
    const uint64_t base = (uint64_t)x;
    i32x4 rsrc;
    rsrc.x = __builtin_amdgcn_readfirstlane((int)(uint32_t)base);
    rsrc.y = __builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
    rsrc.z = __builtin_amdgcn_readfirstlane(nbytes);
    rsrc.w = 0x00020000;
    const uint32_t lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)buf;
    const int lane = threadIdx.x;
    // lane l reads 16 B at element 4*(63-l) (reversed), every 4th lane is sent out of range
    const uint32_t voff = (lane & 3) == 3 ? 0x80000000u : (uint32_t)(63 - lane) * 16u;
    const uint32_t soff = __builtin_amdgcn_readfirstlane(soff_bytes);
    asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(lds + 1024)) : "m0");   // second KiB of buf
    asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff), "s"(rsrc), "s"(soff) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += 64) out[i] = buf[i];
}
int main() {
    const int n = 64 * 4 + 64;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *x, *out;
    hipMalloc(&x, n * 4); hipMalloc(&out, 512 * 4);
    hipMemcpy(x, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int soff = 0; soff <= 64; soff += 64) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, x, out, 64 * 16, soff);
        std::vector<float> o(512);
        hipMemcpy(o.data(), out, 512 * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) if (o[i] != -7.0f) ++bad;                 // first KiB untouched
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 4; ++e) {
                const float want = (l & 3) == 3 ? 0.0f : (float)(4 * (63 - l) + e + soff / 4);
                if (o[256 + 4 * l + e] != want) { if (bad < 5) printf("lane %d elem %d: got %g want %g\n", l, e, o[256 + 4 * l + e], want); ++bad; }
            }
        printf("{\"test\": \"buffer_load_dwordx4_lds\", \"soffset_bytes\": %d, \"mismatches\": %d}\n", soff, bad);
    }
    return 0;
}
