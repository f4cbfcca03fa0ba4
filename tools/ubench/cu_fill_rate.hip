// How many bytes per clock can ONE compute unit pull in from beyond its L1?  Every block streams the same 256 KB (L2
// resident, like the Winograd weights) or its own 1 MB slice (HBM / MALL), with 8 independent 16-byte loads in flight
// per lane; one block per CU (LDS-limited).
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NT, int NOCACHE>
__global__ __launch_bounds__(NT, 1) void k(const float4* __restrict__ src, float* out, int bytes_per_block, int shared_src, int reps) {
    __shared__ float pad[36 * 1024];                               // 144 KB: one block per CU
    const float4* p = src + (shared_src ? 0 : (size_t)blockIdx.x * (bytes_per_block / 16));
    const int n16 = bytes_per_block / 16;
    float4 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = make_float4(0, 0, 0, 0);
    for (int r = 0; r < reps; ++r) {
        for (int base = 0; base < n16; base += 8 * NT) {
            float4 v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const float4* q = p + base + i * NT + threadIdx.x;
                typedef float f4 __attribute__((ext_vector_type(4)));
                if (NOCACHE) { const f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(q)); v[i] = make_float4(t.x, t.y, t.z, t.w); }
                else v[i] = *q;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { a[i].x += v[i].x; a[i].y += v[i].y; a[i].z += v[i].z; a[i].w += v[i].w; }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y + a[i].z + a[i].w;
    if (s == 12345.678f) pad[threadIdx.x] = s;
    out[blockIdx.x * NT + threadIdx.x] = s + pad[0] * 0.0f;
}

template <int NT, int NOCACHE> int run(const char* label, const float4* src, float* out, int bytes, int shared_src, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NT, NOCACHE>), dim3(256), dim3(NT), 0, 0, src, out, bytes, shared_src, reps);
    CK(hipDeviceSynchronize());
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NT, NOCACHE>), dim3(256), dim3(NT), 0, 0, src, out, bytes, shared_src, reps);
    hipEventRecord(e1); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double per_cu = (double)bytes * reps;
    printf("%-58s %6.1f B/clk/CU  (%.2f TB/s all CUs)\n", label, per_cu / (ms * 1e-3 * 2.4e9), per_cu * 256 / (ms * 1e-3) / 1e12);
    return 0;
}
int main() {
    float4* src; float* out;
    CK(hipMalloc(&src, (size_t)256 << 20)); CK(hipMemset(src, 0, (size_t)256 << 20)); CK(hipMalloc(&out, 256 * 1024 * 4));
    run<256, 0>("shared 256 KB (L2), 256 threads", src, out, 256 << 10, 1, 200);
    run<512, 0>("shared 256 KB (L2), 512 threads", src, out, 256 << 10, 1, 200);
    run<1024, 0>("shared 256 KB (L2), 1024 threads", src, out, 256 << 10, 1, 200);
    run<512, 0>("shared 32 KB (L1-sized), 512 threads", src, out, 32 << 10, 1, 1600);
    run<512, 0>("shared 16 KB (fits L1), 512 threads", src, out, 16 << 10, 1, 3200);
    run<512, 0>("private 1 MB per CU (256 MB total: HBM), 512 threads", src, out, 1 << 20, 0, 20);
    run<512, 0>("private 128 KB per CU (32 MB total: L2/MALL), 512 threads", src, out, 128 << 10, 0, 200);
    run<512, 1>("shared 256 KB, nontemporal loads, 512 threads", src, out, 256 << 10, 1, 200);
    return 0;
}
