// Standalone tuning harness for the streaming kernels (not part of the library): K3 in the HWB float4
// mapping with the knobs that matter for an HBM-bound 2-read/1-write stream on MI355X.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/gap_variants tools/ubench/gap_variants.hip && /tmp/gap_variants
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float4 ldv(const float* p, int nt) {
    if (nt) { float4 v; v.x = __builtin_nontemporal_load(p); v.y = __builtin_nontemporal_load(p + 1); v.z = __builtin_nontemporal_load(p + 2); v.w = __builtin_nontemporal_load(p + 3); return v; }
    return *reinterpret_cast<const float4*>(p);
}
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f ld4(const float* p) { return *reinterpret_cast<const v4f*>(p); }
__device__ __forceinline__ v4f ld4nt(const float* p) { return __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)); }
__device__ __forceinline__ void st4(float* p, v4f v) { *reinterpret_cast<v4f*>(p) = v; }
__device__ __forceinline__ void st4nt(float* p, v4f v) { __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p)); }

template <int LP> __device__ __forceinline__ float gsum(float v) {
    if (LP >= 2) v += __shfl_xor(v, 1, 64);
    if (LP >= 4) v += __shfl_xor(v, 2, 64);
    return v;
}

// one "tile" = TBS*UNR float4; PERSIST: grid-stride over tiles
template <int TBS, int UNR, int NTL, int NTS, int PERSIST>
__global__ __launch_bounds__(TBS) void gap_hwb(const float* __restrict__ z, const float* __restrict__ phi, const float* __restrict__ y,
                                               const float* __restrict__ ps, float* __restrict__ z1, long Q /*float4 total*/) {
    constexpr int LP = 2;
    const long ntiles = (Q + (long)TBS * UNR - 1) / ((long)TBS * UNR);
    for (long tile = blockIdx.x; tile < ntiles; tile += PERSIST ? gridDim.x : ntiles) {
        const long base = tile * (TBS * UNR) + threadIdx.x;
        v4f zv[UNR], pv[UNR]; float yv[UNR], sv[UNR];
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            long q = base + j * TBS; if (q >= Q) q = Q - 1;
            zv[j] = NTL ? ld4nt(z + q * 4) : ld4(z + q * 4);
            pv[j] = NTL ? ld4nt(phi + q * 4) : ld4(phi + q * 4);
            yv[j] = y[q / LP]; sv[j] = ps[q / LP];
        }
#pragma unroll
        for (int j = 0; j < UNR; ++j) {
            const long q = base + j * TBS;
            const v4f pr = zv[j] * pv[j];
            const float fb = gsum<LP>(((pr.x + pr.y) + pr.z) + pr.w);
            const float r = (yv[j] - fb) / sv[j];
            const v4f o = zv[j] + r * pv[j];
            if (q < Q) { if (NTS) st4nt(z1 + q * 4, o); else st4(z1 + q * 4, o); }
        }
    }
}

template <int NT> __global__ __launch_bounds__(256) void copy_k(const float* __restrict__ a, float* __restrict__ b, long Q) {
    const long q = (long)blockIdx.x * 256 + threadIdx.x;
    if (q < Q) { v4f v = NT ? ld4nt(a + q * 4) : ld4(a + q * 4); if (NT) st4nt(b + q * 4, v); else st4(b + q * 4, v); }
}

struct Bufs { float *z, *phi, *y, *ps, *z1; };

template <typename F> float timeit(F f, int n) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) f(i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < n; ++i) f(i);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / n * 1e3f;
}

int main() {
    const long bsz = 64, P = 256 * 256, B = 8, N = bsz * P * B, Q = N / 4;
    const int NSETS = 2;                       // rotate two buffer sets (2 x 436 MB >> 256 MiB MALL)
    std::vector<Bufs> sets(NSETS);
    for (auto& s : sets) {
        CK(hipMalloc(&s.z, N * 4)); CK(hipMalloc(&s.phi, N * 4)); CK(hipMalloc(&s.z1, N * 4));
        CK(hipMalloc(&s.y, bsz * P * 4)); CK(hipMalloc(&s.ps, bsz * P * 4));
        CK(hipMemset(s.z, 0x3c, N * 4)); CK(hipMemset(s.phi, 0x3c, N * 4)); CK(hipMemset(s.y, 0x3c, bsz * P * 4)); CK(hipMemset(s.ps, 0x3c, bsz * P * 4));
    }
    const double bytes = (double)bsz * P * (12 * B + 8);
    auto report = [&](const char* name, float us, double by) { printf("%-44s %8.2f us  %7.1f GB/s  %.3f of 8 TB/s\n", name, us, by / us / 1e3, by / us / 1e3 / 8000.0); fflush(stdout); };
#define RUN(TBS, UNR, NTL, NTS, PERSIST, GRID, LABEL) { \
        const long ntiles = (Q + (long)TBS * UNR - 1) / ((long)TBS * UNR); \
        const unsigned grid = PERSIST ? (unsigned)(GRID) : (unsigned)ntiles; \
        float us = timeit([&](int i) { Bufs& s = sets[i % NSETS]; hipLaunchKernelGGL((gap_hwb<TBS, UNR, NTL, NTS, PERSIST>), dim3(grid), dim3(TBS), 0, 0, s.z, s.phi, s.y, s.ps, s.z1, Q); }, 40); \
        report(LABEL, us, bytes); }
    RUN(256, 4, 0, 0, 0, 0, "gap TB256 UNR4 (library)");
    RUN(256, 2, 0, 0, 0, 0, "gap TB256 UNR2");
    RUN(256, 1, 0, 0, 0, 0, "gap TB256 UNR1");
    RUN(256, 8, 0, 0, 0, 0, "gap TB256 UNR8");
    RUN(512, 4, 0, 0, 0, 0, "gap TB512 UNR4");
    RUN(128, 4, 0, 0, 0, 0, "gap TB128 UNR4");
    RUN(256, 4, 0, 1, 0, 0, "gap TB256 UNR4 nt-store");
    RUN(256, 4, 1, 1, 0, 0, "gap TB256 UNR4 nt-load nt-store");
    RUN(256, 4, 1, 0, 0, 0, "gap TB256 UNR4 nt-load");
    RUN(256, 2, 1, 1, 0, 0, "gap TB256 UNR2 nt-load nt-store");
    RUN(256, 4, 0, 0, 1, 2048, "gap TB256 UNR4 persistent 2048");
    RUN(256, 4, 0, 0, 1, 4096, "gap TB256 UNR4 persistent 4096");
    RUN(256, 4, 0, 0, 1, 1024, "gap TB256 UNR4 persistent 1024");
    RUN(256, 2, 0, 0, 1, 2048, "gap TB256 UNR2 persistent 2048");
    RUN(256, 4, 1, 1, 1, 2048, "gap TB256 UNR4 nt persistent 2048");
    RUN(256, 2, 1, 1, 1, 4096, "gap TB256 UNR2 nt persistent 4096");
    {
        float us = timeit([&](int i) { Bufs& s = sets[i % NSETS]; hipLaunchKernelGGL(copy_k<0>, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, 0, s.z, s.z1, Q); }, 40);
        report("copy float4", us, 2.0 * N * 4);
        us = timeit([&](int i) { Bufs& s = sets[i % NSETS]; hipLaunchKernelGGL(copy_k<1>, dim3((unsigned)((Q + 255) / 256)), dim3(256), 0, 0, s.z, s.z1, Q); }, 40);
        report("copy float4 nt", us, 2.0 * N * 4);
        us = timeit([&](int i) { Bufs& s = sets[i % NSETS]; CK(hipMemcpyAsync(s.z1, s.z, N * 4, hipMemcpyDeviceToDevice, 0)); }, 40);
        report("hipMemcpy D2D", us, 2.0 * N * 4);
    }
    return 0;
}
