// Tuning harness (not part of the library): HBM throughput of an NR-read / NW-write fp32 float4 stream on
// MI355X as a function of cache policy (default vs nt loads / nt stores) and per-lane unroll.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/stream_policy tools/ubench/stream_policy.hip && /tmp/stream_policy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
typedef float v4f __attribute__((ext_vector_type(4)));
struct Ptrs { const float* r[8]; float* w[2]; };

template <int NR, int NW, int UNR, int NTL, int NTS>
__global__ __launch_bounds__(256) void stream_k(Ptrs p, long Q) {
    const long base = (long)blockIdx.x * (256 * UNR) + threadIdx.x;
    v4f acc[UNR];
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
        long q = base + j * 256; if (q >= Q) q = Q - 1;
        v4f a = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const v4f* src = reinterpret_cast<const v4f*>(p.r[r]) + q;
            a += NTL ? __builtin_nontemporal_load(src) : *src;
        }
        acc[j] = a;
    }
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
        const long q = base + j * 256;
        if (q < Q) {
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                v4f* dst = reinterpret_cast<v4f*>(p.w[w]) + q;
                if (NTS) __builtin_nontemporal_store(acc[j] + (float)w, dst); else *dst = acc[j] + (float)w;
            }
            if (NW == 0 && acc[j].x == 123.456f) p.w[0][q] = 1.0f;
        }
    }
}

static float* bufs[2][10];
template <int NR, int NW, int UNR, int NTL, int NTS> void run(long Q, const char* label) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((Q + 256 * UNR - 1) / (256 * UNR));
    auto launch = [&](int i) {
        Ptrs p; for (int r = 0; r < 8; ++r) p.r[r] = bufs[i & 1][r]; p.w[0] = bufs[i & 1][8]; p.w[1] = bufs[i & 1][9];
        hipLaunchKernelGGL((stream_k<NR, NW, UNR, NTL, NTS>), dim3(grid), dim3(256), 0, 0, p, Q);
    };
    for (int i = 0; i < 3; ++i) launch(i);
    CK(hipDeviceSynchronize());
    const int n = 30;
    CK(hipEventRecord(e0)); for (int i = 0; i < n; ++i) launch(i); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms / n * 1e3, bytes = (double)(NR + NW) * Q * 16;
    printf("%-10s R%d W%d UNR%d ntl%d nts%d  %8.2f us %7.1f GB/s %.3f\n", label, NR, NW, UNR, NTL, NTS, us, bytes / us / 1e3, bytes / us / 1e3 / 8000);
    fflush(stdout);
}

#define POL(NR, NW, UNR, L) run<NR, NW, UNR, 0, 0>(Q, L); run<NR, NW, UNR, 1, 0>(Q, L); run<NR, NW, UNR, 0, 1>(Q, L); run<NR, NW, UNR, 1, 1>(Q, L);
int main() {
    const long N = 64L * 256 * 256 * 8, Q = N / 4;
    for (int s = 0; s < 2; ++s) for (int b = 0; b < 10; ++b) { CK(hipMalloc(&bufs[s][b], N * 4)); CK(hipMemset(bufs[s][b], 0x3c, N * 4)); }
    POL(2, 0, 4, "forward");
    POL(2, 0, 1, "forward");
    POL(1, 1, 4, "copy");
    POL(1, 1, 1, "copy");
    POL(2, 1, 4, "gap");
    POL(2, 1, 2, "gap");
    POL(6, 2, 1, "mix_gap");
    POL(6, 2, 2, "mix_gap");
    POL(7, 2, 1, "res_store");
    POL(7, 2, 2, "res_store");
    POL(2, 1, 1, "sub");
    return 0;
}
