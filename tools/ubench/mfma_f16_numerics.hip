// What does v_mfma_f32_32x32x16_f16 do numerically?  (a) fp16 subnormal inputs: kept or flushed; (b) how the 16 products and the
// accumulator are summed: against the exactly rounded sum (float64, one rounding) and against a sequential fp32 fma chain;
// (c) the same with a large accumulator (alignment / truncation of small products).  Decides whether a split-fp16 convolution
// (x = hi + lo, three products, fp32 accumulate) can match an fp32 convolution.   hipcc --offload-arch=gfx950 -O2 -o build/ub/mfma_f16_numerics
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));

// A (32 x 16, row major), B (16 x 32, row major), C/D (32 x 32): one wave, one MFMA, `reps` chained MFMAs (K = 16 reps)
__global__ void k(const _Float16* A, const _Float16* B, const float* C, float* D, int reps) {
    const int l = threadIdx.x, r = l & 31, kq = l >> 5;
    f16v acc;
    for (int i = 0; i < 16; ++i) acc[i] = C[((i >> 2) * 8 + kq * 4 + (i & 3)) * 32 + r];
    for (int rep = 0; rep < reps; ++rep) {
        h8 a, b;
        for (int j = 0; j < 8; ++j) {
            a[j] = A[(rep * 32 + r) * 16 + kq * 8 + j];
            b[j] = B[(rep * 16 + kq * 8 + j) * 32 + r];
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    for (int i = 0; i < 16; ++i) D[((i >> 2) * 8 + kq * 4 + (i & 3)) * 32 + r] = acc[i];
}

static double urand() { return (rand() + 0.5) / (RAND_MAX + 1.0); }

static void run(const char* name, int reps, double amag, double bmag, double cmag, bool subnormal) {
    std::vector<_Float16> A(reps * 32 * 16), B(reps * 16 * 32);
    std::vector<float> C(32 * 32), D(32 * 32);
    for (auto& v : A) v = (_Float16)(subnormal ? ldexp(1.0 + (rand() % 7), -24) : (urand() * 2 - 1) * amag);
    for (auto& v : B) v = (_Float16)((urand() * 2 - 1) * bmag);
    for (auto& v : C) v = (float)((urand() * 2 - 1) * cmag);
    _Float16 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
    k<<<1, 64>>>(dA, dB, dC, dD, reps);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    double e_exact = 0, e_chain = 0, e_perrep = 0, nrm = 0; int eq_exact = 0, eq_chain = 0, eq_perrep = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double s = C[i * 32 + j]; float ch = C[i * 32 + j]; float pr = C[i * 32 + j];
        for (int rep = 0; rep < reps; ++rep) {
            double blk = 0;
            for (int kk = 0; kk < 16; ++kk) {
                const double p = (double)A[(rep * 32 + i) * 16 + kk] * (double)B[(rep * 16 + kk) * 32 + j];
                s += p; blk += p; ch = fmaf((float)A[(rep * 32 + i) * 16 + kk], (float)B[(rep * 16 + kk) * 32 + j], ch);
            }
            pr = (float)((double)pr + blk);          // one rounding per MFMA
        }
        const double d = D[i * 32 + j];
        e_exact += (d - s) * (d - s); e_chain += (d - ch) * (d - ch); e_perrep += (d - pr) * (d - pr); nrm += s * s;
        eq_exact += (float)s == D[i * 32 + j]; eq_chain += ch == D[i * 32 + j]; eq_perrep += pr == D[i * 32 + j];
    }
    printf("%-44s reps %3d: rel err vs exact %.3e | == exact-rounded %4d/1024 | == one-rounding-per-MFMA %4d | == fp32 fma chain %4d | D[0] %.9g\n",
           name, reps, sqrt(e_exact / nrm), eq_exact, eq_perrep, eq_chain, D[0]);
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
}

int main() {
    srand(1);
    run("subnormal A (2^-24 .. 7 2^-24), B ~ 1000", 1, 0, 1000.0, 0.0, true);
    run("A, B ~ U(-1,1), C = 0", 1, 1, 1, 0, false);
    run("A, B ~ U(-1,1), C ~ 1", 1, 1, 1, 1, false);
    run("A, B ~ U(-1,1), C ~ 1000 (small products)", 1, 1, 1, 1000, false);
    run("A ~ 1e-3 (lo piece), B ~ 1, C ~ 10", 1, 1e-3, 1, 10, false);
    run("K = 576 chain, C = 0", 36, 1, 1, 0, false);
    run("K = 1728 chain, C = 0", 108, 1, 1, 0, false);
    return 0;
}
