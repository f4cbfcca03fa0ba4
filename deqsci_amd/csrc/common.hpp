// Shared device/host helpers for libdeqsci_hip (gfx950 only: wave64, no portability shims).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "deqsci_hip.h"

namespace deqsci {

constexpr int TB = 256;     // threads per block: 4 wavefronts, one per SIMD
constexpr int WAVE = 64;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// Streaming ("nt") forms for the big once-through tensors (z, Phi, history rows, activations): on MI355X a
// 2-read/1-write fp32 stream moves 5.6 TB/s with default-policy accesses and 6.7 TB/s with nt loads AND nt
// stores (tools/ubench/gap_variants.hip) - the lines are not kept in L2/MALL where nothing would re-read them.
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4s(const float* p) {
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4s(float* p, float4 v) {
    const v4f t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<v4f*>(p));
}

// Cache policy of a kernel's big streams, chosen by the launcher: bit 0 = nt loads, bit 1 = nt stores.
// Same-box sweep of every kernel on rotating buffer sets (tools/policy_sweep.sh, profiles/r01_policy_sweep.txt):
// nt loads AND nt stores win or tie everywhere (K3 77 -> 65 us = 5.7 -> 6.7 TB/s, K7+K3 196 -> 172 us), so every
// launch that touches at least STREAM_MIN_BYTES streams; smaller launches keep the default policy so a small
// working set stays in L2 / Infinity Cache for the next kernel.
constexpr int POL_DEFAULT = 0, POL_NTL = 1, POL_NTS = 2, POL_NTLS = 3;
constexpr int64_t STREAM_MIN_BYTES = 64ll << 20;
template <int POL> __device__ __forceinline__ float4 ldp(const float* p) { return (POL & 1) ? ld4s(p) : ld4(p); }
template <int POL> __device__ __forceinline__ void stp(float* p, float4 v) { if (POL & 2) st4s(p, v); else st4(p, v); }
// The shipped library reads NO environment variable and keeps no mutable state (include/deqsci_hip.h: "re-entrant, no global
// state").  The tuning / diagnostic knobs of the tools exist only in the -DDEQSCI_DIAG build (`make diag` ->
// build/diag/libdeqsci_hip_diag.so, loaded by the tools through DEQSCI_HIP_LIB).
#ifdef DEQSCI_DIAG
inline int diag_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
inline double diag_env_f64(const char* name, double dflt) { const char* e = getenv(name); return e ? atof(e) : dflt; }
inline int forced_policy() { return diag_env_int("DEQSCI_FORCE_POLICY", -1); }      // tools/policy_sweep.sh: 0..3
#else
inline int forced_policy() { return -1; }
#endif
inline int pick_policy(int64_t bytes, int streaming_policy) {
    const int f = forced_policy();
    if (f >= 0 && f <= 3) return f;
    return bytes >= STREAM_MIN_BYTES ? streaming_policy : POL_DEFAULT;
}

__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 operator/(float4 a, float4 b) { return make_float4(a.x / b.x, a.y / b.y, a.z / b.z, a.w / b.w); }
__device__ __forceinline__ float4 operator*(float a, float4 b) { return make_float4(a * b.x, a * b.y, a * b.z, a * b.w); }
__device__ __forceinline__ float4 fma4(float a, float4 b, float4 c) {
    return make_float4(fmaf(a, b.x, c.x), fmaf(a, b.y, c.y), fmaf(a, b.z, c.z), fmaf(a, b.w, c.w));
}
// ((a.x*b.x + a.y*b.y) + a.z*b.z) + a.w*b.w, products and sums rounded separately (no FMA) so the
// frame sum of K1/K3 matches a left-to-right fp32 sum of the reference's materialised x*Phi.
__device__ __forceinline__ float dot4_seq(float4 a, float4 b) { return ((a.x * b.x + a.y * b.y) + a.z * b.z) + a.w * b.w; }
__device__ __forceinline__ float dot4_fma(float4 a, float4 b, float acc) {
    return fmaf(a.w, b.w, fmaf(a.z, b.z, fmaf(a.y, b.y, fmaf(a.x, b.x, acc))));
}

// Sum over the LP adjacent lanes that hold the B = 4*LP frames of one pixel (HWB layout).
// xor-butterfly: every lane of the group ends with the bit-identical total.
template <int LP>
__device__ __forceinline__ float group_sum(float v) {
    if (LP >= 2) v += __shfl_xor(v, 1, WAVE);
    if (LP >= 4) v += __shfl_xor(v, 2, WAVE);
    if (LP >= 8) v += __shfl_xor(v, 4, WAVE);
    return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
    return v;   // valid in lane 0
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
    return v;
}

// ---- power-of-two scales of the "sp16" activations (csrc/conv_s16.hip: a tensor stored as the fp16 pieces hi + lo of 2^e x).
// fp32 is scale-free, fp16 is not, so the exponent e FOLLOWS THE DATA, per image of the batch: an activation's range is the pair (amax, exp)
// - `amax` a device pointer to one word per image, max |x| of that image of the activation as measured by the kernel that produced it
// (NULL: the fixed exponent `exp`), from which every kernel that writes or reads the image derives the same
// e = SP16_TARGET_EXP - floor(log2(amax[image])): 2^e max|x| lies in [2^11, 2^12),
// a factor 16 below fp16's overflow, and an element keeps all 22 bits of its split down to 2^-14 of the maximum (below that the lo
// piece goes subnormal: the ABSOLUTE error stays 2^-36 of the maximum).  Powers of two: every scaling is exact.
constexpr int SP16_DEFAULT_EXP = 8, SP16_TARGET_EXP = 11, SP16_EXP_LIMIT = 64;
__host__ __device__ inline int sp16_act_exp(float amax) {
    uint32_t b;
    __builtin_memcpy(&b, &amax, 4);
    const int e = (int)((b >> 23) & 0xffu);
    if (e == 0 || e == 255) return SP16_DEFAULT_EXP;          // all zeros (or subnormal) / not finite: any exponent does; an overflow shows downstream
    const int a = SP16_TARGET_EXP - (e - 127);
    return a < -SP16_EXP_LIMIT ? -SP16_EXP_LIMIT : a > SP16_EXP_LIMIT ? SP16_EXP_LIMIT : a;
}
__device__ __forceinline__ int sp16_resolve_exp(const float* amax, int exp) { return amax ? sp16_act_exp(*amax) : exp; }
// 2^e as a float; NaN beyond the normal range, so that an absurd combination of ranges is loud instead of silently mis-scaled
__device__ __forceinline__ float sp16_pow2(int e) {
    return (e < -126 || e > 127) ? __builtin_nanf("") : __builtin_bit_cast(float, (uint32_t)(e + 127) << 23);
}
// max over the 64 lanes of a per-lane value v >= 0, as the (wave-uniform) bit pattern of the float: four DPP steps inside each row of 16
// lanes, then the four row results through scalar registers - no LDS, no per-lane address registers
__device__ __forceinline__ uint32_t sp16_wave_max_bits(float v) {
    auto step = [](float x, auto ctrl) {
        return fmaxf(x, __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true)));
    };
    v = step(v, std::integral_constant<int, 0xB1>{});          // quad_perm [1,0,3,2]
    v = step(v, std::integral_constant<int, 0x4E>{});          // quad_perm [2,3,0,1]
    v = step(v, std::integral_constant<int, 0x141>{});         // row_half_mirror
    v = step(v, std::integral_constant<int, 0x140>{});         // row_mirror
    const uint32_t b = __builtin_bit_cast(uint32_t, v);
    const uint32_t r0 = __builtin_amdgcn_readlane(b, 0), r1 = __builtin_amdgcn_readlane(b, 16), r2 = __builtin_amdgcn_readlane(b, 32),
                   r3 = __builtin_amdgcn_readlane(b, 48);
    const uint32_t m01 = r0 > r1 ? r0 : r1, m23 = r2 > r3 ? r2 : r3;
    return m01 > m23 ? m01 : m23;
}
// max over the workgroup of a per-lane value v >= 0, times `unscale`, folded into *track by ONE atomic (non-negative floats order like
// their bit patterns).  `lds`: one word per wave.  Every thread of the workgroup must call it.
__device__ __forceinline__ void sp16_track_block_max(float v, float unscale, float* track, uint32_t* lds) {
    const uint32_t wm = sp16_wave_max_bits(v);
    const int nw = (int)((blockDim.x + WAVE - 1) / WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0) lds[threadIdx.x / WAVE] = wm;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t m = lds[0];
        for (int w = 1; w < nw; ++w) m = lds[w] > m ? lds[w] : m;
        atomicMax(reinterpret_cast<unsigned int*>(track), __builtin_bit_cast(uint32_t, __builtin_bit_cast(float, m) * unscale));
    }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Compute units of the current device (cached per device; a plain attribute query: no allocation, no synchronisation).
inline int num_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cached[dev] = v;
    }
    return cached[dev];
}

// Workgroups of `kernel` the device can hold AT ONCE (occupancy query x CUs; cached per device): what a launch whose workgroups wait for one
// another - the stack launches of conv_s16.hip / conv_w16.hip - may ask for at most.  0: the kernel cannot be resident as built (the caller
// reports DEQSCI_ERR_UNSUPPORTED instead of launching something that would sit in its waits until they time out).
template <typename Kernel> inline int64_t resident_workgroups(Kernel kernel, int threads, int* cache /* [64], zero-initialised */) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cache[dev] == 0) {
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess || per_cu <= 0) {
            (void)hipGetLastError();
            cache[dev] = -1;
        } else {
            cache[dev] = per_cu;
        }
    }
    return cache[dev] > 0 ? (int64_t)cache[dev] * num_cus() : 0;
}

inline int launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : static_cast<int>(e);
}

// history depth / partial layout shared by K4 and K5+K6
constexpr int MAXM = DEQSCI_MAX_M;
constexpr int PART_STRIDE = DEQSCI_PART_STRIDE;      // <G_k,G_j> j<MAXM, then |F_k|^2
constexpr int GRAM_STRIDE = 80;                      // doubles per sample: 64 Gram + ff + gg, padded to 640 B

}  // namespace deqsci
