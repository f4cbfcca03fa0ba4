// Denoiser epilogue for MI355X: h = max(h + bias[c], 0) in ONE in-place pass.
// PyTorch-ROCm runs conv2d (MIOpen Winograd), then a bias-add kernel, then a ReLU kernel: two extra
// read+write sweeps of a (bsz*B, 64, H/2, W/2) fp32 activation per layer (~9 % each of an FFDNet call on
// MI355X, profiles/r01_bench_kernel_stats.csv).  With BatchNorm folded into the conv weights
// (networks/ffdnet/models.py:53-58 in eval mode is an affine map per channel) the whole
// Conv-BN-ReLU block becomes conv + this kernel: HBM-bound, 8 bytes per element.
#include "common.hpp"

namespace deqsci {

// NCHW: one block row per (n,c) plane, float4 along H*W
template <int POL>
__global__ __launch_bounds__(TB) void bias_relu_nchw_kernel(float* __restrict__ h, const float* __restrict__ bias,
                                                            int64_t HW, int C, int blocks_per_plane, int relu) {
    const int64_t plane = blockIdx.x / blocks_per_plane;
    const int64_t i = ((int64_t)(blockIdx.x % blocks_per_plane) * TB + threadIdx.x) * 4;
    if (i >= HW) return;
    const float b = bias[plane % C];
    float* p = h + plane * HW + i;
    float4 v = ldp<POL>(p);
    v.x += b; v.y += b; v.z += b; v.w += b;
    if (relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
    stp<POL>(p, v);
}

// channels_last (N,H,W,C physical): float4 along C
template <int POL>
__global__ __launch_bounds__(TB) void bias_relu_nhwc_kernel(float* __restrict__ h, const float* __restrict__ bias,
                                                            int64_t total4, int C, int relu) {
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i >= total4) return;
    const float4 b = ld4(bias + (int)((i * 4) % C));
    float4 v = ldp<POL>(h + i * 4);
    v = v + b;
    if (relu) { v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); }
    stp<POL>(h + i * 4, v);
}

}  // namespace deqsci

using namespace deqsci;

extern "C" int deqsci_bias_relu_f32(float* h, const float* bias, int64_t n, int64_t c, int64_t hw, int channels_last,
                                    int relu, deqsci_stream_t stream) {
    if (!h || !bias) return DEQSCI_ERR_NULL;
    if (n <= 0 || c <= 0 || hw <= 0) return DEQSCI_ERR_SHAPE;
    if (!aligned16(h) || !aligned16(bias)) return DEQSCI_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // default policy unless forced: the activation is re-read by the next conv right away
    const int pol = forced_policy() >= 0 ? forced_policy() : POL_DEFAULT;
#define BR_DISPATCH(...)                                                   \
    switch (pol) {                                                         \
        case POL_NTL:  { constexpr int POL = POL_NTL; __VA_ARGS__; } break;  \
        case POL_NTS:  { constexpr int POL = POL_NTS; __VA_ARGS__; } break;  \
        case POL_NTLS: { constexpr int POL = POL_NTLS; __VA_ARGS__; } break; \
        default:       { constexpr int POL = POL_DEFAULT; __VA_ARGS__; } break; \
    }
    if (channels_last) {
        if (c % 4 != 0) return DEQSCI_ERR_UNSUPPORTED;
        const int64_t total4 = n * c * hw / 4;
        if (ceil_div(total4, TB) > 0x7fffffffLL) return DEQSCI_ERR_UNSUPPORTED;
        BR_DISPATCH(hipLaunchKernelGGL(bias_relu_nhwc_kernel<POL>, dim3((unsigned)ceil_div(total4, TB)), dim3(TB), 0, st, h, bias, total4, (int)c, relu));
    } else {
        if (hw % 4 != 0) return DEQSCI_ERR_UNSUPPORTED;
        const int64_t bpp = ceil_div(hw / 4, TB);
        if (bpp * n * c > 0x7fffffffLL) return DEQSCI_ERR_UNSUPPORTED;
        BR_DISPATCH(hipLaunchKernelGGL(bias_relu_nchw_kernel<POL>, dim3((unsigned)(bpp * n * c)), dim3(TB), 0, st, h, bias, hw, (int)c, (int)bpp, relu));
    }
    return launch_status();
}
