// 3x3 convolution 64 -> 64 channels (pad 1, stride 1, fp32, channels_last) as Winograd F(2x2,3x3) on the
// fp32 matrix cores of MI355X, with the per-channel bias (folded BatchNorm) and ReLU fused into the epilogue.
//
// Why: this layer is 13 of FFDNet's 15 (networks/ffdnet/models.py:53-58) and 2 of SimpleCNN's 4, i.e. > 80 % of a
// DEQ-SCI reconstruction.  MIOpen runs it as a direct implicit GEMM at 85 % of the 157 TFLOP/s fp32 MFMA peak
// (579 us for 64 images of 128x128) preceded by a zero-fill (34 us) and followed by a bias+ReLU sweep (82 us); a
// direct fp32 convolution cannot get meaningfully faster than that, Winograd does 2.25x fewer multiplications.
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A      F(2x2,3x3): 4x4 input patch d (stride 2) -> 2x2 outputs
//
// One block = 8 x 8 Winograd tiles (16 x 16 output pixels) x all 64 output channels.  The 16 transform positions
// xi are 16 independent GEMMs  M[xi] (64 tiles x 64 cout) += V[xi] (64 x cin) U[xi] (cin x 64):
//   * 4 wavefronts, each owning a 32-tile x 32-cout quadrant of every M[xi]: 16 accumulators of
//     v_mfma_f32_32x32x2_f32 = 256 VGPRs per lane (1 wave per SIMD, the f32 MFMA reaches peak from that);
//   * cin is consumed in chunks of 8: V[xi][tile][8] and U[xi][cout][8] live in LDS with a 12-float row
//     stride (ds_read_b128 conflict-free); a lane half k = lane>>5 reads its 4 consecutive cin with ONE
//     ds_read_b128 per operand and feeds 4 MFMAs (the K index of an MFMA is free to mean "cin s" for k=0 and
//     "cin 4+s" for k=1 as long as A and B agree);
//   * the next chunk's 4x4 input patches (2 channels per lane) and pre-transformed weights are fetched into
//     registers BEFORE the MFMA phase of the current chunk, so global latency hides under 64 MFMAs per wave;
//     the input transform B^T d B (adds only) and the LDS writes are ~10 % of a chunk;
//   * the 16 values A^T M A needs for one (tile, cout) sit in the same accumulator slot of the 16 M[xi], so the
//     output transform, bias and ReLU are pure per-lane register work; a half-wave writes 128 B contiguous.
#include "common.hpp"

namespace deqsci {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int WG_CK = 8;            // input channels per chunk
constexpr int WG_VS = 12;           // LDS row stride in floats (8 + 4 pad)
constexpr int WG_NCHUNK = 64 / WG_CK;

__global__ __launch_bounds__(TB, 1) void winograd_conv64_kernel(const float* __restrict__ x, const float* __restrict__ Ug,
                                                                const float* __restrict__ bias, float* __restrict__ y,
                                                                int H, int W, int relu) {
    __shared__ __attribute__((aligned(16))) float Vs[16 * 64 * WG_VS];
    __shared__ __attribute__((aligned(16))) float Us[16 * 64 * WG_VS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int n = blockIdx.z;
    const int ty0 = blockIdx.y * 8, tx0 = blockIdx.x * 8;          // tile coordinates of the block
    const float* xn = x + (int64_t)n * H * W * 64;

    // transform role: tile t, channel pair cp of the chunk
    const int t = tid >> 2, cp = tid & 3;
    const int iy0 = 2 * (ty0 + (t >> 3)) - 1, ix0 = 2 * (tx0 + (t & 7)) - 1;

    float2 d[16];
    float4 u[8];
    auto fetch = [&](int c) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr)
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {
                const int iy = iy0 + pr, ix = ix0 + pc;
                float2 v = make_float2(0.0f, 0.0f);
                if (iy >= 0 && iy < H && ix >= 0 && ix < W)
                    v = *reinterpret_cast<const float2*>(xn + ((int64_t)iy * W + ix) * 64 + c * WG_CK + 2 * cp);
                d[pr * 4 + pc] = v;
            }
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] = ld4(Ug + (int64_t)c * (16 * 64 * WG_CK) + (j * TB + tid) * 4);
    };

    f32x16 acc[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[xi][r] = 0.0f;

    fetch(0);
#pragma unroll 1
    for (int c = 0; c < WG_NCHUNK; ++c) {
        // ---- input transform V = B^T d B for the 2 channels of this lane; B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
        float2 w[16];
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {                      // rows: w = B^T d
            const float2 d0 = d[pc], d1 = d[4 + pc], d2 = d[8 + pc], d3 = d[12 + pc];
            w[pc] = make_float2(d0.x - d2.x, d0.y - d2.y);
            w[4 + pc] = make_float2(d1.x + d2.x, d1.y + d2.y);
            w[8 + pc] = make_float2(d2.x - d1.x, d2.y - d1.y);
            w[12 + pc] = make_float2(d1.x - d3.x, d1.y - d3.y);
        }
        __syncthreads();                                      // previous chunk's MFMAs are done with LDS
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {                      // columns: V = w B
            const float2 w0 = w[pr * 4], w1 = w[pr * 4 + 1], w2 = w[pr * 4 + 2], w3 = w[pr * 4 + 3];
            float* vrow = Vs + ((pr * 4) * 64 + t) * WG_VS + 2 * cp;
            *reinterpret_cast<float2*>(vrow) = make_float2(w0.x - w2.x, w0.y - w2.y);
            *reinterpret_cast<float2*>(vrow + 64 * WG_VS) = make_float2(w1.x + w2.x, w1.y + w2.y);
            *reinterpret_cast<float2*>(vrow + 2 * 64 * WG_VS) = make_float2(w2.x - w1.x, w2.y - w1.y);
            *reinterpret_cast<float2*>(vrow + 3 * 64 * WG_VS) = make_float2(w1.x - w3.x, w1.y - w3.y);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {                         // weights: Ug[c][xi][cout][8] -> Us[xi][cout][12]
            const int e = (j * TB + tid) * 4;
            *reinterpret_cast<float4*>(Us + (e >> 3) * WG_VS + (e & 7)) = u[j];
        }
        __syncthreads();
        if (c + 1 < WG_NCHUNK) fetch(c + 1);                  // global loads fly under the MFMA phase
        const float* va = Vs + (32 * wm + (lane & 31)) * WG_VS + 4 * (lane >> 5);
        const float* ub = Us + (32 * wn + (lane & 31)) * WG_VS + 4 * (lane >> 5);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            const float4 a = *reinterpret_cast<const float4*>(va + xi * 64 * WG_VS);
            const float4 b = *reinterpret_cast<const float4*>(ub + xi * 64 * WG_VS);
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[xi], 0, 0, 0);
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[xi], 0, 0, 0);
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[xi], 0, 0, 0);
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[xi], 0, 0, 0);
        }
    }

    // ---- output transform Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]; bias; ReLU; store
    const int cout = 32 * wn + (lane & 31);
    const float bv = bias ? bias[cout] : 0.0f;
    float* yn = y + (int64_t)n * H * W * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int tl = 32 * wm + row;
        const int oy = 2 * (ty0 + (tl >> 3)), ox = 2 * (tx0 + (tl & 7));
        float s0[4], s1[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float m0 = acc[b][r], m1 = acc[4 + b][r], m2 = acc[8 + b][r], m3 = acc[12 + b][r];
            s0[b] = (m0 + m1) + m2;
            s1[b] = (m1 - m2) - m3;
        }
        float y00 = (s0[0] + s0[1]) + s0[2] + bv, y01 = (s0[1] - s0[2]) - s0[3] + bv;
        float y10 = (s1[0] + s1[1]) + s1[2] + bv, y11 = (s1[1] - s1[2]) - s1[3] + bv;
        if (relu) { y00 = fmaxf(y00, 0.0f); y01 = fmaxf(y01, 0.0f); y10 = fmaxf(y10, 0.0f); y11 = fmaxf(y11, 0.0f); }
        if (oy < H && ox < W) {
            float* o = yn + ((int64_t)oy * W + ox) * 64 + cout;
            o[0] = y00;
            if (ox + 1 < W) o[64] = y01;
            if (oy + 1 < H) {
                o[(int64_t)W * 64] = y10;
                if (ox + 1 < W) o[(int64_t)W * 64 + 64] = y11;
            }
        }
    }
}

}  // namespace deqsci

using namespace deqsci;

extern "C" int deqsci_conv3x3_c64_winograd_f32(const float* x, const float* u_packed, const float* bias, float* y, int64_t n,
                                               int64_t H, int64_t W, int relu, deqsci_stream_t stream) {
    if (!x || !u_packed || !y) return DEQSCI_ERR_NULL;
    if (n <= 0 || H <= 0 || W <= 0) return DEQSCI_ERR_SHAPE;
    if (n > 65535 || H > (1 << 20) || W > (1 << 20) || x == y) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(x) || !aligned16(u_packed) || !aligned16(y)) return DEQSCI_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)ceil_div(ceil_div(W, 2), 8), (unsigned)ceil_div(ceil_div(H, 2), 8), (unsigned)n);
    hipLaunchKernelGGL(winograd_conv64_kernel, grid, dim3(TB), 0, st, x, u_packed, bias, y, (int)H, (int)W, relu);
    return launch_status();
}
