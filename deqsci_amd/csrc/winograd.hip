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
// One block = 4 x 8 Winograd tiles (8 x 16 output pixels) x all 64 output channels.  The 16 transform positions
// xi are 16 independent GEMMs  M[xi] (32 tiles x 64 cout) += V[xi] (32 x cin) U[xi] (cin x 64):
//   * 4 wavefronts, each owning 16 tiles x 32 couts of every M[xi]: 32 accumulators of v_mfma_f32_16x16x4_f32
//     (128 registers), two blocks per CU so that transform / barrier / epilogue time of one block is MFMA time
//     of the other (a first version with 64-tile blocks, 32x32x2 MFMAs and one block per CU spent 57 % of its
//     time outside the matrix pipe: tools/ubench/winograd_ablate.sh);
//   * cin is consumed in chunks of 8: V[xi][tile][8] and U[xi][cout][8] live in LDS with a 12-float row
//     stride (a 32-lane ds_read_b64 touches 64 distinct banks); lane quarter q = lane>>4 reads its cin
//     {2q, 2q+1} with ONE ds_read_b64 per operand and feeds 2 MFMAs (the K index of an MFMA is free to mean
//     "cin 2q+s" as long as A and B agree);
//   * the next chunk's 4x4 input patches (one channel per lane) and pre-transformed weights are fetched into
//     registers BEFORE the MFMA phase of the current chunk, so global latency hides under the MFMAs;
//   * the 16 values A^T M A needs for one (tile, cout) sit in the same accumulator slot of the 16 M[xi], so the
//     output transform, bias and ReLU are pure per-lane register work.
#include "common.hpp"

namespace deqsci {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WG_CK = 8;            // input channels per chunk
constexpr int WG_VS = 12;           // LDS row stride in floats (8 + 4 pad): ds_read_b64 of 32 lanes hits 64 distinct banks
constexpr int WG_NCHUNK = 64 / WG_CK;
constexpr int WG_T = 32;            // Winograd tiles per block: 4 rows x 8 columns = 8 x 16 output pixels
#ifndef WG_ABLATE        // tuning harness only (tools/ubench/winograd_ablate.sh): 1 = no MFMA, 2 = no global loads/transform
#define WG_ABLATE 0
#endif

// Block = 32 tiles x all 64 output channels, 4 wavefronts, TWO blocks resident per CU (72 KB LDS, <= 256 registers)
// so one block's input transform / barriers / epilogue run under the other block's MFMAs.  Wave w owns tiles
// [16*(w>>1), +16) x couts [32*(w&1), +32) for all 16 transform positions: 16 x 2 accumulators of
// v_mfma_f32_16x16x4_f32 (128 registers), and holds every value the output transform of its (tile, cout) needs.
__global__ __launch_bounds__(TB, 2) void winograd_conv64_kernel(const float* __restrict__ x, const float* __restrict__ Ug,
                                                                const float* __restrict__ bias, float* __restrict__ y,
                                                                int H, int W, int relu) {
    __shared__ __attribute__((aligned(16))) float Vs[16 * WG_T * WG_VS];     // V[xi][tile][cin]   24 KB
    __shared__ __attribute__((aligned(16))) float Us[16 * 64 * WG_VS];       // U[xi][cout][cin]   48 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave >> 1, wn = wave & 1;
    const int n = blockIdx.z;
    const int ty0 = blockIdx.y * 4, tx0 = blockIdx.x * 8;          // tile coordinates of the block
    const float* xn = x + (int64_t)n * H * W * 64;

    // transform role: tile t (0..31), input channel ci (0..7) of the chunk
    const int t = tid >> 3, ci = tid & 7;
    const int iy0 = 2 * (ty0 + (t >> 3)) - 1, ix0 = 2 * (tx0 + (t & 7)) - 1;
    // branch-free patch addressing: out-of-image pixels read a clamped (valid) address and are zeroed by a select
    int poff[16];
    unsigned pmask = 0;
#pragma unroll
    for (int pr = 0; pr < 4; ++pr)
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {
            const int iy = iy0 + pr, ix = ix0 + pc;
            const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int cy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), cx = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
            poff[pr * 4 + pc] = (cy * W + cx) * 64 + ci;
            pmask |= (ok ? 1u : 0u) << (pr * 4 + pc);
        }
    float d[16];
    float4 u[8];
    auto fetch = [&](int c) {
#pragma unroll
        for (int k = 0; k < 16; ++k) d[k] = xn[poff[k] + c * WG_CK];
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] = ld4(Ug + (int64_t)c * (16 * 64 * WG_CK) + (j * TB + tid) * 4);
    };

    f32x4 acc[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[xi][j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    // MFMA operand addresses: lane (i = lane&15, q = lane>>4) reads cin {2q, 2q+1}; k index of the MFMA = q
    const float* va = Vs + (16 * wt + (lane & 15)) * WG_VS + 2 * (lane >> 4);
    const float* ub = Us + (32 * wn + (lane & 15)) * WG_VS + 2 * (lane >> 4);

    fetch(0);
#pragma unroll 1
    for (int c = 0; c < WG_NCHUNK; ++c) {
#if WG_ABLATE == 2
        if (c > 0) goto mfma_phase;
#endif
        {
            // ---- input transform V = B^T d B; B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
#pragma unroll
            for (int k = 0; k < 16; ++k) if (!((pmask >> k) & 1u)) d[k] = 0.0f;
            float w[16];
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {                      // rows: w = B^T d
                const float d0 = d[pc], d1 = d[4 + pc], d2 = d[8 + pc], d3 = d[12 + pc];
                w[pc] = d0 - d2;
                w[4 + pc] = d1 + d2;
                w[8 + pc] = d2 - d1;
                w[12 + pc] = d1 - d3;
            }
            __syncthreads();                                      // previous chunk's MFMAs are done with LDS
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {                      // columns: V = w B
                const float w0 = w[pr * 4], w1 = w[pr * 4 + 1], w2 = w[pr * 4 + 2], w3 = w[pr * 4 + 3];
                float* vrow = Vs + ((pr * 4) * WG_T + t) * WG_VS + ci;
                vrow[0] = w0 - w2;
                vrow[WG_T * WG_VS] = w1 + w2;
                vrow[2 * WG_T * WG_VS] = w2 - w1;
                vrow[3 * WG_T * WG_VS] = w1 - w3;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {                         // weights: Ug[c][xi][cout][8] -> Us[xi][cout][12]
                const int e = (j * TB + tid) * 4;
                *reinterpret_cast<float4*>(Us + (e >> 3) * WG_VS + (e & 7)) = u[j];
            }
            __syncthreads();
            fetch(c + 1 < WG_NCHUNK ? c + 1 : c);                 // next chunk's global loads fly under the MFMA phase
        }
#if WG_ABLATE == 2
    mfma_phase:
#endif
        __builtin_amdgcn_sched_barrier(0);
        // MFMA phase: per xi three ds_read_b64 (A, B for the two cout tiles) feed four MFMAs; the reads of xi+1 are
        // issued before the MFMAs of xi, consecutive MFMAs alternate accumulators (40-cycle dependent latency)
        float2 a = *reinterpret_cast<const float2*>(va);
        float2 b0 = *reinterpret_cast<const float2*>(ub), b1 = *reinterpret_cast<const float2*>(ub + 16 * WG_VS);
        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            float2 an = a, b0n = b0, b1n = b1;
            if (xi + 1 < 16) {
                an = *reinterpret_cast<const float2*>(va + (xi + 1) * WG_T * WG_VS);
                b0n = *reinterpret_cast<const float2*>(ub + (xi + 1) * 64 * WG_VS);
                b1n = *reinterpret_cast<const float2*>(ub + (xi + 1) * 64 * WG_VS + 16 * WG_VS);
            }
#if WG_ABLATE == 1
            acc[xi][0][0] += a.x * b0.x + a.y * b0.y;
            acc[xi][1][0] += a.x * b1.x + a.y * b1.y;
#else
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0.x, acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b1.x, acc[xi][1], 0, 0, 0);
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b0.y, acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1.y, acc[xi][1], 0, 0, 0);
#endif
            a = an;
            b0 = b0n;
            b1 = b1n;
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);   // the three ds_read_b64 of xi+1 ...
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // ... then the four MFMAs of xi
        }
        __builtin_amdgcn_sched_barrier(0);                    // keep the next transform (and its vmcnt wait) BELOW the MFMAs
    }

    // ---- output transform Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]; bias; ReLU; store
    // C/D layout of the 16x16 MFMA: col = lane&15 (cout), row = 4*(lane>>4) + reg (tile)
    float* yn = y + (int64_t)n * H * W * 64;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int cout = 32 * wn + 16 * j + (lane & 15);
        const float bv = bias ? bias[cout] : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tl = 16 * wt + 4 * (lane >> 4) + r;
            const int oy = 2 * (ty0 + (tl >> 3)), ox = 2 * (tx0 + (tl & 7));
            float s0[4], s1[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float m0 = acc[b][j][r], m1 = acc[4 + b][j][r], m2 = acc[8 + b][j][r], m3 = acc[12 + b][j][r];
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
}

}  // namespace deqsci

#ifndef WG_NO_CABI
using namespace deqsci;

extern "C" int deqsci_conv3x3_c64_winograd_f32(const float* x, const float* u_packed, const float* bias, float* y, int64_t n,
                                               int64_t H, int64_t W, int relu, deqsci_stream_t stream) {
    if (!x || !u_packed || !y) return DEQSCI_ERR_NULL;
    if (n <= 0 || H <= 0 || W <= 0) return DEQSCI_ERR_SHAPE;
    if (n > 65535 || H > (1 << 20) || W > (1 << 20) || x == y) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(x) || !aligned16(u_packed) || !aligned16(y)) return DEQSCI_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)ceil_div(ceil_div(W, 2), 8), (unsigned)ceil_div(ceil_div(H, 2), 4), (unsigned)n);
    hipLaunchKernelGGL(winograd_conv64_kernel, grid, dim3(TB), 0, st, x, u_packed, bias, y, (int)H, (int)W, relu);
    return launch_status();
}
#endif  // WG_NO_CABI
