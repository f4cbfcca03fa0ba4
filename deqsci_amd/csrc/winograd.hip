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
#include <hip/hip_ext.h>

namespace deqsci {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int WG_CK = 8;            // input channels per chunk

constexpr int WG_NCHUNK = 64 / WG_CK;

#ifndef WG_ABLATE        // tuning harness only (tools/ubench/winograd_ablate.sh): 1 = no MFMA, 2 = no global loads/transform
#define WG_ABLATE 0
#endif

// Block = 32 tiles x all 64 output channels, 4 wavefronts, TWO blocks resident per CU (<= 80 KB LDS, <= 256 registers)
// so one block's staging / barriers / epilogue run under the other block's MFMAs.  Wave w owns tiles
// [16*(w>>1), +16) x couts [32*(w&1), +32) for all 16 transform positions: 16 x 2 accumulators of
// v_mfma_f32_16x16x4_f32 (128 registers), and holds every value the output transform of its (tile, cout) needs.
//
// Data movement (the first versions were bound by LDS traffic and barriers, not by the matrix pipe):
//   * the RAW input tile (10 x 18 pixels) is staged in LDS one 16-channel quarter at a time (64 B per pixel,
//     coalesced); pixel stride 18 floats, and pixel rows 2,3,6,7 shifted right by one pixel, make the per-lane
//     patch reads (ds_read2_b64, 32 banks) conflict-free;
//   * each MFMA lane (tile i = lane&15, channel pair q = lane>>4) reads ITS OWN 4x4 patch for its 2 channels of
//     the chunk and computes V = B^T d B in registers: the A operands of all 16 xi never touch LDS again;
//   * the pre-transformed weights of a chunk go through LDS in MFMA-lane order (host-packed): staging is a linear
//     32 KB copy, a lane's B operands for one xi are ONE conflict-free ds_read_b128, and the chunk buffers are
//     double-buffered so a chunk costs one barrier.
constexpr int WG_RAW_PS = 18;                 // floats per staged pixel (16 channels + 2)
constexpr int WG_RAW_COLS = 18, WG_RAW_ROWS = 10;
constexpr int WG_RAW_RS = (WG_RAW_COLS + 1) * WG_RAW_PS;      // 342 floats per staged pixel row (one spare pixel for the shift)
constexpr int WG_U_CHUNK = 16 * 2 * 64 * 4;   // floats of one weight chunk in LDS (32 KB)

__device__ __forceinline__ int wg_row_shift(int r) { return (r >> 1) & 1; }

__global__ __launch_bounds__(TB, 2) void winograd_conv64_kernel(const float* __restrict__ x, const float* __restrict__ Ug,
                                                                const float* __restrict__ bias, float* __restrict__ y,
                                                                int H, int W, int relu) {
    __shared__ __attribute__((aligned(16))) float Raw[WG_RAW_ROWS * WG_RAW_RS];   // 13.4 KB
    __shared__ __attribute__((aligned(16))) float Us[2 * WG_U_CHUNK];             // 2 x U[xi][cout half][MFMA lane][j][2]   64 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave >> 1, wn = wave & 1;
    const int n = blockIdx.z;
    const int ty0 = blockIdx.y * 4, tx0 = blockIdx.x * 8;          // tile coordinates of the block
    const float* xn = x + (int64_t)n * H * W * 64;
    const int py0 = 2 * ty0 - 1, px0 = 2 * tx0 - 1;                // image coordinates of staged pixel (0,0)

    // ---- staging roles: raw quarter = 180 pixels x 4 float4 (3 per thread, last partly idle); U chunk = 8 float4 per thread
    constexpr int RAW_F4 = WG_RAW_ROWS * WG_RAW_COLS * 4;          // 720
    constexpr int RAW_PER_THREAD = (RAW_F4 + TB - 1) / TB;         // 3
    float4 rawv[RAW_PER_THREAD];
    auto fetch_raw = [&](int quarter) {
#pragma unroll
        for (int k = 0; k < RAW_PER_THREAD; ++k) {
            const int e = k * TB + tid;
            const int pix = e >> 2, q4 = e & 3;
            const int pr = pix / WG_RAW_COLS, pc = pix - pr * WG_RAW_COLS;
            const int iy = py0 + pr, ix = px0 + pc;
            const bool ok = e < RAW_F4 && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int cy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), cx = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
            float4 v = ld4(xn + ((int64_t)cy * W + cx) * 64 + quarter * 16 + 4 * q4);   // clamped address, zero-select
            if (!ok) v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            rawv[k] = v;
        }
    };
    auto store_raw = [&]() {
#pragma unroll
        for (int k = 0; k < RAW_PER_THREAD; ++k) {
            const int e = k * TB + tid;
            if (e < RAW_F4) {
                const int pix = e >> 2, q4 = e & 3;
                const int pr = pix / WG_RAW_COLS, pc = pix - pr * WG_RAW_COLS;
                float* dst = Raw + pr * WG_RAW_RS + (pc + wg_row_shift(pr)) * WG_RAW_PS + 4 * q4;   // 8-B aligned
                *reinterpret_cast<float2*>(dst) = make_float2(rawv[k].x, rawv[k].y);
                *reinterpret_cast<float2*>(dst + 2) = make_float2(rawv[k].z, rawv[k].w);
            }
        }
    };
    float4 u[8];
    auto fetch_u = [&](int c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) u[j] = ld4(Ug + (int64_t)c * WG_U_CHUNK + (j * TB + tid) * 4);
    };
    auto store_u = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 8; ++j) *reinterpret_cast<float4*>(Us + buf * WG_U_CHUNK + (j * TB + tid) * 4) = u[j];
    };

    f32x4 acc[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[xi][j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    // MFMA roles: lane (i = lane&15, q = lane>>4) owns tile 16*wt + i and channels {2q, 2q+1} of the chunk
    const int mi = lane & 15, mq = lane >> 4;
    const int tl_a = 16 * wt + mi;
    int prow[4];                                             // this lane's 4 patch rows (with their shift) in Raw
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
        const int r = 2 * (tl_a >> 3) + pr;
        prow[pr] = r * WG_RAW_RS + (2 * (tl_a & 7) + wg_row_shift(r)) * WG_RAW_PS + 2 * mq;
    }
    const float* ub = Us + (wn * 64 + lane) * 4;             // this lane's B operands of xi = 0 in buffer 0

    fetch_raw(0);
    fetch_u(0);
    store_raw();
    store_u(0);
    __syncthreads();
#pragma unroll 1
    for (int c = 0; c < WG_NCHUNK; ++c) {
        // entry: Us[c&1] = U(c) and Raw = quarter c>>1 are visible to every wave
#if WG_ABLATE < 3
        if (c + 1 < WG_NCHUNK) fetch_u(c + 1);                // next chunk's / quarter's global loads fly under the MFMAs
        if ((c & 1) == 0 && c + 2 < WG_NCHUNK) fetch_raw((c >> 1) + 1);
#endif

        // ---- input transform in registers: V = B^T d B for (tile, 2 channels); B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
        float2 v[16];
        {
            float2 w[16];
            const float* pp = Raw + 8 * (c & 1);
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) {                  // rows: w = B^T d (column pc of the patch)
                const float2 d0 = *reinterpret_cast<const float2*>(pp + prow[0] + pc * WG_RAW_PS);
                const float2 d1 = *reinterpret_cast<const float2*>(pp + prow[1] + pc * WG_RAW_PS);
                const float2 d2 = *reinterpret_cast<const float2*>(pp + prow[2] + pc * WG_RAW_PS);
                const float2 d3 = *reinterpret_cast<const float2*>(pp + prow[3] + pc * WG_RAW_PS);
                w[pc] = make_float2(d0.x - d2.x, d0.y - d2.y);
                w[4 + pc] = make_float2(d1.x + d2.x, d1.y + d2.y);
                w[8 + pc] = make_float2(d2.x - d1.x, d2.y - d1.y);
                w[12 + pc] = make_float2(d1.x - d3.x, d1.y - d3.y);
            }
#pragma unroll
            for (int pr = 0; pr < 4; ++pr) {                  // columns: V = w B
                const float2 w0 = w[pr * 4], w1 = w[pr * 4 + 1], w2 = w[pr * 4 + 2], w3 = w[pr * 4 + 3];
                v[pr * 4] = make_float2(w0.x - w2.x, w0.y - w2.y);
                v[pr * 4 + 1] = make_float2(w1.x + w2.x, w1.y + w2.y);
                v[pr * 4 + 2] = make_float2(w2.x - w1.x, w2.y - w1.y);
                v[pr * 4 + 3] = make_float2(w1.x - w3.x, w1.y - w3.y);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- MFMA phase: per xi ONE ds_read_b128 (this lane's B operands for both cout tiles and both k-steps, stored
        // in lane order so a wavefront reads 1 KiB contiguous) feeds four MFMAs; the read of xi+1 is issued before the
        // MFMAs of xi; consecutive MFMAs alternate accumulators (40-cycle dependent latency)
        const float* ubc = ub + (c & 1) * WG_U_CHUNK;
        float4 b = *reinterpret_cast<const float4*>(ubc);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            float4 bn = b;
#if WG_ABLATE != 4
            if (xi + 1 < 16) bn = *reinterpret_cast<const float4*>(ubc + (xi + 1) * (2 * 64 * 4));
#endif
#if WG_ABLATE == 1
            acc[xi][0][0] += v[xi].x * b.x + v[xi].y * b.y;
            acc[xi][1][0] += v[xi].x * b.z + v[xi].y * b.w;
#else
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[xi].x, b.x, acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[xi].x, b.z, acc[xi][1], 0, 0, 0);
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[xi].y, b.y, acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[xi].y, b.w, acc[xi][1], 0, 0, 0);
#endif
            b = bn;
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // the ds_read_b128 of xi+1 ...
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // ... then the four MFMAs of xi
        }
        __builtin_amdgcn_sched_barrier(0);
#if WG_ABLATE < 3
        if (c + 1 < WG_NCHUNK) store_u((c + 1) & 1);          // the other buffer: nobody reads it before the barrier
        __syncthreads();                                      // U(c+1) visible; every wave is done with Us[c&1] (and Raw if c is odd)
        if ((c & 1) == 1 && c + 1 < WG_NCHUNK) {              // quarter boundary: one extra barrier per two chunks
            store_raw();
            __syncthreads();
        }
#endif
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

static int winograd_impl(const float* x, const float* u_packed, const float* bias, float* y, int64_t n, int64_t H, int64_t W,
                         int relu, deqsci_stream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    if (!x || !u_packed || !y) return DEQSCI_ERR_NULL;
    if (n <= 0 || H <= 0 || W <= 0) return DEQSCI_ERR_SHAPE;
    if (n > 65535 || H > (1 << 20) || W > (1 << 20) || x == y) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(x) || !aligned16(u_packed) || !aligned16(y)) return DEQSCI_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)ceil_div(ceil_div(W, 2), 8), (unsigned)ceil_div(ceil_div(H, 2), 4), (unsigned)n);
    hipExtLaunchKernelGGL(winograd_conv64_kernel, grid, dim3(TB), 0, st, ev0, ev1, 0, x, u_packed, bias, y, (int)H, (int)W, relu);
    return launch_status();
}

extern "C" int deqsci_conv3x3_c64_winograd_f32(const float* x, const float* u_packed, const float* bias, float* y, int64_t n,
                                               int64_t H, int64_t W, int relu, deqsci_stream_t stream) {
    return winograd_impl(x, u_packed, bias, y, n, H, W, relu, stream, nullptr, nullptr);
}

extern "C" int deqsci_conv3x3_c64_winograd_timed_f32(const float* x, const float* u_packed, const float* bias, float* y, int64_t n,
                                                     int64_t H, int64_t W, int relu, deqsci_stream_t stream, void* start_event,
                                                     void* stop_event) {
    if (!start_event || !stop_event) return DEQSCI_ERR_NULL;
    return winograd_impl(x, u_packed, bias, y, n, H, W, relu, stream, static_cast<hipEvent_t>(start_event),
                         static_cast<hipEvent_t>(stop_event));
}
#endif  // WG_NO_CABI
