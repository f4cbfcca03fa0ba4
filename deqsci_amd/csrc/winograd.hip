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

// Block = 64 tiles (8 x 8) x all 64 output channels, 8 wavefronts (512 threads), ONE block per CU at 2 waves per SIMD:
// per-CU weight traffic is what bounds this kernel (the global -> LDS path sustains ~25 GB/s per CU), and a 64-tile block
// streams each 32 KB weight chunk once for twice the outputs of the earlier 32-tile / two-blocks-per-CU version.
// Wave w owns tiles [16*(w>>1), +16) x couts [32*(w&1), +32) for all 16 transform positions: 16 x 2 accumulators of
// v_mfma_f32_16x16x4_f32 (128 registers), and holds every value the output transform of its (tile, cout) needs.
//
// The 64 input channels are consumed in 8 chunks of 8.  A chunk is one software-pipeline stage with ONE barrier:
//   * weights U(c+1): host-packed in MFMA-lane order, copied global -> LDS by the DMA path (global_load_lds_dwordx4,
//     no registers, no ds_write) into the other of two 32 KB buffers; a lane's B operands for one xi are then ONE
//     conflict-free ds_read_b128 feeding 4 MFMAs;
//   * raw input of chunk c+2 (10 x 18 pixels x 8 channels): two coalesced float4 per lane into registers at the top of
//     the stage, written to the other of two 7.6 KB LDS tiles after the MFMAs (pixel stride 10 floats, pixel rows
//     2,3,6,7 shifted by one pixel: the per-lane patch reads hit 32 distinct banks);
//   * each MFMA lane (tile i = lane&15, channel pair q = lane>>4) reads ITS OWN 4x4 patch of chunk c+1 while the
//     MFMAs of chunk c run, and turns it into V = B^T d B (its A operands for all 16 xi) in registers afterwards:
//     the A operands never go back to LDS, and no LDS read latency is exposed (measured: the un-pipelined input
//     transform was 24 % of a block's lifetime).
constexpr int WG_RAW_PS = 10;                 // floats per staged pixel (8 channels + 2)
constexpr int WG_THREADS = 512;               // 8 wavefronts: wave w owns tiles [16*(w>>1), +16) x couts [32*(w&1), +32)
constexpr int WG_RAW_COLS = 18, WG_RAW_ROWS = 18;  // 8 x 8 tiles = 16 x 16 output pixels (+ halo)
constexpr int WG_RAW_RS = (WG_RAW_COLS + 1) * WG_RAW_PS;      // 190 floats per staged pixel row (one spare pixel for the shift)
constexpr int WG_RAW_BUF = WG_RAW_ROWS * WG_RAW_RS;           // 1900 floats = 7.6 KB
constexpr int WG_U_CHUNK = 16 * 2 * 64 * 4;   // floats of one weight chunk in LDS (32 KB)

__device__ __forceinline__ int wg_row_shift(int r) { return (r >> 1) & 1; }

__global__ __launch_bounds__(WG_THREADS, 2) void winograd_conv64_kernel(const float* __restrict__ x, const float* __restrict__ Ug,
                                                                const float* __restrict__ bias, float* __restrict__ y,
                                                                int H, int W, int relu) {
    __shared__ __attribute__((aligned(16))) float Us[2 * WG_U_CHUNK];             // 2 x U[xi][cout half][MFMA lane][j][2]   64 KB
    __shared__ __attribute__((aligned(16))) float Raw[2 * WG_RAW_BUF];            // 2 x raw chunk tile                    15.2 KB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wt = wave >> 1, wn = wave & 1;
    const int n = blockIdx.z;
    const int ty0 = blockIdx.y * 8, tx0 = blockIdx.x * 8;          // tile coordinates of the block
    const float* xn = x + (int64_t)n * H * W * 64;
    const int py0 = 2 * ty0 - 1, px0 = 2 * tx0 - 1;                // image coordinates of staged pixel (0,0)

    // ---- raw staging role: chunk tile = 324 pixels x 2 float4; lane e handles (pixel e>>1, half e&1), e = tid, tid + 512
    constexpr int RAW_F4 = WG_RAW_ROWS * WG_RAW_COLS * 2;          // 648
    int roff[2], rdst[2];                                          // global element offset (clamped) / LDS float offset, -1 = idle
    bool rok[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = k * WG_THREADS + tid;
        const int pix = e >> 1, q4 = e & 1;
        const int pr = pix / WG_RAW_COLS, pc = pix - pr * WG_RAW_COLS;
        const int iy = py0 + pr, ix = px0 + pc;
        rok[k] = e < RAW_F4 && iy >= 0 && iy < H && ix >= 0 && ix < W;
        const int cy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), cx = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
        roff[k] = (cy * W + cx) * 64 + 4 * q4;
        rdst[k] = e < RAW_F4 ? pr * WG_RAW_RS + (pc + wg_row_shift(pr)) * WG_RAW_PS + 4 * q4 : -1;
    }
    float4 rawv[2];
    auto fetch_raw = [&](int c) {                                  // clamped address + zero-select: no branches
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float4 v = ld4(xn + roff[k] + c * WG_CK);
            if (!rok[k]) v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            rawv[k] = v;
        }
    };
    auto store_raw = [&](int buf) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if (rdst[k] >= 0) {
                float* dst = Raw + buf * WG_RAW_BUF + rdst[k];                          // 8-B aligned (pixel stride 40 B)
                *reinterpret_cast<float2*>(dst) = make_float2(rawv[k].x, rawv[k].y);
                *reinterpret_cast<float2*>(dst + 2) = make_float2(rawv[k].z, rawv[k].w);
            }
    };
    // ---- weight chunk: DMA global -> LDS, 8 x 1 KiB per wavefront, linear
    auto dma_u_piece = [&](int c, int buf, int j) {
        const int blk = j * 8 + wave;                                                   // 1 KiB block of the 32 KB chunk
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)(Ug + (int64_t)c * WG_U_CHUNK + (blk * 64 + lane) * 4),
            (__attribute__((address_space(3))) void*)(Us + buf * WG_U_CHUNK + blk * 256), 16, 0, 0);
    };
    auto dma_u = [&](int c, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) dma_u_piece(c, buf, j);
    };

    f32x4 acc[16][2];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[xi][j] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};

    // MFMA roles: lane (i = lane&15, q = lane>>4) owns tile 16*wt + i and channels {2q, 2q+1} of the chunk
    const int mi = lane & 15, mq = lane >> 4;
    const int tl_a = 16 * wt + mi;
    int prow[4];                                             // this lane's 4 patch rows (with their shift) in a raw tile
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
        const int r = 2 * (tl_a >> 3) + pr;
        prow[pr] = r * WG_RAW_RS + (2 * (tl_a & 7) + wg_row_shift(r)) * WG_RAW_PS + 2 * mq;
    }
    const float* ub = Us + (wn * 64 + lane) * 4;             // this lane's B operands of xi = 0 in buffer 0

    // B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]:  V = B^T d B
    float2 dn[16], v[16];
    auto read_patch = [&](int buf) {
        const float* pp = Raw + buf * WG_RAW_BUF;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr)
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) dn[pr * 4 + pc] = *reinterpret_cast<const float2*>(pp + prow[pr] + pc * WG_RAW_PS);
    };
    auto transform = [&]() {
        float2 w[16];
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {                      // rows: w = B^T d (column pc of the patch)
            const float2 d0 = dn[pc], d1 = dn[4 + pc], d2 = dn[8 + pc], d3 = dn[12 + pc];
            w[pc] = make_float2(d0.x - d2.x, d0.y - d2.y);
            w[4 + pc] = make_float2(d1.x + d2.x, d1.y + d2.y);
            w[8 + pc] = make_float2(d2.x - d1.x, d2.y - d1.y);
            w[12 + pc] = make_float2(d1.x - d3.x, d1.y - d3.y);
        }
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {                      // columns: V = w B
            const float2 w0 = w[pr * 4], w1 = w[pr * 4 + 1], w2 = w[pr * 4 + 2], w3 = w[pr * 4 + 3];
            v[pr * 4] = make_float2(w0.x - w2.x, w0.y - w2.y);
            v[pr * 4 + 1] = make_float2(w1.x + w2.x, w1.y + w2.y);
            v[pr * 4 + 2] = make_float2(w2.x - w1.x, w2.y - w1.y);
            v[pr * 4 + 3] = make_float2(w1.x - w3.x, w1.y - w3.y);
        }
    };

#ifdef WG_STAMP
    unsigned long long* stamp_out = reinterpret_cast<unsigned long long*>(const_cast<float*>(bias)) +
                                    (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 40;
    int stamp_i = 0;
#define WG_MARK() do { if (tid == 0) stamp_out[stamp_i] = __builtin_readcyclecounter(); ++stamp_i; } while (0)
#else
#define WG_MARK() do { } while (0)
#endif
    WG_MARK();                                                // 0: kernel entry
    // ---- prologue: U(0), raw(0), raw(1) staged; V(0) computed
    dma_u(0, 0);
    fetch_raw(0);
    store_raw(0);
    fetch_raw(1);
    store_raw(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_patch(0);
    transform();
    WG_MARK();                                                // 1: prologue done
#pragma unroll 1
    for (int c = 0; c < WG_NCHUNK; ++c) {
        // entry: Us[c&1] = U(c), Raw[(c+1)&1] = raw(c+1) visible; v = V(c) in registers
        __builtin_amdgcn_sched_barrier(0);
        WG_MARK();
        // ---- MFMA phase: per xi ONE ds_read_b128 (B operands) and ONE ds_read_b64 (a patch element of the next chunk)
        // are issued ahead of the four MFMAs of xi; consecutive MFMAs alternate accumulators (40-cycle dependent latency)
        const float* ubc = ub + (c & 1) * WG_U_CHUNK;
        const float* ppn = Raw + ((c + 1) & 1) * WG_RAW_BUF;
        float4 b = *reinterpret_cast<const float4*>(ubc);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            float4 bn = b;
#if WG_ABLATE != 4
            if (xi + 1 < 16) bn = *reinterpret_cast<const float4*>(ubc + (xi + 1) * (2 * 64 * 4));
#endif
            dn[xi] = *reinterpret_cast<const float2*>(ppn + prow[xi >> 2] + (xi & 3) * WG_RAW_PS);
#if WG_ABLATE < 3
            // the next chunk's weights (8 DMA pieces) and the raw tile after next (2 loads) are issued one per xi, so the
            // vector-memory queue never makes the wave wait in front of its MFMAs
            if (xi < 8 && (xi & 1) == 0) { if (c + 1 < WG_NCHUNK) dma_u_piece(c + 1, (c + 1) & 1, xi >> 1); }
            else if (xi == 8) { if (c + 2 < WG_NCHUNK) fetch_raw(c + 2); }
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
            if (xi < 8 && (xi & 1) == 0) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // one DMA piece ...
            else if (xi == 8) __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);              // ... or the two raw loads
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // the B read of xi+1 and one patch read ...
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // ... then the four MFMAs of xi
        }
        __builtin_amdgcn_sched_barrier(0);
        WG_MARK();
#if WG_ABLATE < 3
        if (c + 1 < WG_NCHUNK) transform();                   // V(c+1) from the patch read during the MFMAs
        if (c + 2 < WG_NCHUNK) store_raw(c & 1);              // raw(c+2) into the tile V(c) came from
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's part of U(c+1) has landed in LDS
        __syncthreads();                                      // U(c+1), raw(c+2) visible; every wave is done with Us[c&1]
#endif
        WG_MARK();
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
    WG_MARK();                                                // last: epilogue issued
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
    const dim3 grid((unsigned)ceil_div(ceil_div(W, 2), 8), (unsigned)ceil_div(ceil_div(H, 2), 8), (unsigned)n);
    hipExtLaunchKernelGGL(winograd_conv64_kernel, grid, dim3(WG_THREADS), 0, st, ev0, ev1, 0, x, u_packed, bias, y, (int)H, (int)W, relu);
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
