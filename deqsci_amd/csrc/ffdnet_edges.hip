// FFDNet edge layers fused with the reference's hand-written layout layers (networks/ffdnet/functions.py):
//
//   tail: conv3x3(64 -> 4, pad 1, no bias) + upsamplefeatures (functions.py:62-81, = pixel_shuffle 2)
//         reading the channels_last activation once and writing the planar full-resolution noise image.
//         MIOpen runs this 4-output-channel layer as a 64x16-tile implicit GEMM (424 us at 64 images of
//         128x128x64, plus a 32 us zero-fill and a 10 us shuffle copy); it is 4.8 GFLOP over 268 MB, i.e.
//         a VALU/HBM-balanced stencil, not a GEMM.
//
// Mapping (tail): block = 8 x 32 half-resolution positions, one lane per position, 4 accumulators (the 2x2
// output pixels).  The 64 input channels are staged through LDS in two halves of 32 ([10 x 34 pixels][32+4]
// floats = 49 KB -> 3 blocks per CU); the +4 pad makes the per-lane ds_read_b128 conflict-free.  Weights are
// wave-uniform and pre-packed [half][tap][cin][cout], so they come through the scalar cache (s_load) and
// every v_fma takes its weight from an SGPR: 144 ds_read_b128 feed 2304 FMAs per lane.
#include "common.hpp"

namespace deqsci {

typedef float f32x2 __attribute__((ext_vector_type(2)));
// acc += s.x * w  /  acc += s.y * w  on a register pair, fused (one rounding per lane, exactly fmaf): ONE v_pk_fma_f32, the
// broadcast of s.x / s.y is an operand modifier (op_sel) of the instruction
__device__ __forceinline__ void pk_fma_lo(f32x2& acc, f32x2 s, f32x2 w) {
    acc = __builtin_elementwise_fma(__builtin_shufflevector(s, s, 0, 0), w, acc);
}
__device__ __forceinline__ void pk_fma_hi(f32x2& acc, f32x2 s, f32x2 w) {
    acc = __builtin_elementwise_fma(__builtin_shufflevector(s, s, 1, 1), w, acc);
}

constexpr int TT_H = 8, TT_W = 32;                 // output tile (half-res positions)
constexpr int TT_IW = TT_W + 2, TT_IH = TT_H + 2;  // input tile with halo
constexpr int TT_CH = 32;                          // channels per LDS pass
constexpr int TT_PS = TT_CH + 4;                   // LDS pixel stride (floats)


// ---- "sp16" activations (csrc/conv_s16.hip): [n][chunk c (4)][piece hl (2)][block kb (2)][H][W][8 halfs] holding 2^e x as hi + lo.
// The 64->64 layers behind a head run on that layout when the engine picks the split-fp16 convolution; the 1 -> 64 stencil head then
// writes it directly (OUT_SP16) instead of a conversion pass (FFDNet's head and both tails have matrix-core forms in conv_s16.hip).
typedef _Float16 eh2 __attribute__((ext_vector_type(2)));
typedef unsigned eu4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void sp16_split2(float a, float b, float scale, unsigned& hi, unsigned& lo) {
    a *= scale; b *= scale;
    const _Float16 ha = (_Float16)a, hb = (_Float16)b;
    hi = __builtin_bit_cast(unsigned, (eh2){ha, hb});
    lo = __builtin_bit_cast(unsigned, (eh2){(_Float16)(a - (float)ha), (_Float16)(b - (float)hb)});
}
// 16 lanes per position, lane cq owning couts 4 cq .. 4 cq + 3 (the vector-ALU heads): lanes cq and cq ^ 1 hold the two halves of one block
// of 8 couts.  The even lane ends up with the block's hi piece, the odd lane with its lo piece: one 16-byte store each.
__device__ __forceinline__ void sp16_store_quad(char* ysp, int64_t HW, int64_t pos, int cq, float4 v, float scale, bool ok) {
    unsigned h0, h1, l0, l1;
    sp16_split2(v.x, v.y, scale, h0, l0);
    sp16_split2(v.z, v.w, scale, h1, l1);
    const bool odd = cq & 1;
    const unsigned s0 = odd ? h0 : l0, s1 = odd ? h1 : l1;                 // what the partner stores: the even lane sends its lo, the odd its hi
    const unsigned r0 = (unsigned)__builtin_amdgcn_mov_dpp((int)s0, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]: lane ^ 1
    const unsigned r1 = (unsigned)__builtin_amdgcn_mov_dpp((int)s1, 0xB1, 0xF, 0xF, true);
    const eu4 o = odd ? (eu4){r0, r1, l0, l1} : (eu4){h0, h1, r0, r1};       // couts 0..7 of the block: even lane hi piece, odd lane lo piece
    const int blk = cq >> 1;                                              // chunk blk >> 1, block kb = blk & 1
    if (ok) *reinterpret_cast<eu4*>(ysp + (((int64_t)((blk >> 1) * 4 + (odd ? 2 : 0) + (blk & 1))) * HW + pos) * 16) = o;
}

template <int COUT>   // COUT 4: FFDNet tail (2x2 pixel shuffle on the way out); 1: plain 64 -> 1 layer (SimpleCNN tail)
__global__ __launch_bounds__(TB) void edge_tail_kernel(const float* __restrict__ h, const float* __restrict__ wp,
                                                       const float* __restrict__ bias, float* __restrict__ out, int H, int W) {
    __shared__ __attribute__((aligned(16))) float tile[TT_IH * TT_IW * TT_PS];
    const int n = blockIdx.z;
    const int r0 = blockIdx.y * TT_H, c0 = blockIdx.x * TT_W;
    const int lr = threadIdx.x / TT_W, lc = threadIdx.x % TT_W;

    const float* hn = h + (int64_t)n * H * W * 64;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.0f;
    // staging: [TT_IH][TT_IW][32] per half (zero outside the image), 8 float4 per pixel, NST float4 per lane.  All loads of a
    // half are issued back to back into registers, and the second half is requested BEFORE the first is computed on.
    constexpr int NE = TT_IH * TT_IW * (TT_CH / 4), NST = (NE + TB - 1) / TB;
    float4 stg[NST];
    int sdst[NST];                                            // LDS float offset, -1 = no element
    int64_t soff[NST];                                        // global element offset, -1 = outside the image (zero)
#pragma unroll
    for (int i = 0; i < NST; ++i) {
        const int e = threadIdx.x + i * TB;
        const int pix = e / (TT_CH / 4), q = e % (TT_CH / 4);
        const int pr = pix / TT_IW, pc = pix % TT_IW;
        const int gr = r0 + pr - 1, gc = c0 + pc - 1;
        sdst[i] = e < NE ? pix * TT_PS + 4 * q : -1;
        soff[i] = (e < NE && gr >= 0 && gr < H && gc >= 0 && gc < W) ? ((int64_t)gr * W + gc) * 64 + 4 * q : -1;
    }
    auto fetch_half = [&](int half) {
#pragma unroll
        for (int i = 0; i < NST; ++i) stg[i] = soff[i] >= 0 ? ld4(hn + soff[i] + half * TT_CH) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    };
    auto store_half = [&](int half) {
#pragma unroll
        for (int i = 0; i < NST; ++i) {
            float4 v = stg[i];
            if (bias && soff[i] >= 0) {   // the previous layer's folded-BN bias + ReLU applied on the way in (padding stays 0)
                const float4 b = ld4(bias + half * TT_CH + (sdst[i] % TT_PS));
                v.x = fmaxf(v.x + b.x, 0.0f); v.y = fmaxf(v.y + b.y, 0.0f); v.z = fmaxf(v.z + b.z, 0.0f); v.w = fmaxf(v.w + b.w, 0.0f);
            }
            if (sdst[i] >= 0) *reinterpret_cast<float4*>(tile + sdst[i]) = v;
        }
    };
    fetch_half(0);
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
        store_half(half);
        __syncthreads();
        if (half == 0) fetch_half(1);                         // in flight under the FMAs of the first half
        const float* wh = wp + half * 9 * TT_CH * COUT;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float* tp = tile + ((lr + tap / 3) * TT_IW + (lc + tap % 3)) * TT_PS;
            const float* wt = wh + tap * TT_CH * COUT;
#pragma unroll
            for (int c4 = 0; c4 < TT_CH / 4; ++c4) {
                const float4 v = *reinterpret_cast<const float4*>(tp + 4 * c4);
                const float* w = wt + c4 * 4 * COUT;
#pragma unroll
                for (int o = 0; o < COUT; ++o)
                    acc[o] = fmaf(v.w, w[3 * COUT + o], fmaf(v.z, w[2 * COUT + o], fmaf(v.y, w[COUT + o], fmaf(v.x, w[o], acc[o]))));
            }
        }
        __syncthreads();
    }
    const int r = r0 + lr, c = c0 + lc;
    if (r < H && c < W) {
        if (COUT == 4) {        // pixel_shuffle(2): channel 2i+j -> (2r+i, 2c+j)
            float* o = out + (int64_t)n * 4 * H * W + (int64_t)(2 * r) * (2 * W) + 2 * c;
            *reinterpret_cast<float2*>(o) = make_float2(acc[0], acc[COUT > 1 ? 1 : 0]);
            *reinterpret_cast<float2*>(o + 2 * W) = make_float2(acc[COUT > 2 ? 2 : 0], acc[COUT > 3 ? 3 : 0]);
        } else {
            out[(int64_t)n * H * W + (int64_t)r * W + c] = acc[0];
        }
    }
}

// plain head: conv3x3(1 -> 64, pad 1, no bias) [+ ReLU], planar (n,1,H,W) image in, channels_last (n,H,W,64) out.
// Same mapping as the FFDNet head below: 16 lanes per position x 4 output channels each, 9 x 4 weights in
// registers, a 34 x 34 patch in LDS read by broadcast; the kernel is bound by its 256 B/position store.
constexpr int H1_T = 32, H1_P = H1_T + 2, H1_PS = H1_P + 1;
// OUT_SP16: the output as an sp16 activation (1) or a p32 activation (2: the 16-byte pixels of csrc/conv_w16.hip, 2^e y unsplit, even / odd
// columns of a 64-column block apart) holding 2^e y, e from the range (out_amax, out_exp) (common.hpp); `track` (may be NULL):
// max |y| of the launch folded into *track - the range measurement of the first f-call
template <int OUT_SP16>
__global__ __launch_bounds__(TB) void conv_c1_to_64_kernel(const float* __restrict__ x, const float* __restrict__ wq,
                                                           float* __restrict__ h, int H, int W, int relu, const float* __restrict__ out_amax,
                                                           int out_exp, float* __restrict__ track) {
    __shared__ float patch[H1_P * H1_PS];
    __shared__ uint32_t trk_s[TB / WAVE];
    float tmax = 0.0f;
    const int n = blockIdx.z;
    const int r0 = blockIdx.y * H1_T, c0 = blockIdx.x * H1_T;
    const float oscale = OUT_SP16 ? sp16_pow2(out_amax ? sp16_act_exp(out_amax[n]) : out_exp) : 1.0f;     // (the ranges are per image)
    const float* xn = x + (int64_t)n * H * W;
    for (int e = threadIdx.x; e < H1_P * H1_P; e += TB) {
        const int pr = e / H1_P, pc = e % H1_P;
        const int gr = r0 - 1 + pr, gc = c0 - 1 + pc;
        patch[pr * H1_PS + pc] = (gr >= 0 && gr < H && gc >= 0 && gc < W) ? xn[(int64_t)gr * W + gc] : 0.0f;
    }
    const int cq = threadIdx.x % 16, slot = threadIdx.x / 16;
    float4 wr[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wr[k] = ld4(wq + (k * 16 + cq) * 4);
    __syncthreads();
    float* hn = h + (int64_t)n * H * W * 64;
#pragma unroll 2
    for (int it = 0; it < H1_T * H1_T / 16; ++it) {
        const int q = it * 16 + slot;
        const int lr = q / H1_T, lc = q % H1_T;
        const int r = r0 + lr, c = c0 + lc;
        float4 acc = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) acc = fma4(patch[(lr + tap / 3) * H1_PS + lc + tap % 3], wr[tap], acc);
        if (r < H && c < W) {
            if (relu) { acc.x = fmaxf(acc.x, 0.0f); acc.y = fmaxf(acc.y, 0.0f); acc.z = fmaxf(acc.z, 0.0f); acc.w = fmaxf(acc.w, 0.0f); }
            if (OUT_SP16) {
                if (track) tmax = fmaxf(fmaxf(tmax, fmaxf(fabsf(acc.x), fabsf(acc.y))), fmaxf(fabsf(acc.z), fabsf(acc.w)));   // (uniform branch)
                if (OUT_SP16 == 2) {                            // plane cq = couts 4 cq .. 4 cq + 3
                    const int64_t nbc = (W + 63) >> 6;
                    st4(h + ((((((int64_t)n * 16 + cq) * H + r) * nbc + (c >> 6)) * 2 + (c & 1)) * 32 + ((c & 63) >> 1)) * 4,
                        make_float4(acc.x * oscale, acc.y * oscale, acc.z * oscale, acc.w * oscale));
                } else {
                    sp16_store_quad(reinterpret_cast<char*>(h) + (int64_t)n * H * W * 256, (int64_t)H * W, (int64_t)r * W + c, cq, acc, oscale, true);
                }
            } else st4(hn + ((int64_t)r * W + c) * 64 + 4 * cq, acc);
        }
    }
    if (OUT_SP16 && track) sp16_track_block_max(tmax, 1.0f, track + n, trk_s);
}

// ------------------------------------------------------------------------------------------------
// head: concatenate_input_noise_map (functions.py:16-53: sigma map + 2x2 pixel-unshuffle, channel 2i+j)
//       + conv3x3(5 -> 64, pad 1, no bias) + ReLU, reading the planar full-resolution image and writing the
//       channels_last (n,H,W,64) activation once.  PyTorch runs cat, unshuffle, a layout copy, MIOpen's
//       zero-fill + conv and a ReLU sweep for this (~250 us at 64 images); it is 6 GFLOP and one 268 MB write.
// Mapping: 16 lanes per output position, each owning 4 output channels with their 4 x 45 weights resident in
// VGPRs for the whole 32 x 32-position tile; a wavefront handles 4 neighbouring positions, so its store is one
// contiguous 1 KiB.  The 68 x 68 full-resolution patch sits in LDS (zero outside the image = the conv's zero
// padding of the unshuffled channels); the 16 lanes of a position read it by broadcast.
// HD_T = tile side in half-res positions: 32, or 16 for grids that would leave CUs idle (8 images of 128 x 128 are 128 tiles of
// 32 x 32 - half the chip - each a serial loop of 64 trips)
template <int HD_T>
__global__ __launch_bounds__(TB) void ffdnet_head_kernel(const float* __restrict__ x, const float* __restrict__ wq,
                                                         const float* __restrict__ sigma, int sigma_stride,
                                                         float* __restrict__ h, int H, int W) {
    constexpr int HD_P = 2 * HD_T + 4;            // patch side in full-res pixels
    constexpr int HD_PS = HD_P + 2;               // LDS row stride (even: float2 reads stay 8-B aligned)
    constexpr int HD_SS = HD_T + 3;               // row stride of the sigma plane (HD_T + 2 columns)
    __shared__ __attribute__((aligned(16))) float patch[HD_P * HD_PS];
    __shared__ float sgm[(HD_T + 2) * HD_SS];
    const int n = blockIdx.z;
    const int r0 = blockIdx.y * HD_T, c0 = blockIdx.x * HD_T;
    const int H2 = 2 * H, W2 = 2 * W;
    const float* xn = x + (int64_t)n * H2 * W2;
    for (int e = threadIdx.x; e < HD_P * HD_P; e += TB) {
        const int pr = e / HD_P, pc = e % HD_P;
        const int gr = 2 * r0 - 2 + pr, gc = 2 * c0 - 2 + pc;
        patch[pr * HD_PS + pc] = (gr >= 0 && gr < H2 && gc >= 0 && gc < W2) ? xn[(int64_t)gr * W2 + gc] : 0.0f;
    }
    // the sigma plane of the tile with its ring: sigma inside the image, 0 in the conv's zero padding (9 compares and
    // selects per position otherwise - the loop below is bound by instruction count, not by its FMAs)
    const float sig = sigma[(int64_t)n * sigma_stride];
    for (int e = threadIdx.x; e < (HD_T + 2) * (HD_T + 2); e += TB) {
        const int pr = e / (HD_T + 2), pc = e % (HD_T + 2);
        const int rr = r0 - 1 + pr, cc = c0 - 1 + pc;
        sgm[pr * HD_SS + pc] = (rr >= 0 && rr < H && cc >= 0 && cc < W) ? sig : 0.0f;
    }
    const int cq = threadIdx.x % 16, slot = threadIdx.x / 16;
    // [ch*9 + tap] -> output channels 4cq..4cq+3 as two register pairs: every multiply-add below is ONE v_pk_fma_f32 on a
    // pair of output channels (2 x the scalar FMA rate), its input picked from the low / high half of the float2 the LDS
    // read delivered (op_sel), so no broadcast moves are needed
    f32x2 wl[45], wh[45];
#pragma unroll
    for (int k = 0; k < 45; ++k) {
        const float4 t = ld4(wq + (k * 16 + cq) * 4);
        wl[k] = (f32x2){t.x, t.y};
        wh[k] = (f32x2){t.z, t.w};
    }
    __syncthreads();
    float* hn = h + (int64_t)n * H * W * 64;
#pragma unroll 1
    for (int it = 0; it < HD_T * HD_T / 16; ++it) {
        const int q = it * 16 + slot;
        const int lr = q / HD_T, lc = q % HD_T;
        const int r = r0 + lr, c = c0 + lc;
        f32x2 al = {0.0f, 0.0f}, ah = {0.0f, 0.0f};
#pragma unroll
        for (int dr = 0; dr < 3; ++dr) {            // channel 0: the sigma map (zero in the conv's padding ring), from LDS
            const float* sp = sgm + (lr + dr) * HD_SS + lc;
            const f32x2 s01 = {sp[0], sp[1]}, s2x = {sp[2], sp[2]};
            pk_fma_lo(al, s01, wl[dr * 3]);     pk_fma_lo(ah, s01, wh[dr * 3]);
            pk_fma_hi(al, s01, wl[dr * 3 + 1]); pk_fma_hi(ah, s01, wh[dr * 3 + 1]);
            pk_fma_lo(al, s2x, wl[dr * 3 + 2]); pk_fma_lo(ah, s2x, wh[dr * 3 + 2]);
        }
#pragma unroll
        for (int prow = 0; prow < 6; ++prow) {      // full-res rows 2r-2 .. 2r+3: dr = prow/2 - 1, i = prow%2
            const float* pp = patch + (2 * lr + prow) * HD_PS + 2 * lc;
            f32x2 v[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) v[j] = *reinterpret_cast<const f32x2*>(pp + 2 * j);
#pragma unroll
            for (int pcol = 0; pcol < 6; ++pcol) {  // dc = pcol/2 - 1, j = pcol%2
                const int ch = 1 + 2 * (prow % 2) + (pcol % 2);
                const int tap = (prow / 2) * 3 + (pcol / 2);
                if (pcol % 2 == 0) { pk_fma_lo(al, v[pcol / 2], wl[ch * 9 + tap]); pk_fma_lo(ah, v[pcol / 2], wh[ch * 9 + tap]); }
                else { pk_fma_hi(al, v[pcol / 2], wl[ch * 9 + tap]); pk_fma_hi(ah, v[pcol / 2], wh[ch * 9 + tap]); }
            }
        }
        if (r < H && c < W) {
            const float4 o4 = make_float4(fmaxf(al[0], 0.0f), fmaxf(al[1], 0.0f), fmaxf(ah[0], 0.0f), fmaxf(ah[1], 0.0f));
            st4(hn + ((int64_t)r * W + c) * 64 + 4 * cq, o4);
        }
    }
}

// The same head on the fp32 matrix cores (large launches): per group of 16 neighbouring positions a 64 x 48 x 16 product
//   D[cout][position] = sum_k W[cout][k] * P[k][position],   k = 9 sigma taps (+3 empty) | 4 sub-pixel channels x 9 taps,
// i.e. 48 v_mfma_f32_16x16x4_f32 (4 cout groups x 12 k-steps) instead of 90 packed FMAs per position and lane: the vector ALU is
// left with one LDS gather per k-step (the P operand: lane (k, position) reads ITS tap of ITS position from the patch), the
// ReLU and the 16-byte stores.  A wave keeps all 48 weight operands in registers and walks 16 groups of its 32 x 32 tile.
typedef float f32x4m __attribute__((ext_vector_type(4)));
#ifndef HEAD_ST
#define HEAD_ST st4
#endif
__global__ __launch_bounds__(TB) void ffdnet_head_mfma_kernel(const float* __restrict__ x, const float* __restrict__ wq,
                                                              const float* __restrict__ sigma, int sigma_stride,
                                                              float* __restrict__ h, int H, int W) {
    constexpr int T = 32, P = 2 * T + 4, PS = P + 2, SS = T + 3;
    constexpr int SGM = P * PS;                                   // the sigma plane sits behind the patch in one LDS array
    __shared__ __attribute__((aligned(16))) float lds[P * PS + (T + 2) * SS];
    const int n = blockIdx.z;
    const int r0 = blockIdx.y * T, c0 = blockIdx.x * T;
    const int H2 = 2 * H, W2 = 2 * W;
    const float* xn = x + (int64_t)n * H2 * W2;
    for (int e = threadIdx.x; e < P * P; e += TB) {
        const int pr = e / P, pc = e % P;
        const int gr = 2 * r0 - 2 + pr, gc = 2 * c0 - 2 + pc;
        lds[pr * PS + pc] = (gr >= 0 && gr < H2 && gc >= 0 && gc < W2) ? xn[(int64_t)gr * W2 + gc] : 0.0f;
    }
    const float sig = sigma[(int64_t)n * sigma_stride];
    for (int e = threadIdx.x; e < (T + 2) * (T + 2); e += TB) {
        const int pr = e / (T + 2), pc = e % (T + 2);
        const int rr = r0 - 1 + pr, cc = c0 - 1 + pc;
        lds[SGM + pr * SS + pc] = (rr >= 0 && rr < H && cc >= 0 && cc < W) ? sig : 0.0f;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int pn = lane & 15, kq = lane >> 4;
    // weight operands (A: row = cout 16 cg + pn, column = k-slot 4 ks + kq) and the LDS offset of this lane's tap per k-step
    float wa[4][12];
    int off[12];
#pragma unroll
    for (int ks = 0; ks < 12; ++ks) {
        const int slot = 4 * ks + kq;
        const bool sg = ks < 3;
        const int korig = sg ? slot : 9 + (slot - 12);             // index into the packed [ch*9+tap] weights
        const bool valid = !sg || slot < 9;
        const int tap = sg ? (valid ? slot : 0) : (slot - 12) % 9, ch = sg ? 0 : 1 + (slot - 12) / 9;
        const int dr = tap / 3, dc = tap % 3;
        off[ks] = sg ? SGM + dr * SS + dc + pn : (2 * dr + ((ch - 1) >> 1)) * PS + 2 * dc + ((ch - 1) & 1) + 2 * pn;
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) {
            const int cout = 16 * cg + pn;
            wa[cg][ks] = valid ? wq[(korig * 16 + (cout >> 2)) * 4 + (cout & 3)] : 0.0f;
        }
    }
    __syncthreads();
    float* hn = h + (int64_t)n * H * W * 64;
#pragma unroll 1
    for (int g = wave; g < T * T / 16; g += TB / 64) {
        const int lr = g >> 1, lc0 = 16 * (g & 1);                 // 16 positions of row lr, columns lc0 .. lc0 + 15
        const int gs = lr * SS + lc0, gp = 2 * lr * PS + 2 * lc0;  // (uniform) group offsets in the sigma plane / the patch
        float pv[12];
#pragma unroll
        for (int ks = 0; ks < 12; ++ks) pv[ks] = lds[off[ks] + (ks < 3 ? gs : gp)];
        f32x4m acc[4];
#pragma unroll
        for (int cg = 0; cg < 4; ++cg) acc[cg] = (f32x4m){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int ks = 0; ks < 12; ++ks)
#pragma unroll
            for (int cg = 0; cg < 4; ++cg) acc[cg] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[cg][ks], pv[ks], acc[cg], 0, 0, 0);
        const int r = r0 + lr, c = c0 + lc0 + pn;
        if (r < H && c < W) {
            {
                float* o = hn + ((int64_t)r * W + c) * 64 + 4 * kq;    // D: column = position pn, rows 4 kq .. 4 kq + 3 of the cout group
#pragma unroll
                for (int cg = 0; cg < 4; ++cg)
                    HEAD_ST(o + 16 * cg, make_float4(fmaxf(acc[cg][0], 0.0f), fmaxf(acc[cg][1], 0.0f), fmaxf(acc[cg][2], 0.0f), fmaxf(acc[cg][3], 0.0f)));
            }
        }
    }
}

}  // namespace deqsci

using namespace deqsci;

extern "C" int deqsci_ffdnet_head_f32(const float* x, const float* w_packed, const float* sigma, int64_t sigma_stride, float* h,
                                      int64_t n, int64_t H, int64_t W, deqsci_stream_t stream) {
    if (!x || !w_packed || !sigma || !h) return DEQSCI_ERR_NULL;
    if (n <= 0 || H <= 0 || W <= 0 || sigma_stride < 0) return DEQSCI_ERR_SHAPE;
    if (n > 65535 || H > (1 << 20) || W > (1 << 20)) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(w_packed) || !aligned16(h)) return DEQSCI_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
#ifdef DEQSCI_DIAG
    const int head_vector_only = diag_env_int("DEQSCI_HEAD_VALU", 0);   // (A/B knob of the diagnostic build)
#else
    constexpr int head_vector_only = 0;
#endif
#define HEAD_LAUNCH(KERNEL, T)                                                                                                          \
    do {                                                                                                                                \
        const dim3 grid((unsigned)ceil_div(W, T), (unsigned)ceil_div(H, T), (unsigned)n);                                               \
        hipLaunchKernelGGL(KERNEL, grid, dim3(TB), 0, st, x, w_packed, sigma, (int)sigma_stride, h, (int)H, (int)W);                    \
    } while (0)
    const bool big = ceil_div(W, 32) * ceil_div(H, 32) * n >= 2 * (int64_t)num_cus();
    if (!head_vector_only && big) HEAD_LAUNCH(ffdnet_head_mfma_kernel, 32);
    else if (big) HEAD_LAUNCH((ffdnet_head_kernel<32>), 32);
    else HEAD_LAUNCH((ffdnet_head_kernel<16>), 16);
#undef HEAD_LAUNCH
    return launch_status();
}

template <int COUT>
static int tail_impl(const float* h, const float* w_packed, const float* in_bias, float* out, int64_t n, int64_t H, int64_t W, deqsci_stream_t stream) {
    if (!h || !w_packed || !out) return DEQSCI_ERR_NULL;
    if (in_bias && !aligned16(in_bias)) return DEQSCI_ERR_ALIGN;
    if (n <= 0 || H <= 0 || W <= 0) return DEQSCI_ERR_SHAPE;
    if (n > 65535 || H > (1 << 20) || W > (1 << 20)) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(h) || !aligned16(w_packed) || !aligned16(out)) return DEQSCI_ERR_ALIGN;
    const dim3 grid((unsigned)ceil_div(W, TT_W), (unsigned)ceil_div(H, TT_H), (unsigned)n);
    hipLaunchKernelGGL((edge_tail_kernel<COUT>), grid, dim3(TB), 0, static_cast<hipStream_t>(stream), h, w_packed, in_bias, out, (int)H, (int)W);
    return launch_status();
}

extern "C" int deqsci_ffdnet_tail_f32(const float* h, const float* w_packed, const float* in_bias, float* out, int64_t n, int64_t H,
                                      int64_t W, deqsci_stream_t stream) {
    return tail_impl<4>(h, w_packed, in_bias, out, n, H, W, stream);
}

extern "C" int deqsci_conv3x3_c64_to_1_f32(const float* h, const float* w_packed, const float* in_bias, float* out, int64_t n,
                                           int64_t H, int64_t W, deqsci_stream_t stream) {
    return tail_impl<1>(h, w_packed, in_bias, out, n, H, W, stream);
}

static int c1_to_64_impl(const float* x, const float* w_packed, float* h, int64_t n, int64_t H, int64_t W, int relu, int sp16, const float* out_amax,
                         int out_exp, float* track_amax, deqsci_stream_t stream) {
    if (!x || !w_packed || !h) return DEQSCI_ERR_NULL;
    if (n <= 0 || H <= 0 || W <= 0) return DEQSCI_ERR_SHAPE;
    if (n > 65535 || H > (1 << 20) || W > (1 << 20) || out_exp < -SP16_EXP_LIMIT || out_exp > SP16_EXP_LIMIT) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(w_packed) || !aligned16(h)) return DEQSCI_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)ceil_div(W, H1_T), (unsigned)ceil_div(H, H1_T), (unsigned)n);
    if (sp16 == 2) hipLaunchKernelGGL(conv_c1_to_64_kernel<2>, grid, dim3(TB), 0, st, x, w_packed, h, (int)H, (int)W, relu, out_amax, out_exp, track_amax);
    else if (sp16) hipLaunchKernelGGL(conv_c1_to_64_kernel<1>, grid, dim3(TB), 0, st, x, w_packed, h, (int)H, (int)W, relu, out_amax, out_exp, track_amax);
    else hipLaunchKernelGGL(conv_c1_to_64_kernel<0>, grid, dim3(TB), 0, st, x, w_packed, h, (int)H, (int)W, relu, out_amax, out_exp, track_amax);
    return launch_status();
}

extern "C" int deqsci_conv3x3_c1_to_64_f32(const float* x, const float* w_packed, float* h, int64_t n, int64_t H, int64_t W,
                                           int relu, deqsci_stream_t stream) {
    return c1_to_64_impl(x, w_packed, h, n, H, W, relu, 0, nullptr, 0, nullptr, stream);
}

extern "C" int deqsci_conv3x3_c1_to_64_sp16(const float* x, const float* w_packed, void* h_sp16, int64_t n, int64_t H, int64_t W,
                                            int relu, const float* out_amax, int out_exp, float* track_amax, deqsci_stream_t stream) {
    return c1_to_64_impl(x, w_packed, static_cast<float*>(h_sp16), n, H, W, relu, 1, out_amax, out_exp, track_amax, stream);
}

extern "C" int deqsci_conv3x3_c1_to_64_p32(const float* x, const float* w_packed, void* h_p32, int64_t n, int64_t H, int64_t W,
                                           int relu, const float* out_amax, int out_exp, float* track_amax, deqsci_stream_t stream) {
    return c1_to_64_impl(x, w_packed, static_cast<float*>(h_p32), n, H, W, relu, 2, out_amax, out_exp, track_amax, stream);
}
