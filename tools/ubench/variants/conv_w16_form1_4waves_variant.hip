// 3x3 convolution 64 -> 64 channels (pad 1, stride 1) on the f16 matrix cores with the split-fp16 arithmetic of csrc/conv_s16.hip (every
// operand two fp16 pieces, three MFMAs per product, fp32 accumulation) under a WINOGRAD transform: F(2,3) ALONG X nested in the direct sum
// ALONG Y -
//        y[row][2t + o] = sum_xi A^T[o][xi] M[xi][row][t],     M[xi][row][t] = sum_dy sum_cin U[dy][xi][cout][cin] V[xi][row + dy][t][cin]
//        U[dy][xi] = G g[dy][.]  (host, float64),              V[xi][row][t] = (B^T d[row][2t - 1 .. 2t + 2])_xi   (in the kernel, fp32)
// - 4 multiplications per 2 outputs and tap row instead of 6: 18 f16 MFMA products per output and cin-cout pair instead of the direct
// kernel's 27.  Why this form and not F(2x2,3x3) (12): the matrix pipe of a CU is fed from 160 KB of LDS and 512 registers per lane and
// SIMD.  F(2x2,3x3) holds 16 accumulators per 4 outputs (a wave of 32 tiles x 64 couts: 512 registers), its 256 KB of transformed weights
// neither fit the LDS nor stream through it at the rate the MFMAs want (one operand read per 1.5 MFMAs), and with the weights pinned in
// registers the positions are split over waves and every output crosses the LDS once more (DESIGN section 6.5 has the arithmetic).  The
// nested form holds 4 accumulators per 2 outputs, transforms ROWS (each halo row once per wave, shared by the three tap rows), and its
// operand traffic is the direct kernel's.
//
// Measured on MI355X (tools/ubench/mfma_f16_fillers.hip, profiles/r05_mfma_f16_fillers.jsonl): one wave issues an MFMA 32x32x16 every 32
// cycles as long as the vector-ALU instructions between two MFMAs add up to < ~28 cycles (v_sub_f32 ~5, v_fma_mix / v_cvt_pk_f16_f32 ~8,
// v_pk_add_f32 ~19 - never that one), whether one or two waves share the SIMD.  So: ONE wave per SIMD with all 512 registers - 256
// accumulators (4 xi x 2 rows x 2 cout groups x 16) and the transform in the MFMA lanes' own registers, four VALU instructions behind
// every MFMA.
//
// Geometry.  Block tile = 8 rows x 64 columns of output pixels x 64 couts, one persistent 4-wave workgroup per CU; wave w owns output rows
// 2w, 2w + 1 (halo rows 2w .. 2w + 3), all 64 couts; MFMA N (32 lanes) = the 32 Winograd tiles of a row, lane (t, kb) holds channels
// 8 kb .. + 8 of tile t.  Input channels in chunks of 16 (K of one MFMA), each chunk in TWO half-stages (xi = 0, 1 | xi = 2, 3): a
// half-stage multiplies V of its two positions (computed one half-stage ahead, 64 registers) with 24 KB of weights
// ([xi'][dy][piece][cout group] fragments of 1 KB, host-packed, LDS-DMA, double-buffered by half-stage) - 72 MFMAs per wave - while the
// transform of the next half-stage runs in the gaps: per halo row 6 LDS reads (3 pixels), 16 subtractions, 8 packed converts + 16
// v_fma_mixlo/hi (hi + lo split).  The 10 x 66 pixel halo tile of a chunk (4 planes, 42 KB) is double-buffered by chunk; its LDS order
// separates even and odd columns, so that the lanes of a wave read consecutive 16-byte slots (conflict-free) for every pixel of their
// tiles.  One barrier per half-stage.
//
// Activations between layers: "sp16" as csrc/conv_s16.hip ([n][chunk][piece][k block][H][W][8 halfs]: hi + lo of 2^e x), or - FMT_P32 -
// "p32": the same 16 planes of 16-byte pixels holding 2^e x as fp32 ([n][chunk][k block][half][H][W][4 floats]: plane 2 b8 + j = channels
// 8 b8 + 4 j .. + 4).  p32 spares the kernel the hi + lo -> fp32 conversion of every pixel it transforms (24 v_fma_mix per halo row and
// half-stage: 40 % of the transform's issue time) and the split in its epilogue; the e of an image follows its range slot exactly as for
// sp16 (common.hpp), so the two formats are interchangeable layer by layer (FFDNet's first and last layer have both forms).
#include "common.hpp"
#include <hip/hip_ext.h>
#include <type_traits>
#pragma clang diagnostic ignored "-Winline-asm"

#ifndef W16_ABL
#define W16_ABL 0     // timing ablations only (results wrong): 1 = no DMA inside the half-stages, 2 = no transform, 4 = no epilogue
#endif

namespace deqsci {
namespace w16 {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

constexpr int WAVES = 4, TBW = 64 * WAVES;
constexpr int OUT_ROWS = 8, OUT_COLS = 64, RAW_ROWS = 10, RAW_COLS = 66, RAW_HALF = 33, RAW_PIX = RAW_ROWS * RAW_COLS;   // 660
constexpr int PLANE_B = RAW_PIX * 16;                          // 10560 bytes of one staged plane
constexpr int RAW_SLOTS = 4 * RAW_PIX;                         // 2640 units of 16 bytes per chunk tile
constexpr int RAW_INSTR = 11;                                  // LDS-DMA instructions of 64 units per wave and chunk (44 in all, 2816 slots)
constexpr int RAW_BUF = WAVES * RAW_INSTR * 1024;              // 45056 bytes
constexpr int W_FRAGS = 2 * 3 * 2 * 2;                         // fragments of a half-stage: [xi' (2)][dy (3)][piece: hi, lo (2)][cout group (2)]
constexpr int W_HALF = W_FRAGS * 1024;                         // 24576 bytes
constexpr int W_INSTR = W_FRAGS / WAVES;                       // 6 per wave and half-stage
constexpr uint32_t RAW_OOB = 0x80000000u;                      // beyond num_records: the hardware writes zeros

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int mdiv(int t, uint32_t mg, uint32_t sh) { return (int)(((uint64_t)(uint32_t)t * mg) >> sh); }

// hi + lo of an fp32 pair in three instructions: hi = v_cvt_pk_f16_f32 (round to nearest even), lo = fp16(a - hi) by v_fma_mixlo / mixhi
// (a - hi is exact in fp32: one rounding, the same bits as subtracting in fp32 and converting)
__device__ __forceinline__ void split_pair(float a0, float a1, unsigned& hi, unsigned& lo) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(a0), "v"(a1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(a1));
}
// fp32(hi) + fp32(lo) of the two halves of a register pair (exact: 22 bits)
__device__ __forceinline__ float join_lo(unsigned hi, unsigned lo) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hi), "v"(lo));
    return r;
}
__device__ __forceinline__ float join_hi(unsigned hi, unsigned lo) {
    float r;
    asm("v_fma_mix_f32 %0, %1, 1.0, %2 op_sel:[1,0,1] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(hi), "v"(lo));
    return r;
}

__device__ __forceinline__ float sub1(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float add1(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

struct StackLayer { const char* w; const float* bias; int w_exp; int relu; };
constexpr unsigned STACK_SPIN_LIMIT = 1u << 21;                // polls, one every ~0.1 us: a wait gives up after a quarter of a second
constexpr int STACK_POLLS = 8;                                 // polls of the neighbours' words in flight
constexpr int STACK_FLAG_STRIDE = 32;                          // words between two tiles' progress words: a 128-byte line each

// FMT_P32 = 0: sp16 in, sp16 out; 1: p32 in, p32 out.  STACK: a run of n_layers layers in one launch with dataflow synchronisation between
// the tiles (exactly the protocol of conv_s16_kernel<0, 0, 1>: progress words, write-through stores, agent-scope DMA loads; see there).
template <int FMT_P32, int STACK>
__global__ __launch_bounds__(TBW, 1) void conv_w16_kernel(const char* __restrict__ x, const char* __restrict__ Wp, const float* __restrict__ bias,
                                                          char* __restrict__ y, int H, int W, int relu, int w_exp, const float* __restrict__ in_amax, int in_exp,
                                                          const float* __restrict__ out_amax, int out_exp, int tiles_x, int tiles_y,
                                                          int n_tiles, uint32_t mg_img, uint32_t sh_img, uint32_t mg_tx, uint32_t sh_tx,
                                                          char* __restrict__ y2, const StackLayer* __restrict__ layers, int n_layers, unsigned* flags,
                                                          int range_stride) {
    __shared__ __attribute__((aligned(16))) char Raw[2 * RAW_BUF];
    __shared__ __attribute__((aligned(16))) char Wt[2 * W_HALF];
    __shared__ __attribute__((aligned(16))) float bias_s[STACK ? 128 : 64];     // (STACK: per layer parity)
    __shared__ uint32_t ready_s;                               // (STACK) wave 0's verdict on the next tile's inputs, for all waves
    __shared__ uint32_t poll_s[64];                            // (STACK) the words wave 0 polled in the shadow of half-stage 4 (by LDS-DMA: no register
                                                               // waits for a load that lands a half-stage later)
    __shared__ uint32_t voff_s[RAW_INSTR * TBW];               // per-lane global offsets of the halo-tile DMA instructions (see voff_set)
    const int lane = (int)(threadIdx.x & 63);
    const int wave = (int)uniform((uint32_t)(threadIdx.x >> 6));
    int t_first, t_step, t_end;
    {
        const int nb = (int)gridDim.x, b = (int)blockIdx.x;
        if ((nb & 7) == 0) {                                   // block b runs on XCD b % 8: give every XCD a contiguous range of tiles
            const int per_xcd = (n_tiles + 7) >> 3;
            t_first = (b & 7) * per_xcd + (b >> 3);
            t_step = nb >> 3;
            t_end = min(n_tiles, ((b & 7) + 1) * per_xcd);
        } else { t_first = b; t_step = nb; t_end = n_tiles; }
    }
    if (t_first >= t_end) return;
    const int64_t HW = (int64_t)H * W;
    const int pl = lane & 31, kb = lane >> 5;

    // ---- halo tile by LDS-DMA: slot s = 64 (11 wave + j) + lane of the chunk tile is plane p = s / 660, row (s % 660) / 66, and inside the
    // row the EVEN columns first (33), then the odd ones: lane-linear in LDS, a per-lane byte offset on the global side.
    i32x4 rsrc;
    int ft_py0 = 0, ft_px0 = 0;
    auto fetch_tile_uniform = [&](int t, const char* xb) __attribute__((always_inline)) {
        const int n = mdiv(t, mg_img, sh_img), r = t - n * (tiles_x * tiles_y);
        const int by = mdiv(r, mg_tx, sh_tx), bx = r - by * tiles_x;
        const uint64_t base = (uint64_t)(xb + (int64_t)n * HW * 256);
        rsrc.x = (int)uniform((uint32_t)base);
        rsrc.y = (int)uniform((uint32_t)(base >> 32));
        rsrc.z = (int)uniform((uint32_t)(HW * 256));
        rsrc.w = 0x00020000;
        ft_py0 = OUT_ROWS * by - 1;
        ft_px0 = OUT_COLS * bx - 1;
    };
    auto fetch_lane_offset = [&](int j) __attribute__((always_inline)) -> uint32_t {
        int w_ = wave;
        asm volatile("" : "+s"(w_));                           // (recomputed at every use: twice per tile and instruction)
        const int s = 64 * (RAW_INSTR * w_ + j) + lane;
        const int p = (s * 6356) >> 22;                        // s / 660 for s < 2816
        const int q = s - p * RAW_PIX;
        const int row = (q * 993) >> 16, rem = q - row * RAW_COLS;           // q / 66 for q < 660
        const int par = rem >= RAW_HALF ? 1 : 0, col = 2 * (rem - par * RAW_HALF) + par;
        const int iy = ft_py0 + row, ix = ft_px0 + col;
        const bool ok = s < RAW_SLOTS && (uint32_t)iy < (uint32_t)H && (uint32_t)ix < (uint32_t)W;
        uint32_t off = ((uint32_t)p * (uint32_t)HW + (uint32_t)(iy * W + ix)) * 16u;
        asm volatile("" : "+v"(off));
        return ok ? off : RAW_OOB;
    };
    // the lane offsets of the tile whose chunks are being fetched live in LDS (11 registers the transform needs more): written once per tile,
    // read one DMA instruction ahead
    uint32_t vo_next = 0;
    auto voff_set = [&](int j, uint32_t v) __attribute__((always_inline)) { voff_s[j * TBW + (int)threadIdx.x] = v; };
    auto voff_get = [&](int j) __attribute__((always_inline)) { vo_next = voff_s[j * TBW + (int)threadIdx.x]; };
    const uint32_t raw_lds = (uint32_t)(uintptr_t)(lds_char*)Raw, wt_lds = (uint32_t)(uintptr_t)(lds_char*)Wt;
    auto raw_piece = [&](int c, int buf, int j) __attribute__((always_inline)) {
        int w_ = wave;
        asm volatile("" : "+s"(w_));
        const uint32_t soff = uniform((uint32_t)c * (uint32_t)HW * 64u);                               // 4 planes of 16 HW bytes per chunk
        const uint32_t m0v = uniform(raw_lds + (uint32_t)(buf * RAW_BUF + (RAW_INSTR * w_ + j) * 1024));
        const uint32_t voj = vo_next;                          // (voff_get(j) ran a gap ago)
        if (j + 1 < RAW_INSTR) voff_get(j + 1);
        if (STACK) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds" ::"s"(m0v), "v"(voj), "s"(rsrc), "s"(soff) : "m0");
        else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(voj), "s"(rsrc), "s"(soff) : "m0");
    };
    // ---- weights of half-stage hs = 2 c + h: 24 fragments of 1 KiB, host-packed in LDS order; wave w moves fragments 6 w .. 6 w + 5
    auto w_piece = [&](const char* Wl, int hs, int buf, int j) __attribute__((always_inline)) {
        int w_ = wave;
        asm volatile("" : "+s"(w_));
        const uint32_t off = (uint32_t)((W_INSTR * w_ + j) * 1024);
        const uint64_t g = (uint64_t)(Wl + (int64_t)hs * W_HALF) + off;
        const uint32_t m0v = uniform(wt_lds + (uint32_t)(buf * W_HALF) + off);
        const uint64_t gs = ((uint64_t)uniform((uint32_t)(g >> 32)) << 32) | uniform((uint32_t)g);
        const uint32_t lv = (uint32_t)lane * 16u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(lv), "s"(gs) : "m0");
    };

    // acc[xi][r][g]: ONE accumulation chain per position (36 MFMAs: the cross products of a (dy, chunk) step first, then hi x hi)
    f32x16 acc[4][2][2];
    // V of a half-stage: [buffer][halo row rho (4)][xi' (2)] as the MFMA B operand of lane (tile, kb): hi and lo pieces
    u32x4 Vh[2][4][2], Vl[2][4][2];
    struct Done { i32x4 orsrc; uint32_t pix[2]; int oy0; float oscale, bscale; };

    // operand addresses: the lane's first halo row / tile slot in plane (hi | first half) of its k block; weight fragments lane-linear
    // (opaque to the compiler: folded into the arrays' absolute LDS addresses, every offset beyond 64 KB becomes an address register of its own)
    uint32_t lb0 = (uint32_t)(uintptr_t)(lds_char*)Raw + (uint32_t)((FMT_P32 ? 2 * kb : kb) * PLANE_B + (2 * wave * RAW_COLS + pl) * 16);
    uint32_t lb1 = lb0 + RAW_BUF, ab0 = (uint32_t)(uintptr_t)(lds_char*)Wt + (uint32_t)lane * 16u;
    asm volatile("" : "+v"(lb0), "+v"(lb1), "+v"(ab0));
    const lds_char* const lbase[2] = {(const lds_char*)(uintptr_t)lb0, (const lds_char*)(uintptr_t)lb1};
    constexpr int FRAG2 = FMT_P32 ? PLANE_B : 2 * PLANE_B;     // from a pixel's first 16 bytes (hi | channels 0-3) to its second (lo | channels 4-7)
    const lds_char* abase = (const lds_char*)(uintptr_t)ab0;

    // ---- the input transform of ONE halo row rho for the half-stage (c', h') in micro-steps (placed behind the MFMAs of the half-stage
    // before): raw[q][f] = pixel q + h' of the lane's tile (f: first | second 16 bytes), dd[q][k] = its 8 channels as fp32,
    // h' = 0: V0 = d0 - d2, V1 = d1 + d2;  h' = 1 (d0..2 = pixels 1..3): V2 = d1 - d0, V3 = d0 - d2
    u32x4 raw[3][2];
    float dd[3][8], va[8], vb[8];
    auto t_load = [&](int rbuf, int rho, int hn) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int jj = q + hn;
            const lds_char* p = lbase[rbuf] + (rho * RAW_COLS + (jj & 1) * RAW_HALF + (jj >> 1)) * 16;
            raw[q][0] = *reinterpret_cast<const lds_u32x4*>(p);
            raw[q][1] = *reinterpret_cast<const lds_u32x4*>(p + FRAG2);
        }
    };
    auto t_conv = [&](int s) __attribute__((always_inline)) {      // (sp16 only) s = 0..5: pixel s >> 1, channels 4 (s & 1) .. + 4
        const int q = s >> 1, half = s & 1;
        if (FMT_P32) return;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            dd[q][4 * half + 2 * k] = join_lo(raw[q][0][2 * half + k], raw[q][1][2 * half + k]);
            dd[q][4 * half + 2 * k + 1] = join_hi(raw[q][0][2 * half + k], raw[q][1][2 * half + k]);
        }
    };
    auto px = [&](int q, int k) __attribute__((always_inline)) -> float {
        const unsigned u = raw[q][k >> 2][k & 3];              // (by value: __builtin_bit_cast of a vector ELEMENT reads element 0)
        return FMT_P32 ? __builtin_bit_cast(float, u) : dd[q][k];
    };
    auto t_xf = [&](int s, int hn) __attribute__((always_inline)) {  // s = 0..3: channels 2 s, 2 s + 1
#pragma unroll
        for (int k = 2 * s; k < 2 * s + 2; ++k) {
            // (single v_sub / v_add by inline asm: left to itself hipcc pairs them into v_pk_add_f32, 19 cycles beside an MFMA against 2 x 5)
            if (hn == 0) { va[k] = sub1(px(0, k), px(2, k)); vb[k] = add1(px(1, k), px(2, k)); }
            else { va[k] = sub1(px(1, k), px(0, k)); vb[k] = sub1(px(0, k), px(2, k)); }
        }
    };
    auto t_split = [&](int s, int vbuf, int rho) __attribute__((always_inline)) {   // s = 0..3: channel pair s of both positions
        unsigned hi, lo;
        split_pair(va[2 * s], va[2 * s + 1], hi, lo);
        Vh[vbuf][rho][0][s] = hi; Vl[vbuf][rho][0][s] = lo;
        split_pair(vb[2 * s], vb[2 * s + 1], hi, lo);
        Vh[vbuf][rho][1][s] = hi; Vl[vbuf][rho][1][s] = lo;
    };
    // the micro-step of gap m (0 .. 71) of a half-stage whose successor is (chunk buffer rbuf, xi half hn, V buffer vbuf): 16 gaps per
    // halo row - [load][-][conv x 6][xf x 4][split x 4, the next row's load with the first] (p32: the conv steps are register renames)
    auto xf_step = [&](int m, int rbuf, int hn, int vbuf) __attribute__((always_inline)) {
        if ((W16_ABL & 2) || m >= 64) return;
        const int rho = m >> 4, k = m & 15;
        if (k == 0 && rho == 0) t_load(rbuf, 0, hn);
        else if (k >= 2 && k < 8) t_conv(k - 2);
        else if (k >= 8 && k < 12) t_xf(k - 8, hn);
        else if (k >= 12) {
            if (k == 12 && rho < 3) t_load(rbuf, rho + 1, hn);     // (the row's pixels are dead once its V is formed: the next row's land under the split)
            t_split(k - 12, vbuf, rho);
        }
    };
    auto t_all = [&](int rbuf, int hn, int vbuf) __attribute__((always_inline)) {   // the whole transform at once (prologue, slow path)
#pragma unroll
        for (int m = 0; m < 64; ++m) xf_step(m, rbuf, hn, vbuf);
    };

#ifdef W16_STAMP   // profiling build (tools/w16_stamps.py): cycles per phase, summed over the launch, written over the bias array: [workgroup][wave][8]
    uint32_t st_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};           // MFMA stream h = 0 | h = 1, DMA wait h = 0 | h = 1, barrier, epilogue, rest, slow path
    uint64_t st_t = __builtin_readcyclecounter();
#define W16_MARK(i) do { const uint64_t now_ = __builtin_readcyclecounter(); st_sum[i] += (uint32_t)(now_ - st_t); st_t = now_; } while (0)
    uint32_t* st_out = reinterpret_cast<uint32_t*>(const_cast<float*>(STACK ? layers[0].bias : bias));
#else
#define W16_MARK(i) do { } while (0)
#endif
    int L = 0;                                                 // (STACK) the layer the workgroup is on
    auto tile_done = [&](int t) -> Done {
        Done d;
        const int n = mdiv(t, mg_img, sh_img), rr_ = t - n * (tiles_x * tiles_y);
        const int by = mdiv(rr_, mg_tx, sh_tx), bx = rr_ - by * tiles_x;
        const int ox = OUT_COLS * bx + 2 * pl;
        const uint64_t ob = (uint64_t)(y + (int64_t)n * HW * 256);
        d.orsrc.x = (int)uniform((uint32_t)ob);
        d.orsrc.y = (int)uniform((uint32_t)(ob >> 32));
        d.orsrc.z = (int)uniform((uint32_t)(HW * 256));
        d.orsrc.w = 0x00020000;
        {
            const int e_in = STACK ? (in_amax ? sp16_act_exp(in_amax[(int64_t)L * range_stride + n]) : L == 0 ? in_exp : out_exp)
                                   : in_amax ? sp16_act_exp(in_amax[n]) : in_exp;
            const int e_out = STACK ? (in_amax ? sp16_act_exp(in_amax[(int64_t)(L + 1) * range_stride + n]) : out_exp)
                                    : out_amax ? sp16_act_exp(out_amax[n]) : out_exp;
            d.oscale = sp16_pow2(e_out - e_in - w_exp);
            d.bscale = sp16_pow2(e_out);
        }
        d.oy0 = OUT_ROWS * by + 2 * wave;
        d.pix[0] = ox >= W ? RAW_OOB : (uint32_t)((kb * (int)HW + ox) * 16);
        d.pix[1] = ox + 1 >= W ? RAW_OOB : (uint32_t)((kb * (int)HW + ox + 1) * 16);
        return d;
    };

    const char* Wnx = nullptr;                                 // (STACK) the next layer's weights
    bool s3_raw = false;                                       // (STACK) half-stages 5 and 7 may fetch the next tile's first two chunks (the tiles it reads are written)
    const char* s3_w = nullptr;                                // (STACK) ... and whose weights go with it (this layer's or the next one's)

    // ---- one half-stage hs = 2 c + h of the current tile: 72 MFMAs = 6 groups (xi', dy) of 12 (both cout groups x both rows x three
    // products), V[hs & 1] x Wt[hs & 1]; in the gap behind every MFMA one micro-step of the NEXT half-stage's transform, a weight
    // fragment of the next group, a DMA instruction of what comes after (h = 0: the next chunk's halo tile and this chunk's second
    // weight half; h = 1: the next chunk's first weight half).  `before_barrier` runs between the last MFMA and the barrier, `shadow(m)`
    // in gap m.
    auto half_stage = [&](int hs, bool wnext, bool rnext, auto&& before_barrier, auto&& shadow) __attribute__((always_inline)) {
        const int c = hs >> 1, h = hs & 1, vb_ = hs & 1, wb = hs & 1;
        const int hn = h ^ 1, rbn = (h ? (c + 1) : c) & 1;    // the successor's xi half and chunk buffer
        const lds_char* ab = abase + wb * W_HALF;
        // weight fragments of a group: the lo pieces (first pass only) in one set, the hi pieces in two (the next group's arrive while this
        // group's are multiplied)
        u32x4 Alo[2], Ahi[2][2];                               // [cout group], [group parity][cout group]
        auto loadA = [&](int g, int i) __attribute__((always_inline)) {      // fragment i = 2 piece + cg of group g
            const u32x4 v = *reinterpret_cast<const lds_u32x4*>(ab + ((g * 2 + (i >> 1)) * 2 + (i & 1)) * 1024);
            if (i >> 1) Alo[i & 1] = v; else Ahi[g & 1][i & 1] = v;
        };
        // DMA instruction k of the half-stage: the weights of half-stage hs + 1 first (wnext: there is one), then - h = 1 - the halo tile of
        // chunk c + 2 (c >= 2: of the NEXT tile's chunk c - 2; rnext: there is one and the tiles it reads are written), TWO half-stages
        // ahead of its transform: a half-stage is ~1.2 us, a round trip to memory under load more
        auto dma = [&](int k) __attribute__((always_inline)) {
            if (W16_ABL & 1) return;
            if (k < W_INSTR) { if (wnext) w_piece((STACK && hs == 7) ? s3_w : Wp, (hs + 1) & 7, wb ^ 1, k); }
            else if (h == 1 && k < W_INSTR + RAW_INSTR) { if (rnext) raw_piece((c + 2) & 3, c & 1, k - W_INSTR); }
        };
        W16_MARK(6);
#pragma unroll
        for (int i = 0; i < 4; ++i) loadA(0, i);
        __builtin_amdgcn_sched_barrier(0);
        const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const int xp = g / 3, dy = g % 3, xi = 2 * h + xp;
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int m = 12 * g + i, pass = i >> 2, r = (i >> 1) & 1, cg = i & 1;
                const bool z1st = c == 0 && dy == 0 && pass == 0;      // a tile's first MFMA into an accumulator: C = 0
                const h8 a = __builtin_bit_cast(h8, pass == 0 ? Alo[cg] : Ahi[g & 1][cg]);
                const h8 b = __builtin_bit_cast(h8, pass == 1 ? Vl[vb_][r + dy][xp] : Vh[vb_][r + dy][xp]);
                acc[xi][r][cg] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z1st ? zero16 : acc[xi][r][cg], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (h == 1 && m == 16 && rnext) voff_get(0);
                xf_step(m, rbn, hn, vb_ ^ 1);
                if (g < 5 && i >= 4 && i < 8) loadA(g + 1, (i - 4) ^ 2);   // (the lo pieces are free after the first pass: the next group's first, then its hi pieces)
                if (m % 3 == 1) dma(m / 3);
                shadow(m);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        W16_MARK(h);
        // the weights have to be there; the halo tile issued behind them in this half-stage has another half-stage to land
        if (h == 1 && rnext && !(W16_ABL & 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RAW_INSTR) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W16_MARK(2 + h);
        before_barrier();
        lds_barrier();
        W16_MARK(4);
    };
    auto nothing = [] {};
    auto no_shadow = [](int) {};

    char* const y_even = y;
    unsigned fbase = 0, fgiveup = 0;
    if (STACK) {
        const StackLayer l0 = layers[0];
        Wp = l0.w; bias = l0.bias; w_exp = l0.w_exp; relu = l0.relu;
        fbase = __hip_atomic_load(flags + (int64_t)t_first * STACK_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fgiveup = __hip_atomic_load(flags + (int64_t)n_tiles * STACK_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- prologue: bias, chunk 0 and the first weight half of the first tile, its transform
#ifdef W16_STAMP
    if (wave == 0) bias_s[lane] = 0.0f;                       // (the profiling build takes its stamp buffer through the bias pointer)
#else
    if (wave == 0) bias_s[lane] = bias ? bias[lane] : 0.0f;
#endif
    fetch_tile_uniform(t_first, x);
#pragma unroll
    for (int j = 0; j < RAW_INSTR; ++j) voff_set(j, fetch_lane_offset(j));
    voff_get(0);
#pragma unroll
    for (int j = 0; j < RAW_INSTR; ++j) raw_piece(0, 0, j);
#pragma unroll
    for (int j = 0; j < W_INSTR; ++j) w_piece(Wp, 0, 0, j);
    voff_get(0);
#pragma unroll
    for (int j = 0; j < RAW_INSTR; ++j) raw_piece(1, 1, j);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RAW_INSTR) : "memory");
    __syncthreads();
    t_all(0, 0, 0);

    auto flag_word = [&](int t) __attribute__((always_inline)) -> const unsigned* {
        const int tpi = tiles_x * tiles_y;
        const int n = mdiv(t, mg_img, sh_img), r = t - n * tpi, by = mdiv(r, mg_tx, sh_tx), bx = r - by * tiles_x;
        const int k = lane < 9 ? lane : 4;
        const int ny = by + k / 3 - 1, nx = bx + k % 3 - 1;
        const bool ok = ny >= 0 && ny < tiles_y && nx >= 0 && nx < tiles_x;
        return flags + (int64_t)(ok ? n * tpi + ny * tiles_x + nx : t) * STACK_FLAG_STRIDE;
    };
    int pend_t = -1;                                           // (STACK) a finished tile whose word is published behind the next barrier its stores are waited for at
    unsigned pend_v = 0;
#pragma unroll 1
    for (int t_cur = t_first;;) {
        const bool new_layer = STACK && !(t_cur + t_step < t_end);
        const int t_next = new_layer ? t_first : t_cur + t_step;
        const bool next = STACK ? (!new_layer || L + 1 < n_layers) : t_next < t_end;
        const int L_next = L + (new_layer ? 1 : 0);
        const bool poll = STACK && next && L_next > 0;
        if (STACK && new_layer && next) Wnx = layers[L + 1].w;
        half_stage(0, true, false, nothing, no_shadow);
        // (STACK) behind half-stage 0's wait and barrier every store of the tile before has been acknowledged: its word goes out
        half_stage(1, true, true, nothing, [&](int m) __attribute__((always_inline)) {
            if (STACK && m == 0 && pend_t >= 0 && wave == 2 && lane == 0)
                __hip_atomic_store(flags + (int64_t)pend_t * STACK_FLAG_STRIDE, pend_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        });
        pend_t = -1;
        half_stage(2, true, false, nothing, no_shadow);
        half_stage(3, true, true, nothing, no_shadow);
        // the NEXT tile's fetch descriptor and lane offsets (this tile's last halo DMA went out in half-stage 3), and (STACK) ONE poll of
        // the words of the tiles the next tile reads (wave 0: asked in gap 0, landed by the half-stage's wait, the verdict through LDS)
        half_stage(4, true, false, [&]() __attribute__((always_inline)) {
            if (STACK && wave == 0) {
                const unsigned pv = poll_s[lane];              // (landed: the half-stage's own vmcnt(0) is behind us)
                const unsigned target = fbase + (unsigned)L_next;
                const bool late = poll && lane < 9 && (int)(pv - target) < 0;
                const uint32_t ok = (!poll || fgiveup) ? 1u : (__builtin_amdgcn_ballot_w64(late) == 0 ? 1u : 0u);
                if (lane == 0) ready_s = ok;
            }
        }, [&](int m) __attribute__((always_inline)) {
            if (STACK && m == 0 && wave == 0 && poll) {
                const unsigned* pf = flag_word(t_next);
                const uint32_t m0v = uniform((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)poll_s);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off sc1" ::"s"(m0v), "v"(pf) : "m0", "memory");
            }
            if (m == 20 && next) fetch_tile_uniform(t_next, (STACK && new_layer) ? y : x);
            if (m >= 22 && m < 22 + 2 * RAW_INSTR && !(m & 1) && next) voff_set((m - 22) >> 1, fetch_lane_offset((m - 22) >> 1));
        });
        bool ready = true;
        if (STACK) {
            ready = uniform(ready_s) != 0;
            s3_raw = next && ready;
            s3_w = new_layer ? Wnx : Wp;
        }
        const bool rn = STACK ? s3_raw : next;                 // the next tile's first two halo chunks go out in half-stages 5 and 7
        half_stage(5, true, rn, nothing, no_shadow);
        half_stage(6, true, false, nothing, no_shadow);
        Done d;
        half_stage(7, next, rn, nothing, [&](int m) __attribute__((always_inline)) { if (m == 66) d = tile_done(t_cur); });

        // ---- epilogue: y[2t] = M0 + M1 + M2, y[2t + 1] = M1 - M2 - M3, x 2^(e_out - e_in - w_exp), + bias, ReLU (the NaN-propagating maximum),
        // stores.  acc[.][r][g][i] is cout 32 g + 8 (i >> 2) + 4 kb + (i & 3) of tile pl of row 2 wave + r.
        if (!(W16_ABL & 4) || relu == 77) {
            const float floor_ = relu ? 0.0f : -__builtin_inff();
            const __attribute__((address_space(3))) float* bsl = (const __attribute__((address_space(3))) float*)bias_s + (STACK ? 64 * (L & 1) : 0);
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const bool row_ok = d.oy0 + r < H;             // (uniform)
                const uint32_t rowoff = (uint32_t)((d.oy0 + r) * W) * 16u;
#pragma unroll
                for (int g = 0; g < 2; ++g) {
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {           // 16 couts 32 g + 16 gp .. : the 8-cout blocks b8 = 4 g + 2 gp, + 1 = one input chunk of the next layer
                        f32x4 o[2][2];                         // [pixel 2t, 2t + 1][block parity]: the lane's four consecutive couts 8 b8 + 4 kb .. + 4
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int gq = 2 * gp + q;
                            const f32x4 bz = *reinterpret_cast<const lds_f32x4*>(bsl + 32 * g + 8 * gq + 4 * kb) * d.bscale;
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                const int i = 4 * gq + k;
                                const float m0 = acc[0][r][g][i], m1 = acc[1][r][g][i], m2 = acc[2][r][g][i], m3 = acc[3][r][g][i];
                                const float y0 = (m0 + m1) + m2, y1 = (m1 - m2) - m3;
                                o[0][q][k] = __builtin_elementwise_maximum(__builtin_fmaf(y0, d.oscale, bz[k]), floor_);
                                o[1][q][k] = __builtin_elementwise_maximum(__builtin_fmaf(y1, d.oscale, bz[k]), floor_);
                            }
                        }
                        if (!row_ok) continue;
#define W16_STORE(VAL, PIX, SO)                                                                                                                  \
    do {                                                                                                                                         \
        if (STACK) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc1\n\ts_nop 1" ::"v"(VAL), "v"(PIX), "s"(d.orsrc), "s"(SO) : "memory"); \
        else asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" ::"v"(VAL), "v"(PIX), "s"(d.orsrc), "s"(SO) : "memory");       \
    } while (0)
                        if (FMT_P32) {
                            // plane 2 b8 + kb (the lane's kb rides in pix): the lane's four couts are the pixel's 16 bytes
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                const uint32_t so = uniform((uint32_t)(2 * (4 * g + 2 * gp + q)) * (uint32_t)HW * 16u + rowoff);
#pragma unroll
                                for (int px = 0; px < 2; ++px) W16_STORE(o[px][q], d.pix[px], so);
                            }
                        } else {
                            // sp16: split, and trade halves with the lane 32 away so that each lane holds one whole 16-byte pixel of plane
                            // (chunk 2 g + gp, hl, kb) - the even 8-cout block for lanes < 32, the odd one for lanes >= 32
                            const uint32_t so_h = uniform((uint32_t)((2 * g + gp) * 4 + 0) * (uint32_t)HW * 16u + rowoff);
                            const uint32_t so_l = uniform((uint32_t)((2 * g + gp) * 4 + 2) * (uint32_t)HW * 16u + rowoff);
#pragma unroll
                            for (int px = 0; px < 2; ++px) {
                                unsigned hi[4], lo[4];         // [block parity][pair]
#pragma unroll
                                for (int e = 0; e < 4; ++e) split_pair(o[px][e >> 1][2 * (e & 1)], o[px][e >> 1][2 * (e & 1) + 1], hi[e], lo[e]);
#pragma unroll
                                for (int e = 0; e < 2; ++e) {
                                    auto sh = __builtin_amdgcn_permlane32_swap(hi[e], hi[2 + e], false, false);
                                    hi[e] = sh[0]; hi[2 + e] = sh[1];
                                    auto sl = __builtin_amdgcn_permlane32_swap(lo[e], lo[2 + e], false, false);
                                    lo[e] = sl[0]; lo[2 + e] = sl[1];
                                }
                                const u32x4 oh = {hi[0], hi[1], hi[2], hi[3]}, ol = {lo[0], lo[1], lo[2], lo[3]};
                                W16_STORE(oh, d.pix[px], so_h);
                                W16_STORE(ol, d.pix[px], so_l);
                            }
                        }
#undef W16_STORE
                        __builtin_amdgcn_sched_barrier(0);     // (piece by piece: 32 accumulator reads in flight, not 256)
                    }
                }
            }
        }
        W16_MARK(5);
        if (STACK) {
            const unsigned done_v = fbase + (unsigned)(L + 1);
            if (new_layer && next) {
                const StackLayer ln = layers[L + 1];
                x = y;
                y = ((L + 1) & 1) ? y2 : y_even;
                Wp = ln.w; bias = ln.bias; w_exp = ln.w_exp; relu = ln.relu;
#ifndef W16_STAMP
                if (wave == 1) bias_s[64 * ((L + 1) & 1) + lane] = bias ? bias[lane] : 0.0f;
#endif
            }
            if (next && !ready) {
                // ---- the tiles the next tile reads are not all written (always so with ONE tile per workgroup): wait for the stores, publish,
                // wait for the nine words, fetch, transform
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (wave == 2 && lane == 0)
                    __hip_atomic_store(flags + (int64_t)t_cur * STACK_FLAG_STRIDE, done_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (wave == 0) {
                    const unsigned target = fbase + (unsigned)L_next;
                    const unsigned* f = flag_word(t_next);
                    if (!fgiveup) {
                        unsigned v[STACK_POLLS], spins = 0;
#pragma unroll
                        for (int q = 0; q < STACK_POLLS; ++q) {
                            asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[q]) : "v"(f) : "memory");
                            __builtin_amdgcn_s_sleep(3);
                        }
                        bool waiting = true;
                        while (waiting) {
#pragma unroll
                            for (int q = 0; q < STACK_POLLS; ++q) {
                                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v[q]) : "n"(STACK_POLLS - 1) : "memory");
                                if (__builtin_amdgcn_ballot_w64(lane < 9 && (int)(v[q] - target) < 0) == 0) { waiting = false; break; }
                                if (++spins > STACK_SPIN_LIMIT) {
                                    fgiveup = 1;
                                    if (lane == 0) __hip_atomic_fetch_or(flags + (int64_t)n_tiles * STACK_FLAG_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    waiting = false;
                                    break;
                                }
                                asm volatile("global_load_dword %0, %1, off sc1" : "=v"(v[q]) : "v"(f) : "memory");
                                __builtin_amdgcn_s_sleep(3);
                            }
                        }
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                }
                __syncthreads();
                voff_get(0);
#pragma unroll
                for (int j = 0; j < RAW_INSTR; ++j) raw_piece(0, 0, j);
                voff_get(0);
#pragma unroll
                for (int j = 0; j < RAW_INSTR; ++j) raw_piece(1, 1, j);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RAW_INSTR) : "memory");
                __syncthreads();
                t_all(0, 0, 0);
                W16_MARK(7);
            } else {
                pend_t = t_cur;
                pend_v = done_v;
            }
            if (new_layer) ++L;
        }
        if (!next) break;
        t_cur = t_next;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (STACK) {
        __syncthreads();
        if (pend_t >= 0 && threadIdx.x == 0)
            __hip_atomic_store(flags + (int64_t)pend_t * STACK_FLAG_STRIDE, pend_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef W16_STAMP
    W16_MARK(6);
    if (lane == 0)
        for (int i = 0; i < 8; ++i) st_out[((int)blockIdx.x * WAVES + wave) * 8 + i] = st_sum[i];
#endif
}

}  // namespace w16
}  // namespace deqsci

using namespace deqsci;

static void w16_magic(uint32_t d, uint32_t* mg, uint32_t* sh) {
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    *sh = 31 + s;
    *mg = (uint32_t)(((1ull << (31 + s)) + d - 1) / d);
}

static bool w16_bad_exp(int e) { return e < -SP16_EXP_LIMIT || e > SP16_EXP_LIMIT; }

static_assert(sizeof(w16::StackLayer) == 24, "the layer table of deqsci_conv3x3_c64_wino16_stack is three 8-byte words per layer");

extern "C" int deqsci_conv3x3_c64_wino16(const void* x, const void* u_packed, const float* bias, void* y, int64_t n, int64_t H, int64_t W,
                                         int relu, int w_exp, const float* in_amax, int in_exp, const float* out_amax, int out_exp, int fmt,
                                         deqsci_stream_t stream, void* start_event, void* stop_event) {
    if (!x || !u_packed || !y) return DEQSCI_ERR_NULL;
    if ((start_event == nullptr) != (stop_event == nullptr)) return DEQSCI_ERR_NULL;
    if (n <= 0 || H <= 0 || W <= 0) return DEQSCI_ERR_SHAPE;
    if (x == y || (fmt != DEQSCI_ACT_SP16 && fmt != DEQSCI_ACT_P32) || w16_bad_exp(w_exp) || w16_bad_exp(in_exp) || w16_bad_exp(out_exp)) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(x) || !aligned16(u_packed) || !aligned16(y)) return DEQSCI_ERR_ALIGN;
    const int64_t tiles_x = ceil_div(W, w16::OUT_COLS), tiles_y = ceil_div(H, w16::OUT_ROWS);
    const int64_t n_tiles = n * tiles_x * tiles_y;
    // 32-bit byte offsets inside one image, and the out-of-range sentinel 2^31 must lie beyond the descriptor's range
    if (n_tiles > (int64_t)INT32_MAX / 16 || H * W * 256 + 16 > (int64_t)w16::RAW_OOB) return DEQSCI_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t resident = (int64_t)num_cus();
    const dim3 grid((unsigned)(n_tiles < resident ? n_tiles : resident));
    uint32_t mg_img, sh_img, mg_tx, sh_tx;
    w16_magic((uint32_t)(tiles_x * tiles_y), &mg_img, &sh_img);
    w16_magic((uint32_t)tiles_x, &mg_tx, &sh_tx);
    hipEvent_t ev0 = static_cast<hipEvent_t>(start_event), ev1 = static_cast<hipEvent_t>(stop_event);
#define W16_LAUNCH(KERNEL)                                                                                                                  \
    hipExtLaunchKernelGGL(KERNEL, grid, dim3(w16::TBW), 0, st, ev0, ev1, 0, static_cast<const char*>(x), static_cast<const char*>(u_packed),    \
                          bias, static_cast<char*>(y), (int)H, (int)W, relu, w_exp, in_amax, in_exp, out_amax, out_exp, (int)tiles_x,           \
                          (int)tiles_y, (int)n_tiles, mg_img, sh_img, mg_tx, sh_tx, static_cast<char*>(nullptr),                                \
                          static_cast<const w16::StackLayer*>(nullptr), 1, static_cast<unsigned*>(nullptr), 0)
    if (fmt == DEQSCI_ACT_P32) W16_LAUNCH((w16::conv_w16_kernel<1, 0>));
    else W16_LAUNCH((w16::conv_w16_kernel<0, 0>));
#undef W16_LAUNCH
    return launch_status();
}

extern "C" int deqsci_conv3x3_c64_wino16_stack(const void* x, void* y_even, void* y_odd, const void* layers, int n_layers,
                                               int64_t n, int64_t H, int64_t W, const float* ranges, int64_t range_stride, int in_exp, int out_exp,
                                               int fmt, void* flags, deqsci_stream_t stream, void* start_event, void* stop_event) {
    if (!x || !y_even || !layers || !flags || (n_layers > 1 && !y_odd)) return DEQSCI_ERR_NULL;
    if ((start_event == nullptr) != (stop_event == nullptr)) return DEQSCI_ERR_NULL;
    if (n <= 0 || H <= 0 || W <= 0 || n_layers <= 0 || (ranges && (range_stride < n || range_stride > INT32_MAX))) return DEQSCI_ERR_SHAPE;
    if (x == y_even || x == y_odd || y_even == y_odd || n_layers > 64 || (fmt != DEQSCI_ACT_SP16 && fmt != DEQSCI_ACT_P32) || w16_bad_exp(in_exp) || w16_bad_exp(out_exp))
        return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(x) || !aligned16(y_even) || !aligned16(y_odd) || (reinterpret_cast<uintptr_t>(layers) & 7u) || (reinterpret_cast<uintptr_t>(flags) & 3u))
        return DEQSCI_ERR_ALIGN;
    const int64_t tiles_x = ceil_div(W, w16::OUT_COLS), tiles_y = ceil_div(H, w16::OUT_ROWS);
    const int64_t n_tiles = n * tiles_x * tiles_y;
    if (H * W * 256 + 16 > (int64_t)w16::RAW_OOB) return DEQSCI_ERR_UNSUPPORTED;
    // every workgroup of the launch has to be RESIDENT (they wait for one another): one per CU - the kernel's 137 KB of LDS and 512
    // registers per lane admit no second one - so never more workgroups than CUs; each walks its tiles layer after layer
    if (n_tiles > (int64_t)INT32_MAX / (16 * 32)) return DEQSCI_ERR_UNSUPPORTED;
    const int64_t resident = (int64_t)num_cus();
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint32_t mg_img, sh_img, mg_tx, sh_tx;
    w16_magic((uint32_t)(tiles_x * tiles_y), &mg_img, &sh_img);
    w16_magic((uint32_t)tiles_x, &mg_tx, &sh_tx);
    hipEvent_t ev0 = static_cast<hipEvent_t>(start_event), ev1 = static_cast<hipEvent_t>(stop_event);
#define W16_LAUNCH(KERNEL)                                                                                                                       \
    hipExtLaunchKernelGGL(KERNEL, dim3((unsigned)(n_tiles < resident ? n_tiles : resident)), dim3(w16::TBW), 0, st, ev0, ev1, 0,                     \
                          static_cast<const char*>(x), static_cast<const char*>(nullptr), static_cast<const float*>(nullptr), static_cast<char*>(y_even), \
                          (int)H, (int)W, 0, 0, ranges, in_exp, static_cast<const float*>(nullptr), out_exp, (int)tiles_x, (int)tiles_y, (int)n_tiles,   \
                          mg_img, sh_img, mg_tx, sh_tx, static_cast<char*>(y_odd), static_cast<const w16::StackLayer*>(layers), n_layers,                \
                          static_cast<unsigned*>(flags), (int)range_stride)
    if (fmt == DEQSCI_ACT_P32) W16_LAUNCH((w16::conv_w16_kernel<1, 1>));
    else W16_LAUNCH((w16::conv_w16_kernel<0, 1>));
#undef W16_LAUNCH
    return launch_status();
}
