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
// nested form holds 4 accumulators per 2 outputs, transforms ROWS, and its operand traffic is the direct kernel's.
//
// Two forms were built (profiles/r05_w16_form1_*; tools/ubench/variants/conv_w16_form1_4waves_variant.hip).  Form 1 - FOUR waves of 512
// registers, a wave = two output rows x 64 couts, each halo row transformed once per wave - was instruction-issue bound: a wave alone on
// its SIMD issues one instruction every 5-8 cycles (tools/ubench/mfma_f16_fillers.hip: v_sub_f32 5, v_cvt_pk_f16_f32 / v_fma_mix* 8,
// ds_read_b128 ~8, an LDS-DMA instruction 60+), its ~550 instructions per 72 MFMAs took 4100-4300 cycles against the MFMAs' 2304, and the
// epilogue's 32 stores per lane (store issue: ~7 bytes per cycle and CU) had no MFMA to hide behind: 210 us per layer at 64 images against
// the direct kernel's 173.  TWO waves on a SIMD issue nearly twice the vector-ALU instructions per MFMA slot (same microbenchmark: 8
// transform-type instructions per MFMA at 34 cycles per MFMA against 63 for one wave).  So, form 2, this file:
//
// Geometry.  Block tile = 8 rows x 64 columns of output pixels x 64 couts, one persistent 8-wave workgroup per CU, two waves per SIMD, 256
// registers each: 128 accumulators (4 xi x 2 cout groups x 16).  Wave w owns output row w - halo rows w, w + 1, w + 2, each transformed by
// the wave itself in its MFMA lanes' registers (1.5 x the transform work of form 1, on twice the issue slots) -, all 64 couts; MFMA N (32
// lanes) = the 32 Winograd tiles of the row, lane (t, kb) holds channels 8 kb .. + 8 of tile t.  Input channels in chunks of 16 (K of one
// MFMA), each chunk in TWO half-stages (xi = 0, 1 | xi = 2, 3): a half-stage is three groups (dy) of 12 MFMAs - V of one halo row (two
// positions, hi + lo: 16 registers, a ring of two rows) times the group's eight weight fragments ([xi'][dy][piece][cout group], 1 KB each,
// host-packed, LDS-DMA, 24 KB per half-stage, double-buffered) - while the next row's transform runs in the gaps: 6 LDS reads (3 pixels),
// 16 subtractions, 8 packed converts + 16 v_fma_mixlo/hi (the hi + lo split).  The 10 x 66 pixel halo tile of a chunk (4 planes, 42 KB) is
// double-buffered by chunk and fetched TWO half-stages ahead; its LDS order separates even and odd columns, so that the lanes of a wave
// read consecutive 16-byte slots (conflict-free) for every pixel of their tiles.  One barrier per half-stage.
//
// Activations: "p32" - the 16 planes of 16-byte pixels of sp16 (csrc/conv_s16.hip) holding 2^e x as fp32 instead of hi + lo fp16:
// [n][8 blocks of 8 channels][2 halves][H][ceil(W/64)][2 column parities][32][4 floats], plane 2 b8 + j = channels 8 b8 + 4 j .. + 4, and inside
// every block of 64 columns the 32 even columns first, then the 32 odd ones: a lane of this kernel owns the two pixels of a Winograd tile
// (columns 2t, 2t + 1) and stores 16 bytes at a time, so with the columns in natural order every store (and every halo-tile load) touched
// half of each 128-byte line and every line was written twice - with the parities apart a wave's store is 512 contiguous bytes per
// 8-channel block.  e follows the image's range slot exactly as for sp16 (common.hpp).  Unsplit, because the transform wants fp32 pixels (joining hi + lo costs 24 v_fma_mix per halo row and
// half-stage: 40 % of the transform's issue time in form 1) and the epilogue has nothing to split; FFDNet's first and last layer have p32
// forms (conv_s16.hip: head_s16_kernel / tail_s16_kernel with P32 = 1).
#include "common.hpp"
#include <hip/hip_ext.h>
#include <type_traits>
#pragma clang diagnostic ignored "-Winline-asm"

#ifndef W16_PHASES
#define W16_PHASES 1        // (A/B) > 1: the workgroups start in this many phases spread over W16_SPREAD shader cycles (see the kernel's prologue)
#endif
#ifndef W16_SPREAD
#define W16_SPREAD 40000    // ~ one tile
#endif
#ifndef W16_PRIO
#define W16_PRIO 0          // (A/B) issue priority of the two waves of a SIMD (w, w + 4): 0 = left to the arbiter (the older wave, 0-3, wins), 1 = waves 4-7
                            // at s_setprio 1 throughout, 2 = the winner alternates half-stage by half-stage, 3 = group by group
#endif
#ifndef W16_DMA4
#define W16_DMA4 0          // (A/B) 1: waves 0-3 - the older wave of every SIMD, which the arbiter serves first and which then waits ~1300 cycles per
                            // half-stage at the barrier for its partner - issue ALL the LDS-DMA instructions (two per slot), waves 4-7 none
#endif
#ifndef W16_EPI_PK
#define W16_EPI_PK 1        // (A/B) 0: round 5 - the epilogue's additions and multiply-adds one cout at a time
#endif
#ifndef W16_STORE_NOW
#define W16_STORE_NOW 1     // (A/B) 0: round 5 - a tile's outputs always wait for the end of its epilogue
#endif
#ifndef W16_XF_PK
#define W16_XF_PK 0         // (A/B) 1: the input transform's subtractions as packed fp32 (two channels per instruction, the same bits)
#endif
#ifndef W16_ABL
#define W16_ABL 0     // timing ablations only (results wrong): 1 = no DMA inside the half-stages, 2 = no transform, 4 = no epilogue, 8 = epilogue without its stores,
                      // 16 = what-if: the transform shared through LDS, 32 = what-if: the instruction mix of 4 x 64 tiles with one cout group per wave (see xf_step)
#endif

namespace deqsci {
namespace w16 {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) u32x4 lds_u32x4;
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

constexpr int WAVES = 8, TBW = 64 * WAVES;
constexpr int OUT_ROWS = 8, OUT_COLS = 64, RAW_ROWS = 10, RAW_COLS = 66, RAW_HALF = 33, RAW_PIX = RAW_ROWS * RAW_COLS;   // 660
constexpr int PLANE_B = RAW_PIX * 16;                          // 10560 bytes of one staged plane
constexpr int RAW_SLOTS = 4 * RAW_PIX;                         // 2640 units of 16 bytes per chunk tile
constexpr int RAW_TOTAL = 42;                                  // LDS-DMA instructions of 64 units per chunk tile (2688 slots): instruction 8 j + w is wave w's j-th
constexpr int RAW_INSTR = 6;                                   // ... per wave (the sixth only for waves 0 and 1)
constexpr int RAW_BUF = RAW_TOTAL * 1024;                      // 43008 bytes
constexpr int W_FRAGS = 2 * 3 * 2 * 2;                         // fragments of a half-stage: [xi' (2)][dy (3)][piece: hi, lo (2)][cout group (2)]
constexpr int W_HALF = W_FRAGS * 1024;                         // 24576 bytes
constexpr int W_INSTR = W_FRAGS / WAVES;                       // 3 per wave and half-stage
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
// (single v_sub / v_add by inline asm: left to itself hipcc pairs them into v_pk_add_f32, 19 cycles beside an MFMA against 2 x 5)
__device__ __forceinline__ float sub1(float a, float b) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float add1(float a, float b) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// packed fp32 (two values per instruction, each lane-wise IEEE operation the scalar one): spelled out, because hipcc packs the additions of a
// float2 expression but scalarises its subtractions
__device__ __forceinline__ f32x2 pk_add(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { f32x2 r; asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { f32x2 r; asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

struct StackLayer { const char* w; const float* bias; int w_exp; int relu; };
constexpr unsigned STACK_SPIN_LIMIT = 1u << 21;                // polls, one every ~0.1 us: a wait gives up after a quarter of a second
constexpr int STACK_POLLS = 8;                                 // polls of the neighbours' words in flight
constexpr int STACK_FLAG_STRIDE = 32;                          // words between two tiles' progress words: a 128-byte line each

// STACK: a run of n_layers layers in one launch with dataflow synchronisation between the tiles (the protocol of conv_s16_kernel<0, 0, 1>:
// progress words, write-through stores, agent-scope DMA loads; see there).
template <int STACK>
__global__ __launch_bounds__(TBW, 2) void conv_w16_kernel(const char* __restrict__ x, const char* __restrict__ Wp, const float* __restrict__ bias,
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
    __shared__ __attribute__((aligned(16))) char abl_v[(W16_ABL & 48) ? 4096 : 16];      // (W16_ABL & 16: the what-if's dummy V tile)
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
    const int Wq = tiles_x * OUT_COLS;                         // row pitch of a p32 plane: whole 64-column blocks ([32 even columns][32 odd columns] each)
    const int64_t HW = (int64_t)H * Wq;                        // pixels of a plane
    const int pl = lane & 31, kb = lane >> 5;
    const bool six = wave < RAW_TOTAL - 8 * (RAW_INSTR - 1);   // (uniform) this wave issues a sixth halo-tile instruction
    const bool mover = !W16_DMA4 || wave < 4;                  // (uniform) this wave issues LDS-DMA instructions at all
    // "everything but the halo tile issued last has landed": the instructions of one halo tile that may stay in flight (W16_DMA4: waves 0 and
    // 1 issue 11 of the 42, waves 2 and 3 ten; the others have only stores in flight and keep the old count)
    auto wait_all_but_a_tile = [&]() __attribute__((always_inline)) {
        if (W16_DMA4 && wave < 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * RAW_INSTR - 1) : "memory");
        else if (W16_DMA4 && wave < 4) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * RAW_INSTR - 2) : "memory");
        else if (six) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RAW_INSTR) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(RAW_INSTR - 1) : "memory");
    };

    // ---- halo tile by LDS-DMA: slot s = 64 (8 j + wave) + lane of the chunk tile is plane p = s / 660, row (s % 660) / 66, and inside the
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
    auto fetch_lane_offset = [&](int j) __attribute__((always_inline)) -> uint32_t {       // (W16_DMA4: j = 2 x slot + which of its two instructions)
        int w_ = wave;
        asm volatile("" : "+s"(w_));                           // (recomputed at every use: once per tile and instruction)
        const int s = W16_DMA4 ? 64 * (8 * (j >> 1) + w_ + 4 * (j & 1)) + lane : 64 * (8 * j + w_) + lane;
        const int p = (s * 6356) >> 22;                        // s / 660 for s < 2816
        const int q = s - p * RAW_PIX;
        const int row = (q * 993) >> 16, rem = q - row * RAW_COLS;           // q / 66 for q < 660
        const int par = rem >= RAW_HALF ? 1 : 0, col = 2 * (rem - par * RAW_HALF) + par;
        const int iy = ft_py0 + row, ix = ft_px0 + col;
        const bool ok = s < RAW_SLOTS && (uint32_t)iy < (uint32_t)H && (uint32_t)ix < (uint32_t)W;
        uint32_t off = ((uint32_t)p * (uint32_t)HW + (uint32_t)(iy * Wq + (ix & ~63) + ((ix & 1) << 5) + ((ix & 63) >> 1))) * 16u;
        asm volatile("" : "+v"(off));
        return ok ? off : RAW_OOB;
    };
    // the lane offsets of the tile whose chunks are being fetched live in LDS (registers the accumulators need): written once per tile,
    // read one DMA instruction ahead
    uint32_t vo_next = 0, vo_next1 = 0;
    auto voff_set = [&](int j, uint32_t v) __attribute__((always_inline)) {
        if (W16_DMA4) voff_s[j * (TBW / 2) + (int)(threadIdx.x & (TBW / 2 - 1))] = v;
        else voff_s[j * TBW + (int)threadIdx.x] = v;
    };
    auto voff_get = [&](int j) __attribute__((always_inline)) {
        if (W16_DMA4) {
            vo_next = voff_s[(2 * j) * (TBW / 2) + (int)(threadIdx.x & (TBW / 2 - 1))];
            vo_next1 = voff_s[(2 * j + 1) * (TBW / 2) + (int)(threadIdx.x & (TBW / 2 - 1))];
        } else vo_next = voff_s[j * TBW + (int)threadIdx.x];
    };
    const uint32_t raw_lds = (uint32_t)(uintptr_t)(lds_char*)Raw, wt_lds = (uint32_t)(uintptr_t)(lds_char*)Wt;
    auto raw_piece = [&](int c, int buf, int j) __attribute__((always_inline)) {
        int w_ = wave;
        asm volatile("" : "+s"(w_));
        const uint32_t soff = uniform((uint32_t)c * (uint32_t)HW * 64u);                               // 4 planes of 16 HW bytes per chunk
        if (W16_DMA4) {                                        // instructions 8 j + w and 8 j + w + 4, waves 0-3 only
            const uint32_t vo[2] = {vo_next, vo_next1};
            if (j + 1 < RAW_INSTR) voff_get(j + 1);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int I = 8 * j + w_ + 4 * p;
                if (I >= RAW_TOTAL) continue;                  // (uniform)
                const uint32_t m0p = uniform(raw_lds + (uint32_t)(buf * RAW_BUF + I * 1024));
                if (STACK) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds" ::"s"(m0p), "v"(vo[p]), "s"(rsrc), "s"(soff) : "m0");
                else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0p), "v"(vo[p]), "s"(rsrc), "s"(soff) : "m0");
            }
            return;
        }
        const uint32_t m0v = uniform(raw_lds + (uint32_t)(buf * RAW_BUF + (8 * j + w_) * 1024));
        const uint32_t voj = vo_next;                          // (voff_get(j) ran a gap ago)
        if (j + 1 < RAW_INSTR) voff_get(j + 1);
        if (j == RAW_INSTR - 1 && !six) return;                // (uniform)
        if (STACK) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen sc1 lds" ::"s"(m0v), "v"(voj), "s"(rsrc), "s"(soff) : "m0");
        else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(voj), "s"(rsrc), "s"(soff) : "m0");
    };
    // ---- weights of half-stage hs = 2 c + h: 24 fragments of 1 KiB, host-packed in LDS order; wave w moves fragments 3 w .. 3 w + 2
    auto w_piece = [&](const char* Wl, int hs, int buf, int j) __attribute__((always_inline)) {
        int w_ = wave;
        asm volatile("" : "+s"(w_));
        const uint32_t off = (uint32_t)(((W16_DMA4 ? 2 * W_INSTR : W_INSTR) * w_ + j) * 1024);        // (W16_DMA4: j = 0 .. 5, waves 0-3)
        const uint64_t g = (uint64_t)(Wl + (int64_t)hs * W_HALF) + off;
        const uint32_t m0v = uniform(wt_lds + (uint32_t)(buf * W_HALF) + off);
        const uint64_t gs = ((uint64_t)uniform((uint32_t)(g >> 32)) << 32) | uniform((uint32_t)g);
        const uint32_t lv = (uint32_t)lane * 16u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(lv), "s"(gs) : "m0");
    };

    // acc[xi][g]: ONE accumulation chain per position (36 MFMAs: the cross products of a (dy, chunk) step first, then hi x hi)
    f32x16 acc[4][2];
    // V of a halo row: [ring slot][xi' (2)] as the MFMA B operand of lane (tile, kb): hi and lo pieces
    u32x4 Vh[2][2], Vl[2][2];
    struct Done { i32x4 orsrc; uint32_t pix[2]; int oy; float oscale, bscale; };

    // operand addresses: the lane's first halo row / tile slot in the first half plane of its k block; weight fragments lane-linear
    // (opaque to the compiler: folded into the arrays' absolute LDS addresses, every offset beyond 64 KB becomes an address register of its own)
    uint32_t lb0 = (uint32_t)(uintptr_t)(lds_char*)Raw + (uint32_t)(2 * kb * PLANE_B + (wave * RAW_COLS + pl) * 16);
    uint32_t lb1 = lb0 + RAW_BUF, ab0 = (uint32_t)(uintptr_t)(lds_char*)Wt + (uint32_t)lane * 16u;
    asm volatile("" : "+v"(lb0), "+v"(lb1), "+v"(ab0));
    const lds_char* const lbase[2] = {(const lds_char*)(uintptr_t)lb0, (const lds_char*)(uintptr_t)lb1};
    const lds_char* abase = (const lds_char*)(uintptr_t)ab0;

    // ---- the input transform of ONE halo row rho (0..2 of the wave's three) for the xi half hh from chunk buffer rb into ring slot ns, in two
    // halves f (channels 4 f .. + 4 of the lane's eight: a pixel's f-th 16 bytes) of five micro-steps, placed behind the MFMAs of the group
    // before: raw[q] = pixel q + hh of the lane's tile,
    // hh = 0: V0 = d0 - d2, V1 = d1 + d2;  hh = 1 (d0..2 = pixels 1..3): V2 = d1 - d0, V3 = d0 - d2
    u32x4 raw[3];
    float va[4], vb[4];
    auto t_load = [&](int rb, int rho, int hh, int f) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const int jj = q + hh;
            raw[q] = *reinterpret_cast<const lds_u32x4*>(lbase[rb] + (rho * RAW_COLS + (jj & 1) * RAW_HALF + (jj >> 1)) * 16 + f * PLANE_B);
        }
    };
    auto px = [&](int q, int k) __attribute__((always_inline)) -> float {
        const unsigned u = raw[q][k];                          // (by value: __builtin_bit_cast of a vector ELEMENT reads element 0)
        return __builtin_bit_cast(float, u);
    };
    auto t_xf = [&](int s, int hh) __attribute__((always_inline)) {  // s = 0, 1: channels 2 s, 2 s + 1 of the half
#if W16_XF_PK
        {
            const f32x2 d0 = {px(0, 2 * s), px(0, 2 * s + 1)}, d1 = {px(1, 2 * s), px(1, 2 * s + 1)}, d2 = {px(2, 2 * s), px(2, 2 * s + 1)};
            const f32x2 a = hh == 0 ? pk_sub(d0, d2) : pk_sub(d1, d0), b = hh == 0 ? pk_add(d1, d2) : pk_sub(d0, d2);
            va[2 * s] = a[0]; va[2 * s + 1] = a[1]; vb[2 * s] = b[0]; vb[2 * s + 1] = b[1];
            return;
        }
#endif
#pragma unroll
        for (int k = 2 * s; k < 2 * s + 2; ++k) {
            if (hh == 0) { va[k] = sub1(px(0, k), px(2, k)); vb[k] = add1(px(1, k), px(2, k)); }
            else { va[k] = sub1(px(1, k), px(0, k)); vb[k] = sub1(px(0, k), px(2, k)); }
        }
    };
    auto t_split = [&](int s, int ns, int f) __attribute__((always_inline)) {   // s = 0, 1: channel pair s of the half, both positions
        unsigned hi, lo;
        split_pair(va[2 * s], va[2 * s + 1], hi, lo);
        Vh[ns][0][2 * f + s] = hi; Vl[ns][0][2 * f + s] = lo;
        split_pair(vb[2 * s], vb[2 * s + 1], hi, lo);
        Vh[ns][1][2 * f + s] = hi; Vl[ns][1][2 * f + s] = lo;
    };
    // the micro-step of gap i (0 .. 11) of a group:  [load f0][-][-][-][xf][xf][split + load f1][split][xf][xf][split][split]
    auto xf_step = [&](int i, int rb, int rho, int hh, int ns) __attribute__((always_inline)) {
        if (W16_ABL & 2) return;
        if (W16_ABL & 32) {
            // WHAT-IF (results wrong): the instruction mix of block tiles of 4 x 64 pixels with a wave = one row x ONE cout group (the geometry whose
            // shared V fits the LDS), per 36 MFMAs: two row transforms instead of three (78 vector instructions, 12 raw
            // reads), 6 V writes and 24 V reads from the dummy tile beside the 24 weight-fragment reads
            __attribute__((address_space(3))) u32x4* dv = (__attribute__((address_space(3))) u32x4*)abl_v + lane;
            if (rho == 2) {
                if (i < 8) { u32x4 t_ = dv[64 * (i & 3)]; asm volatile("" :: "v"(t_)); }
                if (i == 0) { Vh[ns][0] = dv[0]; Vl[ns][0] = dv[64]; Vh[ns][1] = dv[128]; Vl[ns][1] = dv[192]; }
                return;
            }
            if (i < 8) { u32x4 t_ = dv[64 * (i & 3)]; asm volatile("" :: "v"(t_)); }      // (16 more V reads over the two transformed rows)
            if (i == 11) {
                t_split(1, ns, 1);
                dv[0] = Vh[ns][0]; dv[64] = Vl[ns][0]; dv[128] = Vh[ns][1];
                return;
            }
        }
        if (W16_ABL & 16) {
            // WHAT-IF (results wrong): every halo row transformed ONCE per workgroup and shared through LDS - a wave transforms 1.25 rows per
            // half-stage instead of 3 (its middle row; the first row of its successor only in waves 0 and 1) and WRITES their V (4 x 16 bytes per
            // lane) to LDS; the rows it skips it READS there (4 x 16 bytes).  The LDS the real thing needs (80 KB of V) does not exist beside
            // the raw tiles: all waves use one 4 KB dummy tile, so this prices the instruction streams only
            const bool mine = rho == 1 || (rho == 0 && wave < 2);
            __attribute__((address_space(3))) u32x4* dv = (__attribute__((address_space(3))) u32x4*)abl_v + lane;
            if (!mine) {
                if (i == 0) { Vh[ns][0] = dv[0]; Vl[ns][0] = dv[64]; Vh[ns][1] = dv[128]; Vl[ns][1] = dv[192]; }
                return;
            }
            if (i == 11) {
                t_split(1, ns, 1);
                dv[0] = Vh[ns][0]; dv[64] = Vl[ns][0]; dv[128] = Vh[ns][1]; dv[192] = Vl[ns][1];
                return;
            }
        }
        if (i == 0) t_load(rb, rho, hh, 0);
        else if (i == 4 || i == 5) t_xf(i - 4, hh);
        else if (i == 6) { t_split(0, ns, 0); t_load(rb, rho, hh, 1); }       // (the half's pixels are dead once its V is formed)
        else if (i == 7) t_split(1, ns, 0);
        else if (i == 8 || i == 9) t_xf(i - 8, hh);
        else if (i == 10 || i == 11) t_split(i - 10, ns, 1);
    };
    auto t_row = [&](int rb, int rho, int hh, int ns) __attribute__((always_inline)) {   // a whole row at once (prologue, slow path)
#pragma unroll
        for (int i = 0; i < 12; ++i) xf_step(i, rb, rho, hh, ns);
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
        const int ox = OUT_COLS * bx + 2 * pl;                 // pixel p of the lane's tile is column ox + p = slot 32 p + pl of block bx
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
        d.oy = OUT_ROWS * by + wave;
        d.pix[0] = ox >= W ? RAW_OOB : (uint32_t)((kb * (int)HW + OUT_COLS * bx + pl) * 16);
        d.pix[1] = ox + 1 >= W ? RAW_OOB : (uint32_t)((kb * (int)HW + OUT_COLS * bx + 32 + pl) * 16);
        return d;
    };

    const char* Wnx = nullptr;                                 // (STACK) the next layer's weights
    bool s3_raw = false;                                       // (STACK) half-stages 5 and 7 may fetch the next tile's first two chunks (the tiles it reads are written)
    const char* s3_w = nullptr;                                // (STACK) ... and whose weights go with its first half-stage (this layer's or the next one's)

    // ---- one half-stage hs = 2 c + h of the current tile: 36 MFMAs = 3 groups (dy) of 12 (both positions x both cout groups x three
    // products): halo row dy's V (ring slot (3 hs + dy) & 1) x the group's eight weight fragments (Alo / Ahi, in registers since the group
    // before); in the gap behind every MFMA a micro-step of the NEXT row's transform (dy + 1 of this half-stage, or row 0 of the successor's:
    // the other xi half of the chunk, or the first of the next chunk) and a weight fragment of the next group.  The half-stage's BARRIER sits
    // between its groups 1 and 2: behind it the halo chunk the first two groups transformed from and the weight half the groups read are
    // dead, and what was fetched for the successor is visible - so group 2 multiplies on (its operands are in registers) while it fetches the
    // successor's first fragments, issues the DMA of what comes after (weights of half-stage hs + 2 into this one's buffer; h = 1: the halo
    // tile of chunk c + 2 - c >= 2: of the NEXT tile's chunk c - 2 - into this chunk's: TWO half-stages ahead of its transform) and
    // transforms the successor's first row.  No MFMA waits behind the barrier.  `raw_prev`: the half-stage before issued a halo tile behind
    // its weights (h = 0 only: it may still be in flight at the barrier); `before_barrier` runs in front of the barrier, `shadow(m)` in gap m.
    u32x4 Alo[2][2], Ahi[2][2];                                // weight fragments of the current group: [xi'][cout group], lo pieces (first pass only) / hi pieces
    auto loadA = [&](int wbuf, int dy, int piece, int xp, int cg) __attribute__((always_inline)) {
        const u32x4 v = *reinterpret_cast<const lds_u32x4*>(abase + wbuf * W_HALF + ((((xp * 3 + dy) * 2 + piece) * 2 + cg) * 1024));
        if (piece) Alo[xp][cg] = v; else Ahi[xp][cg] = v;
    };
    auto half_stage = [&](int hs, bool wnext, bool rnext, bool raw_prev, auto&& before_barrier, auto&& shadow) __attribute__((always_inline)) {
        const int c = hs >> 1, h = hs & 1, wb = hs & 1;
        const int hn = h ^ 1, rbn = (h ? (c + 1) : c) & 1;    // the successor's xi half and chunk buffer
        // DMA instruction k of the half-stage (group 2 only): the weights of half-stage hs + 2 (wnext: there is one; the next tile's come from
        // s3_w, its layer's), then - h = 1 - the halo tile (rnext: there is one and the tiles it reads are written)
        auto dma = [&](int k) __attribute__((always_inline)) {
            if ((W16_ABL & 1) || !mover) return;
            if (k < W_INSTR) {
                if (wnext) {
                    if (W16_DMA4) { w_piece((STACK && hs >= 6) ? s3_w : Wp, (hs + 2) & 7, wb, 2 * k); w_piece((STACK && hs >= 6) ? s3_w : Wp, (hs + 2) & 7, wb, 2 * k + 1); }
                    else w_piece((STACK && hs >= 6) ? s3_w : Wp, (hs + 2) & 7, wb, k);
                }
            } else if (h == 1 && k < W_INSTR + RAW_INSTR) { if (rnext) raw_piece((c + 2) & 3, c & 1, k - W_INSTR); }
        };
        const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (W16_PRIO == 2) { if (((wave >> 2) ^ hs) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int slot = (3 * hs + dy) & 1;
            if (W16_PRIO == 3) { if (((wave >> 2) ^ (3 * hs + dy)) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
            if (dy == 2) {
                W16_MARK(h);
                // what the successor reads has to be there: its weights (issued a half-stage ago) and - h = 1 - its halo chunk (two ago); a halo
                // tile issued behind the weights in the half-stage before has another half-stage to land
                if (h == 0 && raw_prev && !(W16_ABL & 1)) wait_all_but_a_tile();
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                W16_MARK(2 + h);
                before_barrier();
                lds_barrier();
                W16_MARK(4);
            }
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const int m = 12 * dy + i, pass = i >> 2, xp = (i >> 1) & 1, cg = i & 1, xi = 2 * h + xp;
                const bool z1st = c == 0 && dy == 0 && pass == 0;      // a tile's first MFMA into an accumulator: C = 0
                const h8 a = __builtin_bit_cast(h8, pass == 0 ? Alo[xp][cg] : Ahi[xp][cg]);
                const h8 b = __builtin_bit_cast(h8, pass == 1 ? Vl[slot][xp] : Vh[slot][xp]);
                acc[xi][cg] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, z1st ? zero16 : acc[xi][cg], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (dy < 2) xf_step(i, c & 1, dy + 1, h, slot ^ 1);
                else xf_step(i, rbn, 0, hn, slot ^ 1);
                // the next group's fragments: the lo pieces are free after the first pass, each hi piece behind the last MFMA that reads it
                if (i >= 4 && i < 8) loadA(dy < 2 ? wb : wb ^ 1, dy < 2 ? dy + 1 : 0, 1, ((i - 4) >> 1) & 1, (i - 4) & 1);
                if (i >= 8) loadA(dy < 2 ? wb : wb ^ 1, dy < 2 ? dy + 1 : 0, 0, ((i - 8) >> 1) & 1, (i - 8) & 1);
                if (dy == 2) {
                    if (h == 1 && i == 0 && rnext && mover) voff_get(0);
                    if (i >= 1 && i <= 3) dma(i - 1);                  // weights
                    if (i >= 4 && i <= 9) dma(i - 1);                  // halo tile (h = 1)
                }
                shadow(m);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
    auto nothing = [] {};
    auto no_shadow = [](int) {};
    // what a tile starts from when nothing was fetched ahead (prologue, slow path): its first two halo chunks (2 x (6 | 5) instructions), the
    // first row's transform, the first group's weight fragments
    auto fetch_two_chunks_and_wait = [&]() __attribute__((always_inline)) {
        if (mover) {
            voff_get(0);
#pragma unroll
            for (int j = 0; j < RAW_INSTR; ++j) raw_piece(0, 0, j);
            voff_get(0);
#pragma unroll
            for (int j = 0; j < RAW_INSTR; ++j) raw_piece(1, 1, j);
        }
        wait_all_but_a_tile();
    };
    auto first_row_and_fragments = [&]() __attribute__((always_inline)) {
        t_row(0, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) loadA(0, 0, i >> 2, (i >> 1) & 1, i & 1);
    };

    char* const y_even = y;
    unsigned fbase = 0, fgiveup = 0;
    if (STACK) {
        const StackLayer l0 = layers[0];
        Wp = l0.w; bias = l0.bias; w_exp = l0.w_exp; relu = l0.relu;
        fbase = __hip_atomic_load(flags + (int64_t)t_first * STACK_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fgiveup = __hip_atomic_load(flags + (int64_t)n_tiles * STACK_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- (A/B, off) de-phasing: every workgroup does the same work from the same start and stores its tile at the same moment.  Starting them in
    // phases took 18 % off a launch while the kernel's stores were half-line writes (116 -> 95 us at 32 images); with the column parities
    // apart (full 128-byte lines) a tile's stores are 6 % of a launch and the phases change nothing (profiles/r05_w16_form2_phases*.txt).
    if (W16_PHASES > 1 && t_first + t_step < t_end) {
        const int phase = ((int)blockIdx.x >> 3) % W16_PHASES;
        for (int i = 0; i < phase * (W16_SPREAD / W16_PHASES / (64 * 100)); ++i) __builtin_amdgcn_s_sleep(100);
    }
    if (W16_PRIO == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    // ---- prologue: bias, chunks 0 and 1 and the first weight half of the first tile, the transform of its first halo row
#ifdef W16_STAMP
    if (wave == 0) bias_s[lane] = 0.0f;                       // (the profiling build takes its stamp buffer through the bias pointer)
#else
    if (wave == 0) bias_s[lane] = bias ? bias[lane] : 0.0f;
#endif
    fetch_tile_uniform(t_first, x);
    if (mover) {
#pragma unroll
        for (int j = 0; j < (W16_DMA4 ? 2 : 1) * RAW_INSTR; ++j) voff_set(j, fetch_lane_offset(j));
#pragma unroll
        for (int j = 0; j < (W16_DMA4 ? 2 : 1) * W_INSTR; ++j) w_piece(Wp, 0, 0, j);
#pragma unroll
        for (int j = 0; j < (W16_DMA4 ? 2 : 1) * W_INSTR; ++j) w_piece(Wp, 1, 1, j);
    }
    fetch_two_chunks_and_wait();
    __syncthreads();
    first_row_and_fragments();

    auto flag_word = [&](int t) __attribute__((always_inline)) -> const unsigned* {
        const int tpi = tiles_x * tiles_y;
        const int n = mdiv(t, mg_img, sh_img), r = t - n * tpi, by = mdiv(r, mg_tx, sh_tx), bx = r - by * tiles_x;
        const int k = lane < 9 ? lane : 4;
        const int ny = by + k / 3 - 1, nx = bx + k % 3 - 1;
        const bool ok = ny >= 0 && ny < tiles_y && nx >= 0 && nx < tiles_x;
        return flags + (int64_t)(ok ? n * tpi + ny * tiles_x + nx : t) * STACK_FLAG_STRIDE;
    };
    int pend_t = -1;                                           // (STACK) a finished tile whose word is published once its stores are waited for (half-stage 1's barrier)
    unsigned pend_v = 0;
    // ---- the finished tile's outputs wait in registers (the 64 of the positions 2, 3, which the next tile touches only in its second
    // half-stage) and go out in the gaps of the NEXT tile's first half-stage, two gaps apart: the 16 stores of a wave take ~150 cycles each
    // to issue when every CU stores at once (the write path, not the CU) - behind one another at the end of a tile they were a third of it
    f32x4 o[2][4][2];                                          // [cout group][8-cout block of the group][pixel 2t, 2t + 1]
    Done dp = {(i32x4){0, 0, 0, 0}, {RAW_OOB, RAW_OOB}, 0, 0.0f, 0.0f};       // ... and where they go
    bool have_o = false;
    auto store_step = [&](int k) __attribute__((always_inline)) {    // k = 0 .. 15
        const int g = k >> 3, gq = (k >> 1) & 3, p = k & 1;
        if (!(dp.oy < H) || ((W16_ABL & 8) && relu != 77)) return;       // (uniform)
        const uint32_t so = uniform((uint32_t)(2 * (4 * g + gq)) * (uint32_t)HW * 16u + (uint32_t)(dp.oy * Wq) * 16u);
#ifdef W16_STACK_NT   // (timing experiment only: not coherent)
        if (STACK) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" ::"v"(o[g][gq][p]), "v"(dp.pix[p]), "s"(dp.orsrc), "s"(so) : "memory");
#else
        if (STACK) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc1\n\ts_nop 1" ::"v"(o[g][gq][p]), "v"(dp.pix[p]), "s"(dp.orsrc), "s"(so) : "memory");
#endif
        else asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" ::"v"(o[g][gq][p]), "v"(dp.pix[p]), "s"(dp.orsrc), "s"(so) : "memory");
    };
    bool raw_flying = true;                                    // a halo tile went out behind the last weights (the prologue's did)
#pragma unroll 1
    for (int t_cur = t_first;;) {
        const bool new_layer = STACK && !(t_cur + t_step < t_end);
        const int t_next = new_layer ? t_first : t_cur + t_step;
        const bool next = STACK ? (!new_layer || L + 1 < n_layers) : t_next < t_end;
        const int L_next = L + (new_layer ? 1 : 0);
        const bool poll = STACK && next && L_next > 0;
        if (STACK && new_layer && next) Wnx = layers[L + 1].w;
        // (the tile before's outputs: one store every other gap)
        half_stage(0, true, false, raw_flying, nothing, [&](int m) __attribute__((always_inline)) {
            if ((m & 1) && m < 32 && have_o) store_step(m >> 1);
        });
        have_o = false;
        // (STACK) behind half-stage 1's wait and barrier every store of the tile before has been acknowledged: its word goes out
        half_stage(1, true, true, false, nothing, [&](int m) __attribute__((always_inline)) {
            if (STACK && m == 24 && pend_t >= 0 && wave == 2 && lane == 0)
                __hip_atomic_store(flags + (int64_t)pend_t * STACK_FLAG_STRIDE, pend_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        });
        pend_t = -1;
        half_stage(2, true, false, true, nothing, no_shadow);
        // (STACK) ONE poll of the words of the tiles the next tile reads: wave 0 asks behind half-stage 3's barrier, IN FRONT of the DMA of
        // its last group (half-stage 4's wait then covers it), the verdict goes through LDS in front of half-stage 4's barrier
        half_stage(3, true, true, false, nothing, [&](int m) __attribute__((always_inline)) {
            if (STACK && m == 24 && wave == 0 && poll) {
                const unsigned* pf = flag_word(t_next);
                const uint32_t m0v = uniform((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)poll_s);
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off sc1" ::"s"(m0v), "v"(pf) : "m0", "memory");
            }
        });
        // the NEXT tile's fetch descriptor and lane offsets (this tile's last halo DMA went out in half-stage 3)
        half_stage(4, true, false, true, [&]() __attribute__((always_inline)) {
            if (STACK && wave == 0) {
                const unsigned pv = poll_s[lane];
                const unsigned target = fbase + (unsigned)L_next;
                const bool late = poll && lane < 9 && (int)(pv - target) < 0;
                const uint32_t ok = (!poll || fgiveup) ? 1u : (__builtin_amdgcn_ballot_w64(late) == 0 ? 1u : 0u);
                if (lane == 0) ready_s = ok;
            }
        }, [&](int m) __attribute__((always_inline)) {
            if (m == 1 && next) fetch_tile_uniform(t_next, (STACK && new_layer) ? y : x);
            if (W16_DMA4) { if (m >= 2 && m < 2 + 2 * RAW_INSTR && next && mover) voff_set(m - 2, fetch_lane_offset(m - 2)); }
            else if (m >= 2 && m < 2 + 2 * RAW_INSTR && !(m & 1) && next) voff_set((m - 2) >> 1, fetch_lane_offset((m - 2) >> 1));
        });
        bool ready = true;
        if (STACK) {
            ready = uniform(ready_s) != 0;
            s3_raw = next && ready;
            s3_w = new_layer ? Wnx : Wp;
        }
        const bool rn = STACK ? s3_raw : next;                 // the next tile's first two halo chunks go out in half-stages 5 and 7
        half_stage(5, true, rn, false, nothing, no_shadow);
        half_stage(6, next, false, rn, nothing, no_shadow);
        // (where this tile's outputs go: the tile before's went out in half-stage 0)
        half_stage(7, next, rn, false, nothing, [&](int m) __attribute__((always_inline)) { if (m == 28) dp = tile_done(t_cur); });
        raw_flying = rn;

        // ---- epilogue: y[2t] = M0 + M1 + M2, y[2t + 1] = M1 - M2 - M3, x 2^(e_out - e_in - w_exp), + bias, ReLU (the NaN-propagating maximum).
        // acc[.][g][i] is cout 32 g + 8 (i >> 2) + 4 kb + (i & 3) of tile pl of row wave: the lane's four consecutive couts of an 8-cout block
        // b8 = 4 g + gq are the 16 bytes of its pixel in plane 2 b8 + kb (the lane's kb rides in pix).
        // (round 6) a tile whose outputs cannot wait for the next tile's first half-stage - the slow path below, the launch's last tile - stores each
        // pair of pixels as soon as it exists: the 16 stores take ~150 cycles each to issue, and behind the epilogue's arithmetic they were 2400
        // cycles of the hand-over between two layers at one tile per workgroup (one measurement per call)
        const bool store_now = W16_STORE_NOW && (STACK ? (!next || !ready) : !next);
        if (!(W16_ABL & 4) || relu == 77) {
            const float floor_ = relu ? 0.0f : -__builtin_inff();
            const __attribute__((address_space(3))) float* bsl = (const __attribute__((address_space(3))) float*)bias_s + (STACK ? 64 * (L & 1) : 0);
#pragma unroll
            for (int g = 0; g < 2; ++g) {
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 bz = *reinterpret_cast<const lds_f32x4*>(bsl + 32 * g + 8 * gq + 4 * kb) * dp.bscale;
#if W16_EPI_PK
                    // (round 6) two couts per instruction: v_pk_add_f32 / v_pk_fma_f32 - the same IEEE operations lane by lane (the same bits), half the
                    // issue slots; nothing competes for them here: both waves of the SIMD are in their epilogues, no MFMA runs beside them
#pragma unroll
                    for (int k = 0; k < 4; k += 2) {
                        const int i = 4 * gq + k;
                        const f32x2 m0 = {acc[0][g][i], acc[0][g][i + 1]}, m1 = {acc[1][g][i], acc[1][g][i + 1]};
                        const f32x2 m2 = {acc[2][g][i], acc[2][g][i + 1]}, m3 = {acc[3][g][i], acc[3][g][i + 1]};
                        const f32x2 y0 = pk_add(pk_add(m0, m1), m2), y1 = pk_sub(pk_sub(m1, m2), m3);
                        const f32x2 sc = {dp.oscale, dp.oscale}, b2 = {bz[k], bz[k + 1]};
                        const f32x2 z0 = pk_fma(y0, sc, b2), z1 = pk_fma(y1, sc, b2);
                        o[g][gq][0][k] = __builtin_elementwise_maximum(z0.x, floor_);
                        o[g][gq][0][k + 1] = __builtin_elementwise_maximum(z0.y, floor_);
                        o[g][gq][1][k] = __builtin_elementwise_maximum(z1.x, floor_);
                        o[g][gq][1][k + 1] = __builtin_elementwise_maximum(z1.y, floor_);
                    }
#else
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int i = 4 * gq + k;
                        const float m0 = acc[0][g][i], m1 = acc[1][g][i], m2 = acc[2][g][i], m3 = acc[3][g][i];
                        const float y0 = (m0 + m1) + m2, y1 = (m1 - m2) - m3;
                        o[g][gq][0][k] = __builtin_elementwise_maximum(__builtin_fmaf(y0, dp.oscale, bz[k]), floor_);
                        o[g][gq][1][k] = __builtin_elementwise_maximum(__builtin_fmaf(y1, dp.oscale, bz[k]), floor_);
                    }
#endif
                    if (store_now) { store_step(2 * (4 * g + gq)); store_step(2 * (4 * g + gq) + 1); }
                }
            }
            have_o = !store_now;
        }
        // the next tile's second weight half (issued in half-stage 7's last group, in front of a halo tile that may still fly) has to be there
        // before a store goes out behind it: vmcnt counts loads and stores alike
        if (rn && !(W16_ABL & 1)) wait_all_but_a_tile();
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
                // ---- the tiles the next tile reads are not all written (always so with ONE tile per workgroup): store, wait for the stores,
                // publish, wait for the nine words, fetch, transform
                if (have_o) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) store_step(k);
                    have_o = false;
                }
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
                fetch_two_chunks_and_wait();
                __syncthreads();
                first_row_and_fragments();
                raw_flying = true;
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
    if (have_o) {                                              // the last tile's outputs
#pragma unroll
        for (int k = 0; k < 16; ++k) store_step(k);
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

extern "C" int deqsci_conv3x3_c64_wino16(const void* x_p32, const void* u_packed, const float* bias, void* y_p32, int64_t n, int64_t H, int64_t W,
                                         int relu, int w_exp, const float* in_amax, int in_exp, const float* out_amax, int out_exp,
                                         deqsci_stream_t stream, void* start_event, void* stop_event) {
    if (!x_p32 || !u_packed || !y_p32) return DEQSCI_ERR_NULL;
    if ((start_event == nullptr) != (stop_event == nullptr)) return DEQSCI_ERR_NULL;
    if (n <= 0 || H <= 0 || W <= 0) return DEQSCI_ERR_SHAPE;
    if (x_p32 == y_p32 || w16_bad_exp(w_exp) || w16_bad_exp(in_exp) || w16_bad_exp(out_exp)) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(x_p32) || !aligned16(u_packed) || !aligned16(y_p32)) return DEQSCI_ERR_ALIGN;
    const int64_t tiles_x = ceil_div(W, w16::OUT_COLS), tiles_y = ceil_div(H, w16::OUT_ROWS);
    const int64_t n_tiles = n * tiles_x * tiles_y;
    // 32-bit byte offsets inside one image, and the out-of-range sentinel 2^31 must lie beyond the descriptor's range
    if (n_tiles > (int64_t)INT32_MAX / 16 || H * tiles_x * w16::OUT_COLS * 256 + 16 > (int64_t)w16::RAW_OOB) return DEQSCI_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t resident = (int64_t)num_cus();
    const dim3 grid((unsigned)(n_tiles < resident ? n_tiles : resident));
    uint32_t mg_img, sh_img, mg_tx, sh_tx;
    w16_magic((uint32_t)(tiles_x * tiles_y), &mg_img, &sh_img);
    w16_magic((uint32_t)tiles_x, &mg_tx, &sh_tx);
    hipEvent_t ev0 = static_cast<hipEvent_t>(start_event), ev1 = static_cast<hipEvent_t>(stop_event);
    hipExtLaunchKernelGGL((w16::conv_w16_kernel<0>), grid, dim3(w16::TBW), 0, st, ev0, ev1, 0, static_cast<const char*>(x_p32), static_cast<const char*>(u_packed),
                          bias, static_cast<char*>(y_p32), (int)H, (int)W, relu, w_exp, in_amax, in_exp, out_amax, out_exp, (int)tiles_x,
                          (int)tiles_y, (int)n_tiles, mg_img, sh_img, mg_tx, sh_tx, static_cast<char*>(nullptr),
                          static_cast<const w16::StackLayer*>(nullptr), 1, static_cast<unsigned*>(nullptr), 0);
    return launch_status();
}

extern "C" int deqsci_conv3x3_c64_wino16_stack(const void* x_p32, void* y_even, void* y_odd, const void* layers, int n_layers,
                                               int64_t n, int64_t H, int64_t W, const float* ranges, int64_t range_stride, int in_exp, int out_exp,
                                               void* flags, deqsci_stream_t stream, void* start_event, void* stop_event) {
    if (!x_p32 || !y_even || !layers || !flags || (n_layers > 1 && !y_odd)) return DEQSCI_ERR_NULL;
    if ((start_event == nullptr) != (stop_event == nullptr)) return DEQSCI_ERR_NULL;
    if (n <= 0 || H <= 0 || W <= 0 || n_layers <= 0 || (ranges && (range_stride < n || range_stride > INT32_MAX))) return DEQSCI_ERR_SHAPE;
    if (x_p32 == y_even || x_p32 == y_odd || y_even == y_odd || n_layers > 64 || w16_bad_exp(in_exp) || w16_bad_exp(out_exp)) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(x_p32) || !aligned16(y_even) || !aligned16(y_odd) || (reinterpret_cast<uintptr_t>(layers) & 7u) || (reinterpret_cast<uintptr_t>(flags) & 3u))
        return DEQSCI_ERR_ALIGN;
    const int64_t tiles_x = ceil_div(W, w16::OUT_COLS), tiles_y = ceil_div(H, w16::OUT_ROWS);
    const int64_t n_tiles = n * tiles_x * tiles_y;
    if (H * tiles_x * w16::OUT_COLS * 256 + 16 > (int64_t)w16::RAW_OOB) return DEQSCI_ERR_UNSUPPORTED;
    // every workgroup of the launch has to be RESIDENT (they wait for one another): one per CU - the kernel's 148 KB of LDS admit no second
    // one - so never more workgroups than CUs; each walks its tiles layer after layer
    if (n_tiles > (int64_t)INT32_MAX / (16 * 32)) return DEQSCI_ERR_UNSUPPORTED;
    static int occ_cache[64] = {0};
    const int64_t fit = resident_workgroups(w16::conv_w16_kernel<1>, w16::TBW, occ_cache);
    if (fit <= 0) return DEQSCI_ERR_UNSUPPORTED;                // (the waits inside the launch need every workgroup resident: ask the runtime, do not assume)
    const int64_t resident = fit < (int64_t)num_cus() ? fit : (int64_t)num_cus();
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint32_t mg_img, sh_img, mg_tx, sh_tx;
    w16_magic((uint32_t)(tiles_x * tiles_y), &mg_img, &sh_img);
    w16_magic((uint32_t)tiles_x, &mg_tx, &sh_tx);
    hipEvent_t ev0 = static_cast<hipEvent_t>(start_event), ev1 = static_cast<hipEvent_t>(stop_event);
    hipExtLaunchKernelGGL((w16::conv_w16_kernel<1>), dim3((unsigned)(n_tiles < resident ? n_tiles : resident)), dim3(w16::TBW), 0, st, ev0, ev1, 0,
                          static_cast<const char*>(x_p32), static_cast<const char*>(nullptr), static_cast<const float*>(nullptr), static_cast<char*>(y_even),
                          (int)H, (int)W, 0, 0, ranges, in_exp, static_cast<const float*>(nullptr), out_exp, (int)tiles_x, (int)tiles_y, (int)n_tiles,
                          mg_img, sh_img, mg_tx, sh_tx, static_cast<char*>(y_odd), static_cast<const w16::StackLayer*>(layers), n_layers,
                          static_cast<unsigned*>(flags), (int)range_stride);
    return launch_status();
}
