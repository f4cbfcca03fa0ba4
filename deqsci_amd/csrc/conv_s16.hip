// 3x3 convolution 64 -> 64 channels (pad 1, stride 1) as a DIRECT convolution on the f16 matrix cores of MI355X with fp32-class
// accuracy: every fp32 operand is split into two fp16 pieces, x = hi + lo (hi = fp16(x), lo = fp16(x - hi): 22 significant bits),
// and the product is formed from three f16 MFMAs with fp32 accumulation
//        w x  ~=  w_hi x_hi + w_lo x_hi + w_hi x_lo            (the dropped w_lo x_lo term is 2^-22 relative)
// The f16 matrix pipe is 16x faster than the f32 one (v_mfma_f32_32x32x16_f16: 32768 flops per 32 cycles against 2048 for
// v_mfma_f32_16x16x4_f32), so three f16 products of a direct convolution (3 x 9 taps) cost 0.42 of the f32 MFMA time of Winograd
// F(4x4,3x3) (2.25 taps-equivalent) - and there is NO transform: no input transform (13 % of the F(4x4,3x3) kernel), no output
// transform, no exchange, none of the cancellation that makes Winograd forms noisy on rough inputs.  Measured rounding against a
// float64 convolution on FFDNet's own data: see tools/conv_error_real.py (on par with the fp32 forms, below MIOpen's direct fp32
// convolution), tools/ubench/mfma_f16_numerics.hip for what the f16 MFMA does with its 16 products (fp16 subnormals kept, one
// rounding per instruction).
//
// Scaling: fp16 has 5 exponent bits, fp32 is scale-free - so the scales FOLLOW THE DATA.  An activation is stored multiplied by 2^e with
// e = 11 - floor(log2(max |x|)) of that activation (common.hpp: sp16_act_exp; the maximum is measured on the device by the kernel that
// produces the activation - `track_amax` below - at the first f-call of a reconstruction, and every kernel that writes or reads the
// activation derives e from the same device word, so nothing crosses to the host and a captured hipGraph follows its inputs), or with
// a fixed exponent the caller names (2^8 by default).  Each layer's weights carry a power of two chosen at pack time so that max |w|
// lands in [2^13, 2^14).  The lo pieces then stay normal fp16 numbers for every value that matters (down to 2^-14 of the maximum), an
// activation that grows 16-fold beyond the measured maximum shows as inf/NaN in the output, never silently, and the epilogue multiplies
// the fp32 accumulator by the exact power of two that takes it to the output's scale.
//
// Activation layout between layers ("sp16"): [n][cin chunk c (4)][piece hl (2: hi, lo)][k block kb (2)][H][W][8 halfs] - 16 planes
// of 16-byte pixels (256 bytes per pixel in all, as fp32 NHWC).  A pixel's 16 bytes in plane (c, hl, kb) are exactly one lane's
// B operand of v_mfma_f32_32x32x16_f16 (channels 16 c + 8 kb .. + 8), a staged tile row is 544 contiguous bytes per plane, and a
// tap (dy, dx) is a constant offset into the staged plane: conflict-free ds_read_b128 for every tap, no per-tap address arithmetic.
//
// Block tile = 16 x 32 output pixels x 64 couts, one persistent 8-wave workgroup per CU (as csrc/winograd44.hip); wave w owns cout
// group w & 1 (one M tile of 32) x pixel rows 4 (w >> 1) .. + 3 (four N tiles of 32 pixels): four accumulators of 16 registers, twice
// (two accumulation chains).  Input
// channels in chunks of 16 (= K of one MFMA): per chunk the 18 x 34 pixel halo tile (4 planes, 39 KB) and the chunk's weights
// (9 taps x 2 pieces x 2 cout groups x 1 KB = 36 KB, host-packed in LDS order) are double-buffered and fetched by the LDS-DMA path
// one stage ahead; out-of-image pixels are zeros the buffer hardware writes for out-of-range lanes.  One barrier per stage.
#include "common.hpp"
#include <hip/hip_ext.h>
#include <type_traits>
#pragma clang diagnostic ignored "-Winline-asm"

#ifndef S16_RAW_NT
#define S16_RAW_NT 0  // 1: the halo-tile DMA with the non-temporal hint (A/B: tools/s16_variants.sh)
#endif
#if S16_RAW_NT
#define S16_NT " nt"
#else
#define S16_NT ""
#endif
#ifndef S16_EPI_LOCKSTEP
#define S16_EPI_LOCKSTEP 0  // 1: both waves of a SIMD run their epilogue behind the last stage's barrier (the form before the out-of-step one; A/B)
#endif
#ifndef S16_INTERLEAVE
#define S16_INTERLEAVE 1    // 0: a group's operand reads and DMA instruction in front of its six MFMAs, order left to hipcc (A/B)
#endif
#ifndef S16_ST_SEL
#define S16_ST_SEL 0        // cache policy of the output stores (A/B, tools/s16_variants.sh): 0 nt, 1 default, 2 sc1, 3 sc0 sc1, 4 sc0 sc1 nt
#endif
#if S16_ST_SEL == 0
#define S16_ST " nt"
#elif S16_ST_SEL == 1
#define S16_ST ""
#elif S16_ST_SEL == 2
#define S16_ST " sc1"
#elif S16_ST_SEL == 3
#define S16_ST " sc0 sc1"
#else
#define S16_ST " sc0 sc1 nt"
#endif
#ifndef S16_TAIL_XCD
#define S16_TAIL_XCD 1      // 0: the last layer's tiles dealt round-robin over the XCDs (A/B)
#endif
#ifndef S16_ROWS4
#define S16_ROWS4 1   // wave geometry inside the 8-wave workgroup: 1 (shipped since round 4) = FOUR pixel rows x ONE cout group per wave (54 operand
                      // reads from LDS per stage and wave: 18 weight + 36 activation fragments; 242 registers); 0 = two pixel rows x both cout
                      // groups (60: 36 + 24; 256 registers - the round-3 form, kept for the A/B: tools/s16_variants.sh "rows2:-DS16_ROWS4=0")
#endif
#ifndef S16_ZEROC
#define S16_ZEROC S16_ROWS4   // 1: a tile's first MFMA into each accumulator takes the constant 0 as its C operand instead of 128 v_mov_b32 per tile and wave
#endif
#ifndef S16_TILE_VOFF
#define S16_TILE_VOFF S16_ROWS4   // 1: the per-lane offsets of the halo-tile DMA are formed once per TILE (5 registers) instead of once per DMA instruction
#endif
#ifndef S16_ABL
#define S16_ABL 0     // timing ablations only (results wrong; tools/s16_variants.sh): 1 = no DMA inside the stages, 2 = no wait + barrier at the end
                      // of a stage, 4 = no epilogue, 8 = no LDS operand reads (registers reused), 16 = one EXTRA operand read per MFMA group
                      // (S16_ROWS4: + 9 on 54 per stage; what a read costs with the MFMA operands unchanged)
#endif

namespace deqsci {
namespace s16 {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
// hi + lo of an fp32 pair in three instructions (csrc/conv_w16.hip: split_pair): hi = v_cvt_pk_f16_f32 (round to nearest even), lo = fp16(a - hi) by
// v_fma_mixlo / mixhi (a - hi is exact in fp32: one rounding - the bits of converting, converting back, subtracting and converting again)
__device__ __forceinline__ void split_pair(float a0, float a1, unsigned& hi, unsigned& lo) {
    asm("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hi) : "v"(a0), "v"(a1));
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(hi), "v"(a0));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(hi), "v"(a1));
}
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) h8 lds_h8;

constexpr int WAVES = 8, TBW = 64 * WAVES;
constexpr int OUT_ROWS = 16, OUT_COLS = 32, RAW_ROWS = 18, RAW_COLS = 34, RAW_PIX = RAW_ROWS * RAW_COLS;     // 612
constexpr int PLANE_B = RAW_PIX * 16;                          // 9792 bytes of one staged plane
constexpr int RAW_SLOTS = 4 * RAW_PIX;                         // 2448 units of 16 bytes per chunk tile
constexpr int RAW_INSTR = 5;                                   // LDS-DMA instructions of 64 units per wave and chunk (40 in all, 2560 slots)
constexpr int RAW_BUF = WAVES * RAW_INSTR * 1024;              // 40960 bytes
constexpr int W_CHUNK = 9 * 2 * 2 * 1024;                      // 36864 bytes: [tap][hl][cout group][lane][8 halfs]
constexpr uint32_t RAW_BIAS = 4096;                            // the descriptor starts this far below the image (see set_fetch_tile)
constexpr uint32_t RAW_OOB = 0x80000000u;                      // beyond num_records: the hardware writes zeros

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ int mdiv(int t, uint32_t mg, uint32_t sh) { return (int)(((uint64_t)(uint32_t)t * mg) >> sh); }

// OUT_F32 = 0: sp16 output (the next 64->64 layer's input); 1: fp32 channels_last (n, H, W, 64) output (the consumer is not this kernel)
// TRACK = 1: the range MEASUREMENT - the same arithmetic, but instead of storing y the launch folds max |y| (true units, pixels of the
// image only) into track[image] (one atomic per wave and tile)
// The ranges (common.hpp) are PER IMAGE - in_amax, out_amax, track point to n words, one per image of the batch, so that a measurement's
// result does not depend on what else is in the batch: image i of the input holds 2^e_in x, e_in from in_amax[i] (or in_exp), of the sp16
// output 2^e_out y, e_out from out_amax[i] (or out_exp); a tile's scales are formed where its output descriptor is (tile_done)
//
// STACK = 1: a RUN of n_layers 64->64 layers in ONE launch.  The persistent workgroups (one per CU, all resident) walk their tiles layer
// after layer with DATAFLOW synchronisation instead of kernel boundaries: a tile of layer l + 1 needs layer l of itself and its EIGHT
// NEIGHBOURS and nothing else, so every tile has a progress word (flags[32 tile] = layers finished - a 128-byte line per tile -, counted
// on from launch to launch: never reset), written once the tile's stores have been acknowledged and polled by whoever reads the tile.
// With several tiles per workgroup the words a tile waits for were written a tile-time or more ago: ONE poll in the shadow of stage 2
// says so and the next tile's first chunk is fetched during stage 3 as within a layer - the layers run into one another without a
// bubble.  With ONE tile per workgroup (one measurement per call) the next tile is this one, a layer on, and its neighbours finish when
// it does: the workgroup waits for its stores, publishes, waits for the nine words, fetches (the slow path, ~10 k cycles per layer
// against ~14 k for a kernel boundary).  Coherence without cache maintenance: the activations are stored write-through and fetched with
// agent-scope loads (sc1 - the tiles of one image may sit on different XCDs, whose L2s do not snoop each other), the words are
// agent-scope atomics.  What the single launch buys beyond the bubbles: a SLICE of a batch (32 images of 128 x 128: 128 MiB per
// activation) runs its 13 layers back to back, its ping-pong buffers staying in the 256 MiB Infinity Cache - the caller slices.
// Layer l reads x (l = 0) or the buffer layer l - 1 wrote and writes y (l even) / y2 (l odd); its weights, bias, w_exp, relu come from
// `layers`, its ranges from in_amax[l range_stride + image] -> in_amax[(l + 1) range_stride + image] (in_amax = the run's slot table,
// or NULL: in_exp for the run's input, out_exp for every output).  flags[32 n_tiles] != 0: a wait timed out (a workgroup of the
// launch was not resident) - the launch never hangs, the result is invalid and says so.
struct StackLayer { const char* w; const float* bias; int w_exp; int relu; };
constexpr unsigned STACK_SPIN_LIMIT = 1u << 21;                // polls, one every ~0.1 us: a wait gives up after a quarter of a second
#ifndef S16_STACK_POLLS
#define S16_STACK_POLLS 8
#endif
constexpr int STACK_POLLS = S16_STACK_POLLS;                   // polls of the neighbours' words in flight
constexpr int STACK_FLAG_STRIDE = 32;                          // words between two tiles' progress words: a 128-byte line each (256 pollers would queue on eight lines)
template <int OUT_F32, int TRACK, int STACK>
__global__ __launch_bounds__(TBW, 2) void conv_s16_kernel(const char* __restrict__ x, const char* __restrict__ Wp, const float* __restrict__ bias,
                                                          char* __restrict__ y, int H, int W, int relu, int w_exp, const float* __restrict__ in_amax, int in_exp,
                                                          const float* __restrict__ out_amax, int out_exp, float* __restrict__ track, int tiles_x, int tiles_y,
                                                          int n_tiles, uint32_t mg_img, uint32_t sh_img, uint32_t mg_tx, uint32_t sh_tx,
                                                          char* __restrict__ y2, const StackLayer* __restrict__ layers, int n_layers, unsigned* flags,
                                                          int range_stride) {
    __shared__ __attribute__((aligned(16))) char Raw[2 * RAW_BUF];
    __shared__ __attribute__((aligned(16))) char Wt[2 * W_CHUNK];
    __shared__ __attribute__((aligned(16))) float bias_s[STACK ? 128 : 64];     // (STACK: per layer parity - a wave may be a tile ahead of its partner's epilogue)
    __shared__ uint32_t ready_s;                               // (STACK) wave 0's verdict on the next tile's inputs, for all waves
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

    // ---- halo tile by LDS-DMA: slot s = 64 (5 wave + j) + lane of the chunk tile is plane p = s / 612 (p = 2 hl + kb), pixel
    // (row, col) = ((s % 612) / 34, (s % 612) % 34): lane-linear in LDS, a per-lane byte offset on the global side.
    i32x4 rsrc;
    // Split in two so that it can be spread over MFMA groups (see `shadow` below): the wave-uniform part (descriptor, tile origin) and
    // the per-instruction lane offsets.  The slot -> (plane, row, col) arithmetic does not depend on the tile: done once, packed.
    uint32_t slot_rc[RAW_INSTR];                               // row | plane << 5 | col << 8 (col = 2^24 - 1 for slots beyond the tile: never inside an image)
#pragma unroll
    for (int j = 0; j < RAW_INSTR; ++j) {
        const int s = 64 * (RAW_INSTR * wave + j) + lane;
        const int p = (s * 857) >> 19;                         // s / 612 for s < 2560
        const int q = s - p * RAW_PIX;
        const int row = (q * 1928) >> 16, col = q - row * RAW_COLS;          // q / 34 for q < 768
        slot_rc[j] = s < RAW_SLOTS ? (uint32_t)(row | (p << 5) | (col << 8)) : 0xFFFFFF00u;
    }
    int ft_py0 = 0, ft_px0 = 0;
    auto fetch_tile_uniform = [&](int t, const char* xb) __attribute__((always_inline)) {
        const int n = mdiv(t, mg_img, sh_img), r = t - n * (tiles_x * tiles_y);
        const int by = mdiv(r, mg_tx, sh_tx), bx = r - by * tiles_x;
        // the descriptor starts RAW_BIAS bytes BELOW the image: four of a wave's instructions differ only in their immediate offset, which
        // the hardware adds to the LDS AND the global address; the per-lane offsets take it back out
        const uint64_t base = (uint64_t)(xb + (int64_t)n * HW * 256) - RAW_BIAS;
        rsrc.x = (int)uniform((uint32_t)base);
        rsrc.y = (int)uniform((uint32_t)(base >> 32));
        rsrc.z = (int)uniform((uint32_t)(HW * 256) + RAW_BIAS);
        rsrc.w = 0x00020000;
        ft_py0 = OUT_ROWS * by - 1;
        ft_px0 = OUT_COLS * bx - 1;
    };
    auto fetch_lane_offset = [&](int j) __attribute__((always_inline)) -> uint32_t {
        uint32_t rc = slot_rc[j];
        asm volatile("" : "+v"(rc));                           // (opaque: hoisted out of the tile loop, the pieces of this arithmetic spill)
        const int iy = ft_py0 + (int)(rc & 31u), ix = ft_px0 + (int)(rc >> 8);
        const bool ok = (uint32_t)iy < (uint32_t)H && (uint32_t)ix < (uint32_t)W;
        uint32_t off = (((rc >> 5) & 7u) * (uint32_t)HW + (uint32_t)(iy * W + ix)) * 16u + (RAW_BIAS - 1024u * (j & 3));
        asm volatile("" : "+v"(off));                          // (a select, not a branch around the arithmetic)
        return ok ? off : RAW_OOB;
    };
    uint32_t voff[RAW_INSTR];                                  // (S16_TILE_VOFF) the lane offsets of the tile whose chunks are being fetched
    const uint32_t raw_lds = (uint32_t)(uintptr_t)(lds_char*)Raw, wt_lds = (uint32_t)(uintptr_t)(lds_char*)Wt;
    auto raw_piece = [&](int c, int buf, int j) __attribute__((always_inline)) {
        int w_ = wave;
        asm volatile("" : "+s"(w_));                           // (recomputed at every use: hoisted out of the tile loop, these scalars fill the SGPR file)
        const uint32_t soff = uniform((uint32_t)c * (uint32_t)HW * 64u);                               // 4 planes of 16 HW bytes per chunk
        const uint32_t m0v = uniform(raw_lds + (uint32_t)(buf * RAW_BUF + RAW_INSTR * w_ * 1024 + (j == 4 ? 4096 : 0)));
#if S16_TILE_VOFF
        const uint32_t voj = voff[j];
#else
        const uint32_t voj = fetch_lane_offset(j);             // (a dozen vector instructions, in the shadow of the group's MFMAs)
#endif
#define S16_RAW_LOAD(OFFS, MOD) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen" OFFS MOD " lds" ::"s"(m0v), "v"(voj), "s"(rsrc), "s"(soff) : "m0")
        if (STACK) {                                           // agent-scope loads: what another XCD's workgroup wrote through a moment ago
            if (j == 0 || j == 4) S16_RAW_LOAD("", " sc1");
            else if (j == 1) S16_RAW_LOAD(" offset:1024", " sc1");
            else if (j == 2) S16_RAW_LOAD(" offset:2048", " sc1");
            else S16_RAW_LOAD(" offset:3072", " sc1");
        } else {
            if (j == 0 || j == 4) S16_RAW_LOAD("", S16_NT);
            else if (j == 1) S16_RAW_LOAD(" offset:1024", S16_NT);
            else if (j == 2) S16_RAW_LOAD(" offset:2048", S16_NT);
            else S16_RAW_LOAD(" offset:3072", S16_NT);
        }
#undef S16_RAW_LOAD
    };
    // ---- weight chunk: 36 pieces of 1 KiB, host-packed in LDS order; wave w moves the five pieces from 9 w / 2 on
    auto w_piece = [&](const char* Wl, int c, int buf, int j) __attribute__((always_inline)) {
        int w_ = wave;
        asm volatile("" : "+s"(w_));
        const int first = (9 * w_) >> 1;                       // (odd waves own four pieces: their fifth is the next wave's first, fetched twice -
                                                               // the same bytes to the same place - rather than branched around)
        const uint32_t off = (uint32_t)((first + j) * 1024);
        const uint64_t g = (uint64_t)(Wl + (int64_t)c * W_CHUNK) + off;
        const uint32_t m0v = uniform(wt_lds + (uint32_t)(buf * W_CHUNK) + off);
        const uint64_t gs = ((uint64_t)uniform((uint32_t)(g >> 32)) << 32) | uniform((uint32_t)g);
        const uint32_t lv = (uint32_t)lane * 16u;
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(m0v), "v"(lv), "s"(gs) : "m0");
    };

#ifdef S16_STAMP   // profiling build (tools/s16_stamps.py): cycles per phase, summed over the launch, written over the bias array: [workgroup][wave][5]
    uint32_t st_sum[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};         // (STACK: + store drain, barrier, publish + wait for the neighbours, first fetch of the next layer)
    uint64_t st_t = __builtin_readcyclecounter();
#define S16_MARK(i) do { const uint64_t now_ = __builtin_readcyclecounter(); st_sum[i] += (uint32_t)(now_ - st_t); st_t = now_; } while (0)
    uint32_t* st_out = reinterpret_cast<uint32_t*>(const_cast<float*>(STACK ? layers[0].bias : bias));
#else
#define S16_MARK(i) do { } while (0)
#endif
    // [chain][cout group g][pixel row r].  TWO accumulation chains per output: chain 0 takes the hi x hi products, chain 1 the two
    // cross products (2^-11 of the size).  Every MFMA rounds its accumulator once, and the rounding error of a sum of n such steps grows
    // like sqrt(n) ulps OF THAT ACCUMULATOR: with the cross terms out of the way the big chain is 144 steps long instead of 432 (and
    // the small chain's ulps are 2^-11 of the big one's), which takes the per-layer error against float64 from 2.5e-7 to 1.6e-7 - below
    // the fp32 Winograd forms - for 64 more registers and one addition per output in the epilogue.
    // (The measuring launch keeps ONE chain - it only has to find the binary exponent of max |y| - and has its registers to spare.)
    constexpr int CH1 = TRACK ? 0 : 1;
    f32x16 acc[CH1 + 1][2][2];
    // where the finished tile goes: descriptor of its image, per-lane offsets of rows r (S16_ROWS4: of the lane's COLUMN - the rows of a wave
    // are wave-uniform and go into the stores' scalar offset, an out-of-image row is a uniform branch), first pixel row of the wave
    // ... and the tile's scales: acc = 2^(e_in + w_exp) sum w x  ->  oscale acc + bscale bias = 2^e_out y (e of the tile's image)
    struct Done { i32x4 orsrc; uint32_t pix[2]; int oy0; float oscale, bscale; int img; };
    const int pl = lane & 31, kb = lane >> 5;
#if S16_ROWS4
    const int wg = wave & 1, wr4 = 4 * (wave >> 1);             // the wave's cout group and its first pixel row; acc[.][r >> 1][r & 1] is row r
    const lds_char* bbase = (const lds_char*)Raw + kb * PLANE_B + (wr4 * RAW_COLS + pl) * 16;
    const lds_char* abase = (const lds_char*)Wt + wg * 1024 + lane * 16;
#else
    const lds_char* bbase = (const lds_char*)Raw + kb * PLANE_B + (2 * wave * RAW_COLS + pl) * 16;
    const lds_char* abase = (const lds_char*)Wt + lane * 16;
#endif

    // One stage = chunk c of the current tile: Raw[c & 1], Wt[c & 1] hold it; chunk c + 1 (of this tile, or chunk 0 of the next one) is
    // fetched into the other buffers from inside the MFMA stream, one DMA instruction at a time.
    // ---- epilogue of one (r, g, gp) piece: acc[.][g][r][i] is cout 32 g + 8 (i >> 2) + 4 kb + (i & 3) of pixel (2 wave + r, pl)
    typedef __attribute__((address_space(3))) f32x4 lds_f32x4;
    auto ep_piece = [&](const Done& d, int k, const f32x4 (&bz4)[2][4], float& tmax) __attribute__((always_inline)) {
#if S16_ROWS4
        const int r = k >> 1, gp = k & 1, ga = r >> 1, ra = r & 1, gb = 0;   // (ga, ra): where row r sits in acc; gb: the bias set
        const int g = wg;                                      // (wave-uniform, not a constant: the stores' scalar offsets)
        const bool row_ok = d.oy0 + r < H;                     // (uniform)
        const uint32_t pixv = d.pix[0];
        const uint32_t rowoff = (uint32_t)((d.oy0 + r) * W) * (OUT_F32 ? 256u : 16u);
#else
        const int r = k >> 2, g = (k >> 1) & 1, gp = k & 1, ga = g, ra = r, gb = g;
        const bool row_ok = true;
        const uint32_t pixv = d.pix[r], rowoff = 0;
#endif
        // values in PAIRS (k, k + 1): every step below is one packed instruction per pair where the hardware has one.  ReLU is the
        // NaN-propagating maximum (v_maximum3_f32) against 0, or against -inf when the layer has none: an overflow upstream (inf in the
        // fp16 pieces -> inf - inf in the accumulators) stays a NaN all the way to the output instead of being clamped to 0.
        const float floor_ = relu ? 0.0f : -__builtin_inff();
        f32x2 v2[4];                                           // [gq = 2 gp, 2 gp + 1][k pair]
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * (2 * gp + (e >> 1)) + 2 * (e & 1);
            const f32x2 a0 = {acc[0][ga][ra][i], acc[0][ga][ra][i + 1]}, a1 = TRACK ? (f32x2){0.0f, 0.0f} : (f32x2){acc[CH1][ga][ra][i], acc[CH1][ga][ra][i + 1]};
            const f32x4 b4 = bz4[gb][2 * gp + (e >> 1)];
            const f32x2 bz = (e & 1) ? (f32x2){b4.z, b4.w} : (f32x2){b4.x, b4.y};
            f32x2 t = __builtin_elementwise_fma(a0 + a1, (f32x2){d.oscale, d.oscale}, bz);
            t.x = __builtin_elementwise_maximum(t.x, floor_);
            t.y = __builtin_elementwise_maximum(t.y, floor_);
            if (TRACK && row_ok && pixv != RAW_OOB) tmax = fmaxf(tmax, fmaxf(__builtin_fabsf(t.x), __builtin_fabsf(t.y)));   // (pixels of the image only)
            v2[e] = t;
        }
        if (TRACK) {                                           // (the measuring launch writes nothing; pieces one after the other, as the stores keep them)
            asm volatile("" : "+v"(tmax));
            __builtin_amdgcn_sched_barrier(0);
            return;
        }
        if (!row_ok) return;
        if (OUT_F32) {
            // fp32 channels_last: the lane's four consecutive couts of each group are 16 contiguous bytes (pix = byte offset of the pixel)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const f32x4 o = {v2[2 * q].x, v2[2 * q].y, v2[2 * q + 1].x, v2[2 * q + 1].y};
                const uint32_t so = uniform((uint32_t)((32 * g + 8 * (2 * gp + q)) * 4) + rowoff);
                asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" S16_ST "\n\ts_nop 1" ::"v"(o), "v"(pixv), "s"(d.orsrc), "s"(so) : "memory");
            }
        } else {
            // sp16: split, pack, and trade halves with the lane 32 away so that each lane holds one whole 16-byte pixel of
            // plane (chunk 2 g + gp, hl, kb) - even couts-of-8 block for lanes < 32, odd block for lanes >= 32
            unsigned hi[4], lo[4];                             // [block parity (gq & 1)][k pair]
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const h2 hh = __builtin_convertvector(v2[e], h2);                          // v_cvt_pk_f16_f32 (round to nearest even)
                const f32x2 rem = v2[e] - __builtin_convertvector(hh, f32x2);               // exact in fp32
                hi[e] = __builtin_bit_cast(unsigned, hh);
                lo[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(rem, h2));
            }
#pragma unroll
            for (int e = 0; e < 2; ++e) {                      // (E, O) = (block 0, block 1) registers e: swap E[32..63] with O[0..31]
                auto sh = __builtin_amdgcn_permlane32_swap(hi[e], hi[2 + e], false, false);
                hi[e] = sh[0]; hi[2 + e] = sh[1];
                auto sl = __builtin_amdgcn_permlane32_swap(lo[e], lo[2 + e], false, false);
                lo[e] = sl[0]; lo[2 + e] = sl[1];
            }
            // now lanes < 32 hold couts [0, 8) of block 0 as (hi[0], hi[1], hi[2], hi[3]) = (own 0..3, partner's 4..7); lanes >= 32
            // hold block 1 of pixel lane - 32 likewise: plane kb of the lane (inside pix), pixel pl
            const u32x4 oh = {hi[0], hi[1], hi[2], hi[3]}, ol = {lo[0], lo[1], lo[2], lo[3]};
            const uint32_t so_h = uniform((uint32_t)((2 * g + gp) * 4 + 0) * (uint32_t)HW * 16u + rowoff);
            const uint32_t so_l = uniform((uint32_t)((2 * g + gp) * 4 + 2) * (uint32_t)HW * 16u + rowoff);
            if (STACK) {                                       // write-through (agent scope): the neighbours' next layer reads it from another CU, maybe another XCD
                asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc1\n\ts_nop 1" ::"v"(oh), "v"(pixv), "s"(d.orsrc), "s"(so_h) : "memory");
                asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc1\n\ts_nop 1" ::"v"(ol), "v"(pixv), "s"(d.orsrc), "s"(so_l) : "memory");
            } else {
                asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" S16_ST "\n\ts_nop 1" ::"v"(oh), "v"(pixv), "s"(d.orsrc), "s"(so_h) : "memory");
                asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen" S16_ST "\n\ts_nop 1" ::"v"(ol), "v"(pixv), "s"(d.orsrc), "s"(so_l) : "memory");
            }
        }
    };
    int L = 0;                                                 // (STACK) the layer the workgroup is on
    auto tile_done = [&](int t) -> Done {
        Done d;
        const int n = mdiv(t, mg_img, sh_img), rr_ = t - n * (tiles_x * tiles_y);
        const int by = mdiv(rr_, mg_tx, sh_tx), bx = rr_ - by * tiles_x;
        const int ox = OUT_COLS * bx + pl;
        const uint64_t ob = (uint64_t)(y + (int64_t)n * HW * 256);
        d.orsrc.x = (int)uniform((uint32_t)ob);
        d.orsrc.y = (int)uniform((uint32_t)(ob >> 32));
        d.orsrc.z = (int)uniform((uint32_t)(HW * 256));
        d.orsrc.w = 0x00020000;
        {
            // (wave-uniform: scalar loads; the measuring launch: true units; STACK: in_amax is the run's slot table [layer][image])
            const int e_in = STACK ? (in_amax ? sp16_act_exp(in_amax[(int64_t)L * range_stride + n]) : L == 0 ? in_exp : out_exp)
                                   : in_amax ? sp16_act_exp(in_amax[n]) : in_exp;
            const int e_out = STACK ? (in_amax ? sp16_act_exp(in_amax[(int64_t)(L + 1) * range_stride + n]) : out_exp)
                                    : (OUT_F32 || TRACK) ? 0 : out_amax ? sp16_act_exp(out_amax[n]) : out_exp;
            d.oscale = sp16_pow2(e_out - e_in - w_exp);
            d.bscale = sp16_pow2(e_out);
            d.img = n;
        }
#if S16_ROWS4
        d.oy0 = OUT_ROWS * by + wr4;
        d.pix[0] = d.pix[1] = ox >= W ? RAW_OOB : OUT_F32 ? (uint32_t)(ox * 256 + 16 * kb) : (uint32_t)((kb * (int)HW + ox) * 16);
#else
        d.oy0 = 0;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int oy = OUT_ROWS * by + 2 * wave + r;
            const bool ok = oy < H && ox < W;
            d.pix[r] = !ok ? RAW_OOB : OUT_F32 ? (uint32_t)((oy * W + ox) * 256 + 16 * kb) : (uint32_t)((kb * (int)HW + oy * W + ox) * 16);
        }
#endif
        return d;
    };

    const char* Wnx = nullptr;                                 // (STACK) the next layer's weights
    bool s3_raw = false;                                       // (STACK) stage 3 may fetch the next tile's first activation chunk (the tiles it reads are written)
    const char* s3_w = nullptr;                                // (STACK) ... and whose weights go with it (this layer's or the next one's)
    // One stage = chunk c of the current tile: Raw[c & 1], Wt[c & 1] hold it; chunk c + 1 (of this tile, or chunk 0 of
    // the next one) is fetched into the other buffers from inside the MFMA stream, one DMA instruction at a time.
    // `before_barrier` runs between the stage's last MFMA and its barrier, `shadow(i)` inside group i behind four of its MFMAs.
    auto stage = [&](int c, bool more, auto&& before_barrier, auto&& shadow) __attribute__((always_inline)) {
        const int buf = c & 1, nb = buf ^ 1, cn = (c + 1) & 3;
        const lds_char* bb = bbase + buf * RAW_BUF;
        const lds_char* ab = abase + buf * W_CHUNK;
#if S16_ROWS4
        // 9 groups (dx, dy) of 12 MFMAs: the wave's four pixel rows x three products of one tap.  Operands are read from LDS ONE group
        // ahead (= the same twelve MFMAs ahead as in the other geometry): per group the tap's two weight fragments (hi, lo) and the halo
        // rows that come into play - rows 0..3 at dy = 0, row 4 at dy = 1, row 5 at dy = 2 of every dx - into a ring of EIGHT row slots
        // (slot = (6 dx + row) mod 8: while (dx, 2) multiplies rows 2..5, the four dead slots take rows 0..3 of dx + 1).
        h8 Ah[2], Al[2], Bh[8], Bl[8];
        auto loads = [&](int i) __attribute__((always_inline)) {
            if (i >= 9 || ((S16_ABL & 8) && i >= 1)) return;
            const int dx = i / 3, dy = i % 3, tap = dy * 3 + dx;
            Ah[i & 1] = *reinterpret_cast<const lds_h8*>(ab + ((tap * 2 + 0) * 2) * 1024);
            Al[i & 1] = *reinterpret_cast<const lds_h8*>(ab + ((tap * 2 + 1) * 2) * 1024);
#pragma unroll
            for (int rr = (dy == 0 ? 0 : dy + 3); rr <= dy + 3; ++rr) {
                Bh[(6 * dx + rr) & 7] = *reinterpret_cast<const lds_h8*>(bb + (rr * RAW_COLS + dx) * 16);
                Bl[(6 * dx + rr) & 7] = *reinterpret_cast<const lds_h8*>(bb + 2 * PLANE_B + (rr * RAW_COLS + dx) * 16);
            }
        };
        auto dma = [&](int j) __attribute__((always_inline)) {    // the next chunk: 10 DMA instructions, two per group in the first five groups
            if (more && j < 2 * RAW_INSTR && !(S16_ABL & 1)) {
                // (STACK, last stage: the next chunk is chunk 0 of the next tile - maybe of the NEXT LAYER: its weights can come now, its
                // activations only if the tiles it reads have been written: s3_raw)
                if (j < RAW_INSTR) { if (!(STACK && c == 3) || s3_raw) raw_piece(cn, nb, j); }
                else w_piece((STACK && c == 3) ? s3_w : Wp, cn, nb, j - RAW_INSTR);
            }
        };
        S16_MARK(4);
        loads(0);
        __builtin_amdgcn_sched_barrier(0);
        const f32x16 zero16 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        h8 extra[2];                                           // (S16_ABL & 16)
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int dx = i / 3, dy = i % 3;
            const bool z1st = S16_ZEROC && c == 0 && i == 0;   // a tile's first MFMA into an accumulator: C = 0 (an inline constant)
#define S16_SLOT(r) ((6 * dx + dy + (r)) & 7)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[CH1][r >> 1][r & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[i & 1], Bh[S16_SLOT(r)], z1st ? zero16 : acc[CH1][r >> 1][r & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            loads(i + 1);
            if (S16_ABL & 16) {
                extra[i & 1] = *reinterpret_cast<const lds_h8*>(bb + PLANE_B + i * 16);
                if (i > 0) asm volatile("" ::"v"(extra[(i - 1) & 1]));
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[CH1][r >> 1][r & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[i & 1], Bl[S16_SLOT(r)], acc[CH1][r >> 1][r & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            dma(2 * i);
            shadow(2 * i);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[0][r >> 1][r & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[i & 1], Bh[S16_SLOT(r)], (z1st && CH1) ? zero16 : acc[0][r >> 1][r & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            dma(2 * i + 1);
            shadow(2 * i + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 2; r < 4; ++r) acc[0][r >> 1][r & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[i & 1], Bh[S16_SLOT(r)], (z1st && CH1) ? zero16 : acc[0][r >> 1][r & 1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#undef S16_SLOT
        }
        if (S16_ABL & 16) asm volatile("" ::"v"(extra[0]));
#else
        // 18 groups (dx, dy, g) of 6 MFMAs: the two pixel rows x three products of one tap and cout group.  Operands are read from LDS
        // TWO groups ahead (software pipeline pinned by sched_barriers: left to itself hipcc reads each fragment one MFMA before its use
        // and waits for it): per group two weight fragments (hi, lo) and, when a new halo row comes into play, its hi and lo fragments
        // (rows 0, 1 at dy = 0, row 2 at dy = 1, row 3 at dy = 2 of every dx; kept per dx parity).
        h8 Ah[3], Al[3], Bh[2][4], Bl[2][4];
        auto loads = [&](int i) __attribute__((always_inline)) {
            if (i >= 18 || ((S16_ABL & 8) && i >= 2)) return;
            const int dx = i / 6, dy = (i % 6) >> 1, g = i & 1, tap = dy * 3 + dx;
            Ah[i % 3] = *reinterpret_cast<const lds_h8*>(ab + ((tap * 2 + 0) * 2 + g) * 1024);
            Al[i % 3] = *reinterpret_cast<const lds_h8*>(ab + ((tap * 2 + 1) * 2 + g) * 1024);
            if (g == 0) {
#pragma unroll
                for (int rr = (dy == 0 ? 0 : dy + 1); rr <= dy + 1; ++rr) {
                    Bh[dx & 1][rr] = *reinterpret_cast<const lds_h8*>(bb + (rr * RAW_COLS + dx) * 16);
                    Bl[dx & 1][rr] = *reinterpret_cast<const lds_h8*>(bb + 2 * PLANE_B + (rr * RAW_COLS + dx) * 16);
                }
            }
        };
        S16_MARK(4);
        loads(0);
        loads(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 18; ++i) {
            const int dx = i / 6, dy = (i % 6) >> 1, g = i & 1;
            // (hipcc left to itself issues a group's six MFMAs first and everything else behind them: the wave's next MFMA then waits for
            // its own operand reads, DMA set-up and scalar arithmetic to issue - pinned here in the MFMAs' shadow instead)
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[CH1][g][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[i % 3], Bh[dx & 1][dy + r], acc[CH1][g][r], 0, 0, 0);
#if S16_INTERLEAVE
            __builtin_amdgcn_sched_barrier(0);
#endif
            loads(i + 2);
#if S16_INTERLEAVE
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[CH1][g][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[i % 3], Bl[dx & 1][dy + r], acc[CH1][g][r], 0, 0, 0);
#if S16_INTERLEAVE
            __builtin_amdgcn_sched_barrier(0);
#endif
            if (more && i < 2 * RAW_INSTR && !(S16_ABL & 1)) {   // the next chunk: 10 DMA instructions, one per group in the FIRST half of the
                                                               // stage - the last one needs the second half (an HBM round trip) to land
                if (i < RAW_INSTR) { if (!(STACK && c == 3) || s3_raw) raw_piece(cn, nb, i); }
                else w_piece((STACK && c == 3) ? s3_w : Wp, cn, nb, i - RAW_INSTR);
            }
            shadow(i);                                         // (tile bookkeeping rides here, behind four of the group's MFMAs)
#if S16_INTERLEAVE
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int r = 0; r < 2; ++r) acc[0][g][r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[i % 3], Bh[dx & 1][dy + r], acc[0][g][r], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
        S16_MARK(0);
        if (!(S16_ABL & 2)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            S16_MARK(1);
            before_barrier();
            lds_barrier();
            S16_MARK(2);
        }
    };
    auto nothing = [] {};
    auto no_shadow = [](int) {};

#ifdef S16_PRIO
    if (wave >= 4) asm volatile("s_setprio 1");               // (A/B: static priority for the second-dispatched half of the workgroup)
#endif
    const Done none = {(i32x4){0, 0, 0, 0}, {RAW_OOB, RAW_OOB}, 0, 0.0f, 0.0f, 0};
    char* const y_even = y;
    unsigned fbase = 0, fgiveup = 0;
    if (STACK) {
        const StackLayer l0 = layers[0];                       // (wave-uniform: scalar loads)
        Wp = l0.w; bias = l0.bias; w_exp = l0.w_exp; relu = l0.relu;
        fbase = __hip_atomic_load(flags + (int64_t)t_first * STACK_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // where the launch before left the words (all the same)
        fgiveup = __hip_atomic_load(flags + (int64_t)n_tiles * STACK_FLAG_STRIDE, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // an earlier time-out: no more waiting
    }
    // ---- prologue: bias, chunk 0 of the first tile
#ifdef S16_STAMP
    if (wave == 0) bias_s[lane] = 0.0f;                       // (the profiling build takes its stamp buffer through the bias pointer)
#else
    if (wave == 0) bias_s[lane] = bias ? bias[lane] : 0.0f;     // (scaled per tile: the sp16 output of image i carries 2^e_out(i) y, so does its bias)
#endif
    fetch_tile_uniform(t_first, x);
#if S16_TILE_VOFF
#pragma unroll
    for (int j = 0; j < RAW_INSTR; ++j) voff[j] = fetch_lane_offset(j);
#endif
#pragma unroll
    for (int j = 0; j < RAW_INSTR; ++j) { raw_piece(0, 0, j); w_piece(Wp, 0, 0, j); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // (A "loader + 4" form - four compute waves, one per SIMD, each 4 pixel rows x 64 couts, and a fifth wave issuing all 76 DMA
    // instructions of a stage: 40 % less LDS operand traffic, no compute wave ever at the memory pipeline's door - was built and measured
    // (tools/ubench/variants/conv_s16_with_loader_variant.hip, -DS16_V2): 212 us against 193 us.  Five waves put two on one SIMD, so the
    // register budget stays 256 and only one accumulation chain fits; and a wave alone on its SIMD has nobody to cover its stalls.)
    // (Spreading a tile's epilogue over the MFMA stream of the next tile - a second accumulator set, one piece behind every other group -
    // was built and measured: 192 us against 187 us at 64 x 128 x 128.  On random data this kernel runs against the chip's POWER limit
    // (all-zero operands: 142 us), where time follows the energy of the launch, not its idle cycles: the out-of-step epilogue, the
    // pinned MFMA / operand-read / DMA interleave and the tile bookkeeping moved into MFMA shadows took a wave's cycles per launch from
    // 294 k to 280 k (tools/s16_stamps.py, profiles/r03_s16_stamps_*.txt) and the launch from 183 to 180 us - the clock gave the rest back.)
    S16_MARK(4);
    // (STACK) the words of the nine tiles the inputs of tile t come from - its own and its eight neighbours' -, one per lane 0..8 (a lane
    // without a neighbour: the tile's own)
    auto flag_word = [&](int t) __attribute__((always_inline)) -> const unsigned* {
        const int tpi = tiles_x * tiles_y;
        const int n = mdiv(t, mg_img, sh_img), r = t - n * tpi, by = mdiv(r, mg_tx, sh_tx), bx = r - by * tiles_x;
        const int k = lane < 9 ? lane : 4;
        const int ny = by + k / 3 - 1, nx = bx + k % 3 - 1;
        const bool ok = ny >= 0 && ny < tiles_y && nx >= 0 && nx < tiles_x;
        return flags + (int64_t)(ok ? n * tpi + ny * tiles_x + nx : t) * STACK_FLAG_STRIDE;
    };
    int pend_t = -1;                                           // (STACK) a finished tile whose word is published behind the next stage barrier its stores are waited for at
    unsigned pend_v = 0;
    const unsigned* pf = nullptr;                              // (STACK, wave 0) the words the next tile waits for, and what the poll of stage 2 read
    unsigned pv = 0;
#pragma unroll 1
    for (int t_cur = t_first;;) {
        // what comes next: the workgroup's next tile of this layer, or its first tile of the next layer
        const bool new_layer = STACK && !(t_cur + t_step < t_end);
        const int t_next = new_layer ? t_first : t_cur + t_step;
        const bool next = STACK ? (!new_layer || L + 1 < n_layers) : t_next < t_end;
        const int L_next = L + (new_layer ? 1 : 0);
        const bool poll = STACK && next && L_next > 0;         // (a tile of layer 0 reads the run's input: nothing to wait for)
        if (STACK && new_layer && next) Wnx = layers[L + 1].w;
#if !(S16_ZEROC && S16_ROWS4)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[0][g][r][i] = acc[CH1][g][r][i] = 0.0f;
#endif
        stage(0, true, nothing, no_shadow);
        // (STACK) behind stage 0's wait and barrier every store of the tile before has been acknowledged: its word goes out
        stage(1, true, nothing, [&](int i) __attribute__((always_inline)) {
            if (STACK && i == 0 && pend_t >= 0 && wave == 2 && lane == 0)
                __hip_atomic_store(flags + (int64_t)pend_t * STACK_FLAG_STRIDE, pend_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        });
        pend_t = -1;
        // the NEXT tile's fetch descriptor and origin, in the shadow of this stage's MFMA group 10 (its own DMA instructions - the last ones
        // that use this tile's - sit in groups 0..9; the per-lane offsets are formed where each DMA instruction is issued).  Between the
        // tiles this arithmetic cost every wave 1.5 us with the matrix pipe idle
        // (STACK) ... and ONE poll of the words of the tiles the next tile reads (wave 0: asked in group 0, landed by the stage's wait, the
        // verdict through LDS behind its barrier): with several tiles per workgroup they were written a tile or more ago
        stage(2, true, [&]() __attribute__((always_inline)) {
            if (STACK && wave == 0) {
                // (the poll's destination register is only valid behind the wait: said to the compiler too, so that it cannot copy or spill
                //  the register between the load's issue and here - the stage's own wait has already drained the queue, this one is free)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(pv)::"memory");
                const unsigned target = fbase + (unsigned)L_next;
                const bool late = poll && lane < 9 && (int)(pv - target) < 0;
                const uint32_t ok = (!poll || fgiveup) ? 1u : (__builtin_amdgcn_ballot_w64(late) == 0 ? 1u : 0u);
                if (lane == 0) ready_s = ok;
            }
        }, [&](int i) __attribute__((always_inline)) {
            if (STACK && i == 0 && wave == 0 && poll) {
                pf = flag_word(t_next);
                asm volatile("global_load_dword %0, %1, off sc1" : "=v"(pv) : "v"(pf) : "memory");
            }
            if (i == 10 && next) fetch_tile_uniform(t_next, (STACK && new_layer) ? y : x);
#if S16_TILE_VOFF
            if (i >= 11 && i < 11 + RAW_INSTR) voff[i - 11] = fetch_lane_offset(i - 11);     // (this tile's last raw pieces went out in groups 0..4)
#endif
        });
        bool ready = true;
        if (STACK) {
            ready = uniform(ready_s) != 0;
            s3_raw = next && ready;
            s3_w = new_layer ? Wnx : Wp;
        }
        // The epilogue, out of step between the two waves of a SIMD.  Waves 0-3 (one per SIMD, dispatched first: the issue arbiter favours
        // them, they finish a stage's MFMAs in 55 % of its time and idle at the barrier) run their epilogue BEFORE the last stage's barrier,
        // under the MFMAs their SIMD partner is still issuing; waves 4-7 run theirs BEHIND it, under the partner's first MFMAs of the next
        // tile.  The matrix pipe always has a wave feeding it; in lockstep both epilogues of a SIMD ran together with the pipe idle (21 % of
        // a tile: tools/s16_stamps.py, profiles/r03_s16_stamps_before.txt).
        Done d = none;
        auto epilogue = [&]() __attribute__((always_inline)) {
            if (!(S16_ABL & 4) || relu == 77) {
                // the lane's 32 bias values in one burst of LDS reads (operand registers are free here): read piece by piece, each read
                // was a round trip through an LDS the partner wave keeps busy - 16 of them made the epilogue twice as long
                f32x4 bz4[2][4];
                const __attribute__((address_space(3))) float* bsl = (const __attribute__((address_space(3))) float*)bias_s + (STACK ? 64 * (L & 1) : 0);
#if S16_ROWS4
#pragma unroll
                for (int q = 0; q < 4; ++q) bz4[0][q] = bz4[1][q] = *reinterpret_cast<const lds_f32x4*>(bsl + 32 * wg + 8 * q + 4 * kb) * d.bscale;
#else
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int q = 0; q < 4; ++q) bz4[g][q] = *reinterpret_cast<const lds_f32x4*>(bsl + 32 * g + 8 * q + 4 * kb) * d.bscale;
#endif
                float tmax = 0.0f;
#pragma unroll
                for (int k = 0; k < 8; ++k) ep_piece(d, k, bz4, tmax);
                if (TRACK) {                                   // (the measuring launch only: one atomic per tile and wave, on the tile's image)
                    const uint32_t tb = sp16_wave_max_bits(tmax);
                    if (lane == 0 && track) atomicMax(reinterpret_cast<unsigned int*>(track) + d.img, tb);
                }
            }
        };
#if S16_EPI_LOCKSTEP
        stage(3, next, nothing, [&](int i) __attribute__((always_inline)) { if (i == 12) d = tile_done(t_cur); });
        epilogue();
#else
        stage(3, next, [&]() __attribute__((always_inline)) { if (wave < 4) epilogue(); },
              [&](int i) __attribute__((always_inline)) { if (i == 12) d = tile_done(t_cur); });
        if (wave >= 4) epilogue();
#endif
        S16_MARK(3);
        if (STACK) {
            const unsigned done_v = fbase + (unsigned)(L + 1);     // this tile's word once its stores have landed
            if (new_layer && next) {
                // the next tile is of the next layer: this wave's copy of the layer's parameters (the last stage fetched its first weight chunk)
                const StackLayer ln = layers[L + 1];
                x = y;
                y = ((L + 1) & 1) ? y2 : y_even;
                Wp = ln.w; bias = ln.bias; w_exp = ln.w_exp; relu = ln.relu;
#ifndef S16_STAMP
                if (wave == 1) bias_s[64 * ((L + 1) & 1) + lane] = bias ? bias[lane] : 0.0f;
#endif
            }
            if (next && !ready) {
                // ---- the tiles the next tile reads are not all written (always so with ONE tile per workgroup: the next tile is this one, a
                // layer on, and its neighbours finish when it does).  Every wave waits for its stores (write-through: acknowledged = visible
                // to the device); behind the barrier one lane publishes the tile's word and nine lanes wait for the words the next tile
                // needs; then its first chunk.
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                S16_MARK(5);
                __syncthreads();
                S16_MARK(6);
                if (wave == 2 && lane == 0)
                    __hip_atomic_store(flags + (int64_t)t_cur * STACK_FLAG_STRIDE, done_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (wave == 0) {
                    const unsigned target = fbase + (unsigned)L_next;
                    const unsigned* f = flag_word(t_next);
                    // The neighbours finish within a fraction of a microsecond of this tile, and a poll is a round trip to memory (~1 us under the
                    // layer's store burst): one poll at a time almost always needs two.  So STACK_POLLS polls are kept in flight, one every ~200
                    // cycles, and the last neighbour's word is seen half a round trip after it lands.  Loads return in order: with STACK_POLLS
                    // outstanding - this wave has no other memory operation in flight: the tile's own word is published by wave 2, the bias read
                    // by wave 1 - the oldest has landed once vmcnt <= STACK_POLLS - 1.
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
                                if (__builtin_amdgcn_ballot_w64(lane < 9 && (int)(v[q] - target) < 0) == 0) { waiting = false; break; }   // (wrap-safe: the words count on for ever)
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
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // (the polls still in flight write registers)
                    }
                }
                __syncthreads();
                S16_MARK(7);
#pragma unroll
                for (int j = 0; j < RAW_INSTR; ++j) raw_piece(0, 0, j);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                S16_MARK(8);
            } else {
                pend_t = t_cur;                                // (its stores are waited for at the end of the next tile's stage 0, or at the end of the launch)
                pend_v = done_v;
            }
            if (new_layer) ++L;
        }
        if (!next) break;
        t_cur = t_next;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (STACK) {                                               // the last tile's word (its stores: every wave's, hence the barrier)
        __syncthreads();
        if (pend_t >= 0 && threadIdx.x == 0)
            __hip_atomic_store(flags + (int64_t)pend_t * STACK_FLAG_STRIDE, pend_v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
#ifdef S16_STAMP
    if (lane == 0)
        for (int i = 0; i < (STACK ? 9 : 5); ++i) st_out[((int)blockIdx.x * WAVES + wave) * (STACK ? 9 : 5) + i] = st_sum[i];
#endif
}

// fp32 channels_last (n, H, W, 64) -> sp16 holding 2^e x, e from the range (amax, exp): one lane per (pixel, 8-channel block)
__global__ __launch_bounds__(256) void f32_to_sp16_kernel(const float* __restrict__ x, char* __restrict__ y, int64_t HW, int64_t total,
                                                          const float* __restrict__ amax, int exp) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;     // over n * HW * 8
    if (i >= total) return;
    const int blk = (int)(i & 7);
    const int64_t pix = i >> 3, n = pix / HW, p = pix - n * HW;
    const float scale = sp16_pow2(amax ? sp16_act_exp(amax[n]) : exp);
    const float4 a = ld4s(x + pix * 64 + blk * 8), b = ld4s(x + pix * 64 + blk * 8 + 4);
    const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
    h8 hi, lo;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float t = v[k] * scale;
        hi[k] = (_Float16)t;
        lo[k] = (_Float16)(t - (float)hi[k]);
    }
    const int c = blk >> 1, kb = blk & 1;
    char* base = y + n * HW * 256;
    *reinterpret_cast<h8*>(base + ((int64_t)(c * 4 + 0 + kb) * HW + p) * 16) = hi;
    *reinterpret_cast<h8*>(base + ((int64_t)(c * 4 + 2 + kb) * HW + p) * 16) = lo;
}

// max |x| over each image of a batch (n images of `count` contiguous floats; blockIdx.y = image) folded into amax[image] (one atomic per
// workgroup): the range of an activation that no sp16-writing kernel produced (the denoiser's input image; an fp32 activation converted
// by f32_to_sp16_kernel)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, int64_t count, float* __restrict__ amax) {
    __shared__ uint32_t wmax[4];
    float m = 0.0f;
    const float* xi = x + (int64_t)blockIdx.y * count;
    const int64_t stride = (int64_t)gridDim.x * 256 * 4;
    const bool al = ((reinterpret_cast<uintptr_t>(xi) & 15u) == 0);
    for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < count; i += stride) {
        if (al && i + 4 <= count) {
            const float4 v = ld4(xi + i);
            m = fmaxf(fmaxf(m, fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y))), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w)));
        } else {
            for (int64_t k = i; k < count && k < i + 4; ++k) m = fmaxf(m, __builtin_fabsf(xi[k]));
        }
    }
    sp16_track_block_max(m, 1.0f, amax + blockIdx.y, wmax);
}


// ---- the denoisers' LAST layer on the same arithmetic: conv3x3 64 -> COUT (4: FFDNet, followed by its 2x2 pixel shuffle; 1: SimpleCNN), no bias,
// reading the sp16 activation of the last 64->64 layer.  As a matrix product it is tiny in N, so the taps go INTO N:
//     P[pixel q][col = 4 tap + cout] = sum_cin a[q][cin] w[tap][cin][cout]          one 32 x 32 x 16 f16 MFMA tile row per 32 pixels, K = 64 x 3 products
//     out[pixel][cout]               = sum_tap P[pixel + (dy, dx)][4 tap + cout]      nine float4 reads of P from LDS per output pixel
// The activation fragments go from global memory straight into the MFMA's A operand (a lane's 16 bytes of plane (c, hl, kb) ARE its
// operand), the 36-column weight operands (16 KB as hi + lo) stay in registers for the whole workgroup, and the only LDS traffic is P.
// The vector-ALU form of this layer (ffdnet_edges.hip: edge_tail_kernel) was bound by its scalar weight loads: 95-108 us at 64 images of
// 128 x 128 against 45 us for reading its input once.
constexpr int TL_H = 8, TL_W = 32, TL_IW = TL_W + 2, TL_IH = TL_H + 2, TL_PIX = TL_IH * TL_IW;      // 8 x 32 outputs, 10 x 34 = 340 halo pixels
constexpr int TL_MB = (TL_PIX + 31) / 32;                                                      // 11 blocks of 32 halo pixels
// P32 = 1: the input is "p32" (csrc/conv_w16.hip: the same 16 planes holding 2^e x as fp32, plane 2 b8 + j = channels 8 b8 + 4 j .. + 4, the
// columns of every block of 64 with the parities apart): two 16-byte loads per fragment, split into hi + lo on the fly
template <int COUT, int P32>
__global__ __launch_bounds__(256) void tail_s16_kernel(const char* __restrict__ x, const char* __restrict__ Wp, float* __restrict__ out, int H, int W,
                                                       int w_exp, const float* __restrict__ in_amax, int in_exp, int tiles_x, int tiles_y, int n_tiles) {
    constexpr int NCOL = 9 * COUT, NT = (NCOL + 31) / 32, PS = NCOL;              // P row stride in floats (36: float4-aligned; 9)

    __shared__ __attribute__((aligned(16))) float P[TL_PIX * PS + 32 * 36];       // (+ slack: the last pixel block writes 352 rows)
    // Workgroup b runs on XCD b % 8: every XCD takes a contiguous range of tiles (image-major, then rows), so that the two halo rows a
    // tile shares with the tile above and below it are re-read from THAT XCD's L2 while they are hot - with tiles dealt round-robin
    // over the XCDs every halo row came from HBM again (10 rows fetched per 8 produced)
#if S16_TAIL_XCD
    const int per_xcd = (n_tiles + 7) >> 3, t = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
#else
    const int t = (int)blockIdx.x;
#endif
    if (t >= n_tiles) return;
    const int n = t / (tiles_x * tiles_y), rt = t - n * (tiles_x * tiles_y), by = rt / tiles_x;
    const int r0 = by * TL_H, c0 = (rt - by * tiles_x) * TL_W;
    const float oscale = sp16_pow2(-(in_amax ? sp16_act_exp(in_amax[n]) : in_exp) - w_exp);   // acc = 2^(e_in + w_exp) sum w x  ->  true units (the tile's image)
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6), pl = lane & 31, kb = lane >> 5;
    const int Wq = P32 ? 64 * ((W + 63) / 64) : W;             // row pitch of a plane
    const int64_t HW = (int64_t)H * Wq;
    const char* xn = x + (int64_t)n * HW * 256;
    // weight operands (B: column = lane % 32 of N tile nt, k block = lane / 32): [chunk][hl][nt][lane][8 halfs]
    h8 Bw[4][2][NT];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) Bw[c][hl][nt] = *reinterpret_cast<const h8*>(Wp + ((((c * 2 + hl) * NT + nt) * 64 + lane) * 16));
    // a wave's pixel blocks mb = wave, wave + 4, wave + 8.  (Requesting the NEXT block's fragments before multiplying the current one was
    // measured: 74.0 vs 72.3 us - the extra 32 registers cost a workgroup per CU; three resident workgroups already overlap each other.)
    auto fetch = [&](int mb, h8 (&A)[4][2]) __attribute__((always_inline)) {
        const int q = 32 * mb + pl, row = q / TL_IW, col = q - row * TL_IW;
        const int gr = r0 - 1 + row, gc = c0 - 1 + col;
        const bool ok = mb < TL_MB && q < TL_PIX && gr >= 0 && gr < H && gc >= 0 && gc < W;
        if (P32) {
            // plane 2 (2 c + kb) + j: + (4 c + j) HW 16 bytes from the lane's k block; 2^e x as fp32 -> hi + lo
            const char* px = xn + ((int64_t)(2 * kb) * HW + (int64_t)gr * Wq + (gc & ~63) + ((gc & 1) << 5) + ((gc & 63) >> 1)) * 16;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                f32x4 v0 = {0, 0, 0, 0}, v1 = {0, 0, 0, 0};
                if (ok) {
                    v0 = *reinterpret_cast<const f32x4*>(px + (int64_t)(4 * c) * HW * 16);
                    v1 = *reinterpret_cast<const f32x4*>(px + (int64_t)(4 * c + 1) * HW * 16);
                }
                const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                // (the three-instruction split of csrc/conv_w16.hip - v_cvt_pk_f16_f32 + v_fma_mixlo / mixhi by inline asm - was measured here in
                //  round 6: 33.5 -> 39.4 us in the loop; the compiler cannot schedule the asm statements between the loads it is waiting for)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const _Float16 hh = (_Float16)v[k];
                    A[c][0][k] = hh;
                    A[c][1][k] = (_Float16)(v[k] - (float)hh);
                }
            }
            return;
        }
        const char* px = xn + ((int64_t)kb * HW + (int64_t)gr * W + gc) * 16;   // plane (c, hl, kb): + (4 c + 2 hl) HW 16 bytes
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int hl = 0; hl < 2; ++hl) {
                h8 v = {0, 0, 0, 0, 0, 0, 0, 0};
                if (ok) v = *reinterpret_cast<const h8*>(px + (int64_t)(4 * c + 2 * hl) * HW * 16);
                A[c][hl] = v;
            }
    };
    h8 A[4][2];
#pragma unroll 1
    for (int mb = wave; mb < TL_MB; mb += 4) {
        fetch(mb, A);
        f32x16 acc[NT];                                        // (K = 64 per tap column: twelve MFMAs per accumulator - one chain, cross terms first)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nt][i] = 0.0f;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[c][1], Bw[c][0][nt], acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[c][0], Bw[c][1][nt], acc[nt], 0, 0, 0);
            }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[c][0], Bw[c][0][nt], acc[nt], 0, 0, 0);
        // D: row (pixel) 8 (i >> 2) + 4 kb + (i & 3) of the block, column pl of the N tile
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int colp = 32 * nt + pl;
            if (colp < NCOL) {
#pragma unroll
                for (int i = 0; i < 16; ++i) P[(32 * mb + 8 * (i >> 2) + 4 * kb + (i & 3)) * PS + colp] = acc[nt][i] * oscale;
            }
        }
    }
    __syncthreads();
    const int lr = (int)threadIdx.x / TL_W, lc = (int)threadIdx.x % TL_W, r = r0 + lr, cc = c0 + lc;
    float o[COUT];
#pragma unroll
    for (int k = 0; k < COUT; ++k) o[k] = 0.0f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        const float* pp = P + ((lr + tap / 3) * TL_IW + lc + tap % 3) * PS + COUT * tap;
        if (COUT == 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(pp);
            o[0] += v.x; o[COUT > 1 ? 1 : 0] += v.y; o[COUT > 2 ? 2 : 0] += v.z; o[COUT > 3 ? 3 : 0] += v.w;
        } else {
#pragma unroll
            for (int k = 0; k < COUT; ++k) o[k] += pp[k];
        }
    }
    if (r < H && cc < W) {
        if (COUT == 4) {        // pixel_shuffle(2): channel 2i+j -> (2r+i, 2c+j)
            float* op = out + (int64_t)n * 4 * H * W + (int64_t)(2 * r) * (2 * W) + 2 * cc;
            *reinterpret_cast<f32x2*>(op) = (f32x2){o[0], o[COUT > 1 ? 1 : 0]};
            *reinterpret_cast<f32x2*>(op + 2 * W) = (f32x2){o[COUT > 2 ? 2 : 0], o[COUT > 3 ? 3 : 0]};
        } else {
            out[(int64_t)n * H * W + (int64_t)r * W + cc] = o[0];
        }
    }
}


// ---- FFDNet's FIRST layer on the same arithmetic, writing sp16: concatenate_input_noise_map (functions.py:16-53: sigma map + 2x2
// pixel-unshuffle, channel 2i+j) + conv3x3(5 -> 64, pad 1, no bias) + ReLU.  K = 5 channels x 9 taps = 45 (padded to 48 = three MFMA k
// steps); the B operand (k x 32 positions) is GATHERED: lane (position, k block) reads its eight taps from the full-resolution patch /
// the sigma plane in LDS, multiplies by 2^8 and splits them into hi + lo fp16 on the fly; the weight operands (12 fragments) stay in
// registers.  18 f16 MFMAs per 32 positions instead of 96 fp32 ones: the layer is left with its 256 B/position store.
#ifndef S16_HEAD_NT
#define S16_HEAD_NT 1        // (A/B) the p32 head's stores: 1 = nt (streaming), 0 = default policy (the lines stay in the XCD's L2 / the Infinity Cache)
#endif
#ifndef S16_HEAD_ROWS
#define S16_HEAD_ROWS 8      // (A/B) rows of half-resolution positions per workgroup tile of head_s16_kernel
#endif
constexpr int HS_H = S16_HEAD_ROWS, HS_W = 32, HS_P = 2 * HS_H + 4, HS_Q = 2 * HS_W + 4, HS_QS = HS_Q + 2, HS_SW = HS_W + 2, HS_SS = HS_SW + 1;   // 20 x 68 patch, 10 x 34 sigma plane
template <int TRACK, int P32>   // TRACK = 1: additionally folds max |output| into *track (the measuring launch of the first f-call; the per-value maximum
                                // costs the gather-bound kernel a quarter of its time, so the other 180 calls run without it); P32 = 1: the output is
                                // "p32" (csrc/conv_w16.hip): 2^e y as fp32, the lane's four consecutive couts = its pixel's 16 bytes of plane 2 b8 + kb
__global__ __launch_bounds__(256) void head_s16_kernel(const float* __restrict__ x, const char* __restrict__ Wp, const float* __restrict__ sigma,
                                                       int sigma_stride, char* __restrict__ y, int H, int W, int w_exp, const float* __restrict__ in_amax,
                                                       int in_exp, const float* __restrict__ out_amax, int out_exp, float* __restrict__ track) {
    // (round 6) the patch holds every element ALREADY scaled by 2^e_in and split: one word = its hi piece (low half) and its lo piece (high half).
    // An element is gathered ~4.5 times (nine taps of the positions around it); multiplying and splitting it at every gather was 96 of the
    // kernel's ~160 vector instructions per row of 32 positions - and the kernel is issue-bound (it runs right behind the power-capped stack
    // launch, at its clock: 28 us alone, 39 us in the loop).  Same operations on the same values: the same bits.
    __shared__ __attribute__((aligned(16))) unsigned patch[HS_P * HS_QS + (HS_H + 2) * HS_SS];
    __shared__ uint32_t trk_s[4];
    constexpr int SGM = HS_P * HS_QS;
    const int n = blockIdx.z, r0 = blockIdx.y * HS_H, c0 = blockIdx.x * HS_W;
    const int H2 = 2 * H, W2 = 2 * W;
    const float* xn = x + (int64_t)n * H2 * W2;
    const float sig = sigma[(int64_t)n * sigma_stride];
    // the operand gathered below holds 2^e_in (image | sigma): *in_amax is max |image| (the sigma plane is this kernel's own business)
    const int e_in = in_amax ? sp16_act_exp(fmaxf(in_amax[n], __builtin_fabsf(sig))) : in_exp, e_out = out_amax ? sp16_act_exp(out_amax[n]) : out_exp;
    const float in_scale = sp16_pow2(e_in), oscale = sp16_pow2(e_out - e_in - w_exp);
    float tmax = 0.0f;
    // full-resolution pixels (2 r0 - 2 + pr, 2 c0 - 2 + pc): half-res position (r, c), sub-pixel (i, j), tap (dy, dx) reads
    // (2 (r + dy - 1) + i, 2 (c + dx - 1) + j); zero outside the image = the conv's zero padding of the unshuffled channels
    auto patch_px = [&](int e) __attribute__((always_inline)) -> float {
        const int pr = e / HS_Q, pc = e - pr * HS_Q;
        const int gr = 2 * r0 - 2 + pr, gc = 2 * c0 - 2 + pc;
        return (e < HS_P * HS_Q && gr >= 0 && gr < H2 && gc >= 0 && gc < W2) ? xn[(int64_t)gr * W2 + gc] * in_scale : 0.0f;
    };
    for (int e = threadIdx.x; e < HS_P * HS_Q; e += 512) {      // two elements per trip: one split_pair
        const int e1 = e + 256;
        unsigned hh, ll;
        split_pair(patch_px(e), patch_px(e1), hh, ll);
        patch[(e / HS_Q) * HS_QS + e % HS_Q] = (hh & 0xffffu) | (ll << 16);
        if (e1 < HS_P * HS_Q) patch[(e1 / HS_Q) * HS_QS + e1 % HS_Q] = (hh >> 16) | (ll & 0xffff0000u);
    }
    {
        unsigned hh, ll;
        split_pair(sig * in_scale, 0.0f, hh, ll);
        const unsigned sgw = (hh & 0xffffu) | (ll << 16);
        for (int e = threadIdx.x; e < (HS_H + 2) * HS_SW; e += 256) {
            const int pr = e / HS_SW, pc = e - pr * HS_SW;
            const int rr = r0 - 1 + pr, cc = c0 - 1 + pc;
            patch[SGM + pr * HS_SS + pc] = (rr >= 0 && rr < H && cc >= 0 && cc < W) ? sgw : 0u;
        }
    }
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6), pl = lane & 31, kb = lane >> 5;
    // this lane's 24 taps: k = 16 ks + 8 kb + j = 9 ch + tap (k >= 45: zero weight, any address): LDS float offset relative to the
    // position's patch origin (2 lr, 2 lc) / sigma origin (lr, lc)
    int off[3][8];
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int k = 16 * ks + 8 * kb + j, kk = k < 45 ? k : 0;
            const int ch = kk / 9, tap = kk - 9 * ch, dy = tap / 3, dx = tap - 3 * dy;
            off[ks][j] = ch == 0 ? SGM + dy * HS_SS + dx + pl : (2 * dy + ((ch - 1) >> 1)) * HS_QS + 2 * dx + ((ch - 1) & 1) + 2 * pl;
        }
    h8 Aw[3][2][2];                                            // weights: [k step][hi, lo][cout group]
#pragma unroll
    for (int ks = 0; ks < 3; ++ks)
#pragma unroll
        for (int hl = 0; hl < 2; ++hl)
#pragma unroll
            for (int g = 0; g < 2; ++g) Aw[ks][hl][g] = *reinterpret_cast<const h8*>(Wp + ((((ks * 2 + hl) * 2 + g) * 64 + lane) * 16));
    __syncthreads();
    const int Wq = P32 ? 64 * ((W + 63) / 64) : W;             // row pitch of an output plane
    const int64_t HW = (int64_t)H * Wq;
    i32x4 orsrc;
    {
        const uint64_t ob = (uint64_t)(y + (int64_t)n * HW * 256);
        orsrc.x = (int)uniform((uint32_t)ob);
        orsrc.y = (int)uniform((uint32_t)(ob >> 32));
        orsrc.z = (int)uniform((uint32_t)(HW * 256));
        orsrc.w = 0x00020000;
    }
#pragma unroll 1
    for (int lr = wave; lr < HS_H; lr += 4) {
        const int sbase = lr * HS_SS, pbase = 2 * lr * HS_QS;
        h8 Bh[3], Bl[3];
#pragma unroll
        for (int ks = 0; ks < 3; ++ks) {
            unsigned wd[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = 16 * ks + 8 * kb + j;
                const bool is_sigma = (k < 45 ? k : 0) < 9;      // (sigma taps live in their own plane; per lane: k depends on its k block)
                wd[j] = patch[off[ks][j] + (is_sigma ? sbase : pbase)];
            }
            u32x4 bh, bl;
#pragma unroll
            for (int e = 0; e < 4; ++e) {                       // hi pieces = the words' low halves, lo pieces = their high halves: one v_perm_b32 per pair
                bh[e] = __builtin_amdgcn_perm(wd[2 * e + 1], wd[2 * e], 0x05040100u);
                bl[e] = __builtin_amdgcn_perm(wd[2 * e + 1], wd[2 * e], 0x07060302u);
            }
            Bh[ks] = __builtin_bit_cast(h8, bh);
            Bl[ks] = __builtin_bit_cast(h8, bl);
        }
        f32x16 acc[2];                                          // (K = 48: nine MFMAs per accumulator - one chain, cross terms first)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[g][i] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aw[ks][1][g], Bh[ks], acc[g], 0, 0, 0);
                acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aw[ks][0][g], Bl[ks], acc[g], 0, 0, 0);
            }
#pragma unroll
        for (int ks = 0; ks < 3; ++ks)
#pragma unroll
            for (int g = 0; g < 2; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Aw[ks][0][g], Bh[ks], acc[g], 0, 0, 0);
        // D[g][i]: cout 32 g + 8 (i >> 2) + 4 kb + (i & 3) of position (r0 + lr, c0 + pl): ReLU, x 2^8, split, lane exchange, sp16 stores
        const int r = r0 + lr, c = c0 + pl;
        const uint32_t pix = !(r < H && c < W) ? RAW_OOB
                             : P32 ? (uint32_t)((kb * (int)HW + r * Wq + (c & ~63) + ((c & 1) << 5) + ((c & 63) >> 1)) * 16)
                                   : (uint32_t)((kb * (int)HW + r * W + c) * 16);
        if (P32) {
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    f32x4 t;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        t[k] = __builtin_elementwise_maximum(acc[g][4 * gq + k] * oscale, 0.0f);
                        if (TRACK && pix != RAW_OOB) tmax = fmaxf(tmax, t[k]);
                    }
                    const uint32_t so = uniform((uint32_t)(2 * (4 * g + gq)) * (uint32_t)HW * 16u);
#if S16_HEAD_NT
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" ::"v"(t), "v"(pix), "s"(orsrc), "s"(so) : "memory");
#else
                    asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(t), "v"(pix), "s"(orsrc), "s"(so) : "memory");
#endif
                }
            continue;
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                unsigned hi[4], lo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = 4 * (2 * gp + (e >> 1)) + 2 * (e & 1);
                    f32x2 t = (f32x2){acc[g][i], acc[g][i + 1]} * (f32x2){oscale, oscale};
                    t.x = __builtin_elementwise_maximum(t.x, 0.0f);
                    t.y = __builtin_elementwise_maximum(t.y, 0.0f);
                    if (TRACK && pix != RAW_OOB) tmax = fmaxf(tmax, fmaxf(t.x, t.y));               // (positions of the image only)
                    const h2 hh = __builtin_convertvector(t, h2);
                    hi[e] = __builtin_bit_cast(unsigned, hh);
                    lo[e] = __builtin_bit_cast(unsigned, __builtin_convertvector(t - __builtin_convertvector(hh, f32x2), h2));
                }
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    auto sh = __builtin_amdgcn_permlane32_swap(hi[e], hi[2 + e], false, false);
                    hi[e] = sh[0]; hi[2 + e] = sh[1];
                    auto sl = __builtin_amdgcn_permlane32_swap(lo[e], lo[2 + e], false, false);
                    lo[e] = sl[0]; lo[2 + e] = sl[1];
                }
                const u32x4 oh = {hi[0], hi[1], hi[2], hi[3]}, ol = {lo[0], lo[1], lo[2], lo[3]};
                const uint32_t so_h = uniform((uint32_t)((2 * g + gp) * 4 + 0) * (uint32_t)HW * 16u);
                const uint32_t so_l = uniform((uint32_t)((2 * g + gp) * 4 + 2) * (uint32_t)HW * 16u);
                asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" ::"v"(oh), "v"(pix), "s"(orsrc), "s"(so_h) : "memory");
                asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen nt\n\ts_nop 1" ::"v"(ol), "v"(pix), "s"(orsrc), "s"(so_l) : "memory");
            }
    }
    if (TRACK) sp16_track_block_max(tmax, sp16_pow2(-e_out), track + n, trk_s);
}

}  // namespace s16
}  // namespace deqsci

using namespace deqsci;

static void s16_magic(uint32_t d, uint32_t* mg, uint32_t* sh) {
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    *sh = 31 + s;
    *mg = (uint32_t)(((1ull << (31 + s)) + d - 1) / d);
}

static bool bad_exp(int e) { return e < -SP16_EXP_LIMIT || e > SP16_EXP_LIMIT; }

extern "C" int deqsci_conv3x3_c64_split16(const void* x_sp16, const void* w_packed, const float* bias, void* y, int64_t n, int64_t H, int64_t W,
                                          int relu, int w_exp, const float* in_amax, int in_exp, const float* out_amax, int out_exp,
                                          float* track_amax, int out_f32, deqsci_stream_t stream, void* start_event, void* stop_event) {
    if (!x_sp16 || !w_packed || (!y && !track_amax)) return DEQSCI_ERR_NULL;             // (a measuring launch writes no y: it may be NULL)
    if ((start_event == nullptr) != (stop_event == nullptr)) return DEQSCI_ERR_NULL;
    if (n <= 0 || H <= 0 || W <= 0) return DEQSCI_ERR_SHAPE;
    if (x_sp16 == y || (out_f32 != 0 && out_f32 != 1) || bad_exp(w_exp) || bad_exp(in_exp) || bad_exp(out_exp)) return DEQSCI_ERR_UNSUPPORTED;
    if (track_amax && out_f32) return DEQSCI_ERR_UNSUPPORTED;        // (the range measurement serves the sp16 output)
    if (!aligned16(x_sp16) || !aligned16(w_packed) || !aligned16(y)) return DEQSCI_ERR_ALIGN;
    const int64_t tiles_x = ceil_div(W, s16::OUT_COLS), tiles_y = ceil_div(H, s16::OUT_ROWS);
    const int64_t n_tiles = n * tiles_x * tiles_y;
    // 32-bit byte offsets inside one image, and the out-of-range sentinel 2^31 must lie beyond the descriptor's range
    if (n_tiles > (int64_t)INT32_MAX / 16 || H * W * 256 + s16::RAW_BIAS + 4096 + 16 > (int64_t)s16::RAW_OOB) return DEQSCI_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t resident = (int64_t)num_cus();
    const dim3 grid((unsigned)(n_tiles < resident ? n_tiles : resident));
    uint32_t mg_img, sh_img, mg_tx, sh_tx;
    s16_magic((uint32_t)(tiles_x * tiles_y), &mg_img, &sh_img);
    s16_magic((uint32_t)tiles_x, &mg_tx, &sh_tx);
    hipEvent_t ev0 = static_cast<hipEvent_t>(start_event), ev1 = static_cast<hipEvent_t>(stop_event);
#define S16_LAUNCH(KERNEL)                                                                                                                  \
    hipExtLaunchKernelGGL(KERNEL, grid, dim3(s16::TBW), 0, st, ev0, ev1, 0, static_cast<const char*>(x_sp16), static_cast<const char*>(w_packed), \
                          bias, static_cast<char*>(y), (int)H, (int)W, relu, w_exp, in_amax, in_exp, out_amax, out_exp, track_amax, (int)tiles_x,       \
                          (int)tiles_y, (int)n_tiles, mg_img, sh_img, mg_tx, sh_tx, static_cast<char*>(nullptr),                                        \
                          static_cast<const s16::StackLayer*>(nullptr), 1, static_cast<unsigned*>(nullptr), 0)
    if (out_f32) S16_LAUNCH((s16::conv_s16_kernel<1, 0, 0>));
    else if (track_amax) S16_LAUNCH((s16::conv_s16_kernel<0, 1, 0>));
    else S16_LAUNCH((s16::conv_s16_kernel<0, 0, 0>));
#undef S16_LAUNCH
    return launch_status();
}

static_assert(sizeof(s16::StackLayer) == 24, "the layer table of deqsci_conv3x3_c64_split16_stack is three 8-byte words per layer");

extern "C" int deqsci_conv3x3_c64_split16_stack(const void* x_sp16, void* y_even, void* y_odd, const void* layers, int n_layers,
                                                int64_t n, int64_t H, int64_t W, const float* ranges, int64_t range_stride, int in_exp, int out_exp,
                                                void* flags, deqsci_stream_t stream, void* start_event, void* stop_event) {
    if (!x_sp16 || !y_even || !layers || !flags || (n_layers > 1 && !y_odd)) return DEQSCI_ERR_NULL;
    if ((start_event == nullptr) != (stop_event == nullptr)) return DEQSCI_ERR_NULL;
    if (n <= 0 || H <= 0 || W <= 0 || n_layers <= 0 || (ranges && (range_stride < n || range_stride > INT32_MAX))) return DEQSCI_ERR_SHAPE;
    if (x_sp16 == y_even || x_sp16 == y_odd || y_even == y_odd || n_layers > 64 || bad_exp(in_exp) || bad_exp(out_exp)) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(x_sp16) || !aligned16(y_even) || !aligned16(y_odd) || (reinterpret_cast<uintptr_t>(layers) & 7u) || (reinterpret_cast<uintptr_t>(flags) & 3u))
        return DEQSCI_ERR_ALIGN;
    const int64_t tiles_x = ceil_div(W, s16::OUT_COLS), tiles_y = ceil_div(H, s16::OUT_ROWS);
    const int64_t n_tiles = n * tiles_x * tiles_y;
    if (H * W * 256 + s16::RAW_BIAS + 4096 + 16 > (int64_t)s16::RAW_OOB) return DEQSCI_ERR_UNSUPPORTED;
    // every workgroup of the launch has to be RESIDENT (they wait for one another): one per CU - the kernel's 152 KB of LDS admit no
    // second one - so never more workgroups than CUs; each walks its tiles layer after layer
    if (n_tiles > (int64_t)INT32_MAX / (16 * 32)) return DEQSCI_ERR_UNSUPPORTED;
    static int occ_cache[64] = {0};
    const int64_t fit = resident_workgroups(s16::conv_s16_kernel<0, 0, 1>, s16::TBW, occ_cache);
    if (fit <= 0) return DEQSCI_ERR_UNSUPPORTED;                // (the waits inside the launch need every workgroup resident: ask the runtime, do not assume)
    const int64_t resident = fit < (int64_t)num_cus() ? fit : (int64_t)num_cus();
    hipStream_t st = static_cast<hipStream_t>(stream);
    uint32_t mg_img, sh_img, mg_tx, sh_tx;
    s16_magic((uint32_t)(tiles_x * tiles_y), &mg_img, &sh_img);
    s16_magic((uint32_t)tiles_x, &mg_tx, &sh_tx);
    hipEvent_t ev0 = static_cast<hipEvent_t>(start_event), ev1 = static_cast<hipEvent_t>(stop_event);
    hipExtLaunchKernelGGL((s16::conv_s16_kernel<0, 0, 1>), dim3((unsigned)(n_tiles < resident ? n_tiles : resident)), dim3(s16::TBW), 0, st, ev0, ev1, 0, static_cast<const char*>(x_sp16),
                          static_cast<const char*>(nullptr), static_cast<const float*>(nullptr), static_cast<char*>(y_even), (int)H, (int)W, 0, 0, ranges,
                          in_exp, static_cast<const float*>(nullptr), out_exp, static_cast<float*>(nullptr), (int)tiles_x, (int)tiles_y, (int)n_tiles,
                          mg_img, sh_img, mg_tx, sh_tx, static_cast<char*>(y_odd), static_cast<const s16::StackLayer*>(layers), n_layers,
                          static_cast<unsigned*>(flags), (int)range_stride);
    return launch_status();
}

extern "C" int deqsci_f32_to_split16(const float* x_nhwc, void* y_sp16, int64_t n, int64_t H, int64_t W, const float* amax, int exp,
                                     deqsci_stream_t stream) {
    if (!x_nhwc || !y_sp16) return DEQSCI_ERR_NULL;
    if (n <= 0 || H <= 0 || W <= 0) return DEQSCI_ERR_SHAPE;
    if (bad_exp(exp)) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(x_nhwc) || !aligned16(y_sp16)) return DEQSCI_ERR_ALIGN;
    const int64_t total = n * H * W * 8;
    hipLaunchKernelGGL(s16::f32_to_sp16_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), x_nhwc,
                       static_cast<char*>(y_sp16), H * W, total, amax, exp);
    return launch_status();
}

extern "C" int deqsci_absmax_f32(const float* x, int64_t n, int64_t count, float* amax, deqsci_stream_t stream) {
    if (!x || !amax) return DEQSCI_ERR_NULL;
    if (count <= 0 || n <= 0) return DEQSCI_ERR_SHAPE;
    if (n > 65535) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(x)) return DEQSCI_ERR_ALIGN;
    int64_t blocks = ceil_div(count, 256 * 4 * 8);                   // eight float4 per lane
    const int64_t cap = ceil_div(8 * (int64_t)num_cus(), n);
    blocks = blocks < cap ? blocks : cap;
    hipLaunchKernelGGL(s16::absmax_kernel, dim3((unsigned)(blocks < 1 ? 1 : blocks), (unsigned)n), dim3(256), 0, static_cast<hipStream_t>(stream), x, count, amax);
    return launch_status();
}

template <int COUT, int P32 = 0>
static int tail_s16_impl(const void* x_sp16, const void* w_packed, float* out, int64_t n, int64_t H, int64_t W, int w_exp, const float* in_amax,
                         int in_exp, deqsci_stream_t stream) {
    if (!x_sp16 || !w_packed || !out) return DEQSCI_ERR_NULL;
    if (n <= 0 || H <= 0 || W <= 0) return DEQSCI_ERR_SHAPE;
    if (n > 65535 || H > (1 << 20) || W > (1 << 20) || bad_exp(w_exp) || bad_exp(in_exp)) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(x_sp16) || !aligned16(w_packed) || !aligned16(out)) return DEQSCI_ERR_ALIGN;
    const int64_t tiles_x = ceil_div(W, s16::TL_W), tiles_y = ceil_div(H, s16::TL_H), n_tiles = n * tiles_x * tiles_y;
    if (n_tiles > (1 << 30)) return DEQSCI_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)(8 * ceil_div(n_tiles, 8)));
    if (P32 && H * (64 * ceil_div(W, 64)) * 256 + 16 > (int64_t)s16::RAW_OOB) return DEQSCI_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((s16::tail_s16_kernel<COUT, P32>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const char*>(x_sp16),
                       static_cast<const char*>(w_packed), out, (int)H, (int)W, w_exp, in_amax, in_exp, (int)tiles_x, (int)tiles_y, (int)n_tiles);
    return launch_status();
}

extern "C" int deqsci_ffdnet_tail_split16(const void* x_sp16, const void* w_packed, float* out, int64_t n, int64_t H, int64_t W, int w_exp,
                                          const float* in_amax, int in_exp, deqsci_stream_t stream) {
    return tail_s16_impl<4>(x_sp16, w_packed, out, n, H, W, w_exp, in_amax, in_exp, stream);
}

extern "C" int deqsci_ffdnet_tail_p32(const void* x_p32, const void* w_packed, float* out, int64_t n, int64_t H, int64_t W, int w_exp,
                                      const float* in_amax, int in_exp, deqsci_stream_t stream) {
    return tail_s16_impl<4, 1>(x_p32, w_packed, out, n, H, W, w_exp, in_amax, in_exp, stream);
}

extern "C" int deqsci_conv3x3_c64_to_1_p32(const void* x_p32, const void* w_packed, float* out, int64_t n, int64_t H, int64_t W, int w_exp,
                                           const float* in_amax, int in_exp, deqsci_stream_t stream) {
    return tail_s16_impl<1, 1>(x_p32, w_packed, out, n, H, W, w_exp, in_amax, in_exp, stream);
}

extern "C" int deqsci_conv3x3_c64_to_1_split16(const void* x_sp16, const void* w_packed, float* out, int64_t n, int64_t H, int64_t W, int w_exp,
                                               const float* in_amax, int in_exp, deqsci_stream_t stream) {
    return tail_s16_impl<1>(x_sp16, w_packed, out, n, H, W, w_exp, in_amax, in_exp, stream);
}

template <int P32>
static int head_s16_impl(const float* x, const void* w_packed, const float* sigma, int64_t sigma_stride, void* h_out,
                         int64_t n, int64_t H, int64_t W, int w_exp, const float* in_amax, int in_exp, const float* out_amax,
                         int out_exp, float* track_amax, deqsci_stream_t stream) {
    if (!x || !w_packed || !sigma || !h_out) return DEQSCI_ERR_NULL;
    if (n <= 0 || H <= 0 || W <= 0 || sigma_stride < 0) return DEQSCI_ERR_SHAPE;
    const int64_t pitch = P32 ? 64 * ceil_div(W, 64) : W;
    if (n > 65535 || H > (1 << 20) || W > (1 << 20) || H * pitch * 256 + 16 > (int64_t)s16::RAW_OOB) return DEQSCI_ERR_UNSUPPORTED;
    if (bad_exp(w_exp) || bad_exp(in_exp) || bad_exp(out_exp)) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(w_packed) || !aligned16(h_out)) return DEQSCI_ERR_ALIGN;
    const dim3 grid((unsigned)ceil_div(W, s16::HS_W), (unsigned)ceil_div(H, s16::HS_H), (unsigned)n);
    if (track_amax)
        hipLaunchKernelGGL((s16::head_s16_kernel<1, P32>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, static_cast<const char*>(w_packed), sigma,
                           (int)sigma_stride, static_cast<char*>(h_out), (int)H, (int)W, w_exp, in_amax, in_exp, out_amax, out_exp, track_amax);
    else
        hipLaunchKernelGGL((s16::head_s16_kernel<0, P32>), grid, dim3(256), 0, static_cast<hipStream_t>(stream), x, static_cast<const char*>(w_packed), sigma,
                           (int)sigma_stride, static_cast<char*>(h_out), (int)H, (int)W, w_exp, in_amax, in_exp, out_amax, out_exp, track_amax);
    return launch_status();
}

extern "C" int deqsci_ffdnet_head_split16(const float* x, const void* w_packed, const float* sigma, int64_t sigma_stride, void* h_sp16,
                                          int64_t n, int64_t H, int64_t W, int w_exp, const float* in_amax, int in_exp, const float* out_amax,
                                          int out_exp, float* track_amax, deqsci_stream_t stream) {
    return head_s16_impl<0>(x, w_packed, sigma, sigma_stride, h_sp16, n, H, W, w_exp, in_amax, in_exp, out_amax, out_exp, track_amax, stream);
}

extern "C" int deqsci_ffdnet_head_p32(const float* x, const void* w_packed, const float* sigma, int64_t sigma_stride, void* h_p32,
                                      int64_t n, int64_t H, int64_t W, int w_exp, const float* in_amax, int in_exp, const float* out_amax,
                                      int out_exp, deqsci_stream_t stream) {
    return head_s16_impl<1>(x, w_packed, sigma, sigma_stride, h_p32, n, H, W, w_exp, in_amax, in_exp, out_amax, out_exp, nullptr, stream);
}
