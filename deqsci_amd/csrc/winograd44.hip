// 3x3 convolution 64 -> 64 channels (pad 1, stride 1, fp32, channels_last) as Winograd F(4x4,3x3) on the fp32 matrix
// cores of MI355X, bias (folded BatchNorm) and ReLU fused.  Same layer as csrc/winograd.hip (networks/ffdnet/models.py:53-58,
// SimpleCNN_models.py:47-53), 2.25 multiplications per output instead of 4: the large-batch kernel.
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A      6x6 input patch d (stride 4) -> 4x4 outputs, 36 transform positions
//
// Block tile = 4 x 8 Winograd tiles (16 x 32 output pixels) x 64 couts, ONE persistent 8-wave workgroup per CU walking the
// block tiles of "its" XCD, input channels in chunks of 8 (as in winograd.hip).  What is different, and why:
//   * 36 positions x (16 tiles x 16 couts) accumulators do not fit one wave next to its operands, and a wave alone on a
//     SIMD cannot stream its weight operands from LDS at the matrix-pipe rate (44 cycles per MFMA instead of 32,
//     tools/ubench/mfma_f32_stage.hip).  So the 6 ROWS of the position grid are split between two waves: wave
//     (rg, cgp, tg) owns position rows [3 rg, 3 rg + 3) x couts [32 cgp, +32) x tiles [16 tg, +16): 18 x 2 accumulators
//     of v_mfma_f32_16x16x4_f32 (144 registers).  The split is free for the input transform - V = B^T d B restricted to
//     three rows of B^T is exactly half the work (column pass 6 x 6 ops, row pass 3 x 12) - so a wave spends ONE packed
//     vector instruction per MFMA on it (72 : 72), as F(2x2,3x3) does with its two cout halves.
//   * the output transform needs all 6 rows: each wave reduces ITS rows to the 4x4 partial result, keeps two output rows
//     and hands the other two to its partner wave through LDS (4 rounds of 32 KB per block tile, 7 barriers; a round is one
//     cout group and one PAIR of output columns, so that what is exchanged and stored per pixel is the float4 of four
//     consecutive couts and nothing is carried from round to round).
//   * one weight chunk is 36 x 64 x 8 floats = 72 KB: it is SINGLE-buffered and refilled in two halves behind two barriers
//     per stage (positions are consumed in the same order by every wave: after the first 10 of a wave's 18 positions
//     everyone has passed barrier X1 and the DMA may overwrite them with the next chunk, the other 8 after X2), each half
//     with half a stage of lead time.  Raw input tiles (18 x 34 pixels x 8 channels) stay double-buffered.
//   * the 6x6 patch is read from LDS while it is transformed (it would cost 72 registers to hold).  Raw tiles reach LDS by
//     the LDS-DMA path (buffer_load ... lds: lane-linear in LDS, a per-lane offset on the global side, zeros for lanes
//     outside the buffer), so the layout is chosen by WHICH pixel each lane fetches: the columns of one residue class mod 4
//     side by side and the channel halves swapped on every other group of four pixel rows - the patch reads of the 16 tiles of
//     a wave then hit every bank exactly once.
//   * REGISTERS decide everything here: 144 accumulators + 36 operand registers + the transform leave no room, and one
//     spilled register costs a scratch load AND an s_waitcnt vmcnt(0) in front of its use, i.e. a full memory round trip
//     per stage (measured: 330 us with ~20 spilled registers, 262 us without).  Hence: the two row groups run two separate
//     copies of prologue + loop (rg as a template argument: with a run-time rg hipcc keeps both arms of every rg-dependent
//     piece alive); the lane index and everything derived from it is recomputed at its use (v_mbcnt, volatile) instead of
//     being carried; the three per-lane DMA offsets of a block tile live in LDS; the MFMAs are inline asm with the
//     accumulator tied in place; the last stage of a tile transforms BEHIND the output transform.
//   * the output stores are 16 bytes per lane (64 contiguous bytes per pixel; 8-byte stores cost +115 us per launch),
//     non-temporal (kept in L2 they evict the input lines the launch re-reads for every cin chunk: -9 %), and they are
//     issued between the weight DMA and the raw DMA of the next stage so that no wait in the MFMA loop covers them.
// LDS: 72 KB weights + 2 x 21 KB raw + 32 KB exchange + 6 KB DMA offsets + bias = 152.3 KB.  Rounding: 1.2-1.7e-6 per layer
// against fp64 (F(2x2,3x3): 2.1e-7); end to end the FFDNet gates do not move (tools/f44_numerics.py,
// profiles/r02_f44_numerics.jsonl).  64 images of 128 x 128: 242-245 us against 303-314 us for F(2x2,3x3) on the same box; where the
// rest goes (tools/w44_variants.sh + w44_check.py, profiles/r02_w44_*): MFMA phases alone 158 us, input transform +50,
// weight DMA +10, raw DMA +15, output transform + exchange + stores +25.  (W44_ABL=1, "no raw DMA", once measured 190 us:
// hipcc had deleted the input transform together with it - a shared array that nothing writes.  Count the instructions of a
// stage before believing an ablation.)
#include "common.hpp"
#include <hip/hip_ext.h>
#include <type_traits>
#pragma clang diagnostic ignored "-Winline-asm"   // m0 is named as a clobber of the LDS-DMA asm below, on purpose

#ifndef W44_ABL
#define W44_ABL 0     // timing ablations only (results wrong; tools/w44_variants.sh): 1 = no raw DMA, 2 = no weight DMA, 16 = no input
                      // transform, 64 = no patch reads, 128 = raw DMA folded into 64 KB (cache hits), 4096 = no output stores (but
                      // everything in front of them), 8192 = no exchange between the two row groups
#endif

namespace deqsci {
namespace w44 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) f32x2 lds_f32x2;

constexpr int CK = 8, NCHUNK = 64 / CK;
constexpr int WAVES = 8, TBW = 64 * WAVES;
constexpr int TILE_ROWS = 4, TILE_COLS = 8;                  // Winograd tiles per block tile
constexpr int OUT_ROWS = 4 * TILE_ROWS, OUT_COLS = 4 * TILE_COLS;
constexpr int RAW_ROWS = OUT_ROWS + 2, RAW_COLS = OUT_COLS + 2;              // 18 x 34 staged pixels
constexpr int RAW_HALF_U = RAW_ROWS * RAW_COLS;              // 612 units of 16 bytes per channel half
constexpr int RAW_DMA = 21;                                  // DMA instructions of 64 units per chunk tile (1224 units used)
constexpr int RAW_BUF = RAW_DMA * 64 * 4;                    // 5376 floats = 21 KB (1224 units used)
constexpr uint32_t RAW_BIAS = 4096;                          // see set_fetch_tile
constexpr uint32_t RAW_OOB = 0x80000000u;                    // buffer offset of a pixel outside the image: beyond num_records -> zeros
constexpr int NSTEP = 18;                                    // positions per wave
constexpr int U_STEP = 2 * 2 * 64 * 4;                       // floats of one step in LDS: [rg][cgp][lane][j][ks]
constexpr int U_CHUNK = NSTEP * U_STEP;                      // 18432 floats = 72 KB
constexpr int SPLIT = 10;                                    // steps [0, SPLIT) are refilled behind X1, [SPLIT, 18) behind X2
constexpr int XCH = WAVES * 4 * 64 * 4;                      // exchange area: 4 x 16 bytes per lane and wave = 32 KB
constexpr int DMA1_PIECES = SPLIT * U_STEP * 4 / 1024 / WAVES;             // 5 KiB per wave
constexpr int DMA2_PIECES = (NSTEP - SPLIT) * U_STEP * 4 / 1024 / WAVES;   // 4 KiB per wave
static_assert(DMA1_PIECES == 5 && DMA2_PIECES == 4, "DMA split");

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
// The lane index, re-derived where it is needed (volatile: not hoisted, not kept): the kernel has no register to spare for
// carrying it - or anything computed from it - through the loop, and a spilled register costs a scratch load AND a vmcnt(0).
__device__ __forceinline__ int lane_id() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}
__device__ __forceinline__ int mdiv(int t, uint32_t mg, uint32_t sh) { return (int)(((uint64_t)(uint32_t)t * mg) >> sh); }
__device__ __forceinline__ f32x2 fma2(f32x2 a, float k, f32x2 c) { return __builtin_elementwise_fma(a, (f32x2){k, k}, c); }

// float offset of patch pixel (pr, pc) from patch pixel (0, 0) of the same tile in the raw layout (see the kernel)
__device__ __forceinline__ constexpr int patch_off(int pr, int pc) {
    return (pr * RAW_COLS + ((pc & 3) == 0 ? 0 : (pc & 3) == 1 ? 9 : (pc & 3) == 2 ? 18 : 26) + (pc >> 2)) * 8;
}

// One 16-byte non-temporal output store through a buffer descriptor: per-lane byte offset + scalar offset + immediate 64 k bytes
// (k is a constant after unrolling; the chain below folds to one instruction).  s_nop: the wait states between a 16-byte store and the
// next write of its data registers, which the compiler cannot see into the asm to insert.
#define W44_ST(K) asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen offset:" #K " nt\n\ts_nop 1" ::"v"(val), "v"(vo), "s"(rsrc), "s"(so) : "memory")
__device__ __forceinline__ void store_out(f32x4 val, uint32_t vo, i32x4 rsrc, uint32_t so, int k) {
    switch (k) {
        case 0: W44_ST(0); break;      case 1: W44_ST(64); break;    case 4: W44_ST(256); break;   case 5: W44_ST(320); break;
        case 8: W44_ST(512); break;    case 9: W44_ST(576); break;   case 12: W44_ST(768); break;  case 13: W44_ST(832); break;
        default: __builtin_trap();
    }
}
#undef W44_ST

// 1-D input transform of F(4,3): y = B^T x, B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
__device__ __forceinline__ void bt_lo(const f32x2* x, f32x2& y0, f32x2& y1, f32x2& y2) {
    y0 = fma2(x[2], -5.0f, fma2(x[0], 4.0f, x[4]));
    const f32x2 a = fma2(x[2], -4.0f, x[4]), b = fma2(x[1], -4.0f, x[3]);
    y1 = a + b;
    y2 = a - b;
}
__device__ __forceinline__ void bt_hi(const f32x2* x, f32x2& y3, f32x2& y4, f32x2& y5) {
    const f32x2 c = x[4] - x[2], d = x[3] - x[1];
    y3 = fma2(d, 2.0f, c);
    y4 = fma2(d, -2.0f, c);
    y5 = fma2(x[3], -5.0f, fma2(x[1], 4.0f, x[5]));
}

// IN_BLK / OUT_BLK: the activation is not channels_last (n,H,W,64) but "blk32": [n][cin chunk (8)][H][W/32][32][8] - planes of 8
// channels, and inside every block of 32 columns the pixels in the order this kernel stages them (column m of the block at
// position 8 ((m+1) & 3) + ((m+1) >> 2) - ((m+1) & 3 == 0)): a staged pixel row is then two contiguous runs of 512 bytes
// instead of 34 scattered 32-byte pieces, and an output store instruction writes 256 contiguous bytes per tile row instead of
// 64.  A stack of 64->64 layers runs NHWC -> blk32 -> ... -> blk32 -> NHWC; nothing else ever sees the layout.
template <int IN_BLK, int OUT_BLK>
__global__ __launch_bounds__(TBW, 2) void winograd44_conv64_kernel(const float* __restrict__ x, const float* __restrict__ Ug,
                                                                   const float* __restrict__ bias, float* __restrict__ y,
                                                                   int H, int W, int relu, int tiles_x, int tiles_y, int n_tiles,
                                                                   uint32_t mg_img, uint32_t sh_img, uint32_t mg_tx, uint32_t sh_tx) {
    __shared__ __attribute__((aligned(16))) float Us[U_CHUNK];
    __shared__ __attribute__((aligned(16))) float Raw[2 * RAW_BUF];
    __shared__ __attribute__((aligned(16))) float Xs[XCH];
    __shared__ __attribute__((aligned(16))) float bias_s[64];
    __shared__ uint32_t Voff[3 * TBW];
    const int wave = (int)uniform((uint32_t)(threadIdx.x >> 6));
    const int rg = wave & 1;
    int cgp, tg;                                              // (set inside each row group's code copy: see run())

    int t_first, t_step, t_end;
    {
        const int nb = (int)gridDim.x, b = (int)blockIdx.x;
        if ((nb & 7) == 0) {
            const int per_xcd = (n_tiles + 7) >> 3;
            t_first = (b & 7) * per_xcd + (b >> 3);
            t_step = nb >> 3;
            t_end = min(n_tiles, ((b & 7) + 1) * per_xcd);
        } else { t_first = b; t_step = nb; t_end = n_tiles; }
    }
    if (t_first >= t_end) return;

    // ---- raw staging by the LDS-DMA path: the chunk tile is 1224 units of 16 bytes (pixel, channel half) = 21 instructions
    // buffer_load_dwordx4 ... lds of 64 units; wave w < 7 issues instructions 3 w .. 3 w + 2 (wave 7 repeats wave 6's, so that
    // every wave has the same number of operations in flight).  The LDS side is lane-linear (M0 + 16 lane), the global side a
    // per-lane offset, so the tile's LAYOUT is chosen by which pixel each lane fetches:
    //     pixel slot(row, col) = 34 row + COLMAP(col),  COLMAP(col) = {0, 9, 18, 26}[col & 3] + (col >> 2)   (32 bytes per slot)
    //     unit = 2 slot + (half ^ ((row >> 2) & 1))
    // i.e. the columns of one residue class mod 4 are consecutive, the two lanes of a pixel fetch 32 contiguous bytes (one
    // request), and the channel halves are swapped on every other group of four rows.  A patch read touches, per half wave,
    // channel pairs 0, 1 of the pixels (R + 4 ty, C + 4 tx) of 16 tiles: float offsets 8 tx + 4 ty + 2 q mod 64 - every bank once.
    // Pixels outside the image get an offset beyond the buffer descriptor's range: the hardware writes zeros for them
    // (tools/ubench/buffer_lds_oob.hip), so the zero padding costs no instruction.
    // (the three per-lane buffer offsets of a block tile live in LDS, not in registers: the kernel has none to spare, and a
    // spilled register costs a scratch load AND an s_waitcnt vmcnt(0) - measured at ~80 us per launch)
    i32x4 rsrc;
    auto set_fetch_tile = [&](int t) {
        const int n = mdiv(t, mg_img, sh_img), r = t - n * (tiles_x * tiles_y);
        const int by = mdiv(r, mg_tx, sh_tx), bx = r - by * tiles_x;
        // the descriptor starts RAW_BIAS bytes BELOW the image: a wave's three instructions differ in their immediate offset
        // (which the hardware adds to the LDS and to the global address), the per-lane offsets take it back out
        const int64_t img_floats = IN_BLK ? (int64_t)H * tiles_x * 32 * 64 : (int64_t)H * W * 64;
        const uint64_t base = (uint64_t)(x + (int64_t)n * img_floats) - RAW_BIAS;
        rsrc.x = (int)uniform((uint32_t)base);
        rsrc.y = (int)uniform((uint32_t)(base >> 32));            // stride 0: raw buffer, offsets in bytes
        rsrc.z = (int)uniform((uint32_t)img_floats * 4u + RAW_BIAS);   // num_records (< 2^32: launcher)
        rsrc.w = 0x00020000;
        const int py0 = OUT_ROWS * by - 1, px0 = OUT_COLS * bx - 1;   // image coordinates of staged pixel (0,0)
        const int el = lane_id();
        const int w = wave < 7 ? wave : 6;                         // 21 instructions cover the tile: wave 7 repeats wave 6
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int u = 64 * (3 * w + j) + el, slot = u >> 1;
            const int row = (slot * 1928) >> 16, cu = slot - row * RAW_COLS;      // slot / 34 for slot < 768
            const int col = cu < 9 ? 4 * cu : (cu < 18 ? 4 * (cu - 9) + 1 : (cu < 26 ? 4 * (cu - 18) + 2 : 4 * (cu - 26) + 3));
            const int half = (u & 1) ^ ((row >> 2) & 1);
            const int iy = py0 + row, ix = px0 + col;
            const bool ok = slot < RAW_HALF_U && iy >= 0 && iy < H && ix >= 0 && ix < W;
            uint32_t pix_bytes;                                    // of the pixel's 8 channels of chunk 0 inside the image
            if (IN_BLK) {
                const int m1 = (ix & 31) + 1;
                pix_bytes = (uint32_t)((iy * tiles_x + (ix >> 5)) * 32 + 8 * (m1 & 3) + (m1 >> 2) - ((m1 & 3) == 0)) * 32u;
            } else {
                pix_bytes = (uint32_t)(iy * W + ix) * 256u;
            }
            uint32_t vo = ok ? pix_bytes + 16u * (uint32_t)half + (RAW_BIAS - 1024u * j) : RAW_OOB;
#if W44_ABL & 128     // timing experiment (wrong results): the same scatter pattern, folded into the first 64 KB of the image (cache hits)
            vo = ok ? vo & 0xffffu : RAW_OOB;
#endif
            if (!(W44_ABL & 1)) Voff[j * TBW + wave * 64 + el] = vo;
        }
    };
    const uint32_t raw_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)Raw;
    // chunk c of the fetch tile -> Raw[buf]: the scalar offset selects the 8 channels (32 bytes) of the chunk; ONE M0 value per
    // wave (writing M0 between two LDS-DMA instructions costs ~100 cycles each: measured)
    // One chunk tile = three instructions per wave.  Inside the pipeline they are issued ONE AT A TIME from within an MFMA part
    // (raw_begin, then raw_piece(j) between MFMAs): all eight waves issuing their three right behind a barrier kept every wave
    // waiting at the memory pipeline's door for ~1000 cycles per stage with the matrix pipe idle (tools/w44_stamps.py).
    struct RawDma { uint32_t soff, m0v, v[3]; };
    auto raw_begin = [&](int c, int buf) -> RawDma {
        RawDma d;
        d.soff = uniform(IN_BLK ? (uint32_t)c * (uint32_t)(H * tiles_x) * 1024u : (uint32_t)c * (CK * 4));   // chunk plane / channel offset
        int w = wave;
        asm volatile("" : "+s"(w));                                // recompute the M0 value here (scalar ALU is free; SGPRs are not)
        d.m0v = uniform(raw_lds + (uint32_t)(buf * RAW_BUF * 4 + 3 * (w < 7 ? w : 6) * 1024));
        const int vl = w * 64 + lane_id();
        d.v[0] = Voff[vl]; d.v[1] = Voff[TBW + vl]; d.v[2] = Voff[2 * TBW + vl];
        return d;
    };
    auto raw_piece = [&](const RawDma& d, int j) __attribute__((always_inline)) {
        if (W44_ABL & 1) return;
        if (j == 0) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(d.m0v), "v"(d.v[0]), "s"(rsrc), "s"(d.soff) : "memory", "m0");
        else if (j == 1) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:1024 lds" ::"s"(d.m0v), "v"(d.v[1]), "s"(rsrc), "s"(d.soff) : "memory", "m0");
        else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen offset:2048 lds" ::"s"(d.m0v), "v"(d.v[2]), "s"(rsrc), "s"(d.soff) : "memory", "m0");
    };
    auto dma_raw = [&](int c, int buf) {                           // (prologue: all three at once)
        const RawDma d = raw_begin(c, buf);
        raw_piece(d, 0); raw_piece(d, 1); raw_piece(d, 2);
    };

    // ---- weight chunk: host-packed in LDS order; half 1 = bytes [0, 40 KiB), half 2 = [40 KiB, 72 KiB); wave w moves a
    // contiguous share of each half by the LDS-DMA path, pieces of 1 KiB that differ only in their immediate offset
    const uint32_t us_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)Us;
    auto dma_setup = [&](int c, int half) -> uint64_t {
        int w = wave;
        asm volatile("" : "+s"(w));
        const uint32_t share = half == 0 ? (uint32_t)(w * DMA1_PIECES * 1024 + 2048)
                                         : (uint32_t)(SPLIT * U_STEP * 4 + w * DMA2_PIECES * 1024 + 2048);
        const uint64_t g = (uint64_t)(Ug + (int64_t)c * U_CHUNK) + share;
        asm volatile("s_mov_b32 m0, %0" ::"s"(uniform(us_lds + share)) : "m0");
        return ((uint64_t)uniform((uint32_t)(g >> 32)) << 32) | uniform((uint32_t)g);
    };
#define W44_DMA(OFF) asm volatile("global_load_lds_dwordx4 %0, %1 offset:" #OFF ::"v"(dma_voff), "s"(dg) : "memory")
    auto dma_half = [&](int c, int half) {
        const uint64_t dg = dma_setup(c, half);
        const int dl = lane_id();
        const uint32_t dma_voff = (uint32_t)(dl * 16);
        if (W44_ABL & 2) return;
        W44_DMA(-2048); W44_DMA(-1024); W44_DMA(0); W44_DMA(1024);
        if (half == 0) W44_DMA(2048);
    };

    f32x4 acc[NSTEP][2];
    bool stores_in_flight = false;                            // (uniform) the previous tile's epilogue issued exactly 16 stores per wave

    // MFMA roles: lane (i = lane&15, q = lane>>4) owns tile 16 tg + i and channels {2q, 2q+1} of the chunk
    f32x2 v[NSTEP];
    // V = B^T d B, rows [3 rg, 3 rg + 3) only
    auto transform = [&](auto rg_c, int buf) __attribute__((always_inline)) {
        constexpr int RG = decltype(rg_c)::value;
        // (patch reads are VOLATILE LDS loads: hipcc otherwise fuses pairs of them into ds_read2_b64, which is banked mod 32 over 16-lane
        // groups - the layout below is conflict-free for ds_read_b64's 64 banks over 32 lanes and 2-way conflicted for the fused form)
        const lds_float* pp = (const lds_float*)Raw + buf * RAW_BUF;
        // float offset of channel pair q of patch pixel (0,0) of the lane's tile in a raw buffer; patch pixel (pr, pc) is a constant
        // away, with the channel halves the other way round in patch rows 4, 5 (they belong to the next group of four pixel rows).
        // Recomputed here from an opaque copy of the lane index instead of being carried through the loop.
        const int tl = lane_id();
        const int q_ = tl >> 4, ty_ = (tl >> 3) & 1, tx_ = tl & 7;
        const int pbase0 = (4 * (2 * tg + ty_) * RAW_COLS + tx_) * 8 + 2 * (q_ & 1);
        const int pbaseA = pbase0 + 4 * ((q_ >> 1) ^ ty_), pbaseB = pbase0 + 4 * ((q_ >> 1) ^ ty_ ^ 1);
        f32x2 t[3][6];
        if (RG == 0) {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                f32x2 d[6];
#pragma unroll
                for (int pr = 0; pr < 5; ++pr) d[pr] = (W44_ABL & 64) ? v[pr + j] : *reinterpret_cast<const volatile lds_f32x2*>(pp + (pr < 4 ? pbaseA : pbaseB) + patch_off(pr, j));
                bt_lo(d, t[0][j], t[1][j], t[2][j]);          // (rows 0..2 of B^T do not touch patch row 5, rows 3..5 not row 0: a volatile load is not dropped for being unused)
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                f32x2 d[6];
#pragma unroll
                for (int pr = 1; pr < 6; ++pr) d[pr] = (W44_ABL & 64) ? v[pr + j] : *reinterpret_cast<const volatile lds_f32x2*>(pp + (pr < 4 ? pbaseA : pbaseB) + patch_off(pr, j));
                bt_hi(d, t[0][j], t[1][j], t[2][j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            bt_lo(t[r], v[r * 6], v[r * 6 + 1], v[r * 6 + 2]);
            bt_hi(t[r], v[r * 6 + 3], v[r * 6 + 4], v[r * 6 + 5]);
        }
    };

    // ---- output transform Y = A^T M A, A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1].  Each wave reduces its
    // three rows of M: W = M A (3 x 4), then the partial A^T W over its rows (4 x 4); rg 0 finishes output rows 0, 1 and rg 1
    // rows 2, 3, the other two rows of the partial go to the partner wave (wave ^ 1) through Xs.  The bias is already inside:
    // position (1,1) has coefficient 1 in all 16 outputs and its accumulator (rg 0, step 7) starts from the bias.
    auto epilogue = [&](auto rg_c, int t) __attribute__((always_inline)) {
        constexpr int RG = decltype(rg_c)::value;
        const int n = mdiv(t, mg_img, sh_img), r = t - n * (tiles_x * tiles_y);
        const int by = mdiv(r, mg_tx, sh_tx), bx = r - by * tiles_x;
        // Output stores go through a raw buffer descriptor of image n: a lane's address is then ONE 32-bit offset (image bytes < 2^31:
        // launcher) instead of a 64-bit pointer plus the temporaries of its arithmetic, and everything wave-uniform (cout group,
        // channel plane, output row of the pair) rides in the scalar offset - the 64-bit form spilled 8 registers per lane here.
        const uint32_t plane_b = (uint32_t)(H * tiles_x) * 1024u;      // bytes of one 8-channel plane of the blk32 layout
        const int64_t oimg = OUT_BLK ? (int64_t)H * tiles_x * 2048 : (int64_t)H * W * 64;     // floats per image
        i32x4 orsrc;
        {
            const uint64_t ob = (uint64_t)(y + (int64_t)n * oimg);
            orsrc.x = (int)uniform((uint32_t)ob);
            orsrc.y = (int)uniform((uint32_t)(ob >> 32));
            orsrc.z = (int)uniform((uint32_t)oimg * 4u);
            orsrc.w = 0x00020000;
        }
        asm volatile("s_nop 15");                              // (asm MFMAs: the wait states between the last of them and the first vector read of an accumulator)
        stores_in_flight = OUT_ROWS * (by + 1) <= H && OUT_COLS * (bx + 1) <= W;   // every lane stores all 16 values: 16 operations in flight
        // every lane-dependent address of the epilogue is derived from an opaque copy of the lane index: hipcc would otherwise
        // hoist them out of the persistent loop and keep a dozen registers alive through the MFMA stages (= spills there)
        const int el = lane_id();
        const int ei = el & 15, eq = el >> 4;
        const int oy = OUT_ROWS * by + 4 * (2 * tg + (ei >> 3)) + 2 * RG, ox = OUT_COLS * bx + 4 * (ei & 7);
        const uint32_t ovo = OUT_BLK ? (uint32_t)(eq >> 1) * plane_b + (uint32_t)(((oy * tiles_x + bx) * 32 + (ei & 7)) * 32 + 16 * (eq & 1))
                                     : (uint32_t)((oy * W + ox) * 256 + 16 * eq);
        const uint32_t osb = OUT_BLK ? (uint32_t)(4 * cgp) * plane_b : 128u * (uint32_t)cgp;
        float* xw = Xs + (wave * 4 * 64 + el) * 4;
        const float* xr = Xs + ((wave ^ 1) * 4 * 64 + el) * 4;
        // one round = one cout group j and one PAIR of output columns cp, both register pairs h of the accumulators: what a lane
        // sends / keeps per output pixel is then the float4 of 4 consecutive couts it stores (16-byte stores: the four lanes of a
        // pixel write 64 contiguous bytes; 8-byte stores were measured at +115 us per launch), and nothing has to be kept
        // across rounds.  The price: the row sums a, b, cc, d are computed in both rounds of a j (+12 packed operations).
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int cp = 0; cp < 2; ++cp) {
                __builtin_amdgcn_sched_barrier(0);            // one round at a time
                f32x4 mine[4], send[4];                       // [output row of the half (2)][column of the pair (2)]
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x2 Wm[3][2];
#pragma unroll
                    for (int rr = 0; rr < 3; ++rr) {
                        f32x2 m[6];
#pragma unroll
                        for (int c = 0; c < 6; ++c) m[c] = (f32x2){acc[rr * 6 + c][j][2 * h], acc[rr * 6 + c][j][2 * h + 1]};
                        const f32x2 a = m[1] + m[2], b = m[1] - m[2], cc = m[3] + m[4], d = m[3] - m[4];
                        if (cp == 0) {
                            Wm[rr][0] = (m[0] + a) + cc;
                            Wm[rr][1] = fma2(d, 2.0f, b);
                        } else {
                            Wm[rr][0] = fma2(cc, 4.0f, a);
                            Wm[rr][1] = fma2(d, 8.0f, b) + m[5];
                        }
                    }
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        f32x2 k0, k1, s0, s1;                 // kept rows (2), sent rows (2) of this column
                        if (RG == 0) {
                            const f32x2 s = Wm[1][c] + Wm[2][c], dd = Wm[1][c] - Wm[2][c];
                            k0 = Wm[0][c] + s; k1 = dd; s0 = s; s1 = dd;
                        } else {
                            const f32x2 s = Wm[0][c] + Wm[1][c], dd = Wm[0][c] - Wm[1][c];
                            s0 = s; s1 = dd + dd; k0 = s * 4.0f; k1 = fma2(dd, 8.0f, Wm[2][c]);
                        }
                        mine[c][2 * h] = k0.x; mine[c][2 * h + 1] = k0.y;
                        mine[2 + c][2 * h] = k1.x; mine[2 + c][2 * h + 1] = k1.y;
                        send[c][2 * h] = s0.x; send[c][2 * h + 1] = s0.y;
                        send[2 + c][2 * h] = s1.x; send[2 + c][2 * h + 1] = s1.y;
                    }
                }
#if W44_ABL & 8192     // timing experiment: no exchange
#pragma unroll
                for (int q = 0; q < 4; ++q) mine[q] += send[q];
#else
#pragma unroll
                for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(xw + q * 256) = send[q];
                lds_barrier();
#pragma unroll
                for (int q = 0; q < 4; ++q) mine[q] += *reinterpret_cast<const f32x4*>(xr + q * 256);
                if (!(j == 1 && cp == 1)) lds_barrier();              // the partner has read before the next round overwrites
#endif
#pragma unroll
                for (int rr = 0; rr < 2; ++rr)
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        f32x4 val = mine[rr * 2 + c];
                        if (relu) { val.x = fmaxf(val.x, 0.0f); val.y = fmaxf(val.y, 0.0f); val.z = fmaxf(val.z, 0.0f); val.w = fmaxf(val.w, 0.0f); }
                        if ((!(W44_ABL & 4096) || relu == 77) && oy + rr < H && ox + 2 * cp + c < W) {
                            // blk32: position 8 ((col+1)&3) + tx of the block.  Non-temporal; (s_nop: the wait states between a 16-byte
                            // store and the next write of its data registers, which the compiler cannot see into the asm to insert)
                            const uint32_t so = uniform(OUT_BLK ? osb + (uint32_t)(2 * j) * plane_b + (uint32_t)(rr * tiles_x) * 1024u
                                                                : osb + (uint32_t)(rr * W) * 256u);
                            store_out(val, ovo, orsrc, so, OUT_BLK ? 4 * ((2 * cp + c + 1) & 3) : j + 4 * (2 * cp + c));
                        }
                    }
            }
        }
    };

#ifdef W44_STAMP   // profiling build: cycles per phase, summed over the run, written over the bias array: [workgroup][wave][5]
    uint32_t st_sum[5] = {0, 0, 0, 0, 0};
    uint64_t st_t = __builtin_readcyclecounter();
#define W44_MARK(i) do { const uint64_t now_ = __builtin_readcyclecounter(); st_sum[i] += (uint32_t)(now_ - st_t); st_t = now_; } while (0)
    uint32_t* st_out = reinterpret_cast<uint32_t*>(const_cast<float*>(bias));
    bias = nullptr;
#else
#define W44_MARK(i) do { } while (0)
#endif
    int t_fetch = t_first;

    // One stage = chunk c of the current tile.  PAR = c&1: Raw[PAR^1] holds raw(c+1), Raw[PAR] is receiving raw(c+2).
    // Vector memory operations of a wave, in issue order: behind X2 of the previous stage [second half of U(c) x4, (last stage
    // of a tile: the 16 output stores)]; from inside the first MFMA part [raw(c+2) x3]; behind X1 [first half of U(c+1) x5];
    // (first stage of a tile: raw(2) x3 from inside the second MFMA part instead).
    // vmcnt(3) in front of X1 = the weights are in (the raw tile may still be landing: it is needed behind X1 of the NEXT
    // stage), vmcnt(0) in front of X2.  A register spill inside the loop would be a scratch access = one more vector memory
    // operation (the counts stay safe: extra younger operations only make a wait stricter) with a vmcnt(0) in front of its
    // use - and, between the asm MFMAs, a read of an accumulator without the wait states the compiler gives real MFMAs:
    // the loop must compile without spill STORES inside the MFMA parts (tools/w44_variants.sh prints the spill count).
    auto stage = [&](auto rg_c, auto par_c, auto first_c, int c, int t_cur) __attribute__((always_inline)) {
        constexpr int RG = decltype(rg_c)::value;
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        // entry: Us = U(c) (steps >= SPLIT still landing), Raw[PAR^1] = raw(c+1) visible, v = V(c), raw(c+2) landing in Raw[PAR]
        // the bias is the initial value of the accumulators of position (1,1) (row group 0, step 7): read straight into them - in the
        // first stage of a tile they are dead until that step (a separate copy was spilled)
        if (FIRST && RG == 0) {
            const int el = lane_id();
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[7][j] = *reinterpret_cast<const f32x4*>(bias_s + 32 * cgp + 16 * j + 4 * (el >> 4));
        }
        constexpr int PF = 2;                                 // weight operands are read two positions ahead
        float4 bq[PF + 1];
        const int sl = lane_id();
        const float* ub = Us + ((RG * 2 + cgp) * 64 + sl) * 4;
        // The MFMAs are inline asm with the accumulator tied to its own registers: left to itself hipcc allocates many of them
        // out of place (a copy chain through spare registers) and runs out of registers.  Hazards: an accumulator is touched by
        // every second MFMA (64 cycles apart) and read by vector code only in the epilogue, behind a barrier.
#define W44_MFMA(ACC, A, B) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(ACC) : "v"(A), "v"(B))
#define W44_MFMA0(ACC, A, B) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(ACC) : "v"(A), "v"(B))
#define W44_STEP(S)                                                                                                                   \
        {                                                                                                                             \
            const float4 b = bq[(S) % (PF + 1)];                                                                                      \
            __builtin_amdgcn_sched_barrier(0);                                                                                        \
            if (FIRST && ((S) != 7 || RG != 0)) {                                                                                     \
                W44_MFMA0(acc[S][0], b.x, v[S].x);                                                                                    \
                W44_MFMA0(acc[S][1], b.z, v[S].x);                                                                                    \
            } else {                                                                                                                  \
                W44_MFMA(acc[S][0], b.x, v[S].x);                                                                                     \
                W44_MFMA(acc[S][1], b.z, v[S].x);                                                                                     \
            }                                                                                                                         \
            W44_MFMA(acc[S][0], b.y, v[S].y);                                                                                         \
            W44_MFMA(acc[S][1], b.w, v[S].y);                                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                                        \
        }
        // ---- first part: steps [0, SPLIT)
#pragma unroll
        for (int i = 0; i < PF; ++i) bq[i] = *reinterpret_cast<const float4*>(ub + i * U_STEP);
        __builtin_amdgcn_sched_barrier(0);
        // raw(c+2) -> Raw[PAR] (free since X2 of the previous stage; read behind X1 of the next one), one instruction every other
        // position.  Not in the first stage of a tile: there the previous tile's last transform may still be reading Raw[PAR] in
        // a slower wave - it is issued from the second part, behind X1.
        RawDma rd;
        if (!FIRST) rd = raw_begin((c + 2) & 7, PAR);
#pragma unroll
        for (int s = 0; s < SPLIT; ++s) {
            if (s + PF < SPLIT) bq[(s + PF) % (PF + 1)] = *reinterpret_cast<const float4*>(ub + (s + PF) * U_STEP);
            W44_STEP(s);
            if (!FIRST && (s == 1 || s == 3 || s == 5)) raw_piece(rd, s >> 1);
        }
        W44_MARK(0);                                          // first MFMA part
        // second half of U(c) in LDS; the raw fetch (3 operations, issued behind it) stays in flight.  Behind a tile that lies
        // inside the image the 16 output stores of its epilogue (issued behind the weights too) may stay in flight as well:
        // waiting for their acknowledgements here was measured at ~50 us per launch.
        if (FIRST) { if (stores_in_flight) asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        lds_barrier();                                        // X1: steps >= SPLIT of U(c) visible; everyone is done with steps < SPLIT
        W44_MARK(1);                                          // wait + X1
        dma_half((c + 1) & 7, 0);
        if (FIRST) rd = raw_begin(2, 0);
        // ---- second part: steps [SPLIT, 18)
#pragma unroll
        for (int i = 0; i < PF; ++i) bq[(SPLIT + i) % (PF + 1)] = *reinterpret_cast<const float4*>(ub + (SPLIT + i) * U_STEP);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = SPLIT; s < NSTEP; ++s) {
            if (s + PF < NSTEP) bq[(s + PF) % (PF + 1)] = *reinterpret_cast<const float4*>(ub + (s + PF) * U_STEP);
            W44_STEP(s);
            if (FIRST && (s == SPLIT + 1 || s == SPLIT + 3 || s == SPLIT + 5)) raw_piece(rd, (s - SPLIT) >> 1);
        }
        __builtin_amdgcn_sched_barrier(0);                    // (hipcc otherwise starts the transform above the MFMAs that still read v)
        // (the last stage of a tile transforms BEHIND the epilogue: V would otherwise be live across it, 36 registers too many)
        if (!(W44_ABL & 16) && c != 7) transform(rg_c, PAR ^ 1);    // V(c+1)
        __builtin_amdgcn_sched_barrier(0);
        W44_MARK(2);                                          // second MFMA part + transform
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // first half of U(c+1) and raw(c+2) are in LDS
        lds_barrier();
        W44_MARK(3);                                          // wait + X2                                        // X2: both visible; everyone is done with U(c) and raw(c+1)
        dma_half((c + 1) & 7, 1);                             // (in front of the epilogue's stores: see the wait in front of X1)
        if (c == 7) {
            epilogue(rg_c, t_cur);
            __builtin_amdgcn_sched_barrier(0);
            if (!(W44_ABL & 16)) transform(rg_c, PAR ^ 1);   // V(0) of the next tile: the raw buffer is rewritten behind X1 of its stage 0 at the earliest
            __builtin_amdgcn_sched_barrier(0);
        }
        if (c == 5) {                                         // chunks c+3.. of the fetch stream belong to the next tile
            if (t_fetch + t_step < t_end) t_fetch += t_step;  // (past the end of the run: stay, the fetches are dummies)
            set_fetch_tile(t_fetch);
        }

        W44_MARK(4);                                          // DMA issue, tile switch, once per tile: output transform + last input transform
    };
    using std::integral_constant;
    // the two row groups run two separate copies of prologue + loop: with `rg` a run-time value hipcc keeps both arms of every
    // rg-dependent piece (input transform, output transform) live at once, and whatever is computed in front of the two copies
    // is kept alive THROUGH the first copy for the second one - either way it spills
    auto run = [&](auto rg_c) __attribute__((always_inline)) {
        {   // the wave's cout group / tile group, derived again inside each copy from an opaque copy of the wave index: computed once in
            // front of the two copies they are kept alive through the first copy for the second one (an SGPR spill)
            int w = wave;
            asm volatile("" : "+s"(w));
            cgp = (w >> 1) & 1;
            tg = w >> 2;
        }
        // ---- prologue (once per workgroup): bias, U(0) whole, raw(0), raw(1) staged; V(0) computed; raw(2) on its way
        if (wave == 0) { const int bl = lane_id(); bias_s[bl] = bias ? bias[bl] : 0.0f; }
        set_fetch_tile(t_first);
        dma_half(0, 0);
        dma_half(0, 1);
        dma_raw(0, 0);
        dma_raw(1, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        transform(rg_c, 0);
        __syncthreads();                                          // everyone has read raw(0): the fetch of raw(2) may overwrite it
#pragma unroll 1
        for (int t_cur = t_first; t_cur < t_end; t_cur += t_step) {
            stage(rg_c, integral_constant<int, 0>{}, integral_constant<bool, true>{}, 0, t_cur);
            stage(rg_c, integral_constant<int, 1>{}, integral_constant<bool, false>{}, 1, t_cur);
#pragma unroll 1
            for (int c = 2; c < NCHUNK; c += 2) {
                stage(rg_c, integral_constant<int, 0>{}, integral_constant<bool, false>{}, c, t_cur);
                stage(rg_c, integral_constant<int, 1>{}, integral_constant<bool, false>{}, c + 1, t_cur);
            }
        }
    };
    if (rg == 0) run(integral_constant<int, 0>{}); else run(integral_constant<int, 1>{});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef W44_STAMP
    if (lane_id() == 0)
        for (int i = 0; i < 5; ++i) st_out[((int)blockIdx.x * WAVES + wave) * 5 + i] = st_sum[i];
#endif
}

}  // namespace w44
}  // namespace deqsci

using namespace deqsci;

static void w44_magic(uint32_t d, uint32_t* mg, uint32_t* sh) {
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    *sh = 31 + s;
    *mg = (uint32_t)(((1ull << (31 + s)) + d - 1) / d);
}

static int winograd44_impl(const float* x, const float* u_packed, const float* bias, float* y, int64_t n, int64_t H, int64_t W,
                           int relu, int in_layout, int out_layout, deqsci_stream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    if (!x || !u_packed || !y) return DEQSCI_ERR_NULL;
    if (n <= 0 || H <= 0 || W <= 0) return DEQSCI_ERR_SHAPE;
    if ((in_layout != DEQSCI_ACT_NHWC && in_layout != DEQSCI_ACT_BLK32) || (out_layout != DEQSCI_ACT_NHWC && out_layout != DEQSCI_ACT_BLK32))
        return DEQSCI_ERR_UNSUPPORTED;
    if (H > (1 << 20) || W > (1 << 20) || x == y) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(x) || !aligned16(u_packed) || !aligned16(y)) return DEQSCI_ERR_ALIGN;
    const int64_t tiles_x = ceil_div(W, w44::OUT_COLS), tiles_y = ceil_div(H, w44::OUT_ROWS);
    const int64_t n_tiles = n * tiles_x * tiles_y;
    // 32-bit arithmetic in the kernel: tile indices, and byte offsets inside one image (padded to 32 columns in the blk32 layout).
    // The zero padding relies on RAW_OOB lying beyond the buffer descriptor's range for every instruction: num_records = image bytes
    // + RAW_BIAS, and the range check adds the immediate offset (<= 2048) and the 16 bytes of the access.  Larger images (from
    // about 2896 x 2896 on) are refused - the front end then runs the F(2x2,3x3) kernel.
    if (n_tiles > (int64_t)INT32_MAX / 16 || H * tiles_x * 32 * 256 + w44::RAW_BIAS + 2048 + 16 > (int64_t)w44::RAW_OOB) return DEQSCI_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t resident = (int64_t)num_cus();
    const dim3 grid((unsigned)(n_tiles < resident ? n_tiles : resident));
    uint32_t mg_img, sh_img, mg_tx, sh_tx;
    w44_magic((uint32_t)(tiles_x * tiles_y), &mg_img, &sh_img);
    w44_magic((uint32_t)tiles_x, &mg_tx, &sh_tx);
#define W44_LAUNCH(KERNEL)                                                                                                              \
    do {                                                                                                                                \
        if (ev0 || ev1)                                                                                                                 \
            hipExtLaunchKernelGGL(KERNEL, grid, dim3(w44::TBW), 0, st, ev0, ev1, 0, x, u_packed, bias, y, (int)H, (int)W, relu,        \
                                  (int)tiles_x, (int)tiles_y, (int)n_tiles, mg_img, sh_img, mg_tx, sh_tx);                              \
        else                                                                                                                            \
            hipLaunchKernelGGL(KERNEL, grid, dim3(w44::TBW), 0, st, x, u_packed, bias, y, (int)H, (int)W, relu, (int)tiles_x,          \
                               (int)tiles_y, (int)n_tiles, mg_img, sh_img, mg_tx, sh_tx);                                               \
    } while (0)
    if (in_layout == DEQSCI_ACT_NHWC && out_layout == DEQSCI_ACT_NHWC) W44_LAUNCH((w44::winograd44_conv64_kernel<0, 0>));
    else if (in_layout == DEQSCI_ACT_NHWC) W44_LAUNCH((w44::winograd44_conv64_kernel<0, 1>));
    else if (out_layout == DEQSCI_ACT_NHWC) W44_LAUNCH((w44::winograd44_conv64_kernel<1, 0>));
    else W44_LAUNCH((w44::winograd44_conv64_kernel<1, 1>));
#undef W44_LAUNCH
    return launch_status();
}

extern "C" int deqsci_conv3x3_c64_winograd44_f32(const float* x, const float* u_packed, const float* bias, float* y, int64_t n,
                                                 int64_t H, int64_t W, int relu, deqsci_stream_t stream) {
    return winograd44_impl(x, u_packed, bias, y, n, H, W, relu, DEQSCI_ACT_NHWC, DEQSCI_ACT_NHWC, stream, nullptr, nullptr);
}

extern "C" int deqsci_conv3x3_c64_winograd44_layout_f32(const float* x, const float* u_packed, const float* bias, float* y, int64_t n,
                                                        int64_t H, int64_t W, int relu, int in_layout, int out_layout,
                                                        deqsci_stream_t stream, void* start_event, void* stop_event) {
    if ((start_event == nullptr) != (stop_event == nullptr)) return DEQSCI_ERR_NULL;
    return winograd44_impl(x, u_packed, bias, y, n, H, W, relu, in_layout, out_layout, stream, static_cast<hipEvent_t>(start_event),
                           static_cast<hipEvent_t>(stop_event));
}

extern "C" int deqsci_conv3x3_c64_winograd44_timed_f32(const float* x, const float* u_packed, const float* bias, float* y, int64_t n,
                                                       int64_t H, int64_t W, int relu, deqsci_stream_t stream, void* start_event,
                                                       void* stop_event) {
    if (!start_event || !stop_event) return DEQSCI_ERR_NULL;
    return winograd44_impl(x, u_packed, bias, y, n, H, W, relu, DEQSCI_ACT_NHWC, DEQSCI_ACT_NHWC, stream,
                           static_cast<hipEvent_t>(start_event), static_cast<hipEvent_t>(stop_event));
}
