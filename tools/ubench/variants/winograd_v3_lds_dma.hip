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
// Work unit ("block tile") = 8 x 8 Winograd tiles (16 x 16 output pixels) x all 64 output channels.  The 16 transform
// positions xi are 16 independent GEMMs  M[xi] (64 cout x 64 tiles) += U[xi] (64 x cin) V[xi] (cin x 64):
//   * ONE PERSISTENT workgroup of 8 wavefronts per CU (two per SIMD, <= 256 registers, 88 KB LDS), walking a contiguous
//     run of block tiles of "its" XCD (neighbouring tiles share halo pixels in that XCD's L2).  64 tiles per CU is the
//     most the register file holds accumulators for (256 KB), and it is what one pass over the 256 KB of transformed
//     weights is amortised over.  Wave w owns tiles [16*(w>>1), +16) x couts [32*(w&1), +32) of every M[xi]: 16 x 2
//     accumulators of v_mfma_f32_16x16x4_f32 (128 registers) that hold every value the output transform of its
//     (tile, cout) needs.
//   * the 64 input channels are consumed in chunks of 8, and the chunk pipeline runs ACROSS block tiles: one stage =
//     MFMA phase + input transform + ONE barrier.  BOTH operands reach LDS by the DMA path - no staging registers, no
//     ds_write, no per-stage address arithmetic:
//       - weights U(c+1): host-packed in LDS order, so the 32 KB chunk is a LINEAR copy (global_load_lds_dwordx4, scalar
//         base + immediate offsets): every wave moves 4 KB in 4 instructions; a lane's weight operands for one xi are ONE
//         conflict-free ds_read_b128 feeding 4 MFMAs;
//       - raw input of chunk c+2 (18 x 18 pixels x 8 channels, possibly of the NEXT block tile): 12 x
//         buffer_load_dwordx4 ... lds per workgroup.  The LDS side of that instruction is lane-linear (M0 + 16 * lane), the
//         global side is a per-lane offset, so the tile's LAYOUT is chosen by which pixel each lane fetches: 16-byte units
//         (pixel, channel half) at  half * 384 + row * 20 + col/2 + 9 * (col & 1)  - even and odd columns apart, rows 20
//         units apart - which makes every per-lane patch read (16 tiles x 2 channel pairs per half wave) hit all 64 banks
//         exactly once.  Pixels outside the image are given an offset beyond the buffer descriptor's range: the hardware
//         writes ZEROS for them (checked on gfx950: tools/ubench/buffer_lds_oob.hip), so the zero padding of the
//         convolution costs no instruction either;
//       - each MFMA lane (tile i = lane&15, channel pair q = lane>>4) reads ITS OWN 4x4 patch of chunk c+1 into the
//         registers of the V operands the MFMAs of chunk c have just consumed, and turns it into V = B^T d B in place.
//     So a block tile has no prologue of its own: its first two raw chunks and first weight chunk are in flight while the
//     previous tile finishes, and only the output transform sits between two tiles' MFMAs.
//   * the matrix core gets the WEIGHTS as its A operand: D rows (4 per lane, consecutive registers) are 4 consecutive
//     couts of one tile, so the epilogue (Y = A^T M A, ReLU) is per-lane register work ending in 16-byte stores; the bias
//     is the initial value of the xi = 5 accumulator (its coefficient in all four outputs is 1).
//
// What bounds it (tools/ubench/mfma_f32_fillers.hip, mfma_valu_mix.hip, cu_fill_rate.hip, winograd_stamps.py; DESIGN.md
// has the numbers):
//   * the fp32 MFMA shares the SIMD with the vector ALU: next to a stream of v_mfma_f32_16x16x4_f32 every VALU
//     instruction - of EITHER wave of the SIMD, packed or not - costs 5-6 cycles of matrix-pipe time, a ds_write 10-18,
//     while ds_read / s_waitcnt / SALU are free up to about two per MFMA.  The input transform (32 packed adds per wave
//     and chunk, computed twice per tile because the two cout halves live in different waves) and the output transform
//     are therefore what is left of the gap to the peak; everything else that used to be vector work (staging through
//     registers, border selects, address arithmetic) is gone from the loop;
//   * one wave issuing 16x16x4 MFMAs back to back with nothing in between gets one per 40 cycles, not 32 (a second
//     wave, or any cheap instruction between two MFMAs, restores 32) - another reason for two waves per SIMD;
//   * two things the compiler must not be allowed to do, both measured:
//       - __syncthreads() is a fence + s_barrier and the fence becomes `s_waitcnt vmcnt(0)` wherever a DMA is in flight ->
//         wg_lds_barrier() (lgkmcnt only) with the vmcnt wait placed by hand;
//       - after __builtin_amdgcn_global_load_lds / raw_buffer_load_lds hipcc cannot tell which LDS bytes the DMA writes
//         and puts vmcnt(0) in front of the next ds_read of ANY LDS array -> both DMA forms are inline asm.
#include "common.hpp"
#include <hip/hip_ext.h>
#include <type_traits>
#pragma clang diagnostic ignored "-Winline-asm"   // m0 is named as a clobber of the LDS-DMA asm below, on purpose

namespace deqsci {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int WG_CK = 8;                      // input channels per chunk
constexpr int WG_NCHUNK = 64 / WG_CK;
#ifndef WG_PF
#define WG_PF 2
#endif
#ifndef WG_FILL
#define WG_FILL 0                             // experiment: s_nop 0 between MFMAs that would otherwise be back to back
#endif
#ifndef WG_PRIO
#define WG_PRIO 0                             // experiment: 1 = s_setprio 1 for the younger wave of a SIMD, 2 = for the older one
#endif
#ifndef WG_ABL
#define WG_ABL 0                              // timing ablations only (tools/ubench), results are wrong: 1 = no patch reads, 2 = no DMA,
                                              // 4 = no input transform, 16 = no output transform / stores
#endif
constexpr int WG_WAVES = 8;                   // wavefronts per workgroup (one workgroup per CU)
constexpr int WG_TB = 64 * WG_WAVES;
constexpr int WG_TROWS = 8;                   // Winograd tile rows of a block tile (8 tiles per row, 16 tiles per wave pair)
// raw chunk tile in LDS, in 16-byte units (4 channels of one pixel); see the header for the layout
constexpr int WG_RAW_COLS = 18, WG_RAW_ROWS = 2 * WG_TROWS + 2;
constexpr int WG_RAW_ROW_U = 20;                                  // units per staged pixel row (18 + 2 idle): 2 rows = 8 mod 16
constexpr int WG_RAW_HALF_U = 384;                                // units per channel half (18 * 20 = 360 used)
constexpr int WG_RAW_BUF_U = 2 * WG_RAW_HALF_U;                   // 768 units = 12 KB = 12 DMA instructions of one wave
constexpr int WG_RAW_BUF = WG_RAW_BUF_U * 4;                      // floats
constexpr int WG_RAW_DMA = WG_RAW_BUF_U / 64;                     // 12
constexpr int WG_U_CHUNK = 16 * 2 * 64 * 4;   // floats of one weight chunk in LDS (32 KB)
constexpr uint32_t WG_OOB = 0x80000000u;      // buffer offset of a pixel outside the image: beyond num_records -> zeros

// Workgroup barrier that orders LDS traffic only (see the header).
__device__ __forceinline__ void wg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// a - b on a register pair in ONE instruction (hipcc splits the vector subtraction of the epilogue into two v_sub_f32)
__device__ __forceinline__ f32x2 wg_pk_sub(f32x2 a, f32x2 b) {
    f32x2 r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ uint32_t wg_uniform(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

// Exact t / d for 0 <= t < 2^31 as a multiply and a shift (host: wg_magic): the tile decode is scalar code without the
// ~40-instruction software division.
__device__ __forceinline__ int wg_div(int t, uint32_t mg, uint32_t sh) { return (int)(((uint64_t)(uint32_t)t * mg) >> sh); }

// tiles_x / tiles_y: block tiles per image row / column; n_tiles = images * tiles_x * tiles_y;
// (mg_img, sh_img) / (mg_tx, sh_tx): division magic for tiles_x * tiles_y and tiles_x
__global__ __launch_bounds__(WG_TB, 2) void winograd_conv64_kernel(const float* __restrict__ x, const float* __restrict__ Ug,
                                                                   const float* __restrict__ bias, float* __restrict__ y,
                                                                   int H, int W, int relu, int tiles_x, int tiles_y, int n_tiles,
                                                                   uint32_t mg_img, uint32_t sh_img, uint32_t mg_tx, uint32_t sh_tx) {
    __shared__ __attribute__((aligned(16))) float Us[2 * WG_U_CHUNK];             // 2 x U[xi][cout half][MFMA lane][j][2]   64 KB
    __shared__ __attribute__((aligned(16))) float Raw[2 * WG_RAW_BUF];            // 2 x raw chunk tile                      24 KB
    __shared__ __attribute__((aligned(16))) float bias_s[64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (int)wg_uniform((uint32_t)(tid >> 6));          // in an SGPR: the DMA bookkeeping is scalar code
    const int wt = wave >> 1, wn = wave & 1;
#ifndef WG_ROLES
#define WG_ROLES 1                            // 0 (experiment): every wave keeps transform and epilogue in front of the barrier
#endif
    const int role = WG_ROLES ? wave >> 2 : 0;                       // waves w and w + 4 share a SIMD; w is the older one

    // ---- the run of block tiles of this workgroup: XCD k (workgroups k, k+8, ...) owns tiles [k*per_xcd, (k+1)*per_xcd)
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
    if (WG_PRIO == 1 && (wave >> 2)) __builtin_amdgcn_s_setprio(1);
    if (WG_PRIO == 2 && !(wave >> 2)) __builtin_amdgcn_s_setprio(1);

    // ---- raw staging role: the chunk tile is WG_RAW_DMA = 12 DMA instructions of 64 units; wave w issues instruction w and,
    // for w < 4, instruction w + 8.  Lane l of instruction k fills unit 64 k + l; which pixel / channel half that is:
    int dpix[2];                                                   // (staged row << 8) | staged column, -1 = idle unit
    uint32_t dhalf[2];                                             // byte offset of the unit's channel half inside a chunk (0 / 16)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int u = 64 * (wave + 8 * j) + lane;
        const int half = u >= WG_RAW_HALF_U, r = u - half * WG_RAW_HALF_U;
        const int row = r / WG_RAW_ROW_U, cp = r - row * WG_RAW_ROW_U;
        const int col = cp < 9 ? 2 * cp : 2 * (cp - 9) + 1;
        dpix[j] = (row < WG_RAW_ROWS && cp < WG_RAW_COLS) ? ((row << 8) | col) : -1;
        dhalf[j] = 16u * half;
    }
    const int n_dma = wave < WG_RAW_DMA - 8 ? 2 : 1;               // (uniform)
    // fetch stream state: per-lane buffer offsets (bytes from the image base; WG_OOB = outside the image) of the block tile
    // the chunk two stages ahead belongs to, and that image's buffer descriptor
    uint32_t voff[2];
    i32x4 rsrc;
    auto set_fetch_tile = [&](int t) {
        const int n = wg_div(t, mg_img, sh_img), r = t - n * (tiles_x * tiles_y);
        const int by = wg_div(r, mg_tx, sh_tx), bx = r - by * tiles_x;
        const uint64_t base = (uint64_t)(x + (int64_t)n * H * W * 64);
        rsrc.x = (int)wg_uniform((uint32_t)base);
        rsrc.y = (int)wg_uniform((uint32_t)(base >> 32));          // stride 0: raw buffer, offsets in bytes
        rsrc.z = (int)wg_uniform((uint32_t)(H * W) * 256u);        // num_records = bytes of one image (< 2^31: launcher)
        rsrc.w = 0x00020000;
        const int py0 = 2 * WG_TROWS * by - 1, px0 = 16 * bx - 1;  // image coordinates of staged pixel (0,0)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int iy = py0 + (dpix[j] >> 8), ix = px0 + (dpix[j] & 255);
            const bool ok = dpix[j] >= 0 && iy >= 0 && iy < H && ix >= 0 && ix < W;
            voff[j] = ok ? (uint32_t)(iy * W + ix) * 256u + dhalf[j] : WG_OOB;
        }
    };
    const uint32_t raw_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)Raw;
    // chunk c of the fetch tile -> Raw[buf]: the scalar offset selects the 8 channels (32 bytes) of the chunk
    auto dma_raw = [&](int c, int buf) {
        const uint32_t soff = wg_uniform((uint32_t)c * (WG_CK * 4));
#pragma unroll
        for (int j = 0; j < 2; ++j)
            if (j < n_dma) {
                const uint32_t m0v = wg_uniform(raw_lds + (uint32_t)(buf * WG_RAW_BUF * 4 + (wave + 8 * j) * 1024));
                asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
                             ::"s"(m0v), "v"(voff[j]), "s"(rsrc), "s"(soff) : "memory", "m0");
            }
    };
    // ---- weight chunk: DMA global -> LDS.  The chunk is host-packed in LDS order, so it is a linear 32 KB copy: wave w
    // moves bytes [4 KiB * w, +4 KiB) in 4 instructions that differ only in their immediate offset (which the hardware adds
    // to the global AND the LDS address): one scalar base and one M0 value per chunk, no address arithmetic per piece.
    const uint32_t us_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)Us;
    constexpr int DMA_WAVE_BYTES = WG_U_CHUNK * 4 / WG_WAVES;                          // 4 KiB
    const uint32_t dma_voff = (uint32_t)(wave * DMA_WAVE_BYTES + lane * 16);            // bytes
    // returns the scalar global base of the chunk (+ half a wave share) and the M0 value = LDS base of this wave's share (+ half)
    struct UDma { uint64_t g; uint32_t m0v; };
    auto dma_u_setup = [&](int c, int buf) -> UDma {
        const uint64_t g = (uint64_t)(Ug + (int64_t)c * WG_U_CHUNK) + DMA_WAVE_BYTES / 2;
        UDma d;
        d.m0v = wg_uniform(us_lds + (uint32_t)(buf * WG_U_CHUNK * 4 + wave * DMA_WAVE_BYTES + DMA_WAVE_BYTES / 2));
        d.g = ((uint64_t)wg_uniform((uint32_t)(g >> 32)) << 32) | wg_uniform((uint32_t)g);
        return d;
    };
    // (M0 is set once per chunk by dma_u_m0: nothing between the pieces touches it)
    auto dma_u_m0 = [&](const UDma& d) { asm volatile("s_mov_b32 m0, %0" ::"s"(d.m0v) : "m0"); };
#define WG_DMA_PIECE(OFF) asm volatile("global_load_lds_dwordx4 %0, %1 offset:" #OFF ::"v"(dma_voff), "s"(d.g) : "memory")
    constexpr int DMA_PIECES = DMA_WAVE_BYTES / 1024;                                                // 4
    auto dma_u_piece = [&](const UDma& d, int j) {                                                   // KiB j of this wave's share
        switch (j - DMA_PIECES / 2) {
            case -2: WG_DMA_PIECE(-2048); break;
            case -1: WG_DMA_PIECE(-1024); break;
            case 0: WG_DMA_PIECE(0); break;
            default: WG_DMA_PIECE(1024); break;
        }
    };

    f32x4 acc[16][2];                                        // written, not accumulated, by the first chunk of every tile

    // MFMA roles: lane (i = lane&15, q = lane>>4) owns tile 16*wt + i and channels {2q, 2q+1} of the chunk
    const int mi = lane & 15, mq = lane >> 4;
    const int tl_a = 16 * wt + mi;
    // float index of this lane's patch element (0,0) in a raw tile; element (pr,pc) is WG_PATCH(pr,pc) floats further
    const int pbase = ((mq >> 1) * WG_RAW_HALF_U + 2 * (tl_a >> 3) * WG_RAW_ROW_U + (tl_a & 7)) * 4 + 2 * (mq & 1);
#define WG_PATCH(pr, pc) (((pr) * WG_RAW_ROW_U + ((pc) >> 1) + 9 * ((pc) & 1)) * 4)
    const float* ub = Us + (wn * 64 + lane) * 4;             // this lane's weight operands of xi = 0 in buffer 0

    // B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]:  V = B^T d B
    // ONE register array holds the raw patch and then, transformed in place, V: during the MFMA phase element xi is
    // re-loaded with the next chunk's patch as soon as the MFMAs of xi have consumed it (register pairs: every transform
    // step is one v_pk_add_f32).
    f32x2 v[16];
    // Patch reads are single ds_read_b64 with immediate offsets, written as asm: left to itself hipcc pairs them into
    // ds_read2_b64, which serves 16 lanes per LDS cycle on 32 banks - the 16 tiles of a lane group then collide two by two
    // (they use half of each 16-byte unit), while ds_read_b64 serves the 32 lanes (16 tiles x 2 channel pairs) of a group
    // from all 64 banks at once.  The results are claimed by patch_landed() before the transform touches them.
    const uint32_t patch_lds = raw_lds + (uint32_t)pbase * 4u;         // LDS byte address of this lane's patch element (0,0), buffer 0
#define WG_PATCH_READ(dst, buf, pr, pc) \
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(patch_lds), "n"((buf) * WG_RAW_BUF * 4 + WG_PATCH(pr, pc) * 4))
    auto patch_landed = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                       "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15]));
    };
    auto read_patch = [&](auto buf_c) {
        constexpr int BUF = decltype(buf_c)::value;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr)
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) WG_PATCH_READ(v[pr * 4 + pc], BUF, pr, pc);
        patch_landed();
    };
    auto transform = [&]() {
        f32x2 w[16];
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {                      // rows: w = B^T d (column pc of the patch)
            const f32x2 d0 = v[pc], d1 = v[4 + pc], d2 = v[8 + pc], d3 = v[12 + pc];
            w[pc] = d0 - d2;
            w[4 + pc] = d1 + d2;
            w[8 + pc] = d2 - d1;
            w[12 + pc] = d1 - d3;
        }
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {                      // columns: V = w B
            const f32x2 w0 = w[pr * 4], w1 = w[pr * 4 + 1], w2 = w[pr * 4 + 2], w3 = w[pr * 4 + 3];
            v[pr * 4] = w0 - w2;
            v[pr * 4 + 1] = w1 + w2;
            v[pr * 4 + 2] = w2 - w1;
            v[pr * 4 + 3] = w1 - w3;
        }
    };

    // ---- output transform Y = A^T M A, A^T = [1 1 1 0; 0 1 -1 -1]; ReLU; 16-byte stores.  The bias is already inside:
    // the coefficient of M[1][1] (xi = 5) is 1 in all four outputs, so the tile's first MFMA of xi = 5 starts from the
    // bias instead of zero.  All arithmetic on register PAIRS (v_pk_add_f32): non-MFMA vector instructions are what
    // bounds this kernel once the matrix pipe is fed (tools/ubench/mfma_valu_mix.hip).
    // D layout of the 16x16 MFMA with the weights as A operand: col = lane&15 (tile), row = 4*(lane>>4) + reg (cout)
    auto epilogue = [&](int t) __attribute__((always_inline)) {
        const int n = wg_div(t, mg_img, sh_img), r = t - n * (tiles_x * tiles_y);
        const int by = wg_div(r, mg_tx, sh_tx), bx = r - by * tiles_x;
        float* yn = y + (int64_t)n * H * W * 64;
        const int tl = 16 * wt + (lane & 15);
        const int oy = 2 * (WG_TROWS * by + (tl >> 3)), ox = 2 * (8 * bx + (tl & 7));
        float* o = yn + ((int64_t)oy * W + ox) * 64 + 32 * wn + 4 * (lane >> 4);
        const bool in0 = oy < H && ox < W, inx = ox + 1 < W, iny = oy + 1 < H;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x2 o00[2], o01[2], o10[2], o11[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x2 s0[4], s1[4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const f32x2 m0 = {acc[b][j][2 * h], acc[b][j][2 * h + 1]};
                    const f32x2 m1 = {acc[4 + b][j][2 * h], acc[4 + b][j][2 * h + 1]};
                    const f32x2 m2 = {acc[8 + b][j][2 * h], acc[8 + b][j][2 * h + 1]};
                    const f32x2 m3 = {acc[12 + b][j][2 * h], acc[12 + b][j][2 * h + 1]};
                    s0[b] = (m0 + m1) + m2;
                    s1[b] = wg_pk_sub(wg_pk_sub(m1, m2), m3);
                }
                o00[h] = (s0[0] + s0[1]) + s0[2];
                o01[h] = wg_pk_sub(wg_pk_sub(s0[1], s0[2]), s0[3]);
                o10[h] = (s1[0] + s1[1]) + s1[2];
                o11[h] = wg_pk_sub(wg_pk_sub(s1[1], s1[2]), s1[3]);
                if (relu) {
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        o00[h][e] = fmaxf(o00[h][e], 0.0f); o01[h][e] = fmaxf(o01[h][e], 0.0f);
                        o10[h][e] = fmaxf(o10[h][e], 0.0f); o11[h][e] = fmaxf(o11[h][e], 0.0f);
                    }
                }
            }
            if (in0) {
                float* oj = o + 16 * j;
                st4(oj, make_float4(o00[0][0], o00[0][1], o00[1][0], o00[1][1]));
                if (inx) st4(oj + 64, make_float4(o01[0][0], o01[0][1], o01[1][0], o01[1][1]));
                if (iny) {
                    st4(oj + (int64_t)W * 64, make_float4(o10[0][0], o10[0][1], o10[1][0], o10[1][1]));
                    if (inx) st4(oj + (int64_t)W * 64 + 64, make_float4(o11[0][0], o11[0][1], o11[1][0], o11[1][1]));
                }
            }
        }
    };

#ifdef WG_STAMP
    // timing instrumentation (tools/ubench/winograd_stamps.py): waves 0 and 4 - the two waves of one SIMD - record s_memtime at
    // WG_NSTAMP marks after skipping WG_STAMP_SKIP; the `bias` pointer is re-purposed as the output buffer (2 x 48 uint64 / block)
#ifndef WG_STAMP_SKIP
#define WG_STAMP_SKIP 0
#endif
    constexpr int WG_NSTAMP = 46;
    unsigned long long* stamp_out = reinterpret_cast<unsigned long long*>(const_cast<float*>(bias)) + (size_t)blockIdx.x * 96;
    __shared__ unsigned long long stamp_lds[2][48];           // stamps go to LDS: a global store would count in vmcnt
    int stamp_i = 0;
    const bool stamper = (tid & 255) == 0;
#define WG_MARK() do { if (stamp_i >= WG_STAMP_SKIP && stamp_i < WG_STAMP_SKIP + WG_NSTAMP) { if (stamper) stamp_lds[tid >> 8][stamp_i - WG_STAMP_SKIP] = __builtin_readcyclecounter(); } ++stamp_i; } while (0)
    bias = nullptr;
#else
#define WG_MARK() do { } while (0)
#endif
    WG_MARK();                                                // 0: kernel entry
    // ---- prologue (once per workgroup): bias, U(0), raw(0), raw(1) staged; V(0) computed
    if (tid < 64) bias_s[tid] = bias ? bias[tid] : 0.0f;
    set_fetch_tile(t_first);
    dma_raw(0, 0);
    dma_raw(1, 1);
    {
        const UDma d = dma_u_setup(0, 0);
        dma_u_m0(d);
#pragma unroll
        for (int j = 0; j < DMA_PIECES; ++j) dma_u_piece(d, j);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    read_patch(std::integral_constant<int, 0>{});
    transform();
    __syncthreads();                                          // every wave has read its patch of raw(0): stage 0 overwrites it
    WG_MARK();                                                // 1: prologue done
    int t_fetch = t_first;                                    // tile of the fetch stream
    auto finish_tile = [&](int t) __attribute__((always_inline)) {
        if (!(WG_ABL & 16)) epilogue(t);
        else {
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) asm volatile("" ::"v"(acc[xi][0]), "v"(acc[xi][1]));
        }
    };
    // One pipeline stage = chunk c of the current tile.  PAR = c&1 selects the LDS buffers (compile-time: every LDS address
    // of the stage is an immediate offset), FIRST = chunk 0: the accumulators are written from zero / the bias.
    // (always_inline: called as a function, the by-reference captures - accumulators included - live in scratch memory.)
    auto stage = [&](auto par_c, auto first_c, int c, int t_cur) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        // entry: Us[PAR] = U(c), Raw[PAR^1] = raw(c+1) visible; v = V(c) in registers; everyone is done with raw(c) in Raw[PAR]
        // and with U(c-1) in Us[PAR^1]
        WG_MARK();
        if (c == 6) {                                         // chunks c+2.. of the fetch stream belong to the next tile
            if (t_fetch + t_step < t_end) t_fetch += t_step;  // (past the end of the run: stay, the fetches are dummies
            set_fetch_tile(t_fetch);                          //  of valid memory nobody reads)
        }
        // raw(c+2) -> Raw[PAR], at the very start of the stage: the second wave of a SIMD waits for its turn on the matrix
        // pipe after this, and the copy has the whole stage (about 2 us) to land
        if (!(WG_ABL & 2)) dma_raw((c + 2) & 7, PAR);
        WG_MARK();
        // ---- MFMA phase.  Hand-ordered: everything that is not an MFMA sits right behind the FIRST of the four MFMAs of
        // a transform position.
        const float* ubc = ub + PAR * WG_U_CHUNK;
        f32x4 init5[2];
        if (FIRST) {
#pragma unroll
            for (int j = 0; j < 2; ++j) init5[j] = *reinterpret_cast<const f32x4*>(bias_s + 32 * wn + 16 * j + 4 * (lane >> 4));
        }
        const UDma dg = dma_u_setup((c + 1) & 7, PAR ^ 1);    // weight chunk to stage, into the buffer M(c-1) released
        dma_u_m0(dg);
        constexpr int PF = WG_PF;                             // weight operands are read PF transform positions ahead
        float4 bq[PF + 1];
#pragma unroll
        for (int i = 0; i < PF; ++i) bq[i] = *reinterpret_cast<const float4*>(ubc + i * (2 * 64 * 4));
        __builtin_amdgcn_sched_barrier(0);
        WG_MARK();
        // One non-MFMA instruction per MFMA gap, never more: a lone wave issues v_mfma_f32_16x16x4_f32 every 32 cycles with one
        // cheap instruction in each gap, but a gap holding three or four of them (the operand read, its wait, the patch
        // reads) stretches to 50-60 cycles (tools/ubench/mfma_f32_fillers.hip) - and a wave IS alone on the matrix pipe for most
        // of its phase.  Per transform position:  M1 | weights of xi+PF | M2 | DMA piece | M3 | patch element xi-1 | M4 | (wait)
        const float* ppn = Raw + (PAR ^ 1) * WG_RAW_BUF + pbase;
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            if (xi == 4 || xi == 8 || xi == 12) WG_MARK();
            const float4 b = bq[xi % (PF + 1)];
            const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
            __builtin_amdgcn_sched_barrier(0);
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, v[xi].x, FIRST ? (xi == 5 ? init5[0] : zero) : acc[xi][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (xi + PF < 16) bq[(xi + PF) % (PF + 1)] = *reinterpret_cast<const float4*>(ubc + (xi + PF) * (2 * 64 * 4));
            __builtin_amdgcn_sched_barrier(0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, v[xi].x, FIRST ? (xi == 5 ? init5[1] : zero) : acc[xi][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (xi < DMA_PIECES && !(WG_ABL & 2)) dma_u_piece(dg, xi);   // U(c+1): needed one stage from now
            __builtin_amdgcn_sched_barrier(0);
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, v[xi].y, acc[xi][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // V[xi-1] is consumed: its registers take the next chunk's patch element (a plain load - hipcc counts it in its
            // lgkmcnt waits; the fences on both sides keep it from pairing two of them into a ds_read2_b64, which would
            // collide two by two on its 32 banks)
            if (xi >= 1 && !(WG_ABL & 1)) v[xi - 1] = *reinterpret_cast<const f32x2*>(ppn + WG_PATCH((xi - 1) >> 2, (xi - 1) & 3));
            __builtin_amdgcn_sched_barrier(0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.w, v[xi].y, acc[xi][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (!(WG_ABL & 1)) v[15] = *reinterpret_cast<const f32x2*>(ppn + WG_PATCH(3, 3));
        WG_MARK();
        // The two waves of a SIMD cannot overlap their MFMA phases (one matrix pipe; the older wave, role 0, goes first), so
        // everything else of the stage is placed beside the OTHER wave's MFMA phase:
        //   role 0:  MFMA | transform, epilogue(c = 7) | barrier                  (beside role 1's MFMAs of this stage)
        //   role 1:         MFMA | barrier | transform, epilogue(c = 7)          (beside role 0's MFMAs of the next stage)
        // The barrier carries the V registers as operands: without that hipcc sinks the (register-only) transform of both
        // roles behind it, where nobody is issuing MFMAs.
        if (role == 0) {
            if (!(WG_ABL & 4)) transform();                   // V(c+1) from the patch read during the MFMAs
            if (c == 7) finish_tile(t_cur);
        }
        WG_MARK();
        // this wave's parts of U(c+1) and raw(c+2) are in LDS (role 0's stores of the epilogue complete under role 1's MFMAs);
        // after the barrier U(c+1), raw(c+2) are visible and every wave is done with Us[PAR], Raw[PAR^1]
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier"
                     : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]), "+v"(v[8]),
                       "+v"(v[9]), "+v"(v[10]), "+v"(v[11]), "+v"(v[12]), "+v"(v[13]), "+v"(v[14]), "+v"(v[15])
                     :: "memory");
        WG_MARK();
        if (role != 0) {
            if (!(WG_ABL & 4)) transform();
            if (c == 7) finish_tile(t_cur);
        }
    };
    using std::integral_constant;
#pragma unroll 1
    for (int t_cur = t_first; t_cur < t_end; t_cur += t_step) {
        stage(integral_constant<int, 0>{}, integral_constant<bool, true>{}, 0, t_cur);
        stage(integral_constant<int, 1>{}, integral_constant<bool, false>{}, 1, t_cur);
#pragma unroll 1
        for (int c = 2; c < WG_NCHUNK; c += 2) {
            stage(integral_constant<int, 0>{}, integral_constant<bool, false>{}, c, t_cur);
            stage(integral_constant<int, 1>{}, integral_constant<bool, false>{}, c + 1, t_cur);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // nothing may still be landing in LDS when the workgroup retires
    WG_MARK();
#ifdef WG_STAMP
    if (stamper) {
        unsigned long long* o = stamp_out + 48 * (tid >> 8);
        for (int i = 0; i < WG_NSTAMP && i < stamp_i - WG_STAMP_SKIP; ++i) o[i] = stamp_lds[tid >> 8][i];
        o[46] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
        o[47] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID: wave/simd/cu/sh/se
    }
#endif
}

}  // namespace deqsci

#ifndef WG_NO_CABI
using namespace deqsci;

// t / d == (t * mg) >> sh for every 0 <= t < 2^31:  sh = 31 + ceil(log2 d), mg = ceil(2^sh / d) < 2^32
static void wg_magic(uint32_t d, uint32_t* mg, uint32_t* sh) {
    uint32_t s = 0;
    while ((1ull << s) < d) ++s;
    *sh = 31 + s;
    *mg = (uint32_t)(((1ull << (31 + s)) + d - 1) / d);
}

static int winograd_impl(const float* x, const float* u_packed, const float* bias, float* y, int64_t n, int64_t H, int64_t W,
                         int relu, deqsci_stream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    if (!x || !u_packed || !y) return DEQSCI_ERR_NULL;
    if (n <= 0 || H <= 0 || W <= 0) return DEQSCI_ERR_SHAPE;
    if (H > (1 << 20) || W > (1 << 20) || x == y) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(x) || !aligned16(u_packed) || !aligned16(y)) return DEQSCI_ERR_ALIGN;
    const int64_t tiles_x = ceil_div(ceil_div(W, 2), 8), tiles_y = ceil_div(ceil_div(H, 2), WG_TROWS);
    const int64_t n_tiles = n * tiles_x * tiles_y;
    // 32-bit arithmetic in the kernel: tile indices, and the per-image BYTE offset of a pixel, which must stay below the
    // out-of-range marker 2^31 of the buffer loads (H*W*64 channels*4 B < 2^31)
    if (n_tiles > (int64_t)INT32_MAX / 16 || H * W >= (int64_t)1 << 23) return DEQSCI_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t resident = (int64_t)num_cus();                 // persistent workgroups: 8 wavefronts (2 per SIMD) on every CU
    const dim3 grid((unsigned)(n_tiles < resident ? n_tiles : resident));
    uint32_t mg_img, sh_img, mg_tx, sh_tx;
    wg_magic((uint32_t)(tiles_x * tiles_y), &mg_img, &sh_img);
    wg_magic((uint32_t)tiles_x, &mg_tx, &sh_tx);
    hipExtLaunchKernelGGL(winograd_conv64_kernel, grid, dim3(WG_TB), 0, st, ev0, ev1, 0, x, u_packed, bias, y, (int)H, (int)W, relu,
                          (int)tiles_x, (int)tiles_y, (int)n_tiles, mg_img, sh_img, mg_tx, sh_tx);
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
