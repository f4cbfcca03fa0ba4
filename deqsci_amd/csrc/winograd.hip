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
//   * ONE PERSISTENT workgroup of 8 wavefronts per CU (two per SIMD, <= 256 registers, 92 KB LDS), walking a contiguous
//     run of block tiles of "its" XCD (neighbouring tiles share halo pixels in that XCD's L2).  64 tiles per CU is the
//     most the register file holds accumulators for (256 KB), and it is what one pass over the 256 KB of transformed
//     weights is amortised over.  Wave w owns tiles [16*(w>>1), +16) x couts [32*(w&1), +32) of every M[xi]: 16 x 2
//     accumulators of v_mfma_f32_16x16x4_f32 (128 registers) that hold every value the output transform of its
//     (tile, cout) needs.
//   * the 64 input channels are consumed in chunks of 8, and the chunk pipeline runs ACROSS block tiles: one stage =
//     MFMA phase + input transform + ONE barrier, with
//       - weights U(c+1): host-packed in LDS order, so the 32 KB chunk is a LINEAR copy global -> LDS by the DMA path
//         (global_load_lds_dwordx4: no registers, no ds_write): every wave moves 4 KB in 4 instructions that differ only
//         in their immediate offset; a lane's weight operands for one xi are ONE conflict-free ds_read_b128 feeding 4 MFMAs;
//       - raw input of chunk c+2 (18 x 18 pixels x 8 channels, possibly of the NEXT block tile): two coalesced float4 per
//         lane into registers, written to the other of two 13.7 KB LDS tiles after the MFMAs (pixel stride 10 floats,
//         pixel rows 2,3,6,7,.. shifted by one pixel: the per-lane patch reads hit 32 distinct banks);
//       - each MFMA lane (tile i = lane&15, channel pair q = lane>>4) reads ITS OWN 4x4 patch of chunk c+1 while the
//         MFMAs of chunk c run and turns it into V = B^T d B (its operands for all 16 xi) in registers afterwards;
//     so a block tile has no prologue of its own: its first two raw chunks and first weight chunk are in flight while the
//     previous tile finishes, and only the output transform sits between two tiles' MFMAs.
//   * the matrix core gets the WEIGHTS as its A operand: D rows (4 per lane, consecutive registers) are 4 consecutive
//     couts of one tile, so the epilogue (Y = A^T M A, ReLU) is per-lane register work ending in 16-byte stores; the bias
//     is the initial value of the xi = 5 accumulator (its coefficient in all four outputs is 1).
//
// What bounds it (tools/ubench/mfma_valu_mix.hip, cu_fill_rate.hip, winograd_stamps.py; DESIGN.md has the numbers):
//   * the two waves of a SIMD do not overlap their MFMA streams with each other's other work for free: while one wave
//     issues MFMAs back to back, its sibling gets ONE instruction of any kind per MFMA, and two MFMA streams are simply
//     serialised.  A chunk therefore costs a SIMD 2 x (64 MFMAs + the ~60 LDS/DMA/wait instructions inside the MFMA
//     phase) plus whatever is left outside; every instruction that is not an MFMA was counted and cut: packed
//     (v_pk_add_f32) transforms, bias folded into the accumulator, multiply-shift tile decode, immediate-offset DMA,
//     compile-time LDS buffer parity;
//   * three things the compiler must not be allowed to do, all measured:
//       - __syncthreads() is a fence + s_barrier and the fence becomes `s_waitcnt vmcnt(0)`: it would drain the loads and
//         DMA pieces deliberately left in flight across the barrier -> wg_lds_barrier() (lgkmcnt only);
//       - after __builtin_amdgcn_global_load_lds hipcc cannot tell which LDS bytes the DMA writes and puts vmcnt(0) in
//         front of the next ds_read of ANY LDS array -> the DMA is inline asm and its wait is placed by hand;
//       - a register load hipcc believes pending on some path makes it wait before the registers' next use - and with it
//         for every younger DMA piece -> raw_landed() tells it, on every path, that the raw loads are complete.
//   (WG_WAVES=4 builds the 32-tile variant with two independent workgroups per CU; it measures 2-3 % slower.)
#include "common.hpp"
#include <hip/hip_ext.h>
#include <type_traits>
#pragma clang diagnostic ignored "-Winline-asm"   // m0 is named as a clobber of the LDS-DMA asm below, on purpose

namespace deqsci {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int WG_CK = 8;                      // input channels per chunk
constexpr int WG_NCHUNK = 64 / WG_CK;
#ifndef WG_ABL
#define WG_ABL 0                              // timing ablations only (tools/ubench): 1 = no patch reads, 2 = no DMA / raw fetch
#endif
#ifndef WG_WAVES
#define WG_WAVES 8                            // wavefronts per workgroup: 8 (one workgroup per CU) or 4 (two per CU)
#endif
constexpr int WG_TB = 64 * WG_WAVES;
constexpr int WG_TROWS = WG_WAVES;            // Winograd tile rows of a block tile (8 tiles per row, 16 tiles per wave pair)
constexpr int WG_RAW_PS = 10;                 // floats per staged pixel (8 channels + 2)
constexpr int WG_RAW_COLS = 18, WG_RAW_ROWS = 2 * WG_TROWS + 2;
constexpr int WG_RAW_RS = (WG_RAW_COLS + 1) * WG_RAW_PS;      // 190 floats per staged pixel row (one spare pixel for the shift)
constexpr int WG_RAW_BUF = WG_RAW_ROWS * WG_RAW_RS;           // 3420 floats = 13.7 KB
constexpr int WG_U_CHUNK = 16 * 2 * 64 * 4;   // floats of one weight chunk in LDS (32 KB)

__device__ __forceinline__ int wg_row_shift(int r) { return (r >> 1) & 1; }

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
// ~40-instruction software division, which matters because an instruction here costs a full MFMA slot of the sibling wave.
__device__ __forceinline__ int wg_div(int t, uint32_t mg, uint32_t sh) { return (int)(((uint64_t)(uint32_t)t * mg) >> sh); }

// tiles_x / tiles_y: block tiles per image row / column; n_tiles = images * tiles_x * tiles_y;
// (mg_img, sh_img) / (mg_tx, sh_tx): division magic for tiles_x * tiles_y and tiles_x
__global__ __launch_bounds__(WG_TB, 2) void winograd_conv64_kernel(const float* __restrict__ x, const float* __restrict__ Ug,
                                                                   const float* __restrict__ bias, float* __restrict__ y,
                                                                   int H, int W, int relu, int tiles_x, int tiles_y, int n_tiles,
                                                                   uint32_t mg_img, uint32_t sh_img, uint32_t mg_tx, uint32_t sh_tx) {
    __shared__ __attribute__((aligned(16))) float Us[2 * WG_U_CHUNK];             // 2 x U[xi][cout half][MFMA lane][j][2]   64 KB
    __shared__ __attribute__((aligned(16))) float Raw[2 * WG_RAW_BUF];            // 2 x raw chunk tile                    27.4 KB
    __shared__ __attribute__((aligned(16))) float bias_s[64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = (int)wg_uniform((uint32_t)(tid >> 6));          // in an SGPR: the DMA bookkeeping is scalar code
    const int wt = wave >> 1, wn = wave & 1;

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

    // ---- raw staging role: chunk tile = 180 pixels x 2 float4; lane e handles (pixel e>>1, half e&1), e = tid, tid + 256
    constexpr int RAW_F4 = WG_RAW_ROWS * WG_RAW_COLS * 2;          // 648 <= 2 * WG_TB
    int rdst[2], rpr[2], rpc[2];                                   // LDS float offset (-1 = idle), staged pixel row / column
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int e = k * WG_TB + tid;
        const int pix = e >> 1, q4 = e & 1;
        rpr[k] = pix / WG_RAW_COLS;
        rpc[k] = pix - rpr[k] * WG_RAW_COLS;
        rdst[k] = e < RAW_F4 ? rpr[k] * WG_RAW_RS + (rpc[k] + wg_row_shift(rpr[k])) * WG_RAW_PS + 4 * q4 : -1;
    }
    // fetch stream state: addresses of the block tile that chunk g+2 belongs to
    uint32_t roff[2];                                              // global byte offset (clamped into the image)
    bool rok[2];                                                   // pixel inside the image (else zero)
    bool border = true;                                            // (uniform) some staged pixel of the tile is outside the image
    const float* xf = x;                                           // image base of the fetch stream
    auto set_fetch_tile = [&](int t) {
        const int n = wg_div(t, mg_img, sh_img), r = t - n * (tiles_x * tiles_y);
        const int by = wg_div(r, mg_tx, sh_tx), bx = r - by * tiles_x;
        xf = x + (int64_t)n * H * W * 64;
        const int py0 = 2 * WG_TROWS * by - 1, px0 = 16 * bx - 1;  // image coordinates of staged pixel (0,0)
        border = py0 < 0 || px0 < 0 || py0 + WG_RAW_ROWS > H || px0 + WG_RAW_COLS > W;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int iy = py0 + rpr[k], ix = px0 + rpc[k];
            rok[k] = rdst[k] >= 0 && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const int cy = iy < 0 ? 0 : (iy >= H ? H - 1 : iy), cx = ix < 0 ? 0 : (ix >= W ? W - 1 : ix);
            roff[k] = ((uint32_t)(cy * W + cx) * 64u + 4u * (uint32_t)(tid & 1)) * 4u;   // < 2^32: H*W < 2^24 (launcher)
        }
    };
    // Two register sets: the chunk fetched during stage c is stored to LDS at the end of stage c+1, a whole stage (about
    // 5000 cycles, more than an HBM round trip under load) later - the first touch of a 128-byte line (chunks 0 and 4 of a
    // tile) would otherwise be waited for twice per tile.  Each set carries the in-image flags of the tile it was fetched from.
    float4 rawv[2][2];
    bool rawok[2][2], rawbd[2] = {true, true};
    // clamped address, no branches; NOTHING here may consume the loaded value (that would park the wave on the memory
    // latency): out-of-image pixels are zeroed in store_raw
    auto fetch_raw_k = [&](int set, int c, int k) {
        rawv[set][k] = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(xf + c * WG_CK) + roff[k]);
        rawok[set][k] = rok[k];
        rawbd[set] = border;
    };
    auto store_raw = [&](int set, int buf) {
        if (rawbd[set]) {                                          // uniform branch: interior tiles skip the selects
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (rdst[k] >= 0) {
                    float* dst = Raw + buf * WG_RAW_BUF + rdst[k];                      // 8-B aligned (pixel stride 40 B)
                    const float4 val = rawok[set][k] ? rawv[set][k] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    *reinterpret_cast<float2*>(dst) = make_float2(val.x, val.y);
                    *reinterpret_cast<float2*>(dst + 2) = make_float2(val.z, val.w);
                }
        } else {
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if (rdst[k] >= 0) {
                    float* dst = Raw + buf * WG_RAW_BUF + rdst[k];
                    *reinterpret_cast<float2*>(dst) = make_float2(rawv[set][k].x, rawv[set][k].y);
                    *reinterpret_cast<float2*>(dst + 2) = make_float2(rawv[set][k].z, rawv[set][k].w);
                }
        }
    };
    // Unconditional "use" of the raw registers: tells hipcc's wait-count pass that the loads have landed on every path
    // (the stores above are predicated, and a load it believes pending makes it wait - for younger DMA pieces too).
    auto raw_landed = [&](int set) { asm volatile("" ::"v"(rawv[set][0].x), "v"(rawv[set][1].x)); };
    // ---- weight chunk: DMA global -> LDS.  The chunk is host-packed in LDS order, so it is a linear 32 KB copy: wave w
    // moves bytes [8 KiB * w, +8 KiB) in 8 instructions that differ only in their immediate offset (which the hardware adds
    // to the global AND the LDS address): one scalar base and one M0 value per chunk, no address arithmetic per piece.
    const uint32_t us_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) float*)Us;
    constexpr int DMA_WAVE_BYTES = WG_U_CHUNK * 4 / WG_WAVES;                          // 4 KiB (8 waves) or 8 KiB (4 waves)
    const uint32_t dma_voff = (uint32_t)(wave * DMA_WAVE_BYTES + lane * 16);            // bytes
    // returns the scalar global base of the chunk (+ half a wave share); sets M0 = LDS base of this wave's share (+ half).
    // The base is handed to the pieces as a VALUE: kept in a by-reference variable it ended up in vector registers.
    auto dma_u_setup = [&](int c, int buf) -> uint64_t {
        const uint64_t g = (uint64_t)(Ug + (int64_t)c * WG_U_CHUNK) + DMA_WAVE_BYTES / 2;
        const uint32_t dma_m0 = wg_uniform(us_lds + (uint32_t)(buf * WG_U_CHUNK * 4 + wave * DMA_WAVE_BYTES + DMA_WAVE_BYTES / 2));
        asm volatile("s_mov_b32 m0, %0" ::"s"(dma_m0) : "m0");                         // nothing else in this kernel uses M0
        return ((uint64_t)wg_uniform((uint32_t)(g >> 32)) << 32) | wg_uniform((uint32_t)g);
    };
#define WG_DMA_PIECE(OFF) asm volatile("global_load_lds_dwordx4 %0, %1 offset:" #OFF ::"v"(dma_voff), "s"(dg) : "memory")
    constexpr int DMA_PIECES = DMA_WAVE_BYTES / 1024;
    auto dma_u_piece = [&](uint64_t dg, int j) {                                                     // KiB j of this wave's share
        switch (j - DMA_PIECES / 2) {
            case -4: WG_DMA_PIECE(-4096); break;
            case -3: WG_DMA_PIECE(-3072); break;
            case -2: WG_DMA_PIECE(-2048); break;
            case -1: WG_DMA_PIECE(-1024); break;
            case 0: WG_DMA_PIECE(0); break;
            case 1: WG_DMA_PIECE(1024); break;
            case 2: WG_DMA_PIECE(2048); break;
            default: WG_DMA_PIECE(3072); break;
        }
    };

    f32x4 acc[16][2];                                        // written, not accumulated, by the first chunk of every tile

    // MFMA roles: lane (i = lane&15, q = lane>>4) owns tile 16*wt + i and channels {2q, 2q+1} of the chunk
    const int mi = lane & 15, mq = lane >> 4;
    const int tl_a = 16 * wt + mi;
    int prow[4];                                             // this lane's 4 patch rows (with their shift) in a raw tile
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) {
        const int r = 2 * (tl_a >> 3) + pr;
        prow[pr] = r * WG_RAW_RS + (2 * (tl_a & 7) + wg_row_shift(r)) * WG_RAW_PS + 2 * mq;
    }
    const float* ub = Us + (wn * 64 + lane) * 4;             // this lane's weight operands of xi = 0 in buffer 0

    // B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]:  V = B^T d B
    f32x2 dn[16], v[16];                                     // register pairs: every transform step is ONE v_pk_add_f32
    auto read_patch = [&](int buf) {
        const float* pp = Raw + buf * WG_RAW_BUF;
#pragma unroll
        for (int pr = 0; pr < 4; ++pr)
#pragma unroll
            for (int pc = 0; pc < 4; ++pc) dn[pr * 4 + pc] = *reinterpret_cast<const f32x2*>(pp + prow[pr] + pc * WG_RAW_PS);
    };
    auto transform = [&]() {
        f32x2 w[16];
#pragma unroll
        for (int pc = 0; pc < 4; ++pc) {                      // rows: w = B^T d (column pc of the patch)
            const f32x2 d0 = dn[pc], d1 = dn[4 + pc], d2 = dn[8 + pc], d3 = dn[12 + pc];
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
    // (stores are non-temporal: a layer's output is read again only by the next launch, and kept in L2 it evicts the input
    // lines this launch re-reads for every cin chunk: -3 % here, -9 % in the F(4x4,3x3) kernel)
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
                st4s(oj, make_float4(o00[0][0], o00[0][1], o00[1][0], o00[1][1]));
                if (inx) st4s(oj + 64, make_float4(o01[0][0], o01[0][1], o01[1][0], o01[1][1]));
                if (iny) {
                    st4s(oj + (int64_t)W * 64, make_float4(o10[0][0], o10[0][1], o10[1][0], o10[1][1]));
                    if (inx) st4s(oj + (int64_t)W * 64 + 64, make_float4(o11[0][0], o11[0][1], o11[1][0], o11[1][1]));
                }
            }
        }
    };

#ifdef WG_STAMP
#ifndef WG_STAMP_TID
#define WG_STAMP_TID 0
#endif
#ifndef WG_STAMP_SKIP
#define WG_STAMP_SKIP 0                         // marks to skip before recording 38 of them
#endif
    unsigned long long* stamp_out = reinterpret_cast<unsigned long long*>(const_cast<float*>(bias)) + (size_t)blockIdx.x * 40;
    __shared__ unsigned long long stamp_lds[40];              // stamps go to LDS: a global store would count in vmcnt
    int stamp_i = 0;
#define WG_MARK() do { if (stamp_i >= WG_STAMP_SKIP && stamp_i < WG_STAMP_SKIP + 38) { if (tid == WG_STAMP_TID) stamp_lds[stamp_i - WG_STAMP_SKIP] = __builtin_readcyclecounter(); } ++stamp_i; } while (0)
    bias = nullptr;
#else
#define WG_MARK() do { } while (0)
#endif
    WG_MARK();                                                // 0: kernel entry
    // ---- prologue (once per workgroup): bias, U(0), raw(0), raw(1) staged; V(0) computed; fetch stream at chunk 2
    if (tid < 64) bias_s[tid] = bias ? bias[tid] : 0.0f;
    set_fetch_tile(t_first);
    {
        const uint64_t dg = dma_u_setup(0, 0);
#pragma unroll
        for (int j = 0; j < DMA_PIECES; ++j) dma_u_piece(dg, j);
    }
    fetch_raw_k(0, 0, 0); fetch_raw_k(0, 0, 1);
    store_raw(0, 0);
    fetch_raw_k(0, 1, 0); fetch_raw_k(0, 1, 1);
    store_raw(0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_landed(0);
    fetch_raw_k(1, 2, 0); fetch_raw_k(1, 2, 1);               // raw(2): stored at the end of stage 0
    __syncthreads();
    read_patch(0);
    transform();
    __syncthreads();                                          // every wave has read its patch of raw(0): stage 0 overwrites it
    WG_MARK();                                                // 1: prologue done
    int t_fetch = t_first;                                    // tile of the fetch stream
    // One pipeline stage = chunk c of the current tile.  PAR = c&1 selects the LDS buffers (compile-time: every LDS address
    // of the stage is an immediate offset), FIRST = chunk 0: the accumulators are written from zero / the bias.
    // (always_inline: called as a function, the by-reference captures - accumulators included - live in scratch memory.)
    auto stage = [&](auto par_c, auto first_c, int c, int t_cur) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool FIRST = decltype(first_c)::value;
        // entry: Us[PAR] = U(c), Raw[PAR^1] = raw(c+1) visible; v = V(c) in registers; everyone is done with raw(c) in Raw[PAR]
        WG_MARK();
        // raw(c+2), requested a stage ago, goes to LDS BEFORE the MFMA phase: the second wave of a SIMD does it while it
        // waits for its turn on the matrix pipe, and nothing but the input transform is left between the MFMAs and the barrier
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        raw_landed(PAR ^ 1);
        store_raw(PAR ^ 1, PAR);
        // ---- MFMA phase.  Hand-ordered: everything that is not an MFMA sits right behind the FIRST of the four MFMAs of
        // a transform position.  Chunk indices past the end of the run are clamped instead of branched around (a
        // duplicate fetch of valid memory nobody reads).
        const float* ubc = ub + PAR * WG_U_CHUNK;
        const float* ppn = Raw + (PAR ^ 1) * WG_RAW_BUF;
        const int cf = (c + 3) & 7;                           // chunk of the fetch stream within its tile
        f32x4 init5[2];
        if (FIRST) {
#pragma unroll
            for (int j = 0; j < 2; ++j) init5[j] = *reinterpret_cast<const f32x4*>(bias_s + 32 * wn + 16 * j + 4 * (lane >> 4));
        }
        const uint64_t dg = dma_u_setup((c + 1) & 7, PAR ^ 1);   // weight chunk to stage, into the buffer M(c-1) released
        constexpr int PF = 2;                                 // weight operands are read PF transform positions ahead
        float4 bq[PF + 1];
#pragma unroll
        for (int i = 0; i < PF; ++i) bq[i] = *reinterpret_cast<const float4*>(ubc + i * (2 * 64 * 4));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            if (xi + PF < 16) bq[(xi + PF) % (PF + 1)] = *reinterpret_cast<const float4*>(ubc + (xi + PF) * (2 * 64 * 4));
            if ((xi & 1) == 0 && !(WG_ABL & 1)) {             // two patch elements per instruction (ds_read2_b64)
                dn[xi] = *reinterpret_cast<const f32x2*>(ppn + prow[xi >> 2] + (xi & 3) * WG_RAW_PS);
                dn[xi + 1] = *reinterpret_cast<const f32x2*>(ppn + prow[xi >> 2] + ((xi & 3) + 1) * WG_RAW_PS);
            }
            const float4 b = bq[xi % (PF + 1)];
            const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
            __builtin_amdgcn_sched_barrier(0);
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.x, v[xi].x, FIRST ? (xi == 5 ? init5[0] : zero) : acc[xi][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (WG_ABL & 2) { }
            else if (xi < DMA_PIECES) dma_u_piece(dg, xi);    // weights first: they are needed one stage from now,
            else if (xi == DMA_PIECES) fetch_raw_k(PAR, cf, 0);   // the raw chunk three stages from now, into the set
            else if (xi == DMA_PIECES + 1) fetch_raw_k(PAR, cf, 1);   // stage c-1 stored from
            __builtin_amdgcn_sched_barrier(0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.z, v[xi].x, FIRST ? (xi == 5 ? init5[1] : zero) : acc[xi][1], 0, 0, 0);
            acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.y, v[xi].y, acc[xi][0], 0, 0, 0);
            acc[xi][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(b.w, v[xi].y, acc[xi][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        WG_MARK();
        transform();                                          // V(c+1) from the patch read during the MFMAs
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");      // this wave's part of U(c+1) is in LDS (the two raw loads of this
                                                              // stage stay in flight)
        wg_lds_barrier();                                     // U(c+1), raw(c+2) visible; every wave is done with Us[PAR]
        WG_MARK();
        if (c == 4) {                                         // chunks c+4.. of the fetch stream belong to the next tile
            if (t_fetch + t_step < t_end) t_fetch += t_step;  // (past the end of the run: stay, the fetches are dummies)
            set_fetch_tile(t_fetch);
        } else if (c == 7) {
            epilogue(t_cur);                                  // its stores drain under the next tile's MFMAs
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
    if (tid == WG_STAMP_TID) {
        for (int i = 0; i < 38 && i < stamp_i - WG_STAMP_SKIP; ++i) stamp_out[i] = stamp_lds[i];
        stamp_out[38] = __builtin_amdgcn_s_getreg((31 << 11) | 20);      // HW_REG_XCC_ID
        stamp_out[39] = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID: wave/simd/cu/sh/se
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
    // 32-bit arithmetic in the kernel: tile indices, and the per-image BYTE offset of a pixel (H*W*64 channels*4 B < 2^32)
    if (n_tiles > (int64_t)INT32_MAX / 16 || H * W >= (int64_t)1 << 24) return DEQSCI_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t resident = (8 / WG_WAVES) * (int64_t)num_cus(); // persistent workgroups: 16 wavefronts (2 per SIMD) on every CU
    const dim3 grid((unsigned)(n_tiles < resident ? n_tiles : resident));
    uint32_t mg_img, sh_img, mg_tx, sh_tx;
    wg_magic((uint32_t)(tiles_x * tiles_y), &mg_img, &sh_img);
    wg_magic((uint32_t)tiles_x, &mg_tx, &sh_tx);
    // (the timed entry point stamps ev0/ev1 with the dispatch's own begin/end; the plain one is an ordinary launch, which
    // is what a stream capture - DEQSCIEngine's hipGraph - records)
    if (ev0 || ev1)
        hipExtLaunchKernelGGL(winograd_conv64_kernel, grid, dim3(WG_TB), 0, st, ev0, ev1, 0, x, u_packed, bias, y, (int)H, (int)W, relu,
                              (int)tiles_x, (int)tiles_y, (int)n_tiles, mg_img, sh_img, mg_tx, sh_tx);
    else
        hipLaunchKernelGGL(winograd_conv64_kernel, grid, dim3(WG_TB), 0, st, x, u_packed, bias, y, (int)H, (int)W, relu,
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
