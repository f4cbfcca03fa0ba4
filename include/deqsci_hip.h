/*
 * deqsci_hip.h - C ABI of libdeqsci_hip.so: the MI355X (gfx950) kernels of the DEQ-SCI
 * reconstruction hot path (SCI sensing operators, GAP projection, Anderson fixed-point
 * bookkeeping).  Plain pointers and sizes only - no torch / C++ types cross this boundary.
 *
 * The reference (IndigoPurple/DEQSCI) is pure Python and has no FFI; each entry point below
 * replaces the ATen expression(s) the reference evaluates at the cited file:line.  The binding a
 * reference maintainer would add is a ctypes stub - see INTEGRATION.md.
 *
 * Conventions
 *   - All tensor pointers are DEVICE pointers to contiguous fp32, 16-byte aligned.
 *   - Layouts: DEQSCI_LAYOUT_HWB = (bsz,H,W,B), frames innermost - the reference's API layout
 *     (solvers/equilibrium_solvers_yaping.py:397); DEQSCI_LAYOUT_BHW = (bsz,B,H,W) planar - the
 *     denoiser's (bsz*B,1,H,W) layout (ibid. :415).  y / Phi_sum are always (bsz,H,W).
 *   - phi_shared != 0: Phi (and Phi_sum) carry no batch dimension - one mask for every
 *     measurement of a clip (training/sci_equilibrium_training.py:161-176).
 *   - Ownership: the caller owns every buffer including scratch; no entry point allocates,
 *     frees or synchronises.  Work is enqueued on `stream`; entry points are re-entrant,
 *     hold no global state and are safe under HIP-graph stream capture.
 *   - Return: 0 on success; >0 = hipError_t of the failed launch; <0 = DEQSCI_ERR_*.
 */
#ifndef DEQSCI_HIP_H
#define DEQSCI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DEQSCI_LAYOUT_HWB 0
#define DEQSCI_LAYOUT_BHW 1

#define DEQSCI_MAX_M 8            /* largest Anderson history depth (reference uses m=5) */
#define DEQSCI_PART_STRIDE (DEQSCI_MAX_M + 1)

#define DEQSCI_ERR_NULL        (-1)   /* required pointer is NULL                      */
#define DEQSCI_ERR_SHAPE       (-2)   /* non-positive / inconsistent sizes             */
#define DEQSCI_ERR_ALIGN       (-3)   /* pointer not 16-byte aligned                   */
#define DEQSCI_ERR_UNSUPPORTED (-4)   /* m > DEQSCI_MAX_M, unknown layout, ...         */

typedef void* deqsci_stream_t;    /* a hipStream_t (NULL = the legacy default stream) */

const char* deqsci_version(void);
const char* deqsci_error_string(int code);

/* K1  y = Phi x : y[n,h,w] = sum_b x[n,h,w,b] * Phi[n,h,w,b]
 *     replaces A_torch_, utils/cg_utils.py:85-90.  x and phi share `layout`. */
int deqsci_sci_forward_f32(const float* x, const float* phi, float* y,
                           int64_t bsz, int64_t H, int64_t W, int64_t B,
                           int layout, int phi_shared, deqsci_stream_t stream);

/* K2  x = Phi^T y : x[n,h,w,b] = y[n,h,w] * Phi[n,h,w,b]
 *     replaces At_torch_ (utils/cg_utils.py:124-129) and initial_point (:228-229). */
int deqsci_sci_adjoint_f32(const float* y, const float* phi, float* x,
                           int64_t bsz, int64_t H, int64_t W, int64_t B,
                           int layout, int phi_shared, deqsci_stream_t stream);

/* O4  Phi_sum = sum_b Phi with zeros replaced by 1
 *     replaces training/sci_equilibrium_training.py:162-163.  nb = bsz, or 1 for a shared mask. */
int deqsci_phi_sum_f32(const float* phi, float* phisum,
                       int64_t nb, int64_t H, int64_t W, int64_t B,
                       int layout, deqsci_stream_t stream);

/* K3  fused GAP projection  z1 = z + Phi^T((y - Phi z) / Phi_sum)
 *     replaces solvers/equilibrium_solvers_yaping.py:399-400 (5 ATen kernels, 4 temporaries)
 *     and, when layout_out == BHW with layout_in == HWB, also the permute+contiguous at :415/:419.
 *     z and phi are in layout_in; z1 is written in layout_out.  z1 may alias z iff the layouts match. */
int deqsci_gap_update_f32(const float* z, const float* phi, const float* y, const float* phisum,
                          float* z1, int64_t bsz, int64_t H, int64_t W, int64_t B,
                          int layout_in, int layout_out, int phi_shared, deqsci_stream_t stream);

/* (bsz,H,W,B) <-> (bsz,B,H,W) through an LDS tile; `to_layout` is the layout of `out`. */
int deqsci_transpose_f32(const float* in, float* out,
                         int64_t bsz, int64_t H, int64_t W, int64_t B,
                         int to_layout, deqsci_stream_t stream);

/* K4b out = z1 - noise, z1 and noise planar (BHW), out in layout_out
 *     replaces `z - noise.view(bsz,c,w,h).permute(0,2,3,1)`, equilibrium_solvers_yaping.py:417,420. */
int deqsci_residual_out_f32(const float* z1, const float* noise, float* out,
                            int64_t bsz, int64_t H, int64_t W, int64_t B,
                            int layout_out, deqsci_stream_t stream);

/* ---- Anderson / Picard bookkeeping (solvers/new_equilibrium_utils_yaping.py:153-189, 213-222) ----
 * History buffers F_hist, G_hist are (bsz, m, N) with G = F - X kept instead of X
 * (X_i = F_i - G_i), N = H*W*B in whatever element order the caller iterates in.
 * `deqsci_anderson_chunks(bsz,N)` blocks per sample each emit DEQSCI_PART_STRIDE fp32 partial sums.
 */
int64_t deqsci_anderson_chunks(int64_t bsz, int64_t N);
size_t  deqsci_partials_bytes(int64_t bsz, int64_t N);           /* `partials` buffer            */
size_t  deqsci_gram_bytes(int64_t bsz);                          /* persistent fp64 Gram + norms + arrival
                                                                    ticket; ZERO it once before first use */

/* K4  F_k = z1 - noise (noise may be NULL: F_k = z1);  G_k = F_k - x_cur;  store both into history
 *     slot `slot`; optionally x_next = F_k; per-block partial sums of <G_k,G_j> for j < n_filled and
 *     of |F_k|^2.  replaces :163/:183 stores, the G rows of :177 and the two norms of :184. */
int deqsci_residual_store_f32(const float* z1, const float* noise, const float* x_cur,
                              float* F_hist, float* G_hist, float* x_next, float* partials,
                              int64_t bsz, int64_t N, int m, int slot, int n_filled,
                              deqsci_stream_t stream);

/* K5+K6  finish the reduction (fp64), refresh row/column `slot` of the persistent Gram matrix,
 *     solve the bordered (n+1)x(n+1) system [[0,1^T],[1,GG^T+lam I]] [nu;alpha] = e0 by LU with
 *     partial pivoting, one wavefront per sample, and emit the relative residual
 *     res[0] = |G_k| / (eps + |F_k|) over the whole batch (reference semantics, :184) and
 *     res[1+s] per sample.  replaces torch.bmm + torch.solve + 2x .item() at :178-184.
 *     n = 0 skips the solve (ramp-up / Picard).  alpha is (bsz, DEQSCI_MAX_M). */
int deqsci_anderson_solve_f32(const float* partials, void* gram, float* alpha, float* res,
                              int64_t bsz, int64_t N, int m, int slot, int n_filled, int n,
                              float lam, float eps, deqsci_stream_t stream);

/* The same with the REFERENCE's arithmetic for alpha: gram32 = the n x n block G G^T of the first n history rows as the caller's fp32
 *     GEMM produced it ((bsz, n, n) row-major, rows in slot order: torch.bmm(G[:, :n], G[:, :n]^T), :178), the bordered system formed
 *     and factorised in fp32 (torch.solve = sgesv, :180).  The relative residual still comes from the exact sums.  gram32 NULL = the
 *     entry point above.  Why it exists: DESIGN.md section 5, "Config 2" - the ~5e-6 rounding error of that GEMM at N = 2^19 is worth
 *     +0.02 dB on the reference's 180-iteration FFDNet ensemble. */
int deqsci_anderson_solve_gram_f32(const float* partials, void* gram, float* alpha, float* res,
                                   int64_t bsz, int64_t N, int m, int slot, int n_filled, int n,
                                   float lam, float eps, const float* gram32, deqsci_stream_t stream);

/* The same WITHOUT a GEMM library (anderson_arith = "reference" of the engine and of the drop-in DEQFixedPoint): the new row of G G^T in the
 *     summation order of the fp32 torch.bmm behind tests/golden (MKL sgemm for 5 x N times N x 5: sixteen interleaved fused-multiply-add
 *     chains per entry, chain c over k = c, c + 16, ..., summed pairwise at the end; within one ulp of torch.bmm, tools/gram_on_real_history.py).
 *     On the loop's heavy-tailed residuals that order ABSORBS the small products of a 2^15-step chain: the diagonal comes out 3-7e-6 too small -
 *     a bias, and the thing that moves the reference's chaotic FFDNet ensembles (an unbiased fp32 sum of the same error size
 *     does not).  G_hist = the history K4 wrote (bsz, m, N), partials = K4's block sums of the same call; ref_state = deqsci_gram_ref_bytes(bsz, N)
 *     bytes, caller-owned, 16-byte aligned, ZEROED once and then left alone between calls (it carries the MAX_M x MAX_M fp32 Gram of each sample:
 *     row / column `slot` is refreshed per call, the other rows are what a deterministic GEMM would recompute bit for bit).
 *     deqsci_gram_row_chain16_f32 computes the 16 chain sums of every entry <G_slot, G_j>, j < n_filled, into ref_state: serial = 1 runs the chains
 *     as they are written (N / 16 dependent FMAs, ~220 us at N = 2^19); serial = 0 produces THE SAME BITS in two passes - inside one binade of the
 *     running sum a chain step is S + RN_ulp(p), an integer sum that any number of workgroups can form in any order; only the binade crossings are
 *     walked term by term (csrc/anderson.hip) - and falls back to serial = 1 where N % 4 != 0.  deqsci_anderson_solve_ref_f32 = that (serial = 0),
 *     then the bordered system formed and factorised in fp32 (:180, sgesv); residuals and the float64 Gram of `gram` as by the entry points above. */
size_t deqsci_gram_ref_bytes(int64_t bsz, int64_t N);
int deqsci_gram_row_chain16_f32(const float* G_hist, const float* partials, float* ref_state,
                                int64_t bsz, int64_t N, int m, int slot, int n_filled, int serial,
                                deqsci_stream_t stream);
int deqsci_anderson_solve_ref_f32(const float* G_hist, const float* partials, float* ref_state, void* gram, float* alpha, float* res,
                                  int64_t bsz, int64_t N, int m, int slot, int n_filled, int n,
                                  float lam, float eps, deqsci_stream_t stream);

/* (round 6) K4 and the first of those two passes in ONE launch: deqsci_residual_store_ref_f32 = deqsci_residual_store_f32 (same F / G / x_next, the
 *     same block sums in `partials`, bit for bit) whose blocks also leave the records of deqsci_gram_row_chain16_f32's first pass in ref_state - the
 *     history rows are in the block's registers anyway; the binade a block rounds for comes from its predecessors' block sums, which the blocks
 *     publish to one another inside the launch (csrc/anderson.hip: residual_store_round_kernel).  Only where deqsci_gram_ref_fusable(bsz, N) = 1
 *     (blocks of 2048 elements - N bsz <= 2^25 -, N a whole number of at most 256 of them: 256 x 256 x 8 is exactly that), DEQSCI_ERR_UNSUPPORTED otherwise - then the caller uses the two
 *     entry points above.  deqsci_anderson_apply_solve_ref_f32 = deqsci_anderson_solve_ref_f32 without that first pass: it must follow a
 *     deqsci_residual_store_ref_f32 of the same call (same ref_state, slot, n_filled) on the same stream.  replaces :163,:177-184 like the pair above. */
int deqsci_gram_ref_fusable(int64_t bsz, int64_t N);
int deqsci_residual_store_ref_f32(const float* z1, const float* noise, const float* x_cur,
                                  float* F_hist, float* G_hist, float* x_next, float* partials, float* ref_state,
                                  int64_t bsz, int64_t N, int m, int slot, int n_filled, deqsci_stream_t stream);
int deqsci_anderson_apply_solve_ref_f32(const float* G_hist, const float* partials, float* ref_state, void* gram, float* alpha, float* res,
                                        int64_t bsz, int64_t N, int m, int slot, int n_filled, int n,
                                        float lam, float eps, deqsci_stream_t stream);

/* K7  x_out = beta * sum_i alpha_i F_i + (1-beta) * sum_i alpha_i X_i,  X_i = F_i - G_i   (:182) */
int deqsci_anderson_mix_f32(const float* F_hist, const float* G_hist, const float* alpha,
                            float* x_out, float beta, int n, int64_t bsz, int64_t N, int m,
                            deqsci_stream_t stream);

/* K7+K3  the mix above fused with the GAP projection of its result: writes x_out (the new
 *     iterate X_k, needed for G_k and as the value andersonexp returns) and z1 = GAP(x_out).
 *     History, phi, x_out and z1 all in `layout`. */
int deqsci_anderson_mix_gap_f32(const float* F_hist, const float* G_hist, const float* alpha,
                                float beta, int n, int m,
                                const float* phi, const float* y, const float* phisum,
                                float* x_out, float* z1,
                                int64_t bsz, int64_t H, int64_t W, int64_t B,
                                int layout, int phi_shared, deqsci_stream_t stream);

/* Denoiser epilogue: h = max(h + bias[c], 0) in place (relu = 0: bias only), h (n,c,hw) NCHW-contiguous or
 *     channels_last.  Fuses the BatchNorm(eval)+ReLU of networks/ffdnet/models.py:53-58 - folded into the
 *     preceding conv's weights plus this per-channel bias - into one pass instead of PyTorch's two. */
int deqsci_bias_relu_f32(float* h, const float* bias, int64_t n, int64_t c, int64_t hw,
                         int channels_last, int relu, deqsci_stream_t stream);

/* FFDNet tail: conv3x3(64 -> 4, pad 1, no bias) fused with `upsamplefeatures` (networks/ffdnet/functions.py:62-81,
 *     = pixel_shuffle 2).  h is the channels_last activation (n,H,W,64); w_packed is the (4,64,3,3) weight
 *     re-ordered [half(2)][tap(9)][cin(32)][cout(4)]; out is the planar (n,1,2H,2W) predicted-noise image.
 *     in_bias (64 floats, may be NULL): h is the previous layer's RAW conv output and max(h + in_bias[c], 0)
 *     (its folded BatchNorm bias + ReLU) is applied while the tile is staged - one activation sweep less. */
int deqsci_ffdnet_tail_f32(const float* h, const float* w_packed, const float* in_bias, float* out,
                           int64_t n, int64_t H, int64_t W, deqsci_stream_t stream);

/* The same two stencils without the FFDNet layout layers - SimpleCNN's edge layers
 *     (networks/provable/model/SimpleCNN_models.py:43-57): conv3x3(64 -> 1) from a channels_last (n,H,W,64) activation to
 *     a planar (n,1,H,W) image (w_packed [half(2)][tap(9)][cin(32)], optional in_bias + ReLU on the way in), and
 *     conv3x3(1 -> 64) [+ ReLU] from a planar image to channels_last (w_packed [tap(9)][cout/4(16)][cout%4(4)]). */
int deqsci_conv3x3_c64_to_1_f32(const float* h, const float* w_packed, const float* in_bias, float* out,
                                int64_t n, int64_t H, int64_t W, deqsci_stream_t stream);
int deqsci_conv3x3_c1_to_64_f32(const float* x, const float* w_packed, float* h,
                                int64_t n, int64_t H, int64_t W, int relu, deqsci_stream_t stream);

/* FFDNet head: `concatenate_input_noise_map` (networks/ffdnet/functions.py:16-53: sigma map + 2x2 pixel-unshuffle)
 *     + conv3x3(5 -> 64, pad 1, no bias) + ReLU.  x is the planar (n,1,2H,2W) image, sigma[i*sigma_stride] the noise
 *     level of image i (stride 0 = one value for all), w_packed the (64,5,3,3) weight re-ordered
 *     [ch*9+tap (45)][cout/4 (16)][cout%4 (4)]; h is written as the channels_last (n,H,W,64) activation. */
int deqsci_ffdnet_head_f32(const float* x, const float* w_packed, const float* sigma, int64_t sigma_stride,
                           float* h, int64_t n, int64_t H, int64_t W, deqsci_stream_t stream);

/* conv3x3(64 -> 64, pad 1, stride 1) + per-channel bias + ReLU as Winograd F(2x2,3x3) on the fp32 matrix cores.
 *     x, y channels_last (n,H,W,64), x != y; u_packed = the (64,64,3,3) weight transformed U = G g G^T and ordered
 *     [cin chunk c (8)][xi (16)][wn (2)][q (4)][i (16)][j (2)][s (2)], cout = 32 wn + 16 j + i, cin = 8 c + 2 q + s
 *     (the kernel's MFMA lane order); bias (64, may be NULL); relu 0/1.
 *     The middle layers of FFDNet (networks/ffdnet/models.py:53-58, BatchNorm folded) and SimpleCNN. */
int deqsci_conv3x3_c64_winograd_f32(const float* x, const float* u_packed, const float* bias, float* y,
                                    int64_t n, int64_t H, int64_t W, int relu, deqsci_stream_t stream);

/* The same layer as Winograd F(4x4,3x3): 2.25 instead of 4 multiplications per output, the large-batch kernel (block tiles of
 *     16 x 32 output pixels, one persistent workgroup per CU: it wants >= ~2 block tiles per CU; small launches belong to the
 *     F(2x2,3x3) entry point above).  Same arguments; u_packed = U = G g G^T (6x6 per cout, cin) ordered
 *     [cin chunk c (8)][s (18)][rg (2)][cgp (2)][q (4)][i (16)][j (2)][ks (2)] with transform position (3 rg + s / 6, s % 6),
 *     cout = 32 cgp + 16 j + i, cin = 8 c + 2 q + ks.  Rounding ~1.2e-6 per layer (F(2x2,3x3): 2.1e-7). */
int deqsci_conv3x3_c64_winograd44_f32(const float* x, const float* u_packed, const float* bias, float* y,
                                      int64_t n, int64_t H, int64_t W, int relu, deqsci_stream_t stream);

/* The same kernel between the layers of a stack (FFDNet: 13 of them in a row): activations in "blk32" instead of channels_last -
 *     [n][channel chunk (8)][H][ceil(W/32)][32][8]: planes of 8 channels; inside every block of 32 columns, column m at position
 *     8 ((m+1) & 3) + ((m+1) >> 2) - ((m+1) & 3 == 0), i.e. in the order the kernel stages them (its input fetch becomes two
 *     contiguous 512-byte runs per pixel row, its output stores 256 contiguous bytes per tile row).  in_layout / out_layout:
 *     DEQSCI_ACT_NHWC or DEQSCI_ACT_BLK32, independently (first layer NHWC -> BLK32, last BLK32 -> NHWC); the blk32 buffer holds
 *     n * 8 * H * ceil(W/32) * 256 floats, columns >= W are never read or written.  start_event / stop_event: both NULL, or both
 *     raw hipEvent_t handles that receive the dispatch's begin / end (measurement). */
#define DEQSCI_ACT_NHWC 0
#define DEQSCI_ACT_BLK32 1
int deqsci_conv3x3_c64_winograd44_layout_f32(const float* x, const float* u_packed, const float* bias, float* y,
                                             int64_t n, int64_t H, int64_t W, int relu, int in_layout, int out_layout,
                                             deqsci_stream_t stream, void* start_event, void* stop_event);

/* ---- the same layer as a DIRECT convolution on the f16 matrix cores with fp32-class accuracy (csrc/conv_s16.hip): every operand
 *     is two fp16 pieces (x = hi + lo, 22 significant bits), three f16 MFMAs per product (w_hi x_hi + w_lo x_hi + w_hi x_lo), fp32
 *     accumulation.  x_sp16: the activation as [n][4 cin chunks][2 pieces: hi, lo][2 blocks of 8 channels][H][W][8 halfs] holding
 *     2^e x.  w_packed: 2^w_exp w as [4 chunks][9 taps][2 pieces][2 cout groups of 32][64 lanes][8 halfs] (cout = 32 g + lane % 32,
 *     cin = 16 c + 8 (lane / 32) + j), w_exp the power of two that puts max |w| into [2^13, 2^14).
 *
 *     RANGES.  fp32 (the reference's arithmetic, solvers/equilibrium_solvers_yaping.py:397-420) is scale-free, fp16 is not, so the
 *     exponent e of an sp16 activation follows the data - PER IMAGE of the batch, so that one measurement's result never depends on what
 *     else is in the batch.  Every sp16 activation is described by a pair (amax, exp): `amax` a DEVICE pointer to n floats, amax[i] =
 *     max |x| of image i of that activation - then e(i) = DEQSCI_SP16_TARGET_EXP - floor(log2(amax[i])), i.e. 2^e max|x| in
 *     [2^11, 2^12), derived identically by the kernel that writes the activation and the kernel that reads it - or NULL: the fixed
 *     exponent `exp` for every image (DEQSCI_SP16_DEFAULT_EXP = 8 suits activations of a few units).  `track_amax` (may be NULL): a MEASURING launch - the same arithmetic,
 *     but max |y| of image i of the output is folded into track_amax[i] (n floats; zero them first) and, for the 64->64 layer, y itself
 *     is not written: run a layer once with track_amax = the slots of its output, then again with out_amax = those slots.  Nothing crosses to the host: a captured hipGraph follows
 *     its inputs.  An activation that outgrows fp16 (16 x the measured maximum) becomes inf, never a silently wrong finite number.
 *
 *     Output = relu?(conv + bias): out_f32 = 0 -> sp16 with the range (out_amax, out_exp); out_f32 = 1 -> fp32 channels_last
 *     (n,H,W,64).  Images up to 2^31 / 256 - 33 pixels.  start_event / stop_event: both NULL, or both raw hipEvent_t handles. */
#define DEQSCI_SP16_DEFAULT_EXP 8
#define DEQSCI_SP16_TARGET_EXP 11
int deqsci_conv3x3_c64_split16(const void* x_sp16, const void* w_packed, const float* bias, void* y,
                               int64_t n, int64_t H, int64_t W, int relu, int w_exp, const float* in_amax, int in_exp,
                               const float* out_amax, int out_exp, float* track_amax, int out_f32,
                               deqsci_stream_t stream, void* start_event, void* stop_event);
/* A RUN of n_layers such layers (the denoisers' 13 / 2 middle layers, each feeding the next) in ONE launch: the kernel's persistent
 *     workgroups walk their tiles layer after layer with DATAFLOW synchronisation instead of kernel boundaries - a tile of layer l + 1
 *     waits for layer l of itself and its eight neighbour tiles only (one progress word per tile, agent-scope atomics; activations
 *     stored write-through and fetched with agent-scope loads - no cache maintenance, no grid-wide barrier).  Bit-identical to n_layers
 *     single launches.  Two things it buys: at one measurement per call (8 images of 128 x 128 = one tile per CU, the reference's
 *     usage) the kernel boundary that cost a fifth of a layer; and at any batch size a SLICE of the batch whose activations fit the
 *     Infinity Cache (n x H x W x 256 bytes <= 128 MiB: 32 images of 128 x 128) runs all its layers back to back out of that cache -
 *     the caller slices the batch (pointers into it, range_stride = the batch's images) and launches slice after slice.
 *     layers: DEVICE table of n_layers x { const void* w_packed; const float* bias (may be NULL); int32 w_exp; int32 relu } (24 bytes
 *     each).  Layer l reads x_sp16 (l = 0) or the buffer layer l - 1 wrote, and writes y_even (l even) / y_odd (l odd); all sp16,
 *     none aliasing another.  ranges: n_layers + 1 rows of range slots, range_stride (>= n) floats apart - ranges[l * range_stride + i]
 *     = max |image i of the input of layer l|; NULL: the input holds 2^in_exp x, every output 2^out_exp y.
 *     flags: 32 (n_tiles + 1) 32-bit DEVICE words (n_tiles = n ceil(H/16) ceil(W/32); tile t's word is flags[32 t], a 128-byte line
 *     each), zeroed ONCE by the caller and then left alone: the progress words count on from launch to launch (launches of one shape
 *     may share them when they run one after the other; another shape needs its own).  flags[32 n_tiles] != 0 after a launch = a wait
 *     timed out - a workgroup of the launch was not resident, i.e. somebody else holds CUs of this device - and the output of that
 *     launch is invalid; the launch never hangs, and later launches on the same words do not wait at all (zero the words to rearm).
 *     THE RESIDENCY / TIME-OUT CONTRACT (this launch and deqsci_conv3x3_c64_wino16_stack): the workgroups of a launch wait for ONE ANOTHER, so all
 *     of them must be resident at once.  The launcher asks for min(n_tiles, CUs) workgroups, one per CU by their LDS footprint, after an
 *     occupancy query (DEQSCI_ERR_UNSUPPORTED if the kernel cannot be resident at all on this device) - which holds when the device's CUs
 *     are the caller's: no CU mask, no other process, no concurrent kernel of another stream holding CUs for the length of the launch.  It
 *     is NOT enforced by the hardware queue (no cooperative launch: those cannot be captured into a hipGraph).  A caller who cannot
 *     promise it must read flags[32 n_tiles] after the launch (the Python launchers do, and raise; DEQSCIEngine reads it after the first
 *     stack f-call of a reconstruction and at its end, redoes the call with one launch per layer and stays there for 16 calls) - a
 *     waiting workgroup gives up after ~0.25 s, so a broken promise costs time and an invalid output that says so, never a hang. */
int deqsci_conv3x3_c64_split16_stack(const void* x_sp16, void* y_even, void* y_odd, const void* layers, int n_layers,
                                     int64_t n, int64_t H, int64_t W, const float* ranges, int64_t range_stride, int in_exp, int out_exp,
                                     void* flags, deqsci_stream_t stream, void* start_event, void* stop_event);
/* ---- the same layer with a third fewer matrix-core products (csrc/conv_w16.hip): the split-fp16 arithmetic under Winograd F(2,3) ALONG X
 *     nested in the direct sum ALONG Y - per tap row dy and position xi = 0..3 the weights U[dy][xi] = G g[dy][.] (G of F(2,3), formed in
 *     float64 by the caller), the input transform B^T d of every halo row in the kernel (fp32, then hi + lo), ONE fp32 accumulation chain of
 *     36 MFMAs per position, y[2t] = M0 + M1 + M2, y[2t+1] = M1 - M2 - M3.  18 f16 MFMA products per output and (cin, cout) instead of 27;
 *     per layer against float64 on FFDNet's own data the same 1.3e-7 as the direct kernel (tools/wino16_numerics.py).
 *     Replaces nn.Conv2d(64,64,3,padding=1) + BatchNorm(eval) + ReLU, networks/ffdnet/models.py:53-58 (the 13 middle layers :46-64).
 *     u_packed: 2^w_exp U as [4 cin chunks][2 xi halves][2 xi'][3 dy][2 pieces: hi, lo][2 cout groups of 32][64 lanes][8 halfs]
 *     (xi = 2 half + xi', cout = 32 g + lane % 32, cin = 16 c + 8 (lane / 32) + j), w_exp the power of two that puts max |U| into [2^13, 2^14).
 *     Activations x, y in "p32": the 16 planes of 16-byte pixels of sp16 holding 2^e x as fp32 instead of hi + lo fp16 -
 *     [n][8 blocks of 8 channels][2 halves of 4][H][ceil(W/64) column blocks][2 column parities][32][4 floats]: plane 2 b8 + j = channels
 *     8 b8 + 4 j .. + 3, a row is 64 ceil(W/64) pixels long (the tail of the last block is padding nobody reads), and inside a block of 64
 *     columns the 32 EVEN columns come first, then the 32 odd ones - an F(2,3) tile pair reads and writes whole 128-byte lines (nothing to
 *     join in front of the transform, nothing to split behind the output transform); FFDNet's first and last layer write / read it
 *     (deqsci_ffdnet_head_p32, deqsci_ffdnet_tail_p32).  Ranges (in_amax, in_exp),
 *     (out_amax, out_exp): as for sp16 (B^T d is at most twice max |d|: the headroom against fp16's overflow is 8 x the measured maximum
 *     instead of 16 x).  Output = relu?(conv + bias).  Block tiles of 8 x 64 output pixels, one persistent 512-thread workgroup per CU
 *     (a wave = one output row x 64 couts; 148 KB of LDS). */
int deqsci_conv3x3_c64_wino16(const void* x_p32, const void* u_packed, const float* bias, void* y_p32,
                              int64_t n, int64_t H, int64_t W, int relu, int w_exp, const float* in_amax, int in_exp,
                              const float* out_amax, int out_exp,
                              deqsci_stream_t stream, void* start_event, void* stop_event);
/* A RUN of n_layers such layers in ONE launch: deqsci_conv3x3_c64_split16_stack's contract word for word (layer table, ranges, progress
 *     words, time-out word, residency: never more workgroups than CUs, each of them alone on its CU by its LDS footprint), with
 *     n_tiles = n ceil(H/8) ceil(W/64) and every activation (x, y_even, y_odd) in p32. */
int deqsci_conv3x3_c64_wino16_stack(const void* x_p32, void* y_even, void* y_odd, const void* layers, int n_layers,
                                    int64_t n, int64_t H, int64_t W, const float* ranges, int64_t range_stride, int in_exp, int out_exp,
                                    void* flags, deqsci_stream_t stream, void* start_event, void* stop_event);
/* fp32 channels_last (n,H,W,64) -> sp16 with the range (amax, exp); and, for x = n images of `count` contiguous floats each, max |x| of
 *     image i folded into amax[i] (zero them first): the range of an activation no sp16-writing kernel produced (the denoiser's input
 *     image; a converted fp32 activation). */
int deqsci_f32_to_split16(const float* x_nhwc, void* y_sp16, int64_t n, int64_t H, int64_t W, const float* amax, int exp,
                          deqsci_stream_t stream);
int deqsci_absmax_f32(const float* x, int64_t n, int64_t count, float* amax, deqsci_stream_t stream);
/* SimpleCNN's first layer (conv3x3 1 -> 64 [+ ReLU], deqsci_conv3x3_c1_to_64_f32) writing sp16 with the range (out_amax, out_exp)
 *     instead of fp32 channels_last - no conversion pass in front of a run of split16 layers. */
int deqsci_conv3x3_c1_to_64_sp16(const float* x, const float* w_packed, void* h_sp16, int64_t n, int64_t H, int64_t W, int relu,
                                 const float* out_amax, int out_exp, float* track_amax, deqsci_stream_t stream);
/* ... and writing / reading p32 (in front of / behind a run of deqsci_conv3x3_c64_wino16 layers: SimpleCNN_models.py:43-45, 55-56). */
int deqsci_conv3x3_c1_to_64_p32(const float* x, const float* w_packed, void* h_p32, int64_t n, int64_t H, int64_t W, int relu,
                                const float* out_amax, int out_exp, float* track_amax, deqsci_stream_t stream);
int deqsci_conv3x3_c64_to_1_p32(const void* x_p32, const void* w_packed, float* out, int64_t n, int64_t H, int64_t W,
                                int w_exp, const float* in_amax, int in_exp, deqsci_stream_t stream);
/* The denoisers' last layers (conv3x3 64 -> 4 + pixel shuffle / 64 -> 1, no bias) on the f16 matrix cores with the split-fp16 arithmetic of
 *     deqsci_conv3x3_c64_split16: the 9 taps ride in the matrix N dimension (column = COUT tap + cout), the per-tap products are summed from
 *     LDS.  w_packed: 2^w_exp w as [4 chunks][2 pieces][N tiles][64 lanes][8 halfs]; the input's range is (in_amax, in_exp); fp32 out. */
int deqsci_ffdnet_tail_split16(const void* x_sp16, const void* w_packed, float* out, int64_t n, int64_t H, int64_t W,
                               int w_exp, const float* in_amax, int in_exp, deqsci_stream_t stream);
int deqsci_conv3x3_c64_to_1_split16(const void* x_sp16, const void* w_packed, float* out, int64_t n, int64_t H, int64_t W,
                                    int w_exp, const float* in_amax, int in_exp, deqsci_stream_t stream);
/* FFDNet's first layer (sigma map + pixel-unshuffle + conv3x3 5 -> 64 + ReLU) likewise, writing sp16: K = 45 taps padded to 48, the
 *     activation operand gathered from the image and split on the fly at 2^e_in, e_in(i) from max(in_amax[i], sigma(i)) (in_amax[i] =
 *     max |x| of image i; NULL: in_exp).  w_packed: 2^w_exp w as [3 k steps][2 pieces][2 cout groups][64 lanes][8 halfs], k = 9 ch + tap. */
int deqsci_ffdnet_head_split16(const float* x, const void* w_packed, const float* sigma, int64_t sigma_stride, void* h_sp16,
                               int64_t n, int64_t H, int64_t W, int w_exp, const float* in_amax, int in_exp,
                               const float* out_amax, int out_exp, float* track_amax, deqsci_stream_t stream);

/* FFDNet's first and last layer in front of / behind a run of deqsci_conv3x3_c64_wino16 layers: the entry points above writing / reading
 *     "p32" instead of sp16 (the same planes and ranges, 2^e x unsplit as fp32, the columns of every block of 64 with the parities apart:
 *     see deqsci_conv3x3_c64_wino16).  Same weights (w_packed of deqsci_ffdnet_head_split16 / deqsci_ffdnet_tail_split16), same
 *     arithmetic; the tail splits its operand into hi + lo on the fly.  networks/ffdnet/models.py:46-64, functions.py:16-81. */
int deqsci_ffdnet_head_p32(const float* x, const void* w_packed, const float* sigma, int64_t sigma_stride, void* h_p32,
                           int64_t n, int64_t H, int64_t W, int w_exp, const float* in_amax, int in_exp,
                           const float* out_amax, int out_exp, deqsci_stream_t stream);
int deqsci_ffdnet_tail_p32(const void* x_p32, const void* w_packed, float* out, int64_t n, int64_t H, int64_t W,
                           int w_exp, const float* in_amax, int in_exp, deqsci_stream_t stream);

/* ---- measurement only (bench.py): the same launch with the dispatch's own begin/end timestamps
 * written to two raw hipEvent_t handles (hipExtLaunchKernelGGL), i.e. the duration rocprofv3 reports,
 * without the marker-packet overhead of events recorded around a launch. */
int deqsci_anderson_mix_gap_timed_f32(const float* F_hist, const float* G_hist, const float* alpha,
                                      float beta, int n, int m,
                                      const float* phi, const float* y, const float* phisum,
                                      float* x_out, float* z1,
                                      int64_t bsz, int64_t H, int64_t W, int64_t B,
                                      int layout, int phi_shared, deqsci_stream_t stream,
                                      void* start_event, void* stop_event);
int deqsci_conv3x3_c64_winograd_timed_f32(const float* x, const float* u_packed, const float* bias, float* y,
                                          int64_t n, int64_t H, int64_t W, int relu, deqsci_stream_t stream,
                                          void* start_event, void* stop_event);
int deqsci_conv3x3_c64_winograd44_timed_f32(const float* x, const float* u_packed, const float* bias, float* y,
                                            int64_t n, int64_t H, int64_t W, int relu, deqsci_stream_t stream,
                                            void* start_event, void* stop_event);
int deqsci_event_create(void** ev);
int deqsci_event_destroy(void* ev);
int deqsci_event_elapsed_ms(void* start_event, void* stop_event, float* ms);   /* after the stream is synchronised */

#ifdef __cplusplus
}
#endif
#endif /* DEQSCI_HIP_H */
