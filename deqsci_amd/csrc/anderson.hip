// Anderson-acceleration bookkeeping for the DEQ fixed-point loop on MI355X (gfx950).
// Reference: andersonexp, solvers/new_equilibrium_utils_yaping.py:153-189 (and the Picard
// iterator :213-222, which only needs the two norms).
//
// The reference rebuilds G = F - X (n x N), the full n x n Gram matrix (torch.bmm), solves the
// bordered system (torch.solve) and mixes (two 1 x n bmm's) with ~12 ATen launches and two
// .item() host syncs per iteration.  Here the history is kept as F and G = F - X (X_i = F_i - G_i
// is never stored), only the Gram row of the slot that changed is recomputed, and an iteration is
// three launches with no host sync:
//   K4    residual_store  F_k = z1 - noise, G_k = F_k - X_k written into slot k%m while the block
//                         accumulates <G_k,G_j> and |F_k|^2 (wave64 shuffle tree -> LDS -> one
//                         partial row per block; deterministic, no atomics)
//   K5+K6 anderson_solve  fp64 finish of the partial sums, Gram row/column refresh, (n+1)x(n+1)
//                         LU with partial pivoting (one wavefront per sample), residual norms
//   K7    anderson_mix    X_{k+1} = sum_i alpha_i F_i [- (1-beta) sum_i alpha_i G_i], optionally fused
//                         with the GAP projection of the result (mix_gap) so X_{k+1} is not re-read.
// All streaming parts are HBM-bound (per iteration and sample ~ 4N(2m+3) bytes); the solve is a
// few hundred flops.
#include "common.hpp"
#include <hip/hip_ext.h>

namespace deqsci {

// ------------------------------------------------------------------------------------------------
// K4
// ------------------------------------------------------------------------------------------------
template <int NF, int POL>   // NF = number of filled history slots INCLUDING the one being written
__global__ __launch_bounds__(TB) void residual_store_kernel(const float* __restrict__ z1, const float* __restrict__ noise,
                                                            const float* x_cur, float* __restrict__ F_hist,
                                                            float* __restrict__ G_hist, float* x_next,
                                                            float* __restrict__ partials, int64_t N, int m, int slot,
                                                            int64_t chunk, int vec) {
    const int64_t s = blockIdx.y;
    const int64_t beg = (int64_t)blockIdx.x * chunk;
    const int64_t end = (beg + chunk < N) ? beg + chunk : N;
    const float* zs = z1 + s * N;
    const float* ns = noise ? noise + s * N : nullptr;
    const float* xs = x_cur + s * N;
    float* Fs = F_hist + (s * m + slot) * N;
    float* Gs = G_hist + (s * m + slot) * N;
    const float* Gall = G_hist + (s * m) * N;
    float* xn = x_next ? x_next + s * N : nullptr;

    float acc[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) acc[j] = 0.0f;
    float accf = 0.0f;

    if (vec) {
        // two 1024-element sweeps per trip: 2*(NF+2) independent 16-B loads in flight per lane
        int64_t i = beg + 4 * threadIdx.x;
        for (; i + 4 * TB < end; i += 8 * TB) {
            const int64_t i2 = i + 4 * TB;
            float4 f = ldp<POL>(zs + i), f2 = ldp<POL>(zs + i2);
            if (ns) { f = f - ldp<POL>(ns + i); f2 = f2 - ldp<POL>(ns + i2); }
            const float4 g = f - ldp<POL>(xs + i), g2 = f2 - ldp<POL>(xs + i2);
            float4 o[NF], o2[NF];
#pragma unroll
            for (int j = 0; j < NF; ++j) if (j != slot) { o[j] = ldp<POL>(Gall + j * N + i); o2[j] = ldp<POL>(Gall + j * N + i2); }
            stp<POL>(Fs + i, f); stp<POL>(Fs + i2, f2);
            stp<POL>(Gs + i, g); stp<POL>(Gs + i2, g2);
            if (xn) { stp<POL>(xn + i, f); stp<POL>(xn + i2, f2); }
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                acc[j] = dot4_fma(g, (j == slot) ? g : o[j], acc[j]);
                acc[j] = dot4_fma(g2, (j == slot) ? g2 : o2[j], acc[j]);
            }
            accf = dot4_fma(f2, f2, dot4_fma(f, f, accf));
        }
        for (; i < end; i += 4 * TB) {
            float4 f = ldp<POL>(zs + i);
            if (ns) f = f - ldp<POL>(ns + i);
            const float4 g = f - ldp<POL>(xs + i);
            float4 o[NF];
#pragma unroll
            for (int j = 0; j < NF; ++j) if (j != slot) o[j] = ldp<POL>(Gall + j * N + i);
            stp<POL>(Fs + i, f);
            stp<POL>(Gs + i, g);
            if (xn) stp<POL>(xn + i, f);
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[j] = dot4_fma(g, (j == slot) ? g : o[j], acc[j]);
            accf = dot4_fma(f, f, accf);
        }
    } else {
        for (int64_t i = beg + threadIdx.x; i < end; i += TB) {
            float f = zs[i];
            if (ns) f -= ns[i];
            const float g = f - xs[i];
            Fs[i] = f;
            Gs[i] = g;
            if (xn) xn[i] = f;
#pragma unroll
            for (int j = 0; j < NF; ++j) acc[j] = fmaf(g, (j == slot) ? g : Gall[j * N + i], acc[j]);
            accf = fmaf(f, f, accf);
        }
    }

    __shared__ float red[TB / WAVE][PART_STRIDE];
    const int wave = threadIdx.x / WAVE, lane = threadIdx.x % WAVE;
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const float v = wave_sum(acc[j]);
        if (lane == 0) red[wave][j] = v;
    }
    {
        const float v = wave_sum(accf);
        if (lane == 0) red[wave][MAXM] = v;
    }
    __syncthreads();
    if (threadIdx.x < PART_STRIDE) {
        const int j = threadIdx.x;
        float v = 0.0f;
        if (j < NF || j == MAXM) v = ((red[0][j] + red[1][j]) + red[2][j]) + red[3][j];
        partials[(s * gridDim.x + blockIdx.x) * PART_STRIDE + j] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// K5 + K6
// ------------------------------------------------------------------------------------------------
// Latency matters here, not throughput: the kernel sits alone between two f-calls (19 us of a 690 us iteration at batch 1 in
// its first, lane-0-does-everything form).  So: ONE sweep over the partial rows with every load in flight at once, the
// elimination with one matrix COLUMN per lane in registers (pivot / factors broadcast by v_readlane), and only the
// 20-odd operations of the back substitution left to a single lane.  Every floating-point operation is the one the serial
// form did, in the same order per element, so the results are bit-identical to it.
__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ float readlane_t(float v, int src_lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src_lane)); }
__device__ __forceinline__ double readlane_t(double v, int src_lane) { return readlane_f64(v, src_lane); }

// bordered system  [[0, 1^T], [1, G G^T + lam I]] [nu; alpha] = e0   (:169-172,:178-180), augmented with the right-hand side as
// column nn; lane c holds column c.  S = float: the system is formed and factorised in fp32, as the reference does it (H is an
// fp32 tensor and torch.solve = LAPACK sgesv: LU with partial pivoting) - the Gram entries come from the float64 sums, rounded once.
// S = double: the same elimination in float64 (cond(H) ~ 500: alpha to 1e-13 instead of ~3e-5).
template <typename S>
__device__ __forceinline__ void bordered_solve(const double* Gl, double (*M)[MAXM + 2], float* __restrict__ alpha, int s, int lane, int n, float lam) {
    constexpr int NN = MAXM + 1;
    const int nn = n + 1;
    S col[NN];
#pragma unroll
    for (int i = 0; i < NN; ++i) {
        S v = 0;
        if (i < nn && lane <= nn) {
            if (lane == nn) v = (i == 0) ? S(1) : S(0);
            else if (i == 0 && lane == 0) v = 0;
            else if (i == 0 || lane == 0) v = 1;
            else v = (S)Gl[(i - 1) * MAXM + (lane - 1)] + (i == lane ? (S)lam : S(0));
        }
        col[i] = v;
    }
#pragma unroll
    for (int k = 0; k < NN; ++k) {                          // LU, partial pivoting (as LAPACK gesv)
        if (k < nn) {
            int piv = k;
            S best = col[k] < 0 ? -col[k] : col[k];
#pragma unroll
            for (int i = k + 1; i < NN; ++i)
                if (i < nn) { const S v = col[i] < 0 ? -col[i] : col[i]; if (v > best) { best = v; piv = i; } }
            piv = __builtin_amdgcn_readlane(piv, k);        // column k lives in lane k
#pragma unroll
            for (int i = k + 1; i < NN; ++i)
                if (i == piv) { const S t = col[k]; col[k] = col[i]; col[i] = t; }
            const S inv = S(1) / col[k];                    // (meaningful in lane k)
#pragma unroll
            for (int i = k + 1; i < NN; ++i)
                if (i < nn) {
                    const S f = readlane_t(col[i] * inv, k);
                    col[i] -= f * col[k];
                }
        }
    }
    if (lane <= nn) {
#pragma unroll
        for (int i = 0; i < NN; ++i) M[i][lane] = (double)col[i];
    }
    __syncthreads();
    if (lane == 0) {
        for (int i = nn - 1; i >= 0; --i) {
            S v = (S)M[i][nn];
            for (int j = i + 1; j < nn; ++j) v -= (S)M[i][j] * (S)M[j][nn];
            M[i][nn] = (double)(v / (S)M[i][i]);
        }
        for (int i = 0; i < MAXM; ++i) alpha[(int64_t)s * MAXM + i] = i < n ? (float)M[i + 1][nn] : 0.0f;
    }
}

// ------------------------------------------------------------------------------------------------
// anderson_arith = "reference": the new row of G G^T in the SUMMATION ORDER of the reference's fp32 torch.bmm
// ------------------------------------------------------------------------------------------------
// The reference forms G G^T with one fp32 torch.bmm over the N = H W B elements (solvers/new_equilibrium_utils_yaping.py:177-178); on the
// CPU that produced tests/golden that is MKL's sgemm, and for 5 x N times N x 5 its K loop is SIXTEEN interleaved FMA chains per entry -
// chain c takes k = c, c + 16, c + 32, ... one fused multiply-add after the other - summed pairwise at the end (tools/gram_on_real_history.py:
// an emulation of exactly that lands within one ulp of torch.bmm on every entry, bit-equal on most).  What this order does to the loop's own
// residuals (heavy-tailed: a few moving objects carry the energy) is not noise: in a chain of 2^15 steps most products are smaller than half
// an ulp of the running sum and are ABSORBED, so the diagonal <G_k, G_k> - all terms positive - comes out 3-7e-6 too small, the off-diagonal
// entries, whose small products have either sign, only ~1e-6 off.  That bias, not the size of the error, is what moves the config-2
// ensembles: a flat fp32 chain over 64-element partials (round 5's first form of this kernel: as large an error, no bias)
// sits with the exactly accumulated Gram (profiles/r05_config2_reference_arithmetic_chain64.json).  So the order itself is reproduced: one
// lane per (entry, chain), 2^15 dependent FMAs at N = 2^19 - that dependency chain IS the arithmetic; nothing shortens it without changing
// what it rounds - with the operands staged through LDS two tiles ahead so that the chain waits for nothing else.
typedef int i32x4 __attribute__((ext_vector_type(4)));
constexpr int C16 = 16;                                     // chains per entry
constexpr int C16_TILE = 1024;                              // elements of a row per LDS tile: 64 steps of every chain
constexpr int C16_PITCH = C16_TILE + 16;                    // (the four entries of a wavefront read different rows: 16 banks apart)
constexpr int C16_ROWS = 5;                                 // the new row + up to four others per workgroup
constexpr int C16_RING = 7;                                 // tiles in LDS (7 x 5 x 4160 B = 142 KiB)
constexpr int C16_AHEAD = 5;                                // a tile is asked for five chain-tiles (~2 us) before it is read: HBM latency
constexpr int C16_LOADERS = 3;                              // wavefronts 1..3 only move data (LDS-DMA, 20 KiB per tile); wavefront 0 only adds
constexpr int C16_TB = WAVE * (1 + C16_LOADERS);
// the caller's state of this arithmetic, per sample, in 4-byte words: the persistent fp32 Gram, the chain sums of this call, and the records of the
// two-pass form (gram_round_kernel -> gram_chain_apply_kernel): per K4 block the binade the block was rounded for, and per (block, entry, chain,
// candidate binade) the pair {sum of the rounded terms, sum of their magnitudes} in units of that binade's ulp
constexpr int REF_CSUM = MAXM * MAXM;
constexpr int REF_WALKED = REF_CSUM + MAXM * C16;            // (diagnostic) blocks each chain was walked through in the last call
constexpr int REF_TAKEN = REF_WALKED + MAXM * C16;           // per entry: term slots handed out in this call (zeroed again by gram_chain_apply_kernel)
constexpr int REF_CALL = REF_TAKEN + MAXM;                  // the number of reference-Gram calls finished on this state (gram_chain_apply_kernel counts): the tag of the
                                                            // granules the blocks of residual_store_round_kernel publish to one another
constexpr int REF_EP = REF_CALL + 2;
constexpr int REF_NONE = -100000;                           // no prediction for this block (first block, zero or non-finite sums)
__host__ __device__ constexpr int ref_lo(int code) { return (code >> 1) - 1 + (code & 1); }     // the lower of the two candidate binades
constexpr int REF_CAND = 2;                                 // binades a block is rounded for: the predicted one and its nearer neighbour
constexpr int PF_TERMS = 128;                               // terms per chain of a block of 2048 elements (the chunk K4 uses up to N = 2^25 / bsz)
constexpr int TSLOTS = 96;                                  // blocks per entry whose terms gram_round_kernel also stores chain by chain (see there)
__host__ __device__ constexpr int64_t ref_rec(int64_t nchunks) { return REF_EP + nchunks * MAXM; }
__host__ __device__ constexpr int64_t ref_slot(int64_t nchunks) { return ref_rec(nchunks) + nchunks * MAXM * C16 * REF_CAND * 2; }
// (round 6) per K4 block and entry one 8-byte granule {block sum of K4, call number}: how the blocks of the fused K4 + rounding launch tell
// one another where the chains stand (residual_store_round_kernel)
__host__ __device__ constexpr int64_t ref_pub(int64_t nchunks) { return (ref_slot(nchunks) + nchunks * MAXM + 3) / 4 * 4; }
__host__ __device__ constexpr int64_t ref_terms(int64_t nchunks) { return (ref_pub(nchunks) + nchunks * MAXM * 2 + 3) / 4 * 4; }
__host__ __device__ constexpr int64_t ref_words(int64_t nchunks) { return ref_terms(nchunks) + (int64_t)MAXM * TSLOTS * C16 * PF_TERMS * 2; }

__global__ __launch_bounds__(C16_TB) void gram_row_chain16_kernel(const float* __restrict__ G_hist, float* __restrict__ ref_state, int64_t ref_stride,
                                                                  int64_t N, int m, int slot, int n_filled, int vec) {
    const int64_t s = blockIdx.y;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const int lane = threadIdx.x % WAVE, c = lane & (C16 - 1), jj = lane >> 4;
    const int j0 = 4 * blockIdx.x, j = j0 + jj;
    const bool live = j < n_filled;
    const float* Gs = G_hist + s * m * N;
    const float* ga = Gs + (int64_t)slot * N;
    float acc = 0.0f;
    if (vec) {
        __shared__ __attribute__((aligned(16))) float ring[C16_RING][C16_ROWS][C16_PITCH];
        const int tiles = (int)((N + C16_TILE - 1) / C16_TILE);
        // one buffer descriptor per staged row: reads past the row's N floats return 0, and fma(0, 0, x) = x
        i32x4 rsrc[C16_ROWS];
#pragma unroll
        for (int r = 0; r < C16_ROWS; ++r) {
            const int jr = (r == 0) ? slot : ((j0 + r - 1 < n_filled) ? j0 + r - 1 : slot);
            const uint64_t base = (uint64_t)(Gs + (int64_t)jr * N);
            rsrc[r].x = (int)__builtin_amdgcn_readfirstlane((uint32_t)base);
            rsrc[r].y = (int)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
            rsrc[r].z = (int)__builtin_amdgcn_readfirstlane((uint32_t)(N * 4));
            rsrc[r].w = 0x00020000;
        }
        const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)&ring[0][0][0];
        auto ask = [&](int t) __attribute__((always_inline)) {   // tile t -> ring slot t % C16_RING: 20 asynchronous 1 KiB copies, no registers
            const uint32_t slot_lds = ring_lds + (uint32_t)((t % C16_RING) * C16_ROWS * C16_PITCH * 4);
            const uint32_t voff = (uint32_t)lane * 16u;
#pragma unroll
            for (int r = 0; r < C16_ROWS; ++r)
#pragma unroll
                for (int h = 0; h < C16_TILE / 256; ++h) {
                    const uint32_t m0v = __builtin_amdgcn_readfirstlane(slot_lds + (uint32_t)((r * C16_PITCH + 256 * h) * 4));
                    const uint32_t soff = __builtin_amdgcn_readfirstlane((uint32_t)t * (uint32_t)(C16_TILE * 4) + (uint32_t)(h * 1024));
                    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(m0v), "v"(voff), "s"(rsrc[r]), "s"(soff) : "m0", "memory");
                }
        };
        if (wave > 0)
            for (int t = wave - 1; t < C16_AHEAD && t < tiles; t += C16_LOADERS) ask(t);
        for (int t = 0; t < tiles; ++t) {
            if (wave > 0) {
                if (t % C16_LOADERS == wave - 1) {              // my tile: landed?  (behind it I have asked for tile t + 3 at most)
                    if (t + C16_LOADERS < tiles) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if ((t + C16_AHEAD) % C16_LOADERS == wave - 1 && t + C16_AHEAD < tiles) ask(t + C16_AHEAD);   // (into the slot of tile t - 2: read and done)
            }
            __syncthreads();
            if (wave == 0) {
                const float* a = &ring[t % C16_RING][0][c];
                const float* b = &ring[t % C16_RING][1 + jj][c];
                // 64 dependent FMAs; their 128 operands come 12 steps ahead (an LDS read takes ~64 cycles, a dependent FMA ~5)
                float av[C16_TILE / C16], bv[C16_TILE / C16];
#pragma unroll
                for (int i = 0; i < C16_TILE / C16; ++i) { av[i] = a[C16 * i]; bv[i] = b[C16 * i]; }
#pragma unroll
                for (int i = 0; i < C16_TILE / C16; ++i) acc = fmaf(av[i], bv[i], acc);
                __builtin_amdgcn_sched_group_barrier(0x100, 14, 0);
#pragma unroll
                for (int i = 0; i < C16_TILE / C16 - 14; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
        if (wave > 0) return;
    } else {
        if (wave > 0) return;
        const float* gb = Gs + (int64_t)(live ? j : slot) * N;
        for (int64_t k = c; k < N; k += C16) acc = fmaf(ga[k], gb[k], acc);
    }
    if (live) {
        ref_state[s * ref_stride + REF_CSUM + j * C16 + c] = acc;       // (folded by anderson_solve_kernel)
        reinterpret_cast<int*>(ref_state + s * ref_stride)[REF_WALKED + j * C16 + c] = -1;
    }
}

// ------------------------------------------------------------------------------------------------
// The same sums, bit for bit, WITHOUT the 2^15-step dependency chain
// ------------------------------------------------------------------------------------------------
// While the running sum S of a chain stays inside one binade [2^e, 2^(e+1)) it is a multiple of u = 2^(e-23), and then
//     RN(S + p) = S + RN_u(p)                    (RN_u: to the nearest multiple of u; no tie)
// - the rounded step does not depend on S.  Inside a binade the chain is an INTEGER sum (in units of u) of individually rounded products, exact
// in any order, so every K4 block can round and add its 128 terms per chain on its own (gram_round_kernel, the whole machine, one pass over the
// history) - for the binade the chain is predicted to be in there (from K4's block partials: prefix / 16) and its two neighbours -, and only
// the bookkeeping stays serial (gram_chain_apply_kernel, one wavefront per chain): take the blocks 64 at a time, accept the run of blocks whose
// record is for the binade S is in and that cannot leave it (|S| -+ the block's sum of magnitudes stays inside, two ulps to spare), and walk
// through a block that fails - a crossing into the next binade, a term on a rounding tie (then S's parity decides), a term too large for the
// trick, the first blocks where S is still small - term by term with the FMA itself.  tools/gram_chain_prototype.py: ~16 of 256 blocks per
// chain are walked on the loop's own residuals; tests/test_gpu_parity.py holds the two forms bit-equal on those and on random data.
constexpr int RND_TILE = 1024;                              // elements per staged tile: 64 terms of every chain
constexpr int RND_PITCH = RND_TILE + RND_TILE / C16;        // (skewed by one word per 16: a chain's terms fall in different banks)

template <int NF>
__global__ __launch_bounds__(TB) void gram_round_kernel(const float* __restrict__ G_hist, const float* __restrict__ partials, float* __restrict__ ref_state,
                                                        int64_t ref_stride, int64_t N, int m, int slot, int64_t chunk) {
    const int64_t s = blockIdx.y;
    const int b = blockIdx.x, nchunks = gridDim.x;
    const int t = threadIdx.x, wave = t / WAVE, lane = t % WAVE;
    const int64_t beg = (int64_t)b * chunk;
    const float* Gs = G_hist + s * m * N;
    __shared__ float red[TB / WAVE][MAXM];
    __shared__ int ep_s[MAXM], slot_s[MAXM];
    __shared__ __attribute__((aligned(16))) float tile[NF][RND_PITCH];
    float4 pre[NF];
    auto fetch = [&](int tl) {
        const int64_t k = beg + (int64_t)tl * RND_TILE + 4 * t;
        const bool in = k < N;                                  // (N % 4 == 0 on this path)
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const float4 v = ld4(Gs + (int64_t)j * N + (in ? k : 0));
            pre[j] = make_float4(in ? v.x : 0.f, in ? v.y : 0.f, in ? v.z : 0.f, in ? v.w : 0.f);
        }
    };
    fetch(0);                                                   // (the first tile is on its way while the prediction below is formed)
    // ---- where each chain of each entry is when it gets here: (K4's sums of the blocks before this one) / 16
    const float* ps = partials + s * nchunks * PART_STRIDE;
    {
        float x[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) x[j] = 0.0f;
        for (int bb = t; bb < b; bb += TB)
#pragma unroll
            for (int j = 0; j < NF; ++j) x[j] += ps[(int64_t)bb * PART_STRIDE + j];
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const float v = wave_sum(x[j]);
            if (lane == 0) red[wave][j] = v;
        }
        __syncthreads();
        if (t < NF) {
            float v = 0.0f;
            for (int w = 0; w < TB / WAVE; ++w) v += red[w][t];
            v = fabsf(v) * (1.0f / C16);
            int code = REF_NONE;                                // 2 x (predicted binade) + (upper half of it, on the log scale: the neighbour above is the second candidate)
            if (v > 0.0f && v < 3.0e38f) {
                const int e = ilogbf(v);
                if (e >= -90 && e <= 90) code = 2 * e + (ldexpf(v, -e) >= 1.41421356f ? 1 : 0);
            }
            ep_s[t] = code;
            int* sti = reinterpret_cast<int*>(ref_state + s * ref_stride);
            sti[REF_EP + (int64_t)b * MAXM + t] = code;
            // a block that gram_chain_apply_kernel is likely to WALK - the first ones, where the block is a large part of the sum so far, where
            // the chains (each within ~10 % of their mean) change binade - also leaves its terms chain by chain: a chain's 128 terms of a block
            // are 64 bytes apart in the history (8 KB of cache lines per row), 1 KB in a row there
            int sl = -1;
            if (chunk <= (int64_t)PF_TERMS * C16) {
                const float here = fabsf(ps[(int64_t)b * PART_STRIDE + t]) * (1.0f / C16);
                const float lo = v < v + here ? v : v + here, hi = v + here;
                bool mark = code == REF_NONE || b < 24 || !(here < 0.2f * v);
                if (!mark) mark = ilogbf(0.88f * lo) != ilogbf(1.12f * hi);
                if (mark) {
                    sl = atomicAdd(&sti[REF_TAKEN + t], 1);
                    if (sl >= TSLOTS) sl = -1;
                }
            }
            slot_s[t] = sl;
            sti[ref_slot(nchunks) + (int64_t)b * MAXM + t] = sl;
        }
        __syncthreads();
    }
    float cM[NF][REF_CAND], cH[NF][REF_CAND];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int q = 0; q < REF_CAND; ++q) {
            const int e = ep_s[j] == REF_NONE ? 0 : ref_lo(ep_s[j]) + q;
            cM[j][q] = ldexpf(1.5f, e);                         // S + p rounds on the grid of [2^e, 2^(e+1)) <=> 1.5 2^e + p does (|p| < 2^(e-1))
            cH[j][q] = ldexpf(1.0f, e - 24);                    // u / 2: the distance of a tie
        }
    // per (entry, candidate): the sum of the bit patterns of t = RN(1.5 2^e + a b) - inside the binade they are linear in t, so that
    // bits(t) - bits(1.5 2^e) IS the rounded term in ulps -, the sum of the magnitudes, their OR (any term out of the binade shows up as >= 2^22)
    unsigned Ns[NF][REF_CAND], As[NF][REF_CAND], Os[NF][REF_CAND];
    bool tie[NF][REF_CAND];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int q = 0; q < REF_CAND; ++q) { Ns[j][q] = 0; As[j][q] = 0; Os[j][q] = 0; tie[j][q] = ep_s[j] == REF_NONE; }
    const int c = t >> 4, g = t & 15;                          // my chain, and which of its terms: i = g, g + 16, ...
    float* terms = ref_state + s * ref_stride + ref_terms(nchunks);
    const int tiles = (int)(chunk / RND_TILE);
    for (int tl = 0; tl < tiles; ++tl) {
        const int w0 = 4 * t + (t >> 2);
#pragma unroll
        for (int j = 0; j < NF; ++j) { tile[j][w0] = pre[j].x; tile[j][w0 + 1] = pre[j].y; tile[j][w0 + 2] = pre[j].z; tile[j][w0 + 3] = pre[j].w; }
        __syncthreads();
        if (tl + 1 < tiles) fetch(tl + 1);
#pragma unroll
        for (int ii = 0; ii < RND_TILE / C16 / 16; ++ii) {
            const int i = g + 16 * ii, w = 17 * i + c;
            const float a = tile[slot][w];
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const float bb = tile[j][w];
                if (slot_s[j] >= 0)                             // (uniform)
                    *reinterpret_cast<float2*>(terms + ((((int64_t)j * TSLOTS + slot_s[j]) * C16 + c) * PF_TERMS + (tl * (RND_TILE / C16) + i)) * 2) = make_float2(a, bb);
#pragma unroll
                for (int q = 0; q < REF_CAND; ++q) {
                    const float tt = fmaf(a, bb, cM[j][q]);
                    const unsigned ti = __float_as_uint(tt), mi = __float_as_uint(cM[j][q]);
                    const unsigned dl = ti > mi ? ti - mi : mi - ti;        // |RN_u(a b)| / u
                    Ns[j][q] += ti - mi;
                    As[j][q] += dl;
                    Os[j][q] |= dl;
                    const float d = fmaf(a, bb, -(tt - cM[j][q]));          // a b - RN_u(a b): +- u / 2 on a tie (then S's parity decides: walked)
                    tie[j][q] |= fabsf(d) == cH[j][q];
                }
            }
        }
        __syncthreads();
    }
    int* rec = reinterpret_cast<int*>(ref_state + s * ref_stride) + ref_rec(nchunks);
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int q = 0; q < REF_CAND; ++q) {
            unsigned nn = Ns[j][q], aa = As[j][q], oo = Os[j][q] | (tie[j][q] ? 0x80000000u : 0u);
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { nn += __shfl_xor(nn, o, WAVE); aa += __shfl_xor(aa, o, WAVE); oo |= __shfl_xor(oo, o, WAVE); }
            if (g == 0) {
                int* r2 = rec + ((((int64_t)b * MAXM + j) * C16 + c) * REF_CAND + q) * 2;
                r2[0] = (int)nn;
                r2[1] = (oo >> 22) ? -1 : (int)aa;              // (a term of 2^22 ulps or more, a tie: not for the fast path)
            }
        }
}

// ------------------------------------------------------------------------------------------------
// K4 AND the rounding pass in one launch (round 6)
// ------------------------------------------------------------------------------------------------
// gram_round_kernel re-reads the NF history rows K4 has just had in its registers, because the binade a block rounds for comes from K4's sums of
// the blocks BEFORE it.  Here K4's blocks tell one another: a block publishes its NF dot products as 8-byte granules {sum, call number} (one
// agent-scope store each: value and tag arrive together; the call number is a word of the state that gram_chain_apply_kernel advances when it is
// done with the call - nothing to reset, a replayed hipGraph counts on), then reads its predecessors' granules (agent-scope 16-byte loads, two
// granules each, every granule checked by its own tag and re-asked until the tag is this call's).  The wait is BOUNDED and nothing depends on
// it but speed: the prediction only decides how many blocks gram_chain_apply_kernel walks (a record for the wrong binade is never used), so a
// block whose predecessors have not reported in time - they run beside it or have finished whenever workgroups are dispatched in index order,
// which HIP does not promise - rounds for what it has.  (A ticket drawn from an atomic counter would order the blocks by arrival whatever the
// dispatch order; it cost 15 us of a 66 us launch - the block's loads wait for it - and was dropped: profiles/r06_gram_fused_ablations.txt.)
// Then the block rounds its own 2048 elements from registers through the LDS tile exactly as gram_round_kernel does.  Same partials, same
// F / G / x_next, records of the same meaning: the launch replaces residual_store_kernel + gram_round_kernel for chunks of 2048 elements (N bsz <= 2^25)
// when N is a whole number of at most 256 of them (deqsci_gram_ref_fusable).
constexpr unsigned LOOKBACK_SPINS = 1u << 11;                // re-asks per granule (~0.1 us each) before a block stops waiting: ~0.2 ms
#ifndef RSR_ABL
#define RSR_ABL 0     // timing ablations only (tools/gram_fused_time.py; results wrong): 1 = no term stores, 2 = no waiting in the look-back, 4 = no rounding
                      // arithmetic, 16 = no look-back at all
#endif

template <int NF, int POL>
__global__ __launch_bounds__(TB, 4) void residual_store_round_kernel(const float* __restrict__ z1, const float* __restrict__ noise, const float* x_cur,
                                                                  float* __restrict__ F_hist, float* __restrict__ G_hist, float* x_next,
                                                                  float* __restrict__ partials, float* __restrict__ ref_state, int64_t ref_stride,
                                                                  int64_t N, int m, int slot) {
    constexpr int64_t chunk = 2 * RND_TILE;
    const int64_t s = blockIdx.y;
    const int nchunks = gridDim.x;
    const int t = threadIdx.x, wave = t / WAVE, lane = t % WAVE;
    float* st = ref_state + s * ref_stride;
    int* sti = reinterpret_cast<int*>(st);
    __shared__ float red[TB / WAVE][PART_STRIDE];
    __shared__ float redx[TB / WAVE][MAXM];
    __shared__ int ep_s[MAXM], slot_s[MAXM];
    __shared__ __attribute__((aligned(16))) float tile[NF][RND_PITCH];
    const int b = blockIdx.x;
    // (asked for first, needed last: behind the block's own pass)
    const unsigned epoch = __hip_atomic_load(reinterpret_cast<const unsigned*>(sti + REF_CALL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;   // (a zeroed state carries tag 0: nothing published)
    const int64_t beg = (int64_t)b * chunk;
    const float* zs = z1 + s * N;
    const float* ns = noise ? noise + s * N : nullptr;
    const float* xs = x_cur + s * N;
    float* Fs = F_hist + (s * m + slot) * N;
    float* Gs = G_hist + (s * m + slot) * N;
    const float* Gall = G_hist + (s * m) * N;
    float* xn = x_next ? x_next + s * N : nullptr;
    // ---- K4 (residual_store_kernel's arithmetic, operation for operation: the block sums are bit-identical to its)
    const int64_t i = beg + 4 * t, i2 = i + 4 * TB;
    float4 f = ldp<POL>(zs + i), f2 = ldp<POL>(zs + i2);
    if (ns) { f = f - ldp<POL>(ns + i); f2 = f2 - ldp<POL>(ns + i2); }
    const float4 g = f - ldp<POL>(xs + i), g2 = f2 - ldp<POL>(xs + i2);
    float4 o[NF], o2[NF];
#pragma unroll
    // (the residual history with the DEFAULT cache policy whatever POL says: gram_chain_apply_kernel gathers from these rows next - the terms of the
    //  blocks it walks without a term slot - and finds them in the Infinity Cache: 110.6 -> 105.9 us for K4 + apply + solve at eight measurements)
    for (int j = 0; j < NF; ++j) if (j != slot) { o[j] = ld4(Gall + j * N + i); o2[j] = ld4(Gall + j * N + i2); }
    stp<POL>(Fs + i, f); stp<POL>(Fs + i2, f2);
    st4(Gs + i, g); st4(Gs + i2, g2);
    if (xn) { stp<POL>(xn + i, f); stp<POL>(xn + i2, f2); }
    float acc[NF];
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        if (j == slot) { o[j] = g; o2[j] = g2; }
        acc[j] = dot4_fma(g, o[j], 0.0f);
        acc[j] = dot4_fma(g2, o2[j], acc[j]);
    }
    const float accf = dot4_fma(f2, f2, dot4_fma(f, f, 0.0f));
#pragma unroll
    for (int j = 0; j < NF; ++j) {
        const float v = wave_sum(acc[j]);
        if (lane == 0) red[wave][j] = v;
    }
    {
        const float v = wave_sum(accf);
        if (lane == 0) red[wave][MAXM] = v;
    }
    __syncthreads();
    unsigned long long* pub = reinterpret_cast<unsigned long long*>(st + ref_pub(nchunks));
    float mine = 0.0f;                                          // (t < NF: K4's sum of this block for entry t)
    if (t < PART_STRIDE) {
        const int j = t;
        float v = 0.0f;
        if (j < NF || j == MAXM) v = ((red[0][j] + red[1][j]) + red[2][j]) + red[3][j];
        partials[(s * nchunks + b) * PART_STRIDE + j] = v;
        mine = v;
        if (j < NF)
            __hip_atomic_store(pub + (int64_t)b * MAXM + j, ((unsigned long long)epoch << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- where each chain of each entry is when it gets here: the sums of the blocks before this one (their granules) / 16
    {
        float x[NF];
#pragma unroll
        for (int j = 0; j < NF; ++j) x[j] = 0.0f;
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        constexpr int NP = (NF + 1) / 2;                        // 16-byte pieces of a block's granules: {sum, tag, sum, tag}
        for (int bb = t; bb < ((RSR_ABL & 16) ? 0 : b); bb += TB) {
            const u32x4* src = reinterpret_cast<const u32x4*>(pub + (int64_t)bb * MAXM);
            u32x4 gk[NP];
#pragma unroll
            for (int p = 0; p < NP; ++p) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(gk[p]) : "v"(src + p) : "memory");
#pragma unroll
            for (int p = 0; p < NP; ++p) {
                asm volatile("s_waitcnt vmcnt(%1)" : "+v"(gk[p]) : "n"(NP - 1 - p) : "memory");
                unsigned spins = 0;
                while (!(RSR_ABL & 2) && (gk[p].y != epoch || (2 * p + 1 < NF && gk[p].w != epoch)) && ++spins < LOOKBACK_SPINS) {
                    __builtin_amdgcn_s_sleep(2);
                    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(gk[p]) : "v"(src + p) : "memory");
                }
                x[2 * p] += __uint_as_float(gk[p].x);
                if (2 * p + 1 < NF) x[2 * p + 1] += __uint_as_float(gk[p].z);
            }
        }
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const float v = wave_sum(x[j]);
            if (lane == 0) redx[wave][j] = v;
        }
        __syncthreads();
        if (t < NF) {
            float v = 0.0f;
            for (int w = 0; w < TB / WAVE; ++w) v += redx[w][t];
            v = fabsf(v) * (1.0f / C16);
            int code = REF_NONE;
            if (v > 0.0f && v < 3.0e38f) {
                const int e = ilogbf(v);
                if (e >= -90 && e <= 90) code = 2 * e + (ldexpf(v, -e) >= 1.41421356f ? 1 : 0);
            }
            ep_s[t] = code;
            sti[REF_EP + (int64_t)b * MAXM + t] = code;
            int sl = -1;                                        // (the blocks likely to be WALKED also leave their terms chain by chain: gram_round_kernel)
            {
                const float here = fabsf(mine) * (1.0f / C16);
                const float lo = v < v + here ? v : v + here, hi = v + here;
                bool mark = code == REF_NONE || b < 24 || !(here < 0.2f * v);
                if (!mark) mark = ilogbf(0.88f * lo) != ilogbf(1.12f * hi);
                if (mark) {
                    sl = atomicAdd(&sti[REF_TAKEN + t], 1);
                    if (sl >= TSLOTS) sl = -1;
                }
            }
            slot_s[t] = sl;
            sti[ref_slot(nchunks) + (int64_t)b * MAXM + t] = sl;
        }
        __syncthreads();
    }
    // ---- the rounding pass of gram_round_kernel on the block's own two tiles, from registers
    float cM[NF][REF_CAND], cH[NF][REF_CAND];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int q = 0; q < REF_CAND; ++q) {
            const int e = ep_s[j] == REF_NONE ? 0 : ref_lo(ep_s[j]) + q;
            cM[j][q] = ldexpf(1.5f, e);
            cH[j][q] = ldexpf(1.0f, e - 24);
        }
    unsigned Ns[NF][REF_CAND], As[NF][REF_CAND], Os[NF][REF_CAND];
    bool tie[NF][REF_CAND];
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int q = 0; q < REF_CAND; ++q) { Ns[j][q] = 0; As[j][q] = 0; Os[j][q] = 0; tie[j][q] = ep_s[j] == REF_NONE; }
    const int c = t >> 4, gq = t & 15;                         // my chain, and which of its terms: i = gq, gq + 16, ...
    float* terms = st + ref_terms(nchunks);
#pragma unroll
    for (int tl = 0; tl < 2; ++tl) {
        const int w0 = 4 * t + (t >> 2);
#pragma unroll
        for (int j = 0; j < NF; ++j) {
            const float4 v = tl ? o2[j] : o[j];
            tile[j][w0] = v.x; tile[j][w0 + 1] = v.y; tile[j][w0 + 2] = v.z; tile[j][w0 + 3] = v.w;
        }
        __syncthreads();
#pragma unroll
        for (int ii = 0; ii < RND_TILE / C16 / 16; ++ii) {
            const int ix = gq + 16 * ii, w = 17 * ix + c;
            const float a = tile[slot][w];
#pragma unroll
            for (int j = 0; j < NF; ++j) {
                const float bb = tile[j][w];
                if (slot_s[j] >= 0 && !(RSR_ABL & 1))           // (uniform)
                    *reinterpret_cast<float2*>(terms + ((((int64_t)j * TSLOTS + slot_s[j]) * C16 + c) * PF_TERMS + (tl * (RND_TILE / C16) + ix)) * 2) = make_float2(a, bb);
#pragma unroll
                for (int q = 0; q < ((RSR_ABL & 4) ? 0 : REF_CAND); ++q) {
                    const float tt = fmaf(a, bb, cM[j][q]);
                    const unsigned ti = __float_as_uint(tt), mi = __float_as_uint(cM[j][q]);
                    const unsigned dl = ti > mi ? ti - mi : mi - ti;
                    Ns[j][q] += ti - mi;
                    As[j][q] += dl;
                    Os[j][q] |= dl;
                    const float d = fmaf(a, bb, -(tt - cM[j][q]));
                    tie[j][q] |= fabsf(d) == cH[j][q];
                }
            }
        }
        __syncthreads();
    }
    int* rec = sti + ref_rec(nchunks);
#pragma unroll
    for (int j = 0; j < NF; ++j)
#pragma unroll
        for (int q = 0; q < REF_CAND; ++q) {
            unsigned nn = Ns[j][q], aa = As[j][q], oo = Os[j][q] | (tie[j][q] ? 0x80000000u : 0u);
#pragma unroll
            for (int of = 8; of > 0; of >>= 1) { nn += __shfl_xor(nn, of, WAVE); aa += __shfl_xor(aa, of, WAVE); oo |= __shfl_xor(oo, of, WAVE); }
            if (gq == 0) {
                int* r2 = rec + ((((int64_t)b * MAXM + j) * C16 + c) * REF_CAND + q) * 2;
                r2[0] = (int)nn;
                r2[1] = (oo >> 22) ? -1 : (int)aa;
            }
        }
}

// inclusive prefix sum over the 64 lanes: shifts inside the rows of 16 (DPP), then the three row totals
__device__ __forceinline__ int wave_scan_incl(int v, int lane) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, false);    // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, false);    // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, false);    // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, false);    // row_shr:8
    const int t0 = __builtin_amdgcn_readlane(v, 15), t1 = __builtin_amdgcn_readlane(v, 31), t2 = __builtin_amdgcn_readlane(v, 47);
    const int row = lane >> 4;
    return v + (row >= 1 ? t0 : 0) + (row >= 2 ? t1 : 0) + (row >= 3 ? t2 : 0);
}

constexpr int PF_BLOCKS = 32;                               // blocks whose terms are fetched before the chain starts
constexpr int WIN = 256;                                    // blocks whose records are held in LDS at a time

__global__ __launch_bounds__(WAVE) void gram_chain_apply_kernel(const float* __restrict__ G_hist, const float* __restrict__ partials, float* __restrict__ ref_state,
                                                                int64_t ref_stride, int64_t N, int m, int slot, int64_t chunk, int nchunks) {
    const int64_t s = blockIdx.y;
    const int j = blockIdx.x / C16, c = blockIdx.x % C16, lane = threadIdx.x;
    const float* ga = G_hist + (s * m + slot) * N;
    const float* gb = G_hist + (s * m + j) * N;
    float* st = ref_state + s * ref_stride;
    const int* ep = reinterpret_cast<const int*>(st + REF_EP);
    const int* rec = reinterpret_cast<const int*>(st) + ref_rec(nchunks);
    const int* tslot = reinterpret_cast<const int*>(st) + ref_slot(nchunks);
    const float* terms = st + ref_terms(nchunks);
    const bool small = chunk <= (int64_t)PF_TERMS * C16;        // blocks of <= 128 terms per chain: the prefetches below apply
    __shared__ __attribute__((aligned(16))) float wa[1024], wb[1024];                        // the terms of a block that is walked (chunk <= 16384: <= 1024 per chain)
    __shared__ __attribute__((aligned(16))) float pfa[PF_BLOCKS][PF_TERMS], pfb[PF_BLOCKS][PF_TERMS];
    __shared__ int pf_list[PF_BLOCKS];
    __shared__ int code_s[WIN], tslot_s[WIN];
    __shared__ int rec_s[WIN][REF_CAND][2];
    int win0 = -1;                                              // first block of the record window in LDS
    auto load_window = [&](int b0) {                            // (uniform)
        __syncthreads();
        for (int x = lane; x < WIN; x += WAVE) {
            const int bl = b0 + x;
            const bool in = bl < nchunks;
            code_s[x] = in ? ep[(int64_t)bl * MAXM + j] : REF_NONE;
            tslot_s[x] = in ? tslot[(int64_t)bl * MAXM + j] : -1;
            const int4 r4 = in ? *reinterpret_cast<const int4*>(rec + (((int64_t)bl * MAXM + j) * C16 + c) * REF_CAND * 2) : make_int4(0, -1, 0, -1);
            rec_s[x][0][0] = r4.x; rec_s[x][0][1] = r4.y; rec_s[x][1][0] = r4.z; rec_s[x][1][1] = r4.w;
        }
        win0 = b0;
        __syncthreads();
    };
#ifdef DEQSCI_DIAG
    long long stamp0 = __builtin_readcyclecounter(), c_group = 0, c_walk = 0;
    long long stampA = 0, stampB = 0;
    int n_group = 0, n_hit = 0, n_noslot = 0, n_miss = 0;
#endif
    // (K4's sums of this lane's four blocks - the fallback of the prediction below - are asked for together with the record window)
    float avg4[WIN / WAVE];
#pragma unroll
    for (int r = 0; r < WIN / WAVE; ++r) {
        const int bl = (WIN / WAVE) * lane + r;
        avg4[r] = bl < nchunks ? partials[(s * nchunks + bl) * PART_STRIDE + j] * (1.0f / C16) : 0.0f;
    }
    load_window(0);
#ifdef DEQSCI_DIAG
    const long long stamp1 = __builtin_readcyclecounter();
#endif
    // ---- the blocks that will be walked can be told beforehand: THIS chain's way through the binades, to a few ulps, is the running sum of its
    // own block records (K4's block sums / 16 where a record is flagged) - the blocks where that sum changes binade or comes within the block's
    // magnitude of a boundary, the flagged ones, the first ones.  Their terms are asked for before the chain starts, all at once (a walk that has
    // to fetch its own terms waits a round trip to memory, ~2 us).  (All of the loads in flight before the first is parked: one round trip in all.)
    int n_pf = 0;
    auto ask_terms = [&](int bl, float (&da)[PF_TERMS / WAVE], float (&db)[PF_TERMS / WAVE]) {
        const int64_t beg = (int64_t)bl * chunk;
#pragma unroll
        for (int h = 0; h < PF_TERMS / WAVE; ++h) {
            const int i = lane + WAVE * h;
            const int64_t k = beg + c + (int64_t)C16 * i;
            const bool in = k < N && (int64_t)C16 * i + c < chunk;
            const float va = ga[in ? k : 0], vb = gb[in ? k : 0];
            da[h] = in ? va : 0.0f;                             // (past the block: fma(0, 0, S) = S)
            db[h] = in ? vb : 0.0f;
        }
    };
    if (small && nchunks <= WIN) {
        constexpr int PER = WIN / WAVE;                         // blocks per lane: 4 lane + r
        float contrib[PER], mag[PER];
        float mine = 0.0f;
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const int bl = PER * lane + r;
            contrib[r] = 0.0f;
            mag[r] = 0.0f;
            if (bl < nchunks) {
                const int code = code_s[bl];
                const int q = code == REF_NONE ? 0 : (code >> 1) - ref_lo(code);
                const int aa = rec_s[bl][q][1];
                const float avg = avg4[r];
                if (code != REF_NONE && aa >= 0) {
                    const float u = ldexpf(1.0f, (code >> 1) - 23);
                    contrib[r] = (float)rec_s[bl][q][0] * u;
                    mag[r] = (float)aa * u;
                } else {
                    contrib[r] = avg;
                    mag[r] = 3.0e38f;                           // (flagged: walked whatever the sum does)
                }
            }
            mine += contrib[r];
        }
        float run = mine;                                       // exclusive prefix of the lanes' sums (a prediction: float order does not matter)
#pragma unroll
        for (int o = 1; o < WAVE; o <<= 1) {
            const float v = __shfl_up(run, o, WAVE);
            if (lane >= o) run += v;
        }
        run -= mine;
#pragma unroll
        for (int r = 0; r < PER; ++r) {
            const int bl = PER * lane + r;
            const float s0 = run, s1 = run + contrib[r];
            run = s1;
            bool mark = false;
            if (bl < nchunks) {
                const float lo = fminf(fabsf(s0), fabsf(s1)), hi = fmaxf(fabsf(s0), fabsf(s1));
                const float m2 = mag[r] < 1.0e38f ? 1.5f * mag[r] + 1.0e-3f * hi : 3.0e38f;
                mark = !(lo - m2 > 0.0f) || (s0 < 0.0f) != (s1 < 0.0f) || ilogbf(lo - m2) != ilogbf(hi + m2);
                if (!mark) {                                    // is there a record for the binade the chain will be in?
                    const int code = code_s[bl];
                    const int q = ilogbf(lo) - ref_lo(code);
                    mark = q < 0 || q >= REF_CAND || rec_s[bl][q][1] < 0;
                }
                mark = mark && tslot_s[bl] >= 0;                // (gram_round_kernel left its terms in a row; a block it did not is gathered if and when it is walked)
            }
            const unsigned long long mk = __ballot(mark);
            const int pos = n_pf + __popcll(mk & ((1ull << lane) - 1ull));
            if (mark && pos < PF_BLOCKS) pf_list[pos] = bl;
            n_pf += __popcll(mk);
        }
        if (n_pf > PF_BLOCKS) n_pf = PF_BLOCKS;
        __syncthreads();
#ifdef DEQSCI_DIAG
        stampA = __builtin_readcyclecounter();
#endif
        // (all of them in flight before the first is parked: one round trip in all.  A block with a slot in `terms` is 1 KB in a row -
        //  two terms per lane; one without is gathered from the history, 64 bytes apart)
        float4 tq[PF_BLOCKS];
#pragma unroll
        for (int x = 0; x < PF_BLOCKS; ++x) {
            const int sl = x < n_pf ? tslot_s[pf_list[x]] : 0;
            tq[x] = ld4(terms + ((((int64_t)j * TSLOTS + sl) * C16 + c) * PF_TERMS + 2 * lane) * 2);
        }
#pragma unroll
        for (int x = 0; x < PF_BLOCKS; ++x) {
            pfa[x][2 * lane] = tq[x].x; pfb[x][2 * lane] = tq[x].y; pfa[x][2 * lane + 1] = tq[x].z; pfb[x][2 * lane + 1] = tq[x].w;
        }
#ifdef DEQSCI_DIAG
        stampB = __builtin_readcyclecounter();
#endif
        __syncthreads();
    }
#ifdef DEQSCI_DIAG
    const long long stamp2 = __builtin_readcyclecounter();
#endif
    float S = 0.0f;
    int b = 0, walked = 0;
    while (b < nchunks) {
        if (b >= win0 + WIN) load_window(b);
#ifdef DEQSCI_DIAG
        const long long g0 = __builtin_readcyclecounter();
#endif
        // ---- the next 64 blocks at once: which of them does S pass through without leaving its binade?
        const int bl = b + lane;
        const bool valid = bl < nchunks && bl < win0 + WIN;
        const float aS = fabsf(S);
        const int e = (aS > 0.0f && aS < 3.0e38f) ? ilogbf(aS) : -2000;
        const int code = valid ? code_s[bl - win0] : REF_NONE;
        const int q = code == REF_NONE ? -1 : e - ref_lo(code);
        int nn = 0, aa = -1;
        if (q >= 0 && q < REF_CAND) {
            nn = rec_s[bl - win0][q][0];
            aa = rec_s[bl - win0][q][1];
        }
        bool ok = aa >= 0 && aa < (1 << 22);
        if (!ok) nn = 0;
        const int pre = wave_scan_incl(nn, lane);               // inclusive prefix over the lanes (integers: exact)
        const int si = (e > -200 && e < 100) ? (int)(S * ldexpf(1.0f, 23 - e)) : 0;      // S / u: +-[2^23, 2^24), exact
        const int v0 = si + pre - nn;                           // where the chain stands when it reaches my block, if all before it were passed
        const int av = v0 < 0 ? -v0 : v0;
        ok = ok && ((v0 < 0) == (si < 0)) && av - aa >= (1 << 23) + 2 && av + aa <= (1 << 24) - 2;
        const unsigned long long pass = __ballot(ok);
        int f = pass == ~0ull ? WAVE : __builtin_ctzll(~pass);  // the first block that is not passed
        if (b + f > win0 + WIN) f = win0 + WIN - b;             // (the window's end is not a failure: the next round reloads)
        if (f > 0) {
            const int tot = __builtin_amdgcn_readlane(pre, __builtin_amdgcn_readfirstlane(f - 1));
            S = (float)(si + tot) * ldexpf(1.0f, e - 23);       // (|si + tot| < 2^24: exact)
            b += f;
        }
#ifdef DEQSCI_DIAG
        const long long g1 = __builtin_readcyclecounter();
        c_group += g1 - g0;
        n_group += 1;
#endif
        if (f < WAVE && b < nchunks && b < win0 + WIN) {
            // ---- walk block b: its terms k = beg + c + 16 i, one FMA after the other
            const int64_t beg = (int64_t)b * chunk;
            const int64_t end = beg + chunk < N ? beg + chunk : N;
            int T = (int)((end - beg - c + C16 - 1) / C16);
            if (T < 0) T = 0;
            const float *xa = wa, *xb = wb;
            const unsigned long long hit = __ballot(lane < n_pf && pf_list[lane < PF_BLOCKS ? lane : 0] == b);
            bool have = false;
            if (hit) {
                const int x = __builtin_ctzll(hit);
                xa = pfa[x];
                xb = pfb[x];
                have = true;
            }
            if (!have) {
                if (small && b - win0 < WIN && tslot_s[b - win0] >= 0) {
                    const float4 v = ld4(terms + ((((int64_t)j * TSLOTS + tslot_s[b - win0]) * C16 + c) * PF_TERMS + 2 * lane) * 2);
                    wa[2 * lane] = v.x; wb[2 * lane] = v.y; wa[2 * lane + 1] = v.z; wb[2 * lane + 1] = v.w;
                } else if (small) {
                    float da[PF_TERMS / WAVE], db[PF_TERMS / WAVE];
                    ask_terms(b, da, db);
#pragma unroll
                    for (int h = 0; h < PF_TERMS / WAVE; ++h) { wa[lane + WAVE * h] = da[h]; wb[lane + WAVE * h] = db[h]; }
                } else {
                    for (int i = lane; i < T; i += WAVE) { wa[i] = ga[beg + c + (int64_t)C16 * i]; wb[i] = gb[beg + c + (int64_t)C16 * i]; }
                }
            }
            __syncthreads();
            if (small) {
                // 128 dependent FMAs (terms past the block are zeros); the operands come 32 terms ahead of the chain
                const float4* xa4 = reinterpret_cast<const float4*>(xa);
                const float4* xb4 = reinterpret_cast<const float4*>(xb);
                float4 A4[2][8], B4[2][8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { A4[0][k] = xa4[k]; B4[0][k] = xb4[k]; }
#pragma unroll
                for (int ch = 0; ch < PF_TERMS / 32; ++ch) {
                    if (ch + 1 < PF_TERMS / 32) {
#pragma unroll
                        for (int k = 0; k < 8; ++k) { A4[(ch + 1) & 1][k] = xa4[8 * (ch + 1) + k]; B4[(ch + 1) & 1][k] = xb4[8 * (ch + 1) + k]; }
                    }
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const float4 va = A4[ch & 1][k], vb = B4[ch & 1][k];
                        S = fmaf(va.x, vb.x, S); S = fmaf(va.y, vb.y, S); S = fmaf(va.z, vb.z, S); S = fmaf(va.w, vb.w, S);
                    }
                }
            } else {
                int i = 0;
                for (; i + 16 <= T; i += 16) {
                    float av2[16], bv2[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) { av2[u] = xa[i + u]; bv2[u] = xb[i + u]; }
#pragma unroll
                    for (int u = 0; u < 16; ++u) S = fmaf(av2[u], bv2[u], S);
                }
                for (; i < T; ++i) S = fmaf(xa[i], xb[i], S);
            }
            __syncthreads();
            b += 1;
            walked += 1;
#ifdef DEQSCI_DIAG
            c_walk += __builtin_readcyclecounter() - g1;
            if (lane == 0 && j == 0 && c == 0 && walked <= 24) {
                int* dg = reinterpret_cast<int*>(st) + REF_WALKED + 7 * C16;
                dg[walked - 1] = (b - 1) | (hit ? 0x1000 : 0) | ((b - 1 - win0 < WIN && tslot_s[b - 1 - win0] >= 0) ? 0x2000 : 0);      // walked block | was prefetched | has a slot
            }
            n_hit += hit ? 1 : 0;
            n_miss += have ? 0 : 1;
#endif
        }
    }
#ifdef DEQSCI_DIAG
    if (lane == 0 && j == 0 && c == 0) {                        // (the unused rows of the counters: one chain's account of its time, in cycles)
        int* dg = reinterpret_cast<int*>(st) + REF_WALKED + 6 * C16;
        dg[0] = (int)(stamp1 - stamp0); dg[1] = (int)(stamp2 - stamp1); dg[2] = (int)c_group; dg[3] = (int)c_walk;
        dg[4] = n_group; dg[5] = walked; dg[6] = n_hit; dg[7] = n_noslot; dg[8] = n_miss; dg[9] = (int)(__builtin_readcyclecounter() - stamp0);
        dg[10] = (int)(stampA - stamp1); dg[11] = (int)(stampB - stampA); dg[12] = n_pf;
    }
#endif
    if (lane == 0) {
        st[REF_CSUM + j * C16 + c] = S;
        reinterpret_cast<int*>(st)[REF_WALKED + j * C16 + c] = walked;
        if (c == 0) reinterpret_cast<int*>(st)[REF_TAKEN + j] = 0;              // (gram_round_kernel's term slots: free again for the next call)
        if (c == 0 && j == 0) reinterpret_cast<unsigned*>(st)[REF_CALL] += 1u;   // (the call is over: residual_store_round_kernel tags its granules with the next number)
    }
}

__global__ __launch_bounds__(WAVE) void anderson_solve_kernel(const float* __restrict__ partials, double* gram,
                                                              float* __restrict__ alpha, float* res, int bsz,
                                                              int nchunks, int slot, int n_filled, int n, float lam, float eps, int solve_f32,
                                                              const float* __restrict__ gram32, float* ref_state, int64_t ref_stride
#ifdef DEQSCI_DIAG
                                                              , float gram_noise
#endif
                                                              ) {
    // one wavefront per sample; the last block to arrive folds the per-sample norms into the
    // whole-batch residual (agent-scope release -> ticket -> acquire; the ticket resets itself).
    constexpr int NN = MAXM + 1;                                // rows of the largest bordered system
    __shared__ double Gl[GRAM_STRIDE];                          // this sample's Gram matrix (+ |F|^2, |G|^2)
    __shared__ double M[NN][NN + 1];
    const int s = blockIdx.x, lane = threadIdx.x;
    double* gs = gram + (int64_t)s * GRAM_STRIDE;
    unsigned* ticket = reinterpret_cast<unsigned*>(gram + (int64_t)bsz * GRAM_STRIDE);
    const float* ps = partials + (int64_t)s * nchunks * PART_STRIDE;
    // ---- fp64 finish of the block partials: column j < n_filled = <G_slot, G_j>, column MAXM = |F_slot|^2
    double a[PART_STRIDE];
#pragma unroll
    for (int j = 0; j < PART_STRIDE; ++j) a[j] = 0.0;
    for (int c = lane; c < nchunks; c += WAVE) {
        const float* row = ps + (int64_t)c * PART_STRIDE;
#pragma unroll
        for (int j = 0; j < PART_STRIDE; ++j) a[j] += (double)row[j];
    }
    for (int i = lane; i < MAXM * MAXM + 2; i += WAVE) Gl[i] = gs[i];      // the rows of the other slots, from earlier iterations
#pragma unroll
    for (int j = 0; j < PART_STRIDE; ++j) a[j] = wave_sum(a[j]);           // totals in lane 0
#ifdef DEQSCI_DIAG
    if (gram_noise != 0.0f && lane == 0) {
        // DIAGNOSTIC (-DDEQSCI_DIAG build only, env DEQSCI_GRAM_NOISE): uniform relative noise of that amplitude on the new Gram row, to
        // study how the rounding of the reference's fp32 torch.bmm Gram (~1e-6 mean, 5e-6 max at N = 2^19) steers the chaotic
        // FFDNet + Anderson runs
        unsigned h = (unsigned)__double_as_longlong(a[MAXM]) * 2654435761u + (unsigned)(__double_as_longlong(a[MAXM]) >> 32);
#pragma unroll
        for (int j = 0; j < MAXM; ++j) {
            h = h * 1664525u + 1013904223u;
            a[j] *= 1.0 + (double)gram_noise * (((h >> 8) * (1.0 / 16777216.0)) * 2.0 - 1.0);
        }
    }
#endif
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < MAXM; ++j)
            if (j < n_filled) { Gl[slot * MAXM + j] = a[j]; Gl[j * MAXM + slot] = a[j]; gs[slot * MAXM + j] = a[j]; gs[j * MAXM + slot] = a[j]; }
        Gl[MAXM * MAXM] = a[MAXM];                              // |F_k|^2
        gs[MAXM * MAXM] = a[MAXM];
    }
    __syncthreads();
    const double ff = Gl[MAXM * MAXM], gg = Gl[slot * MAXM + slot];
    if (lane == 0) {
        gs[MAXM * MAXM + 1] = gg;                               // |G_k|^2
        res[1 + s] = (float)(sqrt(gg) / ((double)eps + sqrt(ff)));
    }
    if (ref_state) {
        // anderson_arith = "reference": the sixteen chain sums of each entry of the new row (gram_row_chain16_kernel or the two-pass form of the
        // same sums below) folded halves onto halves, row / column `slot` of the persistent fp32 Gram refreshed, the system solved in fp32
        __shared__ float row32[MAXM];
        float* g32 = ref_state + (int64_t)s * ref_stride;
        const float* csum = g32 + REF_CSUM;
#pragma unroll
        for (int jb = 0; jb < MAXM / 4; ++jb) {
            const int j = 4 * jb + (lane >> 4);
            float v = (j < n_filled) ? csum[j * C16 + (lane & (C16 - 1))] : 0.0f;
            v += __shfl_xor(v, 8, WAVE);                        // 16 -> 8 -> 4 -> 2 -> 1
            v += __shfl_xor(v, 4, WAVE);
            v += __shfl_xor(v, 2, WAVE);
            v += __shfl_xor(v, 1, WAVE);
            if (j < n_filled && (lane & (C16 - 1)) == 0) {
                row32[j] = v;
                g32[slot * MAXM + j] = v;
                g32[j * MAXM + slot] = v;
            }
        }
        __syncthreads();
        if (n > 0) {
            for (int i = lane; i < n * n; i += WAVE) {
                const int a = i / n, b = i % n;
                Gl[a * MAXM + b] = (double)((a == slot) ? row32[b] : (b == slot) ? row32[a] : g32[a * MAXM + b]);
            }
            __syncthreads();
            bordered_solve<float>(Gl, M, alpha, s, lane, n, lam);
        }
    } else if (n > 0) {
        if (gram32) {
            // the REFERENCE's arithmetic for alpha (new_equilibrium_utils_yaping.py:177-180): the n x n Gram block as the caller's fp32
            // torch.bmm produced it (rows in slot order), the system formed and factorised in fp32 like torch.solve = sgesv.  The residual
            // above and the persistent float64 Gram keep their own, exact, sums.
            __syncthreads();
            for (int i = lane; i < n * n; i += WAVE) Gl[(i / n) * MAXM + (i % n)] = (double)gram32[(int64_t)s * n * n + i];
            __syncthreads();
            bordered_solve<float>(Gl, M, alpha, s, lane, n, lam);
        } else if (solve_f32) bordered_solve<float>(Gl, M, alpha, s, lane, n, lam);
        else bordered_solve<double>(Gl, M, alpha, s, lane, n, lam);
    }
    if (lane != 0) return;
    if (bsz == 1) { res[0] = res[1]; return; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t == (unsigned)bsz - 1u) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        double sg = 0.0, sf = 0.0;
        for (int k = 0; k < bsz; ++k) {                         // fixed order: deterministic
            sf += __hip_atomic_load(gram + (int64_t)k * GRAM_STRIDE + MAXM * MAXM, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sg += __hip_atomic_load(gram + (int64_t)k * GRAM_STRIDE + MAXM * MAXM + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        res[0] = (float)(sqrt(sg) / ((double)eps + sqrt(sf)));  // whole-batch norms, :184
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------------------------------------
// K7 (flat) and K7+K3 (HWB / BHW)
// ------------------------------------------------------------------------------------------------
struct Coef { float a[MAXM]; };

__device__ __forceinline__ Coef load_coef(const float* alpha, int64_t s, int n) {
    Coef c;
#pragma unroll
    for (int i = 0; i < MAXM; ++i) c.a[i] = i < n ? alpha[s * MAXM + i] : 0.0f;
    return c;
}

// x = beta * sum a_i F_i + (1-beta) * sum a_i (F_i - G_i) = sum a_i F_i - (1-beta) sum a_i G_i
template <int POL>
__device__ __forceinline__ float4 mix4(const float* Fs, const float* Gs, int64_t N, int64_t off, const Coef& c, int n, float omb) {
    float4 x = f4(0.0f);
#pragma unroll
    for (int i = 0; i < MAXM; ++i) if (i < n) x = fma4(c.a[i], ldp<POL>(Fs + i * N + off), x);
    if (omb != 0.0f) {
        float4 g = f4(0.0f);
#pragma unroll
        for (int i = 0; i < MAXM; ++i) if (i < n) g = fma4(c.a[i], ldp<POL>(Gs + i * N + off), g);
        x = fma4(-omb, g, x);
    }
    return x;
}

template <int POL>
__global__ __launch_bounds__(TB) void mix_kernel(const float* __restrict__ F_hist, const float* __restrict__ G_hist,
                                                 const float* __restrict__ alpha, float* __restrict__ x_out, float omb, int n,
                                                 int64_t N, int m, int vec) {
    const int64_t s = blockIdx.y;
    const Coef c = load_coef(alpha, s, n);
    const float* Fs = F_hist + s * m * N;
    const float* Gs = G_hist + s * m * N;
    float* xs = x_out + s * N;
    if (vec) {
        const int64_t i = ((int64_t)blockIdx.x * TB + threadIdx.x) * 4;
        if (i < N) stp<POL>(xs + i, mix4<POL>(Fs, Gs, N, i, c, n, omb));
    } else {
        const int64_t i0 = ((int64_t)blockIdx.x * TB + threadIdx.x) * 4;
        const int64_t i1 = i0 + 4 < N ? i0 + 4 : N;
        for (int64_t i = i0; i < i1; ++i) {
            float x = 0.0f, g = 0.0f;
            for (int k = 0; k < n; ++k) { x = fmaf(c.a[k], Fs[k * N + i], x); g = fmaf(c.a[k], Gs[k * N + i], g); }
            xs[i] = omb != 0.0f ? fmaf(-omb, g, x) : x;
        }
    }
}

constexpr int UNR = 2;

template <int LP, int POL>
__global__ __launch_bounds__(TB) void mix_gap_hwb_kernel(const float* __restrict__ F_hist, const float* __restrict__ G_hist,
                                                         const float* __restrict__ alpha, float omb, int n, int m,
                                                         const float* __restrict__ phi, const float* __restrict__ y,
                                                         const float* __restrict__ phisum, float* __restrict__ x_out,
                                                         float* __restrict__ z1, int64_t P, int phi_shared) {
    const int64_t s = blockIdx.y;
    const int64_t Q = P * LP, N = Q * 4;
    const Coef c = load_coef(alpha, s, n);
    const float* Fs = F_hist + s * m * N;
    const float* Gs = G_hist + s * m * N;
    const float* ps = phi + (phi_shared ? 0 : s * N);
    const float* ys = y + s * P;
    const float* ss = phisum + (phi_shared ? 0 : s * P);
    const int64_t base = (int64_t)blockIdx.x * (TB * UNR) + threadIdx.x;
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
        const int64_t q = base + j * TB;
        const int64_t qc = q < Q ? q : Q - 1;
        const float4 x = mix4<POL>(Fs, Gs, N, qc * 4, c, n, omb);
        const float4 pv = ldp<POL>(ps + qc * 4);
        const float fb = group_sum<LP>(dot4_seq(x, pv));
        const float r = (ys[qc / LP] - fb) / ss[qc / LP];
        if (q < Q) {
            stp<POL>(x_out + s * N + q * 4, x);
            stp<POL>(z1 + s * N + q * 4, x + r * pv);
        }
    }
}

// Planar (BHW) fused mix + GAP.  A block is BT frames x (256/BT) pixel-quads: lane (b,q) builds the
// new iterate for ONE frame of 4 adjacent pixels (n history rows + Phi: n+1 independent 16-B loads,
// 512-B contiguous per frame row and wave), the per-pixel frame column is staged in LDS, every lane
// re-reads its quad's BT partial products (conflict-free ds_read_b128) to form Phi x, and writes
// x and z1.  8x more wavefronts in flight than a lane-owns-the-column mapping at batch 8.
template <int BT, int POL>
__global__ __launch_bounds__(TB) void mix_gap_bhw_kernel(const float* __restrict__ F_hist, const float* __restrict__ G_hist,
                                                         const float* __restrict__ alpha, float omb, int n, int m,
                                                         const float* __restrict__ phi, const float* __restrict__ y,
                                                         const float* __restrict__ phisum, float* __restrict__ x_out,
                                                         float* __restrict__ z1, int64_t P, int phi_shared) {
    constexpr int QPB = TB / BT;
    __shared__ __attribute__((aligned(16))) float4 part[BT][QPB];
    const int b = threadIdx.x / QPB, q = threadIdx.x % QPB;
    const int64_t s = blockIdx.y;
    const int64_t p = ((int64_t)blockIdx.x * QPB + q) * 4;
    const bool ok = p < P;
    const int64_t pc = ok ? p : 0;
    const int64_t N = (int64_t)BT * P;
    const Coef c = load_coef(alpha, s, n);
    const int64_t off = (int64_t)b * P + pc;
    const float4 xv = mix4<POL>(F_hist + s * m * N, G_hist + s * m * N, N, off, c, n, omb);
    const float4 pv = ldp<POL>(phi + (phi_shared ? 0 : s * N) + off);
    const float4 yv = ld4(y + s * P + pc);
    const float4 sv = ld4(phisum + (phi_shared ? 0 : s * P) + pc);
    part[b][q] = xv * pv;
    __syncthreads();
    float4 fb = part[0][q];
#pragma unroll
    for (int k = 1; k < BT; ++k) fb = fb + part[k][q];
    const float4 r = (yv - fb) / sv;
    if (ok) {
        stp<POL>(x_out + s * N + off, xv);
        stp<POL>(z1 + s * N + off, xv + r * pv);
    }
}

#define POL2_DISPATCH(pol, ...)                                            \
    switch (pol) {                                                         \
        case POL_NTL:  { constexpr int POL = POL_NTL; __VA_ARGS__; } break;  \
        case POL_NTS:  { constexpr int POL = POL_NTS; __VA_ARGS__; } break;  \
        case POL_NTLS: { constexpr int POL = POL_NTLS; __VA_ARGS__; } break; \
        default:       { constexpr int POL = POL_DEFAULT; __VA_ARGS__; } break; \
    }

static inline int64_t chunk_elems(int64_t bsz, int64_t N) {
    // target number of blocks over the whole batch (16384 measured best of 1024..32768 at bsz 64, flat at bsz 8;
    // tuning knob DEQSCI_K4_BLOCKS of the -DDEQSCI_DIAG build), each block a whole number of 1024-element sweeps (>= 2)
#ifdef DEQSCI_DIAG
    const int64_t target = diag_env_int("DEQSCI_K4_BLOCKS", 16384);
#else
    constexpr int64_t target = 16384;
#endif
    int64_t per_sample = target / (bsz > 0 ? bsz : 1);
    if (per_sample < 1) per_sample = 1;
    int64_t chunk = ceil_div(ceil_div(N, per_sample), 1024) * 1024;
    if (chunk < 2048) chunk = 2048;
    return chunk;
}

}  // namespace deqsci

using namespace deqsci;

extern "C" {

int64_t deqsci_anderson_chunks(int64_t bsz, int64_t N) {
    if (bsz <= 0 || N <= 0) return 0;
    return ceil_div(N, chunk_elems(bsz, N));
}

size_t deqsci_partials_bytes(int64_t bsz, int64_t N) {
    return (size_t)(bsz * deqsci_anderson_chunks(bsz, N)) * PART_STRIDE * sizeof(float);
}

size_t deqsci_gram_bytes(int64_t bsz) { return (size_t)(bsz * GRAM_STRIDE + 2) * sizeof(double); }   // + arrival ticket

int deqsci_residual_store_f32(const float* z1, const float* noise, const float* x_cur, float* F_hist, float* G_hist,
                              float* x_next, float* partials, int64_t bsz, int64_t N, int m, int slot, int n_filled,
                              deqsci_stream_t stream) {
    if (!z1 || !x_cur || !F_hist || !G_hist || !partials) return DEQSCI_ERR_NULL;
    if (bsz <= 0 || N <= 0 || m <= 0 || slot < 0 || slot >= m || n_filled < 1 || n_filled > m || slot >= n_filled) return DEQSCI_ERR_SHAPE;
    if (m > MAXM || bsz > 65535) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(z1) || (noise && !aligned16(noise)) || !aligned16(x_cur) || !aligned16(F_hist) || !aligned16(G_hist) ||
        (x_next && !aligned16(x_next)))
        return DEQSCI_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t chunk = chunk_elems(bsz, N);
    const dim3 grid(ceil_div(N, chunk), bsz);
    const int vec = (N % 4 == 0) ? 1 : 0;
    const int pol = pick_policy(bsz * N * 4 * (n_filled + 4), POL_NTLS);
#define RS_CASE(NF) case NF: POL2_DISPATCH(pol, hipLaunchKernelGGL((residual_store_kernel<NF, POL>), grid, dim3(TB), 0, st, z1, noise, x_cur, F_hist, G_hist, x_next, partials, N, m, slot, chunk, vec)); break;
    switch (n_filled) {
        RS_CASE(1) RS_CASE(2) RS_CASE(3) RS_CASE(4) RS_CASE(5) RS_CASE(6) RS_CASE(7) RS_CASE(8)
        default: return DEQSCI_ERR_UNSUPPORTED;
    }
#undef RS_CASE
    return launch_status();
}

int deqsci_anderson_solve_gram_f32(const float* partials, void* gram, float* alpha, float* res, int64_t bsz, int64_t N, int m,
                                   int slot, int n_filled, int n, float lam, float eps, const float* gram32, deqsci_stream_t stream);

static int solve_launch(const float* partials, void* gram, float* alpha, float* res, int64_t bsz, int64_t N, int slot, int n_filled, int n, float lam,
                        float eps, const float* gram32, float* ref_state, hipStream_t st) {
    const int nchunks = (int)deqsci_anderson_chunks(bsz, N);
    const int64_t ref_stride = ref_words(nchunks);
#ifdef DEQSCI_DIAG
    hipLaunchKernelGGL(anderson_solve_kernel, dim3((unsigned)bsz), dim3(WAVE), 0, st, partials, static_cast<double*>(gram), alpha, res,
                       (int)bsz, nchunks, slot, n_filled, n, lam, eps, diag_env_int("DEQSCI_SOLVE_F32", 0), gram32, ref_state, ref_stride,
                       (float)diag_env_f64("DEQSCI_GRAM_NOISE", 0.0));
#else
    hipLaunchKernelGGL(anderson_solve_kernel, dim3((unsigned)bsz), dim3(WAVE), 0, st, partials, static_cast<double*>(gram), alpha, res,
                       (int)bsz, nchunks, slot, n_filled, n, lam, eps, 0, gram32, ref_state, ref_stride);
#endif
    return launch_status();
}

int deqsci_anderson_solve_f32(const float* partials, void* gram, float* alpha, float* res, int64_t bsz, int64_t N, int m,
                              int slot, int n_filled, int n, float lam, float eps, deqsci_stream_t stream) {
    return deqsci_anderson_solve_gram_f32(partials, gram, alpha, res, bsz, N, m, slot, n_filled, n, lam, eps, nullptr, stream);
}

int deqsci_anderson_solve_gram_f32(const float* partials, void* gram, float* alpha, float* res, int64_t bsz, int64_t N, int m,
                                   int slot, int n_filled, int n, float lam, float eps, const float* gram32, deqsci_stream_t stream) {
    if (!partials || !gram || !res || (n > 0 && !alpha)) return DEQSCI_ERR_NULL;
    if (gram32 && n <= 0) return DEQSCI_ERR_SHAPE;
    if (bsz <= 0 || N <= 0 || m <= 0 || slot < 0 || slot >= m || n_filled < 1 || n_filled > m || n < 0 || n > n_filled) return DEQSCI_ERR_SHAPE;
    if (m > MAXM || bsz > 65535) return DEQSCI_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    return solve_launch(partials, gram, alpha, res, bsz, N, slot, n_filled, n, lam, eps, gram32, nullptr, st);
}

size_t deqsci_gram_ref_bytes(int64_t bsz, int64_t N) {
    if (bsz <= 0 || N <= 0) return 0;
    return (size_t)bsz * (size_t)ref_words(deqsci_anderson_chunks(bsz, N)) * sizeof(float);
}

static int ref_check(const float* G_hist, const float* partials, float* ref_state, int64_t bsz, int64_t N, int m, int slot, int n_filled) {
    if (!G_hist || !partials || !ref_state) return DEQSCI_ERR_NULL;
    if (bsz <= 0 || N <= 0 || m <= 0 || slot < 0 || slot >= m || n_filled < 1 || n_filled > m || slot >= n_filled) return DEQSCI_ERR_SHAPE;
    if (m > MAXM || bsz > 65535) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(G_hist) || !aligned16(ref_state)) return DEQSCI_ERR_ALIGN;
    return 0;
}

int deqsci_gram_row_chain16_f32(const float* G_hist, const float* partials, float* ref_state, int64_t bsz, int64_t N, int m, int slot, int n_filled,
                                int serial, deqsci_stream_t stream) {
    if (int rc = ref_check(G_hist, partials, ref_state, bsz, N, m, slot, n_filled)) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t chunk = chunk_elems(bsz, N);
    const int nchunks = (int)deqsci_anderson_chunks(bsz, N);
    const int64_t ref_stride = ref_words(nchunks);
    if (serial || N % 4 != 0 || chunk > 16384) {                // (the two-pass form wants float4 rows and <= 1024 terms per chain and block)
        hipLaunchKernelGGL(gram_row_chain16_kernel, dim3((unsigned)ceil_div(n_filled, 4), (unsigned)bsz), dim3(C16_TB), 0, st, G_hist, ref_state, ref_stride, N, m,
                           slot, n_filled, (N % 4 == 0) ? 1 : 0);
        return launch_status();
    }
    const dim3 grid((unsigned)nchunks, (unsigned)bsz);
#define RND_CASE(NF) case NF: hipLaunchKernelGGL(gram_round_kernel<NF>, grid, dim3(TB), 0, st, G_hist, partials, ref_state, ref_stride, N, m, slot, chunk); break;
    switch (n_filled) {
        RND_CASE(1) RND_CASE(2) RND_CASE(3) RND_CASE(4) RND_CASE(5) RND_CASE(6) RND_CASE(7) RND_CASE(8)
        default: return DEQSCI_ERR_UNSUPPORTED;
    }
#undef RND_CASE
    if (int rc = launch_status()) return rc;
    hipLaunchKernelGGL(gram_chain_apply_kernel, dim3((unsigned)(n_filled * C16), (unsigned)bsz), dim3(WAVE), 0, st, G_hist, partials, ref_state, ref_stride, N, m, slot, chunk,
                       nchunks);
    return launch_status();
}

// (N / 2048 <= 256 blocks per sample: every block reads ALL its predecessors' granules - at N = 2^22, 2048 blocks per sample in several residency
//  rounds, the fused launch is SLOWER than the two launches (bsz 2: 241 against 202 us for K4 + Gram + solve; bsz 8: 770 against 557:
//  tools/gram_fused_time.py with GRAM_BIG=1).  A two-level look-back - the group's 255 predecessors + one inclusive prefix per group of 256 - was
//  built and measured: 176 / 563 us for the fused launch against 186 / 642, still behind K4 + the separate pass (48 + 80 / 202 + 140): at four and
//  more residency rounds of blocks that wait for one another the launch is slower than the sum of its parts, not its look-back; not kept)
int deqsci_gram_ref_fusable(int64_t bsz, int64_t N) {
    return (bsz > 0 && N > 0 && N % (2 * RND_TILE) == 0 && N / (2 * RND_TILE) <= 256 && chunk_elems(bsz, N) == 2 * RND_TILE) ? 1 : 0;
}

int deqsci_residual_store_ref_f32(const float* z1, const float* noise, const float* x_cur, float* F_hist, float* G_hist, float* x_next, float* partials,
                                  float* ref_state, int64_t bsz, int64_t N, int m, int slot, int n_filled, deqsci_stream_t stream) {
    if (!z1 || !x_cur || !F_hist || !G_hist || !partials || !ref_state) return DEQSCI_ERR_NULL;
    if (bsz <= 0 || N <= 0 || m <= 0 || slot < 0 || slot >= m || n_filled < 1 || n_filled > m || slot >= n_filled) return DEQSCI_ERR_SHAPE;
    if (m > MAXM || bsz > 65535 || !deqsci_gram_ref_fusable(bsz, N)) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(z1) || (noise && !aligned16(noise)) || !aligned16(x_cur) || !aligned16(F_hist) || !aligned16(G_hist) ||
        (x_next && !aligned16(x_next)) || !aligned16(ref_state))
        return DEQSCI_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nchunks = (int)deqsci_anderson_chunks(bsz, N);
    const int64_t ref_stride = ref_words(nchunks);
    const dim3 grid((unsigned)nchunks, (unsigned)bsz);
    const int pol = pick_policy(bsz * N * 4 * (n_filled + 4), POL_NTLS);
#define RSR_CASE(NF) case NF: POL2_DISPATCH(pol, hipLaunchKernelGGL((residual_store_round_kernel<NF, POL>), grid, dim3(TB), 0, st, z1, noise, x_cur, F_hist, G_hist, x_next, partials, ref_state, ref_stride, N, m, slot)); break;
    switch (n_filled) {
        RSR_CASE(1) RSR_CASE(2) RSR_CASE(3) RSR_CASE(4) RSR_CASE(5) RSR_CASE(6) RSR_CASE(7) RSR_CASE(8)
        default: return DEQSCI_ERR_UNSUPPORTED;
    }
#undef RSR_CASE
    return launch_status();
}

int deqsci_anderson_apply_solve_ref_f32(const float* G_hist, const float* partials, float* ref_state, void* gram, float* alpha, float* res, int64_t bsz,
                                        int64_t N, int m, int slot, int n_filled, int n, float lam, float eps, deqsci_stream_t stream) {
    if (!gram || !res || (n > 0 && !alpha)) return DEQSCI_ERR_NULL;
    if (n < 0 || n > n_filled) return DEQSCI_ERR_SHAPE;
    if (int rc = ref_check(G_hist, partials, ref_state, bsz, N, m, slot, n_filled)) return rc;
    if (!deqsci_gram_ref_fusable(bsz, N)) return DEQSCI_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int nchunks = (int)deqsci_anderson_chunks(bsz, N);
    hipLaunchKernelGGL(gram_chain_apply_kernel, dim3((unsigned)(n_filled * C16), (unsigned)bsz), dim3(WAVE), 0, st, G_hist, partials, ref_state, ref_words(nchunks), N, m, slot,
                       chunk_elems(bsz, N), nchunks);
    if (int rc = launch_status()) return rc;
    return solve_launch(partials, gram, alpha, res, bsz, N, slot, n_filled, n, lam, eps, nullptr, ref_state, st);
}

int deqsci_anderson_solve_ref_f32(const float* G_hist, const float* partials, float* ref_state, void* gram, float* alpha, float* res, int64_t bsz,
                                  int64_t N, int m, int slot, int n_filled, int n, float lam, float eps, deqsci_stream_t stream) {
    if (!gram || !res || (n > 0 && !alpha)) return DEQSCI_ERR_NULL;
    if (n < 0 || n > n_filled) return DEQSCI_ERR_SHAPE;
    if (int rc = deqsci_gram_row_chain16_f32(G_hist, partials, ref_state, bsz, N, m, slot, n_filled, 0, stream)) return rc;
    return solve_launch(partials, gram, alpha, res, bsz, N, slot, n_filled, n, lam, eps, nullptr, ref_state, static_cast<hipStream_t>(stream));
}

int deqsci_anderson_mix_f32(const float* F_hist, const float* G_hist, const float* alpha, float* x_out, float beta, int n,
                            int64_t bsz, int64_t N, int m, deqsci_stream_t stream) {
    if (!F_hist || !G_hist || !alpha || !x_out) return DEQSCI_ERR_NULL;
    if (bsz <= 0 || N <= 0 || m <= 0 || n < 1 || n > m) return DEQSCI_ERR_SHAPE;
    if (m > MAXM || bsz > 65535) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(F_hist) || !aligned16(G_hist) || !aligned16(x_out)) return DEQSCI_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int pol = pick_policy(bsz * N * 4 * (n + 1), POL_NTLS);
    POL2_DISPATCH(pol, hipLaunchKernelGGL(mix_kernel<POL>, dim3(ceil_div(ceil_div(N, 4), TB), bsz), dim3(TB), 0, st, F_hist, G_hist, alpha, x_out,
                                          1.0f - beta, n, N, m, (N % 4 == 0) ? 1 : 0));
    return launch_status();
}

static int mix_gap_impl(const float* F_hist, const float* G_hist, const float* alpha, float beta, int n, int m,
                        const float* phi, const float* y, const float* phisum, float* x_out, float* z1,
                        int64_t bsz, int64_t H, int64_t W, int64_t B, int layout, int phi_shared,
                        deqsci_stream_t stream, hipEvent_t ev0, hipEvent_t ev1) {
    if (!F_hist || !G_hist || !alpha || !phi || !y || !phisum || !x_out || !z1) return DEQSCI_ERR_NULL;
    if (bsz <= 0 || H <= 0 || W <= 0 || B <= 0 || m <= 0 || n < 1 || n > m) return DEQSCI_ERR_SHAPE;
    if (m > MAXM || bsz > 65535 || (layout != DEQSCI_LAYOUT_HWB && layout != DEQSCI_LAYOUT_BHW)) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(F_hist) || !aligned16(G_hist) || !aligned16(phi) || !aligned16(y) || !aligned16(phisum) || !aligned16(x_out) ||
        !aligned16(z1))
        return DEQSCI_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t P = H * W, N = P * B;
    const float omb = 1.0f - beta;
    const int pol = pick_policy(bsz * P * (4 * B * (n + 3) + 8), POL_NTLS);
    // hipExtLaunchKernelGGL stamps ev0/ev1 with the dispatch's own begin/end (what rocprofv3 reports); without events the
    // launch is an ordinary one (capturable into a hipGraph).
#define MG_ARGS F_hist, G_hist, alpha, omb, n, m, phi, y, phisum, x_out, z1, P, phi_shared
#define MG_LAUNCH(KERNEL, TP1, GRID) POL2_DISPATCH(pol, { if (ev0 || ev1) hipExtLaunchKernelGGL((KERNEL<TP1, POL>), GRID, dim3(TB), 0, st, ev0, ev1, 0, MG_ARGS); \
                                                           else hipLaunchKernelGGL((KERNEL<TP1, POL>), GRID, dim3(TB), 0, st, MG_ARGS); })
    if (layout == DEQSCI_LAYOUT_HWB && (B == 4 || B == 8 || B == 16 || B == 32)) {
        const int LPv = (int)(B / 4);
        const dim3 grid(ceil_div(P * LPv, TB * UNR), bsz);
        switch (LPv) {
            case 1: MG_LAUNCH(mix_gap_hwb_kernel, 1, grid); break;
            case 2: MG_LAUNCH(mix_gap_hwb_kernel, 2, grid); break;
            case 4: MG_LAUNCH(mix_gap_hwb_kernel, 4, grid); break;
            default: MG_LAUNCH(mix_gap_hwb_kernel, 8, grid); break;
        }
        return launch_status();
    }
    if (layout == DEQSCI_LAYOUT_BHW && P % 4 == 0 && (B == 4 || B == 8 || B == 16)) {
        const dim3 grid(ceil_div(P / 4, TB / B), bsz);
        if (B == 4) { MG_LAUNCH(mix_gap_bhw_kernel, 4, grid); }
        else if (B == 8) { MG_LAUNCH(mix_gap_bhw_kernel, 8, grid); }
        else { MG_LAUNCH(mix_gap_bhw_kernel, 16, grid); }
        return launch_status();
    }
#undef MG_LAUNCH
    if (ev0 || ev1) return DEQSCI_ERR_UNSUPPORTED;   // the unfused fallback is two launches: nothing single to time
    // any other shape: the two unfused kernels back to back (x_out is the only intermediate)
    int e = deqsci_anderson_mix_f32(F_hist, G_hist, alpha, x_out, beta, n, bsz, N, m, stream);
    if (e) return e;
    return deqsci_gap_update_f32(x_out, phi, y, phisum, z1, bsz, H, W, B, layout, layout, phi_shared, stream);
}

int deqsci_anderson_mix_gap_f32(const float* F_hist, const float* G_hist, const float* alpha, float beta, int n, int m,
                                const float* phi, const float* y, const float* phisum, float* x_out, float* z1,
                                int64_t bsz, int64_t H, int64_t W, int64_t B, int layout, int phi_shared,
                                deqsci_stream_t stream) {
    return mix_gap_impl(F_hist, G_hist, alpha, beta, n, m, phi, y, phisum, x_out, z1, bsz, H, W, B, layout, phi_shared, stream,
                        nullptr, nullptr);
}

int deqsci_anderson_mix_gap_timed_f32(const float* F_hist, const float* G_hist, const float* alpha, float beta, int n, int m,
                                      const float* phi, const float* y, const float* phisum, float* x_out, float* z1,
                                      int64_t bsz, int64_t H, int64_t W, int64_t B, int layout, int phi_shared,
                                      deqsci_stream_t stream, void* start_event, void* stop_event) {
    if (!start_event || !stop_event) return DEQSCI_ERR_NULL;
    return mix_gap_impl(F_hist, G_hist, alpha, beta, n, m, phi, y, phisum, x_out, z1, bsz, H, W, B, layout, phi_shared, stream,
                        static_cast<hipEvent_t>(start_event), static_cast<hipEvent_t>(stop_event));
}

/* measurement helpers: raw hipEvent handles for the timed launch above */
int deqsci_event_create(void** ev) {
    if (!ev) return DEQSCI_ERR_NULL;
    hipEvent_t e;
    hipError_t rc = hipEventCreate(&e);
    *ev = e;
    return (int)rc;
}
int deqsci_event_destroy(void* ev) { return ev ? (int)hipEventDestroy(static_cast<hipEvent_t>(ev)) : DEQSCI_ERR_NULL; }
int deqsci_event_elapsed_ms(void* start_event, void* stop_event, float* ms) {
    if (!start_event || !stop_event || !ms) return DEQSCI_ERR_NULL;
    return (int)hipEventElapsedTime(ms, static_cast<hipEvent_t>(start_event), static_cast<hipEvent_t>(stop_event));
}

}  // extern "C"
