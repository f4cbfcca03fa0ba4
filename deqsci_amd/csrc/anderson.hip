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

__global__ __launch_bounds__(C16_TB) void gram_row_chain16_kernel(const float* __restrict__ G_hist, float* __restrict__ gram32, int64_t N, int m, int slot,
                                                                  int n_filled, int vec) {
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
    acc += __shfl_xor(acc, 8, WAVE);                            // 16 -> 8 -> 4 -> 2 -> 1, halves onto halves
    acc += __shfl_xor(acc, 4, WAVE);
    acc += __shfl_xor(acc, 2, WAVE);
    acc += __shfl_xor(acc, 1, WAVE);
    if (live && c == 0) {
        float* g = gram32 + s * (MAXM * MAXM);
        g[slot * MAXM + j] = acc;
        g[j * MAXM + slot] = acc;
    }
}

__global__ __launch_bounds__(WAVE) void anderson_solve_kernel(const float* __restrict__ partials, double* gram,
                                                              float* __restrict__ alpha, float* res, int bsz,
                                                              int nchunks, int slot, int n_filled, int n, float lam, float eps, int solve_f32,
                                                              const float* __restrict__ gram32, int gram32_pitch
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
    if (n > 0) {
        if (gram32) {
            // the REFERENCE's arithmetic for alpha (new_equilibrium_utils_yaping.py:177-180): the n x n Gram block as the caller's fp32
            // torch.bmm produced it (rows in slot order), the system formed and factorised in fp32 like torch.solve = sgesv.  The residual
            // above and the persistent float64 Gram keep their own, exact, sums.
            __syncthreads();
            const int pitch = gram32_pitch ? gram32_pitch : n;                                  // (MAXM: the persistent Gram of gram_row_chain16_kernel)
            for (int i = lane; i < n * n; i += WAVE) Gl[(i / n) * MAXM + (i % n)] = (double)gram32[(int64_t)s * pitch * pitch + (i / n) * pitch + (i % n)];
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
                        float eps, const float* gram32, int pitch, hipStream_t st) {
    const int nchunks = (int)deqsci_anderson_chunks(bsz, N);
#ifdef DEQSCI_DIAG
    hipLaunchKernelGGL(anderson_solve_kernel, dim3((unsigned)bsz), dim3(WAVE), 0, st, partials, static_cast<double*>(gram), alpha, res,
                       (int)bsz, nchunks, slot, n_filled, n, lam, eps, diag_env_int("DEQSCI_SOLVE_F32", 0), gram32, pitch, (float)diag_env_f64("DEQSCI_GRAM_NOISE", 0.0));
#else
    hipLaunchKernelGGL(anderson_solve_kernel, dim3((unsigned)bsz), dim3(WAVE), 0, st, partials, static_cast<double*>(gram), alpha, res,
                       (int)bsz, nchunks, slot, n_filled, n, lam, eps, 0, gram32, pitch);
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
    return solve_launch(partials, gram, alpha, res, bsz, N, slot, n_filled, n, lam, eps, gram32, 0, st);
}

size_t deqsci_gram_ref_bytes(int64_t bsz) { return (size_t)(bsz > 0 ? bsz : 0) * MAXM * MAXM * sizeof(float); }

int deqsci_anderson_solve_ref_f32(const float* G_hist, const float* partials, float* gram32, void* gram, float* alpha, float* res, int64_t bsz,
                                  int64_t N, int m, int slot, int n_filled, int n, float lam, float eps, deqsci_stream_t stream) {
    if (!G_hist || !partials || !gram32 || !gram || !res || (n > 0 && !alpha)) return DEQSCI_ERR_NULL;
    if (bsz <= 0 || N <= 0 || m <= 0 || slot < 0 || slot >= m || n_filled < 1 || n_filled > m || n < 0 || n > n_filled) return DEQSCI_ERR_SHAPE;
    if (m > MAXM || bsz > 65535) return DEQSCI_ERR_UNSUPPORTED;
    if (!aligned16(G_hist)) return DEQSCI_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(gram_row_chain16_kernel, dim3((unsigned)ceil_div(n_filled, 4), (unsigned)bsz), dim3(C16_TB), 0, st, G_hist, gram32, N, m, slot, n_filled,
                       (N % 4 == 0) ? 1 : 0);
    if (int rc = launch_status()) return rc;
    return solve_launch(partials, gram, alpha, res, bsz, N, slot, n_filled, n, lam, eps, gram32, MAXM, st);
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
