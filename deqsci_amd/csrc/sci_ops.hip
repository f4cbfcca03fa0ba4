// SCI sensing operators and the fused GAP projection for MI355X (gfx950).
//
//   K1 sci_forward   y  = Phi x            (reference: A_torch_,  utils/cg_utils.py:85-90)
//   K2 sci_adjoint   x  = Phi^T y          (reference: At_torch_, utils/cg_utils.py:124-129)
//   O4 phi_sum       sum_b Phi, 0 -> 1     (training/sci_equilibrium_training.py:162-163)
//   K3 gap_update    z + Phi^T((y-Phi z)/Phi_sum)   (solvers/equilibrium_solvers_yaping.py:399-400)
//   transpose / residual_out: (bsz,H,W,B) <-> (bsz,B,H,W) through an LDS tile (ibid. :415,:417)
//
// All of these are streaming fp32 kernels bounded by HBM bandwidth (K3: (4B+2) flop per
// (12B+8) bytes), so the design goal is one fully coalesced 16-byte access per lane per
// instruction and no temporaries:
//   * HWB (frames innermost, the reference's API layout): the tensor is read as a flat float4
//     stream; the LP = B/4 adjacent lanes that hold one pixel's frame column combine their partial
//     dot products with a wave64 xor-butterfly (DPP shuffles), so every load/store instruction
//     moves 1 KiB contiguous per wavefront.
//   * BHW (planar, the denoiser's layout): one lane owns 4 adjacent pixels and walks the B frame
//     planes with stride H*W, keeping the frame column in registers; the B-reduction is in-register.
//   * HWB -> BHW (and back): the block stages a 256-pixel x B tile in LDS, frame-major with a
//     +4 float row pad (HWB-side scalar accesses conflict-free for LP <= 2, 2-way = free for LP = 4),
//     so both the HWB reads and the planar writes are full-width coalesced.
// blockIdx.y is the measurement index, so no 64-bit div/mod appears in any address computation.
#include "common.hpp"

namespace deqsci {

constexpr int UNR = 4;          // independent float4 positions per lane (HWB kernels)
constexpr int TP = 256;         // pixels per LDS tile
constexpr int TS = TP + 4;      // LDS row stride (floats): 16-B aligned rows, bank-shifted by 4

// ------------------------------------------------------------------------------------------------
// HWB fast paths: B = 4*LP, lane -> (pixel, quarter)
// ------------------------------------------------------------------------------------------------
template <int LP, int POL>
__global__ __launch_bounds__(TB) void forward_hwb_kernel(const float* __restrict__ x, const float* __restrict__ phi,
                                                         float* __restrict__ y, int64_t P, int phi_shared) {
    const int64_t n = blockIdx.y;
    const int64_t Q = P * LP;                          // float4 per measurement
    const float* xs = x + n * Q * 4;
    const float* ps = phi + (phi_shared ? 0 : n * Q * 4);
    float* ys = y + n * P;
    const int64_t base = (int64_t)blockIdx.x * (TB * UNR) + threadIdx.x;
    float4 xv[UNR], pv[UNR];
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
        const int64_t q = base + j * TB;
        const int64_t qc = q < Q ? q : Q - 1;
        xv[j] = ldp<POL>(xs + qc * 4);
        pv[j] = ldp<POL>(ps + qc * 4);
    }
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
        const int64_t q = base + j * TB;
        const float s = group_sum<LP>(dot4_seq(xv[j], pv[j]));
        if (q < Q && (q & (LP - 1)) == 0) ys[q / LP] = s;
    }
}

template <int LP, int POL>
__global__ __launch_bounds__(TB) void adjoint_hwb_kernel(const float* __restrict__ y, const float* __restrict__ phi,
                                                         float* __restrict__ x, int64_t P, int phi_shared) {
    const int64_t n = blockIdx.y;
    const int64_t Q = P * LP;
    const float* ys = y + n * P;
    const float* ps = phi + (phi_shared ? 0 : n * Q * 4);
    float* xs = x + n * Q * 4;
    const int64_t base = (int64_t)blockIdx.x * (TB * UNR) + threadIdx.x;
    float4 pv[UNR];
    float yv[UNR];
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
        const int64_t q = base + j * TB;
        const int64_t qc = q < Q ? q : Q - 1;
        pv[j] = ldp<POL>(ps + qc * 4);
        yv[j] = ys[qc / LP];
    }
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
        const int64_t q = base + j * TB;
        if (q < Q) stp<POL>(xs + q * 4, yv[j] * pv[j]);
    }
}

template <int LP, int POL>
__global__ __launch_bounds__(TB) void phisum_hwb_kernel(const float* __restrict__ phi, float* __restrict__ out, int64_t P) {
    const int64_t n = blockIdx.y;
    const int64_t Q = P * LP;
    const float* ps = phi + n * Q * 4;
    float* os = out + n * P;
    const int64_t base = (int64_t)blockIdx.x * (TB * UNR) + threadIdx.x;
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
        const int64_t q = base + j * TB;
        const int64_t qc = q < Q ? q : Q - 1;
        const float4 p = ldp<POL>(ps + qc * 4);
        const float s = group_sum<LP>(((p.x + p.y) + p.z) + p.w);
        if (q < Q && (q & (LP - 1)) == 0) os[q / LP] = (s == 0.0f) ? 1.0f : s;
    }
}

// z1 may alias z: every element is read and written by the same lane, loads precede stores.
template <int LP, int POL>
__global__ __launch_bounds__(TB) void gap_hwb_kernel(const float* z, const float* __restrict__ phi,
                                                     const float* __restrict__ y, const float* __restrict__ phisum,
                                                     float* z1, int64_t P, int phi_shared) {
    const int64_t n = blockIdx.y;
    const int64_t Q = P * LP;
    const float* zs = z + n * Q * 4;
    const float* ps = phi + (phi_shared ? 0 : n * Q * 4);
    const float* ys = y + n * P;
    const float* ss = phisum + (phi_shared ? 0 : n * P);
    float* os = z1 + n * Q * 4;
    const int64_t base = (int64_t)blockIdx.x * (TB * UNR) + threadIdx.x;
    float4 zv[UNR], pv[UNR];
    float yv[UNR], sv[UNR];
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
        const int64_t q = base + j * TB;
        const int64_t qc = q < Q ? q : Q - 1;
        zv[j] = ldp<POL>(zs + qc * 4);
        pv[j] = ldp<POL>(ps + qc * 4);
        yv[j] = ys[qc / LP];
        sv[j] = ss[qc / LP];
    }
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
        const int64_t q = base + j * TB;
        const float fb = group_sum<LP>(dot4_seq(zv[j], pv[j]));
        const float r = (yv[j] - fb) / sv[j];
        if (q < Q) stp<POL>(os + q * 4, zv[j] + r * pv[j]);
    }
}

// HWB in, BHW out: GAP projection fused with the (H,W,B)->(B,H,W) transpose the denoiser needs.
template <int LP, int POL>
__global__ __launch_bounds__(TB) void gap_hwb2bhw_kernel(const float* __restrict__ z, const float* __restrict__ phi,
                                                         const float* __restrict__ y, const float* __restrict__ phisum,
                                                         float* __restrict__ z1, int64_t P, int phi_shared) {
    constexpr int B = 4 * LP;
    __shared__ __attribute__((aligned(16))) float tile[B * TS];
    const int64_t n = blockIdx.y;
    const int64_t pix0 = (int64_t)blockIdx.x * TP;
    const float* zs = z + n * P * B;
    const float* ps = phi + (phi_shared ? 0 : n * P * B);
    const float* ys = y + n * P;
    const float* ss = phisum + (phi_shared ? 0 : n * P);
    float4 zv[LP], pv[LP];
    float yv[LP], sv[LP];
#pragma unroll
    for (int j = 0; j < LP; ++j) {
        const int e = j * TB + threadIdx.x;
        const int pl = e / LP, qq = e % LP;
        int64_t p = pix0 + pl;
        if (p >= P) p = P - 1;
        zv[j] = ldp<POL>(zs + p * B + 4 * qq);
        pv[j] = ldp<POL>(ps + p * B + 4 * qq);
        yv[j] = ys[p];
        sv[j] = ss[p];
    }
#pragma unroll
    for (int j = 0; j < LP; ++j) {
        const int e = j * TB + threadIdx.x;
        const int pl = e / LP, qq = e % LP;
        const float fb = group_sum<LP>(dot4_seq(zv[j], pv[j]));
        const float r = (yv[j] - fb) / sv[j];
        const float4 o = zv[j] + r * pv[j];
        float* t = tile + (4 * qq) * TS + pl;
        t[0] = o.x; t[TS] = o.y; t[2 * TS] = o.z; t[3 * TS] = o.w;
    }
    __syncthreads();
    float* os = z1 + n * P * B;
#pragma unroll
    for (int j = 0; j < LP; ++j) {
        const int e = j * TB + threadIdx.x;
        const int b = e / (TP / 4), p4 = e % (TP / 4);
        const int64_t p = pix0 + 4 * p4;
        if (p < P) stp<POL>(os + b * P + p, *reinterpret_cast<const float4*>(tile + b * TS + 4 * p4));
    }
}

// planar in (z1, noise), HWB out: out = z1 - noise
template <int LP, int POL>
__global__ __launch_bounds__(TB) void residual_out_bhw2hwb_kernel(const float* __restrict__ z1, const float* __restrict__ noise,
                                                                  float* __restrict__ out, int64_t P) {
    constexpr int B = 4 * LP;
    __shared__ __attribute__((aligned(16))) float tile[B * TS];
    const int64_t n = blockIdx.y;
    const int64_t pix0 = (int64_t)blockIdx.x * TP;
    const float* zs = z1 + n * P * B;
    const float* ns = noise ? noise + n * P * B : nullptr;
#pragma unroll
    for (int j = 0; j < LP; ++j) {
        const int e = j * TB + threadIdx.x;
        const int b = e / (TP / 4), p4 = e % (TP / 4);
        const int64_t p = pix0 + 4 * p4;
        if (p < P) {
            float4 v = ldp<POL>(zs + b * P + p);
            if (ns) v = v - ldp<POL>(ns + b * P + p);
            *reinterpret_cast<float4*>(tile + b * TS + 4 * p4) = v;
        }
    }
    __syncthreads();
    float* os = out + n * P * B;
#pragma unroll
    for (int j = 0; j < LP; ++j) {
        const int e = j * TB + threadIdx.x;
        const int pl = e / LP, qq = e % LP;
        const int64_t p = pix0 + pl;
        if (p < P) {
            const float* t = tile + (4 * qq) * TS + pl;
            stp<POL>(os + p * B + 4 * qq, make_float4(t[0], t[TS], t[2 * TS], t[3 * TS]));
        }
    }
}

// HWB -> BHW plain transpose (LP fast path)
template <int LP, int POL>
__global__ __launch_bounds__(TB) void transpose_hwb2bhw_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t P) {
    constexpr int B = 4 * LP;
    __shared__ __attribute__((aligned(16))) float tile[B * TS];
    const int64_t n = blockIdx.y;
    const int64_t pix0 = (int64_t)blockIdx.x * TP;
    const float* is = in + n * P * B;
#pragma unroll
    for (int j = 0; j < LP; ++j) {
        const int e = j * TB + threadIdx.x;
        const int pl = e / LP, qq = e % LP;
        const int64_t p = pix0 + pl;
        if (p < P) {
            const float4 v = ldp<POL>(is + p * B + 4 * qq);
            float* t = tile + (4 * qq) * TS + pl;
            t[0] = v.x; t[TS] = v.y; t[2 * TS] = v.z; t[3 * TS] = v.w;
        }
    }
    __syncthreads();
    float* os = out + n * P * B;
#pragma unroll
    for (int j = 0; j < LP; ++j) {
        const int e = j * TB + threadIdx.x;
        const int b = e / (TP / 4), p4 = e % (TP / 4);
        const int64_t p = pix0 + 4 * p4;
        if (p < P) stp<POL>(os + b * P + p, *reinterpret_cast<const float4*>(tile + b * TS + 4 * p4));
    }
}

// Any-B / any-P transposes: scalar accesses through a dynamic LDS tile [B][TP+1].
__global__ __launch_bounds__(TB) void transpose_generic_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                               int64_t P, int B, int to_bhw) {
    extern __shared__ __attribute__((aligned(16))) float dtile[];
    const int S = TP + 1;
    const int64_t n = blockIdx.y;
    const int64_t pix0 = (int64_t)blockIdx.x * TP;
    const int npix = (int)((P - pix0) < TP ? (P - pix0) : TP);
    const float* is = in + n * P * B;
    float* os = out + n * P * B;
    if (to_bhw) {
        for (int e = threadIdx.x; e < npix * B; e += TB) dtile[(e % B) * S + e / B] = is[pix0 * B + e];
        __syncthreads();
        for (int e = threadIdx.x; e < npix * B; e += TB) { const int b = e / npix, pl = e % npix; os[b * P + pix0 + pl] = dtile[b * S + pl]; }
    } else {
        for (int e = threadIdx.x; e < npix * B; e += TB) { const int b = e / npix, pl = e % npix; dtile[b * S + pl] = is[b * P + pix0 + pl]; }
        __syncthreads();
        for (int e = threadIdx.x; e < npix * B; e += TB) os[pix0 * B + e] = dtile[(e % B) * S + e / B];
    }
}

// ------------------------------------------------------------------------------------------------
// BHW (planar) fast paths: lane -> 4 adjacent pixels, frame column in registers
// ------------------------------------------------------------------------------------------------
template <int BT, int POL>
__global__ __launch_bounds__(TB) void gap_bhw_kernel(const float* z, const float* __restrict__ phi,
                                                     const float* __restrict__ y, const float* __restrict__ phisum,
                                                     float* z1, int64_t P, int phi_shared) {
    const int64_t n = blockIdx.y;
    const int64_t p = ((int64_t)blockIdx.x * TB + threadIdx.x) * 4;
    if (p >= P) return;
    const float* zs = z + n * BT * P + p;
    const float* ps = phi + (phi_shared ? 0 : n * BT * P) + p;
    float4 zv[BT], pv[BT];
#pragma unroll
    for (int b = 0; b < BT; ++b) { zv[b] = ldp<POL>(zs + b * P); pv[b] = ldp<POL>(ps + b * P); }
    const float4 yv = ld4(y + n * P + p);
    const float4 sv = ld4(phisum + (phi_shared ? 0 : n * P) + p);
    float4 fb = zv[0] * pv[0];
#pragma unroll
    for (int b = 1; b < BT; ++b) fb = fb + zv[b] * pv[b];
    const float4 r = (yv - fb) / sv;
    float* os = z1 + n * BT * P + p;
#pragma unroll
    for (int b = 0; b < BT; ++b) stp<POL>(os + b * P, zv[b] + r * pv[b]);
}

template <int POL>
__global__ __launch_bounds__(TB) void forward_bhw_kernel(const float* __restrict__ x, const float* __restrict__ phi,
                                                         float* __restrict__ y, int64_t P, int B, int phi_shared) {
    const int64_t n = blockIdx.y;
    const int64_t p = ((int64_t)blockIdx.x * TB + threadIdx.x) * 4;
    if (p >= P) return;
    const float* xs = x + n * B * P + p;
    const float* ps = phi + (phi_shared ? 0 : n * B * P) + p;
    float4 acc = ldp<POL>(xs) * ldp<POL>(ps);
#pragma unroll 8
    for (int b = 1; b < B; ++b) acc = acc + ldp<POL>(xs + b * P) * ldp<POL>(ps + b * P);
    stp<POL>(y + n * P + p, acc);
}

template <int POL>
__global__ __launch_bounds__(TB) void adjoint_bhw_kernel(const float* __restrict__ y, const float* __restrict__ phi,
                                                         float* __restrict__ x, int64_t P, int B, int phi_shared) {
    const int64_t n = blockIdx.y;
    const int64_t p = ((int64_t)blockIdx.x * TB + threadIdx.x) * 4;
    if (p >= P) return;
    const float4 yv = ld4(y + n * P + p);
    const float* ps = phi + (phi_shared ? 0 : n * B * P) + p;
    float* xs = x + n * B * P + p;
#pragma unroll 8
    for (int b = 0; b < B; ++b) stp<POL>(xs + b * P, yv * ldp<POL>(ps + b * P));
}

template <int POL>
__global__ __launch_bounds__(TB) void phisum_bhw_kernel(const float* __restrict__ phi, float* __restrict__ out, int64_t P, int B) {
    const int64_t n = blockIdx.y;
    const int64_t p = ((int64_t)blockIdx.x * TB + threadIdx.x) * 4;
    if (p >= P) return;
    const float* ps = phi + n * B * P + p;
    float4 acc = ldp<POL>(ps);
#pragma unroll 8
    for (int b = 1; b < B; ++b) acc = acc + ldp<POL>(ps + b * P);
    acc.x = acc.x == 0.0f ? 1.0f : acc.x; acc.y = acc.y == 0.0f ? 1.0f : acc.y;
    acc.z = acc.z == 0.0f ? 1.0f : acc.z; acc.w = acc.w == 0.0f ? 1.0f : acc.w;
    stp<POL>(out + n * P + p, acc);
}

// out = a - b (b may be null), flat float4 stream
template <int POL>
__global__ __launch_bounds__(TB) void sub_flat_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      float* __restrict__ out, int64_t n4, int64_t tail_from, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (i < n4) {
        float4 v = ldp<POL>(a + 4 * i);
        if (b) v = v - ldp<POL>(b + 4 * i);
        stp<POL>(out + 4 * i, v);
    }
    if (i == 0) for (int64_t t = tail_from; t < total; ++t) out[t] = a[t] - (b ? b[t] : 0.0f);
}

// ------------------------------------------------------------------------------------------------
// Generic strided fallbacks: any B, any P, any layout pair; one lane per pixel, frame loop.
// element (n,p,b) lives at n*P*B + p*sp + b*sb with (sp,sb) = (B,1) for HWB, (1,P) for BHW.
// ------------------------------------------------------------------------------------------------
enum { OP_FORWARD = 0, OP_ADJOINT = 1, OP_PHISUM = 2, OP_GAP = 3 };

template <int OP>
__global__ __launch_bounds__(TB) void generic_kernel(const float* a /*x|y|phi|z*/, const float* __restrict__ phi,
                                                     const float* __restrict__ y, const float* __restrict__ phisum, float* out,
                                                     int64_t P, int B, int64_t isp, int64_t isb, int64_t osp, int64_t osb, int phi_shared) {
    const int64_t n = blockIdx.y;
    const int64_t p = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (p >= P) return;
    const int64_t boff = n * P * B, poff = phi_shared ? 0 : boff;
    if (OP == OP_FORWARD) {
        float acc = a[boff + p * isp] * phi[poff + p * isp];
        for (int b = 1; b < B; ++b) acc += a[boff + p * isp + b * isb] * phi[poff + p * isp + b * isb];
        out[n * P + p] = acc;
    } else if (OP == OP_ADJOINT) {
        const float yv = a[n * P + p];
        for (int b = 0; b < B; ++b) out[boff + p * osp + b * osb] = yv * phi[poff + p * isp + b * isb];
    } else if (OP == OP_PHISUM) {
        float acc = a[boff + p * isp];
        for (int b = 1; b < B; ++b) acc += a[boff + p * isp + b * isb];
        out[n * P + p] = acc == 0.0f ? 1.0f : acc;
    } else {
        float fb = a[boff + p * isp] * phi[poff + p * isp];
        for (int b = 1; b < B; ++b) fb += a[boff + p * isp + b * isb] * phi[poff + p * isp + b * isb];
        const float r = (y[n * P + p] - fb) / phisum[(phi_shared ? 0 : n * P) + p];
        for (int b = 0; b < B; ++b) {
            const float zv = a[boff + p * isp + b * isb];
            out[boff + p * osp + b * osb] = zv + r * phi[poff + p * isp + b * isb];
        }
    }
}

static inline void strides(int layout, int64_t P, int64_t B, int64_t& sp, int64_t& sb) {
    if (layout == DEQSCI_LAYOUT_HWB) { sp = B; sb = 1; } else { sp = 1; sb = P; }
}
static inline bool lp_ok(int64_t B) { return B == 4 || B == 8 || B == 16 || B == 32; }
static inline int check_dims(int64_t bsz, int64_t H, int64_t W, int64_t B, int layout) {
    if (bsz <= 0 || H <= 0 || W <= 0 || B <= 0) return DEQSCI_ERR_SHAPE;
    if (bsz > 65535 || B > 4096) return DEQSCI_ERR_UNSUPPORTED;
    if (layout != DEQSCI_LAYOUT_HWB && layout != DEQSCI_LAYOUT_BHW) return DEQSCI_ERR_UNSUPPORTED;
    return 0;
}

#define LP_DISPATCH(B, ...)                          \
    switch ((int)(B)) {                              \
        case 4:  { constexpr int LP = 1; __VA_ARGS__; } break; \
        case 8:  { constexpr int LP = 2; __VA_ARGS__; } break; \
        case 16: { constexpr int LP = 4; __VA_ARGS__; } break; \
        default: { constexpr int LP = 8; __VA_ARGS__; } break; \
    }
#define POL_DISPATCH(pol, ...)                                   \
    switch (pol) {                                               \
        case POL_NTL:  { constexpr int POL = POL_NTL; __VA_ARGS__; } break;  \
        case POL_NTS:  { constexpr int POL = POL_NTS; __VA_ARGS__; } break;  \
        case POL_NTLS: { constexpr int POL = POL_NTLS; __VA_ARGS__; } break; \
        default:       { constexpr int POL = POL_DEFAULT; __VA_ARGS__; } break; \
    }

}  // namespace deqsci

using namespace deqsci;

extern "C" {

const char* deqsci_version(void) { return "deqsci_hip 0.1 (gfx950)"; }

const char* deqsci_error_string(int code) {
    switch (code) {
        case 0: return "success";
        case DEQSCI_ERR_NULL: return "required pointer is NULL";
        case DEQSCI_ERR_SHAPE: return "non-positive or inconsistent sizes";
        case DEQSCI_ERR_ALIGN: return "pointer is not 16-byte aligned";
        case DEQSCI_ERR_UNSUPPORTED: return "unsupported layout / history depth / size";
        default: return code > 0 ? hipGetErrorString(static_cast<hipError_t>(code)) : "unknown deqsci error";
    }
}

int deqsci_sci_forward_f32(const float* x, const float* phi, float* y, int64_t bsz, int64_t H, int64_t W, int64_t B,
                           int layout, int phi_shared, deqsci_stream_t stream) {
    if (!x || !phi || !y) return DEQSCI_ERR_NULL;
    if (int e = check_dims(bsz, H, W, B, layout)) return e;
    if (!aligned16(x) || !aligned16(phi) || !aligned16(y)) return DEQSCI_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t P = H * W;
    const int pol = pick_policy(bsz * P * (8 * B + 4), POL_NTLS);
    if (layout == DEQSCI_LAYOUT_HWB && lp_ok(B)) {
        POL_DISPATCH(pol, LP_DISPATCH(B, hipLaunchKernelGGL((forward_hwb_kernel<LP, POL>), dim3(ceil_div(P * LP, TB * UNR), bsz), dim3(TB), 0, st, x, phi, y, P, phi_shared)));
    } else if (layout == DEQSCI_LAYOUT_BHW && P % 4 == 0) {
        POL_DISPATCH(pol, hipLaunchKernelGGL(forward_bhw_kernel<POL>, dim3(ceil_div(P / 4, TB), bsz), dim3(TB), 0, st, x, phi, y, P, (int)B, phi_shared));
    } else {
        int64_t sp, sb; strides(layout, P, B, sp, sb);
        hipLaunchKernelGGL(generic_kernel<OP_FORWARD>, dim3(ceil_div(P, TB), bsz), dim3(TB), 0, st, x, phi, nullptr, nullptr, y, P, (int)B, sp, sb, sp, sb, phi_shared);
    }
    return launch_status();
}

int deqsci_sci_adjoint_f32(const float* y, const float* phi, float* x, int64_t bsz, int64_t H, int64_t W, int64_t B,
                           int layout, int phi_shared, deqsci_stream_t stream) {
    if (!x || !phi || !y) return DEQSCI_ERR_NULL;
    if (int e = check_dims(bsz, H, W, B, layout)) return e;
    if (!aligned16(x) || !aligned16(phi) || !aligned16(y)) return DEQSCI_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t P = H * W;
    const int pol = pick_policy(bsz * P * (8 * B + 4), POL_NTLS);
    if (layout == DEQSCI_LAYOUT_HWB && lp_ok(B)) {
        POL_DISPATCH(pol, LP_DISPATCH(B, hipLaunchKernelGGL((adjoint_hwb_kernel<LP, POL>), dim3(ceil_div(P * LP, TB * UNR), bsz), dim3(TB), 0, st, y, phi, x, P, phi_shared)));
    } else if (layout == DEQSCI_LAYOUT_BHW && P % 4 == 0) {
        POL_DISPATCH(pol, hipLaunchKernelGGL(adjoint_bhw_kernel<POL>, dim3(ceil_div(P / 4, TB), bsz), dim3(TB), 0, st, y, phi, x, P, (int)B, phi_shared));
    } else {
        int64_t sp, sb; strides(layout, P, B, sp, sb);
        hipLaunchKernelGGL(generic_kernel<OP_ADJOINT>, dim3(ceil_div(P, TB), bsz), dim3(TB), 0, st, y, phi, nullptr, nullptr, x, P, (int)B, sp, sb, sp, sb, phi_shared);
    }
    return launch_status();
}

int deqsci_phi_sum_f32(const float* phi, float* phisum, int64_t nb, int64_t H, int64_t W, int64_t B, int layout,
                       deqsci_stream_t stream) {
    if (!phi || !phisum) return DEQSCI_ERR_NULL;
    if (int e = check_dims(nb, H, W, B, layout)) return e;
    if (!aligned16(phi) || !aligned16(phisum)) return DEQSCI_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t P = H * W;
    const int pol = pick_policy(nb * P * (4 * B + 4), POL_NTLS);
    if (layout == DEQSCI_LAYOUT_HWB && lp_ok(B)) {
        POL_DISPATCH(pol, LP_DISPATCH(B, hipLaunchKernelGGL((phisum_hwb_kernel<LP, POL>), dim3(ceil_div(P * LP, TB * UNR), nb), dim3(TB), 0, st, phi, phisum, P)));
    } else if (layout == DEQSCI_LAYOUT_BHW && P % 4 == 0) {
        POL_DISPATCH(pol, hipLaunchKernelGGL(phisum_bhw_kernel<POL>, dim3(ceil_div(P / 4, TB), nb), dim3(TB), 0, st, phi, phisum, P, (int)B));
    } else {
        int64_t sp, sb; strides(layout, P, B, sp, sb);
        hipLaunchKernelGGL(generic_kernel<OP_PHISUM>, dim3(ceil_div(P, TB), nb), dim3(TB), 0, st, phi, nullptr, nullptr, nullptr, phisum, P, (int)B, sp, sb, sp, sb, 0);
    }
    return launch_status();
}

int deqsci_gap_update_f32(const float* z, const float* phi, const float* y, const float* phisum, float* z1,
                          int64_t bsz, int64_t H, int64_t W, int64_t B, int layout_in, int layout_out, int phi_shared,
                          deqsci_stream_t stream) {
    if (!z || !phi || !y || !phisum || !z1) return DEQSCI_ERR_NULL;
    if (int e = check_dims(bsz, H, W, B, layout_in)) return e;
    if (int e = check_dims(bsz, H, W, B, layout_out)) return e;
    if (!aligned16(z) || !aligned16(phi) || !aligned16(y) || !aligned16(phisum) || !aligned16(z1)) return DEQSCI_ERR_ALIGN;
    if (layout_in != layout_out && z == z1) return DEQSCI_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t P = H * W;
    const int pol = pick_policy(bsz * P * (12 * B + 8), POL_NTLS);
    if (layout_in == DEQSCI_LAYOUT_HWB && layout_out == DEQSCI_LAYOUT_HWB && lp_ok(B)) {
        POL_DISPATCH(pol, LP_DISPATCH(B, hipLaunchKernelGGL((gap_hwb_kernel<LP, POL>), dim3(ceil_div(P * LP, TB * UNR), bsz), dim3(TB), 0, st, z, phi, y, phisum, z1, P, phi_shared)));
    } else if (layout_in == DEQSCI_LAYOUT_BHW && layout_out == DEQSCI_LAYOUT_BHW && P % 4 == 0 && (B == 4 || B == 8 || B == 16)) {
        const dim3 grid(ceil_div(P / 4, TB), bsz);
        if (B == 4) { POL_DISPATCH(pol, hipLaunchKernelGGL((gap_bhw_kernel<4, POL>), grid, dim3(TB), 0, st, z, phi, y, phisum, z1, P, phi_shared)); }
        else if (B == 8) { POL_DISPATCH(pol, hipLaunchKernelGGL((gap_bhw_kernel<8, POL>), grid, dim3(TB), 0, st, z, phi, y, phisum, z1, P, phi_shared)); }
        else { POL_DISPATCH(pol, hipLaunchKernelGGL((gap_bhw_kernel<16, POL>), grid, dim3(TB), 0, st, z, phi, y, phisum, z1, P, phi_shared)); }
    } else if (layout_in == DEQSCI_LAYOUT_HWB && layout_out == DEQSCI_LAYOUT_BHW && lp_ok(B) && P % 4 == 0) {
        POL_DISPATCH(pol, LP_DISPATCH(B, hipLaunchKernelGGL((gap_hwb2bhw_kernel<LP, POL>), dim3(ceil_div(P, TP), bsz), dim3(TB), 0, st, z, phi, y, phisum, z1, P, phi_shared)));
    } else {
        int64_t isp, isb, osp, osb;
        strides(layout_in, P, B, isp, isb);
        strides(layout_out, P, B, osp, osb);
        hipLaunchKernelGGL(generic_kernel<OP_GAP>, dim3(ceil_div(P, TB), bsz), dim3(TB), 0, st, z, phi, y, phisum, z1, P, (int)B, isp, isb, osp, osb, phi_shared);
    }
    return launch_status();
}

int deqsci_transpose_f32(const float* in, float* out, int64_t bsz, int64_t H, int64_t W, int64_t B, int to_layout,
                         deqsci_stream_t stream) {
    if (!in || !out) return DEQSCI_ERR_NULL;
    if (int e = check_dims(bsz, H, W, B, to_layout)) return e;
    if (!aligned16(in) || !aligned16(out)) return DEQSCI_ERR_ALIGN;
    if (in == out) return DEQSCI_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t P = H * W;
    const dim3 grid(ceil_div(P, TP), bsz);
    const int pol = pick_policy(bsz * P * 8 * B, POL_NTLS);
    if (lp_ok(B) && P % 4 == 0) {
        if (to_layout == DEQSCI_LAYOUT_BHW) {
            POL_DISPATCH(pol, LP_DISPATCH(B, hipLaunchKernelGGL((transpose_hwb2bhw_kernel<LP, POL>), grid, dim3(TB), 0, st, in, out, P)));
        } else {
            POL_DISPATCH(pol, LP_DISPATCH(B, hipLaunchKernelGGL((residual_out_bhw2hwb_kernel<LP, POL>), grid, dim3(TB), 0, st, in, (const float*)nullptr, out, P)));
        }
    } else {
        const size_t lds = (size_t)B * (TP + 1) * sizeof(float);
        if (lds > 160 * 1024) return DEQSCI_ERR_UNSUPPORTED;
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(transpose_generic_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(transpose_generic_kernel, grid, dim3(TB), lds, st, in, out, P, (int)B, to_layout == DEQSCI_LAYOUT_BHW ? 1 : 0);
    }
    return launch_status();
}

int deqsci_residual_out_f32(const float* z1, const float* noise, float* out, int64_t bsz, int64_t H, int64_t W, int64_t B,
                            int layout_out, deqsci_stream_t stream) {
    if (!z1 || !noise || !out) return DEQSCI_ERR_NULL;
    if (int e = check_dims(bsz, H, W, B, layout_out)) return e;
    if (!aligned16(z1) || !aligned16(noise) || !aligned16(out)) return DEQSCI_ERR_ALIGN;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t P = H * W;
    const int pol = pick_policy(bsz * P * 12 * B, POL_NTLS);
    if (layout_out == DEQSCI_LAYOUT_BHW) {
        const int64_t total = bsz * P * B, n4 = total / 4;
        POL_DISPATCH(pol, hipLaunchKernelGGL(sub_flat_kernel<POL>, dim3(ceil_div(n4 > 0 ? n4 : 1, TB)), dim3(TB), 0, st, z1, noise, out, n4, n4 * 4, total));
    } else if (lp_ok(B) && P % 4 == 0) {
        if (out == z1 || out == noise) return DEQSCI_ERR_UNSUPPORTED;
        POL_DISPATCH(pol, LP_DISPATCH(B, hipLaunchKernelGGL((residual_out_bhw2hwb_kernel<LP, POL>), dim3(ceil_div(P, TP), bsz), dim3(TB), 0, st, z1, noise, out, P)));
    } else {
        return DEQSCI_ERR_UNSUPPORTED;   // caller composes sub (BHW) + deqsci_transpose_f32
    }
    return launch_status();
}

}  // extern "C"
