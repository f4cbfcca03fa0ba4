/* Plain-C restatement of the elementwise / small-linear-algebra part of the DEQ-SCI hot path.
 * TEST INFRASTRUCTURE - NOT PRODUCT CODE: only tests/ (and __graft_entry__.build(), which merely
 * compiles it) may touch this file; libdeqsci_hip.so never links or calls it.
 * Pinned by tests/test_oracle_c.py against the golden vectors generated from the reference
 * (tests/golden/ops.npz, anderson_toy.npz).  Layout is the reference's (bsz,H,W,B), fp32.
 *
 *   orc_sci_forward   utils/cg_utils.py:85-90      y = sum_b x*Phi   (left-to-right fp32 sum)
 *   orc_sci_adjoint   utils/cg_utils.py:124-129    x = y*Phi
 *   orc_phi_sum       training/sci_equilibrium_training.py:162-163
 *   orc_gap_update    solvers/equilibrium_solvers_yaping.py:399-400
 *   orc_anderson_alpha  solvers/new_equilibrium_utils_yaping.py:177-180  (Gram + bordered solve)
 *   orc_anderson_mix    ibid. :182
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

void orc_sci_forward(const float* x, const float* phi, float* y, int64_t npix, int64_t B) {
    for (int64_t p = 0; p < npix; ++p) {
        float acc = x[p * B] * phi[p * B];
        for (int64_t b = 1; b < B; ++b) acc += x[p * B + b] * phi[p * B + b];
        y[p] = acc;
    }
}

void orc_sci_adjoint(const float* y, const float* phi, float* x, int64_t npix, int64_t B) {
    for (int64_t p = 0; p < npix; ++p)
        for (int64_t b = 0; b < B; ++b) x[p * B + b] = y[p] * phi[p * B + b];
}

void orc_phi_sum(const float* phi, float* out, int64_t npix, int64_t B) {
    for (int64_t p = 0; p < npix; ++p) {
        float acc = phi[p * B];
        for (int64_t b = 1; b < B; ++b) acc += phi[p * B + b];
        out[p] = acc == 0.0f ? 1.0f : acc;
    }
}

void orc_gap_update(const float* z, const float* phi, const float* y, const float* phisum, float* z1, int64_t npix, int64_t B) {
    for (int64_t p = 0; p < npix; ++p) {
        float fb = z[p * B] * phi[p * B];
        for (int64_t b = 1; b < B; ++b) fb += z[p * B + b] * phi[p * B + b];
        const float r = (y[p] - fb) / phisum[p];
        for (int64_t b = 0; b < B; ++b) z1[p * B + b] = z[p * B + b] + r * phi[p * B + b];
    }
}

/* alpha (n) from F,X (n rows of length N): G = F - X, H = [[0,1],[1,GG^T + lam I]], solve H [nu;alpha] = e0
 * by LU with partial pivoting.  Accumulation in double: the checker's job is to be closer to the exact
 * answer than either implementation under test.  Returns 0, or 1 if singular. */
int orc_anderson_alpha(const float* F, const float* X, int n, int64_t N, double lam, float* alpha) {
    const int nn = n + 1;
    double A[10][11];
    if (n < 1 || n > 8) return 2;
    memset(A, 0, sizeof A);
    for (int i = 0; i < n; ++i)
        for (int j = i; j < n; ++j) {
            double s = 0.0;
            for (int64_t e = 0; e < N; ++e)
                s += (double)(F[i * N + e] - X[i * N + e]) * (double)(F[j * N + e] - X[j * N + e]);
            A[i + 1][j + 1] = A[j + 1][i + 1] = s;
        }
    for (int i = 1; i < nn; ++i) { A[i][i] += lam; A[0][i] = 1.0; A[i][0] = 1.0; }
    A[0][nn] = 1.0;
    for (int k = 0; k < nn; ++k) {
        int piv = k;
        for (int i = k + 1; i < nn; ++i) if (fabs(A[i][k]) > fabs(A[piv][k])) piv = i;
        if (A[piv][k] == 0.0) return 1;
        if (piv != k) for (int j = 0; j <= nn; ++j) { double t = A[k][j]; A[k][j] = A[piv][j]; A[piv][j] = t; }
        for (int i = k + 1; i < nn; ++i) {
            const double f = A[i][k] / A[k][k];
            for (int j = k; j <= nn; ++j) A[i][j] -= f * A[k][j];
        }
    }
    for (int i = nn - 1; i >= 0; --i) {
        double v = A[i][nn];
        for (int j = i + 1; j < nn; ++j) v -= A[i][j] * A[j][nn];
        A[i][nn] = v / A[i][i];
    }
    for (int i = 0; i < n; ++i) alpha[i] = (float)A[i + 1][nn];
    return 0;
}

void orc_anderson_mix(const float* F, const float* X, const float* alpha, int n, int64_t N, float beta, float* out) {
    for (int64_t e = 0; e < N; ++e) {
        float sf = 0.0f, sx = 0.0f;
        for (int i = 0; i < n; ++i) { sf += alpha[i] * F[i * N + e]; sx += alpha[i] * X[i * N + e]; }
        out[e] = beta * sf + (1.0f - beta) * sx;
    }
}
