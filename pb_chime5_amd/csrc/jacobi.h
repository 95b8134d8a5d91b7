// Cyclic parallel-order Jacobi eigendecomposition of one complex Hermitian matrix
// held in LDS, executed by ONE 64-lane wavefront that is the whole workgroup
// (so __syncthreads() is a single-wave barrier).
//
// Replaces np.linalg.eigh in pb_bss' ComplexAngularCentralGaussian.from_covariance
// (reached from CACGMMTrainer.fit, reference call site core.py:180-186) and serves
// the pseudo-inverse fallback of the MVDR solve (stable_solve -> lstsq,
// pb_chime5/math/solve.py:95-114).  Jacobi is used because it is branch-free
// across lanes, needs no pivoting, and resolves small eigenvalues to high
// relative accuracy -- the CACGMM floors eigenvalues at 1e-10 of the largest.
#pragma once
#include "gss_internal.h"

struct JacobiScratch {
    double c[16];
    cplx s[16];
    int p[16], q[16];
};

// A: m x m (m even, <= 32), row-major, Hermitian on entry; on exit diag(A) holds
// the eigenvalues.  V: m x m, on exit column j is the eigenvector of A[j][j].
// If m was padded from an odd size the pad row/column must be zero on entry; it
// then stays decoupled.
__device__ inline int jacobi_eigh_wave(cplx *A, cplx *V, JacobiScratch *js, int m, int lane,
                                       int max_sweeps) {
    const int half = m >> 1;
    for (int idx = lane; idx < m * m; idx += 64) {
        const int i = idx / m, j = idx - i * m;
        V[idx] = c_make(i == j ? 1.0 : 0.0, 0.0);
    }
    __syncthreads();
    int sweep = 0;
    for (; sweep < max_sweeps; ++sweep) {
        // convergence: off-diagonal mass against the total
        double off = 0.0, dia = 0.0;
        for (int idx = lane; idx < m * m; idx += 64) {
            const int i = idx / m, j = idx - i * m;
            const double v = c_abs2(A[idx]);
            if (i == j) dia += v; else off += v;
        }
        off = wave_sum(off);
        dia = wave_sum(dia);
        if (off <= 1e-31 * (dia + off)) break;

        for (int step = 0; step < m - 1; ++step) {
            if (lane < half) {
                int p, q;
                if (lane == 0) {
                    p = m - 1;
                    q = step;
                } else {
                    p = (step + lane) % (m - 1);
                    q = (step - lane + (m - 1)) % (m - 1);
                }
                if (p > q) {
                    const int t = p;
                    p = q;
                    q = t;
                }
                const double a = A[p * m + p].x, d = A[q * m + q].x;
                const cplx b = A[p * m + q];
                const double babs = hypot(b.x, b.y);
                double c = 1.0;
                cplx s = c_make(0.0, 0.0);
                if (babs > 0.0) {
                    const double tau = (d - a) / (2.0 * babs);
                    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    c = 1.0 / sqrt(1.0 + t * t);
                    const double sn = t * c;
                    s = c_make(sn * (b.x / babs), sn * (b.y / babs));
                }
                js->c[lane] = c;
                js->s[lane] = s;
                js->p[lane] = p;
                js->q[lane] = q;
            }
            __syncthreads();
            // column update  A <- A J,  V <- V J   (item = (row i, pair))
            for (int it = lane; it < m * half; it += 64) {
                const int i = it / half, pr = it - i * half;
                const int p = js->p[pr], q = js->q[pr];
                const double c = js->c[pr];
                const cplx s = js->s[pr];
                {
                    const cplx ap = A[i * m + p], aq = A[i * m + q];
                    // A'_ip = A_ip c - A_iq conj(s) ;  A'_iq = A_ip s + A_iq c
                    cplx np_ = c_scale(ap, c), nq_ = c_scale(aq, c);
                    np_.x -= aq.x * s.x + aq.y * s.y;
                    np_.y -= aq.y * s.x - aq.x * s.y;
                    nq_.x += ap.x * s.x - ap.y * s.y;
                    nq_.y += ap.x * s.y + ap.y * s.x;
                    A[i * m + p] = np_;
                    A[i * m + q] = nq_;
                }
                {
                    const cplx vp = V[i * m + p], vq = V[i * m + q];
                    cplx np_ = c_scale(vp, c), nq_ = c_scale(vq, c);
                    np_.x -= vq.x * s.x + vq.y * s.y;
                    np_.y -= vq.y * s.x - vq.x * s.y;
                    nq_.x += vp.x * s.x - vp.y * s.y;
                    nq_.y += vp.x * s.y + vp.y * s.x;
                    V[i * m + p] = np_;
                    V[i * m + q] = nq_;
                }
            }
            __syncthreads();
            // row update  A <- J^H A   (item = (pair, column j))
            for (int it = lane; it < m * half; it += 64) {
                const int pr = it / m, j = it - pr * m;
                const int p = js->p[pr], q = js->q[pr];
                const double c = js->c[pr];
                const cplx s = js->s[pr];
                const cplx ap = A[p * m + j], aq = A[q * m + j];
                // A'_pj = c A_pj - s A_qj ;  A'_qj = conj(s) A_pj + c A_qj
                cplx np_ = c_scale(ap, c), nq_ = c_scale(aq, c);
                np_.x -= s.x * aq.x - s.y * aq.y;
                np_.y -= s.x * aq.y + s.y * aq.x;
                nq_.x += s.x * ap.x + s.y * ap.y;
                nq_.y += s.x * ap.y - s.y * ap.x;
                A[p * m + j] = np_;
                A[q * m + j] = nq_;
            }
            __syncthreads();
        }
    }
    return sweep;
}
