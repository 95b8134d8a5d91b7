// Cyclic parallel-order Jacobi eigendecomposition of one complex Hermitian matrix
// held in LDS, executed by ONE 64-lane wavefront that is the whole workgroup
// (synchronised with wave_sync(), so it may also run inside a larger workgroup).
//
// Replaces np.linalg.eigh in pb_bss' ComplexAngularCentralGaussian.from_covariance
// (reached from CACGMMTrainer.fit, reference call site core.py:180-186) and serves
// the pseudo-inverse fallback of the MVDR solve (stable_solve -> lstsq,
// pb_chime5/math/solve.py:95-114).  Jacobi is used because it is branch-free
// across lanes, needs no pivoting, and resolves small eigenvalues to high
// relative accuracy -- the CACGMM floors eigenvalues at 1e-10 of the largest.
//
// Lane mapping: lane = (grp = lane >> 4, pr = lane & 15).  Every lane keeps the
// rotation of pair `pr` in registers (the four groups compute it redundantly),
// and group `grp` applies it to rows / columns grp, grp+4, ...  No integer
// division, no rotation parameters in LDS, two barriers per rotation set.
#pragma once
#include "gss_internal.h"

// A: m x m (m even, <= 32), row-major, Hermitian on entry; on exit diag(A) holds
// the eigenvalues.  V: m x m, on exit column j is the eigenvector of A[j][j].
// If m was padded from an odd size the pad row/column must be zero on entry; it
// then stays decoupled.  Returns the number of sweeps.
// WARM: V holds a unitary W on entry and A = W^H B W (Hermitian): the rotations go on from
// there and V leaves as the eigenvectors of B -- when W is the eigenbasis of a nearby matrix
// (the same class one EM iteration ago) A is nearly diagonal and two or three sweeps do.
__device__ inline int jacobi_eigh_wave(cplx *A, cplx *V, int m, int lane, int max_sweeps,
                                       bool warm = false) {
    const int half = m >> 1;
    const int pr = lane & 15, grp = lane >> 4;
    const bool active = pr < half;
    if (!warm)
        for (int i = grp; i < m; i += 4)
            for (int j = pr; j < m; j += 16) V[i * m + j] = c_make(i == j ? 1.0 : 0.0, 0.0);
    wave_sync();
    int sweep = 0;
    bool last = false;
    for (; sweep < max_sweeps; ++sweep) {
        // Convergence is judged PER PAIR, |a_ij| against the larger of |a_ii|, |a_jj| (the
        // rotation angle it still stands for is |a_ij| / |a_ii - a_jj|): a class covariance
        // estimated from fewer frames than channels has a continuum of eigenvalues from 1e-13
        // to 1 of the largest, the 1e-10 floor of the model cuts through the middle of it, and
        // an off-diagonal mass of 1e-10 of the WHOLE matrix -- where the former global measure
        // let the last sweep start -- still mixes the directions on either side of that cut
        // (posteriors off by 1e-3, wide fuzz seed 202 case 220).  Pairs that lie entirely
        // within the rounding noise of the matrix (both diagonal entries below 1e-13 of the
        // largest, far below that floor) never settle and are left alone.  Once the worst
        // pair is below 1e-10 the (quadratically convergent) next sweep reaches the rounding
        // floor.
        double dmax = 0.0;
        for (int i = lane; i < m; i += 64) dmax = fmax(dmax, fabs(A[i * m + i].x));
        dmax = wave_max(dmax);
        double worst = 0.0;
        for (int i = grp; i < m; i += 4)
            for (int j = pr; j < m; j += 16) {
                if (i == j) continue;
                const double v = c_abs2(A[i * m + j]);
                const double mx = fmax(fabs(A[i * m + i].x), fabs(A[j * m + j].x));
                if (v == 0.0 || mx <= 1e-13 * dmax) continue;
                worst = fmax(worst, v / (mx * mx));
            }
        worst = wave_max(worst);
        if (last || worst <= 1e-30) break;
        if (worst <= 1e-20) last = true;

        for (int step = 0; step < m - 1; ++step) {
            int p = 0, q = 1;
            double c = 1.0;
            cplx s = c_make(0.0, 0.0);
            if (active) {
                if (pr == 0) {
                    p = m - 1;
                    q = step;
                } else {
                    p = step + pr;
                    if (p >= m - 1) p -= m - 1;
                    q = step - pr;
                    if (q < 0) q += m - 1;
                }
                if (p > q) {
                    const int t = p;
                    p = q;
                    q = t;
                }
                const double a = A[p * m + p].x, d = A[q * m + q].x;
                const cplx b = A[p * m + q];
                const double babs = hypot(b.x, b.y);
                if (babs > 0.0) {
                    const double tau = (d - a) / (2.0 * babs);
                    const double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
                    c = 1.0 / sqrt(1.0 + t * t);
                    const double sn = t * c;
                    s = c_make(sn * (b.x / babs), sn * (b.y / babs));
                }
            }
            // column update  A <- A J,  V <- V J   (rows grp, grp + 4, ...)
            if (active) {
                for (int i = grp; i < m; i += 4) {
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
            }
            wave_sync();
            // row update  A <- J^H A   (columns grp, grp + 4, ...)
            if (active) {
                for (int j = grp; j < m; j += 4) {
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
            }
            wave_sync();
        }
    }
    return sweep;
}
