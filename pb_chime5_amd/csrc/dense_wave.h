// Single-wave dense linear algebra on one small matrix in LDS (leading dimension ld),
// shared by the CACGMM model update (cacgmm.hip) and the GEV beamformer (mvdr.hip).
// The caller is ONE 64-lane wavefront (synchronised with wave_sync(), so the wave may
// be part of a larger workgroup).
#pragma once
#include "gss_internal.h"

// Single-wave linear algebra on one D x D matrix in LDS (leading dimension ld).
// Lane grid for 2-D updates: (ri, ci) = (lane >> 3, lane & 7).

// In-place lower Cholesky A = L L^H; only the lower triangle is referenced and
// written.  Returns false (wave-uniform) on a non-positive pivot.
__device__ inline bool cholesky_lower_wave(cplx *A, int n, int ld, int lane) {
    const int ri = lane >> 3, ci = lane & 7;
    for (int j = 0; j < n; ++j) {
        const double ajj = A[j * ld + j].x;
        if (!(ajj > 0.0) || !isfinite(ajj)) return false;
        const double d = sqrt(ajj), dinv = 1.0 / d;
        wave_sync();
        const int i = j + 1 + lane;
        if (i < n) A[i * ld + j] = c_scale(A[i * ld + j], dinv);
        if (lane == 0) A[j * ld + j] = c_make(d, 0.0);
        wave_sync();
        // trailing update of the lower triangle: A[i][k] -= L[i][j] conj(L[k][j]), j < k <= i
        const int r = n - j - 1;
        for (int ii = ri; ii < r; ii += 8) {
            const cplx li = A[(j + 1 + ii) * ld + j];
            for (int kk = ci; kk <= ii; kk += 8) {
                const cplx lk = A[(j + 1 + kk) * ld + j];
                cplx v = A[(j + 1 + ii) * ld + j + 1 + kk];
                v.x -= li.x * lk.x + li.y * lk.y;
                v.y -= li.y * lk.x - li.x * lk.y;
                A[(j + 1 + ii) * ld + j + 1 + kk] = v;
            }
        }
    }
    wave_sync();
    return true;
}

// In-place inverse of the lower-triangular L (n <= 64): column j of L^-1 is
//   Linv[j][j] = 1 / L[j][j],   Linv[i][j] = -Linv[j][j] * sum_{k=j+1..i} Linv[i][k] L[k][j]
// processed for j = n-1 .. 0 so that the trailing block is already inverted.
__device__ inline void invert_lower_wave(cplx *A, int n, int ld, int lane) {
    for (int j = n - 1; j >= 0; --j) {
        const double ajj = 1.0 / A[j * ld + j].x;
        const int i = j + 1 + lane;
        cplx x = c_make(0.0, 0.0);
        if (i < n)
            for (int k = j + 1; k <= i; ++k) c_fma(x, A[i * ld + k], A[k * ld + j]);
        wave_sync();
        if (i < n) A[i * ld + j] = c_make(-ajj * x.x, -ajj * x.y);
        if (lane == 0) A[j * ld + j] = c_make(ajj, 0.0);
        wave_sync();
    }
}


// Cholesky factor AND its inverse in one register-resident sweep (used on the WPE
// diagonal blocks by 256 threads, G = 16, and on the CACGMM class covariances by one
// wave, G = 8).  The n x n Hermitian matrix (n <= G * NR) lives in registers, NR x NR
// entries per thread: entry (a, b) of thread (ty, tx) is (i, k) = (ty + G a, tx + G b).
// On entry reg holds the upper triangle (k >= i) of the matrix and zeros elsewhere.
// One right-looking pass of n steps builds U (A = U^H U, upper part) and
// W = U^-H (strictly lower part) together: per step the owners of row j publish it
// through LDS, ONE barrier, then every thread reads the pivot, its NR row and NR
// column factors and updates its entries.  The dependent chain is n x (LDS round trip
// + rsqrt + NR^2 complex FMAs) instead of two LDS-bound triangular sweeps.
// On exit the LDS array M (leading dimension ld) holds the UNSCALED rows -- row j is
// U[j][k] / dinv[j] for k > j, W[j][k] / dinv[j] for k < j, the pivot a_jj at k = j --
// and dinv[j] = 1 / U[j][j] (0 where the pivot was not positive and finite); callers
// scale when they consume.  Returns false (uniformly) if any pivot failed.
// The steps are grouped by the block row jb = j / G they fall into (an unrolled loop), so
// that "row / column block before, in, or after the pivot's block" is known at compile
// time: row blocks above the pivot's are finished (no work), W entries take no column
// beyond it, and only the pivot's own blocks need per-lane selects.
// DIAG_RING (the WPE diagonal blocks): M is a ring of TWO rows (row j lives in slot j & 1;
// a thread can be at most one step ahead of the slowest one, so two slots are enough) and
// only the G x G diagonal blocks of W are formed -- the panel kernels use nothing else of
// it (16-blocked substitution).  The caller takes U and W from `reg`, which ends up holding
// what the full-matrix form leaves in M.
template <int G, int NR, bool ONE_WAVE = false, bool DIAG_RING = false>
__device__ __forceinline__ bool chol_inverse_sweep(cplx (&reg)[NR][NR], int n, cplx *M, int ld,
                                          double *dinv, int tx, int ty) {
    bool ok = true;
#pragma unroll
    for (int jb = 0; jb < NR; ++jb) {
        for (int jj = 0; jj < G; ++jj) {
            const int j = G * jb + jj;
            if (j >= n) break;                      // uniform
            // the owners of row j (ty == jj, register row jb) publish it
            cplx *Mj = M + (DIAG_RING ? (j & 1) : j) * ld;
            if (ty == jj) {
#pragma unroll
                for (int b = 0; b < NR; ++b) Mj[tx + G * b] = reg[jb][b];
            }
            if (ONE_WAVE) wave_sync();
            else __syncthreads();
            // all LDS reads of the step are issued together, unconditionally
            const double ajj = Mj[j].x;
            cplx ru[NR], rv[NR];
#pragma unroll
            for (int a = jb; a < NR; ++a) ru[a] = Mj[ty + G * a];
#pragma unroll
            for (int b = 0; b < NR; ++b) rv[b] = Mj[tx + G * b];
            double di = 0.0;
            if (ajj > 0.0 && isfinite(ajj)) di = rsqrt(ajj);
            else ok = false;
            if (tx == 0 && ty == 0) dinv[j] = di;
            // conj(U[j][i]) for the rows below the pivot, 0 elsewhere
            cplx u[NR];
#pragma unroll
            for (int a = jb; a < NR; ++a) {
                const double s = a > jb ? di : (ty > jj ? di : 0.0);
                u[a] = c_make(ru[a].x * s, -ru[a].y * s);
            }
            // k > j: U[j][k];  k < j: W[j][k];  k == j: W[j][j] = 1 / U[j][j]
            cplx v[NR];
#pragma unroll
            for (int b = 0; b < NR; ++b) v[b] = c_scale(rv[b], di);
            if (tx == jj) v[jb] = c_make(di, 0.0);
            // W entries only take columns k <= j
            const cplx vw_jb = tx <= jj ? v[jb] : c_make(0.0, 0.0);
#pragma unroll
            for (int a = jb; a < NR; ++a)
#pragma unroll
                for (int b = 0; b < NR; ++b) {
                    // entry (a, b): upper (U) when b > a, W when b < a, per thread on a == b
                    if (b < a && (b > jb || DIAG_RING)) continue;
                    cplx vv;
                    if (b > a) vv = v[b];
                    else if (b < a) vv = b < jb ? v[b] : vw_jb;
                    else if (b == jb) vv = tx >= ty ? v[b] : vw_jb;
                    else vv = tx >= ty ? v[b] : c_make(0.0, 0.0);      // b > jb
                    reg[a][b].x = fma(-u[a].x, vv.x, reg[a][b].x);
                    reg[a][b].x = fma(u[a].y, vv.y, reg[a][b].x);
                    reg[a][b].y = fma(-u[a].x, vv.y, reg[a][b].y);
                    reg[a][b].y = fma(-u[a].y, vv.x, reg[a][b].y);
                }
        }
    }
    if (ONE_WAVE) wave_sync();
    else __syncthreads();
    return ok;
}
