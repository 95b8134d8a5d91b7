// Single-wave dense linear algebra on one small matrix in LDS (leading dimension ld),
// shared by the CACGMM model update (cacgmm.hip) and the GEV beamformer (mvdr.hip).
// The calling workgroup is ONE 64-lane wavefront, so __syncthreads() is a
// single-wave barrier.
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
        __syncthreads();
        const int i = j + 1 + lane;
        if (i < n) A[i * ld + j] = c_scale(A[i * ld + j], dinv);
        if (lane == 0) A[j * ld + j] = c_make(d, 0.0);
        __syncthreads();
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
    __syncthreads();
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
        __syncthreads();
        if (i < n) A[i * ld + j] = c_make(-ajj * x.x, -ajj * x.y);
        if (lane == 0) A[j * ld + j] = c_make(ajj, 0.0);
        __syncthreads();
    }
}

