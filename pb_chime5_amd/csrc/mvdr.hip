// Mask post-processing and mask-based MVDR (Souden) beamforming with blind
// analytic normalisation:
//   enhance_observation mask handling            core.py:537-554
//   beamform_mvdr_souden_from_masks / _Beamformer speech_enhancement/beamforming_wrapper.py:11-124
//   -> pb_bss.extraction.beamformer.{get_power_spectral_density_matrix,
//      get_mvdr_vector_souden(eps=1e-10), blind_analytic_normalization,
//      apply_beamforming_vector}
#include "dense_wave.h"
#include "gss_internal.h"
#include "jacobi.h"

namespace {

constexpr int PSD_TILE = 64;

// gamma (F,K,T) -> target (F,T), distortion (F,T); Python slice semantics for the
// zeroed context frames (masks[:, :start] = 0; if end > 0: masks[:, -end:] = 0).
__global__ void masks_kernel(const double *__restrict__ gamma, int F, int K, int64_t T,
                             int target, int64_t zero_lo_end, int64_t zero_hi_begin,
                             double *__restrict__ mx, double *__restrict__ mn) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)F * T) return;
    const int f = idx / T;
    const int64_t t = idx - (int64_t)f * T;
    double x = 0.0, n = 0.0;
    if (t >= zero_lo_end && t < zero_hi_begin) {
        const double *g = gamma + (int64_t)f * K * T + t;
        x = g[(int64_t)target * T];
        for (int k = 0; k < K; ++k)
            if (k != target) n += g[(int64_t)k * T];
    }
    mx[idx] = x;
    mn[idx] = n;
}

// Pack the two masks into the (F, 2, T) weight layout of the shared covariance
// kernel (cacgmm.hip: wcov_kernel) and sum them over time.  grid (F), block 256.
__global__ __launch_bounds__(256) void mask_pack_kernel(const double *__restrict__ mx,
                                                        const double *__restrict__ mn, int64_t T,
                                                        double *__restrict__ W2,
                                                        double *__restrict__ msum) {
    __shared__ double red[8];
    const int f = blockIdx.x, tid = threadIdx.x;
    double sx = 0.0, sn = 0.0;
    for (int64_t t = tid; t < T; t += blockDim.x) {
        const double a = mx[(int64_t)f * T + t], b = mn[(int64_t)f * T + t];
        W2[((int64_t)f * 2) * T + t] = a;
        W2[((int64_t)f * 2 + 1) * T + t] = b;
        sx += a;
        sn += b;
    }
    sx = wave_sum(sx);
    sn = wave_sum(sn);
    if ((tid & 63) == 0) {
        red[tid >> 6] = sx;
        red[4 + (tid >> 6)] = sn;
    }
    __syncthreads();
    if (tid == 0) {
        msum[f * 2] = (red[0] + red[1]) + (red[2] + red[3]);
        msum[f * 2 + 1] = (red[4] + red[5]) + (red[6] + red[7]);
    }
}

__device__ __forceinline__ cplx c_div(cplx a, cplx b) {
    // Smith's algorithm (what NumPy uses for complex division)
    if (fabs(b.x) >= fabs(b.y)) {
        if (b.x == 0.0 && b.y == 0.0) return c_make(a.x / fabs(b.x), a.y / fabs(b.x));
        const double r = b.y / b.x, den = b.x + b.y * r;
        return c_make((a.x + a.y * r) / den, (a.y - a.x * r) / den);
    }
    const double r = b.x / b.y, den = b.x * r + b.y;
    return c_make((a.x * r + a.y) / den, (a.y * r - a.x) / den);
}

// Per frequency (one workgroup of MVDR_NT threads): Phi_X, Phi_N from the partial sums;
// Psi = solve(Phi_N, Phi_X) by LU with partial pivoting, pseudo-inverse (lstsq) fallback on an
// exactly singular Phi_N; W = Psi / max(Re tr Psi, eps); per-reference-channel SNR terms.
// Every element of every step is computed by exactly the expressions a single wave used until
// round 5 (the rank-1 update of an LU step, the products of the SNR terms are element-wise), so
// the result does not depend on the thread count: four waves take the 23 x 47 element update of
// the first step in 5 trips instead of 17 -- the kernel was one wave's latency chain per
// frequency (120 us for ~10 000 instructions), not work.  Pivot search, back substitution (one
// right-hand side per lane) and the Jacobi fallback stay with wave 0.
#ifndef GSS_MVDR_NT
#define GSS_MVDR_NT 256        // (tools/build_variant.sh NAME -DGSS_MVDR_NT=64: the same bits from one wave)
#endif
constexpr int MVDR_NT = GSS_MVDR_NT;
__global__ __launch_bounds__(MVDR_NT) void mvdr_solve_kernel(
    const cplx *__restrict__ part, const double *__restrict__ msum, int nch, int D, double eps,
    cplx *__restrict__ Phi /* (F,2,D,D) */, cplx *__restrict__ W /* (F,D,D) */,
    cplx *__restrict__ snr /* (F,D,2) */) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = MVDR_NT;
    const int m = D + (D & 1);
    const int NE = tri_count(D);
    const int W2 = 2 * D;
    cplx *aug = reinterpret_cast<cplx *>(smem);   // D x 2D : [Phi_N | Phi_X] -> [U | Z] -> Psi
    cplx *JA = aug + D * W2;                       // m * m
    cplx *JV = JA + m * m;                         // m * m
    // flags live in the dynamic region too: a static __shared__ in front of it
    // would break its 16-byte alignment
    int *flags = reinterpret_cast<int *>(JV + m * m);
    int &s_piv = flags[0];
    int &s_singular = flags[1];
    double &s_dentr = *reinterpret_cast<double *>(flags + 2);
    const int f = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const bool wave0 = tid < 64;                   // (D <= 32: one wave holds a column)

    const double sx = msum[f * 2], sn = msum[f * 2 + 1];
    const double dx = fmax(sx, 1e-10), dn = fmax(sn, 1e-10);
    cplx *PhiX = Phi + (int64_t)f * 2 * D * D;
    cplx *PhiN = PhiX + D * D;
    for (int e = tid; e < NE; e += NT) {
        // invert the packed index
        int d1 = 0, rem = e;
        while (rem >= D - d1) {
            rem -= D - d1;
            ++d1;
        }
        const int d2 = d1 + rem;
        cplx vx = c_make(0.0, 0.0), vn = c_make(0.0, 0.0);
        for (int c = 0; c < nch; ++c) {
            const cplx *pp = part + ((int64_t)f * nch + c) * 2 * NE;
            vx = c_add(vx, pp[e]);
            vn = c_add(vn, pp[NE + e]);
        }
        vx = c_make(vx.x / dx, vx.y / dx);
        vn = c_make(vn.x / dn, vn.y / dn);
        if (d1 == d2) {
            vx.y = 0.0;
            vn.y = 0.0;
        }
        PhiX[d1 * D + d2] = vx;
        PhiX[d2 * D + d1] = c_conj(vx);
        PhiN[d1 * D + d2] = vn;
        PhiN[d2 * D + d1] = c_conj(vn);
        aug[d1 * W2 + d2] = vn;
        aug[d2 * W2 + d1] = c_conj(vn);
        aug[d1 * W2 + D + d2] = vx;
        aug[d2 * W2 + D + d1] = c_conj(vx);
    }
    if (tid == 0) s_singular = 0;
    __syncthreads();

    // ---- LU with partial pivoting (pivot by |re| + |im| like LAPACK izamax)
    for (int j = 0; j < D; ++j) {
        if (wave0) {
            double best = -1.0;
            int bi = j;
            if (lane >= j && lane < D) {
                const cplx v = aug[lane * W2 + j];
                best = fabs(v.x) + fabs(v.y);
                bi = lane;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const double ob = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ob > best || (ob == best && oi < bi)) {
                    best = ob;
                    bi = oi;
                }
            }
            if (lane == 0) {
                s_piv = bi;
                if (!(best > 0.0)) s_singular = 1;   // zero or NaN pivot
            }
        }
        __syncthreads();
        if (s_singular) break;
        const int p = s_piv;
        if (p != j) {
            for (int col = tid; col < W2; col += NT) {
                const cplx t = aug[j * W2 + col];
                aug[j * W2 + col] = aug[p * W2 + col];
                aug[p * W2 + col] = t;
            }
        }
        __syncthreads();
        const cplx piv = aug[j * W2 + j];
        const int rows = D - j - 1, cols = W2 - j - 1;
        // multipliers first (column j), then the rank-1 update
        for (int i = tid; i < rows; i += NT) {
            const int r = j + 1 + i;
            aug[r * W2 + j] = c_div(aug[r * W2 + j], piv);
        }
        __syncthreads();
        for (int it = tid; it < rows * cols; it += NT) {
            const int i = it / cols, cidx = it - i * cols;
            const int r = j + 1 + i, col = j + 1 + cidx;
            const cplx l = aug[r * W2 + j], u = aug[j * W2 + col];
            cplx v = aug[r * W2 + col];
            v.x -= l.x * u.x - l.y * u.y;
            v.y -= l.x * u.y + l.y * u.x;
            aug[r * W2 + col] = v;
        }
        __syncthreads();
    }
    const bool singular = s_singular != 0;
    if (!singular) {
        // back substitution U Psi = Z, one right-hand side per lane
        if (tid < D) {
            const int col = D + tid;
            for (int j = D - 1; j >= 0; --j) {
                cplx v = aug[j * W2 + col];
                for (int k = j + 1; k < D; ++k) {
                    const cplx u = aug[j * W2 + k], x = aug[k * W2 + col];
                    v.x -= u.x * x.x - u.y * x.y;
                    v.y -= u.x * x.y + u.y * x.x;
                }
                aug[j * W2 + col] = c_div(v, aug[j * W2 + j]);
            }
        }
        __syncthreads();
    } else {
        // np.linalg.lstsq(Phi_N, Phi_X): minimum-norm solution via the Hermitian
        // eigendecomposition; singular values below eps * D * max are dropped.
        for (int idx = tid; idx < m * m; idx += NT) {
            const int i = idx / m, jx = idx - i * m;
            JA[idx] = (i < D && jx < D) ? PhiN[i * D + jx] : c_make(0.0, 0.0);
        }
        __syncthreads();
        if (wave0) {
            jacobi_eigh_wave(JA, JV, m, lane, 20);
            double lmax = 0.0;
            for (int i = lane; i < D; i += 64) lmax = fmax(lmax, fabs(JA[i * m + i].x));
            lmax = wave_max(lmax);
            if (lane == 0) s_dentr = 2.220446049250313e-16 * (double)D * lmax;
        }
        __syncthreads();
        const double cut = s_dentr;
        // Psi = V diag(1/l) V^H Phi_X   (two small products through `aug`)
        // step 1: tmp = V^H Phi_X  -> aug[:, 0:D]
        for (int it = tid; it < D * D; it += NT) {
            const int j = it / D, col = it - j * D;
            cplx v = c_make(0.0, 0.0);
            for (int i = 0; i < D; ++i) c_cfma(v, JV[i * m + j], PhiX[i * D + col]);
            const double l = JA[j * m + j].x;
            const double il = fabs(l) > cut ? 1.0 / l : 0.0;
            aug[j * W2 + col] = c_scale(v, il);
        }
        __syncthreads();
        for (int it = tid; it < D * D; it += NT) {
            const int i = it / D, col = it - i * D;
            cplx v = c_make(0.0, 0.0);
            for (int j = 0; j < D; ++j) c_fma(v, JV[i * m + j], aug[j * W2 + col]);
            aug[i * W2 + D + col] = v;
        }
        __syncthreads();
    }
    // Psi = aug[:, D:2D].  W = Psi / max(Re tr Psi, eps)
    if (wave0) {
        double tr = 0.0;
        for (int i = lane; i < D; i += 64) tr += aug[i * W2 + D + i].x;
        tr = wave_sum(tr);
        if (lane == 0) s_dentr = fmax(tr, eps);
    }
    __syncthreads();
    const double dentr = s_dentr;
    cplx *Wf = W + (int64_t)f * D * D;
    // (every thread reads its elements of Psi before anyone overwrites the left half: the two
    // halves are disjoint, W goes to the left one)
    for (int it = tid; it < D * D; it += NT) {
        const int i = it / D, col = it - i * D;
        const cplx v = aug[i * W2 + D + col];
        const cplx wv = c_make(v.x / dentr, v.y / dentr);
        aug[i * W2 + col] = wv;   // keep W in the left half for the SNR terms
        Wf[it] = wv;
    }
    __syncthreads();
    // SNR terms per reference channel r: w_r^H Phi_X w_r and w_r^H Phi_N w_r.  The products
    // T_X = Phi_X W (-> right half of aug) and T_N = Phi_N W (-> JA) are spread over the whole
    // workgroup; the sums over e and then over d run in the same order as one lane per r would
    // take them.
    // (reads: W in the left half of aug, Phi_X / Phi_N in global memory; writes: the right half
    // and JA -- disjoint, no barrier inside)
    for (int it = tid; it < D * D; it += NT) {
        const int d = it / D, r = it - d * D;
        cplx tx = c_make(0.0, 0.0), tn = c_make(0.0, 0.0);
        for (int e = 0; e < D; ++e) {
            const cplx we = aug[e * W2 + r];
            c_fma(tx, PhiX[d * D + e], we);
            c_fma(tn, PhiN[d * D + e], we);
        }
        aug[d * W2 + D + r] = tx;
        JA[d * m + r] = tn;
    }
    __syncthreads();
    if (tid < D) {
        const int r = tid;
        cplx num = c_make(0.0, 0.0), den = c_make(0.0, 0.0);
        for (int d = 0; d < D; ++d) {
            const cplx wd = aug[d * W2 + r];
            c_cfma(num, wd, aug[d * W2 + D + r]);
            c_cfma(den, wd, JA[d * m + r]);
        }
        snr[((int64_t)f * D + r) * 2] = num;
        snr[((int64_t)f * D + r) * 2 + 1] = den;
    }
}

// GEV beamformer (beamforming_wrapper.py:77-89 -> pb_bss get_gev_vector): principal
// generalised eigenvector of (Phi_X, Phi_N), normalised like the generalised
// Hermitian eigensolvers do (w^H Phi_N w = 1; the phase is arbitrary, as upstream).
//   Phi_N = L L^H,  C = L^-1 Phi_X L^-H,  C u = lambda_max u,  w = L^-H u
// One wave per frequency; w is written to column 0 of W so that mvdr_apply (with
// ref = 0) does the BAN and the filtering.  A Phi_N that is not positive definite makes
// scipy.linalg.eigh -- and with it the reference -- raise LinAlgError: the frequency is
// recorded (ref[0] = -2, ref[1] = the lowest such frequency; the host presets 0 and a
// large number), mvdr_apply fills Xhat with NaN and the host raises.
__global__ __launch_bounds__(64) void gev_solve_kernel(const cplx *__restrict__ part,
                                                       const double *__restrict__ msum, int nch,
                                                       int D, cplx *__restrict__ Phi,
                                                       cplx *__restrict__ W,
                                                       int32_t *__restrict__ ref) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m = D + (D & 1);
    const int NE = tri_count(D);
    cplx *Ln = reinterpret_cast<cplx *>(smem);    // m * m : Phi_N -> L -> L^-1
    cplx *JA = Ln + m * m;                         // m * m : C
    cplx *JV = JA + m * m;                         // m * m : eigenvectors
    cplx *Tm = JV + m * m;                         // m * m : Linv Phi_X
    const int f = blockIdx.x, lane = threadIdx.x;

    const double dx = fmax(msum[f * 2], 1e-10), dn = fmax(msum[f * 2 + 1], 1e-10);
    cplx *PhiX = Phi + (int64_t)f * 2 * D * D;
    cplx *PhiN = PhiX + D * D;
    for (int idx = lane; idx < m * m; idx += 64) {
        Ln[idx] = c_make(0.0, 0.0);
        JA[idx] = c_make(0.0, 0.0);
    }
    __syncthreads();
    for (int e = lane; e < NE; e += 64) {
        int d1 = 0, rem = e;
        while (rem >= D - d1) {
            rem -= D - d1;
            ++d1;
        }
        const int d2 = d1 + rem;
        cplx vx = c_make(0.0, 0.0), vn = c_make(0.0, 0.0);
        for (int c = 0; c < nch; ++c) {
            const cplx *pp = part + ((int64_t)f * nch + c) * 2 * NE;
            vx = c_add(vx, pp[e]);
            vn = c_add(vn, pp[NE + e]);
        }
        vx = c_make(vx.x / dx, vx.y / dx);
        vn = c_make(vn.x / dn, vn.y / dn);
        if (d1 == d2) {
            vx.y = 0.0;
            vn.y = 0.0;
        }
        PhiX[d1 * D + d2] = vx;
        PhiX[d2 * D + d1] = c_conj(vx);
        PhiN[d1 * D + d2] = vn;
        PhiN[d2 * D + d1] = c_conj(vn);
        Ln[d2 * m + d1] = c_conj(vn);          // lower triangle
        if (d1 == d2) Ln[d1 * m + d1] = vn;
    }
    __syncthreads();
    cplx *Wf = W + (int64_t)f * D * D;
    if (!cholesky_lower_wave(Ln, D, m, lane)) {
        for (int idx = lane; idx < D * D; idx += 64) Wf[idx] = c_make(NAN, NAN);
        if (lane == 0) {
            atomicMin(&ref[0], -2);
            atomicMin(&ref[1], f);
        }
        return;
    }
    invert_lower_wave(Ln, D, m, lane);          // Ln = L^-1 (lower)
    // Tm = Linv Phi_X
    for (int idx = lane; idx < D * D; idx += 64) {
        const int i = idx / D, j = idx - i * D;
        cplx v = c_make(0.0, 0.0);
        for (int k = 0; k <= i; ++k) c_fma(v, Ln[i * m + k], PhiX[k * D + j]);
        Tm[i * m + j] = v;
    }
    __syncthreads();
    // C = Tm Linv^H  (Hermitian)
    for (int idx = lane; idx < D * D; idx += 64) {
        const int i = idx / D, j = idx - i * D;
        cplx v = c_make(0.0, 0.0);
        for (int k = 0; k <= j; ++k) c_fmac(v, Tm[i * m + k], Ln[j * m + k]);
        JA[i * m + j] = v;
    }
    __syncthreads();
    // exact Hermitian symmetry for the Jacobi sweeps
    for (int idx = lane; idx < D * D; idx += 64) {
        const int i = idx / D, j = idx - i * D;
        if (i < j) {
            const cplx a = JA[i * m + j], b = JA[j * m + i];
            const cplx h = c_make(0.5 * (a.x + b.x), 0.5 * (a.y - b.y));
            Tm[i * m + j] = h;
            Tm[j * m + i] = c_conj(h);
        } else if (i == j) {
            Tm[i * m + i] = c_make(JA[i * m + i].x, 0.0);
        }
    }
    __syncthreads();
    for (int idx = lane; idx < m * m; idx += 64) {
        const int i = idx / m, j = idx - i * m;
        JA[idx] = (i < D && j < D) ? Tm[idx] : c_make(0.0, 0.0);
    }
    __syncthreads();
    jacobi_eigh_wave(JA, JV, m, lane, 20);
    // largest eigenvalue (first index on ties)
    double best = lane < D ? JA[lane * m + lane].x : -INFINITY;
    int bi = lane < D ? lane : 1 << 30;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ob > best || (ob == best && oi < bi)) {
            best = ob;
            bi = oi;
        }
    }
    // w = Linv^H u :  w[d] = sum_{i >= d} conj(Linv[i][d]) u[i]
    for (int idx = lane; idx < D * D; idx += 64) {
        const int d = idx / D, col = idx - d * D;
        cplx v = c_make(0.0, 0.0);
        if (col == 0)
            for (int i = d; i < D; ++i) c_cfma(v, Ln[i * m + d], JV[i * m + bi]);
        Wf[idx] = v;
    }
}

// get_optimal_reference_channel: one reference channel for all frequencies.
// forced >= 0: the caller names the channel (pb_bss get_mvdr_vector_souden(ref_channel=...)).
// A non-finite SNR makes the reference abort the utterance (`assert np.all(np.isfinite(SNR))`):
// the channel becomes -1, mvdr_apply fills Xhat with NaN and the host raises.
constexpr int MVDR_REF_NT = 1024, MVDR_REF_CHUNK = 64;     // frequencies staged per round
__global__ __launch_bounds__(MVDR_REF_NT) void mvdr_ref_kernel(const cplx *__restrict__ snr, int F,
                                                              int D, double eps, int forced,
                                                              int32_t *__restrict__ ref) {
    // snr (F, D, 2) is one contiguous run: the whole workgroup copies MVDR_REF_CHUNK frequencies
    // of it to LDS at a time (coalesced, every load independent), then lane r of wave 0 adds its
    // channel's terms in ascending frequency -- the order one lane walking global memory took
    // until round 5 (64 dependent round trips to L2: 37 us for 25 000 numbers).
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx *buf = reinterpret_cast<cplx *>(smem);            // MVDR_REF_CHUNK * D * 2
    const int tid = threadIdx.x, lane = tid & 63;
    if (forced >= 0) {
        if (tid == 0) ref[0] = forced;
        return;
    }
    double val = -INFINITY;
    bool isnan_ = false;
    bool bad = false;
    cplx num = c_make(0.0, 0.0), den = c_make(0.0, 0.0);
    for (int f0 = 0; f0 < F; f0 += MVDR_REF_CHUNK) {
        const int nf = min(MVDR_REF_CHUNK, F - f0);
        const int total = nf * D * 2;
        __syncthreads();
        for (int i = tid; i < total; i += MVDR_REF_NT) buf[i] = snr[(int64_t)f0 * D * 2 + i];
        __syncthreads();
        if (tid < D) {
            for (int f = 0; f < nf; ++f) {
                num = c_add(num, buf[(f * D + tid) * 2]);
                den = c_add(den, buf[(f * D + tid) * 2 + 1]);
            }
        }
    }
    if (tid >= 64) return;
    if (lane < D) {
    // np.maximum(den, eps) on complex: lexicographic (real, then imag)
        if (!(den.x > eps || (den.x == eps && den.y > 0.0))) den = c_make(eps, 0.0);
        const cplx q = c_div(num, den);
        val = q.x;
        isnan_ = val != val;
        bad = !(isfinite(q.x) && isfinite(q.y));
    }
    const bool any_bad = __any(bad);
    // np.argmax: first maximum, NaN counts as maximum
    double best = val;
    int bi = lane < D ? lane : 1 << 30;
    bool bn = isnan_;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        const bool on = __shfl_xor((int)bn, o, 64) != 0;
        bool take;
        if (on != bn) take = on;
        else if (on) take = oi < bi;
        else take = ob > best || (ob == best && oi < bi);
        if (take) {
            best = ob;
            bi = oi;
            bn = on;
        }
    }
    if (lane == 0) ref[0] = any_bad ? -1 : bi;
}

// w = W[:, ref] (optionally BAN-normalised), Xhat[t][f] = w^H y_t.  grid (chunks, F)
__global__ __launch_bounds__(256) void mvdr_apply_kernel(
    const cplx *__restrict__ Y, const cplx *__restrict__ W, const cplx *__restrict__ Phi,
    const int32_t *__restrict__ ref, int F, int64_t T, int D, int ban, int chunk_frames,
    cplx *__restrict__ Xhat, int32_t *__restrict__ ref_out, int32_t *__restrict__ status) {
    __shared__ cplx w[GSS_MAX_CHANNELS];
    __shared__ cplx t1[GSS_MAX_CHANNELS];
    __shared__ cplx t2[GSS_MAX_CHANNELS];
    __shared__ double s_norm;
    const int f = blockIdx.y, tid = threadIdx.x;
    const int r = ref[0];
    if (blockIdx.x == 0 && f == 0 && tid == 0) {
        // -1: non-finite SNR (MVDR); -2 - f: Phi_N of frequency f not positive definite (GEV)
        const int code = r == -2 ? -2 - ref[1] : r;
        if (ref_out) ref_out[0] = code;
        if (status) __hip_atomic_store(status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const int64_t c0 = (int64_t)blockIdx.x * chunk_frames;
    const int64_t c1 = c0 + chunk_frames < T ? c0 + chunk_frames : T;
    if (r < 0) {   // the reference raises (see above), nothing meaningful to write
        const double qnan = __longlong_as_double(0x7ff8000000000000LL);
        for (int64_t t = c0 + tid; t < c1; t += blockDim.x) Xhat[t * F + f] = c_make(qnan, qnan);
        return;
    }
    if (tid < D) w[tid] = W[((int64_t)f * D + tid) * D + r];
    __syncthreads();
    if (ban) {
        const cplx *PhiN = Phi + ((int64_t)f * 2 + 1) * D * D;
        if (tid < D) {
            cplx v = c_make(0.0, 0.0);
            for (int e = 0; e < D; ++e) c_fma(v, PhiN[tid * D + e], w[e]);
            t1[tid] = v;   // Phi_N w
            cplx u = c_make(0.0, 0.0);   // (w^H Phi_N)_a = sum_d conj(w_d) Phi_N[d][a], a = tid
            for (int d = 0; d < D; ++d) c_cfma(u, w[d], PhiN[d * D + tid]);
            t2[tid] = u;
        }
        __syncthreads();
        if (tid == 0) {
            // nominator = w^H Phi_N Phi_N w ; denominator = w^H Phi_N w
            cplx nom = c_make(0.0, 0.0), den = c_make(0.0, 0.0);
            for (int a = 0; a < D; ++a) {
                c_fma(nom, t2[a], t1[a]);
                c_cfma(den, w[a], t1[a]);
            }
            const double n = sqrt(hypot(nom.x, nom.y));   // |sqrt(z)|
            const double dd = hypot(den.x, den.y);
            s_norm = n / dd;   // eps = 0 upstream: 0/0 -> NaN like the reference
        }
        __syncthreads();
        if (tid < D) w[tid] = c_scale(w[tid], s_norm);
        __syncthreads();
    }
    const cplx *Yf = Y + (int64_t)f * T * D;
    for (int64_t t = c0 + tid; t < c1; t += blockDim.x) {
        const cplx *y = Yf + t * D;
        cplx v = c_make(0.0, 0.0);
        for (int d = 0; d < D; ++d) c_cfma(v, w[d], y[d]);
        Xhat[t * F + f] = v;
    }
}

int psd_chunks(int F, int64_t T, int *chunk_frames) {
    int64_t tiles = (T + PSD_TILE - 1) / PSD_TILE;
    int64_t want = (2048 + F - 1) / F;
    int64_t tpc = (tiles + want - 1) / want;
    if (tpc < 1) tpc = 1;
    *chunk_frames = (int)(tpc * PSD_TILE);
    return (int)((tiles + tpc - 1) / tpc);
}

}  // namespace

int masks_from_posteriors_run(gss_ctx *ctx, const double *gamma, int F, int K, int64_t T,
                              int target, int drop, int64_t sf, int64_t ef, double *mx,
                              double *mn) {
    int64_t lo_end = 0, hi_begin = T;
    if (drop) {
        lo_end = sf >= 0 ? (sf < T ? sf : T) : (T + sf > 0 ? T + sf : 0);
        if (ef > 0) hi_begin = T - ef > 0 ? T - ef : 0;
    }
    GSS_PROF(ctx, "masks");
    const int64_t total = (int64_t)F * T;
    hipLaunchKernelGGL(masks_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       ctx->stream, gamma, F, K, T, target, lo_end, hi_begin, mx, mn);
    GSS_LAUNCH_CHECK(ctx, "masks_kernel");
    return GSS_OK;
}

size_t mvdr_workspace_bytes(int F, int64_t T, int D) {
    const size_t NE = tri_count(D);
    int cf;
    const int nch = psd_chunks(F, T, &cf);
    size_t b = 0;
    b += align_up(sizeof(cplx) * (size_t)F * nch * 2 * NE);
    b += align_up(sizeof(double) * (size_t)F * 2);
    b += align_up(sizeof(double) * (size_t)F * 2 * T);
    b += align_up(sizeof(cplx) * (size_t)F * 2 * D * D);
    b += align_up(sizeof(cplx) * (size_t)F * D * D);
    b += align_up(sizeof(cplx) * (size_t)F * D * 2);
    b += 256;
    return b + 4096;
}

int mvdr_run(gss_ctx *ctx, const cplx *Y, int F, int64_t T, int D, const double *mx,
             const double *mn, int ban, cplx *Xhat, int32_t *ref_channel, int gev,
             int forced_ref) {
    const int NE = tri_count(D);
    int cf;
    const int nch = psd_chunks(F, T, &cf);
    cplx *part = arena_alloc_t<cplx>(ctx, (size_t)F * nch * 2 * NE);
    double *msum = arena_alloc_t<double>(ctx, (size_t)F * 2);
    double *W2 = arena_alloc_t<double>(ctx, (size_t)F * 2 * T);
    cplx *Phi = arena_alloc_t<cplx>(ctx, (size_t)F * 2 * D * D);
    cplx *W = arena_alloc_t<cplx>(ctx, (size_t)F * D * D);
    cplx *snr = arena_alloc_t<cplx>(ctx, (size_t)F * D * 2);
    int32_t *ref = arena_alloc_t<int32_t>(ctx, 4);
    GSS_REQUIRE(ctx, part && msum && W2 && Phi && W && snr && ref, GSS_ERR_NOMEM,
                "mvdr workspace");
    {
        GSS_PROF(ctx, "psd");
        hipLaunchKernelGGL(mask_pack_kernel, dim3(F), dim3(256), 0, ctx->stream, mx, mn, T, W2,
                           msum);
        GSS_LAUNCH_CHECK(ctx, "mask_pack_kernel");
        GSS_TRY(psd_partials_run(ctx, Y, F, T, D, W2, nch, cf, part));
    }
    if (gev) {
        GSS_PROF(ctx, "gev_solve");
        const int m = D + (D & 1);
        const size_t lds = (sizeof(cplx) * 4 * (size_t)m * m + 15) / 16 * 16;
        if (lds > 64 * 1024)
            GSS_HIP_CHECK(ctx, hipFuncSetAttribute(
                                   reinterpret_cast<const void *>(gev_solve_kernel),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        GSS_HIP_CHECK(ctx, hipMemsetAsync(ref, 0, sizeof(int32_t), ctx->stream));
        GSS_HIP_CHECK(ctx, hipMemsetAsync(ref + 1, 0x7f, sizeof(int32_t), ctx->stream));
        hipLaunchKernelGGL(gev_solve_kernel, dim3(F), dim3(64), lds, ctx->stream, part, msum, nch,
                           D, Phi, W, ref);
        GSS_LAUNCH_CHECK(ctx, "gev_solve_kernel");
    } else {
        {
            GSS_PROF(ctx, "mvdr_solve");
            const int m = D + (D & 1);
            const size_t lds = (sizeof(cplx) * ((size_t)D * 2 * D + 2 * (size_t)m * m) +
                                32 + 15) / 16 * 16;
            if (lds > 64 * 1024)
                GSS_HIP_CHECK(ctx, hipFuncSetAttribute(
                                       reinterpret_cast<const void *>(mvdr_solve_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            hipLaunchKernelGGL(mvdr_solve_kernel, dim3(F), dim3(MVDR_NT), lds, ctx->stream, part, msum, nch,
                               D, 1e-10, Phi, W, snr);
            GSS_LAUNCH_CHECK(ctx, "mvdr_solve_kernel");
        }
        {
            GSS_PROF(ctx, "mvdr_ref");
            hipLaunchKernelGGL(mvdr_ref_kernel, dim3(1), dim3(MVDR_REF_NT),
                               sizeof(cplx) * MVDR_REF_CHUNK * (size_t)D * 2, ctx->stream, snr, F, D, 1e-10,
                               forced_ref, ref);
            GSS_LAUNCH_CHECK(ctx, "mvdr_ref_kernel");
        }
    }
    {
        GSS_PROF(ctx, "mvdr_apply");
        const int chunk = 256;
        hipLaunchKernelGGL(mvdr_apply_kernel, dim3((unsigned)((T + chunk - 1) / chunk), F),
                           dim3(256), 0, ctx->stream, Y, W, Phi, ref, F, T, D, ban, chunk, Xhat,
                           ref_channel, ctx->status_dev);
        GSS_LAUNCH_CHECK(ctx, "mvdr_apply_kernel");
    }
    return GSS_OK;
}
