// Guided CACGMM:  GSS.__call__ (/root/reference/pb_chime5/core.py:154-214) ->
// pb_bss CACGMMTrainer.fit / CACGMM.predict, all 513 frequencies at once.
//
// One EM iteration = two launches:
//   em_step  grid (chunks, F): a single fused pass over the frames of the chunk
//            E-step   q_kt = | y^H B_k^-1 y |,  log p = -D ln q - ln det,  softmax
//                     with weights pi_k and the activity mask, clip to [eps, 1-eps]
//            M-step   partial sums of  (gamma_kt / q_kt) y y^H  and of gamma_kt
//            (the posteriors and quadratic forms never leave the chip)
//   em_eig   grid (K, F): reduce the partial sums, B_k = D * sum / sum(gamma),
//            Hermitian eigendecomposition (Jacobi, LDS), eigenvalues / max, floor
//            1e-10, then  B_k^-1 = V diag(1/lambda) V^H,  ln det,  pi_k.
//
// Hermitian structure: with P_de(t) = y_d conj(y_e) (class independent, d <= e)
//    q_kt   = sum_{d<=e}  Re(Mq_k,de) Re(P_de) + Im(Mq_k,de) Im(P_de)
//    B_k,de = sum_t w_kt P_de(t)
// where Mq holds B_k^-1 with the off-diagonals doubled.  That is 4 real FMAs per
// (entry, class, frame) instead of 12 for the dense form.
#include "gss_internal.h"
#include <cstdlib>

#include "jacobi.h"

namespace {

constexpr int EM_TILE = 64;          // frames per tile (one per lane of a wave)
constexpr int EM_TS = EM_TILE + 1;   // padded LDS row stride (complex elements)
constexpr int EM_SLOTS = 3;          // ceil(528 / 256): accumulator slots per thread

enum { MODE_FIRST = 0, MODE_EM = 1, MODE_PREDICT = 2 };

struct EmArgs {
    const cplx *Y;          // (F,T,D)
    const uint8_t *act;     // (K,act_stride), first T columns used
    int64_t act_stride;
    const cplx *Mq;         // (F,NE,K)
    const double *logdet;   // (F,K)
    const double *pi;       // (F,K)
    cplx *Bp;               // (F,NCH,K,NE)
    double *Sg;             // (F,NCH,K)
    double *gamma;          // (F,K,T)   MODE_PREDICT only
    int64_t T;
    int D, NE, nch, chunk_frames;
    int masked;             // multiply the activity mask into the posteriors
    double aff_eps;         // clip, 0 = none
};

template <int K, int MODE>
__global__ __launch_bounds__(256) void em_step_kernel(EmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D = a.D, NE = a.NE;
    cplx *ys = reinterpret_cast<cplx *>(smem);                 // D * EM_TS
    cplx *Ms = ys + D * EM_TS;                                 // NE * K
    double *qpart = reinterpret_cast<double *>(Ms + NE * K);   // 4 * K * EM_TILE
    double *wk = qpart + 4 * K * EM_TILE;                      // K * EM_TILE
    double *ldet = wk + K * EM_TILE;                           // K
    double *pis = ldet + K;                                    // K
    unsigned char *ed = reinterpret_cast<unsigned char *>(pis + K);   // 2 * NE

    const int f = blockIdx.y, chunk = blockIdx.x;
    const int tid = threadIdx.x;
    const int tl = tid & 63, g = tid >> 6;
    const int64_t T = a.T;
    const int64_t c0 = (int64_t)chunk * a.chunk_frames;
    const int64_t c1 = c0 + a.chunk_frames < T ? c0 + a.chunk_frames : T;
    const cplx *Yf = a.Y + (int64_t)f * T * D;

    for (int d1 = tid; d1 < D; d1 += blockDim.x)
        for (int d2 = d1; d2 < D; ++d2) {
            const int e = tri_index(d1, d2, D);
            ed[2 * e] = (unsigned char)d1;
            ed[2 * e + 1] = (unsigned char)d2;
        }
    if (MODE != MODE_FIRST) {
        const cplx *Mf = a.Mq + (int64_t)f * NE * K;
        for (int i = tid; i < NE * K; i += blockDim.x) Ms[i] = Mf[i];
        if (tid < K) {
            ldet[tid] = a.logdet[f * K + tid];
            pis[tid] = a.pi[f * K + tid];
        }
    }

    cplx acc[EM_SLOTS][K];
    double sg[K];
#pragma unroll
    for (int s = 0; s < EM_SLOTS; ++s)
#pragma unroll
        for (int k = 0; k < K; ++k) acc[s][k] = c_make(0.0, 0.0);
#pragma unroll
    for (int k = 0; k < K; ++k) sg[k] = 0.0;

    const int neq = (NE + 3) / 4;

    for (int64_t t0 = c0; t0 < c1; t0 += EM_TILE) {
        __syncthreads();
        // ---- load + unit-normalise the tile: ys[d][tl] = y_t[d] / ||y_t||
        // thread (tl, g) loads channels d = g, g+4, ... of frame t0 + tl
        {
            const int64_t t = t0 + tl;
            double nrm = 0.0;
            cplx v[8];   // D <= 32 -> at most 8 channels per group
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = g + 4 * j;
                v[j] = c_make(0.0, 0.0);
                if (d < D && t < c1) v[j] = Yf[t * D + d];
                nrm += c_abs2(v[j]);
            }
            qpart[g * EM_TILE + tl] = nrm;
            __syncthreads();
            nrm = qpart[tl] + qpart[EM_TILE + tl] + qpart[2 * EM_TILE + tl] +
                  qpart[3 * EM_TILE + tl];
            nrm = sqrt(nrm);
            if (nrm == 0.0) nrm = GSS_TINY;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int d = g + 4 * j;
                if (d < D) ys[d * EM_TS + tl] = c_make(v[j].x / nrm, v[j].y / nrm);
            }
        }
        __syncthreads();

        // ---- E-step quadratic forms: partial sums over a quarter of the entries
        if (MODE != MODE_FIRST) {
            double q[K];
#pragma unroll
            for (int k = 0; k < K; ++k) q[k] = 0.0;
            const int e0 = g * neq, e1 = min(e0 + neq, NE);
            for (int e = e0; e < e1; ++e) {
                const cplx y1 = ys[ed[2 * e] * EM_TS + tl];
                const cplx y2 = ys[ed[2 * e + 1] * EM_TS + tl];
                const double pr = y1.x * y2.x + y1.y * y2.y;
                const double pim = y1.y * y2.x - y1.x * y2.y;
                const cplx *mrow = Ms + e * K;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const cplx m = mrow[k];
                    q[k] = fma(m.x, pr, q[k]);
                    q[k] = fma(m.y, pim, q[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < K; ++k) qpart[(g * K + k) * EM_TILE + tl] = q[k];
        }
        __syncthreads();

        // ---- posteriors (first wave: one frame per lane)
        if (g == 0) {
            const int64_t t = t0 + tl;
            const bool valid = t < c1;
            double gam[K], qq[K];
            if (MODE == MODE_FIRST) {
                // GSS initialisation (core.py:156-160): where(act == 0, 1e-10, act) / sum_k
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const double v = (valid && a.act[(int64_t)k * a.act_stride + t]) ? 1.0 : 1e-10;
                    gam[k] = v;
                    s += v;
                }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    gam[k] = gam[k] / s;
                    qq[k] = 1.0;
                }
            } else {
                double lp[K], mx = -INFINITY;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    double q = qpart[k * EM_TILE + tl] + qpart[(K + k) * EM_TILE + tl] +
                               qpart[(2 * K + k) * EM_TILE + tl] +
                               qpart[(3 * K + k) * EM_TILE + tl];
                    q = fmax(fabs(q), GSS_TINY);
                    qq[k] = q;
                    lp[k] = -(double)D * log(q) - ldet[k];
                    mx = fmax(mx, lp[k]);
                }
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    double v = exp(lp[k] - mx) * pis[k];
                    if (a.masked) v *= (valid && a.act[(int64_t)k * a.act_stride + t]) ? 1.0 : 0.0;
                    gam[k] = v;
                    s += v;
                }
                s = fmax(s, GSS_TINY);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    double v = gam[k] / s;
                    if (a.aff_eps != 0.0) v = fmin(fmax(v, a.aff_eps), 1.0 - a.aff_eps);
                    gam[k] = v;
                }
            }
            if (MODE == MODE_PREDICT) {
                if (valid) {
#pragma unroll
                    for (int k = 0; k < K; ++k)
                        a.gamma[((int64_t)f * K + k) * T + t] = gam[k];
                }
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const double gk = valid ? gam[k] : 0.0;
                    sg[k] += gk;
                    wk[k * EM_TILE + tl] = gk / fmax(qq[k], 10.0 * GSS_TINY);
                }
            }
        }
        if (MODE == MODE_PREDICT) continue;
        __syncthreads();

        // ---- M-step accumulation: thread owns entries tid, tid+256, tid+512
        const int nfr = (int)min((int64_t)EM_TILE, c1 - t0);
#pragma unroll
        for (int s = 0; s < EM_SLOTS; ++s) {
            const int e = tid + 256 * s;
            if (e < NE) {
                const cplx *r1 = ys + ed[2 * e] * EM_TS;
                const cplx *r2 = ys + ed[2 * e + 1] * EM_TS;
                for (int j = 0; j < nfr; ++j) {
                    const cplx y1 = r1[j], y2 = r2[j];
                    const double pr = y1.x * y2.x + y1.y * y2.y;
                    const double pim = y1.y * y2.x - y1.x * y2.y;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const double w = wk[k * EM_TILE + j];
                        acc[s][k].x = fma(w, pr, acc[s][k].x);
                        acc[s][k].y = fma(w, pim, acc[s][k].y);
                    }
                }
            }
        }
    }

    if (MODE == MODE_PREDICT) return;
    cplx *Bp = a.Bp + ((int64_t)f * a.nch + chunk) * K * NE;
#pragma unroll
    for (int s = 0; s < EM_SLOTS; ++s) {
        const int e = tid + 256 * s;
        if (e < NE) {
#pragma unroll
            for (int k = 0; k < K; ++k) Bp[k * NE + e] = acc[s][k];
        }
    }
    if (g == 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double s = wave_sum(sg[k]);
            if (tl == 0) a.Sg[((int64_t)f * a.nch + chunk) * K + k] = s;
        }
    }
}

// In-place lower Cholesky of the Hermitian positive definite n x n matrix A (LDS,
// leading dimension ld) by one wave; returns false (wave-uniform) on a
// non-positive pivot.  Only the lower triangle is referenced / written.
__device__ inline bool cholesky_lower_wave(cplx *A, int n, int ld, int lane) {
    for (int j = 0; j < n; ++j) {
        const double ajj = A[j * ld + j].x;
        if (!(ajj > 0.0) || !isfinite(ajj)) return false;
        const double d = sqrt(ajj), dinv = 1.0 / d;
        __syncthreads();
        for (int i = j + lane; i < n; i += 64) {
            if (i == j) A[j * ld + j] = c_make(d, 0.0);
            else A[i * ld + j] = c_scale(A[i * ld + j], dinv);
        }
        __syncthreads();
        // trailing update of the lower triangle: A[i][k] -= L[i][j] conj(L[k][j]), j < k <= i
        const int r = n - j - 1;
        for (int it = lane; it < r * r; it += 64) {
            const int ii = it / r, kk = it - ii * r;
            if (kk > ii) continue;
            const int i = j + 1 + ii, k = j + 1 + kk;
            const cplx li = A[i * ld + j], lk = A[k * ld + j];
            cplx v = A[i * ld + k];
            v.x -= li.x * lk.x + li.y * lk.y;
            v.y -= li.y * lk.x - li.x * lk.y;
            A[i * ld + k] = v;
        }
        __syncthreads();
    }
    return true;
}

// grid (K, F), block 64 (one wave per class matrix).
//
// The reference keeps (V, lambda) with lambda <- max(lambda / lambda_max, 1e-10) and
// evaluates q = y^H V diag(1/lambda) V^H y and ln det = sum ln lambda.  A per-class
// scale of lambda cancels between -D ln q and -ln det, and rescales the next
// covariance by a constant that the next normalisation removes again.  So whenever
// NO eigenvalue is floored -- lambda_min > 1e-10 lambda_max, certified here by a
// successful Cholesky factorisation of B - 1e-10 tr(B) I (tr B >= lambda_max) --
// B^-1 and ln det B from a Cholesky factorisation give the same posteriors, and
// the eigendecomposition is only run for matrices that fail the certificate.
__global__ __launch_bounds__(64) void em_eig_kernel(const cplx *__restrict__ Bp,
                                                    const double *__restrict__ Sg, int nch, int D,
                                                    int K, int64_t T, double eig_floor,
                                                    int force_eigh, cplx *__restrict__ Mq,
                                                    double *__restrict__ logdet,
                                                    double *__restrict__ pi,
                                                    int *__restrict__ slow_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int m = D + (D & 1);
    const int NE = tri_count(D);
    cplx *A = reinterpret_cast<cplx *>(smem);   // m * m
    cplx *V = A + m * m;                         // m * m
    double *lam = reinterpret_cast<double *>(V + m * m);   // m
    const int k = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;

    double sg = 0.0;
    for (int c = 0; c < nch; ++c) sg += Sg[((int64_t)f * nch + c) * K + k];
    const double den = fmax(sg, GSS_TINY);

    for (int idx = lane; idx < m * m; idx += 64) A[idx] = c_make(0.0, 0.0);
    __syncthreads();
    double tr = 0.0;
    for (int e = lane; e < NE; e += 64) {
        int d1 = 0, rem = e;
        while (rem >= D - d1) {
            rem -= D - d1;
            ++d1;
        }
        const int d2 = d1 + rem;
        cplx v = c_make(0.0, 0.0);
        for (int c = 0; c < nch; ++c)
            v = c_add(v, Bp[(((int64_t)f * nch + c) * K + k) * NE + e]);
        v.x = ((double)D * v.x) / den;
        v.y = ((double)D * v.y) / den;
        if (d1 == d2) {
            v.y = 0.0;
            tr += v.x;
        }
        A[d1 * m + d2] = v;
        A[d2 * m + d1] = c_conj(v);
    }
    tr = wave_sum(tr);
    __syncthreads();

    bool fast = false;
    if (!force_eigh && tr > 0.0 && isfinite(tr)) {
        // certificate: B - floor * tr(B) * I positive definite  (work in V)
        const double shift = eig_floor * tr;
        for (int idx = lane; idx < D * D; idx += 64) {
            const int i = idx / D, j = idx - i * D;
            cplx v = A[i * m + j];
            if (i == j) v.x -= shift;
            V[i * m + j] = v;
        }
        __syncthreads();
        fast = cholesky_lower_wave(V, D, m, lane);
        __syncthreads();
    }
    if (fast) {
        // factor B itself (it is positive definite a fortiori)
        for (int idx = lane; idx < D * D; idx += 64) {
            const int i = idx / D, j = idx - i * D;
            V[i * m + j] = A[i * m + j];
        }
        __syncthreads();
        fast = cholesky_lower_wave(V, D, m, lane);
        __syncthreads();
    }
    double ld = 0.0;
    if (fast) {
        // ln det B = 2 sum ln L_ii ;  Linv = L^-1 (lower), one column per lane, into A
        for (int i = lane; i < D; i += 64) ld += 2.0 * log(V[i * m + i].x);
        ld = wave_sum(ld);
        __syncthreads();
        if (lane < D) {
            const int c = lane;
            for (int i = 0; i < c; ++i) A[i * m + c] = c_make(0.0, 0.0);
            A[c * m + c] = c_make(1.0 / V[c * m + c].x, 0.0);
            for (int i = c + 1; i < D; ++i) {
                cplx acc = c_make(0.0, 0.0);
                for (int j = c; j < i; ++j) c_fma(acc, V[i * m + j], A[j * m + c]);
                const double dinv = 1.0 / V[i * m + i].x;
                A[i * m + c] = c_make(-acc.x * dinv, -acc.y * dinv);
            }
        }
        __syncthreads();
        // B^-1 = Linv^H Linv :  (d1,d2) = sum_{j >= d2} conj(Linv[j][d1]) Linv[j][d2]
        for (int e = lane; e < NE; e += 64) {
            int d1 = 0, rem = e;
            while (rem >= D - d1) {
                rem -= D - d1;
                ++d1;
            }
            const int d2 = d1 + rem;
            cplx v = c_make(0.0, 0.0);
            for (int j = d2; j < D; ++j) c_cfma(v, A[j * m + d1], A[j * m + d2]);
            if (d1 == d2) {
                v.y = 0.0;
            } else {
                v.x *= 2.0;
                v.y *= 2.0;
            }
            Mq[((int64_t)f * NE + e) * K + k] = v;
        }
    } else {
        if (lane == 0 && slow_count) atomicAdd(slow_count, 1);
        jacobi_eigh_wave(A, V, m, lane, 20);
        double lmax = -INFINITY;
        for (int i = lane; i < D; i += 64) lmax = fmax(lmax, A[i * m + i].x);
        lmax = wave_max(lmax);
        for (int i = lane; i < D; i += 64) {
            double l = A[i * m + i].x / fmax(lmax, GSS_TINY);
            l = fmax(l, eig_floor);
            lam[i] = 1.0 / l;
            ld += log(l);
        }
        ld = wave_sum(ld);
        __syncthreads();
        for (int e = lane; e < NE; e += 64) {
            int d1 = 0, rem = e;
            while (rem >= D - d1) {
                rem -= D - d1;
                ++d1;
            }
            const int d2 = d1 + rem;
            cplx v = c_make(0.0, 0.0);
            for (int j = 0; j < D; ++j) {
                const cplx a = V[d1 * m + j], b = V[d2 * m + j];
                const double il = lam[j];
                v.x += il * (a.x * b.x + a.y * b.y);
                v.y += il * (a.y * b.x - a.x * b.y);
            }
            if (d1 == d2) {
                v.y = 0.0;
            } else {
                v.x *= 2.0;
                v.y *= 2.0;
            }
            Mq[((int64_t)f * NE + e) * K + k] = v;
        }
    }
    if (lane == 0) {
        logdet[f * K + k] = ld;
        pi[f * K + k] = sg / (double)T;
    }
}

size_t em_step_lds(int D, int K) {
    const int NE = tri_count(D);
    size_t b = sizeof(cplx) * ((size_t)D * EM_TS + (size_t)NE * K);
    b += sizeof(double) * ((size_t)4 * K * EM_TILE + (size_t)K * EM_TILE + 2 * K);
    b += 2 * NE;
    return (b + 15) / 16 * 16;
}

int em_chunks(int F, int64_t T, int *chunk_frames) {
    // enough workgroups to fill 256 CUs a few times over, whole tiles per chunk
    int64_t tiles = (T + EM_TILE - 1) / EM_TILE;
    int64_t want = (4096 + F - 1) / F;      // ~4096 workgroups
    if (want < 1) want = 1;
    int64_t tiles_per_chunk = (tiles + want - 1) / want;
    if (tiles_per_chunk < 1) tiles_per_chunk = 1;
    *chunk_frames = (int)(tiles_per_chunk * EM_TILE);
    return (int)((tiles + tiles_per_chunk - 1) / tiles_per_chunk);
}

template <int K>
int launch_step(gss_ctx *ctx, int mode, const EmArgs &a, int F) {
    const size_t lds = em_step_lds(a.D, K);
    dim3 grid(a.nch, F), block(256);
    if (lds > 64 * 1024) {
        const void *fn = mode == MODE_FIRST
                             ? reinterpret_cast<const void *>(em_step_kernel<K, MODE_FIRST>)
                             : mode == MODE_EM
                                   ? reinterpret_cast<const void *>(em_step_kernel<K, MODE_EM>)
                                   : reinterpret_cast<const void *>(em_step_kernel<K, MODE_PREDICT>);
        GSS_HIP_CHECK(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)lds));
    }
    if (mode == MODE_FIRST) {
        GSS_PROF(ctx, "em_step");
        hipLaunchKernelGGL((em_step_kernel<K, MODE_FIRST>), grid, block, lds, ctx->stream, a);
    } else if (mode == MODE_EM) {
        GSS_PROF(ctx, "em_step");
        hipLaunchKernelGGL((em_step_kernel<K, MODE_EM>), grid, block, lds, ctx->stream, a);
    } else {
        GSS_PROF(ctx, "em_predict");
        hipLaunchKernelGGL((em_step_kernel<K, MODE_PREDICT>), grid, block, lds, ctx->stream, a);
    }
    GSS_LAUNCH_CHECK(ctx, "em_step_kernel");
    return GSS_OK;
}

int launch_step_k(gss_ctx *ctx, int K, int mode, const EmArgs &a, int F) {
    switch (K) {
        case 1: return launch_step<1>(ctx, mode, a, F);
        case 2: return launch_step<2>(ctx, mode, a, F);
        case 3: return launch_step<3>(ctx, mode, a, F);
        case 4: return launch_step<4>(ctx, mode, a, F);
        case 5: return launch_step<5>(ctx, mode, a, F);
        case 6: return launch_step<6>(ctx, mode, a, F);
        case 7: return launch_step<7>(ctx, mode, a, F);
        case 8: return launch_step<8>(ctx, mode, a, F);
    }
    return gss_fail(ctx, GSS_ERR_UNSUPPORTED, "cacgmm: K=%d", K);
}

}  // namespace

size_t cacgmm_workspace_bytes(int F, int64_t T, int D, int K) {
    const size_t NE = tri_count(D);
    int cf;
    const int nch = em_chunks(F, T, &cf);
    size_t b = 0;
    b += align_up(sizeof(cplx) * (size_t)F * NE * K);            // Mq
    b += 2 * align_up(sizeof(double) * (size_t)F * K);           // logdet, pi
    b += align_up(sizeof(cplx) * (size_t)F * nch * K * NE);      // Bp
    b += align_up(sizeof(double) * (size_t)F * nch * K);         // Sg
    return b + 4096;
}

int cacgmm_run(gss_ctx *ctx, const cplx *Y, int F, int64_t T, int D, const uint8_t *act,
               int64_t act_stride, int K,
               int iterations, int iterations_post, double *gamma) {
    const int NE = tri_count(D);
    EmArgs a{};
    a.Y = Y;
    a.act = act;
    a.act_stride = act_stride;
    a.T = T;
    a.D = D;
    a.NE = NE;
    a.nch = em_chunks(F, T, &a.chunk_frames);
    cplx *Mq = arena_alloc_t<cplx>(ctx, (size_t)F * NE * K);
    double *logdet = arena_alloc_t<double>(ctx, (size_t)F * K);
    double *pi = arena_alloc_t<double>(ctx, (size_t)F * K);
    a.Bp = arena_alloc_t<cplx>(ctx, (size_t)F * a.nch * K * NE);
    a.Sg = arena_alloc_t<double>(ctx, (size_t)F * a.nch * K);
    GSS_REQUIRE(ctx, Mq && logdet && pi && a.Bp && a.Sg, GSS_ERR_NOMEM, "cacgmm workspace");
    a.Mq = Mq;
    a.logdet = logdet;
    a.pi = pi;
    a.gamma = gamma;

    const size_t lds = em_step_lds(D, K);
    GSS_REQUIRE(ctx, lds <= 160 * 1024, GSS_ERR_UNSUPPORTED, "cacgmm: D=%d K=%d LDS", D, K);
    const int m = D + (D & 1);
    const size_t eig_lds = (sizeof(cplx) * 2 * m * m + sizeof(double) * m + 15) / 16 * 16;
    const int force_eigh = getenv("GSS_FORCE_EIGH") != nullptr;

    auto eig = [&]() -> int {
        GSS_PROF(ctx, "em_eig");
        hipLaunchKernelGGL(em_eig_kernel, dim3(K, F), dim3(64), eig_lds, ctx->stream, a.Bp, a.Sg,
                           a.nch, D, K, T, 1e-10, force_eigh, Mq, logdet, pi, (int *)nullptr);
        GSS_LAUNCH_CHECK(ctx, "em_eig_kernel");
        return GSS_OK;
    };

    // CACGMMTrainer.fit(initialization=array, iterations=I, source_activity_mask)
    for (int it = 0; it < iterations; ++it) {
        a.masked = 1;
        a.aff_eps = 1e-10;
        GSS_TRY(launch_step_k(ctx, K, it == 0 ? MODE_FIRST : MODE_EM, a, F));
        GSS_TRY(eig());
    }
    if (iterations_post > 1) {
        // fit(initialization=model, iterations=post-1): no mask, default clip
        for (int it = 0; it < iterations_post - 1; ++it) {
            a.masked = 0;
            a.aff_eps = 1e-10;
            GSS_TRY(launch_step_k(ctx, K, MODE_EM, a, F));
            GSS_TRY(eig());
        }
    }
    // predict: affiliation_eps = 0; mask only when iterations_post == 0
    a.masked = iterations_post == 0 ? 1 : 0;
    a.aff_eps = 0.0;
    GSS_TRY(launch_step_k(ctx, K, MODE_PREDICT, a, F));
    return GSS_OK;
}
