// WPE dereverberation: nara_wpe.wpe.wpe_v8(statistics_mode='full', psd_context=0)
// as called from WPE.__call__ (/root/reference/pb_chime5/core.py:48-58).
//
// Per frequency f and iteration:
//   lambda_t = mean_d |X[t,d]|^2 ;  w_t = 1 / max(lambda_t, 1e-10 * max_t lambda_t)
//   R = sum_t w_t  yt_t yt_t^H  (n x n, n = taps*D),   P = sum_t w_t yt_t y_t^H  (n x D)
//   G = solve(R, P) ;  X[t] = Y[t] - G^H yt_t
// where yt_t stacks the `taps` delayed frames.  With the stacking order of
// nara_wpe's build_y_tilde (largest delay first) and Y stored (T,D) row-major per
// frequency, yt_t is a CONTIGUOUS window of the flat array:
//       yt_t[r] = Yflat[(t - c) * D + r],  c = delay + taps - 1,  r in [0, n)
// and y_t is the same window continued at r in [c*D, c*D + D).  So R and P are
// both slices of one sliding-window Gram matrix -- the only true dense
// contraction of the hot path, done here with the f64 MFMA
// (v_mfma_f64_16x16x4_f64) straight out of an LDS copy of the window.
#include "gss_internal.h"

typedef double v4d __attribute__((ext_vector_type(4)));

namespace {

// ------------------------------------------------------------------ power
__global__ __launch_bounds__(256) void wpe_power_kernel(const cplx *__restrict__ X, int64_t T,
                                                        int D, double *__restrict__ w) {
    __shared__ double red[4];
    const int f = blockIdx.x;
    const cplx *Xf = X + (int64_t)f * T * D;
    double *wf = w + (int64_t)f * T;
    double mx = 0.0;
    for (int64_t t = threadIdx.x; t < T; t += blockDim.x) {
        const cplx *y = Xf + t * D;
        double p = 0.0;
        for (int d = 0; d < D; ++d) p += c_abs2(y[d]);
        p = p / (double)D;
        wf[t] = p;
        mx = fmax(mx, p);
    }
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    const double eps = 1e-10 * mx;
    for (int64_t t = threadIdx.x; t < T; t += blockDim.x) wf[t] = 1.0 / fmax(wf[t], eps);
}

// ------------------------------------------------------------------ correlation (MFMA)
constexpr int CT = 48;        // wave tile edge: 3 x 3 MFMA tiles of 16 x 16
constexpr int CORR_KT = 64;   // frames staged per chunk

struct CorrTile {
    int row_off, col_off, is_p, pad;
};

// grid: (tile groups, F); block: 256 = 4 waves, one 48 x 48 tile each.
__global__ __launch_bounds__(256) void wpe_corr_kernel(
    const cplx *__restrict__ Y, const double *__restrict__ w, int64_t T, int D, int n, int c,
    int padf, const CorrTile *__restrict__ tiles, int ntiles, cplx *__restrict__ R,
    cplx *__restrict__ P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int frames_lds = CORR_KT + c + padf;
    cplx *S = reinterpret_cast<cplx *>(smem);                      // frames_lds * D
    double *wS = reinterpret_cast<double *>(S + frames_lds * D);   // CORR_KT

    const int f = blockIdx.y;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int tile_id = blockIdx.x * 4 + wave;
    const bool active = tile_id < ntiles;
    CorrTile tl = tiles[active ? tile_id : 0];
    const int li = lane & 15, lk = lane >> 4;
    const cplx *Yf = Y + (int64_t)f * T * D;
    const double *wf = w + (int64_t)f * T;

    v4d acc_re[3][3], acc_im[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            acc_re[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
            acc_im[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
        }

    for (int64_t t0 = 0; t0 < T; t0 += CORR_KT) {
        __syncthreads();
        // stage frames [t0 - c, t0 - c + frames_lds) of Yflat, zero outside [0, T)
        const int64_t fr0 = t0 - c;
        for (int idx = threadIdx.x; idx < frames_lds * D; idx += blockDim.x) {
            const int64_t fr = fr0 + idx / D;
            cplx v = c_make(0.0, 0.0);
            if (fr >= 0 && fr < T) v = Yf[fr0 * D + idx];
            S[idx] = v;
        }
        for (int k = threadIdx.x; k < CORR_KT; k += blockDim.x)
            wS[k] = (t0 + k < T) ? wf[t0 + k] : 0.0;
        __syncthreads();
        if (!active) continue;
        const int ksteps = CORR_KT / 4;
        for (int ks = 0; ks < ksteps; ++ks) {
            const int kf = 4 * ks + lk;
            const double wt = wS[kf];
            const cplx *base = S + kf * D + li;
            double ar[3], ai[3], br[3], bi[3], nbi[3];
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                const cplx a = base[tl.row_off + 16 * m];
                ar[m] = a.x * wt;
                ai[m] = a.y * wt;
                const cplx b = base[tl.col_off + 16 * m];
                br[m] = b.x;
                bi[m] = b.y;
                nbi[m] = -b.y;
            }
            // (a_r + i a_i) * (b_r - i b_i)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[a], br[b], acc_re[a][b], 0, 0, 0);
                    acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[a], br[b], acc_im[a][b], 0, 0, 0);
                }
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[a], bi[b], acc_re[a][b], 0, 0, 0);
                    acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[a], nbi[b], acc_im[a][b], 0, 0, 0);
                }
        }
    }
    if (!active) return;
    // C/D fragment of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
    cplx *Rf = R + (int64_t)f * n * n;
    cplx *Pf = P + (int64_t)f * n * D;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = tl.row_off + 16 * a + lk + 4 * reg;
                const int cc = tl.col_off + 16 * b + li;
                const cplx v = c_make(acc_re[a][b][reg], acc_im[a][b][reg]);
                if (r >= n) continue;
                if (tl.is_p) {
                    const int d = cc - c * D;
                    if (d >= 0 && d < D) Pf[r * D + d] = v;
                } else if (cc < n) {
                    Rf[r * n + cc] = v;
                }
            }
}

// ------------------------------------------------------------------ solve
// Augmented right-looking Cholesky  [R | P] -> [U | Z] with R = U^H U (upper
// triangle of R is the input), then back substitution U G = Z.  G overwrites P.
// A non-positive pivot (exactly singular system, e.g. an all-zero channel) zeroes
// the row, which reproduces the minimum-norm lstsq fallback of stable_solve
// (pb_chime5/math/solve.py:95-114) for zero rows/columns.
__global__ __launch_bounds__(256) void wpe_solve_kernel(cplx *__restrict__ R,
                                                        cplx *__restrict__ P, int n, int D) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx *rowj = reinterpret_cast<cplx *>(smem);   // n + D
    double &s_dinv = *reinterpret_cast<double *>(rowj + n + D);
    const int f = blockIdx.x;
    cplx *A = R + (int64_t)f * n * n;
    cplx *Z = P + (int64_t)f * n * D;
    const int tid = threadIdx.x;

    for (int j = 0; j < n; ++j) {
        if (tid == 0) {
            const double a = A[(int64_t)j * n + j].x;
            double d = 0.0, dinv = 0.0;
            if (a > 0.0 && isfinite(a)) {
                d = sqrt(a);
                dinv = 1.0 / d;
            }
            A[(int64_t)j * n + j] = c_make(d, 0.0);
            s_dinv = dinv;
        }
        __syncthreads();
        const double dinv = s_dinv;
        const int m = n - j - 1;
        for (int idx = tid; idx < m + D; idx += blockDim.x) {
            cplx *p = idx < m ? &A[(int64_t)j * n + j + 1 + idx] : &Z[j * D + (idx - m)];
            const cplx v = c_scale(*p, dinv);
            *p = v;
            rowj[idx] = v;
        }
        __syncthreads();
        // trailing update: A[i][k] -= conj(U[j][i]) U[j][k]  (k >= i),  Z[i] -= conj(U[j][i]) Z[j]
        const int tx = tid & 31, ty = tid >> 5;
        for (int ii = ty; ii < m; ii += 8) {
            const cplx u = rowj[ii];
            if (u.x == 0.0 && u.y == 0.0) continue;
            const int i = j + 1 + ii;
            cplx *Ai = A + (int64_t)i * n;
            for (int kk = ii + tx; kk < m; kk += 32) {
                cplx v = Ai[j + 1 + kk];
                const cplx r = rowj[kk];
                v.x -= u.x * r.x + u.y * r.y;
                v.y -= u.x * r.y - u.y * r.x;
                Ai[j + 1 + kk] = v;
            }
            for (int dd = tx; dd < D; dd += 32) {
                cplx v = Z[i * D + dd];
                const cplx r = rowj[m + dd];
                v.x -= u.x * r.x + u.y * r.y;
                v.y -= u.x * r.y - u.y * r.x;
                Z[i * D + dd] = v;
            }
        }
        __syncthreads();
    }
    // back substitution (column oriented): g_j = z_j / U[j][j]; z_i -= U[i][j] g_j (i < j)
    for (int j = n - 1; j >= 0; --j) {
        const double d = A[(int64_t)j * n + j].x;
        const double dinv = d > 0.0 ? 1.0 / d : 0.0;
        if (tid < D) {
            const cplx g = c_scale(Z[j * D + tid], dinv);
            Z[j * D + tid] = g;
            rowj[tid] = g;
        }
        __syncthreads();
        for (int idx = tid; idx < j * D; idx += blockDim.x) {
            const int i = idx / D, dd = idx - i * D;
            const cplx u = A[(int64_t)i * n + j];
            const cplx g = rowj[dd];
            cplx v = Z[idx];
            v.x -= u.x * g.x - u.y * g.y;
            v.y -= u.x * g.y + u.y * g.x;
            Z[idx] = v;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ apply
// X[t][d] = Y[t][d] - sum_r conj(G[r][d]) Yflat[(t - c) D + r]
constexpr int APPLY_TB = 4;

__global__ __launch_bounds__(256) void wpe_apply_kernel(const cplx *__restrict__ Y,
                                                        const cplx *__restrict__ G, int64_t T,
                                                        int D, int n, int c, int tc,
                                                        cplx *__restrict__ X) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx *Gs = reinterpret_cast<cplx *>(smem);   // n * D
    cplx *S = Gs + n * D;                        // (tc + c) * D
    const int f = blockIdx.y;
    const int64_t t0 = (int64_t)blockIdx.x * tc;
    const cplx *Yf = Y + (int64_t)f * T * D;
    const cplx *Gf = G + (int64_t)f * n * D;
    for (int idx = threadIdx.x; idx < n * D; idx += blockDim.x) Gs[idx] = Gf[idx];
    const int64_t fr0 = t0 - c;
    for (int idx = threadIdx.x; idx < (tc + c) * D; idx += blockDim.x) {
        const int64_t fr = fr0 + idx / D;
        cplx v = c_make(0.0, 0.0);
        if (fr >= 0 && fr < T) v = Yf[fr0 * D + idx];
        S[idx] = v;
    }
    __syncthreads();
    const int ntq = blockDim.x / D;
    const int d = threadIdx.x % D, tq = threadIdx.x / D;
    if (tq >= ntq) return;
    cplx acc[APPLY_TB];
#pragma unroll
    for (int b = 0; b < APPLY_TB; ++b) acc[b] = c_make(0.0, 0.0);
    const cplx *Sb = S + tq * APPLY_TB * D;
    for (int r = 0; r < n; ++r) {
        const cplx g = Gs[r * D + d];
#pragma unroll
        for (int b = 0; b < APPLY_TB; ++b) c_cfma(acc[b], g, Sb[b * D + r]);
    }
#pragma unroll
    for (int b = 0; b < APPLY_TB; ++b) {
        const int64_t t = t0 + tq * APPLY_TB + b;
        if (t < T) {
            const cplx y = S[(tq * APPLY_TB + b + c) * D + d];
            X[((int64_t)f * T + t) * D + d] = c_sub(y, acc[b]);
        }
    }
}

// ------------------------------------------------------------------ MFMA layout self-test
__global__ void mfma_selftest_kernel(double *out) {
    // A[i][k] = i + 1 (k = 0 only), B[k][j] = 100 * (j + 1) (k = 0 only)
    const int lane = threadIdx.x;
    const int li = lane & 15, lk = lane >> 4;
    const double a = lk == 0 ? (double)(li + 1) : 0.0;
    const double b = lk == 0 ? 100.0 * (double)(li + 1) : 0.0;
    v4d acc = {0.0, 0.0, 0.0, 0.0};
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    for (int reg = 0; reg < 4; ++reg) {
        const int row = lk + 4 * reg, col = li;
        out[row * 16 + col] = acc[reg];
    }
}

}  // namespace

static int corr_tiles(int n, int D, int c, std::vector<CorrTile> &tiles) {
    const int nt = (n + CT - 1) / CT;
    for (int i = 0; i < nt; ++i)
        for (int j = i; j < nt; ++j) tiles.push_back({i * CT, j * CT, 0, 0});
    const int np = (D + CT - 1) / CT;
    for (int i = 0; i < nt; ++i)
        for (int j = 0; j < np; ++j) tiles.push_back({i * CT, c * D + j * CT, 1, 0});
    return (int)tiles.size();
}

static int corr_padf(int D) { return (CT + D - 1) / D + 1; }

size_t wpe_workspace_bytes(int F, int64_t T, int D, int taps, int delay) {
    const size_t n = (size_t)taps * D;
    size_t b = 0;
    b += align_up(sizeof(double) * (size_t)F * T);       // w
    b += align_up(sizeof(cplx) * (size_t)F * n * n);     // R
    b += align_up(sizeof(cplx) * (size_t)F * n * D);     // P / G
    b += align_up(sizeof(CorrTile) * 1024);
    return b + 4096;
}

int wpe_run(gss_ctx *ctx, const cplx *Y, int F, int64_t T, int D, int taps, int delay,
            int iterations, cplx *X) {
    const int n = taps * D;
    const int c = delay + taps - 1;
    if (iterations == 0) {
        if (X != Y)
            GSS_HIP_CHECK(ctx, hipMemcpyAsync(X, Y, sizeof(cplx) * (size_t)F * T * D,
                                              hipMemcpyDeviceToDevice, ctx->stream));
        return GSS_OK;
    }
    GSS_REQUIRE(ctx, X != Y, GSS_ERR_INVALID, "gss_wpe: X must not alias Y");
    double *w = arena_alloc_t<double>(ctx, (size_t)F * T);
    cplx *R = arena_alloc_t<cplx>(ctx, (size_t)F * n * n);
    cplx *P = arena_alloc_t<cplx>(ctx, (size_t)F * n * D);
    CorrTile *tiles_dev = arena_alloc_t<CorrTile>(ctx, 1024);
    GSS_REQUIRE(ctx, w && R && P && tiles_dev, GSS_ERR_NOMEM, "wpe workspace");

    std::vector<CorrTile> tiles;
    const int ntiles = corr_tiles(n, D, c, tiles);
    GSS_REQUIRE(ctx, ntiles <= 1024, GSS_ERR_UNSUPPORTED, "wpe: taps*D=%d too large", n);
    GSS_HIP_CHECK(ctx, hipMemcpyAsync(tiles_dev, tiles.data(), sizeof(CorrTile) * ntiles,
                                      hipMemcpyHostToDevice, ctx->stream));
    // the tile list lives on the host stack: make sure the copy has been staged
    GSS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));

    const int padf = corr_padf(D);
    const size_t corr_lds = sizeof(cplx) * (size_t)(CORR_KT + c + padf) * D + sizeof(double) * CORR_KT;
    const size_t solve_lds = sizeof(cplx) * (size_t)(n + D) + 16;
    const int ntq = 256 / D;
    const int tc = ntq * APPLY_TB;
    const size_t apply_lds = sizeof(cplx) * ((size_t)n * D + (size_t)(tc + c) * D);
    GSS_REQUIRE(ctx, corr_lds <= 160 * 1024 && apply_lds <= 160 * 1024, GSS_ERR_UNSUPPORTED,
                "wpe: taps=%d D=%d needs more LDS than a CU has", taps, D);
    if (apply_lds > 64 * 1024)
        GSS_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(wpe_apply_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)apply_lds));
    if (corr_lds > 64 * 1024)
        GSS_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(wpe_corr_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)corr_lds));

    for (int it = 0; it < iterations; ++it) {
        const cplx *cur = it == 0 ? Y : X;
        {
            GSS_PROF(ctx, "wpe_power");
            hipLaunchKernelGGL(wpe_power_kernel, dim3(F), dim3(256), 0, ctx->stream, cur, T, D, w);
            GSS_LAUNCH_CHECK(ctx, "wpe_power_kernel");
        }
        {
            GSS_PROF(ctx, "wpe_corr");
            hipLaunchKernelGGL(wpe_corr_kernel, dim3((ntiles + 3) / 4, F), dim3(256), corr_lds,
                               ctx->stream, Y, w, T, D, n, c, padf, tiles_dev, ntiles, R, P);
            GSS_LAUNCH_CHECK(ctx, "wpe_corr_kernel");
        }
        {
            GSS_PROF(ctx, "wpe_solve");
            hipLaunchKernelGGL(wpe_solve_kernel, dim3(F), dim3(256), solve_lds, ctx->stream, R, P,
                               n, D);
            GSS_LAUNCH_CHECK(ctx, "wpe_solve_kernel");
        }
        {
            GSS_PROF(ctx, "wpe_apply");
            hipLaunchKernelGGL(wpe_apply_kernel, dim3((unsigned)((T + tc - 1) / tc), F), dim3(256),
                               apply_lds, ctx->stream, Y, P, T, D, n, c, tc, X);
            GSS_LAUNCH_CHECK(ctx, "wpe_apply_kernel");
        }
    }
    return GSS_OK;
}

int selftest_mfma_run(gss_ctx *ctx) {
    double *out = nullptr;
    GSS_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void **>(&out), sizeof(double) * 256));
    hipLaunchKernelGGL(mfma_selftest_kernel, dim3(1), dim3(64), 0, ctx->stream, out);
    double host[256];
    hipError_t e = hipMemcpyAsync(host, out, sizeof(host), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(out);
    if (e != hipSuccess) return gss_fail(ctx, GSS_ERR_HIP, "mfma selftest: %s", hipGetErrorString(e));
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            const double want = (double)(i + 1) * 100.0 * (double)(j + 1);
            if (host[i * 16 + j] != want)
                return gss_fail(ctx, GSS_ERR_UNSUPPORTED,
                                "f64 MFMA fragment layout mismatch at (%d,%d): got %g want %g", i,
                                j, host[i * 16 + j], want);
        }
    return GSS_OK;
}
