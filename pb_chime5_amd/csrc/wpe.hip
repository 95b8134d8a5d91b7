// WPE dereverberation: nara_wpe.wpe.wpe_v8(statistics_mode='full', psd_context)
// as called from WPE.__call__ (/root/reference/pb_chime5/core.py:48-58).
//
// Per frequency f and iteration:
//   lambda_t = mean_d |X[t,d]|^2   (psd_context = p > 0: averaged over the frames
//   t-p..t+p that exist) ;  w_t = 1 / max(lambda_t, 1e-10 * max_t lambda_t)
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
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "gss_internal.h"
#include <type_traits>
#include "dense_wave.h"

typedef double v4d __attribute__((ext_vector_type(4)));

namespace {

// ------------------------------------------------------------------ power
// One workgroup per frequency (the maximum over the frames comes before the inversion).
// Frames are taken 256 at a time: the 256 x D squared magnitudes are read with coalesced
// loads (consecutive lanes, consecutive bins -- a lane walking its own frame touches 64
// cache lines per load), parked in LDS, and summed per frame from there.
constexpr int POW_FRAMES = 256;
// psd_context > 0 (nara_wpe.wpe.get_power): the per-frame power is first written to `raw`
// and w takes its moving average over the existing frames of [t - p, t + p] -- np.correlate
// with ones(2p + 1), mode 'full', divided by the same correlation of ones -- before the
// maximum and the floor.
__global__ __launch_bounds__(POW_FRAMES) void wpe_power_kernel(const cplx *__restrict__ X,
                                                               int64_t T, int D, int psd_context,
                                                               double *__restrict__ raw,
                                                               double *__restrict__ w) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    double *ps = reinterpret_cast<double *>(smem);      // POW_FRAMES * (D + 1)
    __shared__ double red[POW_FRAMES / 64];
    const int f = blockIdx.x, tid = threadIdx.x;
    const cplx *Xf = X + (int64_t)f * T * D;
    double *wout = w + (int64_t)f * T;
    double *wf = psd_context > 0 ? raw + (int64_t)f * T : wout;
    const int DP = D + 1;                                // odd-ish stride against bank conflicts
    double mx = 0.0;
    for (int64_t t0 = 0; t0 < T; t0 += POW_FRAMES) {
        const int nfr = (int)min((int64_t)POW_FRAMES, T - t0);
        const int total = nfr * D;
        const cplx *src = Xf + t0 * D;
        __syncthreads();
        // eight loads of a thread in flight before the first LDS store (one per trip was a
        // chain of round trips: 513 workgroups x 256 threads x 16 bytes in flight are a
        // quarter of what the memory system needs to stream), (frame, channel) of an element
        // advanced without dividing: idx + 256 = (fr + 256 / D, d + 256 % D) with one carry
        {
            constexpr int LB = 8;
            const int qf = POW_FRAMES / D, rf = POW_FRAMES - qf * D;
            int fr = tid / D, d = tid - fr * D;
            for (int base = tid; base < total; base += LB * POW_FRAMES) {
                cplx v[LB];
#pragma unroll
                for (int j = 0; j < LB; ++j) {
                    const int idx = base + j * POW_FRAMES;
                    v[j] = idx < total ? src[idx] : c_make(0.0, 0.0);
                }
#pragma unroll
                for (int j = 0; j < LB; ++j) {
                    if (base + j * POW_FRAMES < total) ps[fr * DP + d] = c_abs2(v[j]);
                    fr += qf;
                    d += rf;
                    if (d >= D) {
                        d -= D;
                        ++fr;
                    }
                }
            }
        }
        __syncthreads();
        if (tid < nfr) {
            double p = 0.0;
            for (int d = 0; d < D; ++d) p += ps[tid * DP + d];
            p = p / (double)D;
            wf[t0 + tid] = p;
            mx = fmax(mx, p);
        }
    }
    if (psd_context > 0) {
        __syncthreads();      // the raw powers of this frequency are complete (same workgroup)
        mx = 0.0;
        if (psd_context <= 16) {
            // short windows: the plain sum, in np.correlate's order
            for (int64_t t = tid; t < T; t += POW_FRAMES) {
                const int64_t lo = t - psd_context > 0 ? t - psd_context : 0;
                const int64_t hi = t + psd_context < T - 1 ? t + psd_context : T - 1;
                double sum = 0.0;
                for (int64_t u = lo; u <= hi; ++u) sum += wf[u];
                const double p = sum / (double)(hi - lo + 1);
                wout[t] = p;
                mx = fmax(mx, p);
            }
        } else {
            // long windows: every thread takes a run of consecutive frames, sums the first
            // window and slides it (O(T / 256 + p) per thread instead of O(T p / 256)).  The
            // running sum is kept as an unevaluated pair (sum, carry) updated with error-free
            // additions: a plain `sum += in; sum -= out` keeps eps * (the loudest frame that
            // ever was in the window) after that frame has left -- a relative error of 1e-6
            // in the power of frames 100 dB below it, and a result that depends on where the
            // 256 runs start -- while np.correlate sums every window afresh.  With the pair
            // the slid sum equals the fresh one to ~eps^2 of the loud frame.
            const int64_t run = (T + POW_FRAMES - 1) / POW_FRAMES;
            const int64_t t_beg = (int64_t)tid * run, t_end = t_beg + run < T ? t_beg + run : T;
            double sum = 0.0, carry = 0.0;
            auto add = [&](double x) {
                const double s = sum + x;
                const double bb = s - sum;
                const double err = (sum - (s - bb)) + (x - bb);       // two-sum: s + err == sum + x
                carry += err;
                sum = s;
            };
            int64_t lo = 0, hi = -1;
            for (int64_t t = t_beg; t < t_end; ++t) {
                const int64_t nlo = t - psd_context > 0 ? t - psd_context : 0;
                const int64_t nhi = t + psd_context < T - 1 ? t + psd_context : T - 1;
                if (t == t_beg) {
                    for (int64_t u = nlo; u <= nhi; ++u) add(wf[u]);
                } else {
                    if (nhi > hi) add(wf[nhi]);
                    if (nlo > lo) add(-wf[lo]);
                }
                lo = nlo;
                hi = nhi;
                const double p = (sum + carry) / (double)(hi - lo + 1);
                wout[t] = p;
                mx = fmax(mx, p);
            }
        }
    }
    mx = wave_max(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    const double eps = 1e-10 * mx;
    for (int64_t t = tid; t < T; t += POW_FRAMES) wout[t] = 1.0 / fmax(wout[t], eps);
}

// ------------------------------------------------------------------ correlation (MFMA)
#ifndef GSS_CORR_KT
#define GSS_CORR_KT 64
#endif
constexpr int CORR_KT = GSS_CORR_KT;   // frames staged per chunk
constexpr int CORR_FINE_MAX_SUBTILES = 48;   // up to here (D <= 12 at 10 taps): one wave per 16 x 16 sub-tile

struct CorrTile {
    int row_off, col_off, is_p, mask;   // mask: needed 16 x 16 sub-tiles, bit a * TS + b
};

// 1-D XCD-mapped grid over (tile groups, F); block: NW waves (4, or 2 when that packs the
// chip better, see wpe_run), one (16 TS) x (16 TS) tile each.  MASK (bit a * TS + b) says which 16 x 16 sub-tiles of
// the wave's tile are needed: diagonal tiles skip the sub-tile below the diagonal and
// tiles on the edge of n (240 = 7.5 x 32) the half that lies outside, which removes
// 15 % of the MFMAs at taps * D = 240.  The host sorts the tile list by mask so that the
// four waves of a workgroup carry similar loads.
// The MFMAs of one staged chunk (CORR_KT frames = CORR_KT / 4 k-steps) of a wave's tile:
// S = the chunk's window (frame kf of the chunk at S + kf * D), wS = its CORR_KT weights.
// 3M complex product: with t1 = sum ar br, t2 = sum ai bi, t3 = sum (ar+ai)(br-bi)
//   re(a conj b) = t1 + t2,   im(a conj b) = t3 - t1 + t2
// -- three real MFMAs per tile and k-step instead of four.
// KSTEPS: k-steps of the chunk that are run (frames 0 .. 4 KSTEPS - 1 of the window).
template <int TS, int MASK, int KSTEPS = CORR_KT / 4>
__device__ __forceinline__ void corr_chunk(const cplx *S, const double *wS, int D,
                                           const CorrTile &tl, v4d (&t1)[TS][TS],
                                           v4d (&t2)[TS][TS], v4d (&t3)[TS][TS]) {
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    auto need = [](int a, int b) { return ((MASK >> (a * TS + b)) & 1) != 0; };
    auto need_row = [&](int a) {
        bool r = false;
        for (int b = 0; b < TS; ++b) r = r || need(a, b);
        return r;
    };
    auto need_col = [&](int b) {
        bool r = false;
        for (int a = 0; a < TS; ++a) r = r || need(a, b);
        return r;
    };
    const int ksteps = KSTEPS;
    // operands of k-step ks+1 are fetched from LDS while the MFMAs of ks run
    cplx a_cur[TS], b_cur[TS], a_nxt[TS], b_nxt[TS];
    double w_cur, w_nxt = 0.0;
    {
        const cplx *base = S + lk * D + li;
        w_cur = wS[lk];
#pragma unroll
        for (int m = 0; m < TS; ++m) {
            a_cur[m] = base[tl.row_off + 16 * m];
            b_cur[m] = base[tl.col_off + 16 * m];
        }
    }
    // (one-sub-tile waves are not unrolled all the way: hoisting 16 k-steps of operand
    // loads costs the registers that let a fourth and fifth wave share the SIMD)
    constexpr int KU = TS == 1 || KSTEPS < 16 ? 4 : 16;
#pragma unroll KU
    for (int ks = 0; ks < ksteps; ++ks) {
        if (ks + 1 < ksteps) {
            const int kf = 4 * (ks + 1) + lk;
            const cplx *base = S + kf * D + li;
            w_nxt = wS[kf];
#pragma unroll
            for (int m = 0; m < TS; ++m) {
                a_nxt[m] = need_row(m) ? base[tl.row_off + 16 * m] : c_make(0.0, 0.0);
                b_nxt[m] = need_col(m) ? base[tl.col_off + 16 * m] : c_make(0.0, 0.0);
            }
        }
        double ar[TS], ai[TS], as[TS], br[TS], bi[TS], bd[TS];
#pragma unroll
        for (int m = 0; m < TS; ++m) {
            ar[m] = a_cur[m].x * w_cur;
            ai[m] = a_cur[m].y * w_cur;
            as[m] = ar[m] + ai[m];
            br[m] = b_cur[m].x;
            bi[m] = b_cur[m].y;
            bd[m] = b_cur[m].x - b_cur[m].y;
        }
#pragma unroll
        for (int a = 0; a < TS; ++a)
#pragma unroll
            for (int b = 0; b < TS; ++b) {
                if (!need(a, b)) continue;
                t1[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar[a], br[b], t1[a][b], 0, 0, 0);
                t2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai[a], bi[b], t2[a][b], 0, 0, 0);
                t3[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[a], bd[b], t3[a][b], 0, 0, 0);
            }
#pragma unroll
        for (int m = 0; m < TS; ++m) {
            a_cur[m] = a_nxt[m];
            b_cur[m] = b_nxt[m];
        }
        w_cur = w_nxt;
    }
}

// The finished tile of a wave -> R (upper tiles) or P.
// C/D fragment of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
// (RAW: t1 and t3 already hold the real and imaginary part -- the block-wise accumulation)
template <int TS, int MASK, bool RAW = false>
__device__ __forceinline__ void corr_store(const CorrTile &tl, int f, int n, int c, int D,
                                           const v4d (&t1)[TS][TS], const v4d (&t2)[TS][TS],
                                           const v4d (&t3)[TS][TS], cplx *__restrict__ R,
                                           cplx *__restrict__ P) {
    const int lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    auto need = [](int a, int b) { return ((MASK >> (a * TS + b)) & 1) != 0; };
    cplx *Rf = R + (int64_t)f * n * n;
    cplx *Pf = P + (int64_t)f * n * D;
#pragma unroll
    for (int a = 0; a < TS; ++a)
#pragma unroll
        for (int b = 0; b < TS; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                if (!need(a, b)) continue;
                const int r = tl.row_off + 16 * a + lk + 4 * reg;
                const int cc = tl.col_off + 16 * b + li;
                const cplx v = RAW ? c_make(t1[a][b][reg], t3[a][b][reg])
                                   : c_make(t1[a][b][reg] + t2[a][b][reg],
                                            (t3[a][b][reg] - t1[a][b][reg]) + t2[a][b][reg]);
                if (r >= n) continue;
                if (tl.is_p == 1) {
                    const int d = cc - c * D;
                    if (d >= 0 && d < D) Pf[r * D + d] = v;
                } else {
                    if (cc < n) Rf[r * n + cc] = v;
                    if (tl.is_p == 2) {          // a tile of R's last column that also holds P
                        const int d = cc - c * D;      // (P overlaps R's own columns when c D < n)
                        if (d >= 0 && d < D) Pf[r * D + d] = v;
                    }
                }
            }
}

template <int TS>
__device__ __forceinline__ void corr_zero(v4d (&t1)[TS][TS], v4d (&t2)[TS][TS], v4d (&t3)[TS][TS]) {
#pragma unroll
    for (int a = 0; a < TS; ++a)
#pragma unroll
        for (int b = 0; b < TS; ++b) {
            t1[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
            t2[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
            t3[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
        }
}

template <int TS, int MASK, int NW, int STG1 = 8>
__device__ __forceinline__ void corr_tile_body(
    const cplx *__restrict__ Yf, const double *__restrict__ wf, int64_t T, int D, int n, int c,
    int frames_lds, cplx *S, double *wS, const CorrTile tl, bool active, int f,
    cplx *__restrict__ R, cplx *__restrict__ P) {
    v4d t1[TS][TS], t2[TS][TS], t3[TS][TS];
    corr_zero<TS>(t1, t2, t3);

    // The window of chunk i+1 is fetched from global memory into registers before the
    // MFMAs of chunk i are issued and written to LDS after them, so the global-load
    // latency hides under the matrix work (elements beyond CORR_STG * 256 -- only for
    // very wide windows -- are staged synchronously).
    constexpr int NT = 64 * NW;              // threads of the workgroup
#ifndef GSS_CORR_STG_ELEMS
#define GSS_CORR_STG_ELEMS 2048
#endif
    // (single-wave workgroups: STG1 elements per lane -- 5 cover one array's 80-frame window of
    // 320 elements, and the 12 registers that 8 would cost more are what lets a FOURTH wave
    // share the SIMD: at 138 registers 3072 waves were resident and the last 6 of the
    // 513 x 6 = 3078 ran alone in a second round -- 0.27 ms per launch of which 0.1 ms were
    // that tail)
    constexpr int CORR_STG = NW == 1 ? STG1 : (TS == 1 ? 1024 : GSS_CORR_STG_ELEMS) / NT;
    const int total = frames_lds * D;
    cplx stg[CORR_STG];
    double stg_w = 0.0;
    auto stage_load = [&](int64_t t0) {
        const int64_t fr0 = t0 - c;
#pragma unroll
        for (int s = 0; s < CORR_STG; ++s) {
            const int idx = threadIdx.x + NT * s;
            const int64_t fr = fr0 + idx / D;
            stg[s] = c_make(0.0, 0.0);
            if (idx < total && fr >= 0 && fr < T) stg[s] = Yf[fr0 * D + idx];
        }
        if (threadIdx.x < CORR_KT) stg_w = (t0 + threadIdx.x < T) ? wf[t0 + threadIdx.x] : 0.0;
    };
    auto stage_store = [&](int64_t t0) {
#pragma unroll
        for (int s = 0; s < CORR_STG; ++s) {
            const int idx = threadIdx.x + NT * s;
            if (idx < total) S[idx] = stg[s];
        }
        if (threadIdx.x < CORR_KT) wS[threadIdx.x] = stg_w;
        const int64_t fr0 = t0 - c;
        for (int idx = threadIdx.x + NT * CORR_STG; idx < total; idx += blockDim.x) {
            const int64_t fr = fr0 + idx / D;
            S[idx] = (fr >= 0 && fr < T) ? Yf[fr0 * D + idx] : c_make(0.0, 0.0);
        }
    };
    stage_load(0);
    for (int64_t t0 = 0; t0 < T; t0 += CORR_KT) {
        __syncthreads();          // every wave is done with the previous chunk
        stage_store(t0);
        __syncthreads();
        if (t0 + CORR_KT < T) stage_load(t0 + CORR_KT);
        if (active) corr_chunk<TS, MASK>(S, wS, D, tl, t1, t2, t3);
    }
    if (active) corr_store<TS, MASK>(tl, f, n, c, D, t1, t2, t3, R, P);
}

template <int TS, int NW, int STG1 = 8>
__global__ __launch_bounds__(64 * NW, (NW == 1 && STG1 < 8) ? 4 : 1) void wpe_corr_kernel(
    const cplx *__restrict__ Y, const double *__restrict__ w, int F, int64_t T, int D, int n,
    int c, int padf, const CorrTile *__restrict__ tiles, int ntiles, cplx *__restrict__ R,
    cplx *__restrict__ P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int frames_lds = CORR_KT + c + padf;
    cplx *S = reinterpret_cast<cplx *>(smem);                      // frames_lds * D
    double *wS = reinterpret_cast<double *>(S + frames_lds * D);   // CORR_KT

    int f, grp;
    if (!xcd_group_map((ntiles + NW - 1) / NW, F, f, grp)) return;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tile_id = grp * NW + wave;
    const bool active = tile_id < ntiles;
    const CorrTile tl = tiles[active ? tile_id : 0];
    const cplx *Yf = Y + (int64_t)f * T * D;
    const double *wf = w + (int64_t)f * T;
    constexpr int FULL = (1 << (TS * TS)) - 1;
    const int mask = __builtin_amdgcn_readfirstlane(active ? tl.mask : 1);
#define CORR_CASE(M) \
    case M: corr_tile_body<TS, M, NW, STG1>(Yf, wf, T, D, n, c, frames_lds, S, wS, tl, active, f, R, P); break
    if (TS == 2) {
        switch (mask) {
            CORR_CASE(1);
            CORR_CASE(3);
            CORR_CASE(5);
            CORR_CASE(11);
            default:
                corr_tile_body<TS, FULL, NW, STG1>(Yf, wf, T, D, n, c, frames_lds, S, wS, tl, active, f, R, P);
        }
    } else {
        corr_tile_body<TS, FULL, NW, STG1>(Yf, wf, T, D, n, c, frames_lds, S, wS, tl, active, f, R, P);
    }
#undef CORR_CASE
}

// ---- few channels (one array): the waves of a workgroup split the FRAMES of one sub-tile
// One array at 10 taps is 6 sub-tiles of 16 x 16 per frequency: 3078 single waves of 34 chunks
// each -- 3 x 1024 + 6.  The dispatcher fills SIMDs up to their limit, so whatever the limit is
// the launch ends on a handful of SIMDs that hold one wave more than the rest (three waves per
// SIMD: the last six waves ran alone in a second round; four: a quarter of the chip idle while
// the rest works through four): wave lifetime 158 us, launch 270 us (profiles/r05b_1a_*).
// Here a workgroup is still ONE sub-tile of one frequency, but its KS waves take a quarter of
// the 64-frame chunks each (own window in LDS, no workgroup barrier in the loop) and the
// partial tiles are added in wave order through LDS at the end (deterministic): 12 312 waves a
// quarter as long, handed out as slots become free -- the SIMDs stay evenly loaded and the tail
// is a quarter-wave.  The sums are blocked (four partial sums) instead of one run over T.
template <int KS, int STG>
__global__ __launch_bounds__(64 * KS, 4) void wpe_corr_ksplit_kernel(
    const cplx *__restrict__ Y, const double *__restrict__ w, int F, int64_t T, int D, int n,
    int c, int padf, const CorrTile *__restrict__ tiles, int ntiles, cplx *__restrict__ R,
    cplx *__restrict__ P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int frames_lds = CORR_KT + c + padf;
    const int total = frames_lds * D;                               // <= 64 * STG
    const size_t wbytes = sizeof(cplx) * (size_t)total + sizeof(double) * CORR_KT;
    int f, tile_id;
    if (!xcd_group_map(ntiles, F, f, tile_id)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    cplx *S = reinterpret_cast<cplx *>(smem + wave * wbytes);
    double *wS = reinterpret_cast<double *>(S + total);
    const CorrTile tl = tiles[tile_id];
    const cplx *Yf = Y + (int64_t)f * T * D;
    const double *wf = w + (int64_t)f * T;
    v4d t1[1][1], t2[1][1], t3[1][1];
    corr_zero<1>(t1, t2, t3);

    const int nchunk = (int)((T + CORR_KT - 1) / CORR_KT);
    const int ch0 = (int)((int64_t)wave * nchunk / KS), ch1 = (int)((int64_t)(wave + 1) * nchunk / KS);
    cplx stg[STG];
    double stg_w = 0.0;
    auto stage_load = [&](int64_t t0) {
        const int64_t fr0 = t0 - c;
#pragma unroll
        for (int s = 0; s < STG; ++s) {
            const int idx = lane + 64 * s;
            const int64_t fr = fr0 + idx / D;
            stg[s] = c_make(0.0, 0.0);
            if (idx < total && fr >= 0 && fr < T) stg[s] = Yf[fr0 * D + idx];
        }
        stg_w = (t0 + lane < T) ? wf[t0 + lane] : 0.0;
    };
    if (ch0 < ch1) stage_load((int64_t)ch0 * CORR_KT);
    for (int ch = ch0; ch < ch1; ++ch) {
        wave_sync();              // this wave is done with the previous chunk's window
#pragma unroll
        for (int s = 0; s < STG; ++s) {
            const int idx = lane + 64 * s;
            if (idx < total) S[idx] = stg[s];
        }
        wS[lane] = stg_w;
        wave_sync();
        if (ch + 1 < ch1) stage_load((int64_t)(ch + 1) * CORR_KT);
        corr_chunk<1, 1>(S, wS, D, tl, t1, t2, t3);
    }
    // re / im of this wave's partial tile; waves 1 ... KS - 1 park theirs in LDS (the windows
    // are idle by then), wave 0 adds them in wave order and stores
    v4d re, im;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        re[r] = t1[0][0][r] + t2[0][0][r];
        im[r] = (t3[0][0][r] - t1[0][0][r]) + t2[0][0][r];
    }
    __syncthreads();
    cplx *red = reinterpret_cast<cplx *>(smem);
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) red[((wave - 1) * 4 + r) * 64 + lane] = c_make(re[r], im[r]);
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int o = 0; o < KS - 1; ++o)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const cplx v = red[(o * 4 + r) * 64 + lane];
                re[r] += v.x;
                im[r] += v.y;
            }
        t1[0][0] = re;
        t3[0][0] = im;
        corr_store<1, 1, true>(tl, f, n, c, D, t1, t1, t3, R, P);
    }
}

// ---- the same tiles with the window brought in by LDS-DMA (`buffer_load_dwordx4 ... lds`)
// Why: on MI355X the f64 MFMA runs on the f64 vector lanes (matrix peak = vector peak), so
// every VALU instruction a wave issues -- the index arithmetic, bounds selects and the
// register -> LDS pass of the staged window above are ~200 of them per 64-frame chunk, next
// to 128 f64 operand multiplications -- is time the SIMD does not spend on the 192 MFMAs of
// that chunk.  The window of a chunk is one contiguous run of the frequency's flat (T, D)
// slab, which is exactly what LDS-DMA copies: a wave instruction moves 64 consecutive
// complex values (1 KiB) to a wave-uniform LDS base + 16 * lane; the buffer resource is
// sized to the slab, so the elements past the last frame arrive as zeros without a compare.
// No staging registers, no ds_write pass, one v_add per piece; two LDS windows alternate so
// that the DMA of chunk i + 1 is in flight during the MFMAs of chunk i and ONE barrier per
// chunk remains.  (Frames before the first one -- only while t0 < c -- are the lanes of a
// piece with a negative element index: they store a zero to their slot instead.)
using lds_void_ptr = __attribute__((address_space(3))) void *;

#ifdef GSS_CORR_TRACE
// tools/corr_trace.py: per workgroup {block, start, first window landed, loop done, stored,
// HW_ID} on the 100 MHz wall clock (build with tools/build_variant.sh NAME -DGSS_CORR_TRACE=1)
__device__ long long g_corr_trace[8192 * 6];
__device__ long long g_corr_phase[8192 * 4];     // persistent kernel, wave 0: cycles per phase
extern "C" int gss_debug_corr_trace(long long *host, int entries) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_corr_trace), sizeof(long long) * 6 * entries);
}
extern "C" int gss_debug_corr_phase(long long *host, int entries, int reset) {
    int rc = (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_corr_phase), sizeof(long long) * 4 * entries);
    if (reset) {
        void *p = nullptr;
        (void)hipGetSymbolAddress(&p, HIP_SYMBOL(g_corr_phase));
        (void)hipMemset(p, 0, sizeof(long long) * 4 * 8192);
    }
    return rc;
}
#define CORR_TRACE(slot)                                                                  \
    if (threadIdx.x == 0 && blockIdx.x < 8192) g_corr_trace[blockIdx.x * 6 + (slot)] = wall_clock64()
#else
#define CORR_TRACE(slot)
#endif

// ---- persistent form: resident workgroups pull (frequency, tile group) items from queues
// What the workgroup timeline of the one-workgroup-per-item launch shows (tools/corr_trace.py,
// config 2: 5643 items of ~175 us on 512 resident slots): 10 % of the slot time is the tail of
// the launch (items are 1/11 of a slot's whole share, so the slots run dry over the last ~1.7
// item lengths), 3.6 % the gap between an item's end and the dispatch of the next workgroup
// into its slot, 2 % the first window of every item arriving from memory with nothing to do.
// Here the launch is `slots` workgroups that stay resident and take items from per-XCD queues
// (a frequency's items stay on the XCD whose L2 holds its slab; an empty queue steals from the
// next XCD's): the heavy tile groups of all frequencies first, the light ones (fewer needed
// 16 x 16 sub-tiles per wave) last, so that the launch ends on short items; the window DMA
// runs on across items -- the first window of item i + 1 is requested during the last chunk of
// item i -- and the next item's index is fetched (one returning atomic by one thread) two
// items ahead.
struct CorrQueue {
    int ngroups, gh;      // tile groups per frequency; the first gh are "heavy"
};

__device__ __forceinline__ int corr_queue_len(int F, int x) { return (F - x + 7) / 8; }

// item q of queue x -> f * 4096 + group: the heavy tile groups of all of the XCD's frequencies
// first, then the light ones (fewer needed sub-tiles per wave), so that the launch ends on
// short items.  (The light groups of a frequency then run long after its heavy ones and its
// slab is fetched into the L2 3.2 times per launch; queue orders that keep a frequency's
// groups together -- blocks of 4 - 16 frequencies, or strictly frequency-major -- bring the
// fetch from 597 MB down to 453 - 589 MB and cost 1.5 - 2.7 % of the launch: measured in
// round 4, tools/experiments/corr_fmajor_ab.sh, not kept.)
__device__ __forceinline__ int corr_queue_item(int F, int x, int q, const CorrQueue cq) {
    const int heavy = corr_queue_len(F, x) * cq.gh;
    if (q < heavy) return (x + 8 * (q / cq.gh)) * 4096 + q % cq.gh;
    q -= heavy;
    const int gl = max(cq.ngroups - cq.gh, 1);
    return (x + 8 * (q / gl)) * 4096 + cq.gh + q % gl;
}

// next item for this workgroup (thread 0 only): own XCD's queue first, then the others;
// -1 when every queue is drained.  `live` remembers the first queue that may still hold work.
__device__ __forceinline__ int corr_fetch(int *counters, int F, int xcd, int &live, const CorrQueue cq) {
    for (; live < 8; ++live) {
        const int x = (xcd + live) & 7;
        const int len = corr_queue_len(F, x) * cq.ngroups;
        const int q = atomicAdd(counters + 16 * x, 1);
        if (q < len) return corr_queue_item(F, x, q, cq);
    }
    return -1;
}

// BLOCKED (GSS_CORR_BLOCKED=1): the sums of every 64-frame chunk start from zero and are added
// to the totals (re, im: two accumulators per tile next to the three of the 3M form, +64
// registers) when the chunk is done -- a sum over T frames is then 64 + T / 64 roundings deep
// instead of T.  The f64 MFMA rounds after every frame; BLAS, which the reference's einsum
// runs on, sums in blocks: this is the factor 1.8 - 2 between this path and the oracle in
// their distances from an extended-precision WPE (tests: ..._extended_precision).
template <int MASK, int NW, bool BLOCKED, class Issue>
__device__ __forceinline__ void corr_item_dma(
    int64_t T, int D, int n, int c, int win, cplx *S0, double *w0, const CorrTile tl, bool active,
    int f, int f_next, int &b, Issue &issue, int *ring_slot, int fetched, cplx *__restrict__ R,
    cplx *__restrict__ P) {
    constexpr int TS = 2;
    v4d t1[TS][TS], t2[TS][TS], t3[TS][TS];
    corr_zero<TS>(t1, t2, t3);
    v4d sre[BLOCKED ? TS : 1][BLOCKED ? TS : 1], sim[BLOCKED ? TS : 1][BLOCKED ? TS : 1];
    if constexpr (BLOCKED) {
#pragma unroll
        for (int a = 0; a < TS; ++a)
#pragma unroll
            for (int bb = 0; bb < TS; ++bb) sre[a][bb] = sim[a][bb] = (v4d){0.0, 0.0, 0.0, 0.0};
    }
    // One chunk (window b).  SHORT: the last chunk of the frequency when it holds at most
    // CORR_KT - 16 frames -- the k-steps past the last frame multiply the zeros the window and
    // the weights are padded with (4 of 240 k-steps at T = 941, a tenth of the MFMAs of a 3 s
    // utterance), so only its groups of four k-steps (16 frames) that hold frames are run; the
    // sums are the same (the skipped terms are + 0.0).  It is a peeled iteration BEHIND the loop
    // of full chunks: as a branch inside that loop it cost the loop its register allocation
    // (173 -> 286 registers, accumulators copied between VGPRs and AGPRs around the MFMAs).
    auto chunk_iter = [&](int64_t t0, auto short_tag) {
        constexpr bool SHORT = decltype(short_tag)::value;
#ifdef GSS_CORR_TRACE
        const long long ta = clock64();
#endif
        if (!SHORT && t0 + CORR_KT < T)
            issue(f, t0 + CORR_KT, b ^ 1);
        else if (f_next >= 0)
            issue(f_next, 0, b ^ 1);          // the next item's first window
#ifdef GSS_CORR_TRACE
        const long long tb = clock64();
#endif
        if (active) {
            if constexpr (SHORT) {
                const int frames = (int)(T - t0);
                for (int g = 0; 16 * g < frames; ++g)
                    corr_chunk<TS, MASK, 4>(S0 + b * win + 16 * g * D, w0 + b * CORR_KT + 16 * g, D,
                                            tl, t1, t2, t3);
            } else {
                corr_chunk<TS, MASK>(S0 + b * win, w0 + b * CORR_KT, D, tl, t1, t2, t3);
            }
        }
        if constexpr (BLOCKED) if (active) {
#pragma unroll
            for (int a = 0; a < TS; ++a)
#pragma unroll
                for (int bb = 0; bb < TS; ++bb) {
                    if (!((MASK >> (a * TS + bb)) & 1)) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sre[a][bb][r] += t1[a][bb][r] + t2[a][bb][r];
                        sim[a][bb][r] += (t3[a][bb][r] - t1[a][bb][r]) + t2[a][bb][r];
                    }
                    t1[a][bb] = t2[a][bb] = t3[a][bb] = (v4d){0.0, 0.0, 0.0, 0.0};
                }
        }
        // the item fetched at the start of this one (its atomic has had a chunk to return):
        // into the ring slot of the previous item, published by the barrier below
        if (t0 == 0 && threadIdx.x == 0) *ring_slot = fetched;
#ifdef GSS_CORR_TRACE
        const long long tc = clock64();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long td = clock64();
#endif
        __syncthreads();       // window b ^ 1 has landed (vmcnt(0) of every wave) and b is free
#ifdef GSS_CORR_TRACE
        if (threadIdx.x == 0 && blockIdx.x < 8192) {
            const long long te = clock64();
            g_corr_phase[blockIdx.x * 4 + 0] += tb - ta;     // issue
            g_corr_phase[blockIdx.x * 4 + 1] += tc - tb;     // MFMA chunk
            g_corr_phase[blockIdx.x * 4 + 2] += td - tc;     // waiting for the DMA
            g_corr_phase[blockIdx.x * 4 + 3] += te - td;     // barrier
        }
#endif
        b ^= 1;
    };
    int64_t t0 = 0;
    for (; T - t0 > CORR_KT - 16; t0 += CORR_KT) chunk_iter(t0, std::false_type{});
    if (t0 < T) chunk_iter(t0, std::true_type{});
    if (active)
    {
        if constexpr (BLOCKED) corr_store<TS, MASK, true>(tl, f, n, c, D, sre, sre, sim, R, P);    // (re, -, im)
        else corr_store<TS, MASK>(tl, f, n, c, D, t1, t2, t3, R, P);
    }
}

template <int NW, bool BLOCKED = false>
__global__ __launch_bounds__(64 * NW) void wpe_corr_persist_kernel(
    const cplx *__restrict__ Y, const double *__restrict__ w, int F, int64_t T, int D, int n,
    int c, int pieces, const CorrTile *__restrict__ tiles, int ntiles, CorrQueue cq,
    int *__restrict__ counters, cplx *__restrict__ R, cplx *__restrict__ P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx *S0 = reinterpret_cast<cplx *>(smem);                       // 2 x pieces * 64
    double *w0 = reinterpret_cast<double *>(S0 + 2 * pieces * 64);   // 2 x CORR_KT
    __shared__ int ring[3];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const int win = pieces * 64;
    // HW_REG_XCC_ID (20), bits 3:0: the XCD this workgroup runs on
    const int xcd = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7;
#ifdef GSS_CORR_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        g_corr_trace[blockIdx.x * 6 + 0] = xcd;
        g_corr_trace[blockIdx.x * 6 + 1] = wall_clock64();
        g_corr_trace[blockIdx.x * 6 + 2] = clock64();
    }
#endif
    int live = 0;
    if (threadIdx.x == 0) {
        ring[0] = corr_fetch(counters, F, xcd, live, cq);
        ring[1] = ring[0] < 0 ? -1 : corr_fetch(counters, F, xcd, live, cq);
        ring[2] = -1;
    }
    __syncthreads();

    // One window.  Inside the slab (all but the first and the last chunk of an item) a piece
    // costs NO vector instruction: the lane part of the address (16 * lane) is one VGPR for
    // the whole kernel, everything else -- slab offset of the piece, LDS destination -- is
    // scalar (soffset, M0).  That matters here because a wave's VALU instruction can only
    // issue between the 64-cycle f64 MFMAs of the other wave on its SIMD: 45 address / compare /
    // select instructions per window kept wave 0 in this routine for 21 % of its life
    // (tools/corr_trace_persist.py).  At the edges of the slab the elements before the first
    // frame are lanes that store a zero instead, the elements past the last frame read as
    // zeros through the raw buffer resource sized to the slab.
    const uint32_t lane16 = lane * 16;
    auto issue = [&](int fq, int64_t t0, int bq) {
        const __amdgpu_buffer_rsrc_t rsrc_y = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<cplx *>(Y + (int64_t)fq * T * D), 0, (int)(T * D * 16), 0x00020000);
        cplx *Sb = S0 + bq * win;
        const int64_t g0 = (t0 - c) * (int64_t)D;
        if (g0 >= 0 && g0 + win <= T * (int64_t)D) {
            for (int p = wave; p < pieces; p += NW)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (lds_void_ptr)(Sb + p * 64), 16,
                                                         lane16, (int)((g0 + p * 64) * 16), 0, 0);
        } else {
            for (int p = wave; p < pieces; p += NW) {
                const int64_t gp = g0 + p * 64;            // wave-uniform
                const uint32_t voff = (uint32_t)((gp + lane) * 16);
                if (gp >= 0) {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (lds_void_ptr)(Sb + p * 64),
                                                             16, voff, 0, 0, 0);
                } else if (gp + lane < 0) {
                    Sb[p * 64 + lane] = c_make(0.0, 0.0);
                } else {
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_y, (lds_void_ptr)(Sb + p * 64),
                                                             16, voff, 0, 0, 0);
                }
            }
        }
        if (wave == NW - 1 && lane < CORR_KT / 2) {
            // (weights past the last frame meet window elements that are zero)
            const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<double *>(w + (int64_t)fq * T), 0, (int)(T * 8), 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (lds_void_ptr)(w0 + bq * CORR_KT), 16,
                                                     lane16, (int)(t0 * 8), 0, 0);
        }
    };

    int item = __builtin_amdgcn_readfirstlane(ring[0]);
    if (item < 0) return;
    int b = 0;
    issue(item >> 12, 0, 0);
    __syncthreads();
    for (int k = 0; item >= 0; ++k) {
        const int next = __builtin_amdgcn_readfirstlane(ring[(k + 1) % 3]);
        int fetched = -1;
        if (threadIdx.x == 0 && next >= 0) fetched = corr_fetch(counters, F, xcd, live, cq);
        const int f = item >> 12, grp = item & 4095;
        const int tile_id = grp * NW + wave;
        const bool active = tile_id < ntiles;
        const CorrTile tl = tiles[active ? tile_id : 0];
        const int mask = __builtin_amdgcn_readfirstlane(active ? tl.mask : 1);
        // (ring[(k + 2) % 3] was the slot of item k - 1: nobody reads it any more)
        int *ring_slot = &ring[(k + 2) % 3];
#define CORR_CASE(M)                                                                              \
    case M:                                                                                       \
        corr_item_dma<M, NW, BLOCKED>(T, D, n, c, win, S0, w0, tl, active, f, next < 0 ? -1 : next >> 12, \
                                 b, issue, ring_slot, fetched, R, P);                             \
        break
        switch (mask) {
            CORR_CASE(1);
            CORR_CASE(3);
            CORR_CASE(5);
            CORR_CASE(11);
            default:
                corr_item_dma<15, NW, BLOCKED>(T, D, n, c, win, S0, w0, tl, active, f,
                                          next < 0 ? -1 : next >> 12, b, issue, ring_slot, fetched,
                                          R, P);
        }
#undef CORR_CASE
        item = next;
#ifdef GSS_CORR_TRACE
        if (threadIdx.x == 0 && blockIdx.x < 8192) {
            g_corr_trace[blockIdx.x * 6 + 3] = wall_clock64();
            g_corr_trace[blockIdx.x * 6 + 4] = clock64();
            g_corr_trace[blockIdx.x * 6 + 5] = k + 1;
        }
#endif
    }
}

// ------------------------------------------------------------------ solve
// G = solve(R, P) for every frequency: blocked right-looking Cholesky R = U^H U on
// the upper triangle with the right-hand sides carried along ([R | P] -> [U | Z],
// Z = U^-H P), then blocked back substitution U G = Z (G overwrites P).
// Per block column J of CH_NB = 48 rows:
//   chol_diag    grid (F), 256 threads: factor the 48 x 48 diagonal block and invert it in
//                                      one register-resident sweep: W = U_JJ^-H (lower
//                                      triangular) is kept in the unused strictly-lower
//                                      triangle
//   chol_trsm    XCD grid (chunks x F): row panel  U_J = W A_J  (and Z_J = W P_J)
//   chol_update  XCD grid (tiles x F): trailing update C -= U_J^H U_J with the f64
//                                      MFMA, one 16 x 16 tile per wave, accumulators
//                                      loaded from / stored to global memory in
//                                      fragment layout, operands double-buffered; block
//                                      columns in pairs (see wpe_run): after an even block
//                                      only the next block row, after an odd one everything
//                                      below with both panels (K = 96)
//   chol_backsolve grid (F):           G_J = W^H (Z_J - U_J,>J G_>J), J descending
// A non-positive pivot (exactly singular system, e.g. an all-zero channel) zeroes
// that row, which reproduces the minimum-norm lstsq fallback of stable_solve
// (pb_chime5/math/solve.py:95-114) for zero rows / columns.
#ifndef GSS_UPD_PREFETCH
#define GSS_UPD_PREFETCH 1     // k-steps of operands in flight beyond the next one (chol_update)
#endif
constexpr int CH_NB = 48;
constexpr int UD_LD = CH_NB + 1;   // LDS leading dimension of the diagonal block (bank spread)

// Factor the diagonal block at j0 of one frequency's matrix A (256 threads): U_JJ (upper)
// and the inverses W_tt = U_tt^-H of its three 16 x 16 diagonal blocks (strictly lower part
// of those blocks; nothing else of U_JJ^-H is needed by the 16-blocked substitutions of the
// panel kernels) are written back to A, the diagonal as U_ii.  The block lives in registers,
// 3 x 3 entries per thread, and one pass of nb steps builds U and W together
// (chol_inverse_sweep in dense_wave.h, two-row ring in LDS): Ud = 2 * UD_LD complex,
// dinv = CH_NB doubles.
constexpr size_t DIAG_LDS = sizeof(cplx) * 2 * UD_LD + sizeof(double) * CH_NB;

__device__ inline void chol_diag_block(cplx *A, int n, int j0, cplx *Ud, double *dinv,
                                       int32_t *zero_pivots) {
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int nb = min(CH_NB, n - j0);
    cplx reg[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int i = ty + 16 * a, k = tx + 16 * b;
            reg[a][b] = (k >= i && k < nb) ? A[(int64_t)(j0 + i) * n + j0 + k] : c_make(0.0, 0.0);
        }
    if (!chol_inverse_sweep<16, 3, false, true>(reg, nb, Ud, UD_LD, dinv, tx, ty)) {
        // rows with a non-positive pivot were zeroed (the lstsq branch of stable_solve):
        // counted for gss_last_wpe_zero_pivots()
        if (tid < nb && dinv[tid] == 0.0) atomicAdd(zero_pivots, 1);
    }
    // the registers hold the unscaled rows: scale them, fix the diagonal
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const int i = ty + 16 * a;
        const double di = i < nb ? dinv[i] : 0.0;
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int k = tx + 16 * b;
            if (b < a || i >= nb || k >= nb) continue;       // off-diagonal W blocks: unused
            const cplx raw = reg[a][b];
            A[(int64_t)(j0 + i) * n + j0 + k] =
                k == i ? c_make(di > 0.0 ? sqrt(raw.x) : 0.0, 0.0) : c_scale(raw, di);
        }
    }
}

__global__ __launch_bounds__(256) void chol_diag_kernel(cplx *__restrict__ R, int n, int j0,
                                                        int32_t *__restrict__ zero_pivots) {
    __shared__ __attribute__((aligned(16))) char smem[DIAG_LDS];
    cplx *Ud = reinterpret_cast<cplx *>(smem);                       // 2 * UD_LD
    double *dinv = reinterpret_cast<double *>(Ud + 2 * UD_LD);       // CH_NB
    chol_diag_block(R + (int64_t)blockIdx.x * n * n, n, j0, Ud, dinv, zero_pivots);
}

// LDS copy of a factored diagonal block for the row panel: block row b (16 rows) keeps its
// columns from 16 b on -- U above the diagonal and, inside the three diagonal 16 x 16 blocks, W
// below it; the three blocks under the block diagonal are never read.  96 instead of 144 rows of
// 16: 25.7 KB with the reciprocal diagonal instead of 38 KB, i.e. FIVE workgroups per CU where
// LDS allowed four -- and four are 1024 slots for the 1026 workgroups (2 x 513) that the third
// and the fourth block column of the 24-channel solve launch: a second round for two of them.
// Odd leading dimensions (49, 33, 17 complex): the 16 lanes of a fragment row hit 16 bank groups.
constexpr int UDP_SIZE = 16 * (CH_NB + 1) + 16 * (CH_NB - 16 + 1) + 16 * (CH_NB - 32 + 1);
__device__ __forceinline__ int udp_off(int row, int col) {
    const int b = row >> 4;
    const int base = b == 0 ? 0 : b == 1 ? 16 * (CH_NB + 1) : 16 * (CH_NB + 1) + 16 * (CH_NB - 16 + 1);
    return base + (row & 15) * (CH_NB - 16 * b + 1) + col - 16 * b;
}

// Row panel on the MFMA: U_J[:, tile] = U_JJ^-H A_J[:, tile] for one tile `ct` of 16 trailing
// columns (tiles past the trailing block address the right-hand sides), by one wave.
// Forward substitution in 16-row blocks with the EXPLICIT INVERSES OF THE 16 x 16 DIAGONAL
// BLOCKS only (L = U_JJ^H, W_ii = L_ii^-1 from the sweep):
//     X_0 = W_00 B_0,   X_1 = W_11 (B_1 - L_10 X_0),   X_2 = W_22 (B_2 - L_20 X_0 - L_21 X_1)
// -- the same 24 k-steps as the product with the explicit 48 x 48 inverse that rounds 1-2
// used, but backward stable like a triangular solve: the explicit 48 x 48 inverse costs a
// factor cond(U_JJ) (3x the error of LAPACK on the bench workload, amplified 30x per WPE
// iteration by the power weights; tests/golden/make_wpe_truth.py), a 16 x 16 one does not
// show.  No data movement between the stages: the C fragment of v_mfma_f64_16x16x4_f64
// (row = lk + 4 reg) IS the B operand sequence of the next product (k-step ks takes rows
// 4 ks + lk, i.e. register ks).
// The diagonal block comes from the LDS copy Ud: upper part U_JJ, strictly lower part W
// (only its diagonal 16 x 16 blocks are used), 1 / U_ii from dinv.
__device__ inline void chol_panel_tile(cplx *A, cplx *Z, int n, int D, int j0, int nb, int ct,
                                       const cplx *Ud, const double *dinv, int lane) {
    const int ntrail = n - j0 - nb;
    const int li = lane & 15, lk = lane >> 4;
    const int nct_a = (ntrail + 15) / 16;
    cplx *base;
    int64_t stride;
    int ncols;
    if (ct < nct_a) {
        base = A + (int64_t)j0 * n + j0 + nb + 16 * ct;
        stride = n;
        ncols = min(16, ntrail - 16 * ct);
    } else {
        base = Z + (int64_t)j0 * D + 16 * (ct - nct_a);
        stride = D;
        ncols = min(16, D - 16 * (ct - nct_a));
    }
    cplx bv[CH_NB / 4];
#pragma unroll
    for (int ks = 0; ks < CH_NB / 4; ++ks) {
        const int row = 4 * ks + lk;
        bv[ks] = (row < nb && li < ncols) ? base[row * stride + li] : c_make(0.0, 0.0);
    }
    v4d xre[3], xim[3];
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        // T = B_it - sum_{jt < it} L_it,jt X_jt   (L[i][k] = conj(U[k][i]))
        v4d tre, tim;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            tre[r] = bv[4 * it + r].x;
            tim[r] = bv[4 * it + r].y;
        }
        const int i = 16 * it + li;
#pragma unroll
        for (int jt = 0; jt < it; ++jt)
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int k = 16 * jt + 4 * ks + lk;
                const cplx u = Ud[udp_off(k, i)];
                // T -= conj(u) x
                tre = __builtin_amdgcn_mfma_f64_16x16x4f64(-u.x, xre[jt][ks], tre, 0, 0, 0);
                tim = __builtin_amdgcn_mfma_f64_16x16x4f64(-u.x, xim[jt][ks], tim, 0, 0, 0);
                tre = __builtin_amdgcn_mfma_f64_16x16x4f64(-u.y, xim[jt][ks], tre, 0, 0, 0);
                tim = __builtin_amdgcn_mfma_f64_16x16x4f64(u.y, xre[jt][ks], tim, 0, 0, 0);
            }
        // X_it = W_ii T
        xre[it] = (v4d){0.0, 0.0, 0.0, 0.0};
        xim[it] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int k = 16 * it + 4 * ks + lk;
            cplx w = Ud[udp_off(i, k)];
            if (k == i) w = c_make(i < nb ? dinv[i] : 0.0, 0.0);
            if (k > i) w = c_make(0.0, 0.0);
            xre[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(w.x, tre[ks], xre[it], 0, 0, 0);
            xim[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(w.x, tim[ks], xim[it], 0, 0, 0);
            xre[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(-w.y, tim[ks], xre[it], 0, 0, 0);
            xim[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(w.y, tre[ks], xim[it], 0, 0, 0);
        }
    }
#pragma unroll
    for (int it = 0; it < 3; ++it)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * it + lk + 4 * r;
            if (row < nb && li < ncols) base[row * stride + li] = c_make(xre[it][r], xim[it][r]);
        }
}

// The factored diagonal block (U_JJ above, W below the diagonal) -> LDS, 1 / U_ii -> dinv, by
// 256 threads: all 9 loads of a thread in flight before the first store.
__device__ inline void chol_stage_diag(const cplx *A, int n, int j0, int nb, cplx *Ud, double *dinv) {
    const int tid = threadIdx.x;
    constexpr int WL = CH_NB * CH_NB / 256;
    static_assert(WL * 256 == CH_NB * CH_NB, "block size");
    cplx wv[WL];
#pragma unroll
    for (int s = 0; s < WL; ++s) {
        const int idx = tid + 256 * s;
        const int i = idx / CH_NB, k = idx - i * CH_NB;
        wv[s] = c_make(0.0, 0.0);
        if (i < nb && k < nb && k >= (i & ~15)) wv[s] = A[(int64_t)(j0 + i) * n + j0 + k];
    }
#pragma unroll
    for (int s = 0; s < WL; ++s) {
        const int idx = tid + 256 * s;
        const int i = idx / CH_NB, k = idx - i * CH_NB;
        if (k >= (i & ~15)) Ud[udp_off(i, k)] = wv[s];
        if (k == i) dinv[i] = wv[s].x > 0.0 ? 1.0 / wv[s].x : 0.0;
    }
}

// grid (ceil(tiles / 4), F), block 256: every workgroup stages the factored diagonal
// block of its frequency in LDS, then each wave takes one column tile.
__global__ __launch_bounds__(256) void chol_trsm_kernel(cplx *__restrict__ R,
                                                        cplx *__restrict__ P, int F, int n,
                                                        int D, int j0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx *Ud = reinterpret_cast<cplx *>(smem);                       // UDP_SIZE (udp_off)
    double *dinv = reinterpret_cast<double *>(Ud + UDP_SIZE);        // CH_NB
    const int tid = threadIdx.x;
    const int nb = min(CH_NB, n - j0);
    const int ntrail = n - j0 - nb;
    const int npanel = (ntrail + 15) / 16 + (D + 15) / 16;
    int f, grp;       // the workgroups of one frequency share one XCD (one L2)
    if (!xcd_group_map((npanel + 3) / 4, F, f, grp)) return;
    cplx *A = R + (int64_t)f * n * n;
    cplx *Z = P + (int64_t)f * n * D;
    chol_stage_diag(A, n, j0, nb, Ud, dinv);
    __syncthreads();
    const int ct = grp * 4 + (tid >> 6);
    if (ct < npanel) chol_panel_tile(A, Z, n, D, j0, nb, ct, Ud, dinv, tid & 63);
}

struct UpdTile {
    int row_off, col_off, is_p, pad;
};

// Trailing update of one (16 TM) x (16 TN) tile by one wave:  C -= U_J^H U_J.
template <int TM, int TN, bool PREFETCH>
__device__ inline void chol_update_tile(cplx *A, cplx *Z, int n, int D, int j0, int nb,
                                        const UpdTile tl, int lane) {
    const int li = lane & 15, lk = lane >> 4;
    const cplx *panel = A + (int64_t)j0 * n;      // rows j0 .. j0+nb of U
    const cplx *zpanel = Z + (int64_t)j0 * D;
    const int ncols = tl.is_p ? D : n;

    auto load_ops = [&](int ks, cplx (&a)[TM], cplx (&b)[TN]) {
        const int kk = 4 * ks + lk;
        const bool kv = kk < nb;
#pragma unroll
        for (int m = 0; m < TM; ++m) {
            const int ri = tl.row_off + 16 * m + li;
            a[m] = (kv && ri < n) ? panel[(int64_t)kk * n + ri] : c_make(0.0, 0.0);
        }
#pragma unroll
        for (int m = 0; m < TN; ++m) {
            const int ci = tl.col_off + 16 * m + li;
            b[m] = c_make(0.0, 0.0);
            if (kv && ci < ncols)
                b[m] = tl.is_p ? zpanel[(int64_t)kk * D + ci] : panel[(int64_t)kk * n + ci];
        }
    };

    const int ksteps = (nb + 3) / 4;
    cplx a_cur[TM], b_cur[TN];
    load_ops(0, a_cur, b_cur);

    // C -= conj(a) b with THREE real MFMAs per complex product, as in the correlation and the
    // filter: t1 = sum ar br, t2 = sum ai bi, t3 = sum (ar + ai)(br - bi) accumulate from zero,
    // re(conj(a) b) = t1 + t2, im = t1 - t2 - t3 are taken off C once at the end (the tile of C
    // is requested up front and first touched after the loop).  A quarter fewer MFMAs in the
    // kernel that carries most of the factorisation's flops.
    cplx cv[TM][TN][4];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = tl.row_off + 16 * a + lk + 4 * reg;
                const int c = tl.col_off + 16 * b + li;
                cv[a][b][reg] = c_make(0.0, 0.0);
                if (r < n && c < ncols) cv[a][b][reg] = tl.is_p ? Z[(int64_t)r * D + c] : A[(int64_t)r * n + c];
            }
    v4d t1[TM][TN], t2[TM][TN], t3[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b) {
            t1[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
            t2[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
            t3[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
        }
    // operands 1 + PD k-steps ahead in a register ring (one k-step ahead was not enough: an L2
    // round trip is three k-steps of MFMA time for one wave; J1 at config 2 with 1 / 2 / 3 / 4 /
    // 6 k-steps ahead: 153.8 / 147.6 / 168.5 / 173.3 / 193.4 us -- the ring's register moves
    // and the occupancy cost more than the latency beyond two)
    constexpr int PD = GSS_UPD_PREFETCH;
    cplx a_ring[PD][TM], b_ring[PD][TN];
#pragma unroll
    for (int p = 0; p < PD; ++p)
        if (PREFETCH && p + 1 < ksteps) load_ops(p + 1, a_ring[p], b_ring[p]);
    for (int ks = 0; ks < ksteps; ++ks) {
        cplx a_new[TM], b_new[TN];
        if (PREFETCH) {
            if (ks + PD + 1 < ksteps) load_ops(ks + PD + 1, a_new, b_new);
        } else if (ks > 0) {
            load_ops(ks, a_cur, b_cur);
        }
        double as[TM], bd[TN];
#pragma unroll
        for (int m = 0; m < TM; ++m) as[m] = a_cur[m].x + a_cur[m].y;
#pragma unroll
        for (int m = 0; m < TN; ++m) bd[m] = b_cur[m].x - b_cur[m].y;
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int b = 0; b < TN; ++b) {
                t1[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[a].x, b_cur[b].x, t1[a][b], 0, 0, 0);
                t2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(a_cur[a].y, b_cur[b].y, t2[a][b], 0, 0, 0);
                t3[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(as[a], bd[b], t3[a][b], 0, 0, 0);
            }
        if (PREFETCH) {
#pragma unroll
            for (int m = 0; m < TM; ++m) {
                a_cur[m] = a_ring[0][m];
#pragma unroll
                for (int p = 0; p + 1 < PD; ++p) a_ring[p][m] = a_ring[p + 1][m];
                a_ring[PD - 1][m] = a_new[m];
            }
#pragma unroll
            for (int m = 0; m < TN; ++m) {
                b_cur[m] = b_ring[0][m];
#pragma unroll
                for (int p = 0; p + 1 < PD; ++p) b_ring[p][m] = b_ring[p + 1][m];
                b_ring[PD - 1][m] = b_new[m];
            }
        }
    }
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int r = tl.row_off + 16 * a + lk + 4 * reg;
                const int c = tl.col_off + 16 * b + li;
                if (r < n && c < ncols) {
                    const cplx v = c_make(cv[a][b][reg].x - (t1[a][b][reg] + t2[a][b][reg]),
                                          cv[a][b][reg].y - ((t1[a][b][reg] - t2[a][b][reg]) - t3[a][b][reg]));
                    if (tl.is_p) Z[(int64_t)r * D + c] = v;
                    else A[(int64_t)r * n + c] = v;
                }
            }
}

// XCD-mapped 1-D grid over (tile groups, F), block 256 = 4 waves, one tile each.
// The first `ndiag` tiles of the list are those of the NEXT diagonal block (j0 + K rows
// onward, 48 x 48: up to 6 tiles).  Group 0 of every frequency takes all of them and then
// factors that block right here (chol_diag_block, all 256 threads) while the other groups
// are still updating the rest of the trailing matrix: the 35 us latency chain of the
// diagonal sweep is hidden under the update instead of being a launch of its own between
// two launches that wait for it (4 of the 5 chol_diag launches per solve, 0.4 ms per
// utterance at config 2).
constexpr int UPD_WAVES = 4;    // tiles per workgroup (8 waves: the 6 diagonal tiles in one round, but the bulk runs 7 % slower)

template <int TM, int TN, bool PREFETCH>
__global__ __launch_bounds__(64 * UPD_WAVES) void chol_update_kernel(cplx *__restrict__ R,
                                                          cplx *__restrict__ P, int F, int n,
                                                          int D, int j0, int nb,
                                                          const UpdTile *__restrict__ tiles,
                                                          int ntiles, int ndiag, int j0_next,
                                                          int32_t *__restrict__ zero_pivots) {
    __shared__ __attribute__((aligned(16))) char smem[DIAG_LDS];
    // 1-D XCD-mapped grid: the tiles of one frequency run on one XCD, so its panel is
    // fetched into one L2 instead of all eight.
    // Block ids [0, F8) are the diagonal groups of all frequencies (dispatched FIRST: their
    // latency chain must start at the beginning of the launch, not whenever the dispatcher
    // reaches them), the rest the ordinary groups; block L of either part runs on XCD L % 8
    // = f % 8.
    const int F8 = ndiag > 0 ? (F + 7) / 8 * 8 : 0;
    const bool diag_group = (int)blockIdx.x < F8;
    int f, grp;
    if (diag_group) {
        f = blockIdx.x;
        grp = 0;
        if (f >= F) return;
    } else {
        const int L = blockIdx.x - F8, nsub = (ntiles - ndiag + UPD_WAVES - 1) / UPD_WAVES;
        const int sg = L / (8 * nsub), rem = L - sg * 8 * nsub;
        grp = rem >> 3;
        f = sg * 8 + (rem & 7);
        if (f >= F) return;
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    cplx *A = R + (int64_t)f * n * n, *Z = P + (int64_t)f * n * D;
    if (diag_group) {
        __builtin_amdgcn_s_setprio(3);     // the critical path of the launch
        for (int t = wave; t < ndiag; t += UPD_WAVES)
            chol_update_tile<TM, TN, PREFETCH>(A, Z, n, D, j0, nb, tiles[t], lane);
        __syncthreads();      // the block is complete and visible to the whole workgroup
        if (wave >= 4) return;      // the sweep is a 256-thread routine
        cplx *Ud = reinterpret_cast<cplx *>(smem);                       // 2 * UD_LD
        double *dinv = reinterpret_cast<double *>(Ud + 2 * UD_LD);       // CH_NB
        chol_diag_block(A, n, j0_next, Ud, dinv, zero_pivots);
        return;
    }
    const int tile_id = ndiag + grp * UPD_WAVES + wave;
    if (tile_id >= ntiles) return;
    chol_update_tile<TM, TN, PREFETCH>(A, Z, n, D, j0, nb, tiles[tile_id], lane);
}

// Blocked back substitution U G = Z, G overwrites Z:  G_J = U_JJ^-1 (Z_J - U_J,>J G_>J),
// J descending, on the f64 MFMA.  grid (F), block 256.  Per block column: waves 0..2 own
// one 16-row tile of S = Z_J - U_J,>J G_>J each (2 column tiles = 32 right-hand sides per
// pass), so every frequency reads its U exactly once; wave 3 meanwhile parks the diagonal
// block in LDS as ready-made A operands (6 blocks of 16 x 16 in fragment order).  Then waves
// 3 and 2 take one column tile each through the 16-blocked back substitution with the
// explicit inverses of the 16 x 16 diagonal blocks only (see chol_panel_tile: backward
// stable, same MFMA count as the product with the explicit 48 x 48 inverse):
//     X_2 = W_22^H S_2,  X_1 = W_11^H (S_1 - U_12 X_2),  X_0 = W_00^H (S_0 - U_01 X_1 - U_02 X_2)
constexpr int BS_LD = 33;   // padded leading dimension of the LDS copy of S
constexpr int BS_OPS = 6 * 4 * 64;   // A operands of the substitution: [block][k-step][lane]

__device__ inline void chol_backsolve_body(const cplx *A, cplx *Z, int n, int D, cplx *S,
                                           cplx *OP) {
    // (wave index in an SGPR: the roles below are separate paths of the control flow graph,
    // so the registers of one do not stay allocated across the other)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int nblk = (n + CH_NB - 1) / CH_NB;
    const int a = wave;                       // row tile of this wave (waves 0..2)
    for (int c0 = 0; c0 < D; c0 += 32) {      // 32 right-hand sides per pass
        for (int J = nblk - 1; J >= 0; --J) {
            const int j0 = J * CH_NB, nb = min(CH_NB, n - j0);
            if (a == 3) {
                // A operands (lane li = row i within the tile, lk = k within the k-step):
                //   blocks 0..2: (W_tt^H)[i][k] = conj(W[k][i]) (k > i), 1 / U_ii (k == i), 0 (k < i),
                //                stored NEGATED;  blocks 3..5: U[i][k] of the tile pairs (1,2), (0,1), (0,2)
                const cplx *Ad = A + (int64_t)j0 * n + j0;
                auto fetch = [&](int row, int col) {      // unconditional, clamped
                    const bool ok = row < nb && col < nb;
                    const cplx v = Ad[(int64_t)(ok ? row : 0) * n + (ok ? col : 0)];
                    return c_make(ok ? v.x : 0.0, ok ? v.y : 0.0);
                };
                cplx wop[3][4], uop[3][4];
                double dg[3];
#pragma unroll
                for (int it = 0; it < 3; ++it) {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) wop[it][ks] = fetch(16 * it + 4 * ks + lk, 16 * it + li);
                    dg[it] = fetch(16 * it + li, 16 * it + li).x;
                }
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    uop[0][ks] = fetch(16 + li, 32 + 4 * ks + lk);
                    uop[1][ks] = fetch(li, 16 + 4 * ks + lk);
                    uop[2][ks] = fetch(li, 32 + 4 * ks + lk);
                }
#pragma unroll
                for (int it = 0; it < 3; ++it) {
                    const double ndinv = dg[it] > 0.0 ? -1.0 / dg[it] : 0.0;
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const int k = 4 * ks + lk;
                        const cplx w = wop[it][ks];
                        OP[(it * 4 + ks) * 64 + lane] =
                            c_make(k == li ? ndinv : (k < li ? 0.0 : -w.x), k <= li ? 0.0 : -w.y);
                        OP[((3 + it) * 4 + ks) * 64 + lane] = uop[it][ks];
                    }
                }
            } else {
                // ---- S = Z_J - U_J,>J G_>J  (rows j0 + 16a .., columns c0 .. c0+31)
                v4d acc_re[2], acc_im[2];
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int rl = 16 * a + lk + 4 * reg, col = c0 + 16 * b + li;
                        cplx v = c_make(0.0, 0.0);
                        if (rl < nb && col < D) v = Z[(int64_t)(j0 + rl) * D + col];
                        acc_re[b][reg] = v.x;
                        acc_im[b][reg] = v.y;
                    }
                const int kbeg = j0 + nb;
                const int ksteps = (n - kbeg + 3) / 4;
                const int arow = 16 * a + li;
                auto load_ops = [&](int ks, cplx &u, cplx (&g)[2]) {
                    const int kk = kbeg + 4 * ks + lk;
                    const bool kv = kk < n;
                    u = (kv && arow < nb) ? A[(int64_t)(j0 + arow) * n + kk] : c_make(0.0, 0.0);
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const int col = c0 + 16 * b + li;
                        g[b] = (kv && col < D) ? Z[(int64_t)kk * D + col] : c_make(0.0, 0.0);
                    }
                };
                // One workgroup per frequency and 1.5 waves per SIMD: nothing hides the operand
                // latency but the wave itself, so the operands run PD k-steps ahead in a ring
                // (static slots: the k loop advances PD steps per trip; steps past the end load
                // zeros and add nothing).
                constexpr int PD = 6;
                cplx u_r[PD], g_r[PD][2];
#pragma unroll
                for (int p = 0; p < PD; ++p) load_ops(p, u_r[p], g_r[p]);
                for (int ks0 = 0; ks0 < ksteps; ks0 += PD) {
#pragma unroll
                    for (int p = 0; p < PD; ++p) {
                        // acc -= u * g
                        const double nur = -u_r[p].x, nui = -u_r[p].y, ui = u_r[p].y;
                        const double g0r = g_r[p][0].x, g0i = g_r[p][0].y;
                        const double g1r = g_r[p][1].x, g1i = g_r[p][1].y;
                        load_ops(ks0 + p + PD, u_r[p], g_r[p]);
                        __builtin_amdgcn_sched_barrier(0);
                        acc_re[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(nur, g0r, acc_re[0], 0, 0, 0);
                        acc_im[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(nur, g0i, acc_im[0], 0, 0, 0);
                        acc_re[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(nur, g1r, acc_re[1], 0, 0, 0);
                        acc_im[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(nur, g1i, acc_im[1], 0, 0, 0);
                        acc_re[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ui, g0i, acc_re[0], 0, 0, 0);
                        acc_im[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(nui, g0r, acc_im[0], 0, 0, 0);
                        acc_re[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ui, g1i, acc_re[1], 0, 0, 0);
                        acc_im[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(nui, g1r, acc_im[1], 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg)
                        S[(16 * a + lk + 4 * reg) * BS_LD + 16 * b + li] =
                            c_make(acc_re[b][reg], acc_im[b][reg]);
            }
            __syncthreads();
            const int b = 3 - a;                 // column tile of waves 3 and 2
            if (a >= 2 && c0 + 16 * b < D) {
                // ---- G_J = U_JJ^-1 S, rows 32.., 16.., 0..
                // The MFMA only adds and a complex product needs one subtraction: the operands
                // are stored so that it falls on the B side (4 sign flips per stage).
                // Tn = -(S - U X):  Tn += u x ;  X = -conj(w) Tn, with p = -w stored
                v4d xre[3], xim[3], nxim[3];
#pragma unroll
                for (int st = 0; st < 3; ++st) {
                    const int it = 2 - st;
                    v4d tre, tim;
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const cplx sv = S[(16 * it + lk + 4 * reg) * BS_LD + 16 * b + li];
                        tre[reg] = -sv.x;
                        tim[reg] = -sv.y;
                    }
#pragma unroll
                    for (int sj = 0; sj < st; ++sj)
#pragma unroll
                        for (int ks = 0; ks < 4; ++ks) {
                            const int jt = 2 - sj;
                            const cplx u = OP[((3 + (it == 1 ? 0 : jt)) * 4 + ks) * 64 + lane];
                            tre = __builtin_amdgcn_mfma_f64_16x16x4f64(u.x, xre[jt][ks], tre, 0, 0, 0);
                            tim = __builtin_amdgcn_mfma_f64_16x16x4f64(u.x, xim[jt][ks], tim, 0, 0, 0);
                            tre = __builtin_amdgcn_mfma_f64_16x16x4f64(u.y, nxim[jt][ks], tre, 0, 0, 0);
                            tim = __builtin_amdgcn_mfma_f64_16x16x4f64(u.y, xre[jt][ks], tim, 0, 0, 0);
                        }
                    // X_it = p (conj applied) Tn:  re = px tr + py ti,  im = px ti - py tr
                    const v4d ntre = -tre;
                    xre[it] = (v4d){0.0, 0.0, 0.0, 0.0};
                    xim[it] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        const cplx pw = OP[(it * 4 + ks) * 64 + lane];
                        xre[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(pw.x, tre[ks], xre[it], 0, 0, 0);
                        xim[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(pw.x, tim[ks], xim[it], 0, 0, 0);
                        xre[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(pw.y, tim[ks], xre[it], 0, 0, 0);
                        xim[it] = __builtin_amdgcn_mfma_f64_16x16x4f64(pw.y, ntre[ks], xim[it], 0, 0, 0);
                    }
                    nxim[it] = -xim[it];
#pragma unroll
                    for (int reg = 0; reg < 4; ++reg) {
                        const int rl = 16 * it + lk + 4 * reg, col = c0 + 16 * b + li;
                        if (rl < nb && col < D)
                            Z[(int64_t)(j0 + rl) * D + col] = c_make(xre[it][reg], xim[it][reg]);
                    }
                }
            }
            __syncthreads();
        }
    }
}

// (3 waves per SIMD asked for explicitly: 513 = 2 * 256 + 1 workgroups need 3 resident per
// CU to run in one round, and the allocator otherwise stops one register above the limit)
__global__ __launch_bounds__(256, 3) void chol_backsolve_kernel(const cplx *__restrict__ R,
                                                             cplx *__restrict__ P, int n, int D) {
    __shared__ cplx S[CH_NB * BS_LD];
    __shared__ cplx OP[BS_OPS];
    const int f = blockIdx.x;
    chol_backsolve_body(R + (int64_t)f * n * n, P + (int64_t)f * n * D, n, D, S, OP);
}

// ------------------------------------------------------------------ apply
// X[t][d] = Y[t][d] - sum_r conj(G[r][d]) Yflat[(t - c) D + r]
// As a GEMM per frequency: out(frames x channels) = U conj(G), K = n = taps * D, on the
// f64 MFMA.  A[i = frame][k = r] comes from the LDS copy of the frames the workgroup
// touches -- stored with an odd frame stride DP so that the 16 frames of a fragment
// hit 16 different bank groups; (frame fl, r) lives at (fl + r / D) * DP + r % D --
// B[k = r][j = d] straight from G in global memory (L2 / L1 resident, shared by all
// waves of a frequency), double-buffered in registers.
// grid (ceil(T / (64 TA)), F), block 256: each wave owns 16 TA frames x 16 NB channel
// slots.  NB (channel tiles) is a template parameter: a run-time tile count inside the
// k loop makes the compiler shuttle the accumulators between VGPRs and AGPRs around
// every MFMA and wait for each result.
// M3: three real MFMAs per complex product as in the correlation (t1 = sum ur gr,
// t2 = sum ui gi, t3 = sum (ur + ui)(gr - gi); re = t1 + t2, im = t3 - t1 + t2).
// WRAPS: carries of the running (r / D, r % D) per k-step -- one is enough from D = 4 on (r
// advances by 4), and every wrap is three VALU instructions that the MFMA pipe waits for.
// TAIL: rows r >= n have to be masked -- n is not a multiple of 8 (the loop runs two k-steps per
// trip, so an odd k-step count executes one k-step past the end).
template <int TA, int NB, int NWV = 4, int WRAPS = 4, bool TAIL = true>
__global__ __launch_bounds__(64 * NWV) void wpe_apply_kernel(const cplx *__restrict__ Y,
                                                        const cplx *__restrict__ G, int F,
                                                        int64_t T, int D, int n, int c,
                                                        cplx *__restrict__ X) {
    constexpr int WAVE_FRAMES = 16 * TA, WG_FRAMES = NWV * WAVE_FRAMES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx *S = reinterpret_cast<cplx *>(smem);   // (WG_FRAMES + c + 2) * DP
    const int DP = D | 1;
    int f, chunk;
    if (!xcd_group_map((int)((T + WG_FRAMES - 1) / WG_FRAMES), F, f, chunk)) return;
    const int64_t t0 = (int64_t)chunk * WG_FRAMES;
    const cplx *Yf = Y + (int64_t)f * T * D;
    const cplx *Gf = G + (int64_t)f * n * D;
    const int frames_lds = WG_FRAMES + c + 2;
    const int64_t fr0 = t0 - c;
    // window -> LDS, 8 loads in flight per thread (one load per trip is a chain of
    // dependent round trips as long as the whole k loop)
    constexpr int PRE = 14, NT = 64 * NWV;
    const int total = frames_lds * D;
    for (int base = 0; base < total; base += NT * PRE) {
        cplx v[PRE];
#pragma unroll
        for (int j = 0; j < PRE; ++j) {
            const int idx = base + (int)threadIdx.x + NT * j;
            const int64_t fr = fr0 + idx / D;
            v[j] = c_make(0.0, 0.0);
            if (idx < total && fr >= 0 && fr < T) v[j] = Yf[fr0 * D + idx];
        }
#pragma unroll
        for (int j = 0; j < PRE; ++j) {
            const int idx = base + (int)threadIdx.x + NT * j;
            const int fl = idx / D, d = idx - fl * D;
            if (idx < total) S[fl * DP + d] = v[j];
        }
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    const int wf0 = wave * WAVE_FRAMES;          // first frame of this wave, tile relative
    // (no barrier below: a wave whose frames all lie past the last one -- two of the 32 wave
    // tiles at T = 941 -- leaves its SIMD to the other workgroup's wave)
    if (t0 + wf0 >= T) return;
    v4d acc_re[TA][NB], acc_im[TA][NB], acc_t2[TA][NB];
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            acc_re[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
            acc_im[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
            acc_t2[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
        }
    // Two k-steps per trip with ping-pong operand registers: G rows (global, L2 resident)
    // and window fragments (LDS) of k-step ks + 1 are requested before the MFMAs of ks are
    // issued and first touched a k-step later, so neither latency sits between two groups
    // of MFMAs.  Loads are unconditional (clamped addresses); rows r >= n are cancelled by
    // zeroing the window operand, channel slots d >= D are computed and never stored.  The
    // k loop has no branch: with control flow inside it the compiler keeps the accumulators
    // in VGPRs and copies them to AGPRs and back around every group of MFMAs.
    const int ksteps = (n + 3) / 4;
    const int lds_last = frames_lds * DP - 1;
    // Operand addresses are ADVANCED, not recomputed: every integer VALU instruction of the k
    // loop is time the f64 MFMA pipe of this SIMD does not get, and the 64-bit
    // multiply-adds of "row * D + column" were the most expensive of them.  G: one 32-bit
    // element offset per channel tile, + 4 D per k-step, clamped to the last row (k-steps
    // past the end are requested and never used).  Window: lin = (r / D) DP + r % D of the
    // running row r = 4 ks + lk, + 4 per k-step, + DP - D when r % D wraps.
    int goff[NB], goff_max[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int col = min(16 * b + li, D - 1);
        goff[b] = min(lk, n - 1) * D + col;
        goff_max[b] = (n - 1) * D + col;
    }
    auto load_g = [&](int, cplx (&g)[NB]) {      // (k-steps are requested in order)
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            g[b] = Gf[goff[b]];
            goff[b] = min(goff[b] + 4 * D, goff_max[b]);
        }
    };
    int rm = lk % D, lin = (lk / D) * DP + rm;
    int ubase[TA];
#pragma unroll
    for (int a = 0; a < TA; ++a) ubase[a] = (wf0 + 16 * a + li) * DP;
    auto load_u = [&](cplx (&u)[TA]) {
#pragma unroll
        for (int a = 0; a < TA; ++a) u[a] = S[min(ubase[a] + lin, lds_last)];
        rm += 4;
        lin += 4;
#pragma unroll
        for (int w = 0; w < WRAPS; ++w) {   // D >= 1: at most 4 wraps, branch free
            const bool wrap = rm >= D;
            rm -= wrap ? D : 0;
            lin += wrap ? DP - D : 0;
        }
    };
    auto step = [&](int ks, const cplx (&g)[NB], const cplx (&u)[TA]) {
        const bool ok = !TAIL || 4 * ks + lk < n;
        double ur[TA], ui[TA], gr[NB], gi[NB];
#pragma unroll
        for (int a = 0; a < TA; ++a) {
            ur[a] = ok ? u[a].x : 0.0;
            ui[a] = ok ? u[a].y : 0.0;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            gr[b] = g[b].x;
            gi[b] = g[b].y;
        }
        // u * conj(g)
        // acc_re = t1, acc_t2 = t2, acc_im = t3
        double us[TA], gd[NB];
#pragma unroll
        for (int a = 0; a < TA; ++a) us[a] = ur[a] + ui[a];
#pragma unroll
        for (int b = 0; b < NB; ++b) gd[b] = gr[b] - gi[b];
#pragma unroll
        for (int a = 0; a < TA; ++a)
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                acc_re[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ur[a], gr[b], acc_re[a][b], 0, 0, 0);
                acc_t2[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ui[a], gi[b], acc_t2[a][b], 0, 0, 0);
                acc_im[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(us[a], gd[b], acc_im[a][b], 0, 0, 0);
            }
    };
    cplx g0[NB], g1[NB], u0[TA], u1[TA];
    load_g(0, g0);
    load_u(u0);
    for (int ks = 0; ks < ksteps; ks += 2) {
        // (scheduling barriers: left alone, the compiler gathers both loads at the top of
        // the trip and waits for them before the first MFMA)
        load_g(ks + 1, g1);
        load_u(u1);
        __builtin_amdgcn_sched_barrier(0);
        step(ks, g0, u0);
        __builtin_amdgcn_sched_barrier(0);
        load_g(ks + 2, g0);
        load_u(u0);
        __builtin_amdgcn_sched_barrier(0);
        step(ks + 1, g1, u1);
        __builtin_amdgcn_sched_barrier(0);
    }
    // C/D fragment: col = li (channel), row = lk + 4 * reg (frame)
#pragma unroll
    for (int a = 0; a < TA; ++a)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int fl = wf0 + 16 * a + lk + 4 * reg;
                const int d = 16 * b + li;
                const int64_t t = t0 + fl;
                if (d < D && t < T) {
                    const cplx y = S[(fl + c) * DP + d];
                    const double pre = acc_re[a][b][reg] + acc_t2[a][b][reg];
                    const double pim = (acc_im[a][b][reg] - acc_re[a][b][reg]) + acc_t2[a][b][reg];
                    X[((int64_t)f * T + t) * D + d] = c_make(y.x - pre, y.y - pim);
                }
            }
}

// ------------------------------------------------------------------ apply, frame phases packed into N
// The GEMM above has N = D channels: at D = 24 a quarter of its MFMAs work on the empty half of
// the second column tile, at D = 4 three quarters on an almost empty first one.  Consecutive
// frames read the SAME flat window shifted by D elements:
//     pred[t + p][d] = sum_r conj(G[r][d]) Yflat[(t - c) D + r + p D]
//                    = sum_k conj(G[k - p D][d]) Yflat[(t - c) D + k],     k = r + p D,
// so PH consecutive frames share one A row (the window that starts at frame t, K' = n +
// (PH - 1) D long) against a B operand of PH D columns, column p D + d holding G shifted down by
// p D rows: N = 48 = 3 full tiles at D = 24 with PH = 2 (17.5 % fewer MFMAs), N = 16 at D = 4
// with PH = 4 (a third of the MFMAs).  G is staged in LDS once per workgroup (the shifted
// columns would otherwise be three operand loads per k-step from L1 -- the form that was
// measured slower in round 1), the window as before; frame f of the window lives at
// f DP + f / PH so that the rows of a fragment (frames PH i) are an odd number of elements
// apart.  grid (ceil(T / (64 PH)), F), block 256: one 16-row tile (16 PH frames) per wave.
// GLDS = false (20 / 24 channels, round 6): G stays in global memory (92 KB per frequency, L2
// resident and shared by the 8 workgroups of a frequency) -- with G in LDS the packed form owns
// a CU alone at these channel counts (1.82 vs 1.35 ms in round 4); the operands are requested
// two k-steps (18 MFMAs) ahead.  Opt-in (GSS_VARIANT apply_gglobal): measured slower than the
// unpacked kernel, see wpe_run.
// PREW / PREG: elements per thread of the staging passes (window, G).  14 covers every shape;
// one array (820 window elements, 160 of G) needs 4 and 1 -- the kernel is 36 MFMAs per wave
// behind two staging passes, and at 14 predicated slots each those passes were 1170 of its 1857
// instructions (tools/isa_scratch_by_loop.py).
template <int PH, int NT, bool GLDS = true, int PREW = 14, int PREG = 14>
__global__ __launch_bounds__(256) void wpe_apply_packed_kernel(const cplx *__restrict__ Y,
                                                               const cplx *__restrict__ G, int F,
                                                               int64_t T, int D, int n, int c,
                                                               cplx *__restrict__ X) {
    constexpr int WG_ROWS = 64, WG_FRAMES = PH * WG_ROWS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int DP = D | 1, DG = GLDS ? (D | 1) : D;
    const int frames_lds = WG_FRAMES + c + 2;
    cplx *S = reinterpret_cast<cplx *>(smem);                   // frames_lds * DP + frames_lds / PH + 1
    cplx *Gl = S + (frames_lds * DP + frames_lds / PH + 1);     // (n + 1) * DG, last row zeros
    int f, chunk;
    if (!xcd_group_map((int)((T + WG_FRAMES - 1) / WG_FRAMES), F, f, chunk)) return;
    const int64_t t0 = (int64_t)chunk * WG_FRAMES;
    const cplx *Yf = Y + (int64_t)f * T * D;
    const cplx *Gf = G + (int64_t)f * n * D;
    const int64_t fr0 = t0 - c;
    constexpr int NTHR = 256;
    {   // window -> LDS (all loads of a batch in flight before the first store)
        constexpr int PRE = PREW;
        const int total = frames_lds * D;
        for (int base = 0; base < total; base += NTHR * PRE) {
            cplx v[PRE];
#pragma unroll
            for (int j = 0; j < PRE; ++j) {
                const int idx = base + (int)threadIdx.x + NTHR * j;
                const int64_t fr = fr0 + idx / D;
                v[j] = c_make(0.0, 0.0);
                if (idx < total && fr >= 0 && fr < T) v[j] = Yf[fr0 * D + idx];
            }
#pragma unroll
            for (int j = 0; j < PRE; ++j) {
                const int idx = base + (int)threadIdx.x + NTHR * j;
                const int fl = idx / D, d = idx - fl * D;
                if (idx < total) S[fl * DP + fl / PH + d] = v[j];
            }
        }
    }
    if (GLDS) {   // G -> LDS
        constexpr int PRE = PREG;
        cplx *Gs = Gl;
        const int total = n * D;
        for (int base = 0; base < total; base += NTHR * PRE) {
            cplx v[PRE];
#pragma unroll
            for (int j = 0; j < PRE; ++j) {
                const int idx = base + (int)threadIdx.x + NTHR * j;
                v[j] = idx < total ? Gf[idx] : c_make(0.0, 0.0);
            }
#pragma unroll
            for (int j = 0; j < PRE; ++j) {
                const int idx = base + (int)threadIdx.x + NTHR * j;
                const int r = idx / D, d = idx - r * D;
                if (idx < total) Gs[r * DG + d] = v[j];
            }
        }
        for (int d = threadIdx.x; d < DG; d += NTHR) Gs[n * DG + d] = c_make(0.0, 0.0);
    }
    __syncthreads();
    // where the B operand comes from: the LDS copy (row n = zeros) or G itself (columns past
    // PH D then read row 0: their products are computed and never stored)
    const cplx *Gs = GLDS ? Gl : Gf;
    const int gzero = GLDS ? n * DG : 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 15, lk = lane >> 4;
    if (t0 + (int64_t)16 * wave * PH >= T) return;          // (no barrier below) all frames past the end
    const int row = 16 * wave + li;                         // A row of this lane: frames PH row + p
    const int abase = row * (PH * DP + 1);                  // LDS address of frame PH row
    const int Kp = n + (PH - 1) * D;
    const int ksteps = (Kp + 3) / 4;
    // column of this lane in tile b: col = 16 b + li = p D + d
    int colp[NT], cold[NT];
    bool colv[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int col = 16 * b + li;
        colv[b] = col < PH * D;
        colp[b] = colv[b] ? col / D : 0;
        cold[b] = colv[b] ? col - colp[b] * D : 0;
    }
    v4d acc_re[NT], acc_t2[NT], acc_im[NT];
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        acc_re[b] = (v4d){0.0, 0.0, 0.0, 0.0};
        acc_t2[b] = (v4d){0.0, 0.0, 0.0, 0.0};
        acc_im[b] = (v4d){0.0, 0.0, 0.0, 0.0};
    }
    // The k loop in three sections.  Head: k-steps that hold a k < (PH - 1) D (a shifted column
    // has not started yet); tail: k-steps that hold a k >= n (the unshifted column has ended, the
    // window operand may run past K'); body: every operand of every lane is valid and the
    // addresses advance by constants -- no compare, no select, no division in between the MFMAs
    // (with 153 KB of LDS one workgroup owns a CU: one wave per SIMD, and whatever VALU work sits
    // between two groups of MFMAs is not hidden by anybody).  Columns past PH D (a partly
    // filled last tile) read a row of zeros behind G with a zero address increment.
    const int ks_head = min(((PH - 1) * D + 3) / 4, ksteps);
    const int ks_tail = max(min(n / 4, ksteps), ks_head);
    // k = 4 ks + lk = rq D + rm
    int k = lk, rq = lk / D, rm = lk - rq * D;
    auto u_addr = [&]() { return abase + rq * DP + rq / PH + rm; };
    auto advance = [&]() {
        k += 4;
        rm += 4;
#pragma unroll
        for (int w = 0; w < 4; ++w) {   // D >= 1: at most 4 wraps, branch free
            const bool wrap = rm >= D;
            rm -= wrap ? D : 0;
            rq += wrap ? 1 : 0;
        }
    };
    auto mfma = [&](double ur, double ui, const double (&gr)[NT], const double (&gi)[NT]) {
        const double us = ur + ui;
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            acc_re[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ur, gr[b], acc_re[b], 0, 0, 0);
            acc_t2[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(ui, gi[b], acc_t2[b], 0, 0, 0);
            acc_im[b] = __builtin_amdgcn_mfma_f64_16x16x4f64(us, gr[b] - gi[b], acc_im[b], 0, 0, 0);
        }
    };
    // checked k-step (head and tail): validity by compare and select
    auto checked_step = [&]() {
        const bool ok = k < Kp;
        const cplx u = S[ok ? u_addr() : abase];
        double gr[NT], gi[NT];
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            const int kr = k - colp[b] * D;
            const bool gv = colv[b] && kr >= 0 && kr < n;
            const cplx g = Gs[(gv ? kr : 0) * DG + cold[b]];
            gr[b] = gv ? g.x : 0.0;
            gi[b] = gv ? g.y : 0.0;
        }
        mfma(ok ? u.x : 0.0, ok ? u.y : 0.0, gr, gi);
        advance();
    };
    for (int ks = 0; ks < ks_head; ++ks) checked_step();
    if (!GLDS && ks_tail > ks_head) {
        // body with G from global memory (L2): operands TWO k-steps ahead in a ring of three
        // register sets (one k-step = 9 MFMAs does not cover an L2 round trip under load:
        // one-ahead measured 1.41 ms per utterance against 1.26 for the unpacked kernel).  The
        // first (body length mod 3) k-steps go through the checked form.
        int ks = ks_head;
        for (int r = (ks_tail - ks_head) % 3; r > 0; --r, ++ks) checked_step();
        int ga[NT], ginc[NT];
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            ga[b] = colv[b] ? (k - colp[b] * D) * DG + cold[b] : gzero;
            ginc[b] = colv[b] ? 4 * DG : 0;
        }
        auto load = [&](cplx &u, cplx (&g)[NT]) {      // operands of the load cursor's k-step
            u = S[u_addr()];
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                g[b] = Gs[ga[b]];
                ga[b] += ginc[b];
            }
            advance();
        };
        auto run = [&](const cplx &u, const cplx (&g)[NT]) {
            double gr[NT], gi[NT];
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                gr[b] = g[b].x;
                gi[b] = g[b].y;
            }
            mfma(u.x, u.y, gr, gi);
        };
        cplx u0, u1, u2 = c_make(0.0, 0.0), g0[NT], g1[NT], g2[NT];
#pragma unroll
        for (int b = 0; b < NT; ++b) g2[b] = c_make(0.0, 0.0);
        if (ks < ks_tail) {
            load(u0, g0);
            load(u1, g1);
        }
        for (; ks < ks_tail; ks += 3) {
            if (ks + 2 < ks_tail) load(u2, g2);
            __builtin_amdgcn_sched_barrier(0);
            run(u0, g0);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 3 < ks_tail) load(u0, g0);
            __builtin_amdgcn_sched_barrier(0);
            run(u1, g1);
            __builtin_amdgcn_sched_barrier(0);
            if (ks + 4 < ks_tail) load(u1, g1);
            __builtin_amdgcn_sched_barrier(0);
            run(u2, g2);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else
    if (ks_tail > ks_head) {
        // body: operands one k-step ahead, addresses by increments
        int ga[NT], ginc[NT];
#pragma unroll
        for (int b = 0; b < NT; ++b) {
            ga[b] = colv[b] ? (k - colp[b] * D) * DG + cold[b] : gzero;      // row n of Gs: zeros
            ginc[b] = colv[b] ? 4 * DG : 0;
        }
        cplx u_cur = S[u_addr()], g_cur[NT];
#pragma unroll
        for (int b = 0; b < NT; ++b) g_cur[b] = Gs[ga[b]];
        for (int ks = ks_head; ks < ks_tail; ++ks) {
            advance();
            cplx u_nxt = u_cur, g_nxt[NT];
            const bool more = ks + 1 < ks_tail;                 // uniform
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                ga[b] += ginc[b];
                g_nxt[b] = g_cur[b];
            }
            if (more) {
                u_nxt = S[u_addr()];
#pragma unroll
                for (int b = 0; b < NT; ++b) g_nxt[b] = Gs[ga[b]];
            }
            __builtin_amdgcn_sched_barrier(0);
            double gr[NT], gi[NT];
#pragma unroll
            for (int b = 0; b < NT; ++b) {
                gr[b] = g_cur[b].x;
                gi[b] = g_cur[b].y;
            }
            mfma(u_cur.x, u_cur.y, gr, gi);
            __builtin_amdgcn_sched_barrier(0);
            u_cur = u_nxt;
#pragma unroll
            for (int b = 0; b < NT; ++b) g_cur[b] = g_nxt[b];
        }
    }
    for (int ks = ks_tail; ks < ksteps; ++ks) checked_step();
    // C/D fragment: col = li, row = lk + 4 reg  ->  frame PH (16 wave + row) + p, channel d
#pragma unroll
    for (int b = 0; b < NT; ++b)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
            const int r = 16 * wave + lk + 4 * reg;
            const int fl = PH * r + colp[b];
            const int64_t t = t0 + fl;
            if (colv[b] && t < T) {
                const int fw = fl + c;
                const cplx y = S[fw * DP + fw / PH + cold[b]];
                const double pre = acc_re[b][reg] + acc_t2[b][reg];
                const double pim = (acc_im[b][reg] - acc_re[b][reg]) + acc_t2[b][reg];
                X[((int64_t)f * T + t) * D + cold[b]] = c_make(y.x - pre, y.y - pim);
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

static int corr_tiles(int n, int D, int c, int ct, std::vector<CorrTile> &tiles) {
    // R: every ct x ct tile that reaches the upper triangle; P: all rows x D columns.
    // mask = the 16 x 16 sub-tiles that hold at least one needed entry.
    const int ts = ct / 16;
    auto mask_of = [&](int r0, int c0, int ncols_limit, bool upper_only) {
        int m = 0;
        for (int a = 0; a < ts; ++a)
            for (int b = 0; b < ts; ++b) {
                const int rr = r0 + 16 * a, cc = c0 + 16 * b;
                bool needed = rr < n && cc < ncols_limit;
                if (upper_only && cc + 15 < rr) needed = false;     // entirely below the diagonal
                if (needed) m |= 1 << (a * ts + b);
            }
        return m;
    };
    // P = Gram columns [c D, c D + D).  When they all fall into the LAST column tile of R
    // (one array at 10 taps, delay 2: n = 40 is padded to 48 and P sits at 44..47), every row
    // block's tile of that column already computes them -- the tile is marked 2 ("R and P") and
    // no P tile is launched: 6 waves per frequency instead of 9.
    const int last_col = (n - 1) / ct;
    // (16 x 16 tiles only: a diagonal 32 x 32 tile skips its sub-tile below the diagonal)
    const bool p_folded = ct == 16 && (c * D) / ct == last_col && (c * D + D - 1) / ct == last_col &&
                          !gss_variant_set("corr_p_tiles");
    for (int r0 = 0; r0 < n; r0 += ct)
        for (int c0 = 0; c0 < n; c0 += ct)
            if (c0 + ct > r0) {
                const bool with_p = p_folded && c0 / ct == last_col;
                // (the sub-tiles that hold P columns are needed even where R ends before them)
                tiles.push_back({r0, c0, with_p ? 2 : 0,
                                 mask_of(r0, c0, with_p ? std::max(n, c * D + D) : n, true)});
            }
    if (!p_folded)
        for (int r0 = 0; r0 < n; r0 += ct)
            for (int c0 = 0; c0 < D; c0 += ct)
                tiles.push_back({r0, c * D + c0, 1, mask_of(r0, c0, D, false)});
    // heaviest first, so that the four waves of a workgroup carry similar loads
    std::stable_sort(tiles.begin(), tiles.end(), [](const CorrTile &x, const CorrTile &y) {
        return __builtin_popcount(x.mask) > __builtin_popcount(y.mask);
    });
    return (int)tiles.size();
}

// 16 x 32 (R) and 32 x 16 (P) wave tiles for the channel counts between one array and the
// 32 x 32 tiling (D = 8 ... 12 at 10 taps): pairs of needed 16 x 16 sub-tiles -- two column
// sub-tiles of one row for R (mask 3), two row sub-tiles of one column for P (mask 5) -- run
// through the 32 x 32 kernels, whose masks skip the other half of the tile with its operand
// loads.  Every wave carries two sub-tiles (the odd one of a row: one): 6 MFMAs per 5 operand
// products and k-step, where one sub-tile per wave has 3 per 4 and the 32 x 32 tiling leaves
// the four waves of a workgroup with 4, 3, 2 and 1 sub-tiles at n = 120.
static int corr_tiles_pairs(int n, int D, int c, std::vector<CorrTile> &tiles) {
    for (int r0 = 0; r0 < n; r0 += 16)
        for (int c0 = r0; c0 < n; c0 += 32) tiles.push_back({r0, c0, 0, c0 + 16 < n ? 3 : 1});
    for (int r0 = 0; r0 < n; r0 += 32)
        for (int c0 = 0; c0 < D; c0 += 16) tiles.push_back({r0, c * D + c0, 1, r0 + 16 < n ? 5 : 1});
    std::stable_sort(tiles.begin(), tiles.end(), [](const CorrTile &x, const CorrTile &y) {
        return __builtin_popcount(x.mask) > __builtin_popcount(y.mask);
    });
    return (int)tiles.size();
}

static int corr_padf(int D, int ct) { return (ct + D - 1) / D + 1; }

size_t wpe_workspace_bytes(int F, int64_t T, int D, int taps, int delay) {
    const size_t n = (size_t)taps * D;
    size_t b = 0;
    b += 2 * align_up(sizeof(double) * (size_t)F * T);   // w, raw power (psd_context > 0)
    b += align_up(sizeof(cplx) * (size_t)F * n * n);     // R
    b += align_up(sizeof(cplx) * (size_t)F * n * D);     // P / G
    return b + 4096;
}

static int wpe_power_launch(gss_ctx *ctx, const cplx *cur, int F, int64_t T, int D,
                            int psd_context, double *raw, double *w) {
    const size_t pow_lds = sizeof(double) * POW_FRAMES * (D + 1);
    if (pow_lds > 64 * 1024)
        GSS_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(wpe_power_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)pow_lds));
    hipLaunchKernelGGL(wpe_power_kernel, dim3(F), dim3(POW_FRAMES), pow_lds, ctx->stream,
                       cur, T, D, psd_context, raw, w);
    GSS_LAUNCH_CHECK(ctx, "wpe_power_kernel");
    return GSS_OK;
}

// gss_wpe_inverse_power: the weights of one WPE iteration on their own
int wpe_inverse_power_run(gss_ctx *ctx, const cplx *Y, int F, int64_t T, int D, int psd_context,
                          double *w) {
    double *raw = psd_context > 0 ? arena_alloc_t<double>(ctx, (size_t)F * T) : nullptr;
    GSS_REQUIRE(ctx, raw || psd_context == 0, GSS_ERR_NOMEM, "wpe power workspace");
    return wpe_power_launch(ctx, Y, F, T, D, psd_context, raw, w);
}

int wpe_run(gss_ctx *ctx, const cplx *Y, int F, int64_t T, int D, int taps, int delay,
            int iterations, int psd_context, cplx *X, int part) {
    const int n = taps * D;
    const int c = delay + taps - 1;
    if (iterations == 0) {
        // nothing is solved: gss_last_wpe_zero_pivots() must not report an earlier call's count
        GSS_HIP_CHECK(ctx, hipMemsetAsync(ctx->status_dev + 2, 0, sizeof(int32_t), ctx->stream));
        if (X != Y)
            GSS_HIP_CHECK(ctx, hipMemcpyAsync(X, Y, sizeof(cplx) * (size_t)F * T * D,
                                              hipMemcpyDeviceToDevice, ctx->stream));
        return GSS_OK;
    }
    GSS_REQUIRE(ctx, X != Y, GSS_ERR_INVALID, "gss_wpe: X must not alias Y");
    double *w = arena_alloc_t<double>(ctx, (size_t)F * T);
    // (NULL without smoothing: the kernel's two __restrict__ outputs must not alias)
    double *raw = psd_context > 0 ? arena_alloc_t<double>(ctx, (size_t)F * T) : nullptr;
    cplx *R = arena_alloc_t<cplx>(ctx, (size_t)F * n * n);
    cplx *P = arena_alloc_t<cplx>(ctx, (size_t)F * n * D);
    GSS_REQUIRE(ctx, w && (raw || psd_context == 0) && R && P, GSS_ERR_NOMEM, "wpe workspace");

    // tile lists: correlation tiles, then one trailing-update list per block column
    std::vector<CorrTile> tiles;
    // 32 x 32 wave tiles (48 x 48 measured slower).  With few channels (one array: taps * D
    // = 40 is 3 sub-tiles of 16 wide, 9 needed in all) the 32 x 32 list is a handful of
    // unequal tiles per frequency and the launch is bound by its heaviest waves, so there
    // every needed 16 x 16 sub-tile becomes a wave of its own: equal loads, more waves,
    // fewer registers each.  Measured (T = 2169, ms per launch, 32 x 32 -> 16 x 16):
    // D = 4: 0.53 -> 0.33, D = 10: 1.33 -> 1.12, D = 12: 1.50 -> 1.36, D = 20: 3.40 -> 3.92,
    // D = 24: 4.16 -> 5.11.
    const int sub16 = (n + 15) / 16;
    int corr_ts = sub16 * (sub16 + 1) / 2 + sub16 * ((D + 15) / 16) <= CORR_FINE_MAX_SUBTILES ? 1 : 2;
    // GSS_VARIANT corr_ts=1|2: force the fine / 32 x 32 tiling; corr_ts=3: the pair tiling
    // (corr_tiles_pairs) on the 32 x 32 kernels
    // GSS_VARIANT corr_ts=3, the pair tiling on the 32 x 32 kernels (every wave two sub-tiles
    // instead of one), measures -2.9 % at D = 10 and -0.8 ... -2 % at D = 12 -- and, on the
    // persistent kernel's queue order, fetches the slab four times (config 5: 2.9 GB per launch
    // against 0.99 GB for the fine tiling, profiles/r05b_cfg5_traffic.json): not the default
    bool corr_pairs = false;
    if (const int e = gss_variant("corr_ts", 0)) {
        corr_ts = e == 1 ? 1 : 2;
        corr_pairs = e == 3;
    }
    const int ntiles = corr_pairs ? corr_tiles_pairs(n, D, c, tiles) : corr_tiles(n, D, c, 16 * corr_ts, tiles);
    // trailing-update tiles: 16 x 16, every tile that reaches the upper triangle (larger
    // register tiles -- 2 x 2, 1 x 2, 2 x 1 MFMA tiles per wave -- measured slower: the update
    // is bound by the read-modify-write of the trailing matrix and wants many small
    // workgroups in flight).  To halve that traffic the block columns are taken in pairs:
    // after an even block J only block row J + 1 is updated (K = 48, panel J), after an odd
    // block J everything below it is updated once with both panels J - 1 and J (K = 96).
    constexpr int tm16 = 16, tn16 = 16;
    std::vector<UpdTile> upd;
    std::vector<int> upd_start, upd_count, upd_j0, upd_k, upd_ndiag;
    const bool fold_diag = !gss_variant_set("chol_diag_unfolded");
    {
        const int nblk = (n + CH_NB - 1) / CH_NB;
        for (int J = 0; J < nblk; ++J) {
            upd_start.push_back((int)upd.size());
            const int rs = (J + 1) * CH_NB;
            const bool narrow = J % 2 == 0;
            const int r_end = narrow ? std::min(rs + CH_NB, n) : n;
            // the tiles of the next diagonal block first (group 0 factors it in the same
            // launch, see chol_update_kernel), then everything else
            const int d_end = std::min(rs + CH_NB, n);
            for (int r0 = rs; r0 < d_end; r0 += tm16)
                for (int c0 = rs; c0 < d_end; c0 += tn16)
                    if (c0 + tn16 > r0) upd.push_back({r0, c0, 0, 0});
            upd_ndiag.push_back(fold_diag ? (int)upd.size() - upd_start.back() : 0);
            for (int r0 = rs; r0 < r_end; r0 += tm16) {
                for (int c0 = rs; c0 < n; c0 += tn16)
                    if (c0 + tn16 > r0 && !(r0 < d_end && c0 < d_end)) upd.push_back({r0, c0, 0, 0});
                for (int cc = 0; cc < D; cc += tn16) upd.push_back({r0, cc, 1, 0});
            }
            upd_count.push_back((int)upd.size() - upd_start.back());
            const bool wide = J % 2 == 1;
            upd_j0.push_back(wide ? (J - 1) * CH_NB : J * CH_NB);
            upd_k.push_back(wide ? 2 * CH_NB : CH_NB);
        }
    }
    GSS_REQUIRE(ctx, ntiles <= 1024 && upd.size() <= 4096, GSS_ERR_UNSUPPORTED,
                "wpe: taps*D=%d too large", n);
    static_assert(sizeof(CorrTile) == sizeof(UpdTile), "tile structs share one buffer");
    if (ctx->wpe_tiles_key[0] != taps || ctx->wpe_tiles_key[1] != delay ||
        ctx->wpe_tiles_key[2] != D || ctx->wpe_tiles_key[3] != corr_ts + (fold_diag ? 0 : 16) + (corr_pairs ? 32 : 0)) {
        GSS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
        if (!ctx->wpe_tiles)
            GSS_HIP_CHECK(ctx, hipMalloc(&ctx->wpe_tiles, sizeof(CorrTile) * (1024 + 4096 + 1 + 64)));
        GSS_HIP_CHECK(ctx, hipMemcpy(ctx->wpe_tiles, tiles.data(), sizeof(CorrTile) * ntiles,
                                     hipMemcpyHostToDevice));
        if (!upd.empty())
            GSS_HIP_CHECK(ctx, hipMemcpy(reinterpret_cast<CorrTile *>(ctx->wpe_tiles) + 1024,
                                         upd.data(), sizeof(UpdTile) * upd.size(),
                                         hipMemcpyHostToDevice));
        ctx->wpe_tiles_key[0] = taps;
        ctx->wpe_tiles_key[1] = delay;
        ctx->wpe_tiles_key[2] = D;
        ctx->wpe_tiles_key[3] = corr_ts + (fold_diag ? 0 : 16) + (corr_pairs ? 32 : 0);
    }
    CorrTile *tiles_dev = reinterpret_cast<CorrTile *>(ctx->wpe_tiles);
    UpdTile *upd_dev = reinterpret_cast<UpdTile *>(tiles_dev + 1024);
    // zeroed pivots of this call (all iterations, all frequencies): counted on the device,
    // copied to the context's status words at the end (gss_last_wpe_zero_pivots)
    int32_t *zero_pivots = reinterpret_cast<int32_t *>(tiles_dev + 1024 + 4096);
    if (part <= 0) GSS_HIP_CHECK(ctx, hipMemsetAsync(zero_pivots, 0, sizeof(int32_t), ctx->stream));
    if (part == 0) GSS_HIP_CHECK(ctx, hipEventRecord(ctx->ev_fork, ctx->stream));
    // work-queue heads of the persistent correlation kernel: 8 counters, 64 bytes apart (a second
    // set for the second of two parts that run side by side)
    int *corr_counters = reinterpret_cast<int *>(tiles_dev + 1024 + 4096 + 1) + (part == 1 ? 128 : 0);

    const int padf = corr_padf(D, 16 * corr_ts);
    // (everywhere: 3 real MFMAs per complex product -- t1 = ar br, t2 = ai bi,
    // t3 = (ar + ai)(br - bi); the 4-product forms were removed in round 5)
    // Waves per workgroup.  A workgroup lives for the whole frame loop, so the launch runs in
    // ceil(workgroups / resident slots) rounds: with F = 513 = 2 * 256 + 1 frequencies the
    // 4-wave form can leave a last round almost empty (D = 12, taps 10: 2052 workgroups on
    // 1024 slots = 3 rounds for 2.004 rounds of work).  Two waves per workgroup stage the
    // window twice as often per tile but pack 8 workgroups per CU.
    const size_t corr_lds_probe = sizeof(cplx) * (size_t)(CORR_KT + c + corr_padf(D, 16 * corr_ts)) * D +
                                  sizeof(double) * CORR_KT;
    auto round_eff = [&](int nw) {
        const double wgs = (double)((ntiles + nw - 1) / nw) * F;
        const int by_regs = 16 / nw, by_lds = (int)(160 * 1024 / corr_lds_probe);
        const double rounds = wgs / (256.0 * std::max(1, std::min(by_regs, by_lds)));
        return rounds / std::ceil(rounds);
    };
    int corr_nw = corr_ts == 2 && round_eff(2) > round_eff(4) + 0.15 ? 2 : 4;
    const int nw_forced = gss_variant("corr_nw", 0);
    if (nw_forced) corr_nw = nw_forced == 2 ? 2 : nw_forced == 1 && corr_ts == 1 ? 1 : 4;
    // one array: the window is a few KB, every wave stages its own and the hardware
    // balances single waves (no idle wave in a workgroup, no barrier partner to wait for)
    if (corr_ts == 1 && (size_t)(CORR_KT + c + padf) * D <= 512 && !nw_forced) corr_nw = 1;
    // (single waves whose window is at most 320 elements -- one array at 10 taps -- stage 5
    // elements per lane and fit four to a SIMD; GSS_VARIANT corr_stg8: the 8-element form)
    const bool corr_stg5 = (size_t)(CORR_KT + c + padf) * D <= 320 && !gss_variant_set("corr_stg8");
    auto corr_fn = corr_ts == 1 && corr_nw == 1 ? (corr_stg5 ? wpe_corr_kernel<1, 1, 5> : wpe_corr_kernel<1, 1>)
                   : corr_ts == 1 ? (corr_nw == 2 ? wpe_corr_kernel<1, 2> : wpe_corr_kernel<1, 4>)
                                  : (corr_nw == 2 ? wpe_corr_kernel<2, 2> : wpe_corr_kernel<2, 4>);
    size_t corr_lds = sizeof(cplx) * (size_t)(CORR_KT + c + padf) * D + sizeof(double) * CORR_KT;
    // few channels, single-wave workgroups: the waves of a workgroup split the frames of one
    // sub-tile instead (wpe_corr_ksplit_kernel; GSS_VARIANT corr_ksplit=1: single waves)
    const int ksplit_forced = gss_variant("corr_ksplit", 0);
    const int corr_ks = (corr_ts == 1 && corr_nw == 1 && ksplit_forced != 1)
                            ? (ksplit_forced == 2 ? 2 : 4) : 1;      // (8: 0.687 vs 0.667 ms per utterance)
    using corr_fn_t = void (*)(const cplx *, const double *, int, int64_t, int, int, int, int,
                               const CorrTile *, int, cplx *, cplx *);
    corr_fn_t ksplit_fn = nullptr;
    if (corr_ks > 1) {
        ksplit_fn = corr_stg5 ? (corr_ks == 2 ? wpe_corr_ksplit_kernel<2, 5> : wpe_corr_ksplit_kernel<4, 5>)
                              : (corr_ks == 2 ? wpe_corr_ksplit_kernel<2, 8> : wpe_corr_ksplit_kernel<4, 8>);
        corr_lds = std::max(corr_lds * corr_ks, sizeof(cplx) * 256 * (size_t)(corr_ks - 1));
    }
    const size_t panel_lds = sizeof(cplx) * UDP_SIZE + sizeof(double) * CH_NB;
    constexpr int apply_ta = 2;
    constexpr int apply_nwv = 4;       // (2 or 3 waves per workgroup: 1.41 / 1.47 vs 1.37 ms, round 4)
    const int apply_frames = 16 * apply_ta * apply_nwv;
    auto apply_fn = D <= 16 ? wpe_apply_kernel<apply_ta, 1> : wpe_apply_kernel<apply_ta, 2>;
    // the common case -- four or more channels: one carry per k-step; taps * D a multiple of 8:
    // no row mask (GSS_VARIANT apply_generic: the general form)
    // (the k loop runs two k-steps per trip: without the row mask the k-step count must be
    // even, i.e. 8 | taps * D -- 36 rows are 9 k-steps and the tenth would add clamped garbage)
    if (D >= 4 && !gss_variant_set("apply_generic")) {
        if (n % 8 == 0)
            apply_fn = D <= 16 ? wpe_apply_kernel<apply_ta, 1, 4, 1, false>
                               : wpe_apply_kernel<apply_ta, 2, 4, 1, false>;
        else
            apply_fn = D <= 16 ? wpe_apply_kernel<apply_ta, 1, 4, 1, true>
                               : wpe_apply_kernel<apply_ta, 2, 4, 1, true>;
    }
    const size_t apply_lds = sizeof(cplx) * (size_t)(apply_frames + c + 2) * (D | 1);
    // frame phases packed into the N dimension (wpe_apply_packed_kernel): pick the number of
    // phases that minimises the MFMAs per frame, ceil(PH D / 16) (n + (PH - 1) D) / PH
    int apply_ph = 1, apply_nt = (D + 15) / 16;
    size_t packed_lds = 0;
    {
        const int ph_env = gss_variant("apply_ph", 0);   // tests
        bool ph_env_taken = ph_env <= 1;
        double best = (double)((D + 15) / 16) * n;
        for (int ph = 2; ph <= 4; ++ph) {
            if (ph_env > 0 && ph != ph_env) continue;
            const int nt = (ph * D + 15) / 16;
            if (nt > 3 || D < 2) continue;
            const double cost = (double)nt * (n + (ph - 1) * D) / ph;
            const int fl = 64 * ph + c + 2;
            const size_t lds = sizeof(cplx) * ((size_t)fl * (D | 1) + fl / ph + 1 + (size_t)(n + 1) * (D | 1));
            // (with G and the window in LDS only small channel counts keep two workgroups per CU;
            // at D = 24 / 20 the packed form owns a CU alone, nothing overlaps its staging
            // prologue, and it is slower: 1.82 vs 1.35 ms, 3.10 vs 2.28 ms per utterance)
            if (lds > (ph_env == ph ? 160 : 64) * 1024) continue;
            if (cost < 0.75 * best || ph_env == ph) {
                ph_env_taken = ph_env_taken || ph_env == ph;
                best = cost;
                apply_ph = ph;
                apply_nt = nt;
                packed_lds = lds;
            }
        }
        if (ph_env == 1) apply_ph = 1;
        if (!ph_env_taken) {
            static bool warned = false;      // a forced variant that does not exist for this shape
            if (!warned)
                fprintf(stderr, "libgss_hip: GSS_VARIANT apply_ph=%d is not available for D=%d taps=%d "
                                "(column tiles or LDS); using the default\n", ph_env, D, taps);
            warned = true;
        }
    }
    using apply_packed_t = void (*)(const cplx *, const cplx *, int, int64_t, int, int, int, cplx *);
    apply_packed_t packed_fn = nullptr;
    // 17 - 24 channels (20, 24: all microphones of five / six arrays): two frame phases make
    // 2 D columns = 3 column tiles (the unpacked form's second tile is half empty: 17.5 % fewer
    // MFMAs), with G read from global memory -- in LDS it would own the CU.  Needs 4 | D (the
    // phases' first / last k-steps are whole k-steps).  MEASURED in round 6 and NOT the default
    // (GSS_VARIANT apply_gglobal switches it on): 1.41 ms per utterance with the operands one
    // k-step ahead, 1.33 with two, against 1.27 for the unpacked kernel -- one row tile per wave
    // is 9 MFMAs per four operand loads instead of 12, and three of the four come from L2.
    const bool apply_gglobal = apply_ph == 1 && D > 16 && 2 * D <= 48 && D % 4 == 0 && n % 4 == 0 &&
                               gss_variant_set("apply_gglobal") && gss_variant("apply_ph", 0) == 0 &&
                               !gss_variant_set("apply_generic");
    if (apply_gglobal) {
        apply_ph = 2;
        apply_nt = 3;
        const int fl = 64 * 2 + c + 2;
        packed_lds = sizeof(cplx) * ((size_t)fl * (D | 1) + fl / 2 + 1);
        packed_fn = wpe_apply_packed_kernel<2, 3, false>;
    } else
    if (apply_ph > 1) {
        static const apply_packed_t table[3][3] = {
            {wpe_apply_packed_kernel<2, 1>, wpe_apply_packed_kernel<2, 2>, wpe_apply_packed_kernel<2, 3>},
            {wpe_apply_packed_kernel<3, 1>, wpe_apply_packed_kernel<3, 2>, wpe_apply_packed_kernel<3, 3>},
            {wpe_apply_packed_kernel<4, 1>, wpe_apply_packed_kernel<4, 2>, wpe_apply_packed_kernel<4, 3>}};
        packed_fn = table[apply_ph - 2][apply_nt - 1];
        // few staged elements per thread (one array): the form with 4 + 1 staging slots
        static const apply_packed_t table_small[3][3] = {
            {wpe_apply_packed_kernel<2, 1, true, 4, 1>, wpe_apply_packed_kernel<2, 2, true, 4, 1>, wpe_apply_packed_kernel<2, 3, true, 4, 1>},
            {wpe_apply_packed_kernel<3, 1, true, 4, 1>, wpe_apply_packed_kernel<3, 2, true, 4, 1>, wpe_apply_packed_kernel<3, 3, true, 4, 1>},
            {wpe_apply_packed_kernel<4, 1, true, 4, 1>, wpe_apply_packed_kernel<4, 2, true, 4, 1>, wpe_apply_packed_kernel<4, 3, true, 4, 1>}};
        if ((64 * apply_ph + c + 2) * D <= 4 * 256 && n * D <= 256 && !gss_variant_set("apply_generic"))
            packed_fn = table_small[apply_ph - 2][apply_nt - 1];
        if (packed_lds > 64 * 1024)
            GSS_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(packed_fn),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)packed_lds));
    }
    GSS_REQUIRE(ctx, corr_lds <= 160 * 1024 && apply_lds <= 160 * 1024, GSS_ERR_UNSUPPORTED,
                "wpe: taps=%d D=%d needs more LDS than a CU has", taps, D);
    if (apply_lds > 64 * 1024)
        GSS_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(apply_fn),
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)apply_lds));
    if (corr_lds > 64 * 1024)
        GSS_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(corr_fn),
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)corr_lds));
    // 32 x 32 tiles: resident workgroups fed from per-XCD item queues, the window by LDS-DMA
    // into two alternating LDS windows (wpe_corr_persist_kernel); needs the slab's byte
    // offsets in 31 bits and two windows in half a CU's LDS, else the register-staged kernel
    const int corr_pieces = (int)(((size_t)(CORR_KT + c + padf) * D + 63) / 64);
    const size_t corr_dma_lds = 2 * (sizeof(cplx) * 64 * (size_t)corr_pieces + sizeof(double) * CORR_KT);
    const bool corr_persist = corr_ts == 2 && (int64_t)T * D * 16 < (1LL << 31) &&
                              corr_dma_lds <= 80 * 1024;
    // (two windows of one workgroup take half a CU's LDS at most, so that two workgroups share
    // a CU; resident workgroups do not care how the item count packs into rounds: 4 waves)
    if (corr_persist && !nw_forced) corr_nw = 4;
    // GSS_VARIANT corr_blocked (read on every call): chunk-wise accumulation of R and P
    const bool corr_blocked = corr_persist && gss_variant_set("corr_blocked");
    auto corr_persist_fn = corr_blocked ? (corr_nw == 2 ? wpe_corr_persist_kernel<2, true>
                                                        : wpe_corr_persist_kernel<4, true>)
                                        : (corr_nw == 2 ? wpe_corr_persist_kernel<2>
                                                        : wpe_corr_persist_kernel<4>);
    CorrQueue corr_queue{(ntiles + corr_nw - 1) / corr_nw, 0};
    int corr_slots = 0;
    if (corr_persist) {
        // heavy groups: the heaviest wave needs 3 or 4 of its tile's 4 sub-tiles
        for (int g = 0; g < corr_queue.ngroups; ++g)
            if (__builtin_popcount(tiles[g * corr_nw].mask) >= 3) corr_queue.gh = g + 1;
        if (corr_dma_lds > 64 * 1024)
            GSS_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(corr_persist_fn),
                                                   hipFuncAttributeMaxDynamicSharedMemorySize,
                                                   (int)corr_dma_lds));
        int per_cu = 0, cus = 0;
        GSS_HIP_CHECK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(
                               &per_cu, reinterpret_cast<const void *>(corr_persist_fn), 64 * corr_nw,
                               corr_dma_lds));
        GSS_HIP_CHECK(ctx, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
        corr_slots = std::min(std::max(per_cu, 1) * cus, corr_queue.ngroups * F);
    }

    for (int it = 0; it < iterations; ++it) {
        const cplx *cur = it == 0 ? Y : X;
        {
            GSS_PROF(ctx, "wpe_power");
            GSS_TRY(wpe_power_launch(ctx, cur, F, T, D, psd_context, raw, w));
        }
        if (corr_persist) {
            GSS_HIP_CHECK(ctx, hipMemsetAsync(corr_counters, 0, 8 * 16 * sizeof(int), ctx->stream));
            GSS_PROF(ctx, "wpe_corr");
            hipLaunchKernelGGL(corr_persist_fn, dim3(corr_slots), dim3(64 * corr_nw), corr_dma_lds,
                               ctx->stream, Y, w, F, T, D, n, c, corr_pieces, tiles_dev, ntiles,
                               corr_queue, corr_counters, R, P);
            GSS_LAUNCH_CHECK(ctx, "wpe_corr_persist_kernel");
        } else {
            GSS_PROF(ctx, "wpe_corr");
            if (ksplit_fn)
                hipLaunchKernelGGL(ksplit_fn, dim3(xcd_grid(ntiles, F)), dim3(64 * corr_ks), corr_lds,
                                   ctx->stream, Y, w, F, T, D, n, c, padf, tiles_dev, ntiles, R, P);
            else
                hipLaunchKernelGGL(corr_fn, dim3(xcd_grid((ntiles + corr_nw - 1) / corr_nw, F)),
                                   dim3(64 * corr_nw), corr_lds,
                                   ctx->stream, Y, w, F, T, D, n, c, padf, tiles_dev, ntiles, R, P);
            GSS_LAUNCH_CHECK(ctx, "wpe_corr_kernel");
        }
        {
            const int nblk = (n + CH_NB - 1) / CH_NB;
            // GSS_PROF_DETAIL=1: one profile row per block column (tools/wpe_kprof.py)
            const bool detail = gss_variant_set("prof_detail");
            static const char *trsm_names[] = {"wpe_chol_trsm_J0", "wpe_chol_trsm_J1", "wpe_chol_trsm_J2",
                                               "wpe_chol_trsm_J3", "wpe_chol_trsm_J4", "wpe_chol_trsm_J5+"};
            static const char *upd_names[] = {"wpe_chol_update_J0", "wpe_chol_update_J1", "wpe_chol_update_J2",
                                              "wpe_chol_update_J3", "wpe_chol_update_J4", "wpe_chol_update_J5+"};
            for (int J = 0; J < nblk; ++J) {
                const int j0 = J * CH_NB, nb = std::min(CH_NB, n - j0);
                // (blocks J > 0 are factored by group 0 of the preceding trailing update)
                if (J == 0 || upd_ndiag[J - 1] == 0) {
                    GSS_PROF(ctx, "wpe_chol_diag");
                    hipLaunchKernelGGL(chol_diag_kernel, dim3(F), dim3(256), 0, ctx->stream, R, n, j0,
                                       zero_pivots);
                    GSS_LAUNCH_CHECK(ctx, "chol_diag_kernel");
                }
                {
                    GSS_PROF(ctx, detail ? trsm_names[std::min(J, 5)] : "wpe_chol_trsm");
                    const int npanel = (n - j0 - nb + 15) / 16 + (D + 15) / 16;
                    hipLaunchKernelGGL(chol_trsm_kernel, dim3(xcd_grid((npanel + 3) / 4, F)),
                                       dim3(256), panel_lds, ctx->stream, R, P, F, n, D, j0);
                    GSS_LAUNCH_CHECK(ctx, "chol_trsm_kernel");
                }
                const int nupd = upd_count[J];
                if (nupd > 0) {
                    GSS_PROF(ctx, detail ? upd_names[std::min(J, 5)] : "wpe_chol_update");
                    const int ndiag = upd_ndiag[J];
                    const dim3 g((ndiag > 0 ? (F + 7) / 8 * 8 : 0) +
                                 xcd_grid((nupd - ndiag + UPD_WAVES - 1) / UPD_WAVES, F)), b(64 * UPD_WAVES);
                    const UpdTile *tl = upd_dev + upd_start[J];
                    hipLaunchKernelGGL((chol_update_kernel<1, 1, true>), g, b, 0, ctx->stream, R, P, F, n, D, upd_j0[J],
                                       std::min(upd_k[J], n - upd_j0[J]), tl, nupd, ndiag,
                                       (J + 1) * CH_NB, zero_pivots);
                    GSS_LAUNCH_CHECK(ctx, "chol_update_kernel");
                }
            }
            {
                GSS_PROF(ctx, "wpe_backsolve");
                hipLaunchKernelGGL(chol_backsolve_kernel, dim3(F), dim3(256), 0, ctx->stream, R, P, n,
                                   D);
                GSS_LAUNCH_CHECK(ctx, "chol_backsolve_kernel");
            }
        }
        {
            GSS_PROF(ctx, "wpe_apply");
            if (packed_fn) {
                const int wg_frames = 64 * apply_ph;
                hipLaunchKernelGGL(packed_fn, dim3(xcd_grid((int)((T + wg_frames - 1) / wg_frames), F)),
                                   dim3(256), packed_lds, ctx->stream, Y, P, F, T, D, n, c, X);
            } else {
                hipLaunchKernelGGL(apply_fn,
                                   dim3(xcd_grid((int)((T + apply_frames - 1) / apply_frames), F)),
                                   dim3(64 * apply_nwv), apply_lds, ctx->stream, Y, P, F, T, D, n, c, X);
            }
            GSS_LAUNCH_CHECK(ctx, "wpe_apply_kernel");
        }
    }
    // (two parts: the caller copies the count after both have finished)
    if (part < 0)
        GSS_HIP_CHECK(ctx, hipMemcpyAsync(ctx->status_host + 2, zero_pivots, sizeof(int32_t),
                                          hipMemcpyDeviceToHost, ctx->stream));
    return GSS_OK;
}


int wpe_copy_zero_pivots(gss_ctx *ctx) {
    if (!ctx->wpe_tiles) return GSS_OK;
    const int32_t *zero_pivots =
        reinterpret_cast<const int32_t *>(reinterpret_cast<const CorrTile *>(ctx->wpe_tiles) + 1024 + 4096);
    GSS_HIP_CHECK(ctx, hipMemcpyAsync(ctx->status_host + 2, zero_pivots, sizeof(int32_t),
                                      hipMemcpyDeviceToHost, ctx->stream));
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
