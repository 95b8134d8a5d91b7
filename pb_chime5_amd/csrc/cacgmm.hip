// Guided CACGMM:  GSS.__call__ (/root/reference/pb_chime5/core.py:154-214) ->
// pb_bss CACGMMTrainer.fit / CACGMM.predict, all 513 frequencies at once.
//
// One EM iteration = three launches (the split keeps register use low enough for
// 4+ waves per SIMD, which is what hides the LDS / scalar-load latencies):
//   em_estep grid (chunks, F): q_kt = | y^H B_k^-1 y |,  log p = -D ln q - ln det,
//            softmax with weights pi_k and the activity mask, clip to [eps, 1-eps];
//            writes the M-step weights  w_kt = gamma_kt / q_kt  (F,K,T) and partial
//            sums of gamma.  (First iteration: gamma from the activity, q = 1;
//            final predict: writes gamma.)
//   em_mstep grid (chunks, F): partial sums of  w_kt y y^H  over the chunk's frames
//   em_eig   grid (K, F): reduce the partial sums, B_k = D * sum / sum(gamma), then
//            B_k^-1 and ln det (Cholesky when no eigenvalue can be floored, Jacobi
//            eigendecomposition with the 1e-10 floor otherwise) and pi_k.
//
// Hermitian structure: with P_de(t) = y_d conj(y_e) (class independent, d <= e)
//    q_kt   = sum_{d<=e}  Re(Mq_k,de) Re(P_de) + Im(Mq_k,de) Im(P_de)
//    B_k,de = sum_t w_kt P_de(t)
// where Mq holds B_k^-1 with the off-diagonals doubled.  That is 4 real FMAs per
// (entry, class, frame) instead of 12 for the dense form.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "gss_internal.h"
#include "dense_wave.h"
#include "jacobi.h"

namespace {

#ifndef GSS_CHOL_PRIO
#define GSS_CHOL_PRIO 1       // the same in em_chol (loads 3, sweep 2, product 0): -2 %
#endif
#ifndef GSS_ESTEP_PRIO
#define GSS_ESTEP_PRIO 1      // wave priority falling with progress in the register-form E-step
#endif
constexpr int EM_TILE = 64;          // frames per tile (one per lane of a wave)
constexpr int EM_TS = EM_TILE + 1;   // padded LDS row stride (complex elements)

enum { MODE_FIRST = 0, MODE_EM = 1, MODE_PREDICT = 2 };

// Wave priority that falls with the progress of a workgroup (quarters of its work done: 3, 2,
// 1, 0).  For launches whose workgroups are all resident from the start and should END
// together: the issue arbiter serves the oldest wave first, so the co-resident workgroups of a
// CU otherwise finish one after the other and the tail of the launch runs at a fraction of
// the occupancy; with this whoever is ahead waits for the others.
__device__ __forceinline__ void set_progress_priority(int done, int total) {
    const int q4 = 4 * done / total;
    if (q4 <= 0) __builtin_amdgcn_s_setprio(3);
    else if (q4 == 1) __builtin_amdgcn_s_setprio(2);
    else if (q4 == 2) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
}

// Static balanced partition of the M-step (D > 4; measured at T = 2172: D = 8 / 10 / 12 / 24
// -25 / -21 / -18 / -14 % against equal chunks per frequency): the (frequency, 64-frame tile) items of an
// utterance, frequency major, are cut into S equal runs, one per RESIDENT workgroup (S = what
// the chip holds at once), instead of F x nch workgroups of nch equal chunks per frequency:
// every workgroup starts at t = 0 and ends with the others (the chunked launch drained for
// the last fifth of its span at falling residency: tools/wcov_trace.py), and a frequency is
// cut into 2 - 3 segments instead of 5 - 6, so the fixed cost of a segment (cross-group
// reduction, scattered store of the partial sums: 12 - 23 % of a workgroup's life) and the
// partial sums em_chol re-reads halve.  Segment j of frequency f (its j-th run) writes
// part[(f * maxseg + j) ...]; S = 0: the chunked form.  The partition depends on (F, T, S)
// only -- the same summation order on every run on the same chip.
struct MsegPlan {
    int S, ntile, base, rem, maxseg;
};
__host__ __device__ __forceinline__ int mseg_begin(const MsegPlan &p, int slot) {
    return slot * p.base + (slot < p.rem ? slot : p.rem);
}
__host__ __device__ __forceinline__ int mseg_slot_of(const MsegPlan &p, int item) {
    const int head = p.rem * (p.base + 1);
    return item < head ? item / (p.base + 1) : p.rem + (item - head) / p.base;
}
// number of segments (partial sums) of frequency f
__host__ __device__ __forceinline__ int mseg_count(const MsegPlan &p, int f) {
    return mseg_slot_of(p, f * p.ntile + p.ntile - 1) - mseg_slot_of(p, f * p.ntile) + 1;
}

struct EmArgs {
    const cplx *Y;          // (F,T,D)
    const uint8_t *act;     // (K,act_stride), first T columns used
    int64_t act_stride;
    const double *logdet;   // (F,K)
    const double *pi;       // (F,K)
    double *W;              // (F,K,T) M-step weights gamma / q
    cplx *Bp;               // (F,NCH,K,NE)
    double *Sg;             // (F,NCH,K)
    double *gamma;          // (F,K,T)   MODE_PREDICT only
    int64_t T;
    int F, D, NE, nch, chunk_frames;
    int masked;             // multiply the activity mask into the posteriors
    double aff_eps;         // clip, 0 = none
    MsegPlan mseg;          // M-step partition (S = 0: nch chunks per frequency)
};

// Load frames [t0, t0+64) of one frequency into LDS as ys[d][tl], optionally unit
// normalised per frame (pb_bss normalize_observation).  `scratch` holds 4 * 64
// doubles.  Thread (tl, g) loads channels d = g, g+4, ...
// NV: elements of the 64 x D tile per thread = ceil(D / 4) (8 covers D <= 32; callers that know
// their channel count pass the exact number: the pass is a row of predicated loads and stores)
template <bool NORMALISE, int NV = 8>
__device__ __forceinline__ void load_tile(const cplx *Yf, int D, int64_t t0, int64_t c1, int tl,
                                          int g, cplx *ys, double *scratch) {
    // The 64 x D elements of the tile are one contiguous run of the (T, D) slab: consecutive
    // threads load consecutive elements (a lane walking its own frame touched 64 cache lines
    // per load instruction; em_prepare 0.073 -> 0.060 ms) and scatter them to ys[d][frame];
    // (frame, channel) of element idx + 256 follows from that of idx with one carry.
    const int tid = g * EM_TILE + tl;
    const int nfr = (int)min((int64_t)EM_TILE, c1 - t0);
    const int total = nfr * D;
    const cplx *src = Yf + t0 * D;
    const int qf = 256 / D, rf = 256 - qf * D;
    cplx v[NV];   // D <= 4 NV
#pragma unroll
    for (int j = 0; j < NV; ++j) {
        const int idx = tid + 256 * j;
        v[j] = idx < total ? src[idx] : c_make(0.0, 0.0);
    }
    {
        int fr = tid / D, d = tid - fr * D;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            if (tid + 256 * j < EM_TILE * D) ys[d * EM_TS + fr] = v[j];     // (zeros past the last frame)
            fr += qf;
            d += rf;
            if (d >= D) {
                d -= D;
                ++fr;
            }
        }
    }
    if (NORMALISE) {
        __syncthreads();
        // per frame: the four groups' partial sums over d = g, g + 4, ... in that order, then
        // the groups in order -- the summation order of the loader this one replaces
        double nrm = 0.0;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int d = g + 4 * j;
            nrm += d < D ? c_abs2(ys[d * EM_TS + tl]) : 0.0;
        }
        scratch[g * EM_TILE + tl] = nrm;
        __syncthreads();
        nrm = scratch[tl] + scratch[EM_TILE + tl] + scratch[2 * EM_TILE + tl] +
              scratch[3 * EM_TILE + tl];
        nrm = sqrt(nrm);
        if (nrm < GSS_TINY) nrm = GSS_TINY;      // np.maximum(norm, tiny): NaN stays NaN
#pragma unroll
        for (int j = 0; j < NV; ++j) {
            const int d = g + 4 * j;
            if (d < D) {
                const cplx y = ys[d * EM_TS + tl];
                ys[d * EM_TS + tl] = c_make(y.x / nrm, y.y / nrm);
            }
        }
    }
}

// ------------------------------------------------------------------ E-step
// Mq is a separate __restrict__ argument: its reads are wave-uniform and are issued
// as scalar loads, so the model never occupies LDS bandwidth.
template <int K, int MODE>
__global__ __launch_bounds__(256) void em_estep_kernel(EmArgs a, const cplx *__restrict__ Mq) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D = a.D, NE = a.NE;
    cplx *ys = reinterpret_cast<cplx *>(smem);                       // D * EM_TS
    double *qpart = reinterpret_cast<double *>(ys + D * EM_TS);      // 4 * K * EM_TILE
    double *lpS = qpart + 4 * K * EM_TILE;                           // K * EM_TILE
    double *qqS = lpS + K * EM_TILE;                                 // K * EM_TILE
    double *ldet = qqS + K * EM_TILE;                                // K
    double *pis = ldet + K;                                          // K
    double *vvS = qpart;   // the partial sums are dead once lp / qq are written

    int f, chunk;
    if (!xcd_group_map(a.nch, a.F, f, chunk)) return;
    const int tid = threadIdx.x;
    const int tl = tid & 63;
    const int g = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave index, uniform
    const int64_t T = a.T;
    const int64_t c0 = (int64_t)chunk * a.chunk_frames;
    const int64_t c1 = c0 + a.chunk_frames < T ? c0 + a.chunk_frames : T;
    const cplx *Yf = a.Y + (int64_t)f * T * D;
    const cplx *Mf = Mq + (int64_t)f * NE * K;

    if (MODE != MODE_FIRST && tid < K) {
        ldet[tid] = a.logdet[f * K + tid];
        pis[tid] = a.pi[f * K + tid];
    }
    constexpr int NS = (K + 3) / 4;   // classes per wave in the softmax: g, g + 4, ...
    double sg[NS];               // sums of gamma for those classes
#pragma unroll
    for (int s = 0; s < NS; ++s) sg[s] = 0.0;

    for (int64_t t0 = c0; t0 < c1; t0 += EM_TILE) {
        const int64_t t = t0 + tl;
        const bool valid = t < c1;
        if (MODE == MODE_FIRST) {
            // GSS initialisation (core.py:156-160): where(act == 0, 1e-10, act) / sum_k
            double ssum = 0.0;
#pragma unroll
            for (int k = 0; k < K; ++k)
                ssum += (valid && a.act[(int64_t)k * a.act_stride + t]) ? 1.0 : 1e-10;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const int k = g + 4 * s;
                if (k < K && valid) {
                    const double v = a.act[(int64_t)k * a.act_stride + t] ? 1.0 : 1e-10;
                    const double gk = v / ssum;
                    sg[s] += gk;
                    a.W[((int64_t)f * K + k) * T + t] = gk;          // quadratic form = 1
                }
            }
            continue;
        }
        __syncthreads();
        load_tile<true>(Yf, D, t0, c1, tl, g, ys, qpart);
        __syncthreads();

        // ---- quadratic forms.  Wave g takes the row pairs (p, D-1-p), p = g, g+4, ...
        // (every pair holds D+1 entries: balanced); y_row stays in registers, the
        // model row is read with wave-uniform (scalar) loads.
        double q[K];
#pragma unroll
        for (int k = 0; k < K; ++k) q[k] = 0.0;
        const int npairs = (D + 1) / 2;
        for (int p = g; p < npairs; p += 4) {
#pragma unroll 1
            for (int side = 0; side < 2; ++side) {
                const int r = side == 0 ? p : D - 1 - p;
                if (side == 1 && r == p) break;          // middle row of an odd D
                const cplx y1 = ys[r * EM_TS + tl];
                const cplx *mrow = Mf + (int64_t)tri_index(r, r, D) * K;
                {
                    const double pr = y1.x * y1.x + y1.y * y1.y;
#pragma unroll
                    for (int k = 0; k < K; ++k) q[k] = fma(mrow[k].x, pr, q[k]);
                }
                for (int d2 = r + 1; d2 < D; ++d2) {
                    mrow += K;
                    const cplx y2 = ys[d2 * EM_TS + tl];
                    const double pr = y1.x * y2.x + y1.y * y2.y;
                    const double pim = y1.y * y2.x - y1.x * y2.y;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const cplx m = mrow[k];
                        q[k] = fma(m.x, pr, q[k]);
                        q[k] = fma(m.y, pim, q[k]);
                    }
                }
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) qpart[(g * K + k) * EM_TILE + tl] = q[k];
        __syncthreads();
        // ---- log-likelihoods: thread (tl, g) handles classes g, g + 4, ...
        double myq[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) myq[s] = 1.0;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int k = g + 4 * s;
            if (k < K) {
                double qv = qpart[k * EM_TILE + tl] + qpart[(K + k) * EM_TILE + tl] +
                            qpart[(2 * K + k) * EM_TILE + tl] + qpart[(3 * K + k) * EM_TILE + tl];
                qv = fmax(fabs(qv), GSS_TINY);
                myq[s] = qv;
                lpS[k * EM_TILE + tl] = -(double)D * log(qv) - ldet[k];
            }
        }
        __syncthreads();
        double mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < K; ++k) mx = fmax(mx, lpS[k * EM_TILE + tl]);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int k = g + 4 * s;
            if (k < K) {
                double v = exp(lpS[k * EM_TILE + tl] - mx) * pis[k];
                if (a.masked) v *= (valid && a.act[(int64_t)k * a.act_stride + t]) ? 1.0 : 0.0;
                vvS[k * EM_TILE + tl] = v;
            }
        }
        __syncthreads();
        double ssum = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) ssum += vvS[k * EM_TILE + tl];
        ssum = fmax(ssum, GSS_TINY);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int k = g + 4 * s;
            if (k < K && valid) {
                double gam = vvS[k * EM_TILE + tl] / ssum;
                if (a.aff_eps != 0.0) gam = fmin(fmax(gam, a.aff_eps), 1.0 - a.aff_eps);
                if (MODE == MODE_PREDICT) {
                    a.gamma[((int64_t)f * K + k) * T + t] = gam;
                } else {
                    sg[s] += gam;
                    a.W[((int64_t)f * K + k) * T + t] = gam / fmax(myq[s], 10.0 * GSS_TINY);
                }
            }
        }
    }
    if (MODE == MODE_PREDICT) return;
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int k = g + 4 * s;
        const double tot = wave_sum(sg[s]);
        if (k < K && tl == 0) a.Sg[((int64_t)f * a.nch + chunk) * K + k] = tot;
    }
}

// ------------------------------------------------------------------ E-step, register form
// For the channel counts of the corpus (all / outer / reference-array microphones of its
// 5 and 6 array sessions: D = 24, 20, 12, 10, 4) the E-step runs without LDS and without
// any barrier: the unit-normalised observation is kept once per utterance in (F, D, T) layout (em_prepare), a lane owns
// one frame, pulls its D channel values with coalesced loads into registers, walks the
// packed upper triangle fully unrolled with the model row in SGPRs (scalar loads), and
// finishes the softmax in registers.  Lanes never exchange data.
template <int NV>
__global__ __launch_bounds__(256) void em_prepare_kernel(const cplx *__restrict__ Y, int F,
                                                         int64_t T, int D,
                                                         cplx *__restrict__ Yn,
                                                         int *__restrict__ zero_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx *ys = reinterpret_cast<cplx *>(smem);                  // D * EM_TS
    double *scratch = reinterpret_cast<double *>(ys + D * EM_TS);
    int f, tile;
    const int ntile = (int)((T + EM_TILE - 1) / EM_TILE);
    if (!xcd_group_map(ntile, F, f, tile)) return;
    const int tid = threadIdx.x, tl = tid & 63, g = tid >> 6;
    const int64_t t0 = (int64_t)tile * EM_TILE;
    load_tile<true, NV>(Y + (int64_t)f * T * D, D, t0, T, tl, g, ys, scratch);
    __syncthreads();
    const int64_t t = t0 + tl;
    bool nonzero = false;
    if (t < T)
        for (int d = g; d < D; d += 4) {
            const cplx v = ys[d * EM_TS + tl];
            Yn[((int64_t)f * D + d) * T + t] = v;
            nonzero = nonzero || v.x != 0.0 || v.y != 0.0;
        }
    // Frames whose normalised observation is EXACTLY zero (digital silence): their quadratic
    // forms sit on the clamp max(|q|, tiny), which is the one place where the scale of B_k does
    // not cancel between -D ln q and -ln det -- the frequency then has to take the reference's
    // eigenvalue-normalised model update (see em_chol_kernel).  One word per (frequency, tile),
    // written by every workgroup (no initialisation, no atomics).
    scratch[g * EM_TILE + tl] = nonzero ? 1.0 : 0.0;
    __syncthreads();
    const bool zero_frame = t < T && g == 0 &&
                            scratch[tl] + scratch[EM_TILE + tl] + scratch[2 * EM_TILE + tl] +
                                    scratch[3 * EM_TILE + tl] == 0.0;
    const int any_zero = __syncthreads_or(zero_frame ? 1 : 0);
    if (tid == 0) zero_tiles[(int64_t)f * ntile + tile] = any_zero;
}

// whether frequency f holds a frame on the clamp (em_prepare_kernel's flags; one wave asks)
__device__ __forceinline__ bool frequency_has_zero_frames(const int *__restrict__ zero_tiles,
                                                          int ntile, int f, int lane) {
    int z = 0;
    for (int i = lane; i < ntile; i += 64) z |= zero_tiles[(int64_t)f * ntile + i];
    return __any(z != 0);
}

template <int K, int D, int MODE>
__global__ __launch_bounds__(256) void em_estep_reg_kernel(EmArgs a, const cplx *__restrict__ Mq,
                                                           const cplx *__restrict__ Yn) {
    constexpr int NE = D * (D + 1) / 2;
    const int64_t T = a.T;
    // waves are independent (no LDS, no barrier): the workgroup size only decides how the
    // launch packs into rounds of resident workgroups (see estep_waves_per_block)
    const int wpb = blockDim.x >> 6;
    const int ntile = (int)((T + 64 * wpb - 1) / (64 * wpb));
    int f, tile;
    if (!xcd_group_map(ntile, a.F, f, tile)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t t = ((int64_t)tile * wpb + wave) * 64 + lane;
    const bool valid = t < T;
    const int64_t tc = valid ? t : T - 1;
    const cplx *Mf = Mq + (int64_t)f * NE * K;
    const cplx *yf = Yn + (int64_t)f * D * T + tc;

#if GSS_ESTEP_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    cplx y[D];
#pragma unroll
    for (int d = 0; d < D; ++d) y[d] = yf[(int64_t)d * T];

    double q[K];
#pragma unroll
    for (int k = 0; k < K; ++k) q[k] = 0.0;
    // Walk the packed upper triangle with a ring of model rows: the row of entry e + P
    // is requested (scalar loads, wave uniform) before entry e is evaluated, and a
    // scheduling barrier per entry keeps the compiler from hoisting every row of the
    // fully unrolled triangle to the top (which would spill thousands of SGPRs).
    constexpr int P = 2, RING = P + 1;
    cplx ring[RING][K];
#pragma unroll
    for (int j = 0; j < P; ++j)
#pragma unroll
        for (int k = 0; k < K; ++k) ring[j][k] = Mf[j * K + k];
    int e = 0;
#pragma unroll
    for (int d1 = 0; d1 < D; ++d1) {
#pragma unroll
        for (int d2 = d1; d2 < D; ++d2) {
            if (e + P < NE) {
#pragma unroll
                for (int k = 0; k < K; ++k) ring[(e + P) % RING][k] = Mf[(e + P) * K + k];
            }
            const double pr = y[d1].x * y[d2].x + y[d1].y * y[d2].y;
            if (d1 == d2) {
                // (the imaginary parts of a diagonal entry and of its product are zero: skipping
                // them changes no bit of q)
#pragma unroll
                for (int k = 0; k < K; ++k) q[k] = fma(ring[e % RING][k].x, pr, q[k]);
            } else {
                const double pim = y[d1].y * y[d2].x - y[d1].x * y[d2].y;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const cplx m = ring[e % RING][k];
                    q[k] = fma(m.x, pr, q[k]);
                    q[k] = fma(m.y, pim, q[k]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            ++e;
#if GSS_ESTEP_PRIO
            // priority falls with the wave's progress: the waves of a SIMD converge instead
            // of finishing oldest first, and the end of the launch is not a row of single
            // waves finishing alone
            if (e == NE / 4) __builtin_amdgcn_s_setprio(2);
            if (e == NE / 2) __builtin_amdgcn_s_setprio(1);
            if (e == 3 * NE / 4) __builtin_amdgcn_s_setprio(0);
#endif
        }
    }
    // Posterior  pi_k exp(-D ln q_k - ln det_k) / sum  without a logarithm per frame: relative
    // to the class with the smallest q it is  (q_min / q_k)^D c_k  with
    // c_k = pi_k exp(ln det_min - ln det_k)  -- the common factor cancels in the normalisation.
    // c_k is one exponential per class and WAVE (lane k computes it, an SGPR pair hands it to
    // all lanes) instead of K logarithms and K exponentials per FRAME: the transcendental
    // functions were 600 of the kernel's 5200 instructions per wave, and the integer power is
    // the more accurate form (24 roundings against the 1e-13 of exp(500 +- ...)).  One
    // reciprocal per class serves the ratio and the M-step weight gamma / q, one more the
    // normalisation.  A frame whose terms all underflow (sum < 1e-280: the classes' ln det
    // hundreds apart AND the q ratios against them) takes the log form below, which is also
    // the form of em_estep_kernel.
    // (The walk sits at the 128 registers of four waves per SIMD and at the SGPR limit, and
    // nothing orders the epilogue's loads after it: activity bytes, ln det, pi -- or the lane
    // masks made of them -- were moved above the walk and spilled its ring of model rows lane
    // by lane (5 x slower).  So their addresses depend on the walk's result: `dep` is 0
    // unless q is NaN.)
    const int dep = q[0] != q[0] ? 1 : 0;
    const int tcl = (int)tc - dep;
    auto lane_bcast = [](double x, int k) {
        const unsigned long long u = __double_as_longlong(x);
        const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, k);
        const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), k);
        return __longlong_as_double(((unsigned long long)hi << 32) | lo);
    };
    double cls, ldl, pil;
    {
        const int kk = (lane < K ? lane : K - 1) - (lane > 0 ? dep : 0);
        ldl = a.logdet[f * K + kk];
        pil = a.pi[f * K + kk];
        double ldmin = INFINITY;
#pragma unroll
        for (int k = 0; k < K; ++k) ldmin = fmin(ldmin, lane_bcast(ldl, k));
        cls = pil * exp(ldmin - ldl);
    }
    double qmin = INFINITY;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        q[k] = fmax(fabs(q[k]), GSS_TINY);
        qmin = fmin(qmin, q[k]);
    }
    uint8_t act[K];
#pragma unroll
    for (int k = 0; k < K; ++k) act[k] = a.act[(int64_t)k * a.act_stride + (tcl < 0 ? 0 : tcl)];
    double v[K], iq[K], ssum = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        iq[k] = 1.0 / q[k];
        const double r = qmin * iq[k];
        double p = 1.0, b = r;            // r^D by squaring (D is a template parameter)
#pragma unroll
        for (int bit = D; bit > 0; bit >>= 1) {
            if (bit & 1) p = p * b;
            b = b * b;
        }
        v[k] = p * lane_bcast(cls, k);
        v[k] = (!a.masked || (valid && act[k])) ? v[k] : 0.0;
        ssum += v[k];
    }
    if (ssum < 1e-280) {
        double lp[K], mx = -INFINITY;
        ssum = 0.0;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            lp[k] = -(double)D * log(q[k]) - lane_bcast(ldl, k);
            mx = fmax(mx, lp[k]);
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            v[k] = exp(lp[k] - mx) * lane_bcast(pil, k);
            v[k] = (!a.masked || (valid && act[k])) ? v[k] : 0.0;
            ssum += v[k];
        }
    }
    const double is = 1.0 / fmax(ssum, GSS_TINY);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        double gam = v[k] * is;
        if (a.aff_eps != 0.0) gam = fmin(fmax(gam, a.aff_eps), 1.0 - a.aff_eps);
        if (MODE == MODE_PREDICT) {
            if (valid) a.gamma[((int64_t)f * K + k) * T + t] = gam;
        } else {
            // gamma / max(q, 10 tiny)
            if (valid)
                a.W[((int64_t)f * K + k) * T + t] =
                    gam * (q[k] < 10.0 * GSS_TINY ? 1.0 / (10.0 * GSS_TINY) : iq[k]);
            const double tot = wave_sum(valid ? gam : 0.0);
            if (lane == 0) a.Sg[(((int64_t)f * ntile + tile) * wpb + wave) * K + k] = tot;
        }
    }
}

// ------------------------------------------------------------------ M-step
// Partial sums over a chunk of frames of  w_k(t) y y^H  for KW weight rows, upper
// triangle in 2 x 2 register blocks; thread = (block, frame group).  Shared by the
// CACGMM M-step (unit-normalised y, K class weights) and by the PSD matrices of the
// beamformer (raw y, target / distortion masks).
struct WcovLds {
    int Dp, nblk, nfg, red_groups;   // red_groups: frame groups reduced per round
    size_t ys, wk, scratch, blk, total;
};

__host__ __device__ inline WcovLds wcov_lds_layout(int D, int KW) {
    WcovLds L;
    L.Dp = D + (D & 1);
    const int nb2 = L.Dp / 2;
    L.nblk = nb2 * (nb2 + 1) / 2;
    int nfg = 256 / L.nblk;
    if (nfg > EM_TILE) nfg = EM_TILE;
    if (nfg < 1) nfg = 1;
    L.nfg = nfg;
    size_t off = 0;
    const size_t ys_bytes = sizeof(cplx) * (size_t)L.Dp * EM_TS;
    // the final reduction over the frame groups reuses the tile; with few channels there
    // are up to 64 groups, which go through LDS many at a time (up to 32 KB)
    const size_t group_bytes = sizeof(cplx) * (size_t)L.nblk * 4 * KW;
    int rg = (int)((32 * 1024) / group_bytes);
    if (rg > nfg - 1) rg = nfg - 1;
    if (rg < 1) rg = 1;
    L.red_groups = rg;
    const size_t red_bytes = group_bytes * rg;
    L.ys = off;
    off += ys_bytes > red_bytes ? ys_bytes : red_bytes;
    L.wk = off;      off += sizeof(double) * KW * EM_TILE;
    L.scratch = off; off += sizeof(double) * 4 * EM_TILE;
    L.blk = off;     off += 2 * (size_t)L.nblk;
    L.total = (off + 15) / 16 * 16;
    return L;
}

#ifdef GSS_WCOV_TRACE
// tools/wcov_trace.py: per workgroup (wave 0): start, end (100 MHz clock) and shader cycles in
// staging (loads + LDS stores + barriers), accumulation, final reduction + store
__device__ long long g_wcov_phase[8192 * 6];
extern "C" int gss_debug_wcov_phase(long long *host, int entries) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wcov_phase), sizeof(long long) * 6 * entries);
}
#define WCOV_T(var) const long long var = __builtin_readcyclecounter()
#else
#define WCOV_T(var)
#endif

// KW weight rows per launch out of the Ktot rows of W / part, starting at row k0 (the
// M-step of more than 8 classes runs in groups).
// YP: elements of the (F, D, T) tile a thread stages = ceil(D / 4) -- 8 covers every channel
// count, 3 the D <= 12 of the prefetching form (the staging pass of a tile is instruction
// bound: tools/wcov_trace.py at D = 12, T = 7504 gives it 1600 of a tile's 4600 cycles, and the
// ISA shows why -- eight predicated loads and eight predicated stores, a branch around each,
// for three live elements).
// YEXACT: D == 4 YP (12, 20, 24 channels): every staged element is a live channel, no predicate.
template <int KW, bool NORMALISE, bool SRC_FDT, bool PREFETCH = false, int YP = 8, bool YEXACT = false>
__global__ __launch_bounds__(256) void wcov_kernel(const cplx *__restrict__ Y,
                                                   const double *__restrict__ W, int F,
                                                   int64_t T, int D, int NE, int nch,
                                                   int chunk_frames, cplx *__restrict__ part,
                                                   int Ktot, int k0, MsegPlan plan) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const WcovLds L = wcov_lds_layout(D, KW);
    const int Dp = L.Dp;
    cplx *ys = reinterpret_cast<cplx *>(smem + L.ys);
    double *wk = reinterpret_cast<double *>(smem + L.wk);
    double *scratch = reinterpret_cast<double *>(smem + L.scratch);
    unsigned char *blk = reinterpret_cast<unsigned char *>(smem + L.blk);

    // chunked form: ONE segment (f, chunk); static partition: the run of items of this slot,
    // one segment per frequency it touches
    int f = 0, chunk = 0, item = 0, item_end = 1;
    if (plan.S > 0) {
        item = mseg_begin(plan, blockIdx.x);
        item_end = mseg_begin(plan, blockIdx.x + 1);
    }
    const int item_begin = item;
    if (plan.S > 0) {
    } else if (!xcd_group_map(nch, F, f, chunk)) {
        return;
    }
    const int tid = threadIdx.x;
    const int tl = tid & 63, g = tid >> 6;
#ifdef GSS_WCOV_TRACE
    long long tr_stage = 0, tr_acc = 0;
    const long long tr_wall0 = __builtin_amdgcn_s_memrealtime();
#endif
    WCOV_T(tr_c0);
#ifdef GSS_WCOV_TRACE
    long long tr_red = 0;
#endif

    {
        const int nb2 = Dp / 2;
        for (int bi = tid; bi < nb2; bi += blockDim.x) {
            int idx = bi * nb2 - (bi * (bi - 1)) / 2;
            for (int bj = bi; bj < nb2; ++bj, ++idx) {
                blk[2 * idx] = (unsigned char)bi;
                blk[2 * idx + 1] = (unsigned char)bj;
            }
        }
    }
    const int mb = tid % L.nblk, fg = tid / L.nblk;
    const bool m_active = fg < L.nfg;
  for (bool first_seg = true; item < item_end; first_seg = false) {
    if (Dp != D) {
        // (the padding row shares LDS with the reduction of the previous segment)
        if (!first_seg) __syncthreads();
        for (int j = tid; j < EM_TS; j += blockDim.x) ys[D * EM_TS + j] = c_make(0.0, 0.0);
    }
    int64_t c0, c1;
    int nch_f = nch;
    if (plan.S > 0) {
        f = item / plan.ntile;
        const int tile0 = item - f * plan.ntile;
        const int ntl = min(plan.ntile - tile0, item_end - item);
        c0 = (int64_t)tile0 * EM_TILE;
        c1 = min(T, (int64_t)(tile0 + ntl) * EM_TILE);
        chunk = (int)blockIdx.x - mseg_slot_of(plan, f * plan.ntile);
        nch_f = plan.maxseg;
        item += ntl;
    } else {
        c0 = (int64_t)chunk * chunk_frames;
        c1 = c0 + chunk_frames < T ? c0 + chunk_frames : T;
        item = item_end;
    }
    const cplx *Yf = Y + (int64_t)f * T * D;
    const double *Wf = W + ((int64_t)f * Ktot + k0) * T;
    cplx acc[4][KW];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int k = 0; k < KW; ++k) acc[s][k] = c_make(0.0, 0.0);

    // PREFETCH (few channels, where a workgroup has next to no arithmetic per tile and would
    // sit at the barriers waiting for its loads): the NEXT tile is fetched into registers
    // before the accumulation over the current one and written to LDS after it.
    constexpr int WPRE = (KW * EM_TILE + 255) / 256;
    cplx ypre[YP];
    double wpre[WPRE];
    auto prefetch = [&](int64_t t0) {
        const cplx *src = Y + (int64_t)f * D * T;
#pragma unroll
        for (int j = 0; j < YP; ++j) {
            const int d = g + 4 * j;
            ypre[j] = ((YEXACT || d < D) && t0 + tl < c1) ? src[(int64_t)d * T + t0 + tl] : c_make(0.0, 0.0);
        }
#pragma unroll
        for (int j = 0; j < WPRE; ++j) {
            const int idx = tid + 256 * j;
            const int k = idx / EM_TILE, jj = idx - k * EM_TILE;
            wpre[j] = (idx < KW * EM_TILE && t0 + jj < c1) ? Wf[(int64_t)k * T + t0 + jj] : 0.0;
        }
    };
    if (SRC_FDT && PREFETCH) prefetch(c0);
    for (int64_t t0 = c0; t0 < c1; t0 += EM_TILE) {
        WCOV_T(tr_a);
        if (plan.S > 0) {
            // Static partition: the workgroups of a CU must END together (nothing refills a
            // free slot), but the issue arbiter serves the oldest first -- they finished one
            // after the other and the last third of the work ran with one wave per SIMD
            // (tools/wcov_trace.py).  Priority falls with progress: whoever is ahead waits.
            set_progress_priority(item - (int)((c1 - t0 + EM_TILE - 1) / EM_TILE) - item_begin,
                                  item_end - item_begin);
        }
        __syncthreads();
        if (SRC_FDT) {
            // Y is already the (F, D, T) unit-normalised copy: rows are contiguous.  Without
            // PREFETCH the tile is requested here -- all loads of a thread in flight before
            // its first LDS store (one load per trip of a loop is a chain of round trips).
            if (!PREFETCH) prefetch(t0);
#pragma unroll
            for (int j = 0; j < YP; ++j) {
                const int d = g + 4 * j;
                if (YEXACT || d < D) ys[d * EM_TS + tl] = ypre[j];
            }
#pragma unroll
            for (int j = 0; j < WPRE; ++j) {
                const int idx = tid + 256 * j;
                if (idx < KW * EM_TILE) wk[idx] = wpre[j];
            }
            __syncthreads();
            if (PREFETCH && t0 + EM_TILE < c1) prefetch(t0 + EM_TILE);
        } else {
            load_tile<NORMALISE, YP>(Yf, D, t0, c1, tl, g, ys, scratch);
            for (int idx = tid; idx < KW * EM_TILE; idx += blockDim.x) {
                const int k = idx / EM_TILE, j = idx - k * EM_TILE;
                wk[idx] = t0 + j < c1 ? Wf[(int64_t)k * T + t0 + j] : 0.0;
            }
            __syncthreads();
        }
        WCOV_T(tr_b);
        if (m_active) {
            const int nfr = (int)min((int64_t)EM_TILE, c1 - t0);
            const int bi = blk[2 * mb], bj = blk[2 * mb + 1];
            const cplx *ra0 = ys + (2 * bi) * EM_TS, *ra1 = ra0 + EM_TS;
            const cplx *rb0 = ys + (2 * bj) * EM_TS, *rb1 = rb0 + EM_TS;
            // (Requesting the next frame's four channel values into the same registers right
            // after the eight products, in front of the 40 multiply-adds, keeps the register
            // count and hides the LDS round trip -- measured in round 5: +-0 at D = 24 / 8,
            // +7 % at D = 12.  The loop is at its issue bound, not at the LDS latency.)
            for (int j = fg; j < nfr; j += L.nfg) {
                const cplx a0 = ra0[j], a1 = ra1[j], b0 = rb0[j], b1 = rb1[j];
                double pr[4], pim[4];
                pr[0] = a0.x * b0.x + a0.y * b0.y;  pim[0] = a0.y * b0.x - a0.x * b0.y;
                pr[1] = a0.x * b1.x + a0.y * b1.y;  pim[1] = a0.y * b1.x - a0.x * b1.y;
                pr[2] = a1.x * b0.x + a1.y * b0.y;  pim[2] = a1.y * b0.x - a1.x * b0.y;
                pr[3] = a1.x * b1.x + a1.y * b1.y;  pim[3] = a1.y * b1.x - a1.x * b1.y;
#pragma unroll
                for (int k = 0; k < KW; ++k) {
                    const double w = wk[k * EM_TILE + j];
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        acc[s][k].x = fma(w, pr[s], acc[s][k].x);
                        acc[s][k].y = fma(w, pim[s], acc[s][k].y);
                    }
                }
            }
        }
#ifdef GSS_WCOV_TRACE
        {
            WCOV_T(tr_c);
            tr_stage += tr_b - tr_a;
            tr_acc += tr_c - tr_b;
        }
#endif
    }
    WCOV_T(tr_c1);
    // reduce the frame groups (groups fg > 0 -> LDS -> group 0, red_groups per round, always
    // added in ascending group order), then store
    cplx *red = reinterpret_cast<cplx *>(smem + L.ys);
    const int gstride = L.nblk * 4 * KW;
    for (int r0 = 1; r0 < L.nfg; r0 += L.red_groups) {
        __syncthreads();
        if (m_active && fg >= r0 && fg < r0 + L.red_groups) {
            cplx *dst = red + (size_t)(fg - r0) * gstride;
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int k = 0; k < KW; ++k) dst[(mb * 4 + s) * KW + k] = acc[s][k];
        }
        __syncthreads();
        if (fg == 0) {
            for (int r = r0; r < r0 + L.red_groups && r < L.nfg; ++r) {
                const cplx *src = red + (size_t)(r - r0) * gstride;
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int k = 0; k < KW; ++k)
                        acc[s][k] = c_add(acc[s][k], src[(mb * 4 + s) * KW + k]);
            }
        }
    }
    if (fg == 0) {
        cplx *pp = part + (((int64_t)f * nch_f + chunk) * Ktot + k0) * NE;
        const int bi = blk[2 * mb], bj = blk[2 * mb + 1];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int d1 = 2 * bi + (s >> 1), d2 = 2 * bj + (s & 1);
            if (d1 <= d2 && d2 < D) {
                const int e = tri_index(d1, d2, D);
#pragma unroll
                for (int k = 0; k < KW; ++k) pp[k * NE + e] = acc[s][k];
            }
        }
    }
#ifdef GSS_WCOV_TRACE
    {
        WCOV_T(tr_c2s);
        tr_red += tr_c2s - tr_c1;
    }
#endif
  }   // segments
#ifdef GSS_WCOV_TRACE
    {
        WCOV_T(tr_c2);
        if (tid == 0 && blockIdx.x < 8192) {
            long long *o = g_wcov_phase + 6 * blockIdx.x;
            o[0] = tr_wall0;
            o[1] = __builtin_amdgcn_s_memrealtime();
            o[2] = tr_stage;
            o[3] = tr_acc;
            o[4] = tr_red;
            o[5] = tr_c2 - tr_c0;
        }
    }
#endif
}

// ------------------------------------------------------------------ M-step, register form
// One reference array (D = 4): the packed triangle of all K classes fits the registers of
// one lane (4 real diagonals + 6 complex off-diagonals = 16 doubles per class), so a lane
// owns frames, a wave owns a chunk, and nothing goes through LDS: coalesced loads from the
// (F, D, T) copy and the (F, K, T) weights, the next frame requested before the current
// one is accumulated, one cross-lane sum per entry at the end (on the DPP network: 80 sums
// through ds_bpermute butterflies were a third of the kernel, 0.052 -> 0.036 ms).  The tiled kernel above
// would spend its time at the barriers of 64-frame tiles that carry 3 blocks of work.
template <int K, int D>
__global__ __launch_bounds__(64) void mstep_reg_kernel(const cplx *__restrict__ Yn,
                                                       const double *__restrict__ W, int F,
                                                       int64_t T, int nch, int chunk_frames,
                                                       cplx *__restrict__ part) {
    constexpr int NE = D * (D + 1) / 2;
    int f, chunk;
    if (!xcd_group_map(nch, F, f, chunk)) return;
    const int lane = threadIdx.x;
    const int64_t c0 = (int64_t)chunk * chunk_frames;
    const int64_t c1 = c0 + chunk_frames < T ? c0 + chunk_frames : T;
    const cplx *yf = Yn + (int64_t)f * D * T;
    const double *wf = W + (int64_t)f * K * T;

    double are[K][NE], aim[K][NE];          // aim of the diagonal entries stays unused
#pragma unroll
    for (int k = 0; k < K; ++k)
#pragma unroll
        for (int e = 0; e < NE; ++e) are[k][e] = aim[k][e] = 0.0;

    cplx y[D], yn[D];
    double w[K], wn[K];
    auto fetch = [&](int64_t t, cplx (&yy)[D], double (&ww)[K]) {
        const bool ok = t < c1;
        const int64_t tc = ok ? t : c1 - 1;
#pragma unroll
        for (int d = 0; d < D; ++d) yy[d] = yf[(int64_t)d * T + tc];
#pragma unroll
        for (int k = 0; k < K; ++k) ww[k] = ok ? wf[(int64_t)k * T + tc] : 0.0;
    };
    fetch(c0 + lane, yn, wn);
    for (int64_t tb = c0; tb < c1; tb += 64) {                      // wave uniform
#pragma unroll
        for (int d = 0; d < D; ++d) y[d] = yn[d];
#pragma unroll
        for (int k = 0; k < K; ++k) w[k] = wn[k];
        if (tb + 64 < c1) fetch(tb + 64 + lane, yn, wn);
        int e = 0;
#pragma unroll
        for (int d1 = 0; d1 < D; ++d1) {
#pragma unroll
            for (int d2 = d1; d2 < D; ++d2, ++e) {
                const double pr = y[d1].x * y[d2].x + y[d1].y * y[d2].y;
#pragma unroll
                for (int k = 0; k < K; ++k) are[k][e] = fma(w[k], pr, are[k][e]);
                if (d2 != d1) {
                    const double pim = y[d1].y * y[d2].x - y[d1].x * y[d2].y;
#pragma unroll
                    for (int k = 0; k < K; ++k) aim[k][e] = fma(w[k], pim, aim[k][e]);
                }
            }
        }
    }
    cplx *pp = part + ((int64_t)f * nch + chunk) * K * NE;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        int e = 0;
#pragma unroll
        for (int d1 = 0; d1 < D; ++d1) {
#pragma unroll
            for (int d2 = d1; d2 < D; ++d2, ++e) {
                const double re = wave_sum(are[k][e]);
                const double im = d2 != d1 ? wave_sum(aim[k][e]) : 0.0;
                if (lane == 0) pp[k * NE + e] = c_make(re, im);
            }
        }
    }
}

// ------------------------------------------------------------------ model update
// Packed upper triangle, row major: row d1 starts at s(d1) = d1 (2 D - d1 + 1) / 2.  Closed
// form (float square root, corrected by at most one either way; checked exhaustively for
// D <= 64) -- a search loop costs up to D trips and the class update unpacks 20 entries.
__device__ __forceinline__ void tri_unpack(int e, int D, int &d1, int &d2) {
    const float b = 2.0f * (float)D + 1.0f;
    int r = (int)((b - sqrtf(b * b - 8.0f * (float)e)) * 0.5f);
    r = max(0, min(r, D - 1));
    if (r * (2 * D - r + 1) / 2 > e) --r;
    else if ((r + 1) * (2 * D - r) / 2 <= e) ++r;
    d1 = r;
    d2 = r + e - r * (2 * D - r + 1) / 2;
}

// (d1, d2) of the packed entries e = lane + 64 s a lane owns, unpacked once per kernel: the
// class update walks them four times (trace, Frobenius norm, scatter to LDS, B^-1 product) and
// the closed form costs a float square root and two corrections each time -- a sixth of the
// instructions of em_chol, which is bound by instruction issue.
struct TriSlots {
    int d12[9];     // d1 << 8 | d2 for slot s (9 = COV_SLOTS below), -1 past the triangle
};
__device__ __forceinline__ TriSlots tri_slots(int D, int lane) {
    TriSlots t;
    const int NE = tri_count(D);
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        const int e = lane + 64 * s;
        int d1 = 0, d2 = 0;
        if (e < NE) tri_unpack(e, D, d1, d2);
        t.d12[s] = e < NE ? (d1 << 8 | d2) : -1;
    }
    return t;
}

// The same from a table (tri_table_kernel: packed[e] for the row-major packed order, then
// packed[NE + e'] for the column-major order e' = d2 (d2 + 1) / 2 + d1): em_chol is bound by
// instruction issue (6800 instructions per class matrix at D = 24, tools/chol_trace.py) and
// the closed form above was 800 of them.
__global__ void tri_table_kernel(int D, int *__restrict__ tab) {
    const int NE = tri_count(D);
    for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < NE; e += gridDim.x * blockDim.x) {
        int d1, d2;
        tri_unpack(e, D, d1, d2);
        tab[e] = d1 << 8 | d2;
        int c = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);     // column of e'
        while (c * (c + 1) / 2 > e) --c;
        while ((c + 1) * (c + 2) / 2 <= e) ++c;
        tab[NE + e] = (e - c * (c + 1) / 2) << 8 | c;
    }
}
__device__ __forceinline__ TriSlots tri_slots_tab(const int *__restrict__ tab, int NE, int lane) {
    TriSlots t;
#pragma unroll
    for (int s = 0; s < 9; ++s) {
        const int e = lane + 64 * s;
        t.d12[s] = e < NE ? tab[e] : -1;
    }
    return t;
}
// packed entries a wave needs for D <= 8 NR channels: ceil(D (D + 1) / 2 / 64)
__host__ __device__ constexpr int cov_slots_for(int NR) { return NR <= 1 ? 1 : NR == 2 ? 3 : NR == 3 ? 5 : 9; }

// sum over the E-step's partial sums of gamma_k: one load per lane and a fixed reduction
// tree (a serial loop is a chain of dependent L2 round trips, one per partial sum).
__device__ inline double sum_gamma(const double *__restrict__ Sg, int sg_nch, int K, int k, int f,
                                   int lane) {
    double s = 0.0;
    for (int c = lane; c < sg_nch; c += 64) s += Sg[((int64_t)f * sg_nch + c) * K + k];
    return wave_sum(s);
}

// B_k = D * (sum over chunks of the partial sums) / max(sum gamma, tiny): every lane
// reduces its packed entries e = lane, lane + 64, ... into `vals` (at most
// COV_SLOTS = ceil(528 / 64) of them).  All chunk loads of an entry are issued before
// they are summed -- the reduction is otherwise a chain of dependent L2 round trips.
constexpr int COV_SLOTS = 9;
static_assert(COV_SLOTS == sizeof(TriSlots::d12) / sizeof(int), "TriSlots");

// (nch: partial-sum records per frequency in the layout, cnt <= nch: how many of them hold sums)
template <int NS = COV_SLOTS>
__device__ inline double reduce_covariance(const cplx *__restrict__ Bp, int nch, int cnt, int D, int K,
                                           int k, int f, double den, cplx (&vals)[COV_SLOTS],
                                           int lane, const TriSlots &ts) {
    const int NE = tri_count(D);
    double tr = 0.0;
    // CB chunks of ALL slots are requested before any of them is added (slot after slot was
    // one memory round trip per slot: 30 % of em_chol's time at D = 24); every entry is still
    // summed in ascending chunk order
    constexpr int CB = NS <= 3 ? 8 : NS <= 5 ? 6 : 2;
#pragma unroll
    for (int s = 0; s < COV_SLOTS; ++s) vals[s] = c_make(0.0, 0.0);
    const cplx *src = Bp + ((int64_t)f * nch * K + k) * NE + lane;
    for (int c = 0; c < cnt; c += CB) {
        cplx t[NS][CB];
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < CB; ++j)
                t[s][j] = (lane + 64 * s < NE && c + j < cnt) ? src[(int64_t)(c + j) * K * NE + 64 * s]
                                                              : c_make(0.0, 0.0);
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int j = 0; j < CB; ++j) vals[s] = c_add(vals[s], t[s][j]);
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (lane + 64 * s < NE) {
            vals[s].x = ((double)D * vals[s].x) / den;
            vals[s].y = ((double)D * vals[s].y) / den;
        }
    }
    // trace: diagonal entries
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int p = ts.d12[s];
        if (p >= 0 && (p >> 8) == (p & 255)) {
            vals[s].y = 0.0;
            tr += vals[s].x;
        }
    }
    return wave_sum(tr);
}

// Scatter the reduced entries into LDS: lower triangle (conjugated), optionally the
// mirrored upper one, with `shift` subtracted from the diagonal.
template <int NS = COV_SLOTS>
__device__ inline void store_covariance(const cplx (&vals)[COV_SLOTS], int D, double shift,
                                        bool full, cplx *A, int ld, int lane, const TriSlots &ts) {
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int p = ts.d12[s];
        if (p >= 0) {
            const int d1 = p >> 8, d2 = p & 255;
            cplx v = vals[s];
            if (d1 == d2) v.x -= shift;
            A[d2 * ld + d1] = c_conj(v);
            if (full) A[d1 * ld + d2] = v;
        }
    }
}

// grid (K, F), block 64 (one wave per class matrix).
//
// The reference keeps (V, lambda) with lambda <- max(lambda / lambda_max, 1e-10) and
// evaluates q = y^H V diag(1/lambda) V^H y and ln det = sum ln lambda.  A per-class
// scale of lambda cancels between -D ln q and -ln det, and rescales the next
// covariance by a constant that the next normalisation removes again -- unless a frame sits
// on the clamp q = max(|q|, tiny) (an all-zero frame: digital silence), where q does not scale;
// frequencies with such frames take the eigendecomposition (em_prepare_kernel flags them).  So whenever
// NO eigenvalue is floored -- lambda_min > 1e-10 lambda_max -- B^-1 and ln det B from a
// Cholesky factorisation give the same posteriors.  The certificate comes out of the
// quantities computed anyway:  lambda_min >= 1 / ||B^-1||_F  and  lambda_max <= ||B||_F,
// so  ||B||_F ||B^-1||_F < 0.5e10  proves that nothing would be floored (conservative
// by at most a factor D).  Matrices that fail it, or whose factorisation breaks down,
// are flagged for em_eigh_kernel, which overwrites their rows.
//
// Factor and inverse come from ONE register-resident sweep (chol_inverse_sweep,
// 8 x 8 lane grid, NR x NR entries per lane, D <= 8 NR).
//
// One class matrix by one wave (the wave may be part of a larger workgroup): `vals` holds
// the packed upper triangle of B (entry e = lane + 64 s).  Writes Mq(:, k) and ln det and
// returns whether the no-floor certificate holds; A = D * (8 NR + 1) complex + D doubles
// of LDS owned by this wave.
#ifdef GSS_CHOL_TRACE
// tools/chol_trace.py: shader-clock stamps of em_chol_kernel's wave per phase
__device__ long long g_chol_phase[4096 * 10];
extern "C" int gss_debug_chol_phase(long long *host, int entries) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_chol_phase), sizeof(long long) * 10 * entries);
}
#define CHOL_STAMP(slot)                                                              \
    do {                                                                              \
        const int wg_ = blockIdx.y * gridDim.x + blockIdx.x;                          \
        if (threadIdx.x == 0 && wg_ < 4096) g_chol_phase[wg_ * 10 + (slot)] = clock64(); \
    } while (0)
#else
#define CHOL_STAMP(slot)
#endif

template <int NR>
__device__ __forceinline__ bool class_update_chol(const cplx (&vals)[COV_SLOTS], int D, int K,
                                                  double eig_floor, cplx *A, int lane,
                                                  cplx *__restrict__ Mq_fk,
                                                  double *__restrict__ logdet_fk,
                                                  const TriSlots &ts,
                                                  const int *__restrict__ tab_cm = nullptr) {
    constexpr int ld = 8 * NR + 1;
    constexpr int NS = cov_slots_for(NR);
    double *dinv = reinterpret_cast<double *>(A + D * ld);
    double nb2 = 0.0;   // ||B||_F^2 from the packed upper triangle
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int p = ts.d12[s];
        if (p >= 0)
            nb2 += ((p >> 8) == (p & 255) ? 1.0 : 2.0) * (vals[s].x * vals[s].x + vals[s].y * vals[s].y);
    }
    nb2 = wave_sum(nb2);
    CHOL_STAMP(4);
    store_covariance<NS>(vals, D, 0.0, true, A, ld, lane, ts);
    wave_sync();
    const int tx = lane & 7, ty = lane >> 3;
    cplx reg[NR][NR];
#pragma unroll
    for (int a = 0; a < NR; ++a)
#pragma unroll
        for (int b = 0; b < NR; ++b) {
            const int i = ty + 8 * a, kk = tx + 8 * b;
            reg[a][b] = (kk >= i && kk < D) ? A[i * ld + kk] : c_make(0.0, 0.0);
        }
    wave_sync();
    CHOL_STAMP(5);
    if (!chol_inverse_sweep<8, NR, true>(reg, D, A, ld, dinv, tx, ty)) return false;
    CHOL_STAMP(6);
#if GSS_CHOL_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    double ldv = 0.0;   // ln det B = 2 sum ln U_ii
    for (int i = lane; i < D; i += 64) ldv -= 2.0 * log(dinv[i]);
    ldv = wave_sum(ldv);
    CHOL_STAMP(7);
    // W = U^-H from the unscaled rows of the sweep: W[j][k] = A[j][k] dinv[j] (k < j),
    // W[j][j] = dinv[j].  B^-1 = W^H W :  (d1,d2) = sum_{j >= d2} conj(W[j][d1]) W[j][d2]
    //   = dinv[d2]^2 (d1 == d2 ? 1 : conj(A[d2][d1]))
    //     + sum_{j > d2} dinv[j]^2 conj(A[j][d1]) A[j][d2]
    // An entry costs D - 1 - d2 terms.  With the table, slot s of a lane is entry
    // lane + 64 s of the COLUMN-major packed order, which sorts the entries by that count: the
    // trips of a slot (the longest of its 64 lanes) add up to 45 at D = 24 instead of 115 in
    // row-major order -- same terms, same order per entry, a third of the instructions.
    double ni2 = 0.0;
    const int NE = tri_count(D);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int p = tab_cm ? (lane + 64 * s < NE ? tab_cm[lane + 64 * s] : -1) : ts.d12[s];
        if (p < 0) continue;
        const int d1 = p >> 8, d2 = p & 255;
        const int e = tab_cm ? tri_index(d1, d2, D) : lane + 64 * s;
        const double s2 = dinv[d2] * dinv[d2];
        cplx v = d1 == d2 ? c_make(s2, 0.0) : c_scale(c_conj(A[d2 * ld + d1]), s2);
        for (int j = d2 + 1; j < D; ++j) {
            const double sj = dinv[j] * dinv[j];
            const cplx p = c_scale(A[j * ld + d1], sj);
            c_cfma(v, p, A[j * ld + d2]);
        }
        if (d1 == d2) {
            v.y = 0.0;
            ni2 += v.x * v.x;
        } else {
            ni2 += 2.0 * (v.x * v.x + v.y * v.y);
            v.x *= 2.0;
            v.y *= 2.0;
        }
        Mq_fk[(int64_t)e * K] = v;
    }
    ni2 = wave_sum(ni2);
    CHOL_STAMP(8);
    if (lane == 0) *logdet_fk = ldv;
    const double bound = 0.5 / eig_floor;
    return isfinite(ni2) && nb2 * ni2 < bound * bound;
}

// Eigendecomposition path (exactly the reference): eigenvalues / max, floor, then
// B^-1 = V diag(1/lambda) V^H and ln det = sum ln lambda.  A, V = m x m complex each and
// lam = m doubles of LDS owned by this wave, m = D rounded up to even.
// `basis` (optional, m * m complex that persist between calls for the same class): on entry, if
// *basis_valid, the eigenvectors of this class's covariance at the previous call -- the sweep
// starts from B rotated into that basis (nearly diagonal one EM iteration later); on exit the
// eigenvectors found now.
__device__ inline void class_update_eigh(const cplx (&vals)[COV_SLOTS], int D, int K,
                                         double eig_floor, cplx *A, int lane,
                                         cplx *__restrict__ Mq_fk,
                                         double *__restrict__ logdet_fk, const TriSlots &ts,
                                         cplx *basis = nullptr, int *basis_valid = nullptr) {
    const int m = D + (D & 1);
    cplx *V = A + m * m;
    double *lam = reinterpret_cast<double *>(V + m * m);
    for (int idx = lane; idx < m * m; idx += 64) A[idx] = c_make(0.0, 0.0);
    wave_sync();
    store_covariance(vals, D, 0.0, true, A, m, lane, ts);
    wave_sync();
    const bool warm = basis != nullptr && *basis_valid != 0;
    if (warm) {
        // A <- W^H B W, upper triangle computed and mirrored (exactly Hermitian), V <- W
        cplx rot[4];                              // m * m <= 256 entries over 64 lanes
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int idx = lane + 64 * s, i = idx / m, j = idx - i * m;
            rot[s] = c_make(0.0, 0.0);
            if (idx < m * m && i <= j) {
                for (int a = 0; a < m; ++a) {
                    cplx t = c_make(0.0, 0.0);
                    for (int b = 0; b < m; ++b) c_fma(t, A[a * m + b], basis[b * m + j]);
                    c_cfma(rot[s], basis[a * m + i], t);
                }
                if (i == j) rot[s].y = 0.0;
            }
        }
        wave_sync();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int idx = lane + 64 * s, i = idx / m, j = idx - i * m;
            if (idx < m * m) {
                V[idx] = basis[idx];
                if (i <= j) {
                    A[i * m + j] = rot[s];
                    A[j * m + i] = c_conj(rot[s]);
                }
            }
        }
        wave_sync();
    }
    jacobi_eigh_wave(A, V, m, lane, 20, warm);
    if (basis != nullptr) {
        for (int idx = lane; idx < m * m; idx += 64) basis[idx] = V[idx];
        if (lane == 0) *basis_valid = 1;
    }
    double lmax = -INFINITY;
    for (int i = lane; i < D; i += 64) lmax = fmax(lmax, A[i * m + i].x);
    lmax = wave_max(lmax);
    double ldv = 0.0;
    for (int i = lane; i < D; i += 64) {
        double l = A[i * m + i].x / fmax(lmax, GSS_TINY);
        l = fmax(l, eig_floor);
        lam[i] = 1.0 / l;
        ldv += log(l);
    }
    ldv = wave_sum(ldv);
    wave_sync();
#pragma unroll
    for (int s = 0; s < COV_SLOTS; ++s) {
        const int p = ts.d12[s];
        if (p < 0) continue;
        const int e = lane + 64 * s, d1 = p >> 8, d2 = p & 255;
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
        Mq_fk[(int64_t)e * K] = v;
    }
    if (lane == 0) *logdet_fk = ldv;
}

template <int NR>
__global__ __launch_bounds__(64) void em_chol_kernel(const cplx *__restrict__ Bp,
                                                     const double *__restrict__ Sg, int nch,
                                                     int sg_nch, int D,
                                                     int K, int64_t T, double eig_floor,
                                                     int force_eigh, cplx *__restrict__ Mq,
                                                     double *__restrict__ logdet,
                                                     double *__restrict__ pi,
                                                     int *__restrict__ need_eigh,
                                                     const int *__restrict__ tri_tab,
                                                     MsegPlan plan,
                                                     const int *__restrict__ zero_tiles, int ntile) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NE = tri_count(D);
    cplx *A = reinterpret_cast<cplx *>(smem);                  // D * (8 NR + 1) + D doubles
    const int k = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
    // A frequency with all-zero frames takes the eigendecomposition for every class: on the
    // clamp q = max(|q|, tiny) a per-class scale of lambda does NOT cancel (see below), and the
    // reference's eigenvalues are normalised by their maximum.
    const bool clamped = frequency_has_zero_frames(zero_tiles, ntile, f, lane);
    const int cnt = plan.S > 0 ? mseg_count(plan, f) : nch;

    CHOL_STAMP(0);
#if GSS_CHOL_PRIO
    __builtin_amdgcn_s_setprio(3);
#endif
    const TriSlots ts = tri_slots_tab(tri_tab, NE, lane);
    const double sg = sum_gamma(Sg, sg_nch, K, k, f, lane);
    const double den = fmax(sg, GSS_TINY);
    if (lane == 0) pi[f * K + k] = sg / (double)T;
    CHOL_STAMP(1);
    CHOL_STAMP(2);
    cplx vals[COV_SLOTS];
    const double tr = reduce_covariance<cov_slots_for(NR)>(Bp, nch, cnt, D, K, k, f, den, vals, lane, ts);
    CHOL_STAMP(3);
#if GSS_CHOL_PRIO
    __builtin_amdgcn_s_setprio(2);
#endif
    bool fast = !force_eigh && !clamped && tr > 0.0 && isfinite(tr);
    if (fast)
        fast = class_update_chol<NR>(vals, D, K, eig_floor, A, lane, Mq + (int64_t)f * NE * K + k,
                                     logdet + f * K + k, ts, tri_tab + NE);
    if (lane == 0) need_eigh[f * K + k] = fast ? 0 : 1;
    CHOL_STAMP(9);
}

// Eigendecomposition path for the flagged matrices; overwrites their Mq / ln det.
__global__ __launch_bounds__(64) void em_eigh_kernel(const cplx *__restrict__ Bp,
                                                     const double *__restrict__ Sg, int nch,
                                                     int sg_nch, int D,
                                                     int K, double eig_floor,
                                                     const int *__restrict__ need_eigh,
                                                     cplx *__restrict__ Mq,
                                                     double *__restrict__ logdet,
                                                     const int *__restrict__ tri_tab,
                                                     MsegPlan plan) {
    const int k = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
    if (!need_eigh[f * K + k]) return;
    const int cnt = plan.S > 0 ? mseg_count(plan, f) : nch;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int NE = tri_count(D);
    const double sg = sum_gamma(Sg, sg_nch, K, k, f, lane);
    const double den = fmax(sg, GSS_TINY);
    const TriSlots ts = tri_slots_tab(tri_tab, NE, lane);
    cplx vals[COV_SLOTS];
    reduce_covariance(Bp, nch, cnt, D, K, k, f, den, vals, lane, ts);
    class_update_eigh(vals, D, K, eig_floor, reinterpret_cast<cplx *>(smem), lane,
                      Mq + (int64_t)f * NE * K + k, logdet + f * K + k, ts);
}

// ------------------------------------------------------------------ one array: the whole EM in one launch
// With one array (D = 4: the reference's default multiarray=False, BASELINE config 1) an EM
// iteration is 71 MB of traffic and a few hundred thousand FMAs -- microseconds of work -- but
// cost three dependent launches of ~25-40 us each (E-step, M-step, model update): ramp-up, two
// memory round trips and a tail per launch, 60 of them per utterance.  Frequencies are
// independent and a frequency's model is 50 complex numbers, so here ONE workgroup owns a
// frequency for all iterations and nothing but the posteriors of the final predict step ever
// leaves the chip: the model lives in LDS, the unit-normalised observation (139 KB per
// frequency at T = 2169, L2 resident) is streamed once per iteration.  Per 256-frame chunk:
//   phase E  lane = frame: quadratic forms against the model in LDS (broadcast reads), softmax
//            -- the arithmetic of em_estep_reg_kernel --, the M-step weights w_kt and the 16
//            real numbers of the frame's Hermitian products P_de into LDS (the (F, K, T) weight
//            tensor does not exist);
//   phase M  thread = (entry i, frame slice s): 16 entries x 16 slices, K accumulators each that
//            run over ALL chunks of the iteration; w and P are read back from LDS.
// After the last chunk a 16-lane DPP row sum finishes the K x 16 sums (no partial buffers,
// no tickets, no fences: one workgroup saw every frame), wave w updates classes w, w + 4
// (Cholesky + no-floor certificate, Jacobi eigh when flagged -- the code of em_chol / em_eigh)
// and the next iteration starts.  20 iterations + predict = 1 launch instead of 61.
__device__ __forceinline__ double row16_sum(double v) {      // total in lane 15 of each row
    double s = v + dpp_shifted<0x111, 0xf, 0xf>(v);      // row_shr:1
    s += dpp_shifted<0x112, 0xf, 0xf>(v);                // row_shr:2
    s += dpp_shifted<0x113, 0xf, 0xf>(v);                // row_shr:3
    s += dpp_shifted<0x114, 0xf, 0xe>(s);                // row_shr:4, banks 1-3
    s += dpp_shifted<0x118, 0xf, 0xc>(s);                // row_shr:8, banks 2-3
    return s;
}

constexpr int OC_FRAMES = 256;

#ifndef GSS_EM4_PRIO
#define GSS_EM4_PRIO 1
#endif
#ifdef GSS_EM4_TRACE
// tools/em4_trace.py: shader cycles wave 0 of every workgroup spends per phase
__device__ long long g_em4_phase[1024 * 6];
extern "C" int gss_debug_em4_phase(long long *host, int entries) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(g_em4_phase), sizeof(long long) * 6 * entries);
}
// (accumulated in registers, written once at the end: a global read-modify-write per stamp
// cost more than the phases it measured)
#define EM4_T(var) const long long var = __builtin_readcyclecounter()
#define EM4_ADD(slot, d) em4_acc[slot] += (d)
#else
#define EM4_T(var)
#define EM4_ADD(slot, d)
#endif

struct OnchipArgs {
    cplx *basis;            // (F, K, 16) eigenvectors of flagged classes, one EM iteration back
    const cplx *Yn;         // (F, 4, T) unit-normalised observation
    const uint8_t *act;     // (K, act_stride)
    int64_t act_stride, T;
    int F, iterations, iterations_post, force_eigh;
    const int *zero_tiles;  // (F, ntile) em_prepare_kernel's flags: frames on the clamp
    int ntile;
    int cold_eigh;          // GSS_VARIANT em4_cold_eigh: every eigendecomposition from the identity
    double eig_floor;
    double *gamma;          // (F, K, T)
};

// One class per LANE (D = 4): Cholesky factor B = U^H U, X = U^-1, B^-1 = X X^H, ln det and
// the no-floor certificate of class_update_chol, all in the registers of one lane -- the
// generic wave-cooperative sweep (8 x 8 lane grid, LDS row ring, one wave per class) took
// 13 - 20 % of the one-launch kernel for 4 x 4 matrices.  A single lane is a chain of
// dependent instructions at ~16 cycles each, so everything that is not the factorisation is
// kept out of it: the 16 divisions by sum gamma are done by 16 threads beforehand and
// ln det = ln (product of the pivots) is one logarithm (the pivots of the trace-D matrix lie
// in [1e-10 D, D] or the certificate fails: no under- or overflow).  `b`: B_k in phase-M order
// (4 diagonals, then re and im of the six upper entries in walk order); `m`: B_k^-1 in the
// same order, off-diagonals doubled (the coefficients of q = sum_i m_i P_i).
__device__ __forceinline__ bool class_update4_lane(const double (&b)[16], double eig_floor,
                                                   double (&m)[16], double &logdet) {
    constexpr int D = 4;
    cplx a[D][D];
    double nb2 = 0.0, tr = 0.0;
    {
        int off = 0;
#pragma unroll
        for (int d1 = 0; d1 < D; ++d1)
#pragma unroll
            for (int d2 = d1; d2 < D; ++d2) {
                if (d1 == d2) {
                    a[d1][d1] = c_make(b[d1], 0.0);
                    tr += b[d1];
                    nb2 += b[d1] * b[d1];
                } else {
                    a[d1][d2] = c_make(b[D + off], b[D + 6 + off]);
                    nb2 += 2.0 * (a[d1][d2].x * a[d1][d2].x + a[d1][d2].y * a[d1][d2].y);
                    ++off;
                }
            }
    }
    bool ok = tr > 0.0 && isfinite(tr);
    double di[D], piv = 1.0;
    cplx U[D][D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
        const double ajj = a[j][j].x;
        const bool good = ajj > 0.0 && isfinite(ajj);
        ok = ok && good;
        piv *= ajj;
        di[j] = good ? rsqrt(ajj) : 0.0;
#pragma unroll
        for (int k = j + 1; k < D; ++k) U[j][k] = c_scale(a[j][k], di[j]);
#pragma unroll
        for (int i = j + 1; i < D; ++i) {
            a[i][i].x = fma(-U[j][i].x, U[j][i].x, a[i][i].x);
            a[i][i].x = fma(-U[j][i].y, U[j][i].y, a[i][i].x);
#pragma unroll
            for (int k = i + 1; k < D; ++k) {           // a[i][k] -= conj(U[j][i]) U[j][k]
                a[i][k].x = fma(-U[j][i].x, U[j][k].x, a[i][k].x);
                a[i][k].x = fma(-U[j][i].y, U[j][k].y, a[i][k].x);
                a[i][k].y = fma(-U[j][i].x, U[j][k].y, a[i][k].y);
                a[i][k].y = fma(U[j][i].y, U[j][k].x, a[i][k].y);
            }
        }
    }
    // X = U^-1 (upper): X[i][i] = di[i], X[i][k] = -di[i] sum_{m = i+1..k} U[i][m] X[m][k]
    cplx X[D][D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        X[k][k] = c_make(di[k], 0.0);
#pragma unroll
        for (int i = k - 1; i >= 0; --i) {
            cplx s = c_make(0.0, 0.0);
#pragma unroll
            for (int mm = i + 1; mm <= k; ++mm) c_fma(s, U[i][mm], X[mm][k]);
            X[i][k] = c_make(-di[i] * s.x, -di[i] * s.y);
        }
    }
    logdet = ok ? log(piv) : 0.0;
    double ni2 = 0.0;
    int off = 0;
#pragma unroll
    for (int d1 = 0; d1 < D; ++d1)
#pragma unroll
        for (int d2 = d1; d2 < D; ++d2) {
            cplx v = c_make(0.0, 0.0);
#pragma unroll
            for (int j = d2; j < D; ++j) c_fmac(v, X[d1][j], X[d2][j]);
            if (d1 == d2) {
                ni2 += v.x * v.x;
                m[d1] = v.x;
            } else {
                ni2 += 2.0 * (v.x * v.x + v.y * v.y);
                m[D + off] = 2.0 * v.x;
                m[D + 6 + off] = 2.0 * v.y;
                ++off;
            }
        }
    const double bound = 0.5 / eig_floor;
    return ok && isfinite(ni2) && nb2 * ni2 < bound * bound;
}

// sum over the four lanes of a quad, in every lane (DPP quad_perm)
__device__ __forceinline__ double quad_sum(double v) {
    v += dpp_shifted<0xB1, 0xf, 0xf>(v);      // quad_perm:[1,0,3,2]
    v += dpp_shifted<0x4E, 0xf, 0xf>(v);      // quad_perm:[2,3,0,1]
    return v;
}

constexpr int OC_LD = 72;        // doubles per LDS row of a wave's 64 frames (see phase M)

template <int K>
__global__ __launch_bounds__(256, 3) void em_onchip4_kernel(OnchipArgs a) {
    constexpr int D = 4, NE = 10, NP = 16;           // NP: real numbers per frame's products
    // Every WAVE owns its 64 frames from phase E through phase M: K weight rows and NP product
    // rows of its own in LDS, so a chunk needs no workgroup barrier (two per chunk cost a
    // quarter of the kernel).  Row stride 72 doubles: the lane groups of a ds_read_b128 meet
    // four different entries `mi`, and 2 * 72 dwords = 9 * 16 puts them 16 banks apart.
    __shared__ __attribute__((aligned(16))) double ldsS[4][K + NP][OC_LD];
    __shared__ double logdetS[K], piS[K], cS[K], sgS[4][K], bS[K][NP];
    // the model: row i holds the K coefficients of product P_i in q_k = sum_i m_ik P_i (B_k^-1
    // in phase-M order, off-diagonals doubled); phase E reads a row with broadcast
    // ds_read_b128s, two rows ahead of the FMAs that use it
    constexpr int KP = (K + 1) / 2;
    __shared__ __attribute__((aligned(16))) double mR[NP][2 * KP];
    __shared__ __attribute__((aligned(16))) cplx MqS[NE * K];      // (flagged classes: em_eigh's layout)
    __shared__ int flagS;
    // flagged classes: the eigenvectors of the previous iteration's covariance, where the next
    // Jacobi sweep starts (a class that fails the certificate tends to fail it in EVERY
    // iteration -- a speaker with a handful of active frames --, and its workgroup was the one
    // the whole launch waited for: 7 - 8 cold sweeps per iteration, 2 - 3 from here)
    // They live in global memory (a.basis: L2 resident, touched by flagged classes only) and pass
    // through the wave's idle rows: 1.5 KB more of LDS per workgroup would cost K = 6 its third
    // workgroup per CU -- and with 513 frequencies on 256 CUs the third is the one that matters.
    __shared__ int basis_validS[K];
    // scratch of the flagged-class path (one wave per class: Cholesky sweep / Jacobi): the
    // rows above are idle during the model update
    constexpr int CH_LD = 9;
    constexpr int SCRATCH = (2 * 4 * 4 + D * CH_LD) * sizeof(cplx) + 64;
    static_assert(SCRATCH <= 80 * sizeof(cplx) && 96 * sizeof(cplx) <= sizeof(double) * (K + NP) * OC_LD,
                  "scratch and the warm-start basis alias a wave's rows");
    const int64_t T = a.T;
    const int f = (int)blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const cplx *yf = a.Yn + (int64_t)f * D * T;
    double (*wS)[OC_LD] = ldsS[wave];                  // K weight rows of this wave
    double (*pS)[OC_LD] = ldsS[wave] + K;              // NP product rows of this wave
    const int nsub = (int)((T + OC_FRAMES - 1) / OC_FRAMES);
    const int mi = lane >> 2, msl = lane & 3;         // phase M: entry number, frame slice
    const TriSlots ts = tri_slots(D, lane);
    if (tid < K) basis_validS[tid] = 0;         // (published by the barriers of the first update)
    // every class through the eigendecomposition: asked for, or the frequency holds all-zero
    // frames (on the clamp of q the scale of B_k does not cancel, see em_chol_kernel)
    const bool force_exact = a.force_eigh != 0 ||
                             frequency_has_zero_frames(a.zero_tiles, a.ntile, f, lane);

#ifdef GSS_EM4_TRACE
    long long em4_acc[6] = {0, 0, 0, 0, 0, 0};
#endif
    EM4_T(c_start);
    // fit(I iterations, masked) + fit(post - 1 iterations, unmasked) + predict
    const int n_fit = a.iterations + (a.iterations_post > 1 ? a.iterations_post - 1 : 0);
    for (int it = 0; it <= n_fit; ++it) {
        const bool first = it == 0 && a.iterations > 0;
        const bool predict = it == n_fit;
        const bool masked = predict ? a.iterations_post == 0 : it < a.iterations;
        const double aff_eps = predict ? 0.0 : 1e-10;
        // (the two workgroups of a CU end together: the older one ran 19 % ahead)
        if (GSS_EM4_PRIO) set_progress_priority(it, n_fit + 1);
        double acc[K], sg[K];
#pragma unroll
        for (int k = 0; k < K; ++k) acc[k] = sg[k] = 0.0;

        // the frame of chunk sub + 1 is requested before chunk sub is evaluated
        cplx yn[D];
        uint8_t an[K];
        auto fetch = [&](int sub) {
            const int64_t t = (int64_t)sub * OC_FRAMES + tid;
            const int64_t tc = t < T ? t : T - 1;
#pragma unroll
            for (int d = 0; d < D; ++d) yn[d] = yf[(int64_t)d * T + tc];
#pragma unroll
            for (int k = 0; k < K; ++k) an[k] = a.act[(int64_t)k * a.act_stride + tc];
        };
        fetch(0);
        for (int sub = 0; sub < nsub; ++sub) {
            // a wave whose 64 frames lie past the end has nothing to do (wave uniform)
            if ((int64_t)sub * OC_FRAMES + 64 * wave >= T) break;
            EM4_T(c_a);
            const int64_t t = (int64_t)sub * OC_FRAMES + tid;
            const bool valid = t < T;
            // ---- phase E
            cplx y[D];
#pragma unroll
            for (int d = 0; d < D; ++d) y[d] = yn[d];
            bool on[K];
#pragma unroll
            for (int k = 0; k < K; ++k) on[k] = valid && an[k] != 0;
            if (sub + 1 < nsub) fetch(sub + 1);
            // products of the frame: slots 0-3 |y_d|^2, then re and im of the 6 upper entries
            double pv[NP];
            {
                int off = 0;
#pragma unroll
                for (int d1 = 0; d1 < D; ++d1)
#pragma unroll
                    for (int d2 = d1; d2 < D; ++d2) {
                        const double pr = y[d1].x * y[d2].x + y[d1].y * y[d2].y;
                        if (d1 == d2) {
                            pv[d1] = pr;
                        } else {
                            pv[D + off] = pr;
                            pv[D + 6 + off] = y[d1].y * y[d2].x - y[d1].x * y[d2].y;
                            ++off;
                        }
                    }
            }
            double q[K], gam[K], wgt[K];
            if (first) {
                // GSS initialisation (core.py:156-160): where(act == 0, 1e-10, act) / sum_k; q = 1
                double ssum = 0.0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    q[k] = 1.0;
                    gam[k] = on[k] ? 1.0 : 1e-10;
                    ssum += gam[k];
                }
#pragma unroll
                for (int k = 0; k < K; ++k) wgt[k] = gam[k] = gam[k] / ssum;
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) q[k] = 0.0;
                {
                    // q_k = sum_i m_ik P_i: the K coefficients of row i + 2 are requested
                    // before the FMAs of row i (LDS returns in order: partial waits)
                    double2 mb[3][KP];
                    auto mrow = [&](int i, int slot) {
#pragma unroll
                        for (int r = 0; r < KP; ++r)
                            mb[slot][r] = *reinterpret_cast<const double2 *>(&mR[i][2 * r]);
                    };
                    mrow(0, 0);
                    mrow(1, 1);
#pragma unroll
                    for (int i = 0; i < NP; ++i) {
                        if (i + 2 < NP) mrow(i + 2, (i + 2) % 3);
#pragma unroll
                        for (int k = 0; k < K; ++k) {
                            q[k] = fma(k & 1 ? mb[i % 3][k / 2].y : mb[i % 3][k / 2].x, pv[i], q[k]);
                        }
                    }
                }
                double ssum = 0.0;
                // pi_k exp(-D ln q_k - ln det_k - max) with D = 4 and no logarithm: relative to
                // the class with the smallest q it is  (q_min / q_k)^4 * cS[k]  with
                // cS[k] = pi_k exp(ln det_min - ln det_k)  from the model update -- K
                // exponentials per iteration instead of 2 K transcendental functions per frame
                // (450 of the 765 VALU instructions of phase E; -27 % on the kernel).  The
                // common factor between the two forms cancels in the normalisation; it is
                // bounded below by (1e-10)^(2 D) = 1e-80 -- eigenvalue floor / no-floor
                // certificate --, so nothing underflows that the log form keeps.
                // One reciprocal per class serves the ratio and the M-step weight gamma / q,
                // one more the normalisation: K + 1 divisions per frame instead of 3 K.
                // -DGSS_EM4_LOG_SOFTMAX builds the log / exp form of em_estep_reg_kernel.
                double qmin = INFINITY, iq[K];
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    q[k] = fmax(fabs(q[k]), GSS_TINY);
                    qmin = fmin(qmin, q[k]);
                }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    iq[k] = 1.0 / q[k];
                    const double r = qmin * iq[k], r2 = r * r;
                    gam[k] = (r2 * r2) * cS[k];
                    if (masked) gam[k] *= on[k] ? 1.0 : 0.0;
                    ssum += gam[k];
                }
                const double is = 1.0 / fmax(ssum, GSS_TINY);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    gam[k] = gam[k] * is;
                    if (aff_eps != 0.0) gam[k] = fmin(fmax(gam[k], aff_eps), 1.0 - aff_eps);
                    // gamma / max(q, 10 tiny)
                    wgt[k] = gam[k] * (q[k] < 10.0 * GSS_TINY ? 1.0 / (10.0 * GSS_TINY) : iq[k]);
                }
            }
            if (predict) {
#pragma unroll
                for (int k = 0; k < K; ++k)
                    if (valid) a.gamma[((int64_t)f * K + k) * T + t] = gam[k];
                continue;
            }
            EM4_T(c_b);
            wave_sync();                      // this wave's phase M of the previous chunk has read its rows
#pragma unroll
            for (int k = 0; k < K; ++k) {
                wS[k][lane] = valid ? wgt[k] : 0.0;
                sg[k] += valid ? gam[k] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) pS[i][lane] = pv[i];
            wave_sync();
            EM4_T(c_d);
            // ---- phase M: lane (entry mi, slice msl) adds w_kt P_mi(t) for the frame pairs
            // t = 2 msl + 8 j of the wave's 64 frames (16-byte LDS reads: one of P, K of w --
            // a broadcast within the 16 lanes of a slice -- per 2 K FMAs; operands of step
            // j + 1 are requested before the FMAs of step j)
            double2 pb[2], wb[2][K];
            auto mload = [&](int j, int slot) {
                const int tt = 2 * msl + 8 * j;
                pb[slot] = *reinterpret_cast<const double2 *>(&pS[mi][tt]);
#pragma unroll
                for (int k = 0; k < K; ++k) wb[slot][k] = *reinterpret_cast<const double2 *>(&wS[k][tt]);
            };
            mload(0, 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (j + 1 < 8) mload(j + 1, (j + 1) & 1);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    acc[k] = fma(wb[j & 1][k].x, pb[j & 1].x, acc[k]);
                    acc[k] = fma(wb[j & 1][k].y, pb[j & 1].y, acc[k]);
                }
            }
#ifdef GSS_EM4_TRACE
            {
                EM4_T(c_e);
                EM4_ADD(0, c_b - c_a);     // phase E
                EM4_ADD(2, c_d - c_b);     // LDS stores
                EM4_ADD(3, c_e - c_d);     // phase M
            }
#endif
        }
        if (predict) break;
        EM4_T(c_f);

        // ---- the iteration's sums: B_k entries (K x 16 real numbers) and sum_t gamma_kt:
        // over the four slices of a wave on the DPP network, over the waves through LDS
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double tot = quad_sum(acc[k]);
            if (msl == 0) (&ldsS[wave][0][0])[k * NP + mi] = tot;     // (its own rows: phase M is done)
            const double g = wave_sum(sg[k]);
            if (lane == 0) sgS[wave][k] = g;
        }
        __syncthreads();
        static_assert(K * NP <= 2 * OC_LD, "the wave sums fit the first rows");
        // B_k = D sum / max(sum gamma, tiny): one entry per thread
        if (tid < K * NP) {
            const int k = tid / NP;
            const double sgk = (sgS[0][k] + sgS[1][k]) + (sgS[2][k] + sgS[3][k]);
            const double tot = ((&ldsS[0][0][0])[tid] + (&ldsS[1][0][0])[tid]) +
                               ((&ldsS[2][0][0])[tid] + (&ldsS[3][0][0])[tid]);
            bS[k][tid % NP] = ((double)D * tot) / fmax(sgk, GSS_TINY);
        }
        if (tid == 0) flagS = 0;
        __syncthreads();
        // ---- model update: lane k of wave 0 takes class k (class_update4_lane); classes whose
        // factorisation breaks down or that fail the no-floor certificate are redone by one
        // wave each with the eigendecomposition of em_eigh
        EM4_T(c_u0);
        if (wave == 0 && lane < K) {
            const int k = lane;
            const double sgk = (sgS[0][k] + sgS[1][k]) + (sgS[2][k] + sgS[3][k]);
            piS[k] = sgk / (double)T;
            double b[NP], m[NP], ld;
#pragma unroll
            for (int i = 0; i < NP; ++i) b[i] = bS[k][i];
            const bool fast = class_update4_lane(b, a.eig_floor, m, ld) && !force_exact;
            if (fast) {
#pragma unroll
                for (int i = 0; i < NP; ++i) mR[i][k] = m[i];
                logdetS[k] = ld;
            } else {
                atomicOr(&flagS, 1 << k);
            }
        }
#ifdef GSS_EM4_TRACE
        {
            EM4_T(c_u1);
            EM4_ADD(1, c_u1 - c_u0);       // class update of wave 0's lanes
        }
#endif
        __syncthreads();
        if (flagS != 0) {
            const int flags = flagS;
            for (int k = wave; k < K; k += 4) {
                if (!(flags >> k & 1)) continue;
                // packed upper triangle (row major): diagonals are entries 0, 4, 7, 9 (slots 0-3
                // of bS), the six others in walk order (re: slots 4-9, im: slots 10-15)
                cplx vals[COV_SLOTS];
#pragma unroll
                for (int s2 = 0; s2 < COV_SLOTS; ++s2) vals[s2] = c_make(0.0, 0.0);
                int d1 = 0, d2 = 0;
                if (lane < NE) {
                    const int p = ts.d12[0];
                    d1 = p >> 8, d2 = p & 255;
                    if (d1 == d2) {
                        vals[0] = c_make(bS[k][d1], 0.0);
                    } else {
                        const int off = lane - d1 - 1;          // entries before it minus diagonals
                        vals[0] = c_make(bS[k][D + off], bS[k][D + 6 + off]);
                    }
                }
                cplx *A = reinterpret_cast<cplx *>(&ldsS[wave][0][0]);
                cplx *basis = A + 80;                  // (behind A, V, lambda and the sweep's ring)
                cplx *basis_g = a.basis + ((int64_t)f * K + k) * 16;
                if (!a.cold_eigh && basis_validS[k] && lane < 16) basis[lane] = basis_g[lane];
                wave_sync();
                class_update_eigh(vals, D, K, a.eig_floor, A, lane, MqS + k, logdetS + k, ts,
                                  a.cold_eigh ? nullptr : basis, basis_validS + k);
                if (!a.cold_eigh && lane < 16) basis_g[lane] = basis[lane];
                wave_sync();
                if (lane < NE) {
                    const cplx v = MqS[lane * K + k];
                    if (d1 == d2) {
                        mR[d1][k] = v.x;
                    } else {
                        mR[D + lane - d1 - 1][k] = v.x;
                        mR[D + 6 + lane - d1 - 1][k] = v.y;
                    }
                }
            }
            __syncthreads();
        }
        if (tid < K) {
            double ldmin = INFINITY;
            for (int k = 0; k < K; ++k) ldmin = fmin(ldmin, logdetS[k]);
            cS[tid] = piS[tid] * exp(ldmin - logdetS[tid]);
        }
        __syncthreads();
#ifdef GSS_EM4_TRACE
        {
            EM4_T(c_g);
            EM4_ADD(4, c_g - c_f);         // sums + model update
        }
#endif
    }
#ifdef GSS_EM4_TRACE
    {
        EM4_T(c_end);
        EM4_ADD(5, c_end - c_start);
        if (tid == 0 && blockIdx.x < 1024)
            for (int i = 0; i < 6; ++i) g_em4_phase[blockIdx.x * 6 + i] = em4_acc[i];
    }
#endif
}

template <int K>
int launch_onchip4(gss_ctx *ctx, const OnchipArgs &a) {
    GSS_PROF(ctx, "em_onchip");
    hipLaunchKernelGGL(em_onchip4_kernel<K>, dim3(a.F), dim3(256), 0, ctx->stream, a);
    GSS_LAUNCH_CHECK(ctx, "em_onchip4_kernel");
    return GSS_OK;
}

size_t em_estep_lds(int D, int K) {
    size_t b = sizeof(cplx) * (size_t)D * EM_TS;
    b += sizeof(double) * ((size_t)4 * K * EM_TILE + 2 * (size_t)K * EM_TILE + 2 * K);
    return (b + 15) / 16 * 16;
}

int em_chunks(int F, int64_t T, int D, int *chunk_frames) {
    // enough workgroups to fill 256 CUs a few times over, whole tiles per chunk; every
    // chunk costs one set of partial covariances that the model update re-reads
    // (measured on config 2: 4 / 6 / 8 chunks -> 21.30 / 21.23 / 21.68 ms per utterance).
    // With few channels a tile carries little arithmetic and longer chunks win (M-step at
    // T = 2172, workgroup targets 1536 / 3072: D = 4 0.0295 / 0.0359 ms, D = 12 0.1075 /
    // 0.1228 ms; D = 20 and 24 are flat within 2 %).
    int64_t tiles = (T + EM_TILE - 1) / EM_TILE;
    const int forced = gss_variant("em_wgs", 0);
    const int target = forced > 0 ? forced : (D <= 12 ? 1536 : 3072);
    int64_t want = (target + F - 1) / F;
    if (want < 1) want = 1;
    int64_t tiles_per_chunk = (tiles + want - 1) / want;
    if (tiles_per_chunk < 1) tiles_per_chunk = 1;
    *chunk_frames = (int)(tiles_per_chunk * EM_TILE);
    return (int)((tiles + tiles_per_chunk - 1) / tiles_per_chunk);
}

template <typename Kern>
int raise_lds_limit(gss_ctx *ctx, Kern kern, size_t lds) {
    if (lds > 64 * 1024)
        GSS_HIP_CHECK(ctx, hipFuncSetAttribute(reinterpret_cast<const void *>(kern),
                                               hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)lds));
    return GSS_OK;
}

template <int K>
int launch_estep(gss_ctx *ctx, int mode, const EmArgs &a, const cplx *Mq, int F) {
    const size_t lds = em_estep_lds(a.D, K);
    dim3 grid(xcd_grid(a.nch, F)), block(256);
    if (mode == MODE_FIRST) {
        GSS_PROF(ctx, "em_estep");
        hipLaunchKernelGGL((em_estep_kernel<K, MODE_FIRST>), grid, block, 0, ctx->stream, a, Mq);
    } else if (mode == MODE_EM) {
        GSS_TRY(raise_lds_limit(ctx, em_estep_kernel<K, MODE_EM>, lds));
        GSS_PROF(ctx, "em_estep");
        hipLaunchKernelGGL((em_estep_kernel<K, MODE_EM>), grid, block, lds, ctx->stream, a, Mq);
    } else {
        GSS_TRY(raise_lds_limit(ctx, em_estep_kernel<K, MODE_PREDICT>, lds));
        GSS_PROF(ctx, "em_predict");
        hipLaunchKernelGGL((em_estep_kernel<K, MODE_PREDICT>), grid, block, lds, ctx->stream, a,
                           Mq);
    }
    GSS_LAUNCH_CHECK(ctx, "em_estep_kernel");
    return GSS_OK;
}

// Waves per workgroup of the register-form E-step: 1 or 4, whichever fills the rounds of
// resident workgroups better (16 waves per CU at ~118 VGPRs).  At T = 941, F = 513: 4 waves
// give 2052 workgroups on 1024 slots -- 2.004 rounds, i.e. a third round for 4 workgroups;
// single waves give 7695 on 4096 = 1.88 rounds.
int estep_waves_per_block(int F, int64_t T) {
    auto eff = [&](int wpb) {
        const double wgs = (double)((T + 64 * wpb - 1) / (64 * wpb)) * F;
        const double rounds = wgs / (256.0 * (16 / wpb));
        return rounds / std::ceil(rounds);
    };
    const int forced = gss_variant("estep_wpb", 0);
    if (forced == 1 || forced == 4) return forced;
    return eff(1) > eff(4) ? 1 : 4;
}

template <int K, int D>
int launch_estep_reg(gss_ctx *ctx, int mode, const EmArgs &a, const cplx *Mq, const cplx *Yn,
                     int F) {
    const int wpb = estep_waves_per_block(F, a.T);
    const dim3 grid(xcd_grid((int)((a.T + 64 * wpb - 1) / (64 * wpb)), F)), block(64 * wpb);
    if (mode == MODE_EM) {
        GSS_PROF(ctx, "em_estep");
        hipLaunchKernelGGL((em_estep_reg_kernel<K, D, MODE_EM>), grid, block, 0, ctx->stream, a,
                           Mq, Yn);
    } else {
        GSS_PROF(ctx, "em_predict");
        hipLaunchKernelGGL((em_estep_reg_kernel<K, D, MODE_PREDICT>), grid, block, 0, ctx->stream,
                           a, Mq, Yn);
    }
    GSS_LAUNCH_CHECK(ctx, "em_estep_reg_kernel");
    return GSS_OK;
}

// Channel / class counts with a register-form E-step.
// D = 4 x arrays (all microphones), 2 x arrays (outer microphones) or 4 (reference array)
// for the 5 and 6 array sessions of the corpus.
bool estep_reg_supported(int D, int K) {
    return (D == 24 || D == 20 || D == 12 || D == 10 || D == 4) && K >= 2 && K <= 6;
}

template <int D>
int launch_estep_reg_k(gss_ctx *ctx, int K, int mode, const EmArgs &a, const cplx *Mq,
                       const cplx *Yn, int F) {
    switch (K) {
        case 2: return launch_estep_reg<2, D>(ctx, mode, a, Mq, Yn, F);
        case 3: return launch_estep_reg<3, D>(ctx, mode, a, Mq, Yn, F);
        case 4: return launch_estep_reg<4, D>(ctx, mode, a, Mq, Yn, F);
        case 5: return launch_estep_reg<5, D>(ctx, mode, a, Mq, Yn, F);
        case 6: return launch_estep_reg<6, D>(ctx, mode, a, Mq, Yn, F);
    }
    return gss_fail(ctx, GSS_ERR_UNSUPPORTED, "cacgmm: K=%d", K);
}

// The tiled M-step kernel for (KW classes, D channels): with the next tile prefetched for
// D <= mstep_prefetch_d (12), and staging exactly ceil(D / 4) elements per thread where that is
// one of the corpus' channel counts.
static int mstep_prefetch_max_d();
using wcov_fn_t = void (*)(const cplx *, const double *, int, int64_t, int, int, int, int, cplx *, int,
                           int, MsegPlan);
template <int KW>
wcov_fn_t mstep_kernel(int D) {
    const bool generic = gss_variant_set("mstep_generic");     // 8 predicated elements per thread
    if (D <= mstep_prefetch_max_d()) {
        if (D == 12 && !generic) return wcov_kernel<KW, false, true, true, 3, true>;
        if (D <= 12 && !generic) return wcov_kernel<KW, false, true, true, 3>;
        return wcov_kernel<KW, false, true, true>;
    }
    if (D == 24 && !generic) return wcov_kernel<KW, false, true, false, 6, true>;
    if (D == 20 && !generic) return wcov_kernel<KW, false, true, false, 5, true>;
    return wcov_kernel<KW, false, true>;
}

// Resident workgroups of the M-step kernel on this device (static partition: one run of
// items each).
template <int KW>
int mstep_resident_slots(gss_ctx *ctx, int D, int *slots) {
    const size_t lds = wcov_lds_layout(D, KW).total;
    const void *fn = reinterpret_cast<const void *>(mstep_kernel<KW>(D));
    if (lds > 64 * 1024)
        GSS_HIP_CHECK(ctx, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0, cus = 0;
    GSS_HIP_CHECK(ctx, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fn, 256, lds));
    GSS_HIP_CHECK(ctx, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, ctx->device));
    *slots = std::max(per_cu, 1) * cus;
    return GSS_OK;
}
// (pf_max_d: up to here the tiled M-step prefetches across tiles and keeps the chunked form)
static int mstep_prefetch_max_d() {
    const int v = gss_variant("mstep_prefetch_d", 12);
    return v;
}

template <int KW>
int launch_mstep(gss_ctx *ctx, const EmArgs &a, const cplx *Yn, int F, int K, int k0) {
    const size_t lds = wcov_lds_layout(a.D, KW).total;
    GSS_PROF(ctx, "em_mstep");
    if (Yn && a.D == 4 && KW == K && K >= 2 && K <= 6 && !gss_variant_set("mstep_tiled")) {
        hipLaunchKernelGGL((mstep_reg_kernel<(KW >= 2 && KW <= 6 ? KW : 2), 4>),
                           dim3(xcd_grid(a.nch, F)), dim3(64), 0, ctx->stream, Yn, a.W, F, a.T,
                           a.nch, a.chunk_frames, a.Bp);
        GSS_LAUNCH_CHECK(ctx, "mstep_reg_kernel");
        return GSS_OK;
    }
    if (Yn) {
        // (cross-tile prefetch against loading the tile in place, ms per launch at T = 2169:
        // D = 10 0.113 / 0.120, D = 12 0.122 / 0.124, D = 20 0.219 / 0.207; D = 24, T = 941:
        // 0.130 / 0.118)
        const wcov_fn_t fn = mstep_kernel<KW>(a.D);
        GSS_TRY(raise_lds_limit(ctx, fn, lds));
        hipLaunchKernelGGL(fn, dim3(a.mseg.S > 0 ? a.mseg.S : xcd_grid(a.nch, F)), dim3(256), lds,
                           ctx->stream, Yn, a.W, F, a.T, a.D, a.NE, a.nch, a.chunk_frames, a.Bp, K, k0,
                           a.mseg);
        GSS_LAUNCH_CHECK(ctx, "wcov_kernel");
        return GSS_OK;
    }
    GSS_TRY(raise_lds_limit(ctx, wcov_kernel<KW, true, false>, lds));
    hipLaunchKernelGGL((wcov_kernel<KW, true, false>), dim3(xcd_grid(a.nch, F)), dim3(256), lds,
                       ctx->stream, a.Y, a.W, F, a.T, a.D, a.NE, a.nch, a.chunk_frames, a.Bp, K, k0,
                       MsegPlan{});
    GSS_LAUNCH_CHECK(ctx, "wcov_kernel");
    return GSS_OK;
}

#define GSS_K_CASE(N, CALL) case N: { constexpr int KK = N; return CALL; }
#define GSS_K_SWITCH8(K, CALL)                                             \
    switch (K) {                                                           \
        GSS_K_CASE(1, CALL) GSS_K_CASE(2, CALL) GSS_K_CASE(3, CALL) GSS_K_CASE(4, CALL)     \
        GSS_K_CASE(5, CALL) GSS_K_CASE(6, CALL) GSS_K_CASE(7, CALL) GSS_K_CASE(8, CALL)     \
    }
// pb_bss: assert K < 20 (CACGMMTrainer.fit)
#define GSS_K_SWITCH19(K, CALL)                                            \
    switch (K) {                                                           \
        GSS_K_CASE(1, CALL) GSS_K_CASE(2, CALL) GSS_K_CASE(3, CALL) GSS_K_CASE(4, CALL)     \
        GSS_K_CASE(5, CALL) GSS_K_CASE(6, CALL) GSS_K_CASE(7, CALL) GSS_K_CASE(8, CALL)     \
        GSS_K_CASE(9, CALL) GSS_K_CASE(10, CALL) GSS_K_CASE(11, CALL) GSS_K_CASE(12, CALL) \
        GSS_K_CASE(13, CALL) GSS_K_CASE(14, CALL) GSS_K_CASE(15, CALL) GSS_K_CASE(16, CALL) \
        GSS_K_CASE(17, CALL) GSS_K_CASE(18, CALL) GSS_K_CASE(19, CALL)                     \
    }

int launch_estep_k(gss_ctx *ctx, int K, int mode, const EmArgs &a, const cplx *Mq, int F) {
    GSS_K_SWITCH19(K, launch_estep<KK>(ctx, mode, a, Mq, F));
    return gss_fail(ctx, GSS_ERR_UNSUPPORTED, "cacgmm: K=%d", K);
}

int launch_mstep_group(gss_ctx *ctx, int KW, const EmArgs &a, const cplx *Yn, int F, int K,
                       int k0) {
    GSS_K_SWITCH8(KW, launch_mstep<KK>(ctx, a, Yn, F, K, k0));
    return gss_fail(ctx, GSS_ERR_UNSUPPORTED, "cacgmm: class group of %d", KW);
}

// The M-step keeps 4 complex accumulators per class and thread: up to 8 classes per launch;
// more classes (pb_bss allows K < 20: RTTM sessions with many speakers) go in groups of
// equal size, each a pass over the observation.
int launch_mstep_k(gss_ctx *ctx, int K, const EmArgs &a, const cplx *Yn, int F) {
    const int ngroups = (K + 7) / 8, per = (K + ngroups - 1) / ngroups;
    for (int k0 = 0; k0 < K; k0 += per)
        GSS_TRY(launch_mstep_group(ctx, std::min(per, K - k0), a, Yn, F, K, k0));
    return GSS_OK;
}

int mstep_resident_slots_k(gss_ctx *ctx, int KW, int D, int *slots) {
    GSS_K_SWITCH8(KW, mstep_resident_slots<KK>(ctx, D, slots));
    return gss_fail(ctx, GSS_ERR_UNSUPPORTED, "cacgmm: KW=%d", KW);
}

// (segments per frequency the static partition may cut: the partial-sum workspace holds as many)
static int mstep_maxseg_limit() { return std::min(std::max(gss_variant("mstep_maxseg", 8), 1), 32); }

// The M-step's static partition for (F, T) on this device; S = 0 where the chunked form stays
// (one array, GSS_MSTEP_CHUNKED=1, few frequencies).
int mstep_plan(gss_ctx *ctx, int F, int64_t T, int D, int K, MsegPlan *plan) {
    *plan = MsegPlan{};
    const int min_d = gss_variant("mstep_plan_min_d", 5);
    if (D < min_d || D <= 4 || gss_variant_set("mstep_chunked")) return GSS_OK;
    const int ngroups = (K + 7) / 8, per = (K + ngroups - 1) / ngroups;
    int slots = 0;
    GSS_TRY(mstep_resident_slots_k(ctx, per, D, &slots));
    if (gss_variant("mstep_slots", 0) > 0) slots = gss_variant("mstep_slots", 0);
    const int64_t ntile = (T + EM_TILE - 1) / EM_TILE, N = ntile * F;
    if (N >= (1LL << 30) || ntile < 1) return GSS_OK;
    plan->ntile = (int)ntile;
    plan->S = (int)std::min<int64_t>(slots, N);
    plan->base = (int)(N / plan->S);
    plan->rem = (int)(N % plan->S);
    for (int f = 0; f < F; ++f) plan->maxseg = std::max(plan->maxseg, mseg_count(*plan, f));
    // (few frequencies on a large chip: a run is a tile or two and a frequency would be cut
    // into more segments than chunks -- the chunked form stays)
    if (plan->maxseg > mstep_maxseg_limit()) *plan = MsegPlan{};
    return GSS_OK;
}


// ------------------------------------------------------------------ frequency blocks
// Long segments (the RTTM front end cuts 120 s - 400 s contexts, core_chime6_rttm.py:360-364):
// the unit-normalised observation of ALL frequencies no longer fits the 256 MB Infinity Cache
// (config 5: 739 MB + 154 MB of weights) and every E-step and M-step launch of every iteration
// streams it from HBM again.  Frequencies are independent, so the EM CAN run over blocks of
// frequencies whose observation + weights stay on die -- all iterations and the final predict
// of a block before the next block starts -- optionally two blocks at a time on two streams
// (one block's model update under the other's E-step / M-step).
//   em_l3_mb      budget in MB for what is in flight; 0 = one block            (default 0)
//   em_l3_fit_mb  up to this many MB of (Yn + W) the EM stays one block anyway  (default 230)
//   em_streams    blocks in flight (1 | 2)                                      (default 2)
// MEASURED in round 6 and NOT the default (EXPERIMENTS.md, round 6 item 1): at config 5 the
// resident blocks are slower per frequency than the launches that stream from HBM (E-step
// 0.53 vs 0.43 us per frequency, M-step 0.71 vs 0.59; 6 blocks 48.3 ms, 2 blocks 42.0 ms, one
// block 41.7 ms per utterance).  A read stream gets the same 6.3 - 6.7 TB/s from the Infinity
// Cache as from HBM (tools/micro/hbm_stream_bench.hip) and the E-step already runs at 0.85 of
// a traffic-only kernel with its access pattern; short launches only add ramp and tail.
// Per-frequency arithmetic is unchanged except for the grouping of the M-step's partial sums
// (the chunking / static partition depends on the number of frequencies of a launch).
struct EmBlockPlan {
    int fb, nblocks, streams;
};
EmBlockPlan em_block_plan(int F, int64_t T, int D, int K) {
    EmBlockPlan p{F, 1, 1};
    const double per_f = (16.0 * D + 8.0 * K) * (double)T;           // Yn + W per frequency
    const double mb = 1024.0 * 1024.0;
    const int fit = gss_variant("em_l3_fit_mb", 230), budget = gss_variant("em_l3_mb", 0);
    if (budget <= 0 || F <= 8 || per_f * F <= fit * mb) return p;
    p.streams = gss_variant("em_streams", 2) >= 2 ? 2 : 1;
    int fb = (int)(budget * mb / p.streams / per_f) / 8 * 8;
    if (fb < 8) fb = 8;
    int nb = (F + fb - 1) / fb;
    if (p.streams == 2 && (nb & 1)) ++nb;                           // pairs
    fb = ((F + nb - 1) / nb + 7) / 8 * 8;                           // balanced, XCD aligned
    nb = (F + fb - 1) / fb;
    if (nb <= 1) return EmBlockPlan{F, 1, 1};
    p.fb = fb;
    p.nblocks = nb;
    return p;
}
int em_block_size(const EmBlockPlan &p, int F, int b) { return std::min(p.fb, F - b * p.fb); }

// Per-frequency record counts of the partial-sum arrays (their strides): the largest over the
// block sizes of the plan.
struct EmStrides {
    int nch, bp_nch, reg_nch;
};
EmStrides em_strides(const EmBlockPlan &p, int F, int64_t T, int D) {
    EmStrides s{0, 0, 0};
    for (int b : {0, p.nblocks - 1}) {
        const int Fb = em_block_size(p, F, b);
        int cf;
        const int nch = em_chunks(Fb, T, D, &cf);
        const int wpb = estep_waves_per_block(Fb, T);
        s.nch = std::max(s.nch, nch);
        s.reg_nch = std::max(s.reg_nch, wpb * (int)((T + 64 * wpb - 1) / (64 * wpb)));
    }
    s.bp_nch = std::max(s.nch, mstep_maxseg_limit());
    return s;
}

}  // namespace

// PSD accumulation of the beamformer: same kernel, raw observations, two masks.
// W = (F, 2, T) [target, distortion]; part = (F, nch, 2, NE).
int psd_partials_run(gss_ctx *ctx, const cplx *Y, int F, int64_t T, int D, const double *W2,
                     int nch, int chunk_frames, cplx *part) {
    const size_t lds = wcov_lds_layout(D, 2).total;
    // (the tile pass of the (F, T, D) layout with exactly as many slots as the channel count needs)
    const wcov_fn_t fn = D <= 4 ? wcov_kernel<2, false, false, false, 1>
                         : D <= 12 ? wcov_kernel<2, false, false, false, 3>
                         : D <= 24 ? wcov_kernel<2, false, false, false, 6> : wcov_kernel<2, false, false>;
    GSS_TRY(raise_lds_limit(ctx, fn, lds));
    hipLaunchKernelGGL(fn, dim3(xcd_grid(nch, F)), dim3(256), lds,
                       ctx->stream, Y, W2, F, T, D, tri_count(D), nch, chunk_frames, part, 2, 0,
                       MsegPlan{});
    GSS_LAUNCH_CHECK(ctx, "wcov_kernel");
    return GSS_OK;
}

size_t cacgmm_workspace_bytes(int F, int64_t T, int D, int K) {
    const size_t NE = tri_count(D);
    const EmBlockPlan bp = em_block_plan(F, T, D, K);
    const EmStrides st = em_strides(bp, F, T, D);
    size_t b = 0;
    b += align_up(sizeof(cplx) * (size_t)F * NE * K);            // Mq
    b += 2 * align_up(sizeof(double) * (size_t)F * K);           // logdet, pi
    b += align_up(sizeof(double) * (size_t)F * K * T);           // W
    b += align_up(sizeof(cplx) * (size_t)F * st.bp_nch * K * NE);   // Bp (chunks, or segments)
    b += align_up(sizeof(double) * (size_t)F * st.nch * K);      // Sg
    b += align_up(sizeof(int) * (size_t)F * K);                  // need_eigh
    b += align_up(sizeof(int) * 2 * NE);                         // tri_tab
    b += align_up(sizeof(cplx) * (size_t)F * D * T);             // Yn (register-form E-step)
    b += align_up(sizeof(double) * (size_t)F * st.reg_nch * K);  // Sg of the register-form E-step
    b += align_up(sizeof(cplx) * (size_t)F * K * 16);            // em_onchip4: warm-start eigenvectors (D = 4)
    return b + 4096;
}

namespace {
// What the launches of one block of frequencies [f0, f0 + F) work on: every per-frequency array
// offset to the block's first frequency, strides from the block's own plan.
struct EmBlock {
    EmArgs a;
    int F;
    cplx *Mq, *Yn;
    int *need_eigh, *zero_tiles;
    double *Sg_lds, *Sg_reg;
    int nch_lds, reg_nch, bp_nch, sg_nch;
    hipStream_t stream;
};
struct StreamRestore {
    gss_ctx *ctx;
    hipStream_t s;
    ~StreamRestore() { ctx->stream = s; }
};
}   // namespace

int cacgmm_run(gss_ctx *ctx, const cplx *Y, int F, int64_t T, int D, const uint8_t *act,
               int64_t act_stride, int K, int iterations, int iterations_post, double *gamma) {
    const int NE = tri_count(D);
    const bool reg = estep_reg_supported(D, K) && !gss_variant_set("estep_lds");
    // one array: the whole EM (all iterations + predict) in one launch (em_onchip4_kernel)
    const bool onchip = D == 4 && K >= 2 && K <= 6 && reg && iterations > 0 &&
                        !gss_variant_set("em_unfused");
    const EmBlockPlan bplan = onchip ? EmBlockPlan{F, 1, 1} : em_block_plan(F, T, D, K);
    const EmStrides st = em_strides(bplan, F, T, D);

    cplx *Mq = arena_alloc_t<cplx>(ctx, (size_t)F * NE * K);
    double *logdet = arena_alloc_t<double>(ctx, (size_t)F * K);
    double *pi = arena_alloc_t<double>(ctx, (size_t)F * K);
    double *W = arena_alloc_t<double>(ctx, (size_t)F * K * T);
    cplx *Bp = arena_alloc_t<cplx>(ctx, (size_t)F * st.bp_nch * K * NE);
    double *Sg_lds = arena_alloc_t<double>(ctx, (size_t)F * st.nch * K);
    // the unit-normalised (F, D, T) copy feeds the register-form E-step and the M-step of every
    // channel count; the register-form E-step has its own (finer) partial sums of gamma
    cplx *Yn = arena_alloc_t<cplx>(ctx, (size_t)F * D * T);
    double *Sg_reg = arena_alloc_t<double>(ctx, (size_t)F * st.reg_nch * K);
    int *need_eigh = arena_alloc_t<int>(ctx, (size_t)F * K);
    // per (frequency, 64-frame tile): holds a frame whose normalised observation is zero
    const int ntile = (int)((T + EM_TILE - 1) / EM_TILE);
    int *zero_tiles = arena_alloc_t<int>(ctx, (size_t)F * ntile);
    // (d1, d2) of the packed triangle, row-major and column-major order (em_chol / em_eigh)
    int *tri_tab = arena_alloc_t<int>(ctx, 2 * (size_t)NE);
    GSS_REQUIRE(ctx, Mq && logdet && pi && W && Bp && Sg_lds && Yn && Sg_reg && need_eigh && tri_tab &&
                         zero_tiles,
                GSS_ERR_NOMEM, "cacgmm workspace");
    GSS_REQUIRE(ctx, em_estep_lds(D, K) <= 160 * 1024 &&
                         wcov_lds_layout(D, std::min(K, 8)).total <= 160 * 1024,
                GSS_ERR_UNSUPPORTED, "cacgmm: D=%d K=%d LDS", D, K);
    const int m = D + (D & 1);
    const size_t eigh_lds = (sizeof(cplx) * 2 * m * m + sizeof(double) * m + 15) / 16 * 16;
    const int force_eigh = gss_variant_set("force_eigh");
    if (!onchip) {
        hipLaunchKernelGGL(tri_table_kernel, dim3(1), dim3(256), 0, ctx->stream, D, tri_tab);
        GSS_LAUNCH_CHECK(ctx, "tri_table_kernel");
    }

    auto make_block = [&](int f0, int Fb, hipStream_t stream, EmBlock *blk) -> int {
        EmBlock &b = *blk;
        b = EmBlock{};
        b.F = Fb;
        b.stream = stream;
        EmArgs &a = b.a;
        a.Y = Y + (int64_t)f0 * T * D;
        a.act = act;
        a.act_stride = act_stride;
        a.T = T;
        a.F = Fb;
        a.D = D;
        a.NE = NE;
        a.nch = em_chunks(Fb, T, D, &a.chunk_frames);
        GSS_TRY(mstep_plan(ctx, Fb, T, D, K, &a.mseg));
        // partial-sum records per frequency: chunks, or segments of the static partition
        b.bp_nch = a.mseg.S > 0 ? a.mseg.maxseg : a.nch;
        b.nch_lds = a.nch;
        const int wpb = estep_waves_per_block(Fb, T);
        b.reg_nch = wpb * (int)((T + 64 * wpb - 1) / (64 * wpb));
        GSS_REQUIRE(ctx, b.bp_nch <= st.bp_nch && b.nch_lds <= st.nch && b.reg_nch <= st.reg_nch,
                    GSS_ERR_INVALID, "cacgmm: block strides");
        b.Mq = Mq + (int64_t)f0 * NE * K;
        a.logdet = logdet + (int64_t)f0 * K;
        a.pi = pi + (int64_t)f0 * K;
        a.W = W + (int64_t)f0 * K * T;
        a.Bp = Bp + (int64_t)f0 * st.bp_nch * K * NE;
        b.Sg_lds = Sg_lds + (int64_t)f0 * st.nch * K;
        b.Sg_reg = Sg_reg + (int64_t)f0 * st.reg_nch * K;
        a.Sg = b.Sg_lds;
        a.gamma = gamma + (int64_t)f0 * K * T;
        b.Yn = Yn + (int64_t)f0 * D * T;
        b.need_eigh = need_eigh + (int64_t)f0 * K;
        b.zero_tiles = zero_tiles + (int64_t)f0 * ntile;
        b.sg_nch = b.nch_lds;
        return GSS_OK;
    };
    auto prepare = [&](EmBlock &b) -> int {
        GSS_PROF(ctx, "em_prepare");
        const size_t plds = sizeof(cplx) * (size_t)D * EM_TS + sizeof(double) * 4 * EM_TILE;
        const auto prep = D <= 4 ? em_prepare_kernel<1> : D <= 12 ? em_prepare_kernel<3>
                          : D <= 24 ? em_prepare_kernel<6> : em_prepare_kernel<8>;
        hipLaunchKernelGGL(prep,
                           dim3(xcd_grid((int)((T + EM_TILE - 1) / EM_TILE), b.F)), dim3(256), plds,
                           ctx->stream, b.a.Y, b.F, T, D, b.Yn, b.zero_tiles);
        GSS_LAUNCH_CHECK(ctx, "em_prepare_kernel");
        return GSS_OK;
    };
    auto estep = [&](EmBlock &b, int mode) -> int {
        EmArgs &a = b.a;
        if (reg && mode != MODE_FIRST) {
            a.Sg = b.Sg_reg;
            switch (D) {
                case 24: return launch_estep_reg_k<24>(ctx, K, mode, a, b.Mq, b.Yn, b.F);
                case 20: return launch_estep_reg_k<20>(ctx, K, mode, a, b.Mq, b.Yn, b.F);
                case 12: return launch_estep_reg_k<12>(ctx, K, mode, a, b.Mq, b.Yn, b.F);
                case 10: return launch_estep_reg_k<10>(ctx, K, mode, a, b.Mq, b.Yn, b.F);
                default: return launch_estep_reg_k<4>(ctx, K, mode, a, b.Mq, b.Yn, b.F);
            }
        }
        a.Sg = b.Sg_lds;
        return launch_estep_k(ctx, K, mode, a, b.Mq, b.F);
    };
    auto eig = [&](EmBlock &b) -> int {
        EmArgs &a = b.a;
        {
            GSS_PROF(ctx, "em_chol");
            const int nr = (D + 7) / 8;
            const size_t lds = sizeof(cplx) * (size_t)D * (8 * nr + 1) + sizeof(double) * D;
            auto kern = nr <= 1 ? em_chol_kernel<1> : nr == 2 ? em_chol_kernel<2>
                        : nr == 3 ? em_chol_kernel<3> : em_chol_kernel<4>;
            hipLaunchKernelGGL(kern, dim3(K, b.F), dim3(64), lds, ctx->stream, a.Bp, a.Sg, b.bp_nch,
                               b.sg_nch, D, K, T, 1e-10, force_eigh, b.Mq,
                               const_cast<double *>(a.logdet), const_cast<double *>(a.pi),
                               b.need_eigh, tri_tab, a.mseg, b.zero_tiles, ntile);
            GSS_LAUNCH_CHECK(ctx, "em_chol_kernel");
        }
        {
            GSS_PROF(ctx, "em_eigh");
            hipLaunchKernelGGL(em_eigh_kernel, dim3(K, b.F), dim3(64), eigh_lds, ctx->stream, a.Bp,
                               a.Sg, b.bp_nch, b.sg_nch, D, K, 1e-10, b.need_eigh, b.Mq,
                               const_cast<double *>(a.logdet), tri_tab, a.mseg);
            GSS_LAUNCH_CHECK(ctx, "em_eigh_kernel");
        }
        return GSS_OK;
    };

    if (onchip) {
        EmBlock b;
        GSS_TRY(make_block(0, F, ctx->stream, &b));
        GSS_TRY(prepare(b));
        OnchipArgs o{};
        o.basis = arena_alloc_t<cplx>(ctx, (size_t)F * K * 16);
        GSS_REQUIRE(ctx, o.basis, GSS_ERR_NOMEM, "cacgmm workspace");
        o.Yn = Yn;
        o.act = act;
        o.act_stride = act_stride;
        o.T = T;
        o.F = F;
        o.iterations = iterations;
        o.iterations_post = iterations_post;
        o.force_eigh = force_eigh;
        o.zero_tiles = b.zero_tiles;
        o.ntile = ntile;
        o.cold_eigh = gss_variant_set("em4_cold_eigh");
        o.eig_floor = 1e-10;
        o.gamma = gamma;
        switch (K) {
            case 2: return launch_onchip4<2>(ctx, o);
            case 3: return launch_onchip4<3>(ctx, o);
            case 4: return launch_onchip4<4>(ctx, o);
            case 5: return launch_onchip4<5>(ctx, o);
            default: return launch_onchip4<6>(ctx, o);
        }
    }

    // CACGMMTrainer.fit(initialization=array, iterations=I, source_activity_mask)
    auto em_iteration = [&](EmBlock &b, bool first) -> int {
        GSS_TRY(estep(b, first ? MODE_FIRST : MODE_EM));
        b.sg_nch = (reg && !first) ? b.reg_nch : b.nch_lds;
        GSS_TRY(launch_mstep_k(ctx, K, b.a, b.Yn, b.F));
        return eig(b);
    };

    // The blocks of a group (1 or 2) advance together, launch by launch, each on its own stream.
    hipStream_t const main_stream = ctx->stream;
    StreamRestore restore{ctx, main_stream};
    const int group = bplan.streams;
    if (group > 1) {
        GSS_TRY(aux_stream_ready(ctx));
        GSS_HIP_CHECK(ctx, hipEventRecord(ctx->ev_fork, main_stream));
        GSS_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
    }
    for (int b0 = 0; b0 < bplan.nblocks; b0 += group) {
        EmBlock blk[2];
        const int nb = std::min(group, bplan.nblocks - b0);
        for (int j = 0; j < nb; ++j)
            GSS_TRY(make_block((b0 + j) * bplan.fb, em_block_size(bplan, F, b0 + j),
                               j == 0 ? main_stream : ctx->aux_stream, &blk[j]));
        auto each = [&](auto &&fn) -> int {
            for (int j = 0; j < nb; ++j) {
                ctx->stream = blk[j].stream;
                GSS_TRY(fn(blk[j]));
            }
            return GSS_OK;
        };
        GSS_TRY(each([&](EmBlock &b) { return prepare(b); }));
        for (int it = 0; it < iterations; ++it)
            GSS_TRY(each([&](EmBlock &b) {
                b.a.masked = 1;
                b.a.aff_eps = 1e-10;
                return em_iteration(b, it == 0);
            }));
        // fit(initialization=model, iterations=post-1): no mask, default clip
        for (int it = 0; it < iterations_post - 1; ++it)
            GSS_TRY(each([&](EmBlock &b) {
                b.a.masked = 0;
                b.a.aff_eps = 1e-10;
                return em_iteration(b, false);
            }));
        // predict: affiliation_eps = 0; mask only when iterations_post == 0
        GSS_TRY(each([&](EmBlock &b) {
            b.a.masked = iterations_post == 0 ? 1 : 0;
            b.a.aff_eps = 0.0;
            return estep(b, MODE_PREDICT);
        }));
    }
    ctx->stream = main_stream;
    if (group > 1) {
        GSS_HIP_CHECK(ctx, hipEventRecord(ctx->ev_join, ctx->aux_stream));
        GSS_HIP_CHECK(ctx, hipStreamWaitEvent(main_stream, ctx->ev_join, 0));
    }
    return GSS_OK;
}
