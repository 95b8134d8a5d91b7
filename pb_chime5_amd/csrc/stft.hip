// STFT / iSTFT / frame-activity kernels and layout helpers.
//
// A1  nara_wpe.utils.stft   (reference call site core.py:305-312)
// A9  nara_wpe.utils.istft  (core.py:314-321)
// A5' activity_time_to_frequency (database/chime5/database.py:409-472)
//
// All three are HBM-bound streaming kernels (<1 % of the path's time).  The FFT
// is a radix-2 decimation-in-time transform held in LDS; two real sequences
// share one complex transform (two channels of a frame in the forward direction,
// two frames in the inverse direction), which halves LDS traffic and lets one
// workgroup write a contiguous run of channels of the (F,T,D) tensor.
#include <cstdlib>

#include "gss_internal.h"

namespace {

constexpr int FFT_THREADS = 256;
constexpr int STFT_RUN = 8;      // consecutive frames per XCD (stft_kernel's grid mapping)

__device__ __forceinline__ unsigned bitrev(unsigned v, int bits) {
    return __brev(v) >> (32 - bits);
}

// In-place radix-2 DIT over `nseq` sequences of length n (already bit-reversed) stored back
// to back in LDS, two stages per pass: a thread takes the four points that stages s and s + 1
// connect, runs both butterflies in registers and writes them back -- the same operations in
// the same order as one stage per pass (the results are the same bits), half the LDS round
// trips and half the barriers.  An odd stage count starts with one single stage.
// inverse: conjugated twiddles, no scaling.
__device__ void fft_lds(cplx *s, int n, int log2n, int nseq, const cplx *tw, bool inverse) {
    int stage = 0;
    if (log2n & 1) {
        const int half_total = nseq * (n >> 1);
        for (int b = threadIdx.x; b < half_total; b += blockDim.x) {
            cplx *p = s + 2 * b;                    // stage 0: neighbours, twiddle 1
            cplx w = tw[0];
            if (inverse) w.y = -w.y;
            const cplx a = p[0], t = c_mul(p[1], w);
            p[0] = c_add(a, t);
            p[1] = c_sub(a, t);
        }
        __syncthreads();
        stage = 1;
    }
    const int quarter = n >> 2, quarter_total = nseq * quarter;
    for (; stage < log2n; stage += 2) {
        const int half = 1 << stage;
        const int tw1 = n >> (stage + 1), tw2 = n >> (stage + 2);
        for (int b = threadIdx.x; b < quarter_total; b += blockDim.x) {
            const int seq = b / quarter;
            const int i = b - seq * quarter;
            const int j = i & (half - 1);
            cplx *p = s + seq * n + ((i >> stage) << (stage + 2)) + j;
            cplx w1 = tw[j * tw1], w2a = tw[j * tw2], w2b = tw[(j + half) * tw2];
            if (inverse) {
                w1.y = -w1.y;
                w2a.y = -w2a.y;
                w2b.y = -w2b.y;
            }
            const cplx x0 = p[0], x1 = p[half], x2 = p[2 * half], x3 = p[3 * half];
            cplx t = c_mul(x1, w1);
            const cplx y0 = c_add(x0, t), y1 = c_sub(x0, t);
            t = c_mul(x3, w1);
            const cplx y2 = c_add(x2, t), y3 = c_sub(x2, t);
            t = c_mul(y2, w2a);
            p[0] = c_add(y0, t);
            p[2 * half] = c_sub(y0, t);
            t = c_mul(y3, w2b);
            p[half] = c_add(y1, t);
            p[3 * half] = c_sub(y1, t);
        }
        __syncthreads();
    }
}

// grid: (T, ceil(D / (2*PAIRS))); one workgroup transforms up to 2*PAIRS channels
// of one frame.
// TIn: double (the reference's float64 samples) or int16_t (PCM as it sits in the WAV file;
// multiplied by in_scale = 2^-15, which is exactly what the reference's loader does).
template <int PAIRS, typename TIn>
__global__ __launch_bounds__(FFT_THREADS) void stft_kernel(
    const TIn *__restrict__ x, double in_scale, int D, int64_t N, int64_t T, int size, int log2n, int shift,
    int pad, const double *__restrict__ window, const cplx *__restrict__ twiddle,
    cplx *__restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx *s = reinterpret_cast<cplx *>(smem);            // PAIRS * size
    cplx *tw = s + PAIRS * size;                          // size / 2
    unsigned *nz = reinterpret_cast<unsigned *>(tw + size / 2);   // bit c: channel c non-zero
    // 1-D grid, XCD-mapped (workgroup id L runs on XCD L % 8): every XCD takes runs of
    // STFT_RUN consecutive frames, all channel groups of a frame next to each other.
    // Consecutive frames share three quarters of their samples, and the channel groups of a
    // frame write neighbouring 64-byte pieces of the same lines of Y (F,T,D): with the frames
    // dealt out round robin every sample was fetched into four L2s.
    const int ngrp = (D + 2 * PAIRS - 1) / (2 * PAIRS);
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int grp = q % ngrp, tl = q / ngrp;
    const int64_t t = (int64_t)(tl / STFT_RUN) * (8 * STFT_RUN) + xcd * STFT_RUN + tl % STFT_RUN;
    if (t >= T) return;
    const int d0 = grp * 2 * PAIRS;
    const int F = size / 2 + 1;

    if (threadIdx.x == 0) *nz = 0u;
    // Loads in batches of LB per thread, all in flight before the first LDS store: a
    // load - store loop is a chain of dependent round trips (three per trip here), and a
    // workgroup lives for little more than its ten FFT passes.
    constexpr int LB = 8;
    for (int base = 0; base < size / 2; base += 2 * FFT_THREADS) {
        cplx tv[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = base + threadIdx.x + j * FFT_THREADS;
            tv[j] = i < size / 2 ? twiddle[i] : c_make(0.0, 0.0);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = base + threadIdx.x + j * FFT_THREADS;
            if (i < size / 2) tw[i] = tv[j];
        }
    }
    __syncthreads();
    const int64_t n0 = t * shift - pad;
    unsigned mine = 0u;
    for (int base = 0; base < PAIRS * size; base += LB * FFT_THREADS) {
        double wv[LB];
        TIn ra[LB], rb[LB];
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            const int idx = base + threadIdx.x + j * FFT_THREADS;
            const int pr = idx / size;
            const int i = idx - pr * size;
            const int64_t n = n0 + i;
            const int da = d0 + 2 * pr, db = da + 1;
            const bool in = idx < PAIRS * size && n >= 0 && n < N;
            wv[j] = in ? window[i] : 0.0;
            ra[j] = in && da < D ? x[(int64_t)da * N + n] : (TIn)0;
            rb[j] = in && db < D ? x[(int64_t)db * N + n] : (TIn)0;
        }
#pragma unroll
        for (int j = 0; j < LB; ++j) {
            const int idx = base + threadIdx.x + j * FFT_THREADS;
            if (idx >= PAIRS * size) continue;
            const int pr = idx / size;
            const int i = idx - pr * size;
            const double va = ((double)ra[j] * in_scale) * wv[j];
            const double vb = ((double)rb[j] * in_scale) * wv[j];
            s[pr * size + bitrev(i, log2n)] = c_make(va, vb);
            if (va != 0.0) mine |= 1u << (2 * pr);
            if (vb != 0.0) mine |= 1u << (2 * pr + 1);
        }
    }
    if (mine) atomicOr(nz, mine);
    __syncthreads();
    fft_lds(s, size, log2n, PAIRS, tw, false);

    // Z = FFT(a + i b):  A[f] = (Z[f] + conj(Z[n-f])) / 2,  B[f] = (Z[f] - conj(Z[n-f])) / (2i)
    // The separation leaks ~1e-16 of the partner channel.  An all-zero frame (a dead
    // microphone, digital silence) must come out as exact zeros, as rfft gives in the
    // reference: an exactly singular WPE correlation matrix takes the reference's lstsq
    // branch (math/solve.py:95-114, minimum norm = zero taps on the dead channel), an almost
    // singular one would be solved with taps of size 1e16 on the leaked noise.
    const unsigned nonzero = *nz;
    const int nch = min(2 * PAIRS, D - d0);
    for (int idx = threadIdx.x; idx < F * nch; idx += blockDim.x) {
        const int f = idx / nch;
        const int c = idx - f * nch;
        const int pr = c >> 1;
        const cplx z = s[pr * size + f];
        const cplx zc = s[pr * size + ((size - f) & (size - 1))];
        cplx v;
        if ((c & 1) == 0) {
            v = c_make(0.5 * (z.x + zc.x), 0.5 * (z.y - zc.y));
        } else {
            v = c_make(0.5 * (z.y + zc.y), 0.5 * (zc.x - z.x));
        }
        if (!((nonzero >> c) & 1u)) v = c_make(0.0, 0.0);
        Y[((int64_t)f * T + t) * D + d0 + c] = v;
    }
}

// Inverse: one workgroup handles frames 2*b and 2*b+1 of X (T,F) and writes
// synthesis-windowed frames into frm (T, size).
__global__ __launch_bounds__(FFT_THREADS) void istft_frames_kernel(
    const cplx *__restrict__ X, int64_t T, int size, int log2n,
    const double *__restrict__ syn, const cplx *__restrict__ twiddle,
    double *__restrict__ frm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx *s = reinterpret_cast<cplx *>(smem);   // size
    cplx *tw = s + size;                         // size / 2
    const int F = size / 2 + 1;
    const int64_t ta = (int64_t)blockIdx.x * 2, tb = ta + 1;
    for (int i = threadIdx.x; i < size / 2; i += blockDim.x) tw[i] = twiddle[i];
    // Hermitian extension of both spectra (irfft ignores Im of DC and Nyquist),
    // Z = A + i B.
    for (int k = threadIdx.x; k < size; k += blockDim.x) {
        const int f = k <= size / 2 ? k : size - k;
        cplx a = X[ta * F + f];
        cplx b = tb < T ? X[tb * F + f] : c_make(0.0, 0.0);
        if (f == 0 || f == size / 2) {
            a.y = 0.0;
            b.y = 0.0;
        }
        if (k > size / 2) {
            a.y = -a.y;
            b.y = -b.y;
        }
        s[bitrev(k, log2n)] = c_make(a.x - b.y, a.y + b.x);
    }
    __syncthreads();
    fft_lds(s, size, log2n, 1, tw, true);
    const double scale = 1.0 / (double)size;
    for (int i = threadIdx.x; i < size; i += blockDim.x) {
        const double w = syn[i];
        frm[ta * size + i] = w * (s[i].x * scale);
        if (tb < T) frm[tb * size + i] = w * (s[i].y * scale);
    }
}

// ---- any even window length (not a power of two): direct DFT.  The reference's default is
// 1024 and every BASELINE config uses it; get_enhancer(stft_size=...) (core.py:577) takes any
// length though, so other sizes get a plain O(size^2) transform per frame -- a few ms per
// utterance instead of 0.2, and the same numbers as numpy's rfft / irfft to 1e-15 relative.
// tw holds exp(-2 pi i j / size) for j < size / 2; the second half is its negative.
__device__ __forceinline__ cplx tw_full(const cplx *tw, int j, int half) {
    const cplx w = tw[j < half ? j : j - half];
    return j < half ? w : c_make(-w.x, -w.y);
}

// grid (T, D), block 256: one frame of one channel; thread = frequency bins tid, tid + 256, ...
template <typename TIn>
__global__ __launch_bounds__(FFT_THREADS) void stft_dft_kernel(
    const TIn *__restrict__ x, double in_scale, int D, int64_t N, int64_t T, int size, int shift,
    int pad, const double *__restrict__ window, const cplx *__restrict__ twiddle,
    cplx *__restrict__ Y) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx *tw = reinterpret_cast<cplx *>(smem);                   // size / 2
    double *xw = reinterpret_cast<double *>(tw + size / 2);      // size
    const int64_t t = blockIdx.x;
    const int d = blockIdx.y, half = size / 2, F = half + 1;
    const int64_t n0 = t * shift - pad;
    for (int i = threadIdx.x; i < half; i += blockDim.x) tw[i] = twiddle[i];
    for (int i = threadIdx.x; i < size; i += blockDim.x) {
        const int64_t n = n0 + i;
        xw[i] = (n >= 0 && n < N) ? ((double)x[(int64_t)d * N + n] * in_scale) * window[i] : 0.0;
    }
    __syncthreads();
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        double re = 0.0, im = 0.0;
        int j = 0;                                  // (n f) mod size
        for (int n = 0; n < size; ++n) {
            const cplx w = tw_full(tw, j, half);
            re = fma(xw[n], w.x, re);
            im = fma(xw[n], w.y, im);
            j += f;
            if (j >= size) j -= size;
        }
        if (f == 0 || f == half) im = 0.0;          // rfft: DC and Nyquist are real
        Y[((int64_t)f * T + t) * D + d] = c_make(re, im);
    }
}

// grid (T), block 256: one frame; thread = samples tid, tid + 256, ...
__global__ __launch_bounds__(FFT_THREADS) void istft_dft_kernel(
    const cplx *__restrict__ X, int64_t T, int size, const double *__restrict__ syn,
    const cplx *__restrict__ twiddle, double *__restrict__ frm) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    cplx *tw = reinterpret_cast<cplx *>(smem);                   // size / 2
    cplx *xs = tw + size / 2;                                    // F
    const int64_t t = blockIdx.x;
    const int half = size / 2, F = half + 1;
    for (int i = threadIdx.x; i < half; i += blockDim.x) tw[i] = twiddle[i];
    for (int f = threadIdx.x; f < F; f += blockDim.x) xs[f] = X[t * F + f];
    __syncthreads();
    const double scale = 1.0 / (double)size;
    for (int i = threadIdx.x; i < size; i += blockDim.x) {
        // irfft: x_i = (X_0 + (-1)^i X_half + 2 sum_{0 < k < half} Re(X_k e^{+2 pi i i k / size})) / size
        double acc = xs[0].x + ((i & 1) ? -xs[half].x : xs[half].x);
        int j = i;                                  // (i k) mod size for k = 1
        for (int k = 1; k < half; ++k) {
            const cplx w = tw_full(tw, j, half);    // e^{-i theta}: Re(X conj(w)) = Xr wr + Xi wi
            acc = fma(2.0 * xs[k].x, w.x, acc);
            acc = fma(2.0 * xs[k].y, w.y, acc);
            j += i;
            if (j >= size) j -= size;
        }
        frm[t * size + i] = syn[i] * (acc * scale);
    }
}

// Overlap-add in increasing frame order (np.add.at upstream), then drop the
// fading pad.
__global__ void istft_ola_kernel(const double *__restrict__ frm, int64_t T, int size, int shift,
                                 int pad, int64_t n_out, double *__restrict__ out) {
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= n_out) return;
    const int64_t m = n + pad;
    int64_t t_lo = m - size + 1;
    t_lo = t_lo <= 0 ? 0 : (t_lo + shift - 1) / shift;
    int64_t t_hi = m / shift;
    if (t_hi > T - 1) t_hi = T - 1;
    double acc = 0.0;
    for (int64_t t = t_lo; t <= t_hi; ++t) acc += frm[t * size + (m - t * shift)];
    out[n] = acc;
}

// One WAVE per (speaker, frame): the lanes read the window's bytes 64 at a time (coalesced, all
// loads independent) and the wave ORs them.  (One THREAD per frame walked its 1024 bytes alone:
// 22 us for 5 x 240 000 bytes, a latency chain of 1024 byte loads on 74 waves.)
__global__ __launch_bounds__(256) void activity_kernel(const uint8_t *__restrict__ act, int K,
                                                       int64_t N, int64_t T, int size, int shift,
                                                       int pad, uint8_t *__restrict__ out) {
    const int64_t idx = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (idx >= (int64_t)K * T) return;
    const int lane = threadIdx.x & 63;
    const int k = (int)(idx / T);
    const int64_t t = idx - (int64_t)k * T;
    int64_t a = t * shift - pad, b = a + size;
    if (a < 0) a = 0;
    if (b > N) b = N;
    int any = 0;
    const uint8_t *p = act + (int64_t)k * N;
    for (int64_t n = a + lane; n < b; n += 64) any |= p[n];
    const bool on = __any(any != 0);
    if (lane == 0) out[idx] = on ? 1 : 0;
}

// (D,T,F) -> (F,T,D) and back, tiled through LDS so both sides are coalesced
// along their fastest index.  Tile: 32 f x 32 d (t fixed per block.y).
__global__ void dtf_to_ftd_kernel(const cplx *__restrict__ src, int D, int64_t T, int F,
                                  cplx *__restrict__ dst) {
    __shared__ cplx tile[32][33];
    const int64_t t = blockIdx.y;
    const int f0 = blockIdx.x * 32;
    for (int d0 = 0; d0 < D; d0 += 32) {
        for (int r = threadIdx.y; r < 32; r += blockDim.y) {
            const int d = d0 + r, f = f0 + threadIdx.x;
            if (d < D && f < F) tile[r][threadIdx.x] = src[((int64_t)d * T + t) * F + f];
        }
        __syncthreads();
        for (int r = threadIdx.y; r < 32; r += blockDim.y) {
            const int f = f0 + r, d = d0 + threadIdx.x;
            if (d < D && f < F) dst[((int64_t)f * T + t) * D + d] = tile[threadIdx.x][r];
        }
        __syncthreads();
    }
}

__global__ void ftd_to_dtf_kernel(const cplx *__restrict__ src, int F, int64_t T, int D,
                                  cplx *__restrict__ dst) {
    __shared__ cplx tile[32][33];
    const int64_t t = blockIdx.y;
    const int f0 = blockIdx.x * 32;
    for (int d0 = 0; d0 < D; d0 += 32) {
        for (int r = threadIdx.y; r < 32; r += blockDim.y) {
            const int f = f0 + r, d = d0 + threadIdx.x;
            if (d < D && f < F) tile[r][threadIdx.x] = src[((int64_t)f * T + t) * D + d];
        }
        __syncthreads();
        for (int r = threadIdx.y; r < 32; r += blockDim.y) {
            const int d = d0 + r, f = f0 + threadIdx.x;
            if (d < D && f < F) dst[((int64_t)d * T + t) * F + f] = tile[threadIdx.x][r];
        }
        __syncthreads();
    }
}

// dst[c][a] = src[a][c] for a in [0,A), c in [0,C), batched over `batch` slabs.
__global__ void transpose_f64_kernel(const double *__restrict__ src, int64_t A, int64_t C,
                                     double *__restrict__ dst) {
    __shared__ double tile[32][33];
    const int64_t a0 = (int64_t)blockIdx.y * 32, c0 = (int64_t)blockIdx.x * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int64_t a = a0 + r, c = c0 + threadIdx.x;
        if (a < A && c < C) tile[r][threadIdx.x] = src[a * C + c];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int64_t c = c0 + r, a = a0 + threadIdx.x;
        if (a < A && c < C) dst[c * A + a] = tile[threadIdx.x][r];
    }
}

// Beamformer types 'ch2' (Obs[2]) and 'sum' (core.py:259-262): Y (F,T,D) -> (T,F)
__global__ void channel_pick_kernel(const cplx *__restrict__ Y, int F, int64_t T, int D,
                                    int mode, cplx *__restrict__ Xhat) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)F * T) return;
    const int f = idx / T;
    const int64_t t = idx - (int64_t)f * T;
    const cplx *y = Y + idx * D;
    cplx v;
    if (mode == 1) {
        v = y[2];
    } else {
        v = c_make(0.0, 0.0);
        for (int d = 0; d < D; ++d) v = c_add(v, y[d]);
    }
    Xhat[t * F + f] = v;
}

// postfilter 'mask_mul' (core.py:270-271): X_hat (T,F) *= target_mask (F,T)
__global__ void mask_mul_kernel(cplx *__restrict__ Xhat, const double *__restrict__ mask, int F,
                                int64_t T) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)F * T) return;
    const int64_t t = idx / F;
    const int f = idx - t * F;
    const double m = mask[(int64_t)f * T + t];
    Xhat[idx] = c_scale(Xhat[idx], m);
}

int ilog2(int n) {
    int l = 0;
    while ((1 << l) < n) ++l;
    return l;
}

}  // namespace

size_t stft_workspace_bytes(int64_t T, int size) {
    return align_up(sizeof(double) * (size_t)(T + 1) * size) + 4096;
}

template <int PAIRS, typename TIn>
static void stft_launch(gss_ctx *ctx, const TIn *x, double in_scale, int D, int64_t N, int64_t T,
                        int size, int shift, int pad, cplx *Y) {
    const int64_t runs = (T + 8 * STFT_RUN - 1) / (8 * STFT_RUN);
    dim3 grid((unsigned)(runs * 8 * STFT_RUN * ((D + 2 * PAIRS - 1) / (2 * PAIRS))));
    const size_t lds = sizeof(cplx) * (PAIRS * size + size / 2) + 16;
    hipLaunchKernelGGL((stft_kernel<PAIRS, TIn>), grid, dim3(FFT_THREADS), lds, ctx->stream, x,
                       in_scale, D, N, T, size, ilog2(size), shift, pad, ctx->win_analysis,
                       ctx->twiddle, Y);
}

// x: (D, N) samples, double (in_type 0) or int16 PCM scaled by 2^-15 (in_type 1)
int stft_run(gss_ctx *ctx, const void *x, int in_type, int D, int64_t N, int fading, cplx *Y) {
    const int size = ctx->stft_size, shift = ctx->stft_shift;
    const int64_t T = gss_stft_num_frames(N, size, shift, fading);
    const int pad = fading ? size - shift : 0;
    GSS_REQUIRE(ctx, T < 2147483647, GSS_ERR_UNSUPPORTED, "too many frames");
    GSS_REQUIRE(ctx, in_type == 0 || in_type == 1, GSS_ERR_INVALID, "sample type %d", in_type);
    GSS_PROF(ctx, "stft");
    const double *xd = static_cast<const double *>(x);
    const int16_t *xi = static_cast<const int16_t *>(x);
    const double pcm = 1.0 / 32768.0;
    if ((size & (size - 1)) != 0) {                 // not a power of two: direct DFT
        const size_t lds = sizeof(cplx) * (size / 2) + sizeof(double) * size;
        const dim3 grid((unsigned)T, (unsigned)D);
        if (in_type == 0)
            hipLaunchKernelGGL((stft_dft_kernel<double>), grid, dim3(FFT_THREADS), lds, ctx->stream, xd,
                               1.0, D, N, T, size, shift, pad, ctx->win_analysis, ctx->twiddle, Y);
        else
            hipLaunchKernelGGL((stft_dft_kernel<int16_t>), grid, dim3(FFT_THREADS), lds, ctx->stream,
                               xi, pcm, D, N, T, size, shift, pad, ctx->win_analysis, ctx->twiddle, Y);
        GSS_LAUNCH_CHECK(ctx, "stft_dft_kernel");
        return GSS_OK;
    }
    // 2 pairs (4 channels, 36 KB of LDS) per workgroup: 4 workgroups per CU hide the ten
    // barriers of the radix-2 passes better than 2 workgroups of 4 pairs (0.20 vs 0.24 ms)
    if (size <= 1024) {
        if (in_type == 0) stft_launch<2>(ctx, xd, 1.0, D, N, T, size, shift, pad, Y);
        else stft_launch<2>(ctx, xi, pcm, D, N, T, size, shift, pad, Y);
    } else {
        if (in_type == 0) stft_launch<1>(ctx, xd, 1.0, D, N, T, size, shift, pad, Y);
        else stft_launch<1>(ctx, xi, pcm, D, N, T, size, shift, pad, Y);
    }
    GSS_LAUNCH_CHECK(ctx, "stft_kernel");
    return GSS_OK;
}

int istft_run(gss_ctx *ctx, const cplx *X, int64_t T, int fading, double *x) {
    const int size = ctx->stft_size, shift = ctx->stft_shift;
    const int pad = fading ? size - shift : 0;
    const int64_t n_out = gss_istft_num_samples(T, size, shift, fading);
    double *frm = arena_alloc_t<double>(ctx, (size_t)T * size);
    GSS_REQUIRE(ctx, frm, GSS_ERR_NOMEM, "istft workspace");
    if ((size & (size - 1)) != 0) {
        GSS_PROF(ctx, "istft_frames");
        const size_t lds = sizeof(cplx) * (size / 2 + size / 2 + 1);
        hipLaunchKernelGGL(istft_dft_kernel, dim3((unsigned)T), dim3(FFT_THREADS), lds, ctx->stream, X,
                           T, size, ctx->win_synthesis, ctx->twiddle, frm);
        GSS_LAUNCH_CHECK(ctx, "istft_dft_kernel");
    } else {
        GSS_PROF(ctx, "istft_frames");
        size_t lds = sizeof(cplx) * (size + size / 2);
        hipLaunchKernelGGL(istft_frames_kernel, dim3((unsigned)((T + 1) / 2)), dim3(FFT_THREADS),
                           lds, ctx->stream, X, T, size, ilog2(size), ctx->win_synthesis,
                           ctx->twiddle, frm);
        GSS_LAUNCH_CHECK(ctx, "istft_frames_kernel");
    }
    if (n_out > 0) {
        GSS_PROF(ctx, "istft_ola");
        hipLaunchKernelGGL(istft_ola_kernel, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0,
                           ctx->stream, frm, T, size, shift, pad, n_out, x);
        GSS_LAUNCH_CHECK(ctx, "istft_ola_kernel");
    }
    return GSS_OK;
}

int activity_run(gss_ctx *ctx, const uint8_t *act, int K, int64_t N, int fading, uint8_t *out) {
    const int size = ctx->stft_size, shift = ctx->stft_shift;
    const int64_t T = gss_stft_num_frames(N, size, shift, fading);
    const int pad = fading ? size - shift : 0;
    GSS_PROF(ctx, "activity");
    const int64_t total = (int64_t)K * T;
    hipLaunchKernelGGL(activity_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0,
                       ctx->stream, act, K, N, T, size, shift, pad, out);
    GSS_LAUNCH_CHECK(ctx, "activity_kernel");
    return GSS_OK;
}

int channel_pick_run(gss_ctx *ctx, const cplx *Y, int F, int64_t T, int D, int mode,
                     cplx *Xhat) {
    GSS_PROF(ctx, "channel_pick");
    const int64_t total = (int64_t)F * T;
    hipLaunchKernelGGL(channel_pick_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       ctx->stream, Y, F, T, D, mode, Xhat);
    GSS_LAUNCH_CHECK(ctx, "channel_pick_kernel");
    return GSS_OK;
}

int mask_mul_run(gss_ctx *ctx, cplx *Xhat, const double *mask_ft, int F, int64_t T) {
    GSS_PROF(ctx, "mask_mul");
    const int64_t total = (int64_t)F * T;
    hipLaunchKernelGGL(mask_mul_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                       ctx->stream, Xhat, mask_ft, F, T);
    GSS_LAUNCH_CHECK(ctx, "mask_mul_kernel");
    return GSS_OK;
}

extern "C" int gss_layout_dtf_to_ftd(gss_ctx *ctx, const gss_cplx *src, int D, int64_t T, int F,
                                     gss_cplx *dst) {
    if (!ctx || !src || !dst) return GSS_ERR_INVALID;
    GSS_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    GSS_REQUIRE(ctx, T <= 65535, GSS_ERR_UNSUPPORTED, "layout helper: T=%lld > 65535",
                (long long)T);
    hipLaunchKernelGGL(dtf_to_ftd_kernel, dim3((F + 31) / 32, (unsigned)T), dim3(32, 8), 0,
                       ctx->stream, reinterpret_cast<const cplx *>(src), D, T, F,
                       reinterpret_cast<cplx *>(dst));
    GSS_LAUNCH_CHECK(ctx, "dtf_to_ftd_kernel");
    return GSS_OK;
}

extern "C" int gss_layout_ftd_to_dtf(gss_ctx *ctx, const gss_cplx *src, int F, int64_t T, int D,
                                     gss_cplx *dst) {
    if (!ctx || !src || !dst) return GSS_ERR_INVALID;
    GSS_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    GSS_REQUIRE(ctx, T <= 65535, GSS_ERR_UNSUPPORTED, "layout helper: T=%lld > 65535",
                (long long)T);
    hipLaunchKernelGGL(ftd_to_dtf_kernel, dim3((F + 31) / 32, (unsigned)T), dim3(32, 8), 0,
                       ctx->stream, reinterpret_cast<const cplx *>(src), F, T, D,
                       reinterpret_cast<cplx *>(dst));
    GSS_LAUNCH_CHECK(ctx, "ftd_to_dtf_kernel");
    return GSS_OK;
}

extern "C" int gss_layout_permute_f64(gss_ctx *ctx, const double *src, int64_t A, int64_t B,
                                      int64_t C, int perm, double *dst) {
    if (!ctx || !src || !dst) return GSS_ERR_INVALID;
    GSS_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    // every supported permutation is a 2-D transpose of a (rows, cols) view
    int64_t rows, cols;
    if (perm == 0) {          // (A,B,C) -> (B,C,A): rows = A, cols = B*C
        rows = A;
        cols = B * C;
    } else if (perm == 1) {   // (A,B,C) -> (C,A,B): rows = A*B, cols = C
        rows = A * B;
        cols = C;
    } else if (perm == 2) {   // (A,B) -> (B,A)
        rows = A;
        cols = B;
    } else {
        return gss_fail(ctx, GSS_ERR_INVALID, "unknown permutation %d", perm);
    }
    GSS_REQUIRE(ctx, (rows + 31) / 32 <= 65535, GSS_ERR_UNSUPPORTED, "permute: too many rows");
    hipLaunchKernelGGL(transpose_f64_kernel,
                       dim3((unsigned)((cols + 31) / 32), (unsigned)((rows + 31) / 32)),
                       dim3(32, 8), 0, ctx->stream, src, rows, cols, dst);
    GSS_LAUNCH_CHECK(ctx, "transpose_f64_kernel");
    return GSS_OK;
}
