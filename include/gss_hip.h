/*
 * gss_hip.h -- C ABI of libgss_hip.so: the MI355X (gfx950) implementation of the
 * pb_chime5 guided-source-separation hot path
 *
 *     STFT -> WPE -> CACGMM (guided EM) -> MVDR-Souden (+BAN) -> iSTFT
 *
 * i.e. everything /root/reference/pb_chime5/core.py:514-571
 * (Enhancer.enhance_observation) executes per utterance.  The reference has no
 * FFI layer of its own: its boundary is a set of Python callables that hand NumPy
 * arrays to nara_wpe / pb_bss.  Each entry point below names the reference call
 * it replaces (file:line).  The Python host package (pb_chime5_amd) binds this
 * header with ctypes and keeps the reference's Python signatures on top of it.
 *
 * Conventions
 *  - every function returns 0 on success or a negative gss_status; the message
 *    for the last failure on a context is available from gss_last_error().
 *  - no C++ types, no exceptions, no torch types cross this boundary.
 *  - one gss_ctx per GPU (per host thread); a context is not thread-safe,
 *    different contexts are independent.  All work of a context is ordered on
 *    one HIP stream (its own, or one adopted with gss_set_stream()); a second
 *    stream the fused pipeline may use inside a call is forked from and joined to
 *    it by events (gss_set_utterances_in_flight()).
 *  - the caller owns every buffer it passes in.  Pointers named *_dev are device
 *    pointers valid on the context's GPU (from gss_dev_malloc(), or any other
 *    allocator of the same process, e.g. torch); pointers named *_host are host
 *    pointers.  Device entry points are asynchronous on the context's stream and
 *    never retain caller pointers after the work they enqueue has run.
 *  - arithmetic type: float64 / complex128 end to end, like the reference
 *    ("gss_cplx" = interleaved {re, im} doubles).
 *
 * Canonical device layouts (row-major, last index fastest)
 *    time signal   x      (D, N)      double
 *    STFT tensor   Y      (F, T, D)   gss_cplx      "FTD"; F = size/2 + 1
 *    activity      act    (K, N)      uint8  (time)  /  (K, T) uint8 (frames)
 *    posteriors    gamma  (F, K, T)   double
 *    masks         m      (F, T)      double
 *    beamformed    Xhat   (T, F)      gss_cplx       (the reference's layout)
 * The reference's (D, T, F) / (K, T, F) / (T, F) layouts are produced / consumed
 * with the gss_layout_* helpers.
 */
#ifndef GSS_HIP_H
#define GSS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gss_ctx gss_ctx;
typedef struct { double re, im; } gss_cplx;

typedef enum {
    GSS_OK = 0,
    GSS_ERR_INVALID = -1,      /* bad argument (-> AssertionError / ValueError)   */
    GSS_ERR_HIP = -2,          /* HIP runtime failure                              */
    GSS_ERR_NOMEM = -3,        /* device allocation failed                         */
    GSS_ERR_UNSUPPORTED = -4   /* configuration outside the built kernels          */
                               /* (-> NotImplementedError)                         */
} gss_status;

/* Limits of the built kernels. */
#define GSS_MAX_CHANNELS 32    /* reference asserts D < 30 (beamforming_wrapper.py:44) */
#define GSS_MAX_CLASSES 19     /* pb_bss asserts K < 20 (CACGMMTrainer.fit); CHiME-5/6: K <= 5 */
#define GSS_MAX_STFT_SIZE 4096 /* any even length; powers of two take the FFT kernels, others a direct DFT */

/* ABI revision of this header.  Bumped whenever an entry point changes its argument list
 * or a struct its layout (round 2 added `psd_context` to gss_wpe and `wpe_psd_context` to
 * gss_params: revision 2; rounds 3, 4 and 5 added entry points only: revisions 3, 4, 5).  A binder compares
 * gss_abi_version() with the GSS_ABI_VERSION it was written against before any other call. */
#define GSS_ABI_VERSION 6
int gss_abi_version(void);

/* ---- context ----------------------------------------------------------- */
/* Number of visible HIP devices (0 if none / no driver).  Ranks of a node pick
 * LOCAL_RANK % gss_device_count() (pb_chime5_amd.parallel, replacing dlp_mpi's
 * rank handling at core.py:363-381). */
int gss_device_count(void);
/* PCI address of a device ("0000:c1:00.0", NUL-terminated, `len` >= 16 bytes): the host side
 * reads /sys/bus/pci/devices/<address>/{numa_node,local_cpulist} to run a rank's threads on
 * the socket its GPU hangs off (the reference leaves placement to mpiexec, README.md:108-111). */
int gss_device_pci_bus_id(int device_id, char *buf, int len);
int gss_create(int device_id, gss_ctx **ctx);
int gss_destroy(gss_ctx *ctx);
const char *gss_last_error(gss_ctx *ctx);
const char *gss_version(void);
/* Adopt an existing hipStream_t (e.g. torch's current stream); NULL restores the
 * context's own stream. */
int gss_set_stream(gss_ctx *ctx, void *hip_stream);
/* How many utterances the caller keeps in flight on this context's GPU (over all of its
 * contexts); 0 = not said (the default).  Exactly 1 -- one utterance at a time, the loop of
 * Enhancer.enhance_example, /root/reference/pb_chime5/core.py:363-392 -- lets
 * gss_enhance_observation*() run the two halves of the frequencies of the WPE stage side by
 * side on a second, internal stream: one half's solve under the other's correlation, the same
 * bits, the utterance ~1.5 % sooner.  With two or more utterances in flight (the session
 * driver) they already fill each other's idle time; with 0 the call stays on the context's
 * stream alone as well (and per-kernel timings mean what they say).  GSS_ERR_INVALID for n < 0. */
int gss_set_utterances_in_flight(gss_ctx *ctx, int n);
int gss_synchronize(gss_ctx *ctx);

/* ---- device memory plumbing (for hosts that have no allocator) ---------- */
int gss_dev_malloc(gss_ctx *ctx, size_t bytes, void **dev_ptr);
int gss_dev_free(gss_ctx *ctx, void *dev_ptr);
int gss_memcpy_h2d(gss_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int gss_memcpy_d2h(gss_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int gss_memset(gss_ctx *ctx, void *dst_dev, int value, size_t bytes);
/* Page-locked host memory and copies that return at once (ordered on the context's stream;
 * the host buffer must stay untouched until gss_synchronize() or a later synchronous call
 * returns).  With pageable memory the *_async forms behave like the plain ones.  The session
 * driver reads WAV samples straight into such a block (replaces the reference's soundfile
 * read + float64 conversion + np.array stacking, io/audioread.py:34-226, core.py:427-470).
 * gss_host_malloc / gss_host_free touch no context state (the context only names the GPU
 * and receives the error message) and may be called from any thread.  gss_host_free does not
 * wait for the context's stream itself -- no copy from / to the block may be in flight -- but
 * hipHostFree underneath waits for the WHOLE device on ROCm: free page-locked blocks when a
 * session ends, not between utterances (the session driver parks outgrown blocks until then). */
int gss_host_malloc(gss_ctx *ctx, size_t bytes, void **host_ptr);
int gss_host_free(gss_ctx *ctx, void *host_ptr);
int gss_memcpy_h2d_async(gss_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int gss_memcpy_d2h_async(gss_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);

/* ---- per-kernel timing (HIP events on the context's stream) -------------- */
int gss_profile_enable(gss_ctx *ctx, int on);
/* Restrict the timing to one kernel name (NULL or "" = all).  Two events per timed
 * launch sit in the stream, which costs about 4 us of stream time each on MI355X:
 * timing all ~260 launches of an utterance slows it by 6 %, timing one kernel does not. */
int gss_profile_filter(gss_ctx *ctx, const char *kernel);
int gss_profile_reset(gss_ctx *ctx);
/* Writes a JSON object {"kernel": {"calls": n, "ms": total}, ...} (synchronises). */
int gss_profile_report(gss_ctx *ctx, char *buf, size_t buf_size);

/* ---- STFT geometry ------------------------------------------------------- */
/* Number of frames of nara_wpe.utils.stft(..., pad=True) (core.py:305-312). */
int64_t gss_stft_num_frames(int64_t num_samples, int size, int shift, int fading);
/* Output length of nara_wpe.utils.istft (core.py:314-321). */
int64_t gss_istft_num_samples(int64_t num_frames, int size, int shift, int fading);
/* nara_wpe.utils._samples_to_stft_frames (core.py:224-237). */
int64_t gss_samples_to_stft_frames(int64_t samples, int size, int shift, int fading);

/* Analysis / synthesis windows (host pointers, `size` doubles each).  The host
 * computes them exactly like nara_wpe (periodic Blackman, biorthogonal synthesis
 * window) so the library holds no window policy of its own. */
int gss_set_windows(gss_ctx *ctx, int size, int shift,
                    const double *analysis_host, const double *synthesis_host);

/* ---- stage entry points (device pointers, asynchronous) ------------------ */

/* A1  Enhancer.stft -> nara_wpe.utils.stft (core.py:305-312).
 * x (D,N) -> Y (F,T,D), T = gss_stft_num_frames(N,...). */
int gss_stft(gss_ctx *ctx, const double *x_dev, int D, int64_t N,
             int fading, gss_cplx *Y_dev);

/* A9  Enhancer.istft -> nara_wpe.utils.istft (core.py:314-321).
 * X (T,F) -> x (gss_istft_num_samples(T,...)). */
int gss_istft(gss_ctx *ctx, const gss_cplx *X_dev, int64_t T, int fading,
              double *x_dev);

/* A5' activity_time_to_frequency (database/chime5/database.py:409-472),
 * stft_pad=True.  act (K,N) uint8 -> (K,T) uint8.  Bit-exact. */
int gss_activity_time_to_frequency(gss_ctx *ctx, const uint8_t *act_dev, int K,
                                   int64_t N, int fading, uint8_t *act_frames_dev);

/* A2  WPE.__call__ -> nara_wpe.wpe.wpe_v8(statistics_mode='full') (core.py:48-58).
 * psd_context (core.py:56,583; nara_wpe.wpe.get_power): the frame power is averaged
 * over the existing frames of [t - psd_context, t + psd_context]; 0 = the reference
 * default.  Y (F,T,D) -> X (F,T,D); X must not alias Y unless iterations == 0. */
int gss_wpe(gss_ctx *ctx, const gss_cplx *Y_dev, int F, int64_t T, int D,
            int taps, int delay, int iterations, int psd_context, gss_cplx *X_dev);

/* The weights of one WPE iteration on their own: nara_wpe.wpe.get_power_inverse(Y,
 * psd_context) as wpe_v6 calls it (mean over channels of |Y|^2, optionally averaged over
 * the existing frames of [t - psd_context, t + psd_context], floored at 1e-10 * its maximum
 * over time, inverted).  Y (F,T,D) -> inverse_power (F,T).  A stage entry point for
 * checking the smoothing on its own; gss_wpe computes the same thing inside. */
int gss_wpe_inverse_power(gss_ctx *ctx, const gss_cplx *Y_dev, int F, int64_t T, int D,
                          int psd_context, double *inverse_power_dev);

/* A4-A6  GSS.__call__ (core.py:154-214): initialisation from the frame activity,
 * CACGMMTrainer.fit(iterations, source_activity_mask) and the post step
 * (iterations_post: 0 = masked predict, 1 = predict, >1 = extra unmasked fit
 * iterations then predict).  Y (F,T,D), act_frames (K,T) -> gamma (F,K,T). */
int gss_cacgmm(gss_ctx *ctx, const gss_cplx *Y_dev, int F, int64_t T, int D,
               const uint8_t *act_frames_dev, int K, int iterations,
               int iterations_post, double *gamma_dev);

/* A0  mask post-processing of enhance_observation (core.py:537-554): zero the
 * context frames, pick the target class, sum the others.
 * gamma (F,K,T) -> target (F,T), distortion (F,T).  drop_context = 0 skips the
 * zeroing (bf_drop_context=False). */
int gss_masks_from_posteriors(gss_ctx *ctx, const double *gamma_dev, int F, int K,
                              int64_t T, int target_index, int drop_context,
                              int64_t start_context_frames,
                              int64_t end_context_frames,
                              double *target_mask_dev, double *distortion_mask_dev);

/* A7+A8  beamform_mvdr_souden_from_masks (beamforming_wrapper.py:108-124) with
 * eps=1e-10: masked PSD matrices, Souden MVDR, one reference channel from the
 * cross-frequency SNR argmax, optional blind analytic normalisation, apply.
 * Y (F,T,D), masks (F,T) -> Xhat (T,F).  ref_channel_dev (device int32, may be
 * NULL) receives the chosen reference channel. */
int gss_mvdr_souden(gss_ctx *ctx, const gss_cplx *Y_dev, int F, int64_t T, int D,
                    const double *target_mask_dev,
                    const double *distortion_mask_dev, int ban,
                    gss_cplx *Xhat_dev, int32_t *ref_channel_dev);

/* The same with the reference channel named by the caller (pb_bss
 * get_mvdr_vector_souden(ref_channel=...), call site beamforming_wrapper.py:58-63). */
int gss_mvdr_souden_ref(gss_ctx *ctx, const gss_cplx *Y_dev, int F, int64_t T, int D,
                        const double *target_mask_dev,
                        const double *distortion_mask_dev, int ban, int ref_channel,
                        gss_cplx *Xhat_dev);

/* Reference channel of the last MVDR run on this context (gss_mvdr_souden or the fused
 * pipeline); synchronises the stream.  -1: a per-channel SNR was not finite -- pb_bss
 * get_optimal_reference_channel asserts np.all(np.isfinite(SNR)) and the reference
 * aborts the utterance with an AssertionError; here Xhat is filled with NaN and the host
 * raises.  <= -2 (after gss_gev or the fused pipeline with the GEV beamformer): the noise
 * PSD matrix of frequency -2 - value is not positive definite -- scipy.linalg.eigh inside
 * pb_bss get_gev_vector raises numpy.linalg.LinAlgError there and the reference aborts the
 * utterance; Xhat is NaN and the host raises the same.  0 after a successful GEV run.
 * INT32_MIN: no beamformer has run yet. */
int gss_last_ref_channel(gss_ctx *ctx, int32_t *ref_channel_host);

/* Number of pivots the WPE solve of the last gss_wpe / fused call on this context zeroed
 * (summed over its iterations and frequencies; synchronises the stream).  The normal
 * equations are solved by Cholesky; a non-positive pivot zeroes that row, which is the
 * minimum-norm answer of stable_solve's lstsq fallback (math/solve.py:95-114) for an
 * all-zero channel.  A count > 0 on live channels means R was rank deficient -- a segment
 * with no more frames than taps * D unknowns -- where np.linalg.solve returns a different
 * (equally arbitrary) minimiser than this library. */
int gss_last_wpe_zero_pivots(gss_ctx *ctx, int64_t *count_host);

/* beamform_gev_from_masks (beamforming_wrapper.py:77-89,192-208): masked PSD
 * matrices, principal generalised eigenvector of (Phi_X, Phi_N) with
 * w^H Phi_N w = 1 (phase arbitrary, as upstream), optional BAN, apply.
 * Y (F,T,D), masks (F,T) -> Xhat (T,F). */
int gss_gev(gss_ctx *ctx, const gss_cplx *Y_dev, int F, int64_t T, int D,
            const double *target_mask_dev, const double *distortion_mask_dev,
            int ban, gss_cplx *Xhat_dev);

/* Layout helpers between the canonical device layouts and the reference's. */
int gss_layout_dtf_to_ftd(gss_ctx *ctx, const gss_cplx *src_dev, int D, int64_t T,
                          int F, gss_cplx *dst_dev);
int gss_layout_ftd_to_dtf(gss_ctx *ctx, const gss_cplx *src_dev, int F, int64_t T,
                          int D, gss_cplx *dst_dev);
/* (A, B, C) double -> (C, A, B) double, e.g. gamma (F,K,T) -> (K,T,F) with
 * A=F,B=K,C=T ... expressed as a generic 3-D permutation dst[p(i)] = src[i]:
 * perm = 0: (A,B,C)->(B,C,A);  perm = 1: (A,B,C)->(C,A,B);  perm = 2: (A,B)->(B,A)
 * (C = 1). */
int gss_layout_permute_f64(gss_ctx *ctx, const double *src_dev, int64_t A,
                           int64_t B, int64_t C, int perm, double *dst_dev);

/* ---- fused per-utterance pipeline --------------------------------------- */
typedef struct {
    int stft_size;            /* 1024 */
    int stft_shift;           /* 256  */
    int stft_fading;          /* 1    */
    int wpe;                  /* 1 = run WPE                                   */
    int wpe_taps;             /* 10   */
    int wpe_delay;            /* 2    */
    int wpe_iterations;       /* 3    */
    int bss_iterations;       /* 20   */
    int bss_iterations_post;  /* 1    */
    int bf_drop_context;      /* 1    */
    int bf;                   /* 0 = 'mvdrSouden_ban', 1 = 'ch2', 2 = 'sum',   */
                              /* 3 = 'gev_ban' (not in the reference's dispatch) */
    int postfilter;           /* 0 = None, 1 = 'mask_mul'                      */
    int wpe_psd_context;      /* 0: frames either side averaged into the WPE power */
} gss_params;

/* Optional taps into the pipeline's intermediates (device pointers; any may be
 * NULL).  This is the `debug=True` contract of the reference blocks
 * (core.py:85-86,210-212,275-276,568-569). */
typedef struct {
    gss_cplx *Obs_ftd;        /* (F,T,D) after WPE                             */
    uint8_t *act_frames;      /* (K,T), the first T frames of the activity     */
    double *gamma;            /* (F,K,T) posteriors before context zeroing     */
    double *target_mask;      /* (F,T)                                         */
    double *distortion_mask;  /* (F,T)                                         */
    gss_cplx *Xhat;           /* (T,F)                                         */
    int32_t *ref_channel;     /* (1,)                                          */
} gss_debug_taps;

/* A0  Enhancer.enhance_observation (core.py:514-571), all intermediates kept in
 * HBM.  obs (D,N) double, act (K,N_act) uint8 in dict order with N_act >= N
 * samples (the reference slices the activity of the reference array, which may be
 * longer than the common length the arrays were cut to, and uses its first T
 * frames: core.py:177-184), target_index = position of speaker_id among the
 * activity keys; start/end_context_samples as computed by
 * start_end_context_frames (core.py:217-222).  out receives
 * gss_istft_num_samples(T,...) samples. */
int gss_enhance_observation(gss_ctx *ctx, const gss_params *params,
                            const double *obs_dev, int D, int64_t N,
                            const uint8_t *act_dev, int K, int64_t N_act,
                            int target_index,
                            int64_t start_context_samples,
                            int64_t end_context_samples,
                            double *out_dev, const gss_debug_taps *taps);

/* Same pipeline fed with the 16-bit PCM samples as they sit in the WAV files: the
 * conversion of the reference's loader, float64(sample) / 2^15 (io/audioread.py:34-226 via
 * soundfile), happens inside the STFT kernel -- bit-identical, a quarter of the H2D bytes and
 * no float64 copy of the recording on the host. */
int gss_enhance_observation_pcm16(gss_ctx *ctx, const gss_params *params,
                                  const int16_t *obs_dev, int D, int64_t N,
                                  const uint8_t *act_dev, int K, int64_t N_act,
                                  int target_index,
                                  int64_t start_context_samples,
                                  int64_t end_context_samples,
                                  double *out_dev, const gss_debug_taps *taps);

/* Same, with host buffers: copies in, runs, copies out, synchronises. */
int gss_enhance_observation_host(gss_ctx *ctx, const gss_params *params,
                                 const double *obs_host, int D, int64_t N,
                                 const uint8_t *act_host, int K, int64_t N_act,
                                 int target_index,
                                 int64_t start_context_samples,
                                 int64_t end_context_samples,
                                 double *out_host);

/* Bytes of context workspace the last call needed (diagnostics / sizing). */
size_t gss_workspace_bytes(gss_ctx *ctx);

/* Device self-test of the f64 MFMA fragment layout the WPE kernel relies on;
 * returns 0 when the layout matches. */
int gss_selftest_mfma(gss_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif /* GSS_HIP_H */
