// Internal declarations shared by the HIP translation units of libgss_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <map>
#include <string>
#include <vector>

#include "../../include/gss_hip.h"

typedef double2 cplx;  // .x = re, .y = im; bit-compatible with gss_cplx

// Kernel-variant switches for tests and A/B runs: ONE environment variable,
//     GSS_VARIANT="key=value,key,..."          (a bare key means 1)
// parsed when its text changes (tests flip it between calls), read through gss_variant().
// The keys that exist are listed in INTEGRATION.md; an unknown key is an error at the first
// library call that looks at the string (a typo must not silently run the default).
int gss_variant(const char *key, int dflt);
inline bool gss_variant_set(const char *key) { return gss_variant(key, 0) != 0; }

// Experiment builds (tools/build_variant.sh: trace instrumentation, other compile-time
// constants) must say so: none of these may leak into the library the package ships.
#if (defined(GSS_CORR_TRACE) || defined(GSS_WCOV_TRACE) || defined(GSS_EM4_TRACE) || \
     defined(GSS_CHOL_TRACE)) && !defined(GSS_EXPERIMENT_BUILD)
#error "trace instrumentation needs -DGSS_EXPERIMENT_BUILD=1 (tools/build_variant.sh)"
#endif

#define GSS_TINY 2.2250738585072014e-308  // np.finfo(np.float64).tiny

// ---------------------------------------------------------------- device math
__device__ __forceinline__ cplx c_make(double r, double i) { return make_double2(r, i); }
__device__ __forceinline__ cplx c_add(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx c_sub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx c_mul(cplx a, cplx b) {
    return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
// a * conj(b)
__device__ __forceinline__ cplx c_mulc(cplx a, cplx b) {
    return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y);
}
// conj(a) * b
__device__ __forceinline__ cplx c_cmul(cplx a, cplx b) {
    return make_double2(a.x * b.x + a.y * b.y, a.x * b.y - a.y * b.x);
}
__device__ __forceinline__ cplx c_scale(cplx a, double s) { return make_double2(a.x * s, a.y * s); }
__device__ __forceinline__ cplx c_conj(cplx a) { return make_double2(a.x, -a.y); }
__device__ __forceinline__ double c_abs2(cplx a) { return a.x * a.x + a.y * a.y; }
// acc += a * b
__device__ __forceinline__ void c_fma(cplx &acc, cplx a, cplx b) {
    acc.x = fma(a.x, b.x, acc.x);
    acc.x = fma(-a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y);
    acc.y = fma(a.y, b.x, acc.y);
}
// acc += a * conj(b)
__device__ __forceinline__ void c_fmac(cplx &acc, cplx a, cplx b) {
    acc.x = fma(a.x, b.x, acc.x);
    acc.x = fma(a.y, b.y, acc.x);
    acc.y = fma(a.y, b.x, acc.y);
    acc.y = fma(-a.x, b.y, acc.y);
}
// acc += conj(a) * b
__device__ __forceinline__ void c_cfma(cplx &acc, cplx a, cplx b) {
    acc.x = fma(a.x, b.x, acc.x);
    acc.x = fma(a.y, b.y, acc.x);
    acc.y = fma(a.x, b.y, acc.y);
    acc.y = fma(-a.y, b.x, acc.y);
}

// Barrier for code that is run by ONE wavefront on data in LDS (the lanes of a wave issue
// their LDS operations in order, so only the compiler has to be kept from reordering).
// In a 64-thread workgroup it is equivalent to __syncthreads(); inside a larger
// workgroup it lets single waves work independently.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Sum over the 64 lanes (all active) on the DPP network (row shifts, then row broadcasts; the
// classic GCN reduction) instead of six ds_bpermute round trips per value: 7 VALU steps, no
// LDS crossbar.  The total forms in lane 63 and is handed to every lane through an SGPR.
// It matters where many values are reduced back to back (register-form M-step: 80 sums per
// wave, a third of the kernel with the butterfly).
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_shifted(double v) {
    const unsigned long long u = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)u, CTRL, ROW_MASK, BANK_MASK, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(u >> 32), CTRL, ROW_MASK, BANK_MASK, true);
    return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ __forceinline__ double wave_sum(double v) {
    double s = v + dpp_shifted<0x111, 0xf, 0xf>(v);      // row_shr:1
    s += dpp_shifted<0x112, 0xf, 0xf>(v);                // row_shr:2
    s += dpp_shifted<0x113, 0xf, 0xf>(v);                // row_shr:3
    s += dpp_shifted<0x114, 0xf, 0xe>(s);                // row_shr:4, banks 1-3
    s += dpp_shifted<0x118, 0xf, 0xc>(s);                // row_shr:8, banks 2-3
    s += dpp_shifted<0x142, 0xa, 0xf>(s);                // row_bcast:15 into rows 1, 3
    s += dpp_shifted<0x143, 0xc, 0xf>(s);                // row_bcast:31 into rows 2, 3
    const unsigned long long u = __double_as_longlong(s);
    const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, 63);
    const unsigned hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), 63);
    return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware block mapping for kernels that launch `nsub` workgroups per frequency:
// the dispatcher is observed to place block b on XCD b % 8 (each XCD has its own
// L2), so the nsub workgroups of one frequency are given linear ids that are equal
// mod 8 and share that frequency's slab through one L2.  1-D grid of
// xcd_grid(nsub, F) blocks; returns false for the padding blocks.  Placement only
// affects speed, never results.
__device__ __forceinline__ bool xcd_group_map(int nsub, int F, int &f, int &sub) {
    const int L = blockIdx.x;
    const int sg = L / (8 * nsub), rem = L - sg * 8 * nsub;
    sub = rem >> 3;
    f = sg * 8 + (rem & 7);
    return f < F;
}
static inline unsigned xcd_grid(int nsub, int F) { return (unsigned)(nsub * ((F + 7) / 8 * 8)); }

// Upper-triangular packed index of (d1 <= d2) in a D x D Hermitian matrix.
__host__ __device__ __forceinline__ int tri_index(int d1, int d2, int D) {
    return d1 * D - (d1 * (d1 - 1)) / 2 + (d2 - d1);
}
__host__ __device__ __forceinline__ int tri_count(int D) { return D * (D + 1) / 2; }

// ---------------------------------------------------------------- context
struct ProfEntry {
    std::string name;
    hipEvent_t start, stop;
};

struct gss_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // second stream of the frequency-blocked EM (cacgmm_run: two blocks of frequencies in
    // flight), forked from / joined to `stream` by events; created on first use
    hipStream_t aux_stream = nullptr;
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    int utterances_in_flight = 0;   // gss_set_utterances_in_flight(): exactly 1 = a call may use aux_stream
    std::string error;

    // bump arena for intermediates of one top-level call
    char *arena = nullptr;
    size_t arena_size = 0;
    size_t arena_off = 0;
    size_t arena_peak = 0;

    // STFT tables
    int stft_size = 0, stft_shift = 0;
    double *win_analysis = nullptr;   // device, stft_size
    double *win_synthesis = nullptr;  // device, stft_size
    cplx *twiddle = nullptr;          // device, stft_size/2: exp(-2 pi i j / size)

    // Status words (mapped host memory).  [0]: the last beamformed utterance, written by
    // mvdr_apply_kernel: the reference channel, -1 = non-finite SNR; INT32_MIN = none yet.
    // [2]: pivots zeroed by the last WPE call (copied from the device counter by wpe_run).
    int32_t *status_host = nullptr;
    int32_t *status_dev = nullptr;

    // WPE tile lists (device), rebuilt when (taps, delay, D) changes
    void *wpe_tiles = nullptr;
    int wpe_tiles_key[4] = {-1, -1, -1, -1};   // taps, delay, D, correlation tile size

    // profiling
    bool profiling = false;
    std::vector<ProfEntry> prof_pending;
    std::vector<hipEvent_t> event_pool;
    std::map<std::string, std::pair<long, double>> prof_acc;
    std::string prof_filter;   // time only this kernel (empty: all)
};

int gss_fail(gss_ctx *ctx, int code, const char *fmt, ...);

#define GSS_HIP_CHECK(ctx, expr)                                                   \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess)                                                      \
            return gss_fail((ctx), GSS_ERR_HIP, "%s failed: %s (%s:%d)", #expr,    \
                            hipGetErrorString(_e), __FILE__, __LINE__);            \
    } while (0)

#define GSS_REQUIRE(ctx, cond, code, ...)                      \
    do {                                                       \
        if (!(cond)) return gss_fail((ctx), (code), __VA_ARGS__); \
    } while (0)

#define GSS_TRY(expr)              \
    do {                           \
        int _s = (expr);           \
        if (_s != GSS_OK) return _s; \
    } while (0)

// Arena: reserve() makes sure `bytes` are available for the coming top-level call
// (may synchronise + reallocate); alloc() bumps.  reset() starts a new call.
int arena_reserve(gss_ctx *ctx, size_t bytes);
void arena_reset(gss_ctx *ctx);
void *arena_alloc(gss_ctx *ctx, size_t bytes);
template <typename T>
static inline T *arena_alloc_t(gss_ctx *ctx, size_t count) {
    return reinterpret_cast<T *>(arena_alloc(ctx, count * sizeof(T)));
}
static inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

// Profiling scope: records HIP events around a launch when enabled.
struct ProfScope {
    gss_ctx *ctx;
    ProfEntry e;
    bool active;
    ProfScope(gss_ctx *c, const char *name);
    ~ProfScope();
};
#define GSS_PROF(ctx, name) ProfScope _prof_scope_##__LINE__((ctx), (name))

// Post-launch error check.
#define GSS_LAUNCH_CHECK(ctx, name)                                               \
    do {                                                                          \
        hipError_t _e = hipGetLastError();                                        \
        if (_e != hipSuccess)                                                     \
            return gss_fail((ctx), GSS_ERR_HIP, "launch of %s failed: %s", (name), \
                            hipGetErrorString(_e));                               \
    } while (0)

// ---------------------------------------------------------------- stage launchers
// (device pointers, workspace from the arena, asynchronous on ctx->stream)
size_t wpe_workspace_bytes(int F, int64_t T, int D, int taps, int delay);
int wpe_inverse_power_run(gss_ctx *ctx, const cplx *Y, int F, int64_t T, int D, int psd_context,
                          double *w);
// part: -1 = the whole call; 0 / 1 = one of two sets of frequencies that run side by side on
// ctx->stream / ctx->aux_stream (gss_enhance_observation, GSS_VARIANT wpe_halves): part 0 zeroes
// the pivot counter and records ctx->ev_fork behind it, neither part copies the count to the
// host, each has its own correlation work queues.
int wpe_run(gss_ctx *ctx, const cplx *Y, int F, int64_t T, int D, int taps, int delay,
            int iterations, int psd_context, cplx *X, int part = -1);
// the pivot count of the last wpe_run parts -> the context's status word (after the join)
int wpe_copy_zero_pivots(gss_ctx *ctx);
// second stream + fork / join events of a context, created on first use
int aux_stream_ready(gss_ctx *ctx);

size_t cacgmm_workspace_bytes(int F, int64_t T, int D, int K);
int cacgmm_run(gss_ctx *ctx, const cplx *Y, int F, int64_t T, int D, const uint8_t *act,
               int64_t act_stride, int K, int iterations, int iterations_post, double *gamma);

int psd_partials_run(gss_ctx *ctx, const cplx *Y, int F, int64_t T, int D, const double *W2,
                     int nch, int chunk_frames, cplx *part);

size_t mvdr_workspace_bytes(int F, int64_t T, int D);
int mvdr_run(gss_ctx *ctx, const cplx *Y, int F, int64_t T, int D, const double *mx,
             const double *mn, int ban, cplx *Xhat, int32_t *ref_channel, int gev = 0,
             int forced_ref = -1);
int masks_from_posteriors_run(gss_ctx *ctx, const double *gamma, int F, int K, int64_t T,
                              int target, int drop, int64_t start_frames,
                              int64_t end_frames, double *mx, double *mn);

size_t stft_workspace_bytes(int64_t T, int size);
int stft_run(gss_ctx *ctx, const void *x, int in_type, int D, int64_t N, int fading, cplx *Y);
int istft_run(gss_ctx *ctx, const cplx *X, int64_t T, int fading, double *x);
int activity_run(gss_ctx *ctx, const uint8_t *act, int K, int64_t N, int fading,
                 uint8_t *out);
int channel_pick_run(gss_ctx *ctx, const cplx *Y, int F, int64_t T, int D, int mode,
                     cplx *Xhat);  // mode 1: 'ch2', 2: 'sum'
int mask_mul_run(gss_ctx *ctx, cplx *Xhat, const double *mask_ft, int F, int64_t T);

int selftest_mfma_run(gss_ctx *ctx);

// Shared device routine: cyclic-Jacobi eigendecomposition of Hermitian matrices
// held in LDS (see jacobi.h).
