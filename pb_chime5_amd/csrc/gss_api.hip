// Context, memory plumbing, profiling and the fused per-utterance pipeline of
// libgss_hip.so (C ABI declared in include/gss_hip.h).
#include <cmath>
#include <cstdarg>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "gss_internal.h"

// Every entry point that touches the device makes the context's GPU current first:
// one process (and thread) may interleave contexts of different GPUs.
#define GSS_ENTER(ctx)                                   \
    if (!(ctx)) return GSS_ERR_INVALID;                  \
    GSS_HIP_CHECK((ctx), hipSetDevice((ctx)->device))
// entry points whose kernels have GSS_VARIANT switches (WPE, EM, the fused pipeline)
#define GSS_ENTER_VARIANTS(ctx) \
    GSS_ENTER(ctx);             \
    GSS_TRY(variant_refresh(ctx))
static int variant_refresh(gss_ctx *ctx);

// ------------------------------------------------------------------ errors
// (gss_host_malloc / gss_host_free are documented as callable from any thread -- the session
// driver's loader threads grow their staging blocks on the first context while its owner
// enqueues: the message of a context is written and read under one lock, and
// gss_last_error() hands out the calling thread's own copy)
static std::mutex g_error_lock;

int gss_fail(gss_ctx *ctx, int code, const char *fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) {
        std::lock_guard<std::mutex> guard(g_error_lock);
        ctx->error = buf;
    }
    return code;
}

static thread_local std::string g_create_error;

// ------------------------------------------------------------------ GSS_VARIANT
namespace {
const char *const kVariantKeys[] = {
    // wpe.hip
    "corr_ts", "corr_nw", "corr_blocked", "corr_p_tiles", "chol_diag_unfolded", "apply_ph",
    "apply_generic", "prof_detail", "corr_stg8", "corr_ksplit", "apply_gglobal", "wpe_halves",
    // cacgmm.hip
    "em_wgs", "estep_wpb", "estep_lds", "mstep_prefetch_d", "mstep_tiled", "mstep_plan_min_d",
    "mstep_chunked", "mstep_slots", "force_eigh", "em_unfused", "em_l3_mb", "em_l3_fit_mb",
    "em_streams", "mstep_maxseg", "em4_cold_eigh", "mstep_generic"};
struct VariantTable {
    std::mutex lock;
    std::string text;
    bool parsed = false;
    std::map<std::string, int> values;
};
VariantTable &variant_table() {
    static VariantTable t;
    return t;
}
}   // namespace

// Re-read GSS_VARIANT (once per top-level call that launches variant-switched kernels -- the
// kernels' host code below only looks at the parsed table).  An unknown key is an error of that
// call, not an abort() of the host process.  The text may only be changed while no library call
// is in flight: getenv() beside a concurrent setenv() is undefined behaviour in glibc.
static int variant_refresh(gss_ctx *ctx) {
    const char *env = getenv("GSS_VARIANT");
    VariantTable &t = variant_table();
    std::lock_guard<std::mutex> guard(t.lock);
    if (t.parsed && t.text == (env ? env : "")) return GSS_OK;
    t.text = env ? env : "";
    t.parsed = true;
    t.values.clear();
    std::string unknown;
    size_t pos = 0;
    while (pos < t.text.size()) {
        size_t end = t.text.find_first_of(", ", pos);
        if (end == std::string::npos) end = t.text.size();
        const std::string tok = t.text.substr(pos, end - pos);
        pos = end + 1;
        if (tok.empty()) continue;
        const size_t eq = tok.find('=');
        const std::string name = tok.substr(0, eq);
        bool known = false;
        for (const char *k : kVariantKeys) known = known || name == k;
        if (!known) {
            unknown = name;
            continue;
        }
        t.values[name] = eq == std::string::npos ? 1 : atoi(tok.c_str() + eq + 1);
    }
    if (!unknown.empty()) {
        t.values.clear();
        t.parsed = false;       // the next call reports it again
        return gss_fail(ctx, GSS_ERR_INVALID, "GSS_VARIANT names no switch '%s'", unknown.c_str());
    }
    return GSS_OK;
}

int gss_variant(const char *key, int dflt) {
    VariantTable &t = variant_table();
    std::lock_guard<std::mutex> guard(t.lock);
    const auto it = t.values.find(key);
    return it == t.values.end() ? dflt : it->second;
}

extern "C" const char *gss_last_error(gss_ctx *ctx) {
    if (!ctx) return g_create_error.c_str();
    static thread_local std::string copy;
    std::lock_guard<std::mutex> guard(g_error_lock);
    copy = ctx->error;
    return copy.c_str();
}

extern "C" const char *gss_version(void) {
#ifdef GSS_EXPERIMENT_BUILD
    return "pb_chime5_amd/libgss_hip 0.4 EXPERIMENT BUILD (gfx950, f64)";
#else
    return "pb_chime5_amd/libgss_hip 0.4 (gfx950, f64)";
#endif
}

extern "C" int gss_abi_version(void) { return GSS_ABI_VERSION; }

// ------------------------------------------------------------------ context
extern "C" int gss_device_count(void) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess) return 0;
    return count;
}

extern "C" int gss_device_pci_bus_id(int device_id, char *buf, int len) {
    if (!buf || len < 16) return GSS_ERR_INVALID;
    buf[0] = 0;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device_id < 0 || device_id >= count)
        return GSS_ERR_INVALID;
    return hipDeviceGetPCIBusId(buf, len, device_id) == hipSuccess ? GSS_OK : GSS_ERR_HIP;
}

extern "C" int gss_create(int device_id, gss_ctx **out) {
    if (!out) return GSS_ERR_INVALID;
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0) {
        g_create_error = "no HIP device available: ";
        g_create_error += hipGetErrorString(e);
        return GSS_ERR_HIP;
    }
    if (device_id < 0 || device_id >= count) {
        g_create_error = "device_id out of range";
        return GSS_ERR_INVALID;
    }
    e = hipSetDevice(device_id);
    if (e != hipSuccess) {
        g_create_error = hipGetErrorString(e);
        return GSS_ERR_HIP;
    }
    gss_ctx *ctx = new gss_ctx();
    ctx->device = device_id;
    e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        g_create_error = hipGetErrorString(e);
        delete ctx;
        return GSS_ERR_HIP;
    }
    ctx->stream = ctx->own_stream;
    e = hipHostMalloc(reinterpret_cast<void **>(&ctx->status_host), 64, hipHostMallocMapped);
    if (e == hipSuccess)
        e = hipHostGetDevicePointer(reinterpret_cast<void **>(&ctx->status_dev), ctx->status_host, 0);
    if (e != hipSuccess) {
        g_create_error = hipGetErrorString(e);
        (void)hipStreamDestroy(ctx->own_stream);
        delete ctx;
        return GSS_ERR_HIP;
    }
    ctx->status_host[0] = INT32_MIN;
    ctx->status_host[2] = 0;
    *out = ctx;
    return GSS_OK;
}

static void free_tables(gss_ctx *ctx) {
    if (ctx->win_analysis) (void)hipFree(ctx->win_analysis);
    if (ctx->win_synthesis) (void)hipFree(ctx->win_synthesis);
    if (ctx->twiddle) (void)hipFree(ctx->twiddle);
    ctx->win_analysis = ctx->win_synthesis = nullptr;
    ctx->twiddle = nullptr;
}

extern "C" int gss_destroy(gss_ctx *ctx) {
    if (!ctx) return GSS_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    free_tables(ctx);
    if (ctx->arena) (void)hipFree(ctx->arena);
    if (ctx->wpe_tiles) (void)hipFree(ctx->wpe_tiles);
    if (ctx->status_host) (void)hipHostFree(ctx->status_host);
    for (auto &p : ctx->prof_pending) {
        (void)hipEventDestroy(p.start);
        (void)hipEventDestroy(p.stop);
    }
    for (auto ev : ctx->event_pool) (void)hipEventDestroy(ev);
    if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
    if (ctx->aux_stream) {
        (void)hipStreamSynchronize(ctx->aux_stream);
        (void)hipStreamDestroy(ctx->aux_stream);
    }
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return GSS_OK;
}

int aux_stream_ready(gss_ctx *ctx) {
    if (!ctx->aux_stream)
        GSS_HIP_CHECK(ctx, hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
    if (!ctx->ev_fork) GSS_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    if (!ctx->ev_join) GSS_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    return GSS_OK;
}

extern "C" int gss_set_stream(gss_ctx *ctx, void *hip_stream) {
    GSS_ENTER(ctx);
    GSS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    ctx->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx->own_stream;
    return GSS_OK;
}

extern "C" int gss_set_utterances_in_flight(gss_ctx *ctx, int n) {
    if (!ctx) return GSS_ERR_INVALID;
    GSS_REQUIRE(ctx, n >= 0, GSS_ERR_INVALID, "gss_set_utterances_in_flight: n=%d", n);
    ctx->utterances_in_flight = n;
    return GSS_OK;
}

extern "C" int gss_synchronize(gss_ctx *ctx) {
    if (!ctx) return GSS_ERR_INVALID;
    GSS_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    GSS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return GSS_OK;
}

// ------------------------------------------------------------------ memory
extern "C" int gss_dev_malloc(gss_ctx *ctx, size_t bytes, void **dev_ptr) {
    if (!ctx || !dev_ptr) return GSS_ERR_INVALID;
    GSS_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    *dev_ptr = nullptr;
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(dev_ptr, bytes);
    if (e != hipSuccess)
        return gss_fail(ctx, GSS_ERR_NOMEM, "hipMalloc(%zu) failed: %s", bytes,
                        hipGetErrorString(e));
    return GSS_OK;
}

extern "C" int gss_dev_free(gss_ctx *ctx, void *dev_ptr) {
    if (!ctx) return GSS_ERR_INVALID;
    if (!dev_ptr) return GSS_OK;
    GSS_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    GSS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    GSS_HIP_CHECK(ctx, hipFree(dev_ptr));
    return GSS_OK;
}

extern "C" int gss_memcpy_h2d(gss_ctx *ctx, void *dst, const void *src, size_t bytes) {
    GSS_ENTER(ctx);
    if (bytes == 0) return GSS_OK;
    GSS_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    // pageable host memory: the copy is staged before the call returns, but be
    // explicit so the caller may reuse `src` immediately.
    GSS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return GSS_OK;
}

extern "C" int gss_memcpy_d2h(gss_ctx *ctx, void *dst, const void *src, size_t bytes) {
    GSS_ENTER(ctx);
    if (bytes == 0) return GSS_OK;
    GSS_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    GSS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return GSS_OK;
}

// Page-locked host memory + copies that do not wait: the session driver's staging buffers.
// WAV samples are read straight into a pinned (D, N) int16 block, one DMA brings it to the
// device while the host thread goes on to enqueue the kernels behind it.
extern "C" int gss_host_malloc(gss_ctx *ctx, size_t bytes, void **host_ptr) {
    if (!ctx || !host_ptr) return GSS_ERR_INVALID;
    GSS_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    *host_ptr = nullptr;
    if (bytes == 0) bytes = 16;
    hipError_t e = hipHostMalloc(host_ptr, bytes, hipHostMallocDefault);
    if (e != hipSuccess)
        return gss_fail(ctx, GSS_ERR_NOMEM, "hipHostMalloc(%zu) failed: %s", bytes,
                        hipGetErrorString(e));
    return GSS_OK;
}

extern "C" int gss_host_free(gss_ctx *ctx, void *host_ptr) {
    if (!ctx) return GSS_ERR_INVALID;
    if (!host_ptr) return GSS_OK;
    // no stream synchronisation of our own (unlike gss_dev_free); the caller guarantees that
    // no copy from / to the block is still in flight.  hipHostFree itself waits for the whole
    // device, which is why the session driver only frees at the end of a session
    GSS_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    GSS_HIP_CHECK(ctx, hipHostFree(host_ptr));
    return GSS_OK;
}

extern "C" int gss_memcpy_h2d_async(gss_ctx *ctx, void *dst, const void *src, size_t bytes) {
    GSS_ENTER(ctx);
    if (bytes == 0) return GSS_OK;
    GSS_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return GSS_OK;
}

extern "C" int gss_memcpy_d2h_async(gss_ctx *ctx, void *dst, const void *src, size_t bytes) {
    GSS_ENTER(ctx);
    if (bytes == 0) return GSS_OK;
    GSS_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    return GSS_OK;
}

extern "C" int gss_memset(gss_ctx *ctx, void *dst, int value, size_t bytes) {
    GSS_ENTER(ctx);
    if (bytes == 0) return GSS_OK;
    GSS_HIP_CHECK(ctx, hipMemsetAsync(dst, value, bytes, ctx->stream));
    return GSS_OK;
}

// ------------------------------------------------------------------ arena
void arena_reset(gss_ctx *ctx) { ctx->arena_off = 0; }

int arena_reserve(gss_ctx *ctx, size_t bytes) {
    bytes = align_up(bytes + 4096, 1 << 20);
    ctx->arena_off = 0;
    if (bytes <= ctx->arena_size) return GSS_OK;
    GSS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->arena) {
        GSS_HIP_CHECK(ctx, hipFree(ctx->arena));
        ctx->arena = nullptr;
        ctx->arena_size = 0;
    }
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&ctx->arena), bytes);
    if (e != hipSuccess)
        return gss_fail(ctx, GSS_ERR_NOMEM, "workspace hipMalloc(%zu) failed: %s", bytes,
                        hipGetErrorString(e));
    ctx->arena_size = bytes;
    return GSS_OK;
}

void *arena_alloc(gss_ctx *ctx, size_t bytes) {
    size_t off = align_up(ctx->arena_off, 256);
    if (off + bytes > ctx->arena_size) return nullptr;  // reserve() was too small: bug
    ctx->arena_off = off + bytes;
    if (ctx->arena_off > ctx->arena_peak) ctx->arena_peak = ctx->arena_off;
    return ctx->arena + off;
}

extern "C" size_t gss_workspace_bytes(gss_ctx *ctx) { return ctx ? ctx->arena_peak : 0; }

// ------------------------------------------------------------------ profiling
ProfScope::ProfScope(gss_ctx *c, const char *name) : ctx(c), active(c->profiling) {
    if (active && !c->prof_filter.empty() && c->prof_filter != name) active = false;
    if (!active) return;
    e.name = name;
    auto get = [&]() {
        hipEvent_t ev;
        if (!ctx->event_pool.empty()) {
            ev = ctx->event_pool.back();
            ctx->event_pool.pop_back();
        } else {
            (void)hipEventCreate(&ev);
        }
        return ev;
    };
    e.start = get();
    e.stop = get();
    (void)hipEventRecord(e.start, ctx->stream);
}

ProfScope::~ProfScope() {
    if (!active) return;
    (void)hipEventRecord(e.stop, ctx->stream);
    ctx->prof_pending.push_back(e);
}

static void prof_drain(gss_ctx *ctx) {
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &p : ctx->prof_pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.start, p.stop) == hipSuccess) {
            auto &acc = ctx->prof_acc[p.name];
            acc.first += 1;
            acc.second += ms;
        }
        ctx->event_pool.push_back(p.start);
        ctx->event_pool.push_back(p.stop);
    }
    ctx->prof_pending.clear();
}

extern "C" int gss_profile_enable(gss_ctx *ctx, int on) {
    if (!ctx) return GSS_ERR_INVALID;
    ctx->profiling = on != 0;
    return GSS_OK;
}

extern "C" int gss_profile_filter(gss_ctx *ctx, const char *kernel) {
    if (!ctx) return GSS_ERR_INVALID;
    ctx->prof_filter = kernel ? kernel : "";
    return GSS_OK;
}

extern "C" int gss_profile_reset(gss_ctx *ctx) {
    if (!ctx) return GSS_ERR_INVALID;
    prof_drain(ctx);
    ctx->prof_acc.clear();
    return GSS_OK;
}

extern "C" int gss_profile_report(gss_ctx *ctx, char *buf, size_t buf_size) {
    if (!ctx || !buf || buf_size < 4) return GSS_ERR_INVALID;
    prof_drain(ctx);
    std::string s = "{";
    bool first = true;
    for (auto &kv : ctx->prof_acc) {
        char item[256];
        snprintf(item, sizeof(item), "%s\"%s\": {\"calls\": %ld, \"ms\": %.6f}",
                 first ? "" : ", ", kv.first.c_str(), kv.second.first, kv.second.second);
        s += item;
        first = false;
    }
    s += "}";
    if (s.size() + 1 > buf_size) return gss_fail(ctx, GSS_ERR_INVALID, "report buffer too small");
    memcpy(buf, s.c_str(), s.size() + 1);
    return GSS_OK;
}

// ------------------------------------------------------------------ geometry
extern "C" int64_t gss_stft_num_frames(int64_t n, int size, int shift, int fading) {
    if (fading) n += 2 * (int64_t)(size - shift);
    if (n < size) return 1;
    return (n - size + shift - 1) / shift + 1;
}

extern "C" int64_t gss_istft_num_samples(int64_t T, int size, int shift, int fading) {
    int64_t n = T * shift + size - shift;
    if (fading) n -= 2 * (int64_t)(size - shift);
    return n < 0 ? 0 : n;
}

extern "C" int64_t gss_samples_to_stft_frames(int64_t samples, int size, int shift,
                                              int fading) {
    if (fading) samples += 2 * (int64_t)(size - shift);
    // ceil((samples - size + shift) / shift), also for negative numerators
    int64_t num = samples - size + shift;
    int64_t q = num / shift;
    if (num % shift != 0 && num > 0) q += 1;
    return q;
}

extern "C" int gss_set_windows(gss_ctx *ctx, int size, int shift, const double *analysis,
                               const double *synthesis) {
    if (!ctx || !analysis || !synthesis) return GSS_ERR_INVALID;
    // (powers of two take the radix-2 kernels, other even lengths a direct DFT)
    GSS_REQUIRE(ctx, size >= 4 && size <= GSS_MAX_STFT_SIZE && size % 2 == 0,
                GSS_ERR_UNSUPPORTED, "stft size %d: need an even length in [4, %d]", size,
                GSS_MAX_STFT_SIZE);
    GSS_REQUIRE(ctx, shift > 0 && shift <= size && size % shift == 0, GSS_ERR_INVALID,
                "stft shift %d must divide size %d", shift, size);
    GSS_HIP_CHECK(ctx, hipSetDevice(ctx->device));
    GSS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    free_tables(ctx);
    size_t wb = sizeof(double) * size;
    GSS_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->win_analysis), wb));
    GSS_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->win_synthesis), wb));
    GSS_HIP_CHECK(ctx, hipMalloc(reinterpret_cast<void **>(&ctx->twiddle),
                                 sizeof(cplx) * (size / 2)));
    std::vector<cplx> tw(size / 2);
    for (int j = 0; j < size / 2; ++j) {
        // exact at the quadrant points, cos/sin of a reduced argument elsewhere
        double a = -2.0 * M_PI * (double)j / (double)size;
        tw[j].x = cos(a);
        tw[j].y = sin(a);
    }
    tw[0].x = 1.0;
    tw[0].y = 0.0;
    if (size % 4 == 0) {
        tw[size / 4].x = 0.0;
        tw[size / 4].y = -1.0;
    }
    GSS_HIP_CHECK(ctx, hipMemcpy(ctx->win_analysis, analysis, wb, hipMemcpyHostToDevice));
    GSS_HIP_CHECK(ctx, hipMemcpy(ctx->win_synthesis, synthesis, wb, hipMemcpyHostToDevice));
    GSS_HIP_CHECK(ctx, hipMemcpy(ctx->twiddle, tw.data(), sizeof(cplx) * (size / 2),
                                 hipMemcpyHostToDevice));
    ctx->stft_size = size;
    ctx->stft_shift = shift;
    return GSS_OK;
}

// ------------------------------------------------------------------ stage wrappers

static int check_windows(gss_ctx *ctx) {
    GSS_REQUIRE(ctx, ctx->stft_size > 0, GSS_ERR_INVALID, "call gss_set_windows() first");
    return GSS_OK;
}

extern "C" int gss_stft(gss_ctx *ctx, const double *x, int D, int64_t N, int fading,
                        gss_cplx *Y) {
    GSS_ENTER(ctx);
    GSS_TRY(check_windows(ctx));
    GSS_REQUIRE(ctx, D >= 1 && N >= 0 && x && Y, GSS_ERR_INVALID, "gss_stft: bad arguments");
    return stft_run(ctx, x, 0, D, N, fading, reinterpret_cast<cplx *>(Y));
}

extern "C" int gss_istft(gss_ctx *ctx, const gss_cplx *X, int64_t T, int fading, double *x) {
    GSS_ENTER(ctx);
    GSS_TRY(check_windows(ctx));
    GSS_REQUIRE(ctx, T >= 1 && X && x, GSS_ERR_INVALID, "gss_istft: bad arguments");
    GSS_TRY(arena_reserve(ctx, stft_workspace_bytes(T, ctx->stft_size)));
    return istft_run(ctx, reinterpret_cast<const cplx *>(X), T, fading, x);
}

extern "C" int gss_activity_time_to_frequency(gss_ctx *ctx, const uint8_t *act, int K,
                                              int64_t N, int fading, uint8_t *out) {
    GSS_ENTER(ctx);
    GSS_TRY(check_windows(ctx));
    GSS_REQUIRE(ctx, K >= 1 && N >= 0 && act && out, GSS_ERR_INVALID,
                "gss_activity_time_to_frequency: bad arguments");
    return activity_run(ctx, act, K, N, fading, out);
}

extern "C" int gss_wpe(gss_ctx *ctx, const gss_cplx *Y, int F, int64_t T, int D, int taps,
                       int delay, int iterations, int psd_context, gss_cplx *X) {
    GSS_ENTER_VARIANTS(ctx);
    GSS_REQUIRE(ctx, Y && X && F >= 1 && T >= 1, GSS_ERR_INVALID, "gss_wpe: bad arguments");
    GSS_REQUIRE(ctx, D >= 1 && D <= GSS_MAX_CHANNELS, GSS_ERR_UNSUPPORTED,
                "gss_wpe: D=%d outside [1, %d]", D, GSS_MAX_CHANNELS);
    GSS_REQUIRE(ctx, taps >= 1 && delay >= 0 && iterations >= 0 && psd_context >= 0,
                GSS_ERR_INVALID, "gss_wpe: taps=%d delay=%d iterations=%d psd_context=%d", taps,
                delay, iterations, psd_context);
    GSS_REQUIRE(ctx, (int64_t)F * T * D < (1LL << 31), GSS_ERR_UNSUPPORTED,
                "F * T * D = %lld STFT bins: 2^31 or more are not supported (some kernels index "
                "the tensor with 32 bits)", (long long)((int64_t)F * T * D));
    GSS_TRY(arena_reserve(ctx, wpe_workspace_bytes(F, T, D, taps, delay)));
    return wpe_run(ctx, reinterpret_cast<const cplx *>(Y), F, T, D, taps, delay, iterations,
                   psd_context, reinterpret_cast<cplx *>(X));
}

extern "C" int gss_wpe_inverse_power(gss_ctx *ctx, const gss_cplx *Y, int F, int64_t T, int D,
                                     int psd_context, double *inverse_power) {
    GSS_ENTER_VARIANTS(ctx);
    GSS_REQUIRE(ctx, Y && inverse_power && F >= 1 && T >= 1 && psd_context >= 0, GSS_ERR_INVALID,
                "gss_wpe_inverse_power: bad arguments");
    GSS_REQUIRE(ctx, D >= 1 && D <= GSS_MAX_CHANNELS, GSS_ERR_UNSUPPORTED,
                "gss_wpe_inverse_power: D=%d outside [1, %d]", D, GSS_MAX_CHANNELS);
    GSS_REQUIRE(ctx, (int64_t)F * T * D < (1LL << 31), GSS_ERR_UNSUPPORTED,
                "F * T * D = %lld STFT bins: 2^31 or more are not supported",
                (long long)((int64_t)F * T * D));
    GSS_TRY(arena_reserve(ctx, sizeof(double) * (size_t)F * T + 8192));
    return wpe_inverse_power_run(ctx, reinterpret_cast<const cplx *>(Y), F, T, D, psd_context,
                                 inverse_power);
}

static int check_cacgmm_args(gss_ctx *ctx, int D, int K, int iterations, int post) {
    GSS_REQUIRE(ctx, D >= 2 && D <= GSS_MAX_CHANNELS, GSS_ERR_UNSUPPORTED,
                "cacgmm: D=%d outside [2, %d]", D, GSS_MAX_CHANNELS);
    // pb_bss CACGMMTrainer.fit: assert K < 20 (-> AssertionError in the reference too)
    GSS_REQUIRE(ctx, K >= 1 && K <= GSS_MAX_CLASSES, GSS_ERR_INVALID,
                "cacgmm: assert 1 <= K < 20 failed: K=%d", K);
    GSS_REQUIRE(ctx, iterations >= 1 && post >= 0, GSS_ERR_INVALID,
                "cacgmm: iterations=%d iterations_post=%d", iterations, post);
    return GSS_OK;
}

extern "C" int gss_cacgmm(gss_ctx *ctx, const gss_cplx *Y, int F, int64_t T, int D,
                          const uint8_t *act, int K, int iterations, int post,
                          double *gamma) {
    GSS_ENTER_VARIANTS(ctx);
    GSS_REQUIRE(ctx, Y && act && gamma && F >= 1 && T >= 1, GSS_ERR_INVALID,
                "gss_cacgmm: bad arguments");
    GSS_TRY(check_cacgmm_args(ctx, D, K, iterations, post));
    GSS_REQUIRE(ctx, (int64_t)F * T * D < (1LL << 31), GSS_ERR_UNSUPPORTED,
                "F * T * D = %lld STFT bins: 2^31 or more are not supported (some kernels index "
                "the tensor with 32 bits)", (long long)((int64_t)F * T * D));
    GSS_TRY(arena_reserve(ctx, cacgmm_workspace_bytes(F, T, D, K)));
    return cacgmm_run(ctx, reinterpret_cast<const cplx *>(Y), F, T, D, act, T, K, iterations,
                      post, gamma);
}

extern "C" int gss_masks_from_posteriors(gss_ctx *ctx, const double *gamma, int F, int K,
                                         int64_t T, int target, int drop, int64_t sf,
                                         int64_t ef, double *mx, double *mn) {
    GSS_ENTER(ctx);
    GSS_REQUIRE(ctx, gamma && mx && mn && F >= 1 && T >= 1 && K >= 1, GSS_ERR_INVALID,
                "gss_masks_from_posteriors: bad arguments");
    GSS_REQUIRE(ctx, target >= 0 && target < K, GSS_ERR_INVALID,
                "target_index %d outside [0, %d)", target, K);
    return masks_from_posteriors_run(ctx, gamma, F, K, T, target, drop, sf, ef, mx, mn);
}

extern "C" int gss_mvdr_souden(gss_ctx *ctx, const gss_cplx *Y, int F, int64_t T, int D,
                               const double *mx, const double *mn, int ban, gss_cplx *Xhat,
                               int32_t *ref) {
    GSS_ENTER(ctx);
    GSS_REQUIRE(ctx, Y && mx && mn && Xhat && F >= 1 && T >= 1, GSS_ERR_INVALID,
                "gss_mvdr_souden: bad arguments");
    // beamforming_wrapper.py:44: assert D < 30
    GSS_REQUIRE(ctx, D >= 1 && D < 30, GSS_ERR_INVALID, "assert D < 30 failed: D=%d", D);
    GSS_REQUIRE(ctx, (int64_t)F * T * D < (1LL << 31), GSS_ERR_UNSUPPORTED,
                "F * T * D = %lld STFT bins: 2^31 or more are not supported (some kernels index "
                "the tensor with 32 bits)", (long long)((int64_t)F * T * D));
    GSS_TRY(arena_reserve(ctx, mvdr_workspace_bytes(F, T, D)));
    return mvdr_run(ctx, reinterpret_cast<const cplx *>(Y), F, T, D, mx, mn, ban,
                    reinterpret_cast<cplx *>(Xhat), ref);
}

extern "C" int gss_mvdr_souden_ref(gss_ctx *ctx, const gss_cplx *Y, int F, int64_t T, int D,
                                   const double *mx, const double *mn, int ban, int ref_channel,
                                   gss_cplx *Xhat) {
    GSS_ENTER(ctx);
    GSS_REQUIRE(ctx, Y && mx && mn && Xhat && F >= 1 && T >= 1, GSS_ERR_INVALID,
                "gss_mvdr_souden_ref: bad arguments");
    GSS_REQUIRE(ctx, D >= 1 && D < 30, GSS_ERR_INVALID, "assert D < 30 failed: D=%d", D);
    GSS_REQUIRE(ctx, ref_channel >= 0 && ref_channel < D, GSS_ERR_INVALID,
                "ref_channel %d outside [0, %d)", ref_channel, D);
    GSS_REQUIRE(ctx, (int64_t)F * T * D < (1LL << 31), GSS_ERR_UNSUPPORTED,
                "F * T * D = %lld STFT bins: 2^31 or more are not supported (some kernels index "
                "the tensor with 32 bits)", (long long)((int64_t)F * T * D));
    GSS_TRY(arena_reserve(ctx, mvdr_workspace_bytes(F, T, D)));
    return mvdr_run(ctx, reinterpret_cast<const cplx *>(Y), F, T, D, mx, mn, ban,
                    reinterpret_cast<cplx *>(Xhat), nullptr, /*gev=*/0, ref_channel);
}

extern "C" int gss_last_ref_channel(gss_ctx *ctx, int32_t *ref_channel) {
    GSS_ENTER(ctx);
    GSS_REQUIRE(ctx, ref_channel, GSS_ERR_INVALID, "gss_last_ref_channel: NULL");
    GSS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *ref_channel = __atomic_load_n(ctx->status_host, __ATOMIC_ACQUIRE);
    return GSS_OK;
}

extern "C" int gss_last_wpe_zero_pivots(gss_ctx *ctx, int64_t *count) {
    GSS_ENTER(ctx);
    GSS_REQUIRE(ctx, count, GSS_ERR_INVALID, "gss_last_wpe_zero_pivots: NULL");
    GSS_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    *count = __atomic_load_n(ctx->status_host + 2, __ATOMIC_ACQUIRE);
    return GSS_OK;
}

extern "C" int gss_gev(gss_ctx *ctx, const gss_cplx *Y, int F, int64_t T, int D, const double *mx,
                       const double *mn, int ban, gss_cplx *Xhat) {
    GSS_ENTER(ctx);
    GSS_REQUIRE(ctx, Y && mx && mn && Xhat && F >= 1 && T >= 1, GSS_ERR_INVALID,
                "gss_gev: bad arguments");
    GSS_REQUIRE(ctx, D >= 1 && D < 30, GSS_ERR_INVALID, "assert D < 30 failed: D=%d", D);
    GSS_REQUIRE(ctx, (int64_t)F * T * D < (1LL << 31), GSS_ERR_UNSUPPORTED,
                "F * T * D = %lld STFT bins: 2^31 or more are not supported (some kernels index "
                "the tensor with 32 bits)", (long long)((int64_t)F * T * D));
    GSS_TRY(arena_reserve(ctx, mvdr_workspace_bytes(F, T, D)));
    return mvdr_run(ctx, reinterpret_cast<const cplx *>(Y), F, T, D, mx, mn, ban,
                    reinterpret_cast<cplx *>(Xhat), nullptr, /*gev=*/1);
}

extern "C" int gss_selftest_mfma(gss_ctx *ctx) {
    GSS_ENTER(ctx);
    return selftest_mfma_run(ctx);
}

// ------------------------------------------------------------------ fused pipeline
static int check_params(gss_ctx *ctx, const gss_params *p) {
    GSS_REQUIRE(ctx, p, GSS_ERR_INVALID, "params is NULL");
    GSS_REQUIRE(ctx, p->stft_size == ctx->stft_size && p->stft_shift == ctx->stft_shift,
                GSS_ERR_INVALID, "params stft %d/%d differ from gss_set_windows() %d/%d",
                p->stft_size, p->stft_shift, ctx->stft_size, ctx->stft_shift);
    GSS_REQUIRE(ctx, p->bf >= 0 && p->bf <= 3, GSS_ERR_UNSUPPORTED, "bf=%d", p->bf);
    GSS_REQUIRE(ctx, p->postfilter >= 0 && p->postfilter <= 1, GSS_ERR_UNSUPPORTED,
                "postfilter=%d", p->postfilter);
    GSS_REQUIRE(ctx, p->wpe_psd_context >= 0, GSS_ERR_INVALID, "wpe_psd_context=%d",
                p->wpe_psd_context);
    return GSS_OK;
}

static size_t pipeline_workspace(const gss_params *p, int F, int64_t T, int64_t T_act, int D,
                                 int K) {
    size_t b = 0;
    size_t ftd = align_up(sizeof(cplx) * (size_t)F * T * D);
    b += 2 * ftd;                                            // Y, X
    b += align_up((size_t)K * T_act);                        // frame activity
    b += align_up(sizeof(double) * (size_t)F * K * T);       // gamma
    b += 2 * align_up(sizeof(double) * (size_t)F * T);       // masks
    b += align_up(sizeof(cplx) * (size_t)F * T);             // Xhat
    b += 4096;
    size_t stage = 0;
    if (p->wpe) stage = std::max(stage, wpe_workspace_bytes(F, T, D, p->wpe_taps, p->wpe_delay) + (1 << 16));
    stage = std::max(stage, cacgmm_workspace_bytes(F, T, D, K));
    stage = std::max(stage, mvdr_workspace_bytes(F, T, D));
    stage = std::max(stage, stft_workspace_bytes(T, p->stft_size));
    return b + stage + (1 << 16);
}

static int enhance_observation_impl(gss_ctx *ctx, const gss_params *p, const void *obs,
                                    int obs_type, int D, int64_t N, const uint8_t *act, int K,
                                    int64_t N_act, int target, int64_t start_ctx, int64_t end_ctx,
                                    double *out, const gss_debug_taps *taps) {
    GSS_TRY(check_windows(ctx));
    GSS_TRY(check_params(ctx, p));
    GSS_REQUIRE(ctx, obs && act && out && N >= 1, GSS_ERR_INVALID,
                "gss_enhance_observation: bad arguments");
    GSS_REQUIRE(ctx, D >= 1 && D <= GSS_MAX_CHANNELS, GSS_ERR_UNSUPPORTED, "D=%d", D);
    GSS_REQUIRE(ctx, target >= 0 && target < K, GSS_ERR_INVALID,
                "target_index %d outside [0, %d)", target, K);
    // core.py:221-222
    GSS_REQUIRE(ctx, start_ctx >= 0 && end_ctx >= 0, GSS_ERR_INVALID,
                "assert context samples >= 0 failed: %lld %lld", (long long)start_ctx,
                (long long)end_ctx);
    GSS_TRY(check_cacgmm_args(ctx, D, K, p->bss_iterations, p->bss_iterations_post));
    if (p->bf == 0 || p->bf == 3)
        GSS_REQUIRE(ctx, D < 30, GSS_ERR_INVALID, "assert D < 30 failed: D=%d", D);
    if (p->bf == 1)
        GSS_REQUIRE(ctx, D > 2, GSS_ERR_INVALID, "bf='ch2' needs more than 2 channels");

    const int size = p->stft_size, shift = p->stft_shift, fading = p->stft_fading;
    const int F = size / 2 + 1;
    const int64_t T = gss_stft_num_frames(N, size, shift, fading);
    const int64_t T_act = gss_stft_num_frames(N_act, size, shift, fading);
    // GSS.__call__ uses initialization[..., :T]: fewer activity frames than STFT
    // frames is a shape error in the reference too
    GSS_REQUIRE(ctx, T_act >= T, GSS_ERR_INVALID,
                "activity covers %lld frames but the observation has %lld",
                (long long)T_act, (long long)T);

    GSS_REQUIRE(ctx, (int64_t)F * T * D < (1LL << 31), GSS_ERR_UNSUPPORTED,
                "F * T * D = %lld STFT bins: 2^31 or more are not supported (some kernels index "
                "the tensor with 32 bits)", (long long)((int64_t)F * T * D));
    GSS_TRY(arena_reserve(ctx, pipeline_workspace(p, F, T, T_act, D, K)));
    cplx *Y = arena_alloc_t<cplx>(ctx, (size_t)F * T * D);
    cplx *X = p->wpe ? arena_alloc_t<cplx>(ctx, (size_t)F * T * D) : Y;
    uint8_t *actf = arena_alloc_t<uint8_t>(ctx, (size_t)K * T_act);
    double *gamma = arena_alloc_t<double>(ctx, (size_t)F * K * T);
    double *mx = arena_alloc_t<double>(ctx, (size_t)F * T);
    double *mn = arena_alloc_t<double>(ctx, (size_t)F * T);
    cplx *Xhat = arena_alloc_t<cplx>(ctx, (size_t)F * T);
    int32_t *ref = arena_alloc_t<int32_t>(ctx, 4);
    GSS_REQUIRE(ctx, Y && X && actf && gamma && mx && mn && Xhat && ref, GSS_ERR_NOMEM,
                "workspace sizing bug");
    const size_t mark = ctx->arena_off;

    GSS_TRY(stft_run(ctx, obs, obs_type, D, N, fading, Y));
    if (!p->wpe)    // no solve in this call: clear the count an earlier utterance left behind
        GSS_HIP_CHECK(ctx, hipMemsetAsync(ctx->status_dev + 2, 0, sizeof(int32_t), ctx->stream));
    // The caller said "one utterance at a time on this GPU" (gss_set_utterances_in_flight(ctx, 1)):
    // the WPE stage runs as two sets of frequencies side by side on the context's stream and
    // its internal second stream -- one set's solve (a chain of latency-bound launches, MFMA
    // busy 0.3) under the other's correlation.  Frequencies are independent: the same bits.
    // Measured +1.1 ... +2.1 % at 4 / 12 / 20 / 24 channels for a single utterance and -1.6 %
    // when two utterances are in flight anyway (EXPERIMENTS round 6, item 8), hence the hint;
    // not the default because overlapped launches no longer have durations of their own (the
    // per-kernel table and the roofline of a profile are taken on one stream).  GSS_VARIANT
    // wpe_halves=0 / 1 forces it off / on, wpe_halves=n (n > 1) puts 8 n frequencies into the
    // first set.
    const int halves = gss_variant("wpe_halves", ctx->utterances_in_flight == 1 ? 1 : 0);
    if (p->wpe && halves > 0 && F >= 32 && p->wpe_iterations > 0) {
        GSS_TRY(aux_stream_ready(ctx));
        hipStream_t const main_stream = ctx->stream;
        const int F0 = halves > 1 ? std::min(halves * 8, F - 8) : (F / 2 + 7) / 8 * 8, F1 = F - F0;
        const size_t off = (size_t)F0 * T * D;
        int st = wpe_run(ctx, Y, F0, T, D, p->wpe_taps, p->wpe_delay, p->wpe_iterations,
                         p->wpe_psd_context, X, 0);
        if (st == GSS_OK) {
            GSS_HIP_CHECK(ctx, hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
            ctx->stream = ctx->aux_stream;
            st = wpe_run(ctx, Y + off, F1, T, D, p->wpe_taps, p->wpe_delay, p->wpe_iterations,
                         p->wpe_psd_context, X + off, 1);
            ctx->stream = main_stream;
            // (joined even when the second part failed to enqueue: nothing may be left running
            // on the internal stream when the call returns)
            GSS_HIP_CHECK(ctx, hipEventRecord(ctx->ev_join, ctx->aux_stream));
            GSS_HIP_CHECK(ctx, hipStreamWaitEvent(main_stream, ctx->ev_join, 0));
        }
        GSS_TRY(st);
        GSS_TRY(wpe_copy_zero_pivots(ctx));
        ctx->arena_off = mark;
    } else if (p->wpe) {
        GSS_TRY(wpe_run(ctx, Y, F, T, D, p->wpe_taps, p->wpe_delay, p->wpe_iterations,
                        p->wpe_psd_context, X));
        ctx->arena_off = mark;
    }
    GSS_TRY(activity_run(ctx, act, K, N_act, fading, actf));
    GSS_TRY(cacgmm_run(ctx, X, F, T, D, actf, T_act, K, p->bss_iterations, p->bss_iterations_post,
                       gamma));
    ctx->arena_off = mark;

    int64_t sf = 0, ef = 0;
    if (p->bf_drop_context) {
        sf = gss_samples_to_stft_frames(start_ctx, size, shift, fading);
        ef = gss_samples_to_stft_frames(end_ctx, size, shift, fading);
    }
    GSS_TRY(masks_from_posteriors_run(ctx, gamma, F, K, T, target, p->bf_drop_context, sf, ef,
                                      mx, mn));
    if (p->bf == 0 || p->bf == 3) {
        GSS_TRY(mvdr_run(ctx, X, F, T, D, mx, mn, /*ban=*/1, Xhat, ref, /*gev=*/p->bf == 3));
        ctx->arena_off = mark;
    } else {
        GSS_TRY(channel_pick_run(ctx, X, F, T, D, p->bf, Xhat));
    }
    if (p->postfilter == 1) GSS_TRY(mask_mul_run(ctx, Xhat, mx, F, T));
    GSS_TRY(istft_run(ctx, Xhat, T, fading, out));

    if (taps) {
        auto cp = [&](void *dst, const void *src, size_t bytes) -> int {
            if (!dst) return GSS_OK;
            GSS_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice,
                                              ctx->stream));
            return GSS_OK;
        };
        GSS_TRY(cp(taps->Obs_ftd, X, sizeof(cplx) * (size_t)F * T * D));
        if (taps->act_frames)
            GSS_HIP_CHECK(ctx, hipMemcpy2DAsync(taps->act_frames, (size_t)T, actf, (size_t)T_act,
                                                (size_t)T, (size_t)K, hipMemcpyDeviceToDevice,
                                                ctx->stream));
        GSS_TRY(cp(taps->gamma, gamma, sizeof(double) * (size_t)F * K * T));
        GSS_TRY(cp(taps->target_mask, mx, sizeof(double) * (size_t)F * T));
        GSS_TRY(cp(taps->distortion_mask, mn, sizeof(double) * (size_t)F * T));
        GSS_TRY(cp(taps->Xhat, Xhat, sizeof(cplx) * (size_t)F * T));
        if (p->bf == 0 || p->bf == 3) GSS_TRY(cp(taps->ref_channel, ref, sizeof(int32_t)));
    }
    return GSS_OK;
}

extern "C" int gss_enhance_observation(gss_ctx *ctx, const gss_params *p, const double *obs,
                                       int D, int64_t N, const uint8_t *act, int K,
                                       int64_t N_act, int target, int64_t start_ctx,
                                       int64_t end_ctx, double *out,
                                       const gss_debug_taps *taps) {
    GSS_ENTER_VARIANTS(ctx);
    return enhance_observation_impl(ctx, p, obs, 0, D, N, act, K, N_act, target, start_ctx, end_ctx,
                                    out, taps);
}

extern "C" int gss_enhance_observation_pcm16(gss_ctx *ctx, const gss_params *p,
                                             const int16_t *obs, int D, int64_t N,
                                             const uint8_t *act, int K, int64_t N_act, int target,
                                             int64_t start_ctx, int64_t end_ctx, double *out,
                                             const gss_debug_taps *taps) {
    GSS_ENTER_VARIANTS(ctx);
    return enhance_observation_impl(ctx, p, obs, 1, D, N, act, K, N_act, target, start_ctx, end_ctx,
                                    out, taps);
}

extern "C" int gss_enhance_observation_host(gss_ctx *ctx, const gss_params *p,
                                            const double *obs, int D, int64_t N,
                                            const uint8_t *act, int K, int64_t N_act,
                                            int target, int64_t start_ctx, int64_t end_ctx,
                                            double *out) {
    GSS_ENTER_VARIANTS(ctx);
    GSS_REQUIRE(ctx, p && obs && act && out && N >= 1 && D >= 1 && K >= 1, GSS_ERR_INVALID,
                "gss_enhance_observation_host: bad arguments");
    const int64_t T = gss_stft_num_frames(N, p->stft_size, p->stft_shift, p->stft_fading);
    const int64_t n_out = gss_istft_num_samples(T, p->stft_size, p->stft_shift, p->stft_fading);
    double *obs_d = nullptr, *out_d = nullptr;
    uint8_t *act_d = nullptr;
    int st = gss_dev_malloc(ctx, sizeof(double) * (size_t)D * N, (void **)&obs_d);
    if (st == GSS_OK) st = gss_dev_malloc(ctx, (size_t)K * N_act, (void **)&act_d);
    if (st == GSS_OK) st = gss_dev_malloc(ctx, sizeof(double) * (size_t)n_out, (void **)&out_d);
    if (st == GSS_OK) st = gss_memcpy_h2d(ctx, obs_d, obs, sizeof(double) * (size_t)D * N);
    if (st == GSS_OK) st = gss_memcpy_h2d(ctx, act_d, act, (size_t)K * N_act);
    if (st == GSS_OK)
        st = gss_enhance_observation(ctx, p, obs_d, D, N, act_d, K, N_act, target, start_ctx,
                                     end_ctx, out_d, nullptr);
    if (st == GSS_OK) st = gss_memcpy_d2h(ctx, out, out_d, sizeof(double) * (size_t)n_out);
    std::string keep = ctx->error;
    (void)hipStreamSynchronize(ctx->stream);
    if (obs_d) (void)hipFree(obs_d);
    if (act_d) (void)hipFree(act_d);
    if (out_d) (void)hipFree(out_d);
    ctx->error = keep;
    return st;
}
