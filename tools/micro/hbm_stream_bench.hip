// HBM stream microbenchmark behind EXPERIMENTS.md (round 6, item 1): what does MI355X sustain for
// the access patterns of the CACGMM E-step and M-step at config 5 (12 channels, T = 7503: 739 MB
// of unit-normalised observation + 154 MB of weights per launch -- larger than the 256 MB
// Infinity Cache), and what for the same pattern on a working set that fits the cache?
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/hbm_stream_bench.hip -o /tmp/hbm_stream_bench && /tmp/hbm_stream_bench
//
// Kernels (256 threads, 16-byte loads, grid-stride, nothing but the traffic):
//   read      sum of a buffer                               (the M-step's pattern: read only)
//   read+write   read `r` bytes, write r * 154 / 739 bytes  (the E-step's pattern)
//   copy      read r, write r
// each on 893 MB (HBM) and on 96 MB (Infinity-Cache resident after the first pass), 20 passes.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void read_kernel(const double2 *__restrict__ in, size_t n,
                                                   double *__restrict__ out) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = in[i];
        s += v.x + v.y;
    }
    if (s == 1.2345e300) out[0] = s;
}

// reads n elements, writes `num` of every `den` 256-element blocks of them (every element read
// feeds a sum that is stored at the end, so no load can be skipped)
__global__ __launch_bounds__(256) void read_write_kernel(const double2 *__restrict__ in, size_t n,
                                                         double2 *__restrict__ out, int num, int den,
                                                         double *__restrict__ sink) {
    double s = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const double2 v = in[i];
        s += v.x + v.y;
        const size_t blk = i >> 8;
        if ((int)(blk % den) < num) out[(blk / den * num + blk % den) * 256 + (i & 255)] = make_double2(v.y, s);
    }
    if (s == 1.2345e300) sink[0] = s;
}

// The M-step's pattern at config 5 (D = 12, T = 7503): a workgroup of 256 threads walks the
// 64-frame tiles of ONE frequency; a tile is D rows of 1 KB that lie T * 16 bytes apart in the
// (F, D, T) layout (ROWS apart), or -- TILE_MAJOR -- one contiguous block of D KB.  `depth` tiles
// are requested before the first is consumed (registers), `wgs` workgroups share the F x tiles
// items like the static partition does.
template <int DEPTH>
__global__ __launch_bounds__(256) void tile_walk_kernel(const double2 *__restrict__ in, int F, int D,
                                                        int ntile, long long row_stride, int tile_major,
                                                        double *__restrict__ sink) {
    const int items = F * ntile;
    const int per = (items + gridDim.x - 1) / gridDim.x;
    const int i0 = blockIdx.x * per, i1 = min(items, i0 + per);
    const int tl = threadIdx.x & 63, g = threadIdx.x >> 6;
    double s = 0.0;
    double2 buf[DEPTH][3];
    auto issue = [&](int item, double2 (&b)[3]) {
        const int f = item / ntile, t = item - f * ntile;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int d = g + 4 * j;
            const long long off = tile_major ? ((long long)(f * ntile + t) * D + d) * 64 + tl
                                             : ((long long)f * D + d) * row_stride + (long long)t * 64 + tl;
            b[j] = d < D ? in[off] : make_double2(0.0, 0.0);
        }
    };
#pragma unroll
    for (int k = 0; k < DEPTH; ++k)
        if (i0 + k < i1) issue(i0 + k, buf[k]);
    for (int i = i0; i < i1; i += DEPTH) {
#pragma unroll
        for (int k = 0; k < DEPTH; ++k) {
            if (i + k < i1) {
#pragma unroll
                for (int j = 0; j < 3; ++j) s += buf[k][j].x + buf[k][j].y;
                if (i + k + DEPTH < i1) issue(i + k + DEPTH, buf[k]);
            }
        }
    }
    if (s == 1.2345e300) sink[0] = s;
}

static void run_walk(const char *name, int wgs, int depth, int tile_major) {
    const int F = 513, D = 12, ntile = 118;
    const long long row_stride = (long long)ntile * 64;
    const size_t n = (size_t)F * D * row_stride;
    double2 *in;
    double *sink;
    if (hipMalloc(&in, n * sizeof(double2)) != hipSuccess || hipMalloc(&sink, 64) != hipSuccess ||
        hipMemset(in, 0, n * sizeof(double2)) != hipSuccess) {
        printf("allocation failed\n");
        return;
    }
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    auto launch = [&]() {
        if (depth == 1) hipLaunchKernelGGL(tile_walk_kernel<1>, dim3(wgs), dim3(256), 0, 0, in, F, D, ntile, row_stride, tile_major, sink);
        else if (depth == 2) hipLaunchKernelGGL(tile_walk_kernel<2>, dim3(wgs), dim3(256), 0, 0, in, F, D, ntile, row_stride, tile_major, sink);
        else hipLaunchKernelGGL(tile_walk_kernel<4>, dim3(wgs), dim3(256), 0, 0, in, F, D, ntile, row_stride, tile_major, sink);
    };
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int p = 0; p < 20; ++p) launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    printf("%-58s %5d workgroups  %8.3f ms/pass  %6.2f TB/s\n", name, wgs, ms / 20,
           (double)n * 16 * 20 / (ms * 1e-3) / 1e12);
    (void)hipFree(in);
    (void)hipFree(sink);
}

static double run(const char *name, size_t bytes_in, int num, int den, int passes) {
    const size_t n = bytes_in / sizeof(double2);
    double2 *in, *out;
    double *sink;
    if (hipMalloc(&in, n * sizeof(double2)) != hipSuccess || hipMalloc(&out, n * sizeof(double2)) != hipSuccess ||
        hipMalloc(&sink, 64) != hipSuccess || hipMemset(in, 0, n * sizeof(double2)) != hipSuccess) {
        printf("allocation failed\n");
        return 0.0;
    }
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int grid = 256 * 16;
    auto launch = [&]() {
        if (num == 0) hipLaunchKernelGGL(read_kernel, dim3(grid), dim3(256), 0, 0, in, n, sink);
        else hipLaunchKernelGGL(read_write_kernel, dim3(grid), dim3(256), 0, 0, in, n, out, num, den, sink);
    };
    launch();
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int p = 0; p < passes; ++p) launch();
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms = 0.f;
    hipEventElapsedTime(&ms, a, b);
    const double moved = (double)bytes_in * (1.0 + (double)num / den);
    const double tbs = moved * passes / (ms * 1e-3) / 1e12;
    printf("%-44s %8.1f MB in %8.1f MB out  %8.3f ms/pass  %6.2f TB/s\n", name, bytes_in / 1e6,
           bytes_in * (double)num / den / 1e6, ms / passes, tbs);
    (void)hipFree(in);
    (void)hipFree(out);
    (void)hipFree(sink);
    return tbs;
}

int main() {
    const size_t big = 739ull << 20, small = 96ull << 20;
    printf("%-44s %s\n", "pattern", "per pass");
    run("read only, 739 MB (HBM)", big, 0, 1, 20);
    run("read only, 96 MB (Infinity Cache)", small, 0, 1, 100);
    run("read 739 MB + write 21 % (E-step), HBM", big, 5, 24, 20);
    run("read 96 MB + write 21 %, Infinity Cache", small, 5, 24, 100);
    run("copy 739 MB -> 739 MB, HBM", big, 1, 1, 20);
    run("copy 96 MB -> 96 MB, Infinity Cache", small, 1, 1, 100);
    printf("\nM-step walk (F = 513, D = 12, 118 tiles of 64 frames = 744 MB), tiles in flight per workgroup:\n");
    for (int wgs : {512, 768, 2048})
        for (int depth : {1, 2, 4}) {
            char name[96];
            snprintf(name, sizeof name, "(F, D, T) rows 118 KB apart, %d tile(s) ahead", depth);
            run_walk(name, wgs, depth, 0);
            snprintf(name, sizeof name, "tile-major (F, T/64, D, 64), %d tile(s) ahead", depth);
            run_walk(name, wgs, depth, 1);
        }
    return 0;
}
