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
    return 0;
}
