// Scalar-memory microbenchmark behind DESIGN.md section 8: how fast can a CU feed wave-uniform
// data (the CACGMM model rows of the register-form E-step) to its waves through s_load?
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/smem_bench.hip -o /tmp/smem_bench && /tmp/smem_bench
//
// Every wave streams `bytes` of one of `nbuf` buffers with s_load_dwordx16 (64 B per load), adds
// one dword of each load into an accumulator (so that nothing is dropped) and waits with
// s_waitcnt lgkmcnt(0) after every `per_wait` loads.  Reported: time per launch, loads per
// microsecond per CU, bytes per cycle per CU (2.4 GHz nominal) for 1..16 waves per CU and for
// a footprint that fits the scalar cache (1 KB per wave, re-read) or not (24 KB per wave).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int PER_WAIT>
__global__ __launch_bounds__(64) void smem_kernel(const unsigned *__restrict__ buf, int span_bytes,
                                                  int loads, int nbuf, unsigned *out) {
    const int which = blockIdx.x % nbuf;
    const unsigned *p = buf + (size_t)which * (span_bytes / 4);
    unsigned acc = 0;
    int off = 0;
    for (int i = 0; i < loads; i += PER_WAIT) {
        typedef unsigned v16 __attribute__((ext_vector_type(16)));
        v16 line[PER_WAIT];
#pragma unroll
        for (int j = 0; j < PER_WAIT; ++j) {        // all requests first ...
            // a uniform pointer into the constant address space: s_load_dwordx16
            line[j] = *reinterpret_cast<const __attribute__((address_space(4))) v16 *>(
                (unsigned long long)(p + off / 4));
            off += 64;
            if (off >= span_bytes) off = 0;
        }
#pragma unroll
        for (int j = 0; j < PER_WAIT; ++j)          // ... then one wait and the use
            acc += line[j][0] ^ line[j][7] ^ line[j][15];
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc;
}

int main() {
    const int CUS = 256;
    const int total_bytes = 64 << 20;
    unsigned *buf, *out;
    hipMalloc(&buf, total_bytes);
    hipMalloc(&out, 1 << 22);
    hipMemset(buf, 1, total_bytes);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int loads = 375 * 8;     // 8 x the model of one frequency at D = 24, K = 5
    printf("%-28s %8s %12s %14s %14s\n", "case", "waves/CU", "us/launch", "loads/us/CU", "B/cycle/CU");
    for (int span : {1024, 24000}) {
        for (int wpc : {1, 2, 4, 8, 16}) {
            for (int per_wait : {1, 2, 4}) {
                const int blocks = CUS * wpc;
                const int nbuf = span == 1024 ? 1 : 513;
                auto launch = [&]() {
                    if (per_wait == 1) hipLaunchKernelGGL(smem_kernel<1>, dim3(blocks), dim3(64), 0, 0, buf, span, loads, nbuf, out);
                    else if (per_wait == 2) hipLaunchKernelGGL(smem_kernel<2>, dim3(blocks), dim3(64), 0, 0, buf, span, loads, nbuf, out);
                    else hipLaunchKernelGGL(smem_kernel<4>, dim3(blocks), dim3(64), 0, 0, buf, span, loads, nbuf, out);
                };
                launch();
                hipDeviceSynchronize();
                hipEventRecord(a);
                for (int r = 0; r < 5; ++r) launch();
                hipEventRecord(b);
                hipEventSynchronize(b);
                float ms;
                hipEventElapsedTime(&ms, a, b);
                const double us = ms * 1e3 / 5;
                const double lpu = (double)loads * wpc / us;
                char name[64];
                snprintf(name, sizeof(name), "%s, wait every %d", span == 1024 ? "1 KB (cache hits)" : "24 KB x 513 bufs", per_wait);
                printf("%-28s %8d %12.1f %14.1f %14.2f\n", name, wpc, us, lpu, lpu * 64 / 2400.0);
            }
        }
    }
    return 0;
}
