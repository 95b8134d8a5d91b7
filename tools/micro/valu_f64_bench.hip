// VALU f64 issue-rate microbenchmark behind DESIGN.md section 6: how many cycles does a wave64
// v_fma_f64 / v_mul_f64 occupy its SIMD on gfx950, with VGPR-only operands and with one SGPR
// operand (the form the CACGMM E-step uses), at 1, 2 and 4 waves per SIMD?
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_f64_bench.hip -o /tmp/valu_f64_bench && /tmp/valu_f64_bench
//
// Every wave runs `iters` x 64 independent accumulating FMAs (16 accumulators, 4 rounds, so a
// dependent instruction is 16 issues away).  Reported: time per launch and cycles per
// instruction per SIMD at the shader clock measured in the kernel (s_memtime against the
// 100 MHz s_memrealtime).
#include <hip/hip_runtime.h>

#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(64) void valu_kernel(double *out, int iters, double s0, double s1,
                                                  long long *clk) {
    const long long c0 = clock64(), r0 = wall_clock64();
    double acc[16];
    const double x = 1.0 + 1e-9 * threadIdx.x, y = 1e-12 * (threadIdx.x + 1);
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = i;
    double sa = s0, sb = s1;
    asm volatile("" : "+s"(sa), "+s"(sb));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                if (MODE == 0) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(x), "v"(y));
                if (MODE == 1) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "s"(sa), "v"(y));
                if (MODE == 2) asm volatile("v_mul_f64 %0, %1, %0" : "+v"(acc[i]) : "v"(x));
                if (MODE == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(reinterpret_cast<float &>(acc[i])) : "v"((float)x), "v"((float)y));
                if (MODE == 4) asm volatile("v_add_f64 %0, %1, %0" : "+v"(acc[i]) : "v"(y));
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = clock64() - c0;            // shader clock (s_memtime)
        clk[1] = wall_clock64() - r0;       // constant 100 MHz (s_memrealtime)
    }
    if (s == 12345.678) out[blockIdx.x * 64 + threadIdx.x] = s;
}

int main() {
    double *out;
    long long *clk, hclk[2];
    hipMalloc(&out, 1 << 24);
    hipMalloc(&clk, 16);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int iters = 2000;
    const char *names[] = {"v_fma_f64 v,v,v", "v_fma_f64 s,v,v", "v_mul_f64 v,v", "v_fma_f32 v,v,v", "v_add_f64 v,v"};
    printf("%-18s %10s %12s %10s %22s\n", "instruction", "waves/SIMD", "us/launch", "clock GHz", "shader cycles/instr/SIMD");
    for (int mode = 0; mode < 5; ++mode)
        for (int wps : {1, 2, 4}) {
            const int blocks = 256 * 4 * wps;
            auto launch = [&]() {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(valu_kernel<0>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0, 2.0, clk); break;
                    case 1: hipLaunchKernelGGL(valu_kernel<1>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0, 2.0, clk); break;
                    case 2: hipLaunchKernelGGL(valu_kernel<2>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0, 2.0, clk); break;
                    case 3: hipLaunchKernelGGL(valu_kernel<3>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0, 2.0, clk); break;
                    case 4: hipLaunchKernelGGL(valu_kernel<4>, dim3(blocks), dim3(64), 0, 0, out, iters, 1.0, 2.0, clk); break;
                }
            };
            launch();
            hipDeviceSynchronize();
            hipEventRecord(a);
            for (int r = 0; r < 5; ++r) launch();
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms;
            hipEventElapsedTime(&ms, a, b);
            const double us = ms * 1000 / 5;
            hipMemcpy(hclk, clk, 16, hipMemcpyDeviceToHost);
            const double ghz = (double)hclk[0] / ((double)hclk[1] * 10.0);     // cycles per ns
            const double instr_per_simd = (double)iters * 64 * wps;
            printf("%-18s %10d %12.1f %10.3f %22.2f\n", names[mode], wps, us, ghz,
                   us * 1000.0 * ghz / instr_per_simd);
        }
    return 0;
}
