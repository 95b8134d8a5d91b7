// f64 MFMA issue-rate microbenchmark behind DESIGN.md section 6 / 8 (round 4): what does
// v_mfma_f64_16x16x4_f64 sustain on gfx950, alone and with the instruction mix of the wpe_corr
// k-step around it (12 MFMAs on 12 independent accumulators; + 8 f64 VALU operand products;
// + 5 LDS operand reads; + a few integer VALU address updates), at 1, 2 and 3 waves per SIMD?
//
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 tools/micro/mfma_f64_bench.hip \
//         -o /tmp/mfma_f64_bench && /tmp/mfma_f64_bench
//
// Reported per (mode, waves/SIMD): us per launch, shader clock, shader cycles per MFMA per SIMD
// (64 = the datasheet rate: 2048 flop / 64 cycles / SIMD = 78.6 TFLOP/s at 2.4 GHz) and the
// TFLOP/s that is.
#include <hip/hip_runtime.h>

#include <cstdio>

typedef double v4d __attribute__((ext_vector_type(4)));

// MODE 0: MFMAs only.  1: + 8 VALU f64 per 12 MFMAs.  2: + 5 ds_read_b128 per 12 MFMAs
// (operands really come from LDS).  3: mode 2 + 3 integer VALU.  4: mode 2 with a workgroup
// barrier every 16 k-steps (4-wave workgroups).
template <int MODE>
__global__ __launch_bounds__(256) void mfma_kernel(double *out, int iters, long long *clk) {
    __shared__ double2 S[4096];
    const long long c0 = clock64(), r0 = wall_clock64();
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) S[i] = make_double2(1e-3 * i, 1e-4 * i);
    __syncthreads();
    v4d acc[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a0 = 1.0 + 1e-9 * threadIdx.x, a1 = 0.5, b0 = 1e-12 * (threadIdx.x + 1), b1 = 0.25;
    double w = 1.0000001;
    int addr = threadIdx.x & 63;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) {
            double ar0 = a0, ai0 = a1, ar1 = a0, ai1 = a1, br0 = b0, bi0 = b1, br1 = b0, bi1 = b1;
            if (MODE >= 2) {
                const double2 x0 = S[(addr + 64 * ks) & 4095], x1 = S[(addr + 64 * ks + 16) & 4095];
                const double2 y0 = S[(addr + 64 * ks + 32) & 4095], y1 = S[(addr + 64 * ks + 48) & 4095];
                ar0 = x0.x; ai0 = x0.y; ar1 = x1.x; ai1 = x1.y;
                br0 = y0.x; bi0 = y0.y; br1 = y1.x; bi1 = y1.y;
                w = S[(addr + ks) & 4095].x;
            }
            double as0 = ar0, as1 = ar1, bd0 = br0, bd1 = br1;
            if (MODE >= 1) {
                ar0 *= w; ai0 *= w; as0 = ar0 + ai0;
                ar1 *= w; ai1 *= w; as1 = ar1 + ai1;
                bd0 = br0 - bi0; bd1 = br1 - bi1;
                asm volatile("" : "+v"(ar0), "+v"(ai0), "+v"(as0), "+v"(ar1), "+v"(ai1), "+v"(as1), "+v"(bd0), "+v"(bd1));
            }
            if (MODE == 3) {
                addr += 24; asm volatile("" : "+v"(addr));
                addr ^= 5; asm volatile("" : "+v"(addr));
                addr += 3; asm volatile("" : "+v"(addr));
            }
            acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar0, br0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai0, bi0, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(as0, bd0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar0, br1, acc[3], 0, 0, 0);
            acc[4] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai0, bi1, acc[4], 0, 0, 0);
            acc[5] = __builtin_amdgcn_mfma_f64_16x16x4f64(as0, bd1, acc[5], 0, 0, 0);
            acc[6] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar1, br0, acc[6], 0, 0, 0);
            acc[7] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai1, bi0, acc[7], 0, 0, 0);
            acc[8] = __builtin_amdgcn_mfma_f64_16x16x4f64(as1, bd0, acc[8], 0, 0, 0);
            acc[9] = __builtin_amdgcn_mfma_f64_16x16x4f64(ar1, br1, acc[9], 0, 0, 0);
            acc[10] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai1, bi1, acc[10], 0, 0, 0);
            acc[11] = __builtin_amdgcn_mfma_f64_16x16x4f64(as1, bd1, acc[11], 0, 0, 0);
        }
        if (MODE == 4) __syncthreads();
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < 12; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = clock64() - c0;
        clk[1] = wall_clock64() - r0;
    }
    if (s == 12345.678) out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    double *out;
    long long *clk, hclk[2];
    hipMalloc(&out, 1 << 26);
    hipMalloc(&clk, 16);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    const int iters = 60;
    const char *names[] = {"12 MFMA", "+ 8 VALU f64", "+ 5 ds_read_b128 (+1 b64)", "+ 3 int VALU",
                           "mode 2 + barrier / 16 k-steps"};
    printf("%-32s %10s %10s %10s %18s %10s\n", "k-step", "waves/SIMD", "us/launch", "clock GHz",
           "cycles/MFMA/SIMD", "TFLOP/s");
    for (int mode = 0; mode < 5; ++mode)
        for (int wps : {1, 2, 3}) {
            const int blocks = 256 * wps;       // 4-wave workgroups: one wave per SIMD each
            auto launch = [&]() {
                switch (mode) {
                    case 0: hipLaunchKernelGGL(mfma_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, iters, clk); break;
                    case 1: hipLaunchKernelGGL(mfma_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, iters, clk); break;
                    case 2: hipLaunchKernelGGL(mfma_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, iters, clk); break;
                    case 3: hipLaunchKernelGGL(mfma_kernel<3>, dim3(blocks), dim3(256), 0, 0, out, iters, clk); break;
                    case 4: hipLaunchKernelGGL(mfma_kernel<4>, dim3(blocks), dim3(256), 0, 0, out, iters, clk); break;
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
            const double ghz = (double)hclk[0] / ((double)hclk[1] * 10.0);   // 100 MHz reference
            const double mfma_per_simd = (double)iters * 16 * 12 * wps;
            const double cyc = us * 1e-6 * ghz * 1e9 / mfma_per_simd;
            const double tflops = mfma_per_simd * 1024 * 2048 / (us * 1e-6) * 1e-12;
            printf("%-32s %10d %10.1f %10.3f %18.2f %10.1f\n", names[mode], wps, us, ghz, cyc, tflops);
        }
    return 0;
}
