// mfma_peak.hip -- micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate on this GPU.
// Used to check the 78.6 TFLOP/s fp64 matrix peak that bench.py prices the GEMM engine against.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(1024) void mfma_loop(double* out, int iters, double a0, double b0) {
    d4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = a0 + threadIdx.x * 1e-9, b = b0;
    long long c0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    long long c1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 123.456) out[0] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[1] = (double)(c1 - c0);
}

template <int NACC>
void run(int blocks, int threads, int iters, const char* tag) {
    double* out; hipMalloc(&out, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop<NACC>), dim3(blocks), dim3(threads), 0, 0, out, 10, 1.0, 1.0);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((mfma_loop<NACC>), dim3(blocks), dim3(threads), 0, 0, out, iters, 1.0, 1.0);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    double flops = (double)blocks * (threads / 64) * iters * NACC * 2048.0;
    double cyc[2]; hipMemcpy(cyc, out, 16, hipMemcpyDeviceToHost);
    printf("%-28s blocks=%d threads=%d acc=%d: %.3f ms  %.2f TFLOP/s | wave0: %.1f cyc/MFMA, eff clock %.2f GHz\n", tag, blocks, threads, NACC, best,
           flops / best / 1e9, cyc[1] / ((double)iters * NACC), cyc[1] / (best * 1e6));
    hipFree(out);
}

int main() {
    run<16>(256, 512, 5000, "2 waves/SIMD (1 WG/CU), 16 acc");
    run<8>(256, 1024, 10000, "4 waves/SIMD (1 WG/CU), 8 acc");
    run<16>(256, 1024, 5000, "4 waves/SIMD (1 WG/CU), 16 acc");
    run<4>(256, 256, 20000, "1 wave/SIMD, 4 acc");
    run<16>(256, 256, 5000, "1 wave/SIMD, 16 acc");
    run<4>(512, 256, 20000, "2 waves/SIMD, 4 acc");
    run<16>(512, 256, 5000, "2 waves/SIMD, 16 acc");
    run<4>(1024, 256, 20000, "4 waves/SIMD, 4 acc");
    run<1>(2048, 256, 40000, "8 waves/SIMD, 1 acc");
    return 0;
}
