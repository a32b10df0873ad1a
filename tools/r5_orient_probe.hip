// r5_orient_probe.hip -- rate of the tile shape hosted by the Cholesky step launches (8 waves, 128x128, two LDS stages, two
// workgroups per CU under an 80 KB static allocation) for the four operand orientations: would a transposed copy of the
// factor's panels (trailing update as KM x MK or KM x KM instead of MK x MK) pay?
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Igpim_amd/csrc tools/r5_orient_probe.hip -o tools/r5_orient_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "gemm_body.hpp"

template <bool A_KM, bool B_KM, int NW, int LB>
__global__ __launch_bounds__(NW * 64, LB) void probe(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) double smem[80 * 128 - 64];
    gemm_tile_body<A_KM, B_KM, EPI_STORE, NW, 128, 128, 2>(g, (int)blockIdx.x, 0, smem);
}

template <bool A_KM, bool B_KM, int NW, int LB>
static void run(const char* name, GemmArgs g, int kb) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((probe<A_KM, B_KM, NW, LB>), dim3(g.ntiles), dim3(NW * 64), 0, 0, g);
    hipEventRecord(e0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((probe<A_KM, B_KM, NW, LB>), dim3(g.ntiles), dim3(NW * 64), 0, 0, g);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 2.0 * g.ntiles * 128.0 * 128.0 * kb * 128.0;
    printf("%-28s tiles %5d kblocks %3d : %8.3f ms  %6.2f TFLOP/s  (%s)\n", name, g.ntiles, kb, ms / reps,
           flop / (ms / reps * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
}

int main() {
    const int nb = 96;
    const int64_t n = (int64_t)nb * 128, ld = n;
    double *A, *C;
    hipMalloc(&A, n * ld * 8);
    hipMalloc(&C, n * ld * 8);
    std::vector<double> h(n * ld);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (double)((i * 2654435761u) % 1024) / 1024.0 - 0.5;
    hipMemcpy(A, h.data(), n * ld * 8, hipMemcpyHostToDevice);
    hipMemset(C, 0, n * ld * 8);
    for (int kb : {8, 32}) {
        for (int side : {16, 32, 48}) {
            GemmArgs g{};
            g.A = A; g.lda = ld; g.B = A; g.ldb = ld; g.C = C; g.ldc = ld;
            g.alpha = -1.0; g.beta = 1.0;
            g.rect_rows = side; g.rect_cols = side; g.ntiles = side * side; g.chunk = 64;
            g.kfix0 = 0; g.kfix1 = kb;
            // operands from block rows / columns 48.. so that the two operands do not alias the same panels
            g.a_roff = 48; g.a_coff = 0; g.b_roff = 48; g.b_coff = 0; g.c_roff = 48; g.c_coff = 0;
            printf("-- %d x %d tiles, k = %d blocks\n", side, side, kb);
            run<false, false, 8, 4>("MK x MK (trailing update)", g, kb);
            run<false, true, 8, 4>("MK x KM (inverse phases)", g, kb);
            run<true, true, 8, 4>("KM x KM", g, kb);
            run<true, false, 8, 4>("KM x MK", g, kb);
            run<false, false, 4, 2>("MK x MK 4 waves", g, kb);
            run<false, true, 4, 2>("MK x KM 4 waves", g, kb);
            run<true, true, 4, 2>("KM x KM 4 waves", g, kb);
        }
    }
    return 0;
}
