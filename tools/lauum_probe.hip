// lauum_probe.hip -- is the K^-1 = L^-T L^-1 tile-engine launch limited by operand traffic?
// Runs the library's own launch (same tile list, same kernel) twice: on the real N x N operand and
// with leading dimension 0, where every operand row aliases row 0 (all loads hit L2).  Development aid.
//   hipcc --offload-arch=gfx950 -O3 tools/lauum_probe.hip -Lgpim_amd -lgpimhip -Wl,-rpath,$PWD/gpim_amd
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <vector>
#include "../include/gpimhip.h"

int launch_lauum(gpimhip_ctx* h, const double* A, double* B, int64_t np, int64_t ld);

int main(int argc, char** argv) {
    const int64_t N = argc > 1 ? atoll(argv[1]) : 16384;
    gpimhip_handle h;
    if (gpimhip_create(&h, 0, nullptr)) { printf("create failed\n"); return 1; }
    double *A, *B;
    hipMalloc(&A, N * N * 8);
    hipMalloc(&B, N * N * 8);
    std::vector<double> row(N);
    srand(1);
    for (auto& v : row) v = rand() / (double)RAND_MAX - 0.5;
    for (int64_t i = 0; i < N; ++i) hipMemcpy(A + i * N, row.data() + (i % 7), (N - 7) * 8, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int64_t ld : {N, (int64_t)0, N}) {
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0, 0);
            if (launch_lauum((gpimhip_ctx*)h, A, B, N, ld)) { printf("launch failed: %s\n", gpimhip_last_error()); return 1; }
            hipEventRecord(e1, 0);
            hipDeviceSynchronize();
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("ld=%lld: %.2f ms  %.1f TFLOP/s\n", (long long)ld, ms, (double)N * N * N / 3 / ms / 1e9);
        }
    }
    return 0;
}
