// smalln_prof.hip -- phase-by-phase cycle counts of fit_small_kernel at iteration 5 (development aid).
#define SMALLN_PROFILE
#include "../gpim_amd/csrc/smalln.hip"
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <vector>
void gpim_set_error(const std::string&) {}
static void run(int N) {
    const int T = 200, d = 2, P = 4;
    std::vector<double> X(N * d), y(N), u(P, 0.0), bc(2 * T);
    for (int i = 0; i < N; ++i) { X[2 * i] = (i * 7) % 25; X[2 * i + 1] = (i * 11) % 25 + 0.25 * (i / 25); y[i] = sin(0.3 * X[2 * i]) * cos(0.2 * X[2 * i + 1]); }
    for (int t = 1; t <= T; ++t) { bc[t - 1] = 0.05 / (1 - pow(0.9, t)); bc[T + t - 1] = sqrt(1 - pow(0.999, t)); }
    double *dX, *dy, *du, *dbc, *hist, *loss; int* info; long long* prof;
    hipMalloc(&dX, N * d * 8); hipMalloc(&dy, N * 8); hipMalloc(&du, P * 8); hipMalloc(&dbc, 2 * T * 8);
    hipMalloc(&hist, T * P * 8); hipMalloc(&loss, T * 8); hipMalloc(&info, 4); hipMalloc(&prof, 16 * 8);
    hipMemcpy(dX, X.data(), N * d * 8, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), N * 8, hipMemcpyHostToDevice);
    hipMemcpy(dbc, bc.data(), 2 * T * 8, hipMemcpyHostToDevice);
    hipMemset(info, 0, 4);
    SmallFitArgs a;
    memset(&a, 0, sizeof(a));
    a.m.kernel = 0; a.m.dim = 2; a.m.n_ls = 2; a.m.amp_lo = 1e-4; a.m.amp_hi = 10; a.m.jitter = 1e-6;
    for (int k = 0; k < 2; ++k) { a.m.ls_lo[k] = 0; a.m.ls_hi[k] = 12.5; }
    a.X = dX; a.y = dy; a.N = N; a.T = T; a.u = du; a.lr_over_bc1 = dbc; a.bc2_sqrt = dbc + T; a.hist = hist; a.loss = loss; a.info = info; a.prof = prof;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(du, u.data(), P * 8, hipMemcpyHostToDevice);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((fit_small_kernel<0>), dim3(1), dim3(512), 0, 0, a);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
    }
    long long hp[16];
    double hl[2];
    hipMemcpy(hp, prof, 16 * 8, hipMemcpyDeviceToHost);
    hipMemcpy(hl, loss + T - 2, 16, hipMemcpyDeviceToHost);
    const char* names[] = {"theta", "scale-x", "K build", "factor", "logdet+trinv16", "inverse levels", "z, alpha", "K^-1 tiles + grad", "reduce", "finalize+Adam"};
    printf("N=%d: %.2f us/iteration over %d; final loss %.12g; stamped iteration total %lld cycles\n", N, 1e3 * ms / T, T, hl[1], hp[10] - hp[0]);
    for (int i = 0; i < 10; ++i) printf("  %-20s %7lld\n", names[i], hp[i + 1] - hp[i]);
    hipFree(dX); hipFree(dy); hipFree(du); hipFree(dbc); hipFree(hist); hipFree(loss); hipFree(info); hipFree(prof);
}

int main(int argc, char** argv) {
    if (argc > 1) { run(atoi(argv[1])); return 0; }
    const int sizes[] = {16, 30, 60, 100, 128};
    for (int N : sizes) run(N);
    return 0;
}
