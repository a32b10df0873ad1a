// potf2_prof.hip -- phase-by-phase cycle counts of potf2_inv_kernel (development aid).
#define POTF2_PROFILE
#include "potf2_kernel.hip"
#include <stdio.h>
#include <string.h>
#include <vector>
void gpim_set_error(const std::string&) {}
int main() {
    const int n = 128;
    std::vector<double> A(n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) A[i * n + j] = (i == j ? 2.0 : 0.0) + 1.0 / (1 + abs(i - j));
    double *dA, *dinv, *ld; int* info; long long* prof;
    hipMalloc(&dA, n * n * 8); hipMalloc(&dinv, n * n * 8); double* l16; hipMalloc(&l16, 8 * 256 * 8); hipMalloc(&ld, 8); hipMalloc(&info, 4); hipMalloc(&prof, 32 * 8);
    hipMemset(info, 0, 4);
    long long hp[32];
    for (int rep = 0; rep < 3; ++rep) {
        hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(potf2_kernel<double>, dim3(1), dim3(512), 0, 0, dA, (int64_t)n, 0, dinv, ld, info, 1, prof);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(hp, prof, 32 * 8, hipMemcpyDeviceToHost);
        printf("rep %d: %.1f us total; cycles: load %lld | factor %lld | write L + logdet %lld | inverse %lld | write inverse %lld | all %lld\n",
               rep, ms * 1e3, hp[1] - hp[0], hp[2] - hp[1], hp[3] - hp[2], hp[4] - hp[3], hp[5] - hp[4], hp[5] - hp[0]);
    }
#if FI_LOOKAHEAD
    {
        long long wp[256], fp[128];
        hipMemcpyFromSymbol(wp, HIP_SYMBOL(g_wprof), sizeof(wp));
        hipMemcpyFromSymbol(fp, HIP_SYMBOL(g_fprof), sizeof(fp));
        printf("look-ahead schedule (lds_factor_inv_la): wave 0 per iteration q: [chol16_lp(q) | wait flag_u | solve (q+1,q) | update (q+1,q+1)], iteration total = start of q+1 - start of q\n");
        for (int q = 0; q < 8; ++q)
            printf("  q %d: %5lld | %5lld | %5lld | %5lld   (iteration %lld)\n", q, fp[8 * q + 1] - fp[8 * q], q < 7 ? fp[8 * q + 2] - fp[8 * q + 1] : 0,
                   q < 7 ? fp[8 * q + 3] - fp[8 * q + 2] : 0, q < 7 ? fp[8 * q + 4] - fp[8 * q + 3] : 0, q < 7 ? fp[8 * q + 8] - fp[8 * q] : 0);
        printf("every wave, cycles since wave 0 left the iteration's barrier: [wave 4: export done | trailing done | inverse row done | solves done]\n");
        for (int q = 0; q < 7; ++q) {
            printf("  q %d:", q);
            for (int w = 1; w < 8; ++w) {
                const long long t0 = fp[8 * q];
                if (w == 4) printf("  w4 %5lld |", wp[(w * 8 + q) * 4] - t0);
                else printf("  w%d %5lld %5lld %5lld |", w, wp[(w * 8 + q) * 4 + 2] - t0, wp[(w * 8 + q) * 4 + 1] - t0, wp[(w * 8 + q) * 4 + 3] - t0);
            }
            printf("\n");
        }
    }
#else
    {
        long long fp[128];
        hipMemcpyFromSymbol(fp, HIP_SYMBOL(g_fprof), sizeof(fp));
        printf("wave 0 inside lds_factor_inv, cycles per 16-column step: [panel solve | wait | own tile update | chol16_lp of the next tile | wait]\n");
        for (int p = 0; p < 8; ++p)
            printf("  step %d: %5lld | %5lld | %5lld | %5lld | %5lld   (step total %lld)\n", p, fp[8 * p + 1] - fp[8 * p], fp[8 * p + 2] - fp[8 * p + 1],
                   fp[8 * p + 3] - fp[8 * p + 2], fp[8 * p + 4] - fp[8 * p + 3], fp[8 * p + 5] - fp[8 * p + 4], fp[8 * p + 5] - fp[8 * p]);
    }
    {
        long long wp[256], fp[128];
        hipMemcpyFromSymbol(wp, HIP_SYMBOL(g_wprof), sizeof(wp));
        hipMemcpyFromSymbol(fp, HIP_SYMBOL(g_fprof), sizeof(fp));
        printf("every wave, cycles since the barrier that opened the step's update phase: [export of block row p | row p of the inverse | trailing tiles | leaves the step's closing barrier]\n");
        for (int p = 0; p < 8; ++p) {
            printf("  step %d:", p);
            for (int w = 0; w < 8; ++w) {
                const long long t0 = fp[8 * p + 2];
                printf("  w%d %5lld %5lld %5lld %5lld |", w, wp[(w * 8 + p) * 4] - t0, wp[(w * 8 + p) * 4 + 1] - t0, wp[(w * 8 + p) * 4 + 2] - t0, wp[(w * 8 + p) * 4 + 3] - t0);
            }
            printf("\n");
        }
    }
#endif
    std::vector<double> L(n * n);
    hipMemcpy(L.data(), dA, n * n * 8, hipMemcpyDeviceToHost);
    // residual check L L^T - A
    double err = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) { double s = 0; for (int k = 0; k <= j; ++k) s += L[i * n + k] * L[j * n + k]; err = fmax(err, fabs(s - A[i * n + j])); }
    printf("max |LL^T - A| = %.3e\n", err);
    std::vector<double> Xi(n * n);
    hipMemcpy(Xi.data(), dinv, n * n * 8, hipMemcpyDeviceToHost);
    double e2 = 0;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s2 = 0; for (int k = 0; k < n; ++k) s2 += L[i * n + k] * Xi[k * n + j]; e2 = fmax(e2, fabs(s2 - (i == j))); }
    printf("max |L Linv - I| = %.3e\n", e2);
    {   // bit pattern of the two outputs (to compare schedules: the same operations in the same order give the same hash)
        unsigned long long hL = 1469598103934665603ull, hX = hL;
        for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) {
            unsigned long long b; memcpy(&b, &L[i * n + j], 8); hL = (hL ^ b) * 1099511628211ull;
            memcpy(&b, &Xi[i * n + j], 8); hX = (hX ^ b) * 1099511628211ull;
        }
        printf("hash L %016llx  hash Linv %016llx\n", hL, hX);
    }
    return 0;
}
