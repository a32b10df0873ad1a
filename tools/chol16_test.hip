// chol16_test.hip -- accuracy and cycle count of the one-wave 16x16 Cholesky / triangular inverse
// of blocklds.hpp against a host long-double reference (development aid).
#include "../gpim_amd/csrc/blocklds.hpp"
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <vector>

__global__ void k16(const double* A, double* Lout, double* Xout, double* invd_out, int* bad_out, long long* cyc) {
    __shared__ double D[16 * LDD];
    __shared__ double X[16 * LDD];
    __shared__ double invd[16];
    const int lane = threadIdx.x;
    for (int e = lane; e < 256; e += 64) D[(e >> 4) * LDD + (e & 15)] = A[e];
    __syncthreads();
    const long long t0 = clock64();
    const int bad = chol16(D, invd, lane);
    __syncthreads();
    const long long t1 = clock64();
    trinv16(D, LDD, invd, X, LDD, lane);
    __syncthreads();
    const long long t2 = clock64();
    for (int e = lane; e < 256; e += 64) {
        Lout[e] = D[(e >> 4) * LDD + (e & 15)];
        Xout[e] = X[(e >> 4) * LDD + (e & 15)];
    }
    if (lane < 16) invd_out[lane] = invd[lane];
    if (lane == 0) { *bad_out = bad; cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}

static void run(const char* name, const std::vector<double>& A) {
    double *dA, *dL, *dX, *dI; int* dbad; long long* dc;
    hipMalloc(&dA, 2048); hipMalloc(&dL, 2048); hipMalloc(&dX, 2048); hipMalloc(&dI, 128); hipMalloc(&dbad, 4); hipMalloc(&dc, 16);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, dA, dL, dX, dI, dbad, dc);
    hipDeviceSynchronize();
    std::vector<double> L(256), X(256), I(16); int bad; long long c[2];
    hipMemcpy(L.data(), dL, 2048, hipMemcpyDeviceToHost); hipMemcpy(X.data(), dX, 2048, hipMemcpyDeviceToHost);
    hipMemcpy(I.data(), dI, 128, hipMemcpyDeviceToHost); hipMemcpy(&bad, dbad, 4, hipMemcpyDeviceToHost); hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
    // host reference
    long double R[16][16] = {};
    int hbad = 0;
    for (int j = 0; j < 16; ++j) {
        long double d = A[j * 16 + j];
        for (int k = 0; k < j; ++k) d -= R[j][k] * R[j][k];
        if (!(d > 0) && !hbad) hbad = j + 1;
        R[j][j] = sqrtl(d);
        for (int i = j + 1; i < 16; ++i) {
            long double s = A[i * 16 + j];
            for (int k = 0; k < j; ++k) s -= R[i][k] * R[j][k];
            R[i][j] = s / R[j][j];
        }
    }
    double eL = 0, eI = 0, eRes = 0, eX = 0, up = 0, nL = 0;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j <= i; ++j) { eL = fmax(eL, fabs((double)(L[i * 16 + j] - R[i][j]))); nL = fmax(nL, fabs((double)R[i][j])); }
    for (int i = 0; i < 16; ++i) eI = fmax(eI, fabs(I[i] * L[i * 16 + i] - 1.0));
    // residual L L^T - A (lower), X L - I
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j <= i; ++j) {
            long double s = 0;
            for (int k = 0; k <= j; ++k) s += (long double)L[i * 16 + k] * L[j * 16 + k];
            eRes = fmax(eRes, fabs((double)(s - A[i * 16 + j])));
        }
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            long double s = 0;
            for (int k = 0; k < 16; ++k) s += (long double)X[i * 16 + k] * ((k >= j) ? L[k * 16 + j] : 0.0);
            eX = fmax(eX, fabs((double)(s - (i == j ? 1.0L : 0.0L))));
            if (j > i) up = fmax(up, fabs(X[i * 16 + j]));
        }
    printf("%-22s bad dev %d host %d | max|L-Lref| %.2e (|L| %.2e) | |LL^T-A| %.2e | invd*Ljj-1 %.2e | |XL-I| %.2e upper(X) %.1e | cycles chol16 %lld trinv16 %lld\n",
           name, bad, hbad, eL, nL, eRes, eI, eX, up, c[0], c[1]);
    hipFree(dA); hipFree(dL); hipFree(dX); hipFree(dI); hipFree(dbad); hipFree(dc);
}

int main() {
    std::vector<double> A(256), B(256);
    srand(3);
    for (auto& v : B) v = rand() / (double)RAND_MAX - 0.5;
    for (int i = 0; i < 16; ++i)
        for (int j = 0; j < 16; ++j) {
            double s = (i == j) ? 1.0 : 0.0;
            for (int k = 0; k < 16; ++k) s += B[i * 16 + k] * B[j * 16 + k];
            A[i * 16 + j] = s;
        }
    run("random SPD", A);
    // lower-only storage: garbage above the diagonal must be ignored
    std::vector<double> Al = A;
    for (int i = 0; i < 16; ++i) for (int j = i + 1; j < 16; ++j) Al[i * 16 + j] = 1e300;
    run("lower only", Al);
    for (double noise : {1e-2, 1e-6, 1e-10}) {
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) { const double d = 0.13 * (i - j); A[i * 16 + j] = exp(-0.5 * d * d) + (i == j ? noise : 0.0); }
        char nm[64]; snprintf(nm, 64, "RBF ls=7.7 noise=%.0e", noise);
        run(nm, A);
    }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) A[i * 16 + j] = (i == j) ? 1.0 : 0.0;
    A[9 * 16 + 9] = -1.0;
    run("not PD at col 10", A);
    return 0;
}
