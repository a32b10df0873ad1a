// factor_inv_test.hip -- lds_factor_inv (blocklds.hpp) against a host reference, block by block.
#include "../gpim_amd/csrc/blocklds.hpp"
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <vector>
__global__ __launch_bounds__(NTH, 1) void k(const double* A, double* Lout, double* Xout, int npan, long long* cyc) {
    __shared__ __attribute__((aligned(16))) double D[NB * LDD];
    __shared__ double invd[NB];
    __shared__ __attribute__((aligned(16))) double Xs[32 * XS_LD];
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    const int ns = npan * 16;
    for (int e = tid; e < ns * ns; e += NTH) D[(e / ns) * LDD + (e % ns)] = A[e];
    if (tid == 0) s_bad = 0;
    __syncthreads();
    const long long t0 = clock64();
    struct Sink {
        double *Lout, *D, *invd;
        int ns;
        __device__ void tile(int t, int p, d4 acc, int lane) const {
            for (int g = 0; g < 4; ++g) Lout[(16 * t + (lane & 15)) * ns + 16 * p + 8 * (g >> 1) + 2 * (lane >> 4) + (g & 1)] = acc[g];
        }
        __device__ void row(int i, int lane) const {
            for (int e = lane; e < 16 * ns; e += 64) {
                const int r = i * 16 + e / ns, c = e % ns;
                // (diagonal tile: column c still times L_cc, chol16lp.hpp)
                if ((c >> 4) == i) Lout[r * ns + c] = (c <= r) ? D[r * LDD + c] * invd[c] : 0.0;
                else if ((c >> 4) == i - 1) Lout[r * ns + c] = D[r * LDD + c];
                else if ((c >> 4) > i) Lout[r * ns + c] = 0.0;
            }
        }
    };
    lds_factor_inv(D, invd, Xs, npan, &s_bad, tid, Sink{Lout, D, invd, ns});
    const long long t1 = clock64();
    for (int e = tid; e < ns * ns; e += NTH) Xout[e] = ((e % ns) <= (e / ns)) ? D[(e / ns) * LDD + (e % ns)] : 0.0;
    if (tid == 0) cyc[0] = t1 - t0;
}
int main() {
    for (int npan : {1, 2, 3, 5, 8}) {
        const int n = npan * 16;
        std::vector<double> A(n * n), B(n * n), L(n * n), X(n * n);
        srand(npan);
        for (auto& v : B) v = rand() / (double)RAND_MAX - 0.5;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = (i == j) ? 1.0 : 0.0; for (int q = 0; q < n; ++q) s += B[i * n + q] * B[j * n + q]; A[i * n + j] = s; }
        double *dA, *dL, *dX; long long* dc;
        hipMalloc(&dA, n * n * 8); hipMalloc(&dL, n * n * 8); hipMalloc(&dX, n * n * 8); hipMalloc(&dc, 8);
        hipMemcpy(dA, A.data(), n * n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(NTH), 0, 0, dA, dL, dX, npan, dc);
        hipDeviceSynchronize();
        hipMemcpy(L.data(), dL, n * n * 8, hipMemcpyDeviceToHost); hipMemcpy(X.data(), dX, n * n * 8, hipMemcpyDeviceToHost);
        long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
        // host reference
        std::vector<long double> R(n * n, 0);
        for (int j = 0; j < n; ++j) {
            long double d = A[j * n + j]; for (int q = 0; q < j; ++q) d -= R[j * n + q] * R[j * n + q];
            R[j * n + j] = sqrtl(d);
            for (int i = j + 1; i < n; ++i) { long double s = A[i * n + j]; for (int q = 0; q < j; ++q) s -= R[i * n + q] * R[j * n + q]; R[i * n + j] = s / R[j * n + j]; }
        }
        printf("npan=%d cycles %lld\n", npan, c);
        for (int bi = 0; bi < npan; ++bi) {
            printf("  row %d: L err", bi);
            for (int bj = 0; bj <= bi; ++bj) { double e = 0; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { int r = bi * 16 + i, cc = bj * 16 + j; if (cc <= r) e = fmax(e, fabs((double)(L[r * n + cc] - R[r * n + cc]))); } printf(" %.1e", e); }
            printf(" | (X L - I) err");
            for (int bj = 0; bj <= bi; ++bj) { double e = 0; for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) { int r = bi * 16 + i, cc = bj * 16 + j; long double s = 0; for (int q = 0; q < n; ++q) s += (long double)X[r * n + q] * R[q * n + cc]; e = fmax(e, fabs((double)(s - (r == cc ? 1.0L : 0.0L)))); } printf(" %.1e", e); }
            printf("\n");
        }
        hipFree(dA); hipFree(dL); hipFree(dX); hipFree(dc);
    }
    return 0;
}
