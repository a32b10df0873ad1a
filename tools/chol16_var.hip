// chol16_var.hip -- cycle counts of the one-wave 16x16 Cholesky alone, with the inverse as a by-product (xinv) and
// with its panels published for a helper wave (pub), one wave on an otherwise idle CU.
#include "../gpim_amd/csrc/blocklds.hpp"
#include <stdio.h>
#include <vector>
__global__ void kv(const double* A, long long* cyc, double* sink) {
    __shared__ double D[3][16 * LDD];
    __shared__ double invd[16];
    __shared__ double pub[CHOL16_PUB_DOUBLES];
    const int lane = threadIdx.x;
    for (int v = 0; v < 3; ++v) for (int e = lane; e < 256; e += 64) D[v][(e >> 4) * LDD + (e & 15)] = A[e];
    __syncthreads();
    long long t0 = clock64();
    int bad = chol16(D[0], invd, lane);
    __syncthreads();
    long long t1 = clock64();
    d4 x;
    bad += chol16(D[1], invd, lane, &x);
    __syncthreads();
    long long t2 = clock64();
    bad += chol16(D[2], invd, lane, nullptr, pub);
    __syncthreads();
    long long t3 = clock64();
    if (lane == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; cyc[2] = t3 - t2; }
    sink[lane] = x[0] + x[1] + x[2] + x[3] + bad + D[0][lane] + D[2][lane];
}
int main() {
    std::vector<double> A(256);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) A[i * 16 + j] = (i == j ? 2.0 : 0.0) + 1.0 / (1 + abs(i - j));
    double *dA, *ds; long long* dc;
    hipMalloc(&dA, 2048); hipMalloc(&ds, 512); hipMalloc(&dc, 64);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kv, dim3(1), dim3(64), 0, 0, dA, dc, ds);
    hipDeviceSynchronize();
    long long c[3]; hipMemcpy(c, dc, 24, hipMemcpyDeviceToHost);
    printf("chol16 alone %lld cycles | + inverse by-product %lld | + published panels %lld\n", c[0], c[1], c[2]);
    return 0;
}
