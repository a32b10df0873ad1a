// jacobi_prof.hip -- where does one round-robin step of the one-sided Jacobi eigen-solver (csrc/kron.hip) spend its
// time?  Phase cycle counts (s_memtime) of a step at n = 64 with the rows in global memory vs in LDS.  Development aid.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <math.h>
#include <vector>
#define W16 16
__device__ __forceinline__ double wsum(double v) {
    v += __shfl_xor(v, 32); v += __shfl_xor(v, 16); v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    return v;
}
template <int INLDS>
__global__ __launch_bounds__(1024) void sweep(double* Wg, double* Vg, int n, int nsweep, long long* prof) {
    extern __shared__ double sm[];
    double* W = INLDS ? sm : Wg;
    double* V = INLDS ? sm + n * n : Vg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (INLDS) { for (int e = tid; e < n * n; e += 1024) { W[e] = Wg[e]; V[e] = Vg[e]; } __syncthreads(); }
    const int ring = n - 1, half = n / 2;
    long long t_load = 0, t_red = 0, t_rot = 0, t_store = 0, t_bar = 0;
    for (int sw = 0; sw < nsweep; ++sw)
        for (int step = 0; step < ring; ++step) {
            for (int k = wave; k < half; k += W16) {
                const int i = (k == 0) ? ring : (step + k) % ring, j = (step + ring - k) % ring;
                long long c0 = clock64();
                const double x = lane < n ? W[i * n + lane] : 0.0, y = lane < n ? W[j * n + lane] : 0.0;
                const double p = lane < n ? V[i * n + lane] : 0.0, q = lane < n ? V[j * n + lane] : 0.0;
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                long long c1 = clock64();
                const double a = wsum(x * x), b = wsum(y * y), c = wsum(x * y);
                long long c2 = clock64();
                double cs = 1.0, sn = 0.0;
                if (fabs(c) > 1e-15 * sqrt(a * b)) {
                    const double zeta = (b - a) / (2.0 * c);
                    const double t = (zeta >= 0.0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                    cs = 1.0 / sqrt(1.0 + t * t); sn = cs * t;
                }
                long long c3 = clock64();
                if (lane < n) {
                    W[i * n + lane] = cs * x - sn * y; W[j * n + lane] = sn * x + cs * y;
                    V[i * n + lane] = cs * p - sn * q; V[j * n + lane] = sn * p + cs * q;
                }
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                long long c4 = clock64();
                t_load += c1 - c0; t_red += c2 - c1; t_rot += c3 - c2; t_store += c4 - c3;
            }
            long long c5 = clock64();
            __syncthreads();
            t_bar += clock64() - c5;
        }
    if (tid == 0) { prof[0] = t_load; prof[1] = t_red; prof[2] = t_rot; prof[3] = t_store; prof[4] = t_bar; }
    if (INLDS) { __syncthreads(); for (int e = tid; e < n * n; e += 1024) { Wg[e] = W[e]; Vg[e] = V[e]; } }
}
int main() {
    const int n = 64, nsweep = 8;
    std::vector<double> K(n * n), I(n * n, 0.0);
    for (int a = 0; a < n; ++a) for (int b = 0; b < n; ++b) { K[a * n + b] = exp(-0.5 * (a - b) * (a - b) / 100.0); I[a * n + a] = 1.0; }
    double *W, *V; long long* prof;
    hipMalloc(&W, n * n * 8); hipMalloc(&V, n * n * 8); hipMalloc(&prof, 64);
    for (int lds = 0; lds < 2; ++lds) {
        hipMemcpy(W, K.data(), n * n * 8, hipMemcpyHostToDevice); hipMemcpy(V, I.data(), n * n * 8, hipMemcpyHostToDevice);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        if (lds) hipLaunchKernelGGL(sweep<1>, dim3(1), dim3(1024), 2 * n * n * 8, 0, W, V, n, nsweep, prof);
        else hipLaunchKernelGGL(sweep<0>, dim3(1), dim3(1024), 0, 0, W, V, n, nsweep, prof);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long p[5]; hipMemcpy(p, prof, 40, hipMemcpyDeviceToHost);
        const double steps = nsweep * (n - 1), pairs = steps * 2;
        printf("%s: %.3f ms, %.2f us per step; wave 0 cycles per pair: load %.0f reduce %.0f rotation %.0f store %.0f; barrier wait per step %.0f\n",
               lds ? "rows in LDS   " : "rows in global", ms, ms * 1e3 / steps, p[0] / pairs, p[1] / pairs, p[2] / pairs, p[3] / pairs, p[4] / steps);
    }
    return 0;
}
