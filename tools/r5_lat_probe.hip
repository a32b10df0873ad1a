// r5_lat_probe.hip -- dependent-issue latencies of the fp64 instructions a lane-parallel 16x16 factorisation is made of
// (one wave alone on a CU), the accuracy of the v_rcp_f64 / v_rsq_f64 seeds, and the cycle count + accuracy of
// chol16_lp (chol16lp.hpp) beside the MFMA formulation chol16 (blocklds.hpp).
#include "../gpim_amd/csrc/blocklds.hpp"
#include "../gpim_amd/csrc/chol16lp.hpp"
#include <stdio.h>
#include <math.h>
#include <stdlib.h>
#include <vector>

#define REP4(...) __VA_ARGS__ __VA_ARGS__ __VA_ARGS__ __VA_ARGS__
#define REP16(...) REP4(REP4(__VA_ARGS__))
#define CHAIN(name, ...)                                                               \
    {                                                                                  \
        __syncthreads();                                                               \
        const long long t0 = clock64();                                                \
        for (int it = 0; it < 16; ++it) { REP16(__VA_ARGS__) }                               \
        const long long t1 = clock64();                                                \
        if (threadIdx.x == 0) cyc[slot] = t1 - t0;                                     \
        ++slot;                                                                        \
    }

__global__ void lat_kernel(const double* in, long long* cyc, double* sink) {
    int slot = 0;
    double x = in[threadIdx.x], c = in[64 + threadIdx.x], y = in[128 + threadIdx.x];
    double a0 = x, a1 = x + 1, a2 = x + 2, a3 = x + 3, a4 = x + 4, a5 = x + 5, a6 = x + 6, a7 = x + 7;
    // 0: dependent v_fma_f64
    CHAIN(fma, asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(y));)
    // 1: dependent v_mul_f64
    CHAIN(mul, asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(c));)
    // 2: dependent v_rcp_f64
    CHAIN(rcp, asm volatile("v_rcp_f64 %0, %0" : "+v"(x));)
    // 3: dependent v_rsq_f64
    CHAIN(rsq, asm volatile("v_rsq_f64 %0, %0" : "+v"(x));)
    // 4: dependent v_mov_b64_dpp row_newbcast (2 wait states needed: s_nop 1)
    CHAIN(dppmov, asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x));)
    // 5: v_fmac_f64_dpp, dependent through the accumulator only
    CHAIN(fmacdpp_acc, asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(c), "v"(y));)
    // 6: v_fmac_f64_dpp, dependent through the DPP source
    CHAIN(fmacdpp_src, asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(c));)
    // 7: 8 independent v_fmac_f64_dpp (throughput; 16 x 16 x 8 instructions)
    CHAIN(fmacdpp_tp,
          asm volatile("v_fmac_f64_dpp %0, %8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                       "v_fmac_f64_dpp %1, %8, %9 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                       "v_fmac_f64_dpp %2, %8, %9 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                       "v_fmac_f64_dpp %3, %8, %9 row_newbcast:6 row_mask:0xf bank_mask:0xf\n\t"
                       "v_fmac_f64_dpp %4, %8, %9 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\t"
                       "v_fmac_f64_dpp %5, %8, %9 row_newbcast:8 row_mask:0xf bank_mask:0xf\n\t"
                       "v_fmac_f64_dpp %6, %8, %9 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                       "v_fmac_f64_dpp %7, %8, %9 row_newbcast:10 row_mask:0xf bank_mask:0xf"
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                       : "v"(c), "v"(y));)
    // 8: 8 independent v_fma_f64
    CHAIN(fma_tp,
          asm volatile("v_fma_f64 %0, %8, %9, %0\n\tv_fma_f64 %1, %8, %9, %1\n\tv_fma_f64 %2, %8, %9, %2\n\tv_fma_f64 %3, %8, %9, %3\n\t"
                       "v_fma_f64 %4, %8, %9, %4\n\tv_fma_f64 %5, %8, %9, %5\n\tv_fma_f64 %6, %8, %9, %6\n\tv_fma_f64 %7, %8, %9, %7"
                       : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                       : "v"(c), "v"(y));)
    // 9: readlane -> fma with the scalar -> readlane ...
    CHAIN(readlane,
          {
              int lo, hi;
              asm volatile("v_readlane_b32 %0, %2, 3\n\tv_readlane_b32 %1, %3, 3"
                           : "=s"(lo), "=s"(hi)
                           : "v"(__double2loint(x)), "v"(__double2hiint(x)));
              const double s = __hiloint2double(hi, lo);
              asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(x) : "s"(s), "v"(c));
          })
    // 10: MFMA chain through the accumulator
    d4 acc = (d4){x, x, x, x};
    CHAIN(mfma_acc, acc = __builtin_amdgcn_mfma_f64_16x16x4f64(c, y, acc, 0, 0, 0); asm volatile("" : "+v"(acc));)
    // 11: MFMA chain through the A operand
    CHAIN(mfma_op, acc = __builtin_amdgcn_mfma_f64_16x16x4f64(acc[0], y, (d4){0.0, 0.0, 0.0, 0.0}, 0, 0, 0); asm volatile("" : "+v"(acc));)
    // 12: f64 -> f32 rcp -> f64
    CHAIN(rcp32, {
        float f;
        asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f) : "v"(x));
        asm volatile("v_rcp_f32 %0, %0" : "+v"(f));
        asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(x) : "v"(f));
    })
    // 13: cndmask pair
    int xi = (int)threadIdx.x, ci = xi + 3;
    CHAIN(cndmask, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(xi) : "v"(ci) : );)
    // 14: 8 independent v_mov_b64_dpp + 8 independent v_fma_f64 (the unfused form of the update)
    CHAIN(movdpp_fma_tp,
          {
              double b0, b1, b2, b3;
              asm volatile("v_mov_b64_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                           "v_mov_b64_dpp %1, %4 row_newbcast:4 row_mask:0xf bank_mask:0xf\n\t"
                           "v_mov_b64_dpp %2, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                           "v_mov_b64_dpp %3, %4 row_newbcast:6 row_mask:0xf bank_mask:0xf"
                           : "=v"(b0), "=v"(b1), "=v"(b2), "=v"(b3)
                           : "v"(c));
              asm volatile("v_fma_f64 %0, %4, %8, %0\n\tv_fma_f64 %1, %5, %8, %1\n\tv_fma_f64 %2, %6, %8, %2\n\tv_fma_f64 %3, %7, %8, %3"
                           : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3)
                           : "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(y));
          })
    // 15: ds write + read round trip
    __shared__ double lds[128];
    CHAIN(lds_rt, {
        lds[threadIdx.x] = x;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        x = lds[threadIdx.x ^ 1];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    })
    sink[threadIdx.x] = xi + x + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + acc[0] + acc[1];
}

__global__ void acc_kernel(const double* in, double* out, int n) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const double d = in[t];
    out[t] = __builtin_amdgcn_rcp(d);
    out[n + t] = __builtin_amdgcn_rsq(d);
    {
        const double y = __builtin_amdgcn_rcp(d), e = fma(-d, y, 1.0), pp = fma(e, e, e);
        out[2 * n + t] = fma(y, pp, y);
    }
    out[3 * n + t] = lp_rsqrt(d);
}

// cycles + accuracy of the two 16x16 factorisations
template <class Lay, int ROWS, bool SCALED>
__global__ void k16(const double* A, double* Lout, double* Xout, double* invd_out, int* bad_out, long long* cyc) {
    __shared__ __attribute__((aligned(16))) double D[2][ROWS];
    __shared__ __attribute__((aligned(16))) double X[2][16 * XS_LD];
    __shared__ double invd[2][16];
    const int lane = threadIdx.x;
    for (int v = 0; v < 2; ++v)
        for (int e = lane; e < 256; e += 64) D[v][Lay::in(e >> 4, e & 15)] = A[e];
    __syncthreads();
    const long long t0 = clock64();
    d4 x0;
    const int bad0 = chol16<Lay>(D[0], invd[0], lane, &x0);
    xs_write(X[0], x0, lane);
    __syncthreads();
    const long long t1 = clock64();
    const int bad1 = chol16_lp<Lay, SCALED>(D[1], invd[1], lane, X[1], XS_LD);
    __syncthreads();
    const long long t2 = clock64();
    for (int v = 0; v < 2; ++v)
        for (int e = lane; e < 256; e += 64) {
            Lout[v * 256 + e] = D[v][Lay::in(e >> 4, e & 15)];
            Xout[v * 256 + e] = X[v][(e >> 4) * XS_LD + (e & 15)];
        }
    if (lane < 16) { invd_out[lane] = invd[0][lane]; invd_out[16 + lane] = invd[1][lane]; }
    if (lane == 0) { bad_out[0] = bad0; bad_out[1] = bad1; cyc[0] = t1 - t0; cyc[1] = t2 - t1; }
}

template <class Lay, int ROWS, bool SCALED = true>
static void run(const char* name, const std::vector<double>& A) {
    double *dA, *dL, *dX, *dI; int* dbad; long long* dc;
    hipMalloc(&dA, 2048); hipMalloc(&dL, 4096); hipMalloc(&dX, 4096); hipMalloc(&dI, 256); hipMalloc(&dbad, 8); hipMalloc(&dc, 16);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k16<Lay, ROWS, SCALED>), dim3(1), dim3(64), 0, 0, dA, dL, dX, dI, dbad, dc);
    hipDeviceSynchronize();
    std::vector<double> Lb(512), Xb(512), Ib(32); int bad[2]; long long c[2];
    hipMemcpy(Lb.data(), dL, 4096, hipMemcpyDeviceToHost); hipMemcpy(Xb.data(), dX, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(Ib.data(), dI, 256, hipMemcpyDeviceToHost); hipMemcpy(bad, dbad, 8, hipMemcpyDeviceToHost); hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
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
    if (!SCALED)
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) Lb[256 + i * 16 + j] = (j <= i) ? Lb[256 + i * 16 + j] * Ib[16 + j] : 0.0;
    for (int v = 0; v < 2; ++v) {
        const double *L = Lb.data() + 256 * v, *X = Xb.data() + 256 * v, *I = Ib.data() + 16 * v;
        double eL = 0, eI = 0, eRes = 0, eX = 0, up = 0, upL = 0, nL = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j <= i; ++j) { eL = fmax(eL, fabs((double)(L[i * 16 + j] - R[i][j]))); nL = fmax(nL, fabs((double)R[i][j])); }
        for (int i = 0; i < 16; ++i) eI = fmax(eI, fabs(I[i] * L[i * 16 + i] - 1.0));
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
                if (j > i) { up = fmax(up, fabs(X[i * 16 + j])); if (v == 1) upL = fmax(upL, fabs(L[i * 16 + j])); }
            }
        printf("%-22s %-9s bad dev %d host %d | max|L-Lref| %.2e (|L| %.2e) | |LL^T-A| %.2e | invd*Ljj-1 %.2e | |XL-I| %.2e upper(X) %.1e upper(L) %.1e | cycles %lld\n",
               name, v ? "chol16_lp" : "chol16", bad[v], hbad, eL, nL, eRes, eI, eX, up, upL, c[v]);
    }
    hipFree(dA); hipFree(dL); hipFree(dX); hipFree(dI); hipFree(dbad); hipFree(dc);
}

int main() {
    {
        std::vector<double> in(192);
        for (int i = 0; i < 64; ++i) { in[i] = 1.0 + 1e-3 * i; in[64 + i] = 0.999; in[128 + i] = 1e-3; }
        double *din, *ds; long long* dc;
        hipMalloc(&din, 192 * 8); hipMalloc(&ds, 512); hipMalloc(&dc, 32 * 8);
        hipMemcpy(din, in.data(), 192 * 8, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(lat_kernel, dim3(1), dim3(64), 0, 0, din, dc, ds);
        hipDeviceSynchronize();
        long long c[32]; hipMemcpy(c, dc, sizeof(c), hipMemcpyDeviceToHost);
        const char* names[] = {"v_fma_f64 dependent", "v_mul_f64 dependent", "v_rcp_f64 dependent", "v_rsq_f64 dependent",
                               "s_nop 1 + v_mov_b64_dpp dependent", "v_fmac_f64_dpp (acc chain)", "s_nop 1 + v_fmac_f64_dpp (src chain)",
                               "v_fmac_f64_dpp x8 independent (per instr)", "v_fma_f64 x8 independent (per instr)",
                               "2 readlane + v_fma (per round)", "mfma f64 16x16x4 acc chain", "mfma f64 16x16x4 operand chain",
                               "cvt + rcp_f32 + cvt (per round)", "v_cndmask_b32 dependent", "4 mov_dpp + 4 fma independent (per pair)",
                               "ds_write + ds_read round trip"};
        const double per[] = {256, 256, 256, 256, 256, 256, 256, 2048, 2048, 256, 256, 256, 256, 256, 1024, 256};
        for (int i = 0; i < 16; ++i) printf("  %-44s %8.1f cycles\n", names[i], c[i] / per[i]);
    }
    {
        const int n = 4096;
        std::vector<double> in(n), out(4 * n);
        srand(1);
        for (int i = 0; i < n; ++i) in[i] = ldexp(0.5 + 0.5 * rand() / RAND_MAX, (rand() % 40) - 20);
        double *din, *dout; hipMalloc(&din, n * 8); hipMalloc(&dout, 4 * n * 8);
        hipMemcpy(din, in.data(), n * 8, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(acc_kernel, dim3(n / 256), dim3(256), 0, 0, din, dout, n);
        hipMemcpy(out.data(), dout, 4 * n * 8, hipMemcpyDeviceToHost);
        double e[4] = {0, 0, 0, 0};
        for (int i = 0; i < n; ++i) {
            const long double r = 1.0L / in[i], s = 1.0L / sqrtl((long double)in[i]);
            e[0] = fmax(e[0], fabs((double)((out[i] - r) / r)));
            e[1] = fmax(e[1], fabs((double)((out[n + i] - s) / s)));
            e[2] = fmax(e[2], fabs((double)((out[2 * n + i] - r) / r)));
            e[3] = fmax(e[3], fabs((double)((out[3 * n + i] - s) / s)));
        }
        printf("  max rel error: v_rcp_f64 %.3e  v_rsq_f64 %.3e  lp_rcp %.3e  lp_rsqrt %.3e  (2^-52 = %.3e)\n", e[0], e[1], e[2], e[3], ldexp(1.0, -52));
    }
    std::vector<double> A(256), B(256);
    srand(3);
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) A[i * 16 + j] = (i == j ? 2.0 : 0.0) + 1.0 / (1 + abs(i - j));
    run<LayPad, 16 * LDD>("toeplitz pad", A);
    run<LayTri, 256>("toeplitz tri", A);
    run<LayTri, 256, false>("toeplitz tri unscaled", A);
    // random SPD, badly scaled
    std::vector<double> G(256);
    for (auto& g : G) g = (double)rand() / RAND_MAX - 0.5;
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double s = 0; for (int k = 0; k < 16; ++k) s += G[i * 16 + k] * G[j * 16 + k];
        B[i * 16 + j] = s * pow(10.0, (i + j) * 0.2) + (i == j ? 1e-6 : 0.0);
    }
    run<LayTri, 256>("random scaled", B);
    // kernel-matrix like: near-singular RBF + jitter
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) A[i * 16 + j] = exp(-0.5 * (i - j) * (i - j) / 25.0) + (i == j ? 1e-5 : 0.0);
    run<LayTri, 256>("rbf l=5 +1e-5", A);
    run<LayTri, 256, false>("rbf unscaled", A);
    // not positive definite at column 7
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) A[i * 16 + j] = (i == j ? 2.0 : 0.0) + 1.0 / (1 + abs(i - j));
    A[6 * 16 + 6] = -1.0;
    run<LayTri, 256>("non-PD col 7", A);
    return 0;
}
