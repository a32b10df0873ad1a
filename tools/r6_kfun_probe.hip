// r6_kfun_probe.hip -- kf_exp_neg / kf_sqrt (gpim_amd/csrc/kfun.hpp, round 6) against the device library's exp / sqrt:
// ulp differences over the argument ranges the covariance functions produce, and special values.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Igpim_amd/csrc tools/r6_kfun_probe.hip -o tools/r6_kfun_probe
#include "../gpim_amd/csrc/kfun.hpp"
#include <stdio.h>
#include <string>
#include <vector>
#include <math.h>
void gpim_set_error(const std::string&) {}
__device__ long long ulps(double a, double b) {
    long long x, y;
    memcpy(&x, &a, 8); memcpy(&y, &b, 8);
    return x > y ? x - y : y - x;
}
__global__ void probe(int n, double lo, double hi, int what, long long* maxd, double* sumd, double* worst) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // log-spaced magnitudes
    const double t = (double)i / (double)(n - 1);
    const double mag = lo * pow(hi / lo, t);
    double ref, got, arg;
    if (what == 0) { arg = -mag; ref = exp(arg); got = kf_exp_neg(arg); }
    else { arg = mag; ref = sqrt(arg); got = kf_sqrt(arg); }
    const long long d = (ref == 0.0 && got == 0.0) ? 0 : ulps(ref, got);
    atomicMax((unsigned long long*)maxd, (unsigned long long)d);
    atomicAdd(sumd, (double)d);
    if (d > 2) *worst = arg;
}
__global__ void specials(double* out) {
    out[0] = kf_exp_neg(0.0); out[1] = kf_exp_neg(-0.0); out[2] = kf_exp_neg(-745.2); out[3] = kf_exp_neg(-800.0);
    out[4] = kf_exp_neg(-1e300); out[5] = kf_exp_neg(nan("")); out[6] = kf_sqrt(1e-12); out[7] = kf_sqrt(nan(""));
    out[8] = kf_exp_neg(-708.0); out[9] = exp(-708.0); out[10] = kf_exp_neg(-740.0); out[11] = exp(-740.0);
}
int main() {
    long long* maxd; double *sumd, *worst, *sp;
    hipMalloc(&maxd, 8); hipMalloc(&sumd, 8); hipMalloc(&worst, 8); hipMalloc(&sp, 16 * 8);
    const int n = 20000000;
    struct { const char* name; int what; double lo, hi; } cases[] = {
        {"exp(-x), x in [1e-12, 1e-3]", 0, 1e-12, 1e-3}, {"exp(-x), x in [1e-3, 40]", 0, 1e-3, 40.0},
        {"exp(-x), x in [40, 700]", 0, 40.0, 700.0}, {"sqrt(x), x in [1e-12, 1]", 1, 1e-12, 1.0}, {"sqrt(x), x in [1, 1e8]", 1, 1.0, 1e8}};
    for (auto& c : cases) {
        hipMemset(maxd, 0, 8); hipMemset(sumd, 0, 8); hipMemset(worst, 0, 8);
        hipLaunchKernelGGL(probe, dim3((n + 255) / 256), dim3(256), 0, 0, n, c.lo, c.hi, c.what, maxd, sumd, worst);
        long long m; double s, w;
        hipMemcpy(&m, maxd, 8, hipMemcpyDeviceToHost); hipMemcpy(&s, sumd, 8, hipMemcpyDeviceToHost); hipMemcpy(&w, worst, 8, hipMemcpyDeviceToHost);
        printf("%-32s max %lld ulp, mean %.4f ulp%s\n", c.name, m, s / n, m > 2 ? " (worst argument above)" : "");
        if (m > 2) printf("    an argument with > 2 ulp: %.17g\n", w);
    }
    hipLaunchKernelGGL(specials, dim3(1), dim3(1), 0, 0, sp);
    double h[16]; hipMemcpy(h, sp, 16 * 8, hipMemcpyDeviceToHost);
    printf("exp(0) %g exp(-0) %g exp(-745.2) %g exp(-800) %g exp(-1e300) %g exp(nan) %g sqrt(1e-12) %.17g sqrt(nan) %g\n", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
    printf("exp(-708): %.17g vs library %.17g; exp(-740) (denormal): %.17g vs %.17g\n", h[8], h[9], h[10], h[11]);
    return 0;
}
