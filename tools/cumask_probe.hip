// cumask_probe.hip -- which compute units does a CU-masked stream leave free, per XCD?
// A long "hog" kernel fills the masked stream; a probe kernel on a second stream records where its
// workgroups ran (XCC_ID / HW_ID registers) and how long they waited.  Development aid.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <map>
#define GETREG(id, off, size) __builtin_amdgcn_s_getreg((id) | ((off) << 6) | (((size) - 1) << 11))

__global__ void hog(long long cycles, unsigned* where) {
    __shared__ char pad[60000];
    pad[threadIdx.x] = 0;
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) { }
    if (threadIdx.x == 0) where[blockIdx.x] = (GETREG(20, 0, 4) << 16) | GETREG(4, 0, 16);
}
__global__ void probe(long long cycles, unsigned* where, long long* when) {
    const long long t0 = clock64();
    while (clock64() - t0 < cycles) { }
    if (threadIdx.x == 0) { where[blockIdx.x] = (GETREG(20, 0, 4) << 16) | GETREG(4, 0, 16); when[blockIdx.x] = wall_clock64(); }
}

static void hist(const char* tag, const std::vector<unsigned>& w) {
    std::map<int, std::map<int, int>> cus;     // xcc -> (se,cu) -> count
    for (unsigned v : w) { const int xcc = v >> 16, cu = (v >> 8) & 0xF, se = (v >> 13) & 0x7; cus[xcc][se * 16 + cu]++; }
    printf("%s:", tag);
    for (auto& x : cus) printf("  xcc%d: %zu CUs/%d wgs", x.first, x.second.size(), [&] { int s = 0; for (auto& c : x.second) s += c.second; return s; }());
    printf("\n");
}

int main(int argc, char** argv) {
    const int reserved = argc > 1 ? atoi(argv[1]) : 16;
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int ncu = prop.multiProcessorCount;
    std::vector<uint32_t> mask((ncu + 31) / 32, 0u);
    for (int c = reserved; c < ncu; ++c) mask[c / 32] |= 1u << (c % 32);
    hipStream_t sa, sb;
    hipExtStreamCreateWithCUMask(&sa, mask.size(), mask.data());
    int lo, hi; hipDeviceGetStreamPriorityRange(&lo, &hi);
    hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, hi);
    const int nh = 4 * ncu, np = 2048;
    unsigned *wh, *wp; long long* tp;
    hipMalloc(&wh, nh * 4); hipMalloc(&wp, np * 4); hipMalloc(&tp, np * 8);
    hipLaunchKernelGGL(hog, dim3(nh), dim3(256), 0, sa, 20000000LL, wh);      // ~8 ms, 2 per CU by LDS
    hipLaunchKernelGGL(probe, dim3(64), dim3(64), 0, sb, 1000LL, wp, tp);      // warm
    hipStreamSynchronize(sb);
    long long t_launch;
    { hipEvent_t e; hipEventCreate(&e); }
    hipLaunchKernelGGL(probe, dim3(np), dim3(256), 0, sb, 24000LL, wp, tp);    // 10 us each
    hipDeviceSynchronize();
    std::vector<unsigned> h(nh), p(np); std::vector<long long> t(np);
    hipMemcpy(h.data(), wh, nh * 4, hipMemcpyDeviceToHost); hipMemcpy(p.data(), wp, np * 4, hipMemcpyDeviceToHost);
    hipMemcpy(t.data(), tp, np * 8, hipMemcpyDeviceToHost);
    printf("ncu=%d reserved=%d\n", ncu, reserved);
    hist("hog (masked stream)", h);
    hist("probe (while hog runs)", p);
    long long tmin = t[0], tmax = t[0];
    for (auto v : t) { tmin = v < tmin ? v : tmin; tmax = v > tmax ? v : tmax; }
    printf("probe workgroups finished over %.1f us (wall clock 100 MHz)\n", (tmax - tmin) / 100.0);
    // per-xcc finishing time of the probe
    std::map<int, long long> last;
    for (int i = 0; i < np; ++i) { const int x = p[i] >> 16; if (t[i] > last[x]) last[x] = t[i]; }
    for (auto& x : last) printf("  xcc%d last probe wg at +%.1f us\n", x.first, (x.second - tmin) / 100.0);
    return 0;
}
