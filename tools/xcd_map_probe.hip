// xcd_map_probe.hip -- does workgroup b of a 1-D grid run on XCD b % 8?  (the tile engine's
// blockIdx -> tile remap relies on it).  Development aid.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define GETREG(id, off, size) __builtin_amdgcn_s_getreg((id) | ((off) << 6) | (((size) - 1) << 11))
__global__ void where(unsigned* w, long long spin) {
    __shared__ char pad[70000];
    pad[threadIdx.x] = 0;
    const long long t0 = clock64();
    while (clock64() - t0 < spin) { }
    if (threadIdx.x == 0) w[blockIdx.x] = GETREG(20, 0, 4);
}
int main() {
    for (int n : {512, 4096, 8256}) {
        unsigned* w; hipMalloc(&w, n * 4);
        hipLaunchKernelGGL(where, dim3(n), dim3(256), 0, 0, w, 200000LL);
        hipDeviceSynchronize();
        std::vector<unsigned> h(n); hipMemcpy(h.data(), w, n * 4, hipMemcpyDeviceToHost);
        int ok = 0; int cnt[8] = {0};
        for (int b = 0; b < n; ++b) { ok += (h[b] == (unsigned)(b % 8)); cnt[h[b] & 7]++; }
        printf("grid %d: %d of %d workgroups on XCD b%%8; per XCD:", n, ok, n);
        for (int x = 0; x < 8; ++x) printf(" %d", cnt[x]);
        printf("\n  first 24:"); for (int b = 0; b < 24; ++b) printf(" %u", h[b]);
        printf("\n");
        hipFree(w);
    }
    return 0;
}
