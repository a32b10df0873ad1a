// order_probe.hip -- (1) does hipExtAnyOrderLaunch let a kernel start while its predecessor in the SAME stream is
// still running on gfx950?  (2) what does a cross-stream event dependency cost next to an in-stream boundary?
// Kernels stamp wall_clock64() (100 MHz) at start and end.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void spin(long long* out, int slot, long long ticks) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) { }
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[2 * slot] = t0; out[2 * slot + 1] = wall_clock64(); }
}
int main() {
    long long* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    long long hbuf[512];
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ev[64]; for (auto& evx : ev) CK(hipEventCreateWithFlags(&evx, hipEventDisableTiming));
    // warm
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, 200, 100);
    CK(hipStreamSynchronize(s1));
    // (1) A: 50 us (5000 ticks); B any-order right behind it
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, 0, 5000);
        hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, nullptr, nullptr, hipExtAnyOrderLaunch, d, 1, (long long)500);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, 2, 500);
        CK(hipStreamSynchronize(s1));
        CK(hipMemcpy(hbuf, d, 4096, hipMemcpyDeviceToHost));
        printf("any-order: A [0, %.2f] us   B(any-order) [%.2f, %.2f]   C(plain) [%.2f, %.2f]\n", (hbuf[1] - hbuf[0]) / 100.0,
               (hbuf[2] - hbuf[0]) / 100.0, (hbuf[3] - hbuf[0]) / 100.0, (hbuf[4] - hbuf[0]) / 100.0, (hbuf[5] - hbuf[0]) / 100.0);
    }
    // (2) in-stream chain of 20 x 5 us kernels: boundary = start(i+1) - end(i)
    for (int rep = 0; rep < 2; ++rep) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, i, 500);
        CK(hipStreamSynchronize(s1));
        CK(hipMemcpy(hbuf, d, 4096, hipMemcpyDeviceToHost));
        double g = 0; for (int i = 1; i < 20; ++i) g += (hbuf[2 * i] - hbuf[2 * i - 1]) / 100.0;
        printf("in-stream boundary: %.2f us avg\n", g / 19);
    }
    // (3) ping-pong between two streams through events
    for (int rep = 0; rep < 2; ++rep) {
        for (int i = 0; i < 20; ++i) {
            hipStream_t s = (i & 1) ? s2 : s1, o = (i & 1) ? s1 : s2;
            if (i) CK(hipStreamWaitEvent(s, ev[i - 1], 0));
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, i, 500);
            CK(hipEventRecord(ev[i], s));
            (void)o;
        }
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        CK(hipMemcpy(hbuf, d, 4096, hipMemcpyDeviceToHost));
        double g = 0; for (int i = 1; i < 20; ++i) g += (hbuf[2 * i] - hbuf[2 * i - 1]) / 100.0;
        printf("cross-stream event boundary: %.2f us avg\n", g / 19);
    }
    // (4) the same ping-pong captured into a graph and replayed
    {
        hipGraph_t graph; hipGraphExec_t exec;
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeRelaxed));
        for (int i = 0; i < 20; ++i) {
            hipStream_t s = (i & 1) ? s2 : s1;
            if (i) CK(hipStreamWaitEvent(s, ev[i - 1], 0));
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, i, 500);
            CK(hipEventRecord(ev[i], s));
        }
        CK(hipStreamWaitEvent(s1, ev[19], 0));
        CK(hipStreamEndCapture(s1, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipGraphLaunch(exec, s1)); CK(hipStreamSynchronize(s1));
            CK(hipMemcpy(hbuf, d, 4096, hipMemcpyDeviceToHost));
            double g = 0; for (int i = 1; i < 20; ++i) g += (hbuf[2 * i] - hbuf[2 * i - 1]) / 100.0;
            printf("graph, cross-stream chain: %.2f us avg boundary\n", g / 19);
        }
        // fork-join inside a graph: A (50 us) on s1 || B on s2, then C on s1 after both
        CK(hipStreamBeginCapture(s1, hipStreamCaptureModeRelaxed));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, 10, 300);
        CK(hipEventRecord(ev[0], s1));
        CK(hipStreamWaitEvent(s2, ev[0], 0));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, 0, 5000);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s2, d, 1, 500);
        CK(hipEventRecord(ev[1], s2));
        CK(hipStreamWaitEvent(s1, ev[1], 0));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, 2, 500);
        CK(hipStreamEndCapture(s1, &graph));
        CK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipGraphLaunch(exec, s1)); CK(hipStreamSynchronize(s1));
            CK(hipMemcpy(hbuf, d, 4096, hipMemcpyDeviceToHost));
            printf("graph fork-join: pre end %.2f | A [%.2f, %.2f]  B [%.2f, %.2f]  C [%.2f, %.2f]\n", (hbuf[21] - hbuf[20]) / 100.0,
                   (hbuf[0] - hbuf[21]) / 100.0, (hbuf[1] - hbuf[21]) / 100.0, (hbuf[2] - hbuf[21]) / 100.0, (hbuf[3] - hbuf[21]) / 100.0,
                   (hbuf[4] - hbuf[21]) / 100.0, (hbuf[5] - hbuf[21]) / 100.0);
        }
    }
    // (5) eager fork-join
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, 10, 300);
        CK(hipEventRecord(ev[0], s1));
        CK(hipStreamWaitEvent(s2, ev[0], 0));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, 0, 5000);
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s2, d, 1, 500);
        CK(hipEventRecord(ev[1], s2));
        CK(hipStreamWaitEvent(s1, ev[1], 0));
        hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s1, d, 2, 500);
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2));
        CK(hipMemcpy(hbuf, d, 4096, hipMemcpyDeviceToHost));
        printf("eager fork-join: A [%.2f, %.2f]  B [%.2f, %.2f]  C [%.2f, %.2f]\n",
               (hbuf[0] - hbuf[21]) / 100.0, (hbuf[1] - hbuf[21]) / 100.0, (hbuf[2] - hbuf[21]) / 100.0, (hbuf[3] - hbuf[21]) / 100.0,
               (hbuf[4] - hbuf[21]) / 100.0, (hbuf[5] - hbuf[21]) / 100.0);
    }
    return 0;
}
