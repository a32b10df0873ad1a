// gemm_ablate2.hip -- candidate inner loops for the fp64 tile engine, round 2.  Development aid.
// tools/gemm_ablate.hip showed that what the 128x128 / 4-wave loop loses against the pure-MFMA rate is
// ISSUE time of its memory instructions (~12 cycles each, 0.625 of them per MFMA).  Variants here:
//   V0  the shipped loop: 4 waves x (64x64), 2 workgroups/CU, m-contiguous operands, ds_read_b64
//   V1  same shape, fragment reads as ds_read_b128 (two adjacent m per lane -> two row blocks per read,
//       LDS row stride 128, conflict-free): 0.375 memory instructions per MFMA
//   V2  256x128 workgroup tile, 4 waves x (128x64), ONE workgroup per CU (one wave per SIMD), b128 reads,
//       3-stage LDS ring with partial vmcnt waits: 0.28 memory instructions per MFMA
//   V3  256x128 workgroup tile, 8 waves x (64x64), one workgroup per CU, b128 reads, 3-stage ring
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define GLDS(gptr, lptr) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gptr), \
                                                          (__attribute__((address_space(3))) void*)(lptr), 16, 0, 0)

// ---------------------------------------------------------------- V0 / V1: 128x128, 4 waves, 2 WG/CU
template <int MODE>
__global__ __launch_bounds__(256, 2) void loop128(const double* __restrict__ G, double* out, int steps) {
    constexpr int LDK = (MODE == 0) ? 144 : 128;
    constexpr int STAGE = 16 * LDK;
    __shared__ __attribute__((aligned(16))) double smem[4 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 4 * STAGE; i += 256) smem[i] = 1e-3 * (i % 97);
    __syncthreads();
    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0, 0, 0, 0};
    const double* gp = G + (size_t)blockIdx.x * 4096 + lane * 2;
    for (int s = 0; s < steps; ++s) {
        const double* As = smem + (s & 1) * 2 * STAGE;
        const double* Bs = As + STAGE;
        double* An = smem + ((s + 1) & 1) * 2 * STAGE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int krow = wave * 4 + i;
            GLDS(gp + ((s * 8 + i) & 63) * 512, An + krow * LDK);
            GLDS(gp + ((s * 8 + 4 + i) & 63) * 512, An + STAGE + krow * LDK);
        }
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double a[4], b[4];
            const int k = kk * 4 + (lane >> 4);
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[i] = As[k * LDK + wm * 64 + i * 16 + (lane & 15)];
                    b[i] = Bs[k * LDK + wn * 64 + i * 16 + (lane & 15)];
                }
            } else {
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const d2 va = *reinterpret_cast<const d2*>(As + k * LDK + wm * 64 + p * 32 + 2 * (lane & 15));
                    const d2 vb = *reinterpret_cast<const d2*>(Bs + k * LDK + wn * 64 + p * 32 + 2 * (lane & 15));
                    a[2 * p] = va[0]; a[2 * p + 1] = va[1];
                    b[2 * p] = vb[0]; b[2 * p + 1] = vb[1];
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    double t = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 1.2345) out[0] = t;
}

// ---------------------------------------------------------------- V2 / V3: 256x128, 1 WG/CU, 3-stage ring
// NW = 4: waves 2 (m) x 2 (n), wave tile 128 x 64 (8 x 4 accumulators)
// NW = 8: waves 4 (m) x 2 (n), wave tile  64 x 64 (4 x 4 accumulators)
template <int NW, int NSTAGE, int PRIO>
__global__ __launch_bounds__(NW * 64, 1) void loop256(const double* __restrict__ G, double* out, int steps) {
    constexpr int MT = (NW == 4) ? 8 : 4;
    constexpr int SA = 16 * 256, SB = 16 * 128, STAGE = SA + SB;     // doubles per stage (48 KB)
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < NSTAGE * STAGE; i += NW * 64) smem[i] = 1e-3 * (i % 97);
    __syncthreads();
    d4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0, 0, 0, 0};
    const double* gp = G + (size_t)blockIdx.x * 4096 + lane * 2;
    constexpr int LA = 32 / NW, LB = 16 / NW;        // wave-wide 1 KB loads per wave per stage
    auto issue = [&](int s) {
        double* An = smem + (s % NSTAGE) * STAGE;
#pragma unroll
        for (int i = 0; i < LA; ++i) {
            const int r = wave * LA + i;             // 1 KB row r of the A stage: k-row r>>1, half r&1
            GLDS(gp + ((s * 8 + i) & 63) * 512, An + (r >> 1) * 256 + (r & 1) * 128);
        }
#pragma unroll
        for (int i = 0; i < LB; ++i) {
            const int r = wave * LB + i;
            GLDS(gp + ((s * 8 + 4 + i) & 63) * 512, An + SA + r * 128);
        }
    };
    for (int s = 0; s < NSTAGE - 1; ++s) issue(s);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int s = 0; s < steps; ++s) {
        const double* As = smem + (s % NSTAGE) * STAGE;
        const double* Bs = As + SA;
        issue(s + NSTAGE - 1);                       // overwrites the stage read in step s-1
        if (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double a[MT], b[4];
            const int k = kk * 4 + (lane >> 4);
#pragma unroll
            for (int p = 0; p < MT / 2; ++p) {
                const d2 va = *reinterpret_cast<const d2*>(As + k * 256 + wm * (MT * 16) + p * 32 + 2 * (lane & 15));
                a[2 * p] = va[0]; a[2 * p + 1] = va[1];
            }
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const d2 vb = *reinterpret_cast<const d2*>(Bs + k * 128 + wn * 64 + p * 32 + 2 * (lane & 15));
                b[2 * p] = vb[0]; b[2 * p + 1] = vb[1];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
        // the loads of stage s+1 (issued one step ago) must have landed; those just issued may fly on
        if (NSTAGE == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (LA + LB == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        __syncthreads();
    }
    double t = 0;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 1.2345) out[0] = t;
}

// ---------------------------------------------------------------- V4: register-direct operands (no LDS, no barriers)
// Every wave loads its own 64x64 tile's operand fragments straight from global memory / L2 into registers in
// MFMA operand layout: lane group q = lane >> 4 takes k = 2q, 2q+1 of an 8-deep half-step (16 contiguous bytes of a
// k-contiguous row) -- any assignment of k to lane groups is fine as long as A and B use the same one.  One
// global_load_dwordx4 per 16x16 row block per half-step: 8 loads feed 32 MFMAs; the next half-step's loads are
// in flight while the current one computes.  2 workgroups x 4 waves per CU as in the shipped kernel.
template <int PRIO>
__global__ __launch_bounds__(256, 2) void loop_regdirect(const double* __restrict__ G, double* out, int steps) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    constexpr int LD = 4112;
    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0, 0, 0, 0};
    const int r = lane & 15, q = lane >> 4;
    const int rowA = (blockIdx.x * 128 + wm * 64 + r) % 3840, rowB = ((blockIdx.x * 37 + 5) * 128 + wn * 64 + r) % 3840;
    const double* pa = G + (size_t)rowA * LD + 2 * q;
    const double* pb = G + (size_t)rowB * LD + 2 * q;
    d2 ca[4], cb[4], na[4], nb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        ca[i] = *reinterpret_cast<const d2*>(pa + (size_t)i * 16 * LD);
        cb[i] = *reinterpret_cast<const d2*>(pb + (size_t)i * 16 * LD);
    }
    for (int s = 0; s < 2 * steps; ++s) {               // half-steps of 8 k
        const int koff = ((s + 1) * 8) & 4095;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            na[i] = *reinterpret_cast<const d2*>(pa + (size_t)i * 16 * LD + koff);
            nb[i] = *reinterpret_cast<const d2*>(pb + (size_t)i * 16 * LD + koff);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca[i][e], cb[j][e], acc[i][j], 0, 0, 0);
        if (PRIO) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int i = 0; i < 4; ++i) { ca[i] = na[i]; cb[i] = nb[i]; }
    }
    double t = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 1.2345) out[0] = t;
}

template <typename F>
static void timeit(const char* tag, double flop, F launch) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    hipError_t e = hipGetLastError();
    printf("%-64s %.3f ms  %.1f TFLOP/s %s\n", tag, best, flop / best / 1e9, e == hipSuccess ? "" : hipGetErrorString(e));
}

int main() {
    double *G, *out;
    hipMalloc(&G, (size_t)4096 * 4112 * 8); hipMemset(G, 0, (size_t)4096 * 4112 * 8); hipMalloc(&out, 16);
    const int steps = 2048;
    {
        const int wgs = 2048;
        const double flop = (double)wgs * 4 * steps * 64 * 2048.0;
        timeit("V0 128x128, 4 waves x 64x64, 2 WG/CU, ds_read_b64", flop, [&] { hipLaunchKernelGGL((loop128<0>), dim3(wgs), dim3(256), 0, 0, G, out, steps); });
        timeit("V1 128x128, 4 waves x 64x64, 2 WG/CU, ds_read_b128", flop, [&] { hipLaunchKernelGGL((loop128<1>), dim3(wgs), dim3(256), 0, 0, G, out, steps); });
        timeit("V4 128x128, 4 waves x 64x64, 2 WG/CU, register-direct operands (no LDS)", flop, [&] { hipLaunchKernelGGL((loop_regdirect<0>), dim3(wgs), dim3(256), 0, 0, G, out, steps); });
        timeit("V4 same + s_setprio around the MFMA block", flop, [&] { hipLaunchKernelGGL((loop_regdirect<1>), dim3(wgs), dim3(256), 0, 0, G, out, steps); });
    }
    {
        const int wgs = 1024;
        const double flop = (double)wgs * steps * 2.0 * 256 * 128 * 16;
        const size_t l3 = 3 * (16 * 256 + 16 * 128) * 8, l2 = 2 * (16 * 256 + 16 * 128) * 8;
        hipFuncSetAttribute((const void*)loop256<4, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3);
        hipFuncSetAttribute((const void*)loop256<4, 3, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3);
        hipFuncSetAttribute((const void*)loop256<8, 3, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3);
        hipFuncSetAttribute((const void*)loop256<8, 3, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l3);
        hipFuncSetAttribute((const void*)loop256<4, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
        hipFuncSetAttribute((const void*)loop256<8, 2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l2);
        timeit("V2 256x128, 4 waves x 128x64, 1 WG/CU, b128, 3 stages, prio", flop, [&] { hipLaunchKernelGGL((loop256<4, 3, 1>), dim3(wgs), dim3(256), l3, 0, G, out, steps); });
        timeit("V2 256x128, 4 waves x 128x64, 1 WG/CU, b128, 3 stages, no prio", flop, [&] { hipLaunchKernelGGL((loop256<4, 3, 0>), dim3(wgs), dim3(256), l3, 0, G, out, steps); });
        timeit("V2 256x128, 4 waves x 128x64, 1 WG/CU, b128, 2 stages, prio", flop, [&] { hipLaunchKernelGGL((loop256<4, 2, 1>), dim3(wgs), dim3(256), l2, 0, G, out, steps); });
        timeit("V3 256x128, 8 waves x 64x64, 1 WG/CU, b128, 3 stages, prio", flop, [&] { hipLaunchKernelGGL((loop256<8, 3, 1>), dim3(wgs), dim3(512), l3, 0, G, out, steps); });
        timeit("V3 256x128, 8 waves x 64x64, 1 WG/CU, b128, 3 stages, no prio", flop, [&] { hipLaunchKernelGGL((loop256<8, 3, 0>), dim3(wgs), dim3(512), l3, 0, G, out, steps); });
        timeit("V3 256x128, 8 waves x 64x64, 1 WG/CU, b128, 2 stages, prio", flop, [&] { hipLaunchKernelGGL((loop256<8, 2, 1>), dim3(wgs), dim3(512), l2, 0, G, out, steps); });
    }
    return 0;
}
