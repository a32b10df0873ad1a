// gemm_ablate.hip -- where does the fp64 tile engine lose MFMA issue slots?  The inner loop of
// gemm_tiles_kernel<.., 4, 128, 128> rebuilt piece by piece: MFMA only -> + LDS fragment reads ->
// + barrier per k-step -> + global loads and LDS stores.  2 workgroups of 4 waves per CU, 73 KB LDS
// each, 16 accumulators per wave, like the real kernel.  Development aid.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));
#define STAGE 2304

template <int MODE>
__global__ __launch_bounds__(256, 2) void loop(const double* __restrict__ G, double* out, int steps) {
    __shared__ __attribute__((aligned(16))) double smem[4 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    for (int i = tid; i < 4 * STAGE; i += 256) smem[i] = 1e-3 * (i % 97);
    __syncthreads();
    d4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (d4){0, 0, 0, 0};
    d2 ra[4], rb[4];
    for (int i = 0; i < 4; ++i) { ra[i] = (d2){1e-3 * tid, 2e-3}; rb[i] = (d2){3e-3, 1e-3 * i}; }
    const double* gp = G + (size_t)blockIdx.x * 4096 + tid * 2;
    for (int s = 0; s < steps; ++s) {
        const double* As = smem + (s & 1) * 2 * STAGE;
        const double* Bs = As + STAGE;
        if (MODE == 3 || MODE == 4 || MODE == 7 || MODE == 8 || MODE == 10) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = *reinterpret_cast<const d2*>(gp + ((s * 8 + i) & 63) * 512);
                rb[i] = *reinterpret_cast<const d2*>(gp + ((s * 8 + 4 + i) & 63) * 512);
            }
        }
        if (MODE == 5 || MODE == 9 || MODE == 11) {
            // direct global -> LDS: each wave fills whole 1 KB k-rows of the next stage
            double* An = smem + ((s + 1) & 1) * 2 * STAGE;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int krow = wave * 4 + i;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + ((s * 8 + i) & 63) * 512 - tid * 2 + lane * 2),
                                                 (__attribute__((address_space(3))) void*)(An + krow * 144), 16, 0, 0);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + ((s * 8 + 4 + i) & 63) * 512 - tid * 2 + lane * 2),
                                                 (__attribute__((address_space(3))) void*)(An + STAGE + krow * 144), 16, 0, 0);
            }
        }
        if (MODE == 8 || MODE == 9 || MODE == 11) __builtin_amdgcn_s_setprio(3);
        if (MODE == 10) __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            double a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (MODE == 11) {
                    // k-contiguous tile staged by direct loads: [m/8][(m%8)*8 + ((k/2) ^ (m%8))][k%2]
                    const int k = kk * 4 + (lane >> 4);
                    const int ma = wm * 64 + i * 16 + (lane & 15), mb = wn * 64 + i * 16 + (lane & 15);
                    a[i] = As[(ma >> 3) * 128 + ((ma & 7) * 8 + ((k >> 1) ^ (ma & 7))) * 2 + (k & 1)];
                    b[i] = Bs[(mb >> 3) * 128 + ((mb & 7) * 8 + ((k >> 1) ^ (mb & 7))) * 2 + (k & 1)];
                } else if (MODE >= 1) {
                    a[i] = As[(kk * 4 + (lane >> 4)) * 144 + wm * 64 + i * 16 + (lane & 15)];
                    b[i] = Bs[(kk * 4 + (lane >> 4)) * 144 + wn * 64 + i * 16 + (lane & 15)];
                } else {
                    a[i] = 1.0 + i + kk + lane * 1e-9;
                    b[i] = 0.5 + i - kk;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            if (MODE == 4 && kk == 1) {
                double* An = smem + ((s + 1) & 1) * 2 * STAGE;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = tid + 256 * i, krow = c / 64, c16 = c % 64;
                    *reinterpret_cast<d2*>(An + krow * 144 + c16 * 2) = ra[i];
                    *reinterpret_cast<d2*>(An + STAGE + krow * 144 + c16 * 2) = rb[i];
                }
            }
        }
        if (MODE == 7) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(ra[i]), "v"(rb[i]));
        }
        if (MODE == 8 || MODE == 9 || MODE == 11) __builtin_amdgcn_s_setprio(0);
        if (MODE == 10) __builtin_amdgcn_s_setprio(3);
        if (MODE == 3 || MODE == 6 || MODE == 8 || MODE == 10) {
            double* An = smem + ((s + 1) & 1) * 2 * STAGE;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int c = tid + 256 * i, krow = c / 64, c16 = c % 64;
                *reinterpret_cast<d2*>(An + krow * 144 + c16 * 2) = ra[i];
                *reinterpret_cast<d2*>(An + STAGE + krow * 144 + c16 * 2) = rb[i];
            }
        }
        if (MODE >= 2) __syncthreads();
    }
    double t = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (t == 1.2345) out[0] = t;
}

template <int MODE>
static void run(const char* tag, const double* G, double* out, int wgs) {
    const int steps = 2048;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop<MODE>), dim3(wgs), dim3(256), 0, 0, G, out, steps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    const double flop = (double)wgs * 4 * steps * 64 * 2048.0;
    printf("%-44s wgs=%d: %.3f ms  %.1f TFLOP/s\n", tag, wgs, best, flop / best / 1e9);
}

int main() {
    double *G, *out;
    hipMalloc(&G, (size_t)4096 * 4096 * 8); hipMemset(G, 0, (size_t)4096 * 4096 * 8); hipMalloc(&out, 16);
    for (int wgs : {2048}) {
        run<0>("MFMA only (register operands)", G, out, wgs);
        run<1>("+ LDS fragment reads", G, out, wgs);
        run<2>("+ barrier per 64 MFMAs", G, out, wgs);
        run<3>("+ global loads and LDS stores (full loop)", G, out, wgs);
        run<6>("barrier + LDS stores only (no global loads)", G, out, wgs);
        run<7>("barrier + global loads only (no LDS stores)", G, out, wgs);
        run<4>("full loop, LDS stores after kk=1", G, out, wgs);
        run<5>("full loop, direct global->LDS loads", G, out, wgs);
        run<8>("full loop + s_setprio 3 around the MFMA block", G, out, wgs);
        run<9>("direct loads + s_setprio 3 around the MFMA block", G, out, wgs);
        run<10>("full loop, staging at prio 3, MFMA block at prio 0", G, out, wgs);
        run<11>("direct loads + setprio, XOR-swizzled k-contiguous fragment reads", G, out, wgs);
    }
    return 0;
}
