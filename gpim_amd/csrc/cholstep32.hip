// cholstep32.hip -- the step schedule of cholstep.hip (torch.linalg.cholesky at
// gpim/gpreg/gpr.py:192,248) for single-precision handles (reconstructor(precision='single'), gpr.py:104-113):
// float matrices and fp32 MFMA tiles, the diagonal blocks factored and inverted in double like everywhere else.
//
// Same plan (step_plan_build), same three launches per block column on ONE in-order stream:
//   H_j  chol_step_kernel_f32   workgroup 0 factors block j (potf2_body<float>), workgroups >= 8 compute pending
//                               trailing-update tiles with the float instantiation of the generic tile engine
//   F_j  panel_solve_kernel_f32 one-shot strip solve with the inverse in MFMA operand order (dinvB, kept in double:
//                               the fragments are narrowed on the fly)
//   D_j  diag_update_kernel_f32 one-shot update of the next diagonal tiles
// What differs from the fp64 kernels is the element type and the C/D row map of v_mfma_f32_16x16x4_f32
// (row = 4 * (lane >> 4) + reg; A and B lane maps are those of the f64 instruction).
//
// Why it exists: until round 3 single-precision handles kept the two-stream look-ahead schedule, whose bulk updates
// live on a CU-masked stream of normal priority -- exposed to the stream-population effect of DESIGN.md section 6
// (43 -> 55 ms per iteration at N = 16384 once the process holds more streams than hardware queues).  One in-order
// stream of launches can be driven from the engine's high-priority chain stream like the fp64 schedule.
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include "potf2_body.hpp"
#include "gemm_kernel.hpp"

typedef float f2 __attribute__((ext_vector_type(2)));

struct StepArgs32 {
    float* A; int64_t ld; int kblk; int nb;
    float* dinv_all; double* dinvB_all; double* logdet; int32_t* info;
    int col_off;
    GemmArgs g;                 // filler tiles (NT, alpha = -1, beta = 1); pointers typed double* like everywhere
};

// blockIdx.x == 0: the factorisation; 1..7 idle (filler b stays on XCD b % 8); >= 8: filler tile b - 8
template <int FTM, int FTN>
__global__ __launch_bounds__(NTH, 4) void chol_step_kernel_f32(StepArgs32 a) {
    constexpr int GS = (gemm_smem_elems_t<float, FTM, FTN>() + 1) / 2;           // in doubles
    constexpr int SM = POTF2_SMEM_DOUBLES > GS ? POTF2_SMEM_DOUBLES : GS;
    static_assert(SM * 8 <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) double smem[SM];
    const int b = blockIdx.x;
    if (b >= 8) {
        gemm_tile_body_t<float, false, false, EPI_STORE, 8, FTM, FTN>(a.g, b - 8, (int)blockIdx.y, reinterpret_cast<float*>(smem));
        return;
    }
    if (b != 0) return;
    potf2_body<float>(smem, (int)blockIdx.y, a.A, a.ld, a.kblk, a.dinv_all, a.dinvB_all, a.logdet, a.info, a.nb, a.col_off);
}

// Panel solve, one workgroup (4 waves) per 32-row strip of the block column below the diagonal block:
// S = P * Dinv^T in place.  Wave w owns the 16-column tiles w and 7 - w (36 k-steps per 16 rows for every wave).
// LDS rows are 130 floats apart: 130 % 32 == 2 puts the 16 rows x 4 k of a fragment read on 64 distinct
// (bank, half) slots; 8-byte row alignment, hence the strip is moved in 8-byte chunks.
__global__ __launch_bounds__(256, 2) void panel_solve_kernel_f32(float* __restrict__ A, int64_t ld, int kblk, int nb,
                                                              const double* __restrict__ dinvB_all) {
    constexpr int LDS_LD = 130;
    __shared__ __attribute__((aligned(16))) float S[32 * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    A += (int64_t)blockIdx.y * nb * NB * ld;
    const d2* DB = reinterpret_cast<const d2*>(dinvB_all + ((int64_t)blockIdx.y * nb + kblk) * (NB * NB));
    float* P = A + ((int64_t)(kblk + 1) * NB + (int64_t)blockIdx.x * 32) * ld + (int64_t)kblk * NB;
    // the strip (32 x 128 floats = 2048 eight-byte chunks) and the B fragments of both tiles, all loads in flight
    f2 rv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + 256 * i, row = c >> 6, c2 = c & 63;
        rv[i] = *reinterpret_cast<const f2*>(P + (int64_t)row * ld + c2 * 2);
    }
    const int t0 = wave, t1 = 7 - wave, n0 = 2 * t0 + 2;
    d2 bf[18];
#pragma unroll
    for (int q = 0; q < 18; ++q) {
        const int t = q < n0 ? t0 : t1, s2 = q < n0 ? q : q - n0;
        bf[q] = DB[(t * 16 + s2) * 64 + lane];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + 256 * i, row = c >> 6, c2 = c & 63;
        *reinterpret_cast<f2*>(S + row * LDS_LD + c2 * 2) = rv[i];
    }
    __syncthreads();
    const f4 zero = (f4){0.f, 0.f, 0.f, 0.f};
    f4 acc[2][2] = {{zero, zero}, {zero, zero}};
    const float* Sa = S + (lane & 15) * LDS_LD + (lane >> 4);
#pragma unroll
    for (int q = 0; q < 18; ++q) {
        const bool first = q < n0;          // wave-uniform
        const int s2 = first ? q : q - n0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int s = 2 * s2 + e;
            const float a0 = Sa[4 * s], a1 = Sa[16 * LDS_LD + 4 * s], bv = (float)bf[q][e];
            if (first) {
                acc[0][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc[0][1], 0, 0, 0);
            } else {
                acc[1][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bv, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bv, acc[1][1], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int t = x == 0 ? t0 : t1;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                P[(int64_t)(mt * 16 + 4 * (lane >> 4) + rg) * ld + t * 16 + (lane & 15)] = acc[x][mt][rg];
    }
}

// Diagonal tiles (jj,jj), jj = kblk+1 .. kblk+ntile, -= L[jj,kblk] L[jj,kblk]^T: one workgroup (4 waves, one 16x16
// MFMA tile each) per 32x32 quadrant of the lower half, one-shot like the panel solve.
__global__ __launch_bounds__(256, 2) void diag_update_kernel_f32(float* __restrict__ A, int64_t ld, int kblk, int nb) {
    constexpr int LDS_LD = 130;
    __shared__ __attribute__((aligned(16))) float S[64 * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    A += (int64_t)blockIdx.y * nb * NB * ld;
    const int jj = kblk + 1 + blockIdx.x / 10, q = blockIdx.x % 10;
    const int a = q < 1 ? 0 : q < 3 ? 1 : q < 6 ? 2 : 3, b = q - a * (a + 1) / 2;     // quadrant (a, b), a >= b
    const float* Pa = A + ((int64_t)jj * NB + a * 32) * ld + (int64_t)kblk * NB;
    const float* Pb = A + ((int64_t)jj * NB + b * 32) * ld + (int64_t)kblk * NB;
    f2 ra[8], rb[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + 256 * i, row = c >> 6, c2 = c & 63;
        ra[i] = *reinterpret_cast<const f2*>(Pa + (int64_t)row * ld + c2 * 2);
        rb[i] = *reinterpret_cast<const f2*>(Pb + (int64_t)row * ld + c2 * 2);
    }
    const int wm = wave >> 1, wn = wave & 1;
    float* C = A + ((int64_t)jj * NB + a * 32 + wm * 16 + 4 * (lane >> 4)) * ld + (int64_t)jj * NB + b * 32 + wn * 16 + (lane & 15);
    float cv[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) cv[rg] = C[(int64_t)rg * ld];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = tid + 256 * i, row = c >> 6, c2 = c & 63;
        *reinterpret_cast<f2*>(S + row * LDS_LD + c2 * 2) = ra[i];
        *reinterpret_cast<f2*>(S + (32 + row) * LDS_LD + c2 * 2) = rb[i];
    }
    __syncthreads();
    f4 acc0 = (f4){0.f, 0.f, 0.f, 0.f}, acc1 = acc0;
    const float* Sa = S + (wm * 16 + (lane & 15)) * LDS_LD + (lane >> 4);
    const float* Sb = S + (32 + wn * 16 + (lane & 15)) * LDS_LD + (lane >> 4);
#pragma unroll
    for (int s = 0; s < 32; s += 2) {
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(Sa[4 * s], Sb[4 * s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(Sa[4 * s + 4], Sb[4 * s + 4], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) C[(int64_t)rg * ld] = cv[rg] - (acc0[rg] + acc1[rg]);
}

// ------------------------------------------------------------------------------------------
// host side (plans: cholstep.hip)
// ------------------------------------------------------------------------------------------
static GemmArgs nt_update32(double* A, int64_t ld, const TileDesc* tiles, int n, int64_t rows) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = ld; g.B = A; g.ldb = ld; g.C = A; g.ldc = ld;
    g.alpha = -1.0; g.beta = 1.0; g.tiles = tiles; g.ntiles = n;
    g.sA = g.sB = g.sC = rows * ld;          // elements
    return g;
}

// launch_potrf_steps of cholstep.hip for a single-precision handle: A is a float matrix behind its double* type
int launch_potrf_steps_f32(gpimhip_ctx* h, double* A_, int64_t np, int64_t ld, int32_t* info) {
    const int nb = (int)(np / NB), W = 4;
    GP_TRY(step_plan_ensure(h, nb));
    const StepPlan& P = h->splan;
    float* A = reinterpret_cast<float*>(A_);
    float* const dinv = reinterpret_cast<float*>(h->dinv);
    double* const Ad = reinterpret_cast<double*>(A);
    const int B = h->nbatch;
    const int quad_max = 128, half_max = 512, host_max_batch = 4;
    for (int j = 0; j < nb; ++j) {
        if (j % W == 0 && P.bulk_rest[j / W].n) {
            GemmArgs g = nt_update32(Ad, ld, P.d_tiles + P.bulk_rest[j / W].off, P.bulk_rest[j / W].n, h->np);
            GP_TRY(launch_gemm(h, false, false, EPI_STORE, g));
        }
        StepArgs32 a;
        a.A = A; a.ld = ld; a.kblk = j; a.nb = nb;
        a.dinv_all = dinv; a.dinvB_all = h->dinvB; a.logdet = h->logdet_part; a.info = info;
        a.col_off = 0;
        a.g = nt_update32(Ad, ld, P.d_tiles + P.fill[j].off, P.fill[j].n, h->np);
        const int nf = P.fill[j].n;
        a.g.chunk = std::max(1, std::min(64, nf / 512));
        if (B > host_max_batch) {
            if (nf) GP_TRY(launch_gemm(h, false, false, EPI_STORE, a.g));
            hipLaunchKernelGGL((chol_step_kernel_f32<64, 64>), dim3(1, B), dim3(NTH), 0, h->stream, a);
        } else if ((int64_t)nf * B <= quad_max)
            hipLaunchKernelGGL((chol_step_kernel_f32<64, 64>), dim3(nf ? 8 + 4 * nf : 1, B), dim3(NTH), 0, h->stream, a);
        else if ((int64_t)nf * B <= half_max)
            hipLaunchKernelGGL((chol_step_kernel_f32<128, 64>), dim3(8 + 2 * nf, B), dim3(NTH), 0, h->stream, a);
        else
            hipLaunchKernelGGL((chol_step_kernel_f32<128, 128>), dim3(8 + nf, B), dim3(NTH), 0, h->stream, a);
        HIP_TRY(hipGetLastError());
        if (j + 1 < nb) {
            hipLaunchKernelGGL(panel_solve_kernel_f32, dim3(4 * (nb - j - 1), B), dim3(256), 0, h->stream, A, ld, j, nb,
                               (const double*)h->dinvB);
            HIP_TRY(hipGetLastError());
            hipLaunchKernelGGL(diag_update_kernel_f32, dim3(10 * P.diag[j].n, B), dim3(256), 0, h->stream, A, ld, j, nb);
            HIP_TRY(hipGetLastError());
        }
    }
    return GPIMHIP_OK;
}
