// potf2_kernel.hip -- the diagonal-block factorisation (gpim_amd/csrc/potf2_body.hpp; in the product it is a role of
// chol_step_kernel, cholstep.hip) as a kernel of its own, for tools/potf2_prof.hip.  The diagonal-block kernels of the blocked Cholesky (replaces
// the diagonal steps of torch.linalg.cholesky, call sites gpim/gpreg/gpr.py:192,248).
//
//   potf2_kernel   one workgroup (8 waves) factors a 128x128 diagonal block that is resident in
//                  LDS and then inverts the factor in place (recursive doubling 16 -> 32 -> 64 -> 128
//                  on MFMA).  The explicit inverse turns the panel triangular solve and the leaves of
//                  the triangular inversion into GEMMs.
//
// potf2 works in 16-column panels:
//   wave 0, 16 lanes : 16x16 Cholesky in registers (row per lane; cross-lane broadcasts by
//                      v_readlane, pivots by v_rsq_f64 + Newton) -- the only serial chain
//   threads, 1/row   : panel solve by forward substitution against the 16x16 factor (LDS broadcast)
//   8 waves          : trailing update D -= P P^T on the lower 16x16 tiles, v_mfma_f64_16x16x4_f64
// A lone wave can issue an fp64 MFMA only every ~140 cycles, so the MFMA phases need >= 2 waves per
// SIMD: hence 512-thread workgroups.
#include "../gpim_amd/csrc/potf2_body.hpp"

#ifndef POTF2_WAVES_PER_EU
#define POTF2_WAVES_PER_EU 4
#endif
template <typename R>
__global__ __launch_bounds__(NTH, POTF2_WAVES_PER_EU) void potf2_kernel(R* __restrict__ A, int64_t ld, int kblk,
                                                       R* __restrict__ dinv_all,
                                                       double* __restrict__ logdet_out,
                                                       int32_t* __restrict__ info, int nb PROF_ARG) {
    __shared__ __attribute__((aligned(16))) double smem[POTF2_SMEM_DOUBLES];
#ifdef POTF2_PROFILE
    potf2_body<R, FI_LOOKAHEAD != 0>(smem, (int)blockIdx.y, A, ld, kblk, dinv_all, nullptr, logdet_out, info, nb, prof);
#else
    potf2_body<R, FI_LOOKAHEAD != 0>(smem, (int)blockIdx.y, A, ld, kblk, dinv_all, nullptr, logdet_out, info, nb);
#endif
}

