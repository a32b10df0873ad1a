/*
 * gpimhip.h -- C ABI of libgpimhip.so, the MI355X (gfx950) engine behind
 * gpim_amd.reconstructor / gpim_amd.boptimizer.
 *
 * The reference (ziatdinovmax/GPim, /root/reference) has no native boundary: its hot
 * path is Python calling pyro.contrib.gp / torch.  Each entry point below replaces the
 * group of third-party calls made at the cited reference call site; the Python host
 * code in gpim_amd/ binds them with ctypes.
 *
 * Conventions
 *   - every data pointer is a DEVICE pointer (HBM) borrowed for the duration of the
 *     call; outputs are caller-allocated; the library owns only its workspace.
 *   - all work is enqueued on the handle's HIP stream; calls return without
 *     synchronising unless stated.  Nothing here allocates per call once the
 *     workspace for a problem size exists.
 *   - matrices are row-major doubles; "lower" means the i >= j part is meaningful.
 *   - return value: 0 = ok, <0 = error (see GPIMHIP_E_*), message via
 *     gpimhip_last_error().  Non-positive-definite detection is asynchronous: the
 *     factorisation records the first failing column in a device word that
 *     gpimhip_fit_exact / gpimhip_predict_exact read back at their final sync and
 *     report as GPIMHIP_E_NOT_PD (torch.linalg.cholesky raises at gpr.py:192,248).  Training loops stop
 *     updating on the device at the failing iteration and stop being enqueued a bounded number of
 *     iterations later (see gpimhip_fit_completed).
 *   - parameter vector u (unconstrained, the thing Adam updates), length
 *     P = 2 + n_ls (+1 for RationalQuadratic):
 *         u[0]            variance      sigma^2 = amp_lo + (amp_hi-amp_lo)*sigmoid(u)
 *         u[1..n_ls]      lengthscale   l_k     = ls_lo_k + (ls_hi_k-ls_lo_k)*sigmoid(u)
 *         u[1+n_ls]       noise         s_n^2   = exp(u)
 *         u[2+n_ls]       scale_mixture alpha   = exp(u)      (RationalQuadratic only)
 *     (pyro_kernels.py:81-94 Uniform priors -> interval constraints)
 *   - three regimes behind gpimhip_fit_exact, same arithmetic: N <= 128 one fused launch per
 *     training (one workgroup, K/L/L^-1 resident in LDS); mid N the blocked path with one iteration
 *     captured in a hipGraph and replayed; N >= 6144 the blocked path with a look-ahead panel
 *     stream (side streams are created on first use).  Environment switches for A/B testing:
 *     GPIMHIP_NO_SMALLN, GPIMHIP_NO_GRAPH, GPIMHIP_NO_CUMASK (any value disables the respective
 *     mechanism), GPIMHIP_LOOKAHEAD_MIN_PANELS, GPIMHIP_RESERVED_CUS.
 */
#ifndef GPIMHIP_H
#define GPIMHIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPIMHIP_OK            0
#define GPIMHIP_E_BADARG     -1
#define GPIMHIP_E_HIP        -2
#define GPIMHIP_E_NOT_PD     -3
#define GPIMHIP_E_NOMEM      -4

#define GPIMHIP_KERNEL_RBF       0
#define GPIMHIP_KERNEL_MATERN52  1
#define GPIMHIP_KERNEL_RQ        2

#define GPIMHIP_ACQ_CB   0
#define GPIMHIP_ACQ_EI   1
#define GPIMHIP_ACQ_POI  2

#define GPIMHIP_MAX_DIM  4
#define GPIMHIP_MAX_PARAMS 8

typedef struct gpimhip_ctx* gpimhip_handle;

/* Model description: kernel family + parameterisation.
 * Replaces gpim/kernels/pyro_kernels.py:14-96 (get_kernel) and the jitter argument of
 * gp.models.GPRegression at gpim/gpreg/gpr.py:141-144. */
typedef struct {
    int32_t kernel;                     /* GPIMHIP_KERNEL_*                                  */
    int32_t dim;                        /* d, 1..4                                           */
    int32_t n_ls;                       /* 1 (isotropic) or dim                              */
    int32_t reserved;
    double  amp_lo, amp_hi;             /* variance prior bounds (default 1e-4, 10)          */
    double  ls_lo[GPIMHIP_MAX_DIM];     /* lengthscale prior bounds                          */
    double  ls_hi[GPIMHIP_MAX_DIM];
    double  jitter;                     /* added to diag(K) with the noise (gpr.py:141)      */
} gpimhip_model_t;

/* ---- lifetime ------------------------------------------------------------------- */
/* hip_stream: the hipStream_t every call is enqueued on; NULL selects the device's default
 * (null) stream, so work is ordered with the caller's own default-stream work. */
int  gpimhip_create(gpimhip_handle* out, int device, void* hip_stream);
int  gpimhip_destroy(gpimhip_handle h);
const char* gpimhip_last_error(void);
int  gpimhip_version(void);
/* bytes of device workspace currently held by the handle */
int64_t gpimhip_workspace_bytes(gpimhip_handle h);

/* ---- operator-level entry points (parity hooks; each is also a stage of fit/predict) -- */

/* K = k(X, Z) (+ diag_add on the diagonal when Z == NULL, i.e. the symmetric case).
 * Replaces the Pyro kernel forward called from GPRegression.model/forward
 * (call sites gpim/gpreg/gpr.py:192,248; formulas SURVEY App. A.3).
 * theta (device, doubles): [sigma^2, l_0..l_{n_ls-1}, alpha_rq].  out: (N x M) row-major,
 * leading dimension ld >= M.  Symmetric case writes the full square. */
int gpimhip_kmat(gpimhip_handle h, const gpimhip_model_t* m,
                 const double* X, int64_t N, const double* Z, int64_t M,
                 const double* theta, double diag_add, double* out, int64_t ld);

/* Arithmetic of the exact-GP path (gpimhip_nll_grad, gpimhip_fit_exact*, gpimhip_predict_exact*,
 * gpimhip_acquire_exact): bits = 64 (default) or 32 = the reconstructor's precision='single'
 * (gpim/gpreg/gpr.py:104-113: float32 tensors end to end).  With 32 the N x N matrices (covariance, factor,
 * inverse, K* slab) are float and every O(N^3) product runs on v_mfma_f32_16x16x4_f32 (twice the fp64 matrix
 * rate, half the HBM bytes); diagonal blocks are factored in double and all O(N) vectors, reductions, the
 * loss, the gradient and Adam stay double (wider than the reference's float32, never narrower).  Inputs and
 * outputs of the C ABI remain double.  The fused trainer for N <= 128 computes in double on either setting.
 * Entry points outside that path (kmat, potrf, vfe, dist) require bits = 64.  Switching releases the workspace. */
int gpimhip_set_precision(gpimhip_handle h, int32_t bits);

/* In-place lower Cholesky of the n x n matrix A (row-major, ld), n any size >= 1;
 * the strict upper triangle is left untouched.  info (device int32): 0, or 1 + first
 * failing column.  Replaces torch.linalg.cholesky inside Pyro (gpr.py:192,248). */
int gpimhip_potrf(gpimhip_handle h, double* A, int64_t n, int64_t ld, int32_t* info);

/* loss (negative log marginal likelihood + prior constant) and d loss / d u at u.
 * Replaces Trace_ELBO().differentiable_loss + loss.backward() (gpr.py:186,192-193).
 * loss_out: 1 double, grad_out: P doubles (device). */
int gpimhip_nll_grad(gpimhip_handle h, const gpimhip_model_t* m,
                     const double* X, const double* y, int64_t N,
                     const double* u, double* loss_out, double* grad_out);

/* ---- the hot path ------------------------------------------------------------------ */

/* T Adam iterations on u, entirely on device (no host sync inside the loop).
 * Replaces the training loop reconstructor.train (gpr.py:185-199): fresh Adam state
 * (t=0, m=v=0), lr, betas (0.9, 0.999), eps 1e-8.
 *   u_inout   P doubles (device), updated in place
 *   hist_out  T x (P) doubles (device): constrained values AFTER each step, in the
 *             order [sigma^2, l_*, s_n^2(, alpha)]  (what gpr.py:195-197 appends)
 *   loss_out  T doubles (device) or NULL: loss evaluated BEFORE each step
 * Synchronises the stream once at the end; returns GPIMHIP_E_NOT_PD if any
 * factorisation failed (info of the first failure via gpimhip_last_error). */
int gpimhip_fit_exact(gpimhip_handle h, const gpimhip_model_t* m,
                      const double* X, const double* y, int64_t N,
                      double* u_inout, double lr, int32_t T,
                      double* hist_out, double* loss_out);

/* Posterior mean and variance (full_cov=False, noiseless=False) at Xs (M x d), NaN rows
 * allowed (they produce NaN outputs).  Replaces GPRegression.forward -> conditional
 * (gpr.py:247-248): K and L are recomputed at u.
 *   mean_out, var_out: M doubles (device); var includes the noise, excludes jitter.
 * Synchronises at the end (to report NOT_PD). */
int gpimhip_predict_exact(gpimhip_handle h, const gpimhip_model_t* m,
                          const double* X, const double* y, int64_t N,
                          const double* u, const double* Xs, int64_t M,
                          double* mean_out, double* var_out);

/* Batched forms: B independent problems with the SAME N (and the same model description), advanced in
 * lock-step by every launch (grid.y = problem index) -- B spectral slices of a cube share each
 * latency-bound step of the blocked factorisation instead of paying for it B times.
 *   X         problem b reads X + b*x_stride (N x d); x_stride = 0 shares one X between all problems
 *   y         B x N,  u_inout / u  B x P,  hist_out  B x T x P,  loss_out  B x T
 *   Xs        M x d test points shared by all problems;  mean_out / var_out  B x M
 * The reference has no such loop (it fits one GP per call, gpr.py:257-283); a caller that loops
 * reconstructor(...).run() over slices gets identical per-slice results from one batched call. */
int gpimhip_fit_exact_batched(gpimhip_handle h, const gpimhip_model_t* m,
                              const double* X, int64_t x_stride, const double* y, int64_t N, int32_t B,
                              double* u_inout, double lr, int32_t T,
                              double* hist_out, double* loss_out);
int gpimhip_predict_exact_batched(gpimhip_handle h, const gpimhip_model_t* m,
                                  const double* X, int64_t x_stride, const double* y, int64_t N, int32_t B,
                                  const double* u, const double* Xs, int64_t M,
                                  double* mean_out, double* var_out);

/* ---- sparse (inducing-point) GP, variational free energy ---------------------------------
 * Replaces pyro.contrib.gp.models.SparseGPRegression(approx="VFE") as constructed by
 * reconstructor(sparse=True) (gpim/gpreg/gpr.py:145-155; formulas SURVEY App. A.7).
 * The parameter vector is u = [ P kernel/noise entries as above | Mu x d inducing inputs Xu,
 * row-major, unconstrained ]; both parts are trained by Adam (the reference trains Xu too and
 * records it every iteration, gpr.py:198-199).
 *   gpimhip_vfe_nll_grad   loss (1 double) and gradient (P + Mu*d doubles) at u
 *   gpimhip_fit_vfe        T Adam iterations; hist_theta T x P, hist_xu T x Mu x d (either may be
 *                          NULL), loss_out T or NULL; synchronises at the end
 *   gpimhip_predict_vfe    posterior mean / variance (full_cov=False, noiseless=False) at Xs */
int gpimhip_vfe_nll_grad(gpimhip_handle h, const gpimhip_model_t* m,
                         const double* X, const double* y, int64_t N, int64_t Mu,
                         const double* u, double* loss_out, double* grad_out);
int gpimhip_fit_vfe(gpimhip_handle h, const gpimhip_model_t* m,
                    const double* X, const double* y, int64_t N, int64_t Mu,
                    double* u_inout, double lr, int32_t T,
                    double* hist_theta, double* hist_xu, double* loss_out);
int gpimhip_predict_vfe(gpimhip_handle h, const gpimhip_model_t* m,
                        const double* X, const double* y, int64_t N, int64_t Mu,
                        const double* u, const double* Xs, int64_t M,
                        double* mean_out, double* var_out);
/* B sparse models of equal N and Mu in lock-step (the slices of a 4D cube, gpim/gpreg/gpr.py:145-155 once per slice):
 * every launch of the training loop carries all of them (blockIdx.y = model); each model's arithmetic is that of its own
 * gpimhip_fit_vfe / gpimhip_predict_vfe call.  X: B x N x d with x_stride doubles between models (0: shared inputs),
 * y: B x N, u: B x (P + Mu*d), hist_theta: B x T x P, hist_xu: B x T x Mu x d, loss_out: B x T (each may be NULL);
 * Xs: test points, xs_stride doubles between models (0: shared); mean_out / var_out: B x M.  A factorisation that fails in
 * any model freezes the whole batch at that iteration (GPIMHIP_E_NOT_PD, gpimhip_fit_completed). */
int gpimhip_fit_vfe_batched(gpimhip_handle h, const gpimhip_model_t* m,
                            const double* X, int64_t x_stride, const double* y, int64_t N, int64_t Mu, int32_t B,
                            double* u_inout, double lr, int32_t T,
                            double* hist_theta, double* hist_xu, double* loss_out);
int gpimhip_predict_vfe_batched(gpimhip_handle h, const gpimhip_model_t* m,
                                const double* X, int64_t x_stride, const double* y, int64_t N, int64_t Mu, int32_t B,
                                const double* u, const double* Xs, int64_t xs_stride, int64_t M,
                                double* mean_out, double* var_out);

/* ---- exact GP on a fully observed regular grid (Kronecker-structured covariance) ---------------
 * Takes the role of the reference's structured-kernel reconstructor (gpim/gpreg/skgpr.py:399-448, there
 * GPyTorch's interpolated SKI approximation): the SAME model and parameterisation as
 * gpimhip_fit_exact / gpimhip_predict_exact (RBF kernel, ARD or isotropic), but for observations on a
 * complete product grid c_1 x ... x c_d the covariance is s2 K_1 (x) ... (x) K_d and loss, gradient,
 * Adam loop and posterior follow from the n_i x n_i eigen-decompositions: O(sum n_i^3 + N sum n_i) work,
 * O(N + sum n_i^2) memory, results equal to the dense path's up to rounding.
 *   d, n[d]      grid shape (HOST array; N = prod n_i observations in C order, last axis fastest)
 *   axes         device, sum n_i doubles: the coordinate vectors c_1, ..., c_d, concatenated
 *   y            device, N doubles (no NaN: the grid must be fully observed)
 *   n_test / axes_test   the prediction grid, the same way (M = prod n_test_i outputs, C order)
 * Only GPIMHIP_KERNEL_RBF factorises over the axes; other kernels return GPIMHIP_E_BADARG. */
int gpimhip_kron_nll_grad(gpimhip_handle h, const gpimhip_model_t* m, int32_t d, const int32_t* n,
                          const double* axes, const double* y, const double* u,
                          double* loss_out, double* grad_out);
int gpimhip_fit_kron(gpimhip_handle h, const gpimhip_model_t* m, int32_t d, const int32_t* n,
                     const double* axes, const double* y, double* u_inout, double lr, int32_t T,
                     double* hist_out, double* loss_out);
int gpimhip_predict_kron(gpimhip_handle h, const gpimhip_model_t* m, int32_t d, const int32_t* n,
                         const double* axes, const double* y, const double* u,
                         const int32_t* n_test, const double* axes_test,
                         double* mean_out, double* var_out);

/* Acquisition sweep over the dense grid (gpim/gpbayes/acqfunc.py:11-92):
 *   CB : p0*mean + p1*sd                                  (alpha, beta)
 *   EI : imp*Phi(imp/sd) + sd*phi(imp/sd), imp = mean - p0 - p1     (best, xi)
 *   POI: Phi((mean - p0 - p1)/sd)
 * mask (M doubles of 1/NaN, or NULL) is multiplied in as boptim.py:307-308 does.
 * mean, sd, acq_out: M doubles (device). */
int gpimhip_acq(gpimhip_handle h, int32_t kind, const double* mean, const double* sd,
                int64_t M, double p0, double p1, const double* mask, double* acq_out);

/* One acquisition evaluation of boptimizer for an exact-GP surrogate, without leaving the device and with
 * ONE factorisation (replaces the predict / predict / nanmax / sweep sequence of gpim/gpbayes/acqfunc.py:27-29,
 * 52-62, 80-91 and boptim.py:303-308):
 *   posterior at the observed rows Xobs (Mobs x dim; EI, POI only) -> incumbent = nanmax of the means (EI) or
 *   of the means and standard deviations (POI: the reference's tuple quirk, acqfunc.py:86-88), kept on the device;
 *   posterior at the grid Xs (M x dim) -> mean_out, sd_out; acq_out = CB(p0 = alpha, p1 = beta) or
 *   EI / POI(incumbent, p1 = xi), times mask (M doubles of 1 / NaN, or NULL).
 * For N <= 384 every prediction is one fused launch whose K(X, X*) panel never leaves LDS (csrc/predict.hip). */
int gpimhip_acquire_exact(gpimhip_handle h, const gpimhip_model_t* m, const double* X, const double* y,
                          int64_t N, const double* u, const double* Xs, int64_t M, const double* Xobs,
                          int64_t Mobs, int32_t kind, double p0, double p1, const double* mask,
                          double* mean_out, double* sd_out, double* acq_out);

/* nanmax over n doubles (device) -> out (1 double, device): the incumbent
 * "np.nanmax(mean_sample)" of acqfunc.py:59,88. */
int gpimhip_nanmax(gpimhip_handle h, const double* x, int64_t n, double* out);

/* Descending top-k of acq (NaNs ranked first when keep_nan != 0, exactly like
 * np.argsort(...)[::-1] at boptim.py:303-306; dropped otherwise, boptim.py:310-315).
 * vals_out: k doubles, idx_out: k int64 flat indices (device).  count_out (device
 * int64): number of valid entries written (< k when fewer non-NaN values exist).
 * Ties rank the larger flat index first.  Grids of more than 2048 points use a multi-block radix
 * selection (O(M) per pass, no host round trip), smaller ones k arg-max passes of one workgroup. */
int gpimhip_topk(gpimhip_handle h, const double* acq, int64_t M, int32_t k, int32_t keep_nan,
                 double* vals_out, int64_t* idx_out, int64_t* count_out);

/* ---- one exact GP across several GPUs: building blocks of a block-column-cyclic Cholesky ----------
 * (SURVEY 8(f) rank 1.  The reference treats a cube as ONE d-dimensional GP, gpim/gpreg/gpr.py:30-43,
 * 115-126; its covariance does not fit one device beyond N ~ 10^5.)  The matrix is dealt to the P ranks
 * of a process group by 512-column panels (panel p -> rank p mod P, the 1 x P case of a 2-D block-cyclic
 * layout); each rank keeps its panels, all rows, side by side in ONE caller-owned row-major array
 *     Aloc : np x (512 * owned panels), leading dimension ldloc,   np = N padded to 128.
 * The host driver (gpim_amd/dist_chol.py, torch.distributed / RCCL) runs, for p = 0, 1, ...:
 *     owner(p):  gpimhip_dist_panel_factor   -> L(:, panel p) in place   (on a side stream: look-ahead)
 *                gpimhip_dist_panel_pack     -> broadcast buffer: rows of the panel + its diagonal-block inverses
 *     all     :  broadcast of the buffer ((np + 128) x 512 doubles, src = owner)
 *     each    :  gpimhip_dist_update for its owned panels right of p (panel p+1 first, the rest in one launch)
 *   gpimhip_dist_setup            per-handle set-up for order n on rank `rank` of `world`: diagonal-block workspace
 *                                 and the launch plans of this rank's share -- none of the n x n buffers of the
 *                                 single-GPU path.  gpimhip_dist_begin(h, n) = gpimhip_dist_setup(h, n, 1, 0).
 *   gpimhip_dist_panel_factor     loc_blk0: block column (128-wide) where the panel starts inside Aloc;
 *                                 glob_blk0: its global block index (a multiple of 4); logdet_out: up to 4 doubles
 *                                 (device), the sums of log L_ii of the panel's 128-blocks, or NULL; info as
 *                                 gpimhip_potrf.  The step kernels of the single-GPU factorisation (cholstep.hip)
 *                                 with the left-looking window restricted to the panel.
 *   gpimhip_dist_panel_pack       buf (ldbuf >= 512, np + 128 rows): rows [128 glob_blk0, np) <- the panel, rows
 *                                 [np, np + 128) <- the inverses of its diagonal blocks side by side
 *   gpimhip_dist_update           buf: a packed panel with global block index panel_glob_blk0; updates the OWNED
 *                                 panels c with panel_first <= c < panel_last (global panel indices) in Aloc
 *                                 (this rank's panels side by side):  C -= P_rows P_cols^T on tiles i >= j.
 *                                 One launch whatever the number of panels.
 *   gpimhip_dist_solve_update     one panel step of the forward substitution W = L^-1 B for this rank's own
 *                                 right-hand sides B (np x mpad row-major, mpad a multiple of 128, destroyed):
 *                                 W rows of the panel -> Wt (512 x mpad), B rows below -= L W, and
 *                                 q[j] += sum_r W[r][j]^2 (q may be NULL).  col_tiles > 0: only the first col_tiles
 *                                 128-column tiles of B take part (the others are known to be zero).  Replaces the
 *                                 solve_triangular /
 *                                 GEMM pair of `conditional` (gpim/gpreg/gpr.py:247-248) for a factor that is
 *                                 streamed through the ranks panel by panel. */
int gpimhip_dist_setup(gpimhip_handle h, int64_t n, int32_t world, int32_t rank);
int gpimhip_dist_begin(gpimhip_handle h, int64_t n);
int gpimhip_dist_panel_factor(gpimhip_handle h, double* Aloc, int64_t ldloc, int32_t loc_blk0,
                              int32_t glob_blk0, double* logdet_out, int32_t* info);
int gpimhip_dist_panel_pack(gpimhip_handle h, const double* Aloc, int64_t ldloc, int32_t loc_blk0,
                            int32_t glob_blk0, double* buf, int64_t ldbuf);
int gpimhip_dist_update(gpimhip_handle h, const double* buf, int64_t ldbuf, int32_t panel_glob_blk0,
                        double* Aloc, int64_t ldloc, int32_t panel_first, int32_t panel_last);
int gpimhip_dist_solve_update(gpimhip_handle h, const double* buf, int64_t ldbuf, int32_t panel_glob_blk0,
                              double* B, int64_t ldb, int64_t mpad, double* Wt, int64_t ldw, double* q,
                              int32_t col_tiles);
/* ... for two panels held side by side in a buffer of 1024 columns (gpim_amd.dist_chol._stream_pairs): the first panel of
 * a pair (second = 0) updates the second panel's block rows only; after the second panel's own solves (second = 1) the
 * rows below receive both panels in ONE pass of k-depth 1024 -- half the read-modify-write traffic over the rank's
 * right-hand sides.  wide: first panel in columns [0, 512), second in [512, 1024), each packed by gpimhip_dist_panel_pack
 * with ldbuf >= 1024; Wt2: 1024 rows.  The arithmetic per element is that of two consecutive gpimhip_dist_solve_update
 * calls with the two subtractions of a row below the pair summed in one accumulation. */
int gpimhip_dist_solve_update2(gpimhip_handle h, const double* wide, int64_t ldbuf, int32_t panel_glob_blk0, double* Bm,
                               int64_t ldb, int64_t mpad, double* Wt2, int64_t ldw, double* q, int32_t col_tiles,
                               int32_t second);

/* Distributed TRAINING of that one exact GP (gpim/gpreg/gpr.py:170-217 for a covariance that does not fit one
 * device; driver: gpim_amd/dist_chol.py exact_gp_fit).  Per Adam iteration, at the unconstrained parameters u:
 *   gpimhip_dist_kmat_cols    columns [col0, col0 + ncols_pad) of K(u) + (jitter + noise) I into out (np rows; identity
 *                             on the padding diagonal) -- every rank builds its own panels
 *   (factorisation as above, then X = L^-1 by streaming the factor once more through gpimhip_dist_solve_update with
 *    B = this rank's identity columns, Wt = the panel's rows of the rank's X share, col_tiles = the block columns that
 *    are not structurally zero yet)
 *   gpimhip_dist_kinv_update  xbuf: the broadcast block-column panel of X with global block index panel_glob_blk0
 *                             (all np rows x 512); rows of that panel of K^-1 = X^T X for the OWNED columns
 *                             (tiles i >= j) -> Kinv (np x owned columns, like Aloc)
 *   gpimhip_dist_grad_sums    the rank's share of  sum_ij (K^-1 - alpha alpha^T)_ij dK_ij/dtheta  reduced to 8
 *                             doubles S_out (device); the ranks all-reduce (sum) them
 *   gpimhip_dist_finalize     loss (given quad = y^T alpha and half_logdet = sum log L_ii), d loss / du, and -- for
 *                             t >= 1, the 1-based Adam iteration -- torch.optim.Adam's step on u (state in the handle,
 *                             reset at t = 1) and the constrained values of the stepped u in hist_row; t = 0:
 *                             loss and gradient only.  Identical inputs on every rank -> identical u.
 *   gpimhip_dist_finalize_dev the same with the scalars read from DEVICE memory, so that the training loop of
 *                             gpim_amd.dist_chol.exact_gp_fit enqueues an iteration without reading anything back:
 *                             red[0..7] = the all-reduced sums, red[8] = sum log L_ii over all ranks, red[9] = number of
 *                             ranks whose factorisation met a non-positive pivot (if not 0: *loss_out = NaN, nothing else
 *                             is written), quad[0] = y^T alpha. */
int gpimhip_dist_kmat_cols(gpimhip_handle h, const gpimhip_model_t* m, const double* X, int64_t N, const double* u,
                           int64_t col0, int64_t ncols_pad, double* out, int64_t ld);
/* The O(N^2) vector solves of alpha = K^-1 y of the distributed model (torch.linalg.solve_triangular over the whole
 * factor in the reference, gpr.py:192-193 through pyro), panel by panel on the OWNER of panel [glob_blk0, glob_blk0 + 4)
 * with the handle that factored it (the inverses of its diagonal blocks live there):
 *   gpimhip_dist_vec_forward   piece (<= 512) = Lpp^-1 (y_p - t)  (t may be NULL);  acc[rows below the panel] +=
 *                              L(below, p) piece   (acc: full-length np vector of this rank's partial sums)
 *   gpimhip_dist_vec_backward  piece = Lpp^-T (z_p - L(below, p)^T a[below])  (a: full-length np vector; work: 512 doubles)
 *   gpimhip_matvec_t           out (ncols) = A^T x, A row-major nrows x ncols (ncols % 64 == 0): K*^T alpha */
int gpimhip_dist_vec_forward(gpimhip_handle h, const double* Aloc, int64_t ldloc, int32_t loc_blk0, int32_t glob_blk0,
                             const double* y_p, const double* t, double* piece, double* acc);
int gpimhip_dist_vec_backward(gpimhip_handle h, const double* Aloc, int64_t ldloc, int32_t loc_blk0, int32_t glob_blk0,
                              const double* z_p, const double* a, double* work, double* piece);
int gpimhip_matvec_t(gpimhip_handle h, const double* A, int64_t ld, int64_t nrows, int64_t ncols, const double* x,
                     double* out);
int gpimhip_dist_kinv_update(gpimhip_handle h, const double* xbuf, int64_t ldx, int32_t panel_glob_blk0,
                             const double* Xloc, int64_t ldloc, double* Kinv, int64_t ldk);
/* ... for `npanels` consecutive panels in ONE launch: xbuf holds them side by side (np rows x 512 * npanels, the first one
 * with global block index panel_glob_blk0).  Early panels have few tiles, each as deep as the whole matrix: one at a time
 * they leave most of the chip idle (round 6: four per launch). */
int gpimhip_dist_kinv_update_n(gpimhip_handle h, const double* xbuf, int64_t ldx, int32_t panel_glob_blk0, int32_t npanels,
                               const double* Xloc, int64_t ldloc, double* Kinv, int64_t ldk);
int gpimhip_dist_grad_sums(gpimhip_handle h, const gpimhip_model_t* m, const double* X, int64_t N, const double* u,
                           const double* Kinv, int64_t ldk, const double* alpha, double* S_out);
int gpimhip_dist_finalize(gpimhip_handle h, const gpimhip_model_t* m, int64_t N, double* u, const double* S,
                          double quad, double half_logdet, double lr, int32_t t, double* loss_out, double* grad_out,
                          double* hist_row);
int gpimhip_dist_finalize_dev(gpimhip_handle h, const gpimhip_model_t* m, int64_t N, double* u, const double* red,
                              const double* quad, double lr, int32_t t, double* loss_out, double* grad_out,
                              double* hist_row);

/* Symmetry-reduced exact GP on a COMPLETE uniform grid, for kernels that do not factorise over the axes (Matern52,
 * RationalQuadratic; RBF works too): the role of the reference's structured class (gpim/gpreg/skgpr.py:399-448 with
 * gpim/kernels/gpytorch_kernels.py:65), exact instead of interpolated.  mask: bit k = dimension k of the grid is
 * reflection-symmetric (its coordinates are symmetric about their centre); twoc[k] = first + last.
 * With a mask set, gpimhip_fit_exact_batched / gpimhip_predict_exact_batched / gpimhip_nll_grad-style calls treat the
 * B = 2^popcount(mask) problems of a batch as the diagonal blocks of ONE model in the reflection-adapted basis:
 *   X         the fundamental domain (the first half of every reflected axis): N / B points, shared (x_stride 0)
 *   y         B stacked vectors  y_s[p] = B^-1/2 sum_g chi_s(g) y[g p]   (sign pattern s = problem index, bit j = the
 *             j-th reflected dimension in ascending order carries sign -1)
 *   u         B copies of ONE parameter vector (all updated alike); hist_out / loss_out: the slots of problem 0
 *   predict   mean_out, var_out: M doubles (NOT B x M) -- the posterior of the full model at Xs
 * Axes of ODD length: the fundamental domain includes the mirror plane; wts (device, B x (N / B rounded up to the domain's
 * size), or NULL when every reflected axis is even) holds per block and point 2^(-m/2), m = the number of mirror planes the
 * point lies on, and 0 for a point that does not exist in the block (it lies on the mirror plane of an axis whose sign is
 * -1: its combination vanishes) -- such rows are identity rows of the block, and y of the block is 0 there.  n_total = the
 * number of observations of the full model (0: B x N).
 * var_count > 0: predictions compute the variance for the first var_count test points only (var_out keeps M entries; the
 * others are not written) -- the posterior variance is invariant under the reflections, so a caller predicting on the
 * training grid orders the fundamental domain first and mirrors the result (gpim_amd/gpr.py).
 * mask = 0 switches back.  Double precision only. */
int gpimhip_set_reflection(gpimhip_handle h, int32_t mask, const double* twoc, const double* wts, int64_t n_total,
                           int64_t var_count);

/* The blocks of the symmetry-reduced model dealt to the ranks of a job (gpim_amd/dist_symm.py): no data-path collective at
 * all -- the blocks only share the hyper-parameters, so the ranks exchange eleven doubles per Adam iteration.
 *   gpimhip_set_reflection_shard  problem b of this handle's batches is the block of sign pattern pb_off + b * pb_stride out
 *                                 of nblocks_total = 2^r; raw != 0: gpimhip_predict_exact_batched returns in var_out the
 *                                 local blocks' summed quadratic form instead of the variance (the ranks add theirs up:
 *                                 var = max(sigma^2 - sum, 0) + noise) and in mean_out their share of the mean.  Reset by
 *                                 gpimhip_set_reflection(h, 0, ...).
 *   gpimhip_refl_sums             one evaluation of the local blocks at u (B copies of the parameter vector): kernel
 *                                 matrices, factorisations, inverses, gradient contraction; sums_out (device, 11 doubles):
 *                                 [0..7] gradient sums, [8] sum log L_ii, [9] 1 if a factorisation failed, [10] sum
 *                                 |L^-1 y|^2.  After the all-reduce, gpimhip_dist_finalize_dev takes [0..9] and [10]. */
int gpimhip_set_reflection_shard(gpimhip_handle h, int32_t pb_off, int32_t pb_stride, int32_t nblocks_total, int32_t raw);
int gpimhip_refl_sums(gpimhip_handle h, const gpimhip_model_t* m, const double* X, const double* y, int64_t N, int32_t B,
                      const double* u, double* sums_out);

/* Batch thinning of boptimizer.update_points (gpim/gpbayes/boptim.py:326-376): among n <= 1024 ranked
 * candidates (vals, flat grid indices into a d-dimensional grid of the given shape) repeatedly keep the
 * largest remaining value and drop every candidate within Euclidean index distance <= dscale of it
 * (scipy.spatial.cKDTree.query_ball_point semantics), until none is left or max_out are kept.
 *   shape      d int64 (device);  keep_out  max_out int32 (device): kept positions 0..n-1, in order;
 *   nkeep_out  1 int32 (device).  The random padding of short batches stays with the caller (np.random). */
int gpimhip_thin_batch(gpimhip_handle h, const double* vals, const int64_t* flat_idx, int32_t n, int32_t d,
                       const int64_t* shape, double dscale, int32_t max_out, int32_t* keep_out,
                       int32_t* nkeep_out);

/* The launch plan of the factorisation (and of the triangular inverse riding in its launches, with_inverse != 0) for a
 * double-precision matrix of nb block columns, as HOST data -- no device involved; for tests that replay the schedule
 * on small blocks (tests/test_step_plan.py).  Records of six int32 in `out` (capacity `cap` records): launch index
 * (< nb: the step launch that factors that block column; >= nb: the launches after the last step), ci, cj, kb0, kb1,
 * kind (0: A[ci,cj] -= sum_k L[ci,k] L[cj,k]^T; 1 / 2: Tm[ci,cj] (=|+=) sum_k A[ci,k] A[k,cj]; 3 / 4: A[ci,cj] (=|-=)
 * -sum_k A[ci,k] Tm[k,cj]; k over block columns [kb0, kb1)).  *n_out = number of records of the plan. */
int gpimhip_step_plan_host(int32_t nb, int32_t with_inverse, int32_t* out, int64_t cap, int64_t* n_out);

/* Stage timing for bench.py (HIP events on the handle's stream, recorded only while enabled).
 * stage: 0 = Cholesky (all launches of one factorisation, including the tile operations of the triangular inverse
 *            they host), 1 = what is left of the triangular inverse after the last step,
 *        2 = K^-1 = L^-T L^-1 (exactly one gemm_tiles_kernel<true,true,0> launch, N^3/3 flop),
 *        3 = predictive-variance product L^-1 K(X,X*) (one launch per test-point slab).
 * gpimhip_timing_read synchronises, returns the summed milliseconds and the number of timed
 * intervals since the last read, and clears them. */
int gpimhip_timing_enable(gpimhip_handle h, int enable);
int gpimhip_timing_read(gpimhip_handle h, int stage, double* total_ms, int64_t* count);

/* Number of Adam iterations the last gpimhip_fit_exact[_batched] / gpimhip_fit_vfe call completed: T,
 * or -- when that call returned GPIMHIP_E_NOT_PD -- the index of the iteration whose covariance was not
 * positive-definite.  The device loop freezes u, the optimiser state and the history rows at that
 * iteration (the reference raises there, gpr.py:192, with the parameters of the previous step), so
 * u_inout and the first gpimhip_fit_completed() rows of hist_out / loss_out are valid after the error. */
int gpimhip_fit_completed(gpimhip_handle h);

/* Releases the process-wide helper streams of the library (created on first use, shared by all handles).  Call once,
 * after the last handle has been destroyed and before the process exits; the Python binding registers it with
 * atexit.  Handles created afterwards recreate what they need. */
int gpimhip_shutdown(void);

/* Block until everything enqueued on the handle's stream has finished. */
int gpimhip_sync(gpimhip_handle h);

#ifdef __cplusplus
}
#endif
#endif /* GPIMHIP_H */
