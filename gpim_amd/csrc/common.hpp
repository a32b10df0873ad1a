// common.hpp -- shared declarations of libgpimhip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include <utility>
#include "../../include/gpimhip.h"

#define NB 128            // block size of every blocked algorithm (rows/cols of one tile)
#define GEMM_BK 16        // k-depth of one LDS stage of the MFMA GEMM
#define MAXP GPIMHIP_MAX_PARAMS

void gpim_set_error(const std::string& s);

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            gpim_set_error(std::string(#expr) + ": " + hipGetErrorString(_e));     \
            return GPIMHIP_E_HIP;                                                  \
        }                                                                          \
    } while (0)

#define GP_TRY(expr)                      \
    do {                                  \
        int _r = (expr);                  \
        if (_r != GPIMHIP_OK) return _r;  \
    } while (0)

// One output tile of a blocked GEMM-like launch: C[ci,cj] (op)= sum_{kb in [kb0,kb1)} A(ci,kb) B(kb,cj)
struct TileDesc {
    int32_t ci, cj, kb0, kb1;
};

// Device-resident constants of one model instance, refreshed from u each iteration.
// Layout shared by host and device code.
struct ThetaDev {
    double var;                     // sigma^2
    double ls[GPIMHIP_MAX_DIM];     // lengthscale per input dim (isotropic: replicated)
    double inv_ls[GPIMHIP_MAX_DIM];
    double noise;                   // sigma_n^2
    double alpha;                   // RationalQuadratic scale_mixture
    double diag_add;                // jitter + noise
    // chain-rule factors d theta / d u (0 where the clipped sigmoid saturates)
    double dvar_du;
    double dls_du[GPIMHIP_MAX_DIM];
    double dnoise_du;
    double dalpha_du;
};

struct AdamStep {
    double lr_over_bc1;     // lr / (1 - beta1^t)
    double bc2_sqrt;        // sqrt(1 - beta2^t)
    double beta1, beta2, eps;
};

// A cached launch plan: tile lists of the level-by-level triangular inverse and of the K^-1 product at a given nb.
struct PlanRange { int64_t off; int32_t n; };

struct LinalgPlan {
    int nb = 0;
    TileDesc* d_tiles = nullptr;          // device copy of all tile lists
    int64_t n_tiles = 0;
    // trtri: per level two launches
    std::vector<PlanRange> tri_t, tri_x;
    // lauum
    PlanRange lauum{0, 0};
};

// Launch plan of the step schedule of the Cholesky (cholstep.hip): per block column the tiles hosted by the launch
// that factors its diagonal block and the diagonal tiles updated after its panel solve; per outer panel what is
// left of the previous panel's bulk update.
struct StepPlan {
    int nb = 0;
    int fp32 = 0;                       // hosting policy the lists were built for (cholstep.hip)
    TileDesc* d_tiles = nullptr;
    int64_t n_tiles = 0;
    std::vector<PlanRange> fill;        // per block column: what its step launch hosts
    std::vector<PlanRange> diag;        // per block column: .n = diagonal tiles updated after its panel solve
    std::vector<PlanRange> bulk_rest;   // per outer panel: bulk update launched before its first step (float handles)
    std::vector<PlanRange> post;        // fused inverse: what is left of it after the last step, launch by launch
    std::vector<int32_t> n_update;      // per block column: k-blocks of trailing update among fill[] (sets the hosted shape)
    std::vector<int32_t> n_all;         // per block column: all hosted k-blocks (shape of the launches of a large batch)
    std::vector<uint8_t> pair;          // per block column: its step launch runs hosted quadrants two per CU (cholstep.hip)
    std::vector<uint8_t> copy;          // per block column: its step launch leaves a copy of A[j+1, j] (TILE_COPY): F_j and D_j fuse
};

// Launch plans of the distributed (block-column-cyclic, 1 x P) factorisation for one rank (api.hip: gpimhip_dist_*)
struct DistPlan {
    int nb = 0, world = 0, rank = 0;
    TileDesc* d_tiles = nullptr;
    int64_t n_tiles = 0;
    std::vector<PlanRange> colfill;     // per global block column j: the in-panel left-looking column update
    std::vector<PlanRange> upd_panel;   // per global panel c: the tiles of its trailing update (empty when not owned)
    std::vector<PlanRange> kinv_panel;  // per global panel c: tiles of K^-1 rows of panel c x owned columns (training)
    PlanRange grad_tiles{0, 0};         // the owned lower tiles of K^-1 for the gradient contraction
    int rect_cols = 0;                  // column-tile count the rectangle list below was built for
    TileDesc* d_rect = nullptr;         // tiles (r, c), r = 0 .. nb-1, c = 0 .. rect_cols-1, row-major
    int64_t n_rect = 0;
};

// Reflection symmetry of a complete, uniform grid under a stationary kernel that is even in every coordinate difference
// (gpimhip_set_reflection; engine.hip: kmat_refl_kernel): mask = the reflected dimensions (bit k), twoc[k] = first + last
// coordinate of dimension k, so that the mirror image of z_k is twoc[k] - z_k.
// wts (optional, device, B x N): per block and point 1 / sqrt(|stabiliser|) -- 1 off the mirror planes, 2^-1/2 on one
// plane (axes of odd length), ... -- and 0 where the point does not exist in that block (a point on the mirror plane of
// an axis whose sign is -1): such rows are identity rows of K_s.  n_total: the number of observations of the full model.
struct ReflArgs {
    int mask;
    double twoc[GPIMHIP_MAX_DIM];
    const double* wts;
    int64_t n_total;
    int64_t var_count;      // > 0: predictions compute the variance for the first var_count test points only
    // blocks dealt to the ranks of a job (gpimhip_set_reflection_shard): problem b of the local batch is the block of sign
    // pattern pb_off + b * pb_stride; |G| = nblocks_total (the K* scale).  raw: predictions return, instead of the variance,
    // the blocks' summed quadratic form  sum_s |L_s^-1 K*_s|^2  (the ranks add theirs up before subtracting from sigma^2)
    int pb_off, pb_stride, nblocks_total, raw;
};
struct gpimhip_ctx {
    ReflArgs refl = {0, {0, 0, 0, 0}, nullptr, 0, 0, 0, 1, 0, 0};
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t capture_stream = nullptr; // internal stream used only to capture one iteration into a hipGraph
    bool capture_stream_tried = false;    // the capture stream is created on first use
    int side_generation = 0;              // generation of the process-wide capture stream the pointer below belongs to (api.hip)
    bool capturing = false;               // fit_impl is recording one iteration into a hipGraph
    // workspace (sized for np = padded N)
    int64_t np = 0;                 // padded matrix order the buffers are sized for
    int64_t ld = 0;                 // leading dimension (doubles) of A, B, Tm: np, or np + 16 (see ws_ensure_b)
    int nbatch = 1;                 // problems processed in lock-step by the current call (grid.y)
    int ws_batch = 0;               // number of problems the workspace is sized for
    double* A = nullptr;            // np x np : K -> L -> L^-1
    double* B = nullptr;            // np x np : K^-1 (lower)
    double* Tm = nullptr;           // np x np : trtri temporary
    double* dinv = nullptr;         // nb x 128 x 128 inverses of diagonal blocks
    double* dinvB = nullptr;        // the same in MFMA B-operand order (panel solve of cholstep.hip); fp64 handles
    double* pcopy = nullptr;        // B x 128 x 128: A[j+1, j] as the step launch of column j left it (cholstep.hip: panel_solve_diag_kernel)
    double* ypad = nullptr;         // np
    double* z = nullptr;            // np  (L^-1 y)
    double* alpha = nullptr;        // np  (K^-1 y)
    double* logdet_part = nullptr;  // nb
    double* grad_part = nullptr;    // ntiles_lower x 8
    double* gemv_part = nullptr;    // gemv_tri_chunks(np) x np: row-chunk partial sums of alpha = L^-T z (engine.hip: gemv_t_tri_kernel)
    uint32_t* fin_counter = nullptr;   // per problem: workgroups of grad_reduce_kernel that have finished (engine.hip: FinFused; the last
                                       // one runs the finalize step and resets it)
    ThetaDev* theta = nullptr;      // [ws_batch]
    ThetaDev* theta1 = nullptr;     // single struct for the operator-level gpimhip_kmat
    int32_t* iter = nullptr;        // [ws_batch] device-side iteration counters
    double* adam_m = nullptr;       // MAXP
    double* adam_v = nullptr;       // MAXP
    int32_t* info = nullptr;        // potrf status word
    // prediction workspace
    int64_t ks_rows = 0, ks_cols = 0, ks_batch = 0;   // ks_cols = CAPACITY in test-point columns (grow-only)
    struct PredList { int nc; TileDesc* tiles; int64_t n; };
    std::vector<PredList> pred_lists;                // variance-product tile lists by number of column blocks
    double* Ks = nullptr;           // np x mc chunk of K(X, X*)
    double* colpart = nullptr;      // nb x mc partial column sums of squares
    double* mean_tmp = nullptr;     // mc
    TileDesc* pred_tiles = nullptr; // tile list of the variance product
    int64_t pred_ntiles = 0;
    int64_t bytes = 0;
    LinalgPlan plan;
    StepPlan splan, splan_inv;      // factorisation alone / with the triangular inverse riding in its launches
    DistPlan dplan;
    // optional stage timing (bench.py): HIP event pairs on the handle's stream
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[4];
    // Adam bias-correction table for the fused small-N trainer
    double* bc = nullptr;
    int64_t bc_cap = 0;
    std::vector<double> bc_host;
    // top-k scratch
    unsigned long long* keys = nullptr;
    int64_t keys_cap = 0;
    double* acq_tmp = nullptr; int64_t acq_tmp_cap = 0;   // observed-rows posterior + incumbent (gpimhip_acquire_exact)
    char* sel_scratch = nullptr;    // radix-select state, candidates, nanmax partials (select.hip)
    // training-loop bookkeeping: iterations completed by the last fit call, pinned copy of the status
    // word + events of the bounded run-ahead check (api.hip: RunAhead)
    int fit_completed = 0;
    int32_t* pinned_info = nullptr;
    hipEvent_t ra_ev[2] = {nullptr, nullptr};
    // sparse (VFE) workspace, owned by vfe.hip (VfeWs*); released by vfe_release()
    void* vfe = nullptr;
    // structured (Kronecker) workspace, owned by kron.hip (KronWs*); released by kron_release()
    void* kron = nullptr;
    double* refine = nullptr; int64_t refine_cap = 0;    // residual, correction and partial sums of the refinement (fp32 handles)
    int fp32 = 0;                   // 1: the N x N matrices of the exact-GP path are float (gpimhip_set_precision)
};

// entry points outside the exact-GP path keep their matrices in double
#define FP64_ONLY(h)                                                                                      \
    do {                                                                                                  \
        if ((h) && (h)->fp32) {                                                                           \
            gpim_set_error("this entry point needs a double-precision handle (gpimhip_set_precision(h, 64))"); \
            return GPIMHIP_E_BADARG;                                                                      \
        }                                                                                                 \
    } while (0)

static inline int64_t pad_to(int64_t n, int64_t m) { return (n + m - 1) / m * m; }
// GemmArgs::rag for an exact-GP matrix of N valid rows padded to np with the identity
static inline int rag_of(int64_t N, int64_t np) {
    const int64_t last = N - (np - NB);
    return (np >= 2 * NB && last > 0 && last <= 64) ? (int)(np / NB) : 0;
}

// stage timers (bench.py): 0 factorisation (with the part of the inverse its launches host), 1 rest of the triangular
// inverse, 2 K^-1 product (one tile-engine launch), 3 predictive-variance product
struct StageTimer {
    gpimhip_ctx* h; int stage; hipEvent_t e1 = nullptr;
    StageTimer(gpimhip_ctx* h_, int s) : h(h_), stage(s) {
        if (!h->timing) return;
        hipEvent_t e0;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e1 = nullptr; return; }
        (void)hipEventRecord(e0, h->stream);
        h->ev[stage].push_back({e0, e1});
    }
    ~StageTimer() { if (e1) (void)hipEventRecord(e1, h->stream); }
};

// ---- drivers shared between translation units ----
// row chunks of the triangular transposed mat-vec alpha = L^-T z (engine.hip: gemv_t_tri_kernel): ~np / 32 rows, a multiple
// of 128 between 128 and 1024; a function of the matrix order alone (a problem gives the same bits alone and in a batch)
static inline int gemv_tri_rc(int64_t np) {
    const int64_t rc = ((np / 32 + NB - 1) / NB) * NB;
    return (int)(rc < NB ? NB : (rc > 1024 ? 1024 : rc));
}
static inline int gemv_tri_chunks(int64_t np) { return (int)((np + gemv_tri_rc(np) - 1) / gemv_tri_rc(np)); }
int ws_ensure(gpimhip_ctx* h, int64_t N);
int ws_ensure_predict(gpimhip_ctx* h, int64_t np, int64_t mc);
int plan_ensure(gpimhip_ctx* h, int nb);
// cholstep.hip
int launch_potrf_steps(gpimhip_ctx* h, double* A, int64_t np, int64_t ld, int32_t* info, double* Tm = nullptr, int rag = 0);
void step_plan_release(gpimhip_ctx* h);
int step_plan_ensure(gpimhip_ctx* h, int nb);
int step_plan_ensure_inv(gpimhip_ctx* h, int nb);
int launch_panel_chain(gpimhip_ctx* h, double* A, int64_t ld, int p0, int p1, int nb, const TileDesc* tiles,
                       const PlanRange* colfill, int32_t* info);
// distops.hip
int launch_dist_pack(gpimhip_ctx* h, const double* P, int64_t ldp, int64_t r0, int64_t np, int w, const double* dinv,
                     int nblk, double* buf, int64_t ldb);
int launch_colsumsq_acc(gpimhip_ctx* h, const double* W, int64_t ldw, int rows, int64_t m, double* q);
int launch_dist_trsv(gpimhip_ctx* h, const double* P, int64_t ld, const double* D, int nblk, int backward, const double* r0,
                     const double* r1, double* out);
int launch_dist_rows_acc(gpimhip_ctx* h, const double* A, int64_t ld, int64_t rows, int w, const double* x, double* acc);
// engine.hip (distributed training)
int launch_grad_reduce_tiles(gpimhip_ctx* h, const gpimhip_model_t* m, const double* Kinv, int64_t ld, const double* X,
                             int64_t N, int64_t np, const double* alpha, const TileDesc* tiles, int ntile, double* part);
int launch_sum7(gpimhip_ctx* h, const double* part, int ntile, double* S);
int launch_kmat_refl(gpimhip_ctx* h, const gpimhip_model_t* m, const double* X, int64_t N, const double* Z, int64_t M,
                     const ThetaDev* theta, double* out, int64_t ld, int64_t rows_pad, int64_t cols_pad, int sym, int64_t x_bs,
                     int64_t z_bs, int64_t out_bs, double scale);
int launch_grad_reduce_refl(gpimhip_ctx* h, const gpimhip_model_t* m, const double* Kinv, int64_t ld, const double* X,
                            int64_t N, int nb, const double* alpha, int64_t x_bs);
int launch_finalize_coupled(gpimhip_ctx* h, const gpimhip_model_t* m, int64_t N, int64_t np, double* u, int do_adam,
                            AdamStep st, double* loss_out, double* grad_out, double* hist_row, int32_t* iter,
                            const double* bc, int T, double* hist_base, double* loss_base);
int launch_coupled_sums(gpimhip_ctx* h, int64_t np, double* out11);
int launch_predict_coupled(gpimhip_ctx* h, int64_t ldp, int nb, int64_t m0, int64_t mcount, int64_t nvar, int64_t mean_bs,
                           double* mean_out, double* var_out);
int launch_dist_finalize_dev(gpimhip_ctx* h, const gpimhip_model_t* m, int64_t N, const double* red, const double* quad,
                             double* u, int do_adam, AdamStep st, double* loss_out, double* grad_out, double* hist_row);
int launch_dist_finalize(gpimhip_ctx* h, const gpimhip_model_t* m, int64_t N, const double* S, double q2, double lg,
                         double* u, int do_adam, AdamStep st, double* loss_out, double* grad_out, double* hist_row);
int launch_add_diag_theta(gpimhip_ctx* h, double* out, int64_t ld, int64_t row0, int64_t n, int64_t npad);
int launch_theta(gpimhip_ctx* h, const gpimhip_model_t* m, const double* u);
// predict.hip
bool fused_predict_fits(int64_t np);
int launch_predict_fused(gpimhip_ctx* h, const gpimhip_model_t* m, const double* X, int64_t x_bs, int64_t N,
                         const double* Xs, int64_t M, double* mean_out, double* var_out, double* sd_out,
                         double* acq_out, int acq_kind, double p0, const double* p0_dev, double p1,
                         const double* mask);
int launch_acq_from_var(gpimhip_ctx* h, int kind, const double* mean, const double* var, int64_t M, double p0,
                        const double* p0_dev, double p1, const double* mask, double* sd_out, double* acq_out);


enum GemmEpi { EPI_STORE = 0, EPI_COLSUMSQ = 1 };
struct GemmArgs {
    const double* A; int64_t lda; int a_roff, a_coff;
    const double* B; int64_t ldb; int b_roff, b_coff;
    double* C; int64_t ldc; int c_roff, c_coff;
    double alpha, beta;
    const TileDesc* tiles; int ntiles;
    int cj_max;                            // > 0: tiles with cj >= cj_max are skipped (structurally zero columns)
    int cmap;                              // C's block column is the tile's kb0 field (operand columns stay cj): output
                                           // stored at local columns (distributed layouts); needs kfix0 / kfix1
    int chunk;                             // XCD dealing: 0 = contiguous slices, >0 = round-robin chunks
    int inplace;                           // C aliases an operand tile (panel solve): one workgroup must own the whole tile
    // INVARIANT of ragged launches: the padding rows of the last block row of Tm and B (workspace matrices) are zeroed at
    // allocation only and never written by a rag launch; every reader of Tm / B under rag skips them too (X phase with
    // kb1 == nb, lauum with krev, the COLSUMSQ predict path).  Non-rag fits and the sparse model on the same handle write
    // those rows fully, so after them they hold stale data: a NEW consumer of Tm or B (a full-np reduction over K^-1, say)
    // must either skip the padding rows or zero them first.
    int rag;                               // > 0: block rag - 1 (the LAST block of the matrix order) holds at most 64 valid
                                           // rows / columns, the rest is identity padding: output rows >= 64 of block row
                                           // rag - 1 and the k-steps >= 64 of a range ending with that block are structurally
                                           // zero and skipped (exact GP with (N - 1) % 128 < 64: rag_of())
    int krev;                              // walk each tile's k-range from its end (ranges sharing their upper end)
    int kfix0, kfix1;                      // when kfix1 > kfix0: every tile uses this k-block range (its own is ignored)
    int rect_rows, rect_cols;              // rect_cols > 0: no list -- the launch covers the rect_rows x rect_cols tiles of a
                                           // rectangle (ntiles = their product) in strips of eight rows, column by column
                                           // inside a strip: 64 consecutive tiles are an 8 x 8 patch (with chunk = 64 one
                                           // patch per XCD at a time: 8 + 8 operand panels for 64 tiles); needs kfix0 / kfix1
    double* colpart; int64_t ld_colpart;   // EPI_COLSUMSQ: colpart[ci*ld + cj*128 + col]
    double* C2;                            // != nullptr: the output tile is ALSO stored here as a dense 128 x 128 block (set per tile by
                                           // the Cholesky step kernel: the copy of A[j+1, j] that the fused panel-solve / diagonal-update launch reads)
    int shape_div;                         // > 1: the launch shape is chosen for ntiles * batch / shape_div tiles (the sparse model's
                                           // lock-step batches: every model gets the tile shapes -- hence the bits -- of its own launch)
    int64_t sA, sB, sC, sColpart;          // per-problem (blockIdx.y) strides in elements
};
int launch_gemm(gpimhip_ctx* h, bool a_km, bool b_km, int epi, const GemmArgs& g);
