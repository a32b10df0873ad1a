// api.hip -- host side of libgpimhip: workspace, launch plans, blocked-algorithm drivers and the
// extern "C" entry points declared in include/gpimhip.h.
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <chrono>
#include <atomic>
#include <mutex>
#include "common.hpp"

// kernels implemented in the other translation units
int launch_theta_raw(gpimhip_ctx* h, const gpimhip_model_t* m, const double* raw);
int launch_kmat(gpimhip_ctx* h, const gpimhip_model_t* m, const double* X, int64_t N, const double* Z,
                int64_t M, const ThetaDev* theta, double diag_add, int use_theta_diag, double* out,
                int64_t ld, int64_t rows_pad, int64_t cols_pad, int sym, int lower_only, int64_t x_bs,
                int64_t z_bs, int64_t out_bs);
int launch_pad_copy(gpimhip_ctx* h, const double* src, int64_t n, double* dst, int64_t np);
int launch_diag_inv_copy(gpimhip_ctx* h, double* A, int64_t ld, int nb);
int launch_pad_matrix_in(gpimhip_ctx* h, const double* src, int64_t n, int64_t ld, double* dst, int64_t np);
int launch_pad_matrix_out_lower(gpimhip_ctx* h, const double* src, int64_t np, double* dst, int64_t n, int64_t ld);
int launch_trmv_lower(gpimhip_ctx* h, const double* L, int64_t ld, int64_t np, const double* y, double* z);
int launch_gemv_t(gpimhip_ctx* h, const double* A, int64_t ld, int64_t nrows, int64_t ncols, const double* x,
                  double* out, int tri, int64_t a_bs, int64_t x_bs, int64_t o_bs);
int launch_grad_reduce(gpimhip_ctx* h, const gpimhip_model_t* m, const double* Kinv, int64_t ld,
                       const double* X, int64_t N, int nb, const double* alpha, int64_t x_bs);
int launch_kres(gpimhip_ctx* h, const gpimhip_model_t* m, const double* X, int64_t x_bs, int64_t N, double* scratch,
                int S, double* res);
int launch_axpy(gpimhip_ctx* h, double* x, const double* d, int64_t n);
int launch_gemv_t_tri(gpimhip_ctx* h, const double* A, int64_t ld, int64_t np, const double* x, double* part, double* out);
int launch_grad_reduce_fin(gpimhip_ctx* h, const gpimhip_model_t* m, const double* Kinv, int64_t ld, const double* X,
                           int64_t N, int nb, const double* alpha, int64_t x_bs, const double* alpha_part, double* u,
                           int do_adam, AdamStep st, double* loss_out, double* grad_out, double* hist_row, int32_t* iter,
                           const double* bc, int T, double* hist_base, double* loss_base, int carry_theta);
int launch_finalize(gpimhip_ctx* h, const gpimhip_model_t* m, int64_t N, int64_t np, double* u, int do_adam,
                    AdamStep st, double* loss_out, double* grad_out, double* hist_row, int32_t* iter,
                    const double* bc, int T, double* hist_base, double* loss_base, int carry_theta = 0);
int launch_predict_var(gpimhip_ctx* h, int64_t ldp, int nb, int64_t m0, int64_t mcount, double* var_out, int64_t M);
int launch_copy_slice(gpimhip_ctx* h, const double* src, double* dst, int64_t n, int64_t s_bs, int64_t d_bs);
int launch_acq(gpimhip_ctx* h, int kind, const double* mean, const double* sd, int64_t M, double p0, double p1,
               const double* mask, double* out);
int launch_nanmax(gpimhip_ctx* h, const double* x, int64_t n, double* out);
int launch_topk(gpimhip_ctx* h, const double* x, int64_t M, int k, int keep_nan, double* vals, int64_t* idx,
                int64_t* count);
int launch_topk_radix(gpimhip_ctx* h, const double* x, int64_t M, int k, int keep_nan, double* vals, int64_t* idx,
                      int64_t* count);
int launch_nanmax_two_stage(gpimhip_ctx* h, const double* x, int64_t n, double* out);
int launch_thin_batch(gpimhip_ctx* h, const double* vals, const int64_t* flat, int n, int d, const int64_t* shape,
                      double dscale, int max_out, int32_t* keep_out, int32_t* nkeep_out);
size_t sel_scratch_bytes();
int launch_fit_small(gpimhip_ctx* h, const gpimhip_model_t* m, const double* X, int64_t x_bs, const double* y,
                     int N, double* u, const double* lr_over_bc1, const double* bc2_sqrt, int T, double* hist,
                     double* loss, double* grad);

static thread_local std::string g_err;
void gpim_set_error(const std::string& s) { g_err = s; }

// XCD dealing chunk for lists sorted by decreasing cost: 64-tile chunks keep neighbouring tiles (shared operand
// panels) on one XCD's L2, but a list of a few hundred tiles dealt 64 at a time puts all the longest tiles on
// XCD 0 (N = 4206: K^-1 product 0.92 -> 0.72 ms with single-tile dealing); grow the chunk with the list.
static int deal_chunk(int ntiles = 1 << 30) { return std::max(1, std::min(64, ntiles / 512)); }
// "large N": from 12 outer panels (6144 unknowns) on an iteration is enqueued launch by launch on the caller's stream
// (below: one captured iteration is replayed; its ~100 launches are then one host call)
#define EAGER_MIN_PANELS 12
#define OUTER_W 4    // outer Cholesky panel = 4 x 128 columns

// ------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------
template <typename T>
static int dev_alloc(gpimhip_ctx* h, T** p, int64_t count) {
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, (size_t)count * sizeof(T));
    if (e != hipSuccess) {
        gpim_set_error(std::string("hipMalloc failed: ") + hipGetErrorString(e));
        return GPIMHIP_E_NOMEM;
    }
    *p = (T*)q;
    h->bytes += count * (int64_t)sizeof(T);
    return GPIMHIP_OK;
}
template <typename T>
static void dev_free(gpimhip_ctx* h, T** p, int64_t count) {
    if (*p) {
        (void)hipFree(*p);
        h->bytes -= count * (int64_t)sizeof(T);
        *p = nullptr;
    }
}

// the N x N matrices hold floats on a single-precision handle: same element counts, half the doubles
static inline int64_t mat_doubles(const gpimhip_ctx* h, int64_t elements) { return h->fp32 ? (elements + 1) / 2 : elements; }

static void ws_release_matrix(gpimhip_ctx* h) {
    const int64_t np = h->np, nb = np / NB, B = h->ws_batch;
    if (!np) return;
    dev_free(h, &h->A, mat_doubles(h, B * np * h->ld));
    dev_free(h, &h->B, mat_doubles(h, B * np * h->ld));
    dev_free(h, &h->Tm, mat_doubles(h, B * np * h->ld));
    dev_free(h, &h->dinv, mat_doubles(h, B * nb * NB * NB));
    dev_free(h, &h->dinvB, B * nb * NB * NB);
    dev_free(h, &h->pcopy, B * NB * NB);
    dev_free(h, &h->ypad, B * np);
    dev_free(h, &h->z, B * np);
    dev_free(h, &h->alpha, B * np);
    dev_free(h, &h->logdet_part, B * nb);
    dev_free(h, &h->grad_part, B * nb * (nb + 1) / 2 * 8);
    dev_free(h, &h->gemv_part, B * (int64_t)gemv_tri_chunks(np) * np);
    dev_free(h, &h->fin_counter, (int64_t)B);
    dev_free(h, &h->theta, B);
    dev_free(h, &h->adam_m, B * MAXP);
    dev_free(h, &h->adam_v, B * MAXP);
    dev_free(h, &h->iter, B);
    h->np = 0;
    h->ws_batch = 0;
}

// Workspace for B problems of N observations processed in lock-step (all per-problem buffers are
// stacked: problem b lives at base + b * size).
// padded != 0: rows of the np x np matrices are np + 16 doubles apart.  With ld = np = 2^k every row
// of a 128-column panel maps to the same L2 sets (row stride 2^(k+3) bytes) and tiles that share a
// panel evict each other's lines; one extra cache line per row spreads the rows over the sets.
// matrices == false: only what the diagonal-block kernels and the launch plan need (dinv, log-det partials,
// theta, Adam state) -- the distributed factorisation keeps its share of the matrix in caller-owned storage
// and never needs the three np x np buffers (3 x 32 GiB at N = 65536).
static int ws_ensure_b(gpimhip_ctx* h, int64_t N, int B, int padded, bool matrices = true) {
    const int64_t np = pad_to(std::max<int64_t>(N, 1), NB);
    const int64_t ld = np + ((padded && np >= 1024) ? (h->fp32 ? 32 : 16) : 0);      // one extra 128-byte line per row
    if (np == h->np && B == h->ws_batch && ld == h->ld && (h->A != nullptr || !matrices)) return GPIMHIP_OK;
    HIP_TRY(hipStreamSynchronize(h->stream));
    ws_release_matrix(h);
    const int64_t nb = np / NB;
    // sizes first: if an allocation in the middle fails, ws_release_matrix() frees what exists (it
    // frees by pointer, with these sizes for the byte accounting)
    h->np = np;
    h->ws_batch = B;
    h->ld = ld;
    int rc = GPIMHIP_OK;
    if ((matrices && ((rc = dev_alloc(h, &h->A, mat_doubles(h, B * np * ld))) ||
                      (rc = dev_alloc(h, &h->B, mat_doubles(h, B * np * ld))) ||
                      (rc = dev_alloc(h, &h->Tm, mat_doubles(h, B * np * ld))))) ||
        (rc = dev_alloc(h, &h->dinv, mat_doubles(h, B * nb * NB * NB))) ||
        (rc = dev_alloc(h, &h->dinvB, B * nb * NB * NB)) ||
        (rc = dev_alloc(h, &h->pcopy, (int64_t)B * NB * NB)) ||
        (rc = dev_alloc(h, &h->ypad, B * np)) || (rc = dev_alloc(h, &h->z, B * np)) ||
        (rc = dev_alloc(h, &h->alpha, B * np)) || (rc = dev_alloc(h, &h->logdet_part, B * nb)) ||
        (rc = dev_alloc(h, &h->grad_part, B * nb * (nb + 1) / 2 * 8)) || (rc = dev_alloc(h, &h->gemv_part, B * (int64_t)gemv_tri_chunks(np) * np)) || (rc = dev_alloc(h, &h->fin_counter, (int64_t)B)) || (rc = dev_alloc(h, &h->theta, (int64_t)B)) ||
        (rc = dev_alloc(h, &h->adam_m, (int64_t)B * MAXP)) || (rc = dev_alloc(h, &h->adam_v, (int64_t)B * MAXP)) ||
        (rc = dev_alloc(h, &h->iter, (int64_t)B))) {
        ws_release_matrix(h);
        return rc;
    }
    HIP_TRY(hipMemsetAsync(h->fin_counter, 0, (size_t)B * sizeof(uint32_t), h->stream));   // (the last workgroups reset them)
    if (matrices) {
        // Tm / B: a ragged last block (GemmArgs::rag) never writes the rows of its identity padding
        HIP_TRY(hipMemsetAsync(h->Tm, 0, (size_t)mat_doubles(h, B * np * ld) * sizeof(double), h->stream));
        HIP_TRY(hipMemsetAsync(h->B, 0, (size_t)mat_doubles(h, B * np * ld) * sizeof(double), h->stream));
    }
    if (h->fp32 && h->refine_cap < 10 * (int64_t)B * np) {
        dev_free(h, &h->refine, h->refine_cap);
        h->refine_cap = 0;
        if ((rc = dev_alloc(h, &h->refine, 10 * (int64_t)B * np))) {
            ws_release_matrix(h);       // the next call with this N must not take the early return without it
            return rc;
        }
        h->refine_cap = 10 * (int64_t)B * np;
    }
    // launch plans of the single-GPU path (built here: building one synchronises the stream, which a graph
    // capture does not allow); the distributed factorisation has its own (gpimhip_dist_setup)
    if (matrices) {
        rc = plan_ensure(h, (int)nb);
        // fit / predict invert the factor in the factorisation's launches (double precision); float matrices and
        // gpimhip_potrf use the plain plan, which is built on first use (never inside a capture)
        if (rc == GPIMHIP_OK) rc = h->fp32 ? step_plan_ensure(h, (int)nb) : step_plan_ensure_inv(h, (int)nb);
        if (rc != GPIMHIP_OK) { ws_release_matrix(h); return rc; }     // (a later call must not find buffers without plans)
    }
    return GPIMHIP_OK;
}
int ws_ensure(gpimhip_ctx* h, int64_t N) { return ws_ensure_b(h, N, h->nbatch, 0); }
static int ws_ensure_padded(gpimhip_ctx* h, int64_t N) { return ws_ensure_b(h, N, h->nbatch, 1); }

static void ws_release_predict(gpimhip_ctx* h) {
    dev_free(h, &h->Ks, mat_doubles(h, h->ks_batch * h->ks_rows * (h->ks_cols + 16)));
    dev_free(h, &h->colpart, h->ks_batch * (h->ks_rows / NB) * h->ks_cols);
    dev_free(h, &h->mean_tmp, h->ks_batch * h->ks_cols);
    for (auto& pl : h->pred_lists) dev_free(h, &pl.tiles, pl.n);
    h->pred_lists.clear();
    h->pred_tiles = nullptr;
    h->pred_ntiles = 0;
    h->ks_rows = h->ks_cols = 0;
    h->ks_batch = 0;
}

// Prediction workspace for slabs of up to mc test points.  Grow-only in mc: a Bayesian-optimisation step
// alternates between the full grid and the handful of observed points, and re-allocating the slab on every
// switch costs more than the prediction itself.  Buffers are laid out for the capacity h->ks_cols (row
// stride of the slab: capacity + 16 doubles -- no power-of-two stride); the tile list of the variance
// product depends on the number of column blocks actually used and is cached per size.
int ws_ensure_predict(gpimhip_ctx* h, int64_t np, int64_t mc) {
    const int B = h->nbatch;
    if (!(h->ks_rows == np && h->ks_batch == B && mc <= h->ks_cols)) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        const int64_t cap = (h->ks_rows == np && h->ks_batch == B) ? std::max(mc, h->ks_cols) : mc;
        ws_release_predict(h);
        h->ks_rows = np;                 // sizes first: a failed allocation is cleaned up by pointer
        h->ks_cols = cap;
        h->ks_batch = B;
        int rc = GPIMHIP_OK;
        if ((rc = dev_alloc(h, &h->Ks, mat_doubles(h, B * np * (cap + 16)))) || (rc = dev_alloc(h, &h->colpart, B * (np / NB) * cap)) ||
            (rc = dev_alloc(h, &h->mean_tmp, B * cap))) {
            ws_release_predict(h);
            return rc;
        }
    }
    const int nb = (int)(np / NB), nc = (int)(mc / NB);
    for (auto& pl : h->pred_lists)
        if (pl.nc == nc) {
            h->pred_tiles = pl.tiles;
            h->pred_ntiles = pl.n;
            return GPIMHIP_OK;
        }
    // tile list of the variance product W = L^-1 K*: 8x8 patches, longest k-ranges first
    std::vector<TileDesc> tl;
    tl.reserve((size_t)nb * nc);
    for (int ig = (nb - 1) / 8; ig >= 0; --ig)
        for (int jg = 0; jg <= (nc - 1) / 8; ++jg)
            for (int ci = std::min(nb - 1, ig * 8 + 7); ci >= ig * 8; --ci)
                for (int cj = jg * 8; cj < std::min(nc, jg * 8 + 8); ++cj) tl.push_back({ci, cj, 0, ci + 1});
    gpimhip_ctx::PredList pl{nc, nullptr, (int64_t)tl.size()};
    GP_TRY(dev_alloc(h, &pl.tiles, pl.n));
    // (async + stream sync rather than hipMemcpy: the latter serialises against the legacy stream and is
    // illegal while another thread captures a graph)
    HIP_TRY(hipMemcpyAsync(pl.tiles, tl.data(), tl.size() * sizeof(TileDesc), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->pred_lists.push_back(pl);
    h->pred_tiles = pl.tiles;
    h->pred_ntiles = pl.n;
    return GPIMHIP_OK;
}

// ------------------------------------------------------------------------------------------
// launch plans (tile lists) for a matrix of nb x nb blocks
// ------------------------------------------------------------------------------------------
static void lower_patch_order(std::vector<TileDesc>& out, int lo, int hi, int kb0, int kb1) {
    // lower-triangular tile set {(i,j): lo <= j <= i < hi} in 8x8 patches (operand-panel reuse in L2)
    for (int ig = lo / 8; ig <= (hi - 1) / 8; ++ig)
        for (int jg = lo / 8; jg <= ig; ++jg)
            for (int i = std::max(lo, ig * 8); i < std::min(hi, ig * 8 + 8); ++i)
                for (int j = std::max(lo, jg * 8); j < std::min(hi, jg * 8 + 8); ++j)
                    if (j <= i) out.push_back({i, j, kb0, kb1});
}

// rectangular tile set rows [r0,r1) x cols [c0,c1) in 8x8 patches; krange(ci,cj) -> {kb0,kb1}
template <typename F>
static void rect_patch_order(std::vector<TileDesc>& out, int r0, int r1, int c0, int c1, bool rows_desc,
                             bool col_major, F krange) {
    // col_major: walk patch columns in the outer loop (use when the k-range depends on cj), else
    // patch rows (k-range depends on ci).  Either way consecutive patches cost about the same, so
    // dealing them round-robin to the 8 XCDs stays balanced.
    const int nrg = (r1 - r0 + 7) / 8, ncg = (c1 - c0 + 7) / 8;
    const int nouter = col_major ? ncg : nrg, ninner = col_major ? nrg : ncg;
    for (int a = 0; a < nouter; ++a)
        for (int bq = 0; bq < ninner; ++bq) {
            int ig = col_major ? bq : a, jg = col_major ? a : bq;
            if (rows_desc) ig = nrg - 1 - ig;
            for (int i = r0 + ig * 8; i < std::min(r1, r0 + ig * 8 + 8); ++i)
                for (int j = c0 + jg * 8; j < std::min(c1, c0 + jg * 8 + 8); ++j) {
                    int k0, k1;
                    krange(i, j, k0, k1);
                    out.push_back({i, j, k0, k1});
                }
        }
}

struct TriNode { int lo, mid, hi; };
static int tri_build(int lo, int hi, std::vector<std::vector<TriNode>>& levels) {
    if (hi - lo <= 1) return 0;
    const int mid = lo + (hi - lo + 1) / 2;
    const int hl = tri_build(lo, mid, levels), hr = tri_build(mid, hi, levels);
    const int ht = 1 + std::max(hl, hr);
    if ((int)levels.size() < ht) levels.resize(ht);
    levels[ht - 1].push_back({lo, mid, hi});
    return ht;
}

int plan_ensure(gpimhip_ctx* h, int nb) {
    LinalgPlan& P = h->plan;
    if (P.nb == nb) return GPIMHIP_OK;
    if (P.d_tiles) { (void)hipFree(P.d_tiles); P.d_tiles = nullptr; }
    std::vector<TileDesc> tl;
    auto mark = [&](size_t start) { return PlanRange{(int64_t)start, (int32_t)(tl.size() - start)}; };
    // triangular inversion, bottom-up by subtree height
    std::vector<std::vector<TriNode>> levels;
    tri_build(0, nb, levels);
    P.tri_t.clear();
    P.tri_x.clear();
    // Levels with several nodes: node after node, each longest-first, is a saw-tooth of costs and the
    // launch ends on the long tail of whichever workgroup slots drew the long tiles last (model in
    // tools/sched_model.py: 12 % / 40 % over the ideal makespan for 2 / 4 nodes).  One list sorted by
    // k-length instead: equal lengths are one tile column (T) or row (X) of every node, which still
    // share an operand panel.
    auto sort_longest_first = [&](size_t nnodes, size_t start) {
        // (lists of at most one chip-load of tiles stay node-major: sorted, one XCD would draw all the
        // long tiles of the single round -- 298 vs 218 us for the 8-node level at nb = 128)
        if (nnodes < 2 || tl.size() - start <= 512) return;
        std::stable_sort(tl.begin() + start, tl.end(),
                         [](const TileDesc& a, const TileDesc& b) { return a.kb1 - a.kb0 > b.kb1 - b.kb0; });
    };
    for (auto& lv : levels) {
        size_t s = tl.size();
        // T = L21 * X11: k-range [cj, mid) -- longest for the leftmost columns
        for (auto& nd : lv)
            rect_patch_order(tl, nd.mid, nd.hi, nd.lo, nd.mid, false, true,
                             [&](int, int cj, int& k0, int& k1) { k0 = cj; k1 = nd.mid; });
        sort_longest_first(lv.size(), s);
        P.tri_t.push_back(mark(s));
        s = tl.size();
        // X21 = -X22 * T: k-range [mid, ci] -- longest for the bottom rows
        for (auto& nd : lv)
            rect_patch_order(tl, nd.mid, nd.hi, nd.lo, nd.mid, true, false,
                             [&](int ci, int, int& k0, int& k1) { k0 = nd.mid; k1 = ci + 1; });
        sort_longest_first(lv.size(), s);
        P.tri_x.push_back(mark(s));
    }
    // K^-1 = L^-T L^-1 (lower): k-range [ci, nb), longest first.  Patches are 2 rows x 32 columns:
    // tiles of one row have the same k-length and stay in lock-step (their shared operand panel is
    // read once per XCD), neighbouring rows differ by one block only.
    {
        size_t s = tl.size();
        // (measured at N = 16384: 1x64 / 2x32 / 4x16 22.0-22.2 ms, 8x8 22.7, 16x4 23.2; L2-miss traffic
        // 61-67 GB for all of them)
        const int PR = 2, PC = 32;
        for (int ig = 0; ig <= (nb - 1) / PR; ++ig)
            for (int jg = 0; jg * PC <= std::min(nb - 1, ig * PR + PR - 1); ++jg)
                for (int i = ig * PR; i < std::min(nb, ig * PR + PR); ++i)
                    for (int j = jg * PC; j < std::min(nb, jg * PC + PC); ++j)
                        if (j <= i) tl.push_back({i, j, i, nb});
        P.lauum = mark(s);
    }
    P.n_tiles = (int64_t)tl.size();
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, std::max<size_t>(tl.size(), 1) * sizeof(TileDesc)));
    P.d_tiles = (TileDesc*)q;
    HIP_TRY(hipMemcpyAsync(P.d_tiles, tl.data(), tl.size() * sizeof(TileDesc), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    P.nb = nb;
    return GPIMHIP_OK;
}

// ------------------------------------------------------------------------------------------
// blocked drivers
// ------------------------------------------------------------------------------------------
static GemmArgs gemm_args(const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc,
                          double alpha, double beta, const TileDesc* tiles, int n, int64_t rows) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.alpha = alpha; g.beta = beta; g.tiles = tiles; g.ntiles = n;
    g.sA = rows * lda; g.sB = rows * ldb; g.sC = rows * ldc;  // stacked np-row workspace matrices
    return g;
}

// The one helper stream of the library: the stream an iteration is CAPTURED on (mid-size N).  It is created on first
// use and shared by every handle of the device for the life of the process.  The training loop itself runs on the
// caller's stream alone.  (Until round 4 large-N iterations ran their two mat-vecs on a high-priority side stream beside
// the K^-1 product and drove the launch chains from a second one.  The side branch was the CAUSE of the "stream
// population" slow-down recorded in DESIGN.md section 6: its wait on the next iteration's event sits in a hardware queue
// of its own for the whole factorisation, and when that queue is the fifth or later the process has created, every
// dependent launch of the main queue starts 30-45 us late -- profiles/r04_queue_sweep.txt.  On the caller's stream the
// mat-vecs cost < 0.1 ms of a 74 ms iteration.)
struct SideStreams {
    hipStream_t capture = nullptr;
    bool capture_tried = false;
};
static std::mutex g_side_mutex;
static SideStreams g_side[64];
// gpimhip_shutdown destroys the shared stream; a handle that outlives it (a C-ABI host that shuts down and goes on)
// must not use the pointer it cached: every shutdown starts a new generation, and a handle of an older one forgets it
static std::atomic<int> g_side_generation{0};
static void side_refresh(gpimhip_ctx* h) {
    const int g = g_side_generation.load();
    if (h->side_generation == g) return;
    h->side_generation = g;
    h->capture_stream = nullptr;
    h->capture_stream_tried = false;
}
// The capture stream is shared by every handle of a device: two threads fitting at the same time must not
// interleave their Begin..EndCapture sections on it (the second BeginCapture would fail and that fit would
// silently run un-graphed).  Capture itself is short (host-side recording of one iteration).
static std::mutex g_capture_mutex[64];
void capture_lock(gpimhip_ctx* h) { g_capture_mutex[h->device & 63].lock(); }
void capture_unlock(gpimhip_ctx* h) { g_capture_mutex[h->device & 63].unlock(); }

hipStream_t ensure_capture_stream(gpimhip_ctx* h) {
    side_refresh(h);
    if (!h->capture_stream && !h->capture_stream_tried) {
        h->capture_stream_tried = true;
        std::lock_guard<std::mutex> lock(g_side_mutex);
        SideStreams& S = g_side[h->device & 63];
        if (!S.capture_tried) {
            S.capture_tried = true;
            if (hipStreamCreateWithFlags(&S.capture, hipStreamNonBlocking) != hipSuccess) S.capture = nullptr;
        }
        h->capture_stream = S.capture;
    }
    return h->capture_stream;
}

// The blocked Cholesky: the step schedule of cholstep.hip (float matrices: cholstep32.hip).
int launch_potrf(gpimhip_ctx* h, double* A, int64_t np, int64_t ld, int32_t* info) {
    return launch_potrf_steps(h, A, np, ld, info, nullptr);
}

// in-place inverse of the lower-triangular factor: recursive halving, all nodes of one subtree
// height in one launch pair:  T = L21 * X11 ;  X21 = -X22 * T.
int launch_trtri(gpimhip_ctx* h, double* A, double* Tm, int64_t np, int64_t ld) {
    const int nb = (int)(np / NB);
    GP_TRY(plan_ensure(h, nb));
    const LinalgPlan& P = h->plan;
    GP_TRY(launch_diag_inv_copy(h, A, ld, nb));
    for (size_t lv = 0; lv < P.tri_t.size(); ++lv) {
        GemmArgs g1 = gemm_args(A, ld, A, ld, Tm, ld, 1.0, 0.0, P.d_tiles + P.tri_t[lv].off, P.tri_t[lv].n, h->np);
        g1.chunk = deal_chunk(g1.ntiles);
        g1.krev = 1;              // ranges [cj, mid) share their end
        GP_TRY(launch_gemm(h, false, true, EPI_STORE, g1));
        GemmArgs g2 = gemm_args(A, ld, Tm, ld, A, ld, -1.0, 0.0, P.d_tiles + P.tri_x[lv].off, P.tri_x[lv].n, h->np);
        g2.chunk = deal_chunk(g2.ntiles);
        GP_TRY(launch_gemm(h, false, true, EPI_STORE, g2));
    }
    return GPIMHIP_OK;
}

// A <- L^-1 for the SPD matrix in A (Tm: np x np temporary).  Double precision: one pass -- the tile operations of the
// inverse ride in the launches of the factorisation (cholstep.hip: plan_inverse); float matrices: factorisation, then
// the level-by-level inverse above.
int launch_potrf_inv(gpimhip_ctx* h, double* A, double* Tm, int64_t np, int64_t ld, int32_t* info, int rag) {
    if (!h->fp32) return launch_potrf_steps(h, A, np, ld, info, Tm, rag);
    { StageTimer t(h, 0); GP_TRY(launch_potrf_steps(h, A, np, ld, info, nullptr)); }
    StageTimer t(h, 1);
    return launch_trtri(h, A, Tm, np, ld);
}

// B(lower) = A^T A for lower-triangular A (= L^-1): K^-1.
int launch_lauum(gpimhip_ctx* h, const double* A, double* B, int64_t np, int64_t ld, int rag) {
    const int nb = (int)(np / NB);
    GP_TRY(plan_ensure(h, nb));
    const LinalgPlan& P = h->plan;
    GemmArgs g = gemm_args(A, ld, A, ld, B, ld, 1.0, 0.0, P.d_tiles + P.lauum.off, P.lauum.n, h->np);
    g.chunk = deal_chunk(g.ntiles);
    g.krev = 1;                   // ranges [ci, nb) share their end
    g.rag = h->fp32 ? 0 : rag;
    return launch_gemm(h, true, true, EPI_STORE, g);
}

// N <= 128: the fused single-workgroup trainer (smalln.hip) replaces the blocked path
static bool use_small_path(int64_t N) { return N <= NB && !getenv("GPIMHIP_NO_SMALLN"); }

int check_model(const gpimhip_model_t* m) {
    if (!m || m->dim < 1 || m->dim > GPIMHIP_MAX_DIM || (m->n_ls != 1 && m->n_ls != m->dim) ||
        m->kernel < 0 || m->kernel > GPIMHIP_KERNEL_RQ) {
        gpim_set_error("invalid gpimhip_model_t");
        return GPIMHIP_E_BADARG;
    }
    return GPIMHIP_OK;
}

// Everything needed at the current u: theta, K, L, L^-1, z, alpha.  (Shared by fit and predict.)
// x_bs: per-problem stride of X in elements (0 when all problems of a batch share one X).
// z = L^-1 y, alpha = L^-T z (two HBM-bound passes over L^-1).
// Single-precision handles refine alpha against the covariance regenerated in double (launch_kres): the fp32
// factor and its explicit fp32 inverse leave an error of eps32 * cond(K) in alpha, every pass
// alpha += K32^-1 (y - K alpha) multiplies it by that factor again (one pass by default; measured at cond = 1e5,
// N = 2300, RBF: posterior-mean error 6e-4 -> 1.2e-4, a float32 LAPACK run: 1.3e-3; loss 0.95 -> 0.06 of 4129;
// a second pass changes nothing -- what is left comes from log det and K^-1 themselves).  The loss then takes its
// quadratic term as y^T alpha (launch_finalize).
static int refine_passes() {
    static const int v = 1;
    return v;
}
// defer_alpha (double precision only): alpha stays as the row-chunk partial sums in h->gemv_part -- the gradient
// contraction adds up the entries it needs itself (launch_grad_reduce_fin), one launch less per iteration; h->alpha is
// then NOT valid.
static bool alpha_deferrable(const gpimhip_ctx* h) { return !h->fp32 && !h->refl.mask && h->np <= 8192 && !getenv("GPIMHIP_NO_FUSED_FINALIZE"); }
static int solve_vectors(gpimhip_ctx* h, const gpimhip_model_t* m, const double* X, int64_t x_bs, int64_t N,
                         bool defer_alpha = false) {
    const int64_t np = h->np, ld = h->ld;
    GP_TRY(launch_trmv_lower(h, h->A, ld, np, h->ypad, h->z));
    GP_TRY(launch_gemv_t_tri(h, h->A, ld, np, h->z, h->gemv_part, defer_alpha && !h->fp32 ? nullptr : h->alpha));
    if (!h->fp32) return GPIMHIP_OK;
    const int B = h->nbatch, nb = (int)(np / NB);
    const int S = std::max(1, std::min(8, 512 / std::max(1, nb * B)));        // >= ~512 workgroups per launch
    double *res = h->refine, *delta = res + (int64_t)B * np, *scratch = delta + (int64_t)B * np;
    for (int pass = 0; pass < refine_passes(); ++pass) {
        GP_TRY(launch_kres(h, m, X, x_bs, N, scratch, S, res));
        GP_TRY(launch_trmv_lower(h, h->A, ld, np, res, h->z));
        GP_TRY(launch_gemv_t_tri(h, h->A, ld, np, h->z, h->gemv_part, delta));
        GP_TRY(launch_axpy(h, h->alpha, delta, (int64_t)B * np));
    }
    return GPIMHIP_OK;
}

// K(u) -> L -> L^-1 (in h->A), then z = L^-1 y and alpha = L^-T z
// theta_current: h->theta already holds theta(u) -- the previous iteration's finalize step wrote it (fit_impl)
static int factor_at_u(gpimhip_ctx* h, const gpimhip_model_t* m, const double* X, int64_t x_bs, int64_t N,
                       const double* u, bool theta_current = false, bool defer_alpha = false) {
    const int64_t np = h->np, ld = h->ld;
    if (!theta_current) GP_TRY(launch_theta(h, m, u));
    if (h->refl.mask)      // symmetry-reduced model: problem b of the batch is the block of sign pattern b (engine.hip)
        GP_TRY(launch_kmat_refl(h, m, X, N, nullptr, N, h->theta, h->A, ld, np, np, 1, x_bs, x_bs, np * ld, 1.0));
    else
        GP_TRY(launch_kmat(h, m, X, N, nullptr, N, h->theta, 0.0, 1, h->A, ld, np, np, 1, 1, x_bs, x_bs, np * ld));
    GP_TRY(launch_potrf_inv(h, h->A, h->Tm, np, ld, h->info, rag_of(N, np)));
    return solve_vectors(h, m, X, x_bs, N, defer_alpha);
}

// theta_carried: the training loop launched theta(u) once before its first iteration; every finalize step then leaves the
// theta of the stepped parameters in h->theta (no theta launch inside the loop)
struct IterTable { int32_t* iter; const double* bc; int T; double* hist_base; double* loss_base; bool theta_carried; };

static int loss_grad_at_u(gpimhip_ctx* h, const gpimhip_model_t* m, const double* X, int64_t x_bs, int64_t N,
                          double* u, int do_adam, AdamStep st, double* loss_out, double* grad_out,
                          double* hist_row, const IterTable* tab = nullptr) {
    const int64_t np = h->np;
    // One launch for the gradient contraction and the finalize step (round 6; GPIMHIP_NO_FUSED_FINALIZE: the two launches
    // of rounds 1-5 -- the same reductions in the same order, the same bits)
    // ... up to np = 8192; beyond, the finalize step is a launch of its own again (the tail's device-scope loads of 8256 tiles'
    // sums take longer than a launch boundary costs: 530 us against 400 + 20 at N = 16384) -- it still leaves the next theta
    const bool modern = !h->refl.mask && !getenv("GPIMHIP_NO_FUSED_FINALIZE");
    const bool fused = modern && np <= 8192;
    const bool carried = modern && tab && tab->theta_carried;
    const bool defer = fused && alpha_deferrable(h);
    GP_TRY(factor_at_u(h, m, X, x_bs, N, u, carried, defer));
    { StageTimer t(h, 2); GP_TRY(launch_lauum(h, h->A, h->B, np, h->ld, rag_of(N, np))); }
    if (h->refl.mask) {
        GP_TRY(launch_grad_reduce_refl(h, m, h->B, h->ld, X, N, (int)(np / NB), h->alpha, x_bs));
        if (tab)
            return launch_finalize_coupled(h, m, N, np, u, do_adam, st, nullptr, nullptr, nullptr, tab->iter, tab->bc, tab->T,
                                           tab->hist_base, tab->loss_base);
        return launch_finalize_coupled(h, m, N, np, u, do_adam, st, loss_out, grad_out, hist_row, nullptr, nullptr, 0, nullptr,
                                       nullptr);
    }
    if (fused) {
        const double* ap = defer ? h->gemv_part : nullptr;
        if (tab)
            return launch_grad_reduce_fin(h, m, h->B, h->ld, X, N, (int)(np / NB), h->alpha, x_bs, ap, u, do_adam, st, nullptr,
                                          nullptr, nullptr, tab->iter, tab->bc, tab->T, tab->hist_base, tab->loss_base,
                                          carried ? 1 : 0);
        return launch_grad_reduce_fin(h, m, h->B, h->ld, X, N, (int)(np / NB), h->alpha, x_bs, ap, u, do_adam, st, loss_out,
                                      grad_out, hist_row, nullptr, nullptr, 0, nullptr, nullptr, 0);
    }
    GP_TRY(launch_grad_reduce(h, m, h->B, h->ld, X, N, (int)(np / NB), h->alpha, x_bs));
    if (tab)
        GP_TRY(launch_finalize(h, m, N, np, u, do_adam, st, nullptr, nullptr, nullptr, tab->iter, tab->bc, tab->T,
                               tab->hist_base, tab->loss_base, carried ? 1 : 0));
    else
        GP_TRY(launch_finalize(h, m, N, np, u, do_adam, st, loss_out, grad_out, hist_row, nullptr, nullptr, 0,
                               nullptr, nullptr));
    return GPIMHIP_OK;
}

// Fill the device table of Adam bias corrections (same libm pow() values for every path).
int upload_bc_table(gpimhip_ctx* h, double lr, int T) {
    if (h->bc_cap < 2 * (int64_t)T) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        dev_free(h, &h->bc, h->bc_cap);
        h->bc_cap = 0;
        GP_TRY(dev_alloc(h, &h->bc, 2 * (int64_t)T));
        h->bc_cap = 2 * (int64_t)T;
    }
    h->bc_host.resize(2 * (size_t)T);
    for (int t = 1; t <= T; ++t) {
        h->bc_host[t - 1] = lr / (1.0 - pow(0.9, (double)t));
        h->bc_host[T + t - 1] = sqrt(1.0 - pow(0.999, (double)t));
    }
    HIP_TRY(hipMemcpyAsync(h->bc, h->bc_host.data(), 2 * (size_t)T * sizeof(double), hipMemcpyHostToDevice,
                           h->stream));
    return GPIMHIP_OK;
}

// info[0]: 0 or 1 + first failing column; info[1]: the number of Adam iterations that had completed
// when a training loop first met a non-PD matrix (min over the problems of a batch; only meaningful
// when info[0] != 0 -- fit_impl presets it to a large value).
int finish_and_check(gpimhip_ctx* h) {
    int32_t info[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(info, h->info, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (info[0] != 0) {
        h->fit_completed = std::max(0, std::min(h->fit_completed, info[1]));
        gpim_set_error("cholesky: the input is not positive-definite (leading minor of order " +
                       std::to_string(info[0]) + ")");
        return GPIMHIP_E_NOT_PD;
    }
    return GPIMHIP_OK;
}

// Bounded run-ahead for training loops: every `period` iterations the status word is copied to pinned
// host memory behind an event; before enqueueing more work the host looks at the copy made two periods
// ago (already complete unless the queue is that short), so a failed factorisation stops the loop
// within 2 * period iterations while the device queue never drains.  The device side freezes u, the
// Adam state and the history at the failing iteration on its own (finalize_kernel), so where exactly
// the host stops enqueueing does not change any result.
struct RunAhead {
    gpimhip_ctx* h;
    int period;
    bool armed[2] = {false, false};
    RunAhead(gpimhip_ctx* h_, int64_t np) : h(h_), period(np >= 4096 ? 4 : 32) {
        if (!h->pinned_info) {
            void* q = nullptr;
            if (hipHostMalloc(&q, 2 * sizeof(int32_t), hipHostMallocDefault) == hipSuccess) h->pinned_info = (int32_t*)q;
            for (auto& e : h->ra_ev)
                if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) e = nullptr;
        }
    }
    // call before enqueueing iteration t; returns true when the loop should stop
    bool stop(int t) {
        if (!h->pinned_info || !h->ra_ev[0] || !h->ra_ev[1] || t == 0 || t % period) return false;
        const int slot = (t / period) & 1;
        if (armed[slot]) {
            (void)hipEventSynchronize(h->ra_ev[slot]);
            if (h->pinned_info[slot] != 0) return true;
        }
        (void)hipMemcpyAsync(h->pinned_info + slot, h->info, sizeof(int32_t), hipMemcpyDeviceToHost, h->stream);
        (void)hipEventRecord(h->ra_ev[slot], h->stream);
        armed[slot] = true;
        return false;
    }
};

int vfe_finish_and_check(gpimhip_ctx* h) { return finish_and_check(h); }
void vfe_release(gpimhip_ctx* h);
void kron_release(gpimhip_ctx* h);
static void dist_plan_release(gpimhip_ctx* h);
// the distributed entry points address the workspace (diagonal-block inverses, batch strides) through the plan's block
// count: a handle whose workspace was re-sized after gpimhip_dist_setup must not be used with the stale plan
static bool dist_plan_ok(const gpimhip_ctx* h) { return h && h->np && h->dplan.nb && h->np == (int64_t)h->dplan.nb * NB; }

// ------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------
extern "C" {

const char* gpimhip_last_error(void) { return g_err.c_str(); }

// Destroys the process-wide capture stream (ensure_capture_stream).  Called by the Python binding at interpreter exit,
// while the HIP runtime is still up.
int gpimhip_shutdown(void) {
    std::lock_guard<std::mutex> lock(g_side_mutex);
    for (auto& S : g_side) {
        if (S.capture) (void)hipStreamDestroy(S.capture);
        S = SideStreams();
    }
    g_side_generation.fetch_add(1);
    return GPIMHIP_OK;
}
int gpimhip_version(void) { return 100; }

int gpimhip_create(gpimhip_handle* out, int device, void* hip_stream) {
    if (!out) return GPIMHIP_E_BADARG;
    HIP_TRY(hipSetDevice(device));
    gpimhip_ctx* h = new gpimhip_ctx();
    h->device = device;
    h->stream = (hipStream_t)hip_stream;     // NULL = the device's default (null) stream
    int rc = GPIMHIP_OK;
    if ((rc = dev_alloc(h, &h->theta1, 1)) || (rc = dev_alloc(h, &h->info, 4))) {
        delete h;
        return rc;
    }
    (void)hipMemsetAsync(h->info, 0, 4 * sizeof(int32_t), h->stream);
    *out = h;
    return GPIMHIP_OK;
}

int gpimhip_set_precision(gpimhip_handle h, int32_t bits) {
    if (!h || (bits != 32 && bits != 64)) return GPIMHIP_E_BADARG;
    const int want = bits == 32 ? 1 : 0;
    if (want == h->fp32) return GPIMHIP_OK;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    ws_release_matrix(h);               // sized by the element type
    ws_release_predict(h);
    h->fp32 = want;
    return GPIMHIP_OK;
}

int gpimhip_destroy(gpimhip_handle h) {
    if (!h) return GPIMHIP_OK;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    vfe_release(h);
    kron_release(h);
    ws_release_matrix(h);
    ws_release_predict(h);
    dev_free(h, &h->keys, h->keys_cap);
    if (h->sel_scratch) dev_free(h, &h->sel_scratch, (int64_t)sel_scratch_bytes());
    dev_free(h, &h->acq_tmp, h->acq_tmp_cap);
    dev_free(h, &h->refine, h->refine_cap);
    dev_free(h, &h->bc, h->bc_cap);
    dev_free(h, &h->theta1, 1);
    dev_free(h, &h->info, 4);
    if (h->plan.d_tiles) (void)hipFree(h->plan.d_tiles);
    step_plan_release(h);
    dist_plan_release(h);
    for (auto e : h->ra_ev)
        if (e) (void)hipEventDestroy(e);
    if (h->pinned_info) (void)hipHostFree(h->pinned_info);
    // (the capture stream belongs to the process, not to the handle)
    delete h;
    return GPIMHIP_OK;
}

int64_t gpimhip_workspace_bytes(gpimhip_handle h) { return h ? h->bytes : 0; }

int gpimhip_timing_enable(gpimhip_handle h, int enable) {
    if (!h) return GPIMHIP_E_BADARG;
    h->timing = enable != 0;
    return GPIMHIP_OK;
}

int gpimhip_timing_read(gpimhip_handle h, int stage, double* total_ms, int64_t* count) {
    if (!h || stage < 0 || stage > 3 || !total_ms || !count) return GPIMHIP_E_BADARG;
    HIP_TRY(hipStreamSynchronize(h->stream));
    double tot = 0.0;
    for (auto& pr : h->ev[stage]) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, pr.first, pr.second);
        tot += ms;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    *total_ms = tot;
    *count = (int64_t)h->ev[stage].size();
    h->ev[stage].clear();
    return GPIMHIP_OK;
}

int gpimhip_fit_completed(gpimhip_handle h) { return h ? h->fit_completed : 0; }

int gpimhip_sync(gpimhip_handle h) {
    if (!h) return GPIMHIP_E_BADARG;
    HIP_TRY(hipStreamSynchronize(h->stream));
    return GPIMHIP_OK;
}

int gpimhip_kmat(gpimhip_handle h, const gpimhip_model_t* m, const double* X, int64_t N, const double* Z,
                 int64_t M, const double* theta, double diag_add, double* out, int64_t ld) {
    FP64_ONLY(h);
    if (!h || !X || !theta || !out || N < 1) return GPIMHIP_E_BADARG;
    GP_TRY(check_model(m));
    HIP_TRY(hipSetDevice(h->device));
    const bool sym = (Z == nullptr);
    const int64_t Mv = sym ? N : M;
    if (Mv < 1 || ld < Mv) return GPIMHIP_E_BADARG;
    h->nbatch = 1;
    GP_TRY(launch_theta_raw(h, m, theta));
    // tiles are 128x128: build into a padded scratch and copy out the valid part
    const int64_t rp = pad_to(N, NB), cp = pad_to(Mv, NB);
    GP_TRY(ws_ensure_predict(h, rp, cp));
    const int64_t kld = h->ks_cols + 16;
    GP_TRY(launch_kmat(h, m, X, N, sym ? nullptr : Z, Mv, h->theta1, diag_add, 0, h->Ks, kld, rp, cp, sym ? 1 : 0, 0, 0,
                       0, 0));
    HIP_TRY(hipMemcpy2DAsync(out, (size_t)ld * sizeof(double), h->Ks, (size_t)kld * sizeof(double),
                             (size_t)Mv * sizeof(double), (size_t)N, hipMemcpyDeviceToDevice, h->stream));
    return GPIMHIP_OK;
}

int gpimhip_potrf(gpimhip_handle h, double* A, int64_t n, int64_t ld, int32_t* info) {
    FP64_ONLY(h);
    if (!h || !A || n < 1 || ld < n || !info) return GPIMHIP_E_BADARG;
    HIP_TRY(hipSetDevice(h->device));
    h->nbatch = 1;
    GP_TRY(ws_ensure(h, n));
    const int64_t np = h->np;
    HIP_TRY(hipMemsetAsync(info, 0, sizeof(int32_t), h->stream));
    GP_TRY(launch_pad_matrix_in(h, A, n, ld, h->A, np));
    GP_TRY(launch_potrf(h, h->A, np, np, info));
    GP_TRY(launch_pad_matrix_out_lower(h, h->A, np, A, n, ld));
    return GPIMHIP_OK;
}

static int fit_impl(gpimhip_ctx* h, const gpimhip_model_t* m, const double* X, int64_t x_bs, const double* y,
                    int64_t N, int B, double* u, double lr, int32_t T, double* hist_out, double* loss_out) {
    HIP_TRY(hipSetDevice(h->device));
    h->nbatch = B;
    HIP_TRY(hipMemsetAsync(h->info, 0, sizeof(int32_t), h->stream));
    HIP_TRY(hipMemsetAsync(h->info + 1, 0x7f, sizeof(int32_t), h->stream));   // "completed" = huge until a failure
    h->fit_completed = T;
    if (T == 0) return GPIMHIP_OK;
    GP_TRY(upload_bc_table(h, lr, T));
    if (use_small_path(N) && !h->refl.mask) {
        // fused single-launch trainer, one workgroup per problem
        GP_TRY(launch_fit_small(h, m, X, x_bs, y, (int)N, u, h->bc, h->bc + T, T, hist_out, loss_out, nullptr));
        return finish_and_check(h);
    }
    GP_TRY(ws_ensure_padded(h, N));
    HIP_TRY(hipMemsetAsync(h->adam_m, 0, (size_t)B * MAXP * sizeof(double), h->stream));
    HIP_TRY(hipMemsetAsync(h->adam_v, 0, (size_t)B * MAXP * sizeof(double), h->stream));
    HIP_TRY(hipMemsetAsync(h->iter, 0, (size_t)B * sizeof(int32_t), h->stream));     // iteration counters
    HIP_TRY(hipMemsetAsync(h->fin_counter, 0, (size_t)B * sizeof(uint32_t), h->stream));   // (zero already, unless a launch was cut short)
    GP_TRY(launch_pad_copy(h, y, N, h->ypad, h->np));
    // theta(u) once; after that every iteration's finalize step leaves the next theta behind (not for the symmetry-reduced
    // model, whose coupled finalize step is a launch of its own)
    const bool carry = !h->refl.mask && !getenv("GPIMHIP_NO_FUSED_FINALIZE");
    if (carry) GP_TRY(launch_theta(h, m, u));
    IterTable tab{h->iter, h->bc, T, hist_out, loss_out, carry};
    AdamStep st;
    st.beta1 = 0.9; st.beta2 = 0.999; st.eps = 1e-8; st.lr_over_bc1 = 0.0; st.bc2_sqrt = 1.0;
    // Every iteration enqueues the same launches (the iteration index lives on the device), so one
    // iteration is captured into a hipGraph and replayed: ~10 us of host work per iteration instead
    // of one launch call per kernel.  Not used while stage timing is on or for very short fits.
    // Large N: the launch cost no longer matters and the iteration is enqueued launch by launch on the caller's stream
    // (ONE in-order stream: factor_at_u, loss_grad_at_u).
    const int npanel = (int)((h->np / NB + OUTER_W - 1) / OUTER_W);
    const bool large = npanel >= EAGER_MIN_PANELS;
    const bool use_graph = T >= 8 && !h->timing && !getenv("GPIMHIP_NO_GRAPH") && !large && ensure_capture_stream(h);
    if (use_graph) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        hipStream_t main_s = h->stream;
        h->stream = h->capture_stream;
        capture_lock(h);
        hipError_t e = hipStreamBeginCapture(h->capture_stream, hipStreamCaptureModeRelaxed);
        int rc = GPIMHIP_OK;
        if (e == hipSuccess) {
            h->capturing = true;
            rc = loss_grad_at_u(h, m, X, x_bs, N, u, 1, st, nullptr, nullptr, nullptr, &tab);
            h->capturing = false;
            e = hipStreamEndCapture(h->capture_stream, &graph);
        }
        capture_unlock(h);
        h->stream = main_s;
        if (rc != GPIMHIP_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        const bool inst_ok = e == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess;
        if (inst_ok) {
            RunAhead ra(h, h->np);
            hipError_t le = hipSuccess;
            for (int t = 0; t < T && le == hipSuccess && !ra.stop(t); ++t) le = hipGraphLaunch(exec, main_s);
            rc = finish_and_check(h);
            (void)hipGraphExecDestroy(exec);
            (void)hipGraphDestroy(graph);
            HIP_TRY(le);
            return rc;
        }
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();                        // capture unavailable: plain launches below
    }
    RunAhead ra(h, h->np);
    for (int t = 0; t < T && !ra.stop(t); ++t)
        GP_TRY(loss_grad_at_u(h, m, X, x_bs, N, u, 1, st, nullptr, nullptr, nullptr, &tab));
    return finish_and_check(h);
}


// posterior mean / variance at M test points from the factorised model in the workspace (theta, L^-1, alpha)
static int predict_cols(gpimhip_ctx* h, const gpimhip_model_t* m, const double* X, int64_t x_bs, int64_t N, int B,
                        const double* Xs, int64_t M, double* mean_out, double* var_out) {
    const int64_t np = h->np;
    const int nb = (int)(np / NB);
    // few observations: one fused launch, K* stays in LDS (predict.hip)
    if (!h->fp32 && fused_predict_fits(np) && !h->refl.mask)
        return launch_predict_fused(h, m, X, x_bs, N, Xs, M, mean_out, var_out, nullptr, nullptr, -1, 0.0, nullptr, 0.0,
                                    nullptr);
    // chunk the test points so that the K* slabs of all problems together stay <= ~1 GiB
    int64_t mc = pad_to(M, NB);
    const int64_t cap = std::max<int64_t>(NB, ((int64_t)1 << 27) / np / B / NB * NB);
    mc = std::min(mc, cap);
    GP_TRY(ws_ensure_predict(h, np, mc));
    const int64_t mcap = h->ks_cols, kld = mcap + 16;       // buffer strides follow the capacity
    for (int64_t m0 = 0; m0 < M; m0 += mc) {
        const int64_t cnt = std::min(mc, M - m0);
        const int64_t cpad = pad_to(cnt, NB);
        // the test grid Xs is shared by all problems of the batch (z stride 0)
        if (h->refl.mask)      // V_s^T k(X, x*): the block's rows against the test point and its mirror images, |G|^-1/2 each
            GP_TRY(launch_kmat_refl(h, m, X, N, Xs + m0 * m->dim, cnt, h->theta, h->Ks, kld, np, cpad, 0, x_bs, 0, np * kld,
                                    1.0 / sqrt((double)(h->refl.nblocks_total > 0 ? h->refl.nblocks_total : B))));
        else
            GP_TRY(launch_kmat(h, m, X, N, Xs + m0 * m->dim, cnt, h->theta, 0.0, 0, h->Ks, kld, np, cpad, 0, 0, x_bs, 0,
                               np * kld));
        GP_TRY(launch_gemv_t(h, h->Ks, kld, np, cpad, h->alpha, h->mean_tmp, 0, np * kld, np, mcap));
        if (!h->refl.mask) GP_TRY(launch_copy_slice(h, h->mean_tmp, mean_out + m0, cnt, mcap, M));
        // reflection blocks: the variance may be wanted for the first var_count test points only (it is invariant under
        // the reflections: a caller predicting on the training grid asks for it on the fundamental domain)
        const int64_t nvar = (h->refl.mask && h->refl.var_count > 0) ? std::max<int64_t>(0, std::min(cnt, h->refl.var_count - m0)) : cnt;
        if (nvar == 0) {
            GP_TRY(launch_predict_coupled(h, mcap, nb, m0, cnt, 0, mcap, mean_out, var_out));
            continue;
        }
        GemmArgs g = gemm_args(h->A, h->ld, h->Ks, kld, nullptr, 0, 1.0, 0.0, h->pred_tiles, 0, h->np);
        if (nvar < cnt) g.cj_max = (int)((nvar + NB - 1) / NB);
        g.sB = np * kld;
        g.colpart = h->colpart;
        g.ld_colpart = mcap;
        g.sColpart = (int64_t)nb * mcap;
        // a ragged last chunk still sweeps all column tiles of the slab (stale columns are ignored)
        g.ntiles = (int)h->pred_ntiles;
        g.chunk = deal_chunk(g.ntiles);
        g.rag = h->fp32 ? 0 : rag_of(N, np);
        { StageTimer t(h, 3); GP_TRY(launch_gemm(h, false, true, EPI_COLSUMSQ, g)); }
        if (h->refl.mask) GP_TRY(launch_predict_coupled(h, mcap, nb, m0, cnt, nvar, mcap, mean_out, var_out));
        else GP_TRY(launch_predict_var(h, mcap, nb, m0, cnt, var_out, M));
    }
    return GPIMHIP_OK;
}

static int predict_impl(gpimhip_ctx* h, const gpimhip_model_t* m, const double* X, int64_t x_bs, const double* y,
                        int64_t N, int B, const double* u, const double* Xs, int64_t M, double* mean_out,
                        double* var_out) {
    HIP_TRY(hipSetDevice(h->device));
    h->nbatch = B;
    GP_TRY(ws_ensure_padded(h, N));
    const int64_t np = h->np;
    HIP_TRY(hipMemsetAsync(h->info, 0, sizeof(int32_t), h->stream));
    GP_TRY(launch_pad_copy(h, y, N, h->ypad, np));
    GP_TRY(factor_at_u(h, m, X, x_bs, N, u));
    GP_TRY(predict_cols(h, m, X, x_bs, N, B, Xs, M, mean_out, var_out));
    return finish_and_check(h);
}

int gpimhip_nll_grad(gpimhip_handle h, const gpimhip_model_t* m, const double* X, const double* y, int64_t N,
                     const double* u, double* loss_out, double* grad_out) {
    if (!h || !X || !y || !u || N < 1) return GPIMHIP_E_BADARG;
    GP_TRY(check_model(m));
    HIP_TRY(hipSetDevice(h->device));
    h->nbatch = 1;
    HIP_TRY(hipMemsetAsync(h->info, 0, sizeof(int32_t), h->stream));
    if (use_small_path(N)) {
        GP_TRY(launch_fit_small(h, m, X, 0, y, (int)N, const_cast<double*>(u), nullptr, nullptr, 0, nullptr,
                                loss_out, grad_out));
        return finish_and_check(h);
    }
    GP_TRY(ws_ensure_padded(h, N));
    GP_TRY(launch_pad_copy(h, y, N, h->ypad, h->np));
    AdamStep st;
    memset(&st, 0, sizeof(st));
    GP_TRY(loss_grad_at_u(h, m, X, 0, N, const_cast<double*>(u), 0, st, loss_out, grad_out, nullptr));
    return finish_and_check(h);
}

int gpimhip_fit_exact(gpimhip_handle h, const gpimhip_model_t* m, const double* X, const double* y, int64_t N,
                      double* u_inout, double lr, int32_t T, double* hist_out, double* loss_out) {
    if (!h || !X || !y || !u_inout || N < 1 || T < 0) return GPIMHIP_E_BADARG;
    GP_TRY(check_model(m));
    return fit_impl(h, m, X, 0, y, N, 1, u_inout, lr, T, hist_out, loss_out);
}

int gpimhip_predict_exact(gpimhip_handle h, const gpimhip_model_t* m, const double* X, const double* y, int64_t N,
                          const double* u, const double* Xs, int64_t M, double* mean_out, double* var_out) {
    if (!h || !X || !y || !u || !Xs || M < 1 || N < 1 || !mean_out || !var_out) return GPIMHIP_E_BADARG;
    GP_TRY(check_model(m));
    return predict_impl(h, m, X, 0, y, N, 1, u, Xs, M, mean_out, var_out);
}

int gpimhip_fit_exact_batched(gpimhip_handle h, const gpimhip_model_t* m, const double* X, int64_t x_stride,
                              const double* y, int64_t N, int32_t B, double* u_inout, double lr, int32_t T,
                              double* hist_out, double* loss_out) {
    if (!h || !X || !y || !u_inout || N < 1 || T < 0 || B < 1 || B > 65535 ||
        (x_stride != 0 && x_stride < N * (m ? m->dim : 1)))
        return GPIMHIP_E_BADARG;
    GP_TRY(check_model(m));
    return fit_impl(h, m, X, x_stride, y, N, B, u_inout, lr, T, hist_out, loss_out);
}

int gpimhip_predict_exact_batched(gpimhip_handle h, const gpimhip_model_t* m, const double* X, int64_t x_stride,
                                  const double* y, int64_t N, int32_t B, const double* u, const double* Xs,
                                  int64_t M, double* mean_out, double* var_out) {
    if (!h || !X || !y || !u || !Xs || M < 1 || N < 1 || !mean_out || !var_out || B < 1 || B > 65535)
        return GPIMHIP_E_BADARG;
    GP_TRY(check_model(m));
    return predict_impl(h, m, X, x_stride, y, N, B, u, Xs, M, mean_out, var_out);
}

int gpimhip_acq(gpimhip_handle h, int32_t kind, const double* mean, const double* sd, int64_t M, double p0,
                double p1, const double* mask, double* acq_out) {
    if (!h || !mean || !sd || !acq_out || M < 1 || kind < 0 || kind > GPIMHIP_ACQ_POI) return GPIMHIP_E_BADARG;
    HIP_TRY(hipSetDevice(h->device));
    return launch_acq(h, kind, mean, sd, M, p0, p1, mask, acq_out);
}

static int sel_scratch_ensure(gpimhip_ctx* h);

int gpimhip_acquire_exact(gpimhip_handle h, const gpimhip_model_t* m, const double* X, const double* y, int64_t N,
                          const double* u, const double* Xs, int64_t M, const double* Xobs, int64_t Mobs, int32_t kind,
                          double p0, double p1, const double* mask, double* mean_out, double* sd_out, double* acq_out) {
    if (!h || !X || !y || !u || !Xs || M < 1 || N < 1 || !mean_out || !sd_out || !acq_out || kind < 0 ||
        kind > GPIMHIP_ACQ_POI || (kind != GPIMHIP_ACQ_CB && (!Xobs || Mobs < 1)))
        return GPIMHIP_E_BADARG;
    GP_TRY(check_model(m));
    HIP_TRY(hipSetDevice(h->device));
    h->nbatch = 1;
    GP_TRY(ws_ensure_padded(h, N));
    const int64_t np = h->np;
    HIP_TRY(hipMemsetAsync(h->info, 0, sizeof(int32_t), h->stream));
    GP_TRY(launch_pad_copy(h, y, N, h->ypad, np));
    GP_TRY(factor_at_u(h, m, X, 0, N, u));
    const bool fused = !h->fp32 && fused_predict_fits(np);
    const double* inc = nullptr;
    if (kind != GPIMHIP_ACQ_CB) {
        // incumbent = nanmax of the posterior at the observed rows (EI: means; POI: means and sds), on device
        if (h->acq_tmp_cap < 2 * Mobs + 2) {
            HIP_TRY(hipStreamSynchronize(h->stream));
            dev_free(h, &h->acq_tmp, h->acq_tmp_cap);

            h->acq_tmp_cap = 0;
            GP_TRY(dev_alloc(h, &h->acq_tmp, 2 * Mobs + 2));
            h->acq_tmp_cap = 2 * Mobs + 2;
        }
        double *mo = h->acq_tmp, *so = h->acq_tmp + Mobs, *best = h->acq_tmp + 2 * Mobs;
        if (fused) {
            GP_TRY(launch_predict_fused(h, m, X, 0, N, Xobs, Mobs, mo, nullptr, so, nullptr, -1, 0.0, nullptr, 0.0, nullptr));
        } else {
            GP_TRY(predict_cols(h, m, X, 0, N, 1, Xobs, Mobs, mo, so));
            GP_TRY(launch_acq_from_var(h, kind, mo, so, Mobs, 0.0, nullptr, 0.0, nullptr, so, nullptr));
        }
        const int64_t nmax = (kind == GPIMHIP_ACQ_EI) ? Mobs : 2 * Mobs;
        if (nmax <= 4096) {
            GP_TRY(launch_nanmax(h, mo, nmax, best));
        } else {
            GP_TRY(sel_scratch_ensure(h));
            GP_TRY(launch_nanmax_two_stage(h, mo, nmax, best));
        }
        inc = best;
    }
    if (fused) {
        GP_TRY(launch_predict_fused(h, m, X, 0, N, Xs, M, mean_out, nullptr, sd_out, acq_out, kind, p0, inc, p1, mask));
    } else {
        // sd_out doubles as the variance buffer of the slab path
        GP_TRY(predict_cols(h, m, X, 0, N, 1, Xs, M, mean_out, sd_out));
        GP_TRY(launch_acq_from_var(h, kind, mean_out, sd_out, M, p0, inc, p1, mask, sd_out, acq_out));
    }
    return finish_and_check(h);
}

static int sel_scratch_ensure(gpimhip_ctx* h) {
    if (h->sel_scratch) return GPIMHIP_OK;
    return dev_alloc(h, &h->sel_scratch, (int64_t)sel_scratch_bytes());
}

int gpimhip_nanmax(gpimhip_handle h, const double* x, int64_t n, double* out) {
    if (!h || !x || !out || n < 1) return GPIMHIP_E_BADARG;
    HIP_TRY(hipSetDevice(h->device));
    if (n <= 4096) return launch_nanmax(h, x, n, out);
    GP_TRY(sel_scratch_ensure(h));
    return launch_nanmax_two_stage(h, x, n, out);
}

int gpimhip_topk(gpimhip_handle h, const double* acq, int64_t M, int32_t k, int32_t keep_nan, double* vals_out,
                 int64_t* idx_out, int64_t* count_out) {
    if (!h || !acq || !vals_out || !idx_out || !count_out || M < 1 || k < 1) return GPIMHIP_E_BADARG;
    HIP_TRY(hipSetDevice(h->device));
    if (h->keys_cap < M) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        dev_free(h, &h->keys, h->keys_cap);
        h->keys_cap = 0;
        GP_TRY(dev_alloc(h, &h->keys, M));
        h->keys_cap = M;
    }
    // large grids: multi-block radix selection (15 launches, O(M) each); small ones: k arg-max passes of one
    // workgroup (2 launches)
    if (M > 2048 && k <= 1024 && M < ((int64_t)1 << 32) && !getenv("GPIMHIP_NO_RADIX_TOPK")) {
        GP_TRY(sel_scratch_ensure(h));
        return launch_topk_radix(h, acq, M, k, keep_nan, vals_out, idx_out, count_out);
    }
    return launch_topk(h, acq, M, k, keep_nan, vals_out, idx_out, count_out);
}

// ---- distributed (block-column-cyclic, 1 x P) exact GP: building blocks (see include/gpimhip.h) ----
static void dist_plan_release(gpimhip_ctx* h) {
    DistPlan& D = h->dplan;
    if (D.d_tiles) (void)hipFree(D.d_tiles);
    if (D.d_rect) (void)hipFree(D.d_rect);
    D = DistPlan();
}

static int upload_tiles(gpimhip_ctx* h, const std::vector<TileDesc>& tl, TileDesc** out) {
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, std::max<size_t>(tl.size(), 1) * sizeof(TileDesc)));
    *out = (TileDesc*)q;
    HIP_TRY(hipMemcpyAsync(q, tl.data(), tl.size() * sizeof(TileDesc), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return GPIMHIP_OK;
}

int gpimhip_dist_setup(gpimhip_handle h, int64_t n, int32_t world, int32_t rank) {
    FP64_ONLY(h);
    if (!h || n < 1 || world < 1 || rank < 0 || rank >= world) return GPIMHIP_E_BADARG;
    HIP_TRY(hipSetDevice(h->device));
    h->nbatch = 1;
    GP_TRY(ws_ensure_b(h, n, 1, 0, false));
    HIP_TRY(hipMemsetAsync(h->info, 0, sizeof(int32_t), h->stream));
    const int nb = (int)(h->np / NB);
    DistPlan& D = h->dplan;
    if (D.nb == nb && D.world == world && D.rank == rank) return GPIMHIP_OK;
    dist_plan_release(h);
    std::vector<TileDesc> tl;
    auto mark = [&](size_t start) { return PlanRange{(int64_t)start, (int32_t)(tl.size() - start)}; };
    D.colfill.assign(nb, {0, 0});
    const int npanel = (nb + OUTER_W - 1) / OUTER_W;
    D.upd_panel.assign(npanel, {0, 0});
    for (int j = 0; j < nb; ++j) {
        const int p0 = (j / OUTER_W) * OUTER_W;
        size_t s0 = tl.size();
        if (j > p0)
            for (int i = j + 1; i < nb; ++i) tl.push_back({i, j, p0, j});
        D.colfill[j] = mark(s0);
    }
    // trailing-update tiles of the owned panels, ascending: a round updates a contiguous range of them.  kb0 carries
    // the LOCAL block column of the output (GemmArgs::cmap), the k-range comes from kfix.
    int slot = 0;
    for (int c = 0; c < npanel; ++c) {
        if (c % world != rank) continue;
        size_t s0 = tl.size();
        const int j0 = c * OUTER_W, j1 = std::min(j0 + OUTER_W, nb);
        for (int ig = j0 / 8; ig <= (nb - 1) / 8; ++ig)
            for (int i = std::max(j0, ig * 8); i < std::min(nb, ig * 8 + 8); ++i)
                for (int j = j0; j < std::min(j1, i + 1); ++j) tl.push_back({i, j, slot * OUTER_W + (j - j0), 0});
        D.upd_panel[c] = mark(s0);
        ++slot;
    }
    // training: K^-1 = X^T X row panel by row panel (X = L^-1, broadcast panel c) against the owned columns, and the
    // owned lower tiles of K^-1 for the gradient contraction.  Local block lb <-> global block gblk[lb].
    std::vector<int> gblk;
    for (int c = 0; c < npanel; ++c)
        if (c % world == rank)
            for (int j = c * OUTER_W; j < std::min(c * OUTER_W + OUTER_W, nb); ++j) gblk.push_back(j);
    // (a rank's last owned panel may be the ragged one: its local blocks still sit at slot * 4 + offset)
    auto local_of = [&](int lbidx) { return (gblk[lbidx] / OUTER_W / world) * OUTER_W + gblk[lbidx] % OUTER_W; };
    D.kinv_panel.assign(npanel, {0, 0});
    for (int c = 0; c < npanel; ++c) {
        size_t s0 = tl.size();
        // 64 consecutive tiles = the panel's four block rows x 16 owned columns: dealt to the XCDs in chunks of 64
        // (gpimhip_dist_kinv_update) a patch streams 4 + 16 operand panels for 64 tiles (row by row it was 1 + 64, and the
        // pass ran at 56 TFLOP/s at N = 65536)
        const int ci1 = std::min(c * OUTER_W + OUTER_W, nb);
        for (size_t q0 = 0; q0 < gblk.size(); q0 += 16)
            for (int ci = c * OUTER_W; ci < ci1; ++ci)
                for (size_t q = q0; q < std::min(gblk.size(), q0 + 16); ++q)
                    if (gblk[q] <= ci) tl.push_back({ci, local_of((int)q), ci, nb});
        D.kinv_panel[c] = mark(s0);
    }
    {
        size_t s0 = tl.size();
        for (size_t q = 0; q < gblk.size(); ++q)
            for (int ci = gblk[q]; ci < nb; ++ci) tl.push_back({ci, gblk[q], local_of((int)q), 0});
        D.grad_tiles = mark(s0);
    }
    D.n_tiles = (int64_t)tl.size();
    GP_TRY(upload_tiles(h, tl, &D.d_tiles));
    D.nb = nb; D.world = world; D.rank = rank;
    return GPIMHIP_OK;
}

int gpimhip_dist_begin(gpimhip_handle h, int64_t n) { return gpimhip_dist_setup(h, n, 1, 0); }

int gpimhip_dist_panel_factor(gpimhip_handle h, double* Aloc, int64_t ldloc, int32_t loc_blk0, int32_t glob_blk0,
                              double* logdet_out, int32_t* info) {
    FP64_ONLY(h);
    if (!h || !Aloc || !info || loc_blk0 < 0 || glob_blk0 < 0 || !dist_plan_ok(h)) return GPIMHIP_E_BADARG;
    HIP_TRY(hipSetDevice(h->device));
    const int nb = (int)(h->np / NB);
    if (glob_blk0 >= nb || glob_blk0 % OUTER_W || ldloc < (int64_t)(loc_blk0 + std::min(OUTER_W, nb - glob_blk0)) * NB)
        return GPIMHIP_E_BADARG;
    h->nbatch = 1;
    // the panel's columns live at local block offset loc_blk0: shift the base so that GLOBAL column indices
    // address them (the chain only touches columns of this panel)
    double* As = Aloc - (int64_t)(glob_blk0 - loc_blk0) * NB;
    const int p1 = std::min(glob_blk0 + OUTER_W, nb);
    GP_TRY(launch_panel_chain(h, As, ldloc, glob_blk0, p1, nb, h->dplan.d_tiles, h->dplan.colfill.data() + glob_blk0, info));
    if (logdet_out)
        HIP_TRY(hipMemcpyAsync(logdet_out, h->logdet_part + glob_blk0, (size_t)(p1 - glob_blk0) * sizeof(double),
                               hipMemcpyDeviceToDevice, h->stream));
    return GPIMHIP_OK;
}

// The panel-wise vector solves of alpha = K^-1 y on the owner of panel [glob_blk0, glob_blk0 + 4) (distops.hip):
//   forward : piece (512) = Lpp^-1 (y_p - t);  acc[rows below the panel] += L(below, p) piece
//   backward: piece = Lpp^-T (z_p - L(below, p)^T a[below])
// y_p / z_p / t / piece: the panel's 512 entries (device); acc / a: full-length (np) vectors.  Needs the inverses of the
// panel's diagonal blocks, i.e. the handle that factored the panel (gpimhip_dist_panel_factor).
int gpimhip_dist_vec_forward(gpimhip_handle h, const double* Aloc, int64_t ldloc, int32_t loc_blk0, int32_t glob_blk0,
                             const double* y_p, const double* t, double* piece, double* acc) {
    FP64_ONLY(h);
    if (!h || !Aloc || !y_p || !piece || !acc || !dist_plan_ok(h) || glob_blk0 < 0 || loc_blk0 < 0 || glob_blk0 % OUTER_W)
        return GPIMHIP_E_BADARG;
    HIP_TRY(hipSetDevice(h->device));
    const int nb = (int)(h->np / NB);
    if (glob_blk0 >= nb) return GPIMHIP_E_BADARG;
    const int nblk = std::min(OUTER_W, nb - glob_blk0), w = nblk * NB;
    const int64_t r0 = (int64_t)glob_blk0 * NB;
    const double* P = Aloc + r0 * ldloc + (int64_t)loc_blk0 * NB;
    GP_TRY(launch_dist_trsv(h, P, ldloc, h->dinv + (int64_t)glob_blk0 * NB * NB, nblk, 0, y_p, t, piece));
    return launch_dist_rows_acc(h, P + (int64_t)w * ldloc, ldloc, h->np - r0 - w, w, piece, acc + r0 + w);
}
int gpimhip_dist_vec_backward(gpimhip_handle h, const double* Aloc, int64_t ldloc, int32_t loc_blk0, int32_t glob_blk0,
                              const double* z_p, const double* a, double* work, double* piece) {
    FP64_ONLY(h);
    if (!h || !Aloc || !z_p || !a || !work || !piece || !dist_plan_ok(h) || glob_blk0 < 0 || loc_blk0 < 0 ||
        glob_blk0 % OUTER_W)
        return GPIMHIP_E_BADARG;
    HIP_TRY(hipSetDevice(h->device));
    const int nb = (int)(h->np / NB);
    if (glob_blk0 >= nb) return GPIMHIP_E_BADARG;
    const int nblk = std::min(OUTER_W, nb - glob_blk0), w = nblk * NB;
    const int64_t r0 = (int64_t)glob_blk0 * NB, below = h->np - r0 - w;
    const double* P = Aloc + r0 * ldloc + (int64_t)loc_blk0 * NB;
    h->nbatch = 1;
    // work (512) = L(below, p)^T a[below]
    if (below > 0) GP_TRY(launch_gemv_t(h, P + (int64_t)w * ldloc, ldloc, below, w, a + r0 + w, work, 0, 0, 0, 0));
    else HIP_TRY(hipMemsetAsync(work, 0, (size_t)w * sizeof(double), h->stream));
    return launch_dist_trsv(h, P, ldloc, h->dinv + (int64_t)glob_blk0 * NB * NB, nblk, 1, z_p, work, piece);
}
// out (ncols) = A^T x for a row-major nrows x ncols matrix (ncols a multiple of 64): the posterior mean K*^T alpha of
// the distributed model (HBM-bound, fixed summation order)
int gpimhip_matvec_t(gpimhip_handle h, const double* A, int64_t ld, int64_t nrows, int64_t ncols, const double* x,
                     double* out) {
    FP64_ONLY(h);
    if (!h || !A || !x || !out || nrows < 1 || ncols < 64 || ncols % 64 || ld < ncols) return GPIMHIP_E_BADARG;
    HIP_TRY(hipSetDevice(h->device));
    h->nbatch = 1;
    return launch_gemv_t(h, A, ld, nrows, ncols, x, out, 0, 0, 0, 0);
}

int gpimhip_dist_panel_pack(gpimhip_handle h, const double* Aloc, int64_t ldloc, int32_t loc_blk0, int32_t glob_blk0,
                            double* buf, int64_t ldbuf) {
    FP64_ONLY(h);
    if (!h || !Aloc || !buf || !dist_plan_ok(h) || glob_blk0 < 0 || loc_blk0 < 0 || ldbuf < OUTER_W * NB)
        return GPIMHIP_E_BADARG;
    HIP_TRY(hipSetDevice(h->device));
    const int nb = (int)(h->np / NB);
    if (glob_blk0 >= nb || glob_blk0 % OUTER_W) return GPIMHIP_E_BADARG;
    const int nblk = std::min(OUTER_W, nb - glob_blk0);
    return launch_dist_pack(h, Aloc + (int64_t)loc_blk0 * NB, ldloc, (int64_t)glob_blk0 * NB, h->np, nblk * NB,
                            h->dinv + (int64_t)glob_blk0 * NB * NB, nblk, buf, ldbuf);
}

int gpimhip_dist_update(gpimhip_handle h, const double* buf, int64_t ldbuf, int32_t panel_glob_blk0, double* Aloc,
                        int64_t ldloc, int32_t panel_first, int32_t panel_last) {
    FP64_ONLY(h);
    if (!h || !buf || !Aloc || !dist_plan_ok(h) || panel_glob_blk0 < 0 || panel_glob_blk0 % OUTER_W)
        return GPIMHIP_E_BADARG;
    HIP_TRY(hipSetDevice(h->device));
    const DistPlan& D = h->dplan;
    const int nb = D.nb, npanel = (int)D.upd_panel.size();
    if (panel_glob_blk0 >= nb || panel_first <= panel_glob_blk0 / OUTER_W) return GPIMHIP_E_BADARG;
    h->nbatch = 1;
    panel_last = std::min<int32_t>(panel_last, npanel);
    int64_t off = -1, cnt = 0;
    for (int c = panel_first; c < panel_last; ++c) {
        if (!D.upd_panel[c].n) continue;
        if (off < 0) off = D.upd_panel[c].off;
        cnt += D.upd_panel[c].n;                     // owned panels are stored back to back, ascending
    }
    if (cnt == 0) return GPIMHIP_OK;
    // operands: the broadcast buffer holds columns [panel_glob_blk0, +kblk) of L for all rows; the shifted base lets
    // the tile engine use global block indices for them
    const int kblk = std::min(OUTER_W, nb - panel_glob_blk0);
    const double* Ps = buf - (int64_t)panel_glob_blk0 * NB;
    GemmArgs g = gemm_args(Ps, ldbuf, Ps, ldbuf, Aloc, ldloc, -1.0, 1.0, D.d_tiles + off, (int)cnt, h->np);
    g.kfix0 = panel_glob_blk0;
    g.kfix1 = panel_glob_blk0 + kblk;
    g.cmap = 1;
    return launch_gemm(h, false, false, EPI_STORE, g);
}

static int dist_rect_ensure(gpimhip_ctx* h, int cols) {
    DistPlan& D = h->dplan;
    if (D.rect_cols == cols && D.d_rect) return GPIMHIP_OK;
    if (D.d_rect) { (void)hipFree(D.d_rect); D.d_rect = nullptr; }
    std::vector<TileDesc> tl;
    tl.reserve((size_t)D.nb * cols);
    for (int r = 0; r < D.nb; ++r)
        for (int c = 0; c < cols; ++c) tl.push_back({r, c, 0, 1});
    D.n_rect = (int64_t)tl.size();
    GP_TRY(upload_tiles(h, tl, &D.d_rect));
    D.rect_cols = cols;
    return GPIMHIP_OK;
}

// One panel step of the forward substitution  W = L^-1 B  for this rank's right-hand sides B (np x mpad, row-major,
// destroyed), against the broadcast buffer of panel p (gpimhip_dist_panel_pack):
//   for b = 0..3:  W_b = Dinv_b B[4p+b]   ->  Wt rows [128 b, +128)
//                  B[4p+b+1 .. 4p+3] -= L[., 4p+b] W_b                       (rows inside the panel)
//   B[4p+4 ..] -= L[., panel p] Wt                                           (k-depth 512)
//   q[j] += sum over the panel's rows of W[r][j]^2
// The last line rewrites the rank's whole share of B below the panel once per panel: at N = 65536 the streamed inverse
// moved 4.4 TB that way and ran at 39 TFLOP/s.  Panels therefore go in PAIRS where the caller can hold two of them
// (gpimhip_dist_solve_update2): the first panel of a pair updates only the block rows of the second, and after the second
// panel's own solves ONE update of k-depth 1024 -- both panels side by side in a buffer of 1024 columns, both W's
// one below the other -- brings the rest of B up to date: half the passes over B.
static int dist_solve_panel(gpimhip_ctx* h, const double* buf, int64_t ldbuf, int g0, double* Bm, int64_t ldb, int cols,
                            double* Wt, int64_t ldw, int32_t col_tiles) {
    const DistPlan& D = h->dplan;
    const int nb = D.nb, nblk = std::min(OUTER_W, nb - g0);
    const double* dinv = buf + h->np * ldbuf;               // 128 x (128 nblk): the panel's diagonal-block inverses
    const double* Ps = buf - (int64_t)g0 * NB;              // global k-block index -> buffer column
    for (int b = 0; b < nblk; ++b) {
        // W_b = Dinv_b * B[g0 + b]   (NN: A = Dinv_b k-contiguous, B = right-hand sides m-contiguous)
        GemmArgs g = gemm_args(dinv + (int64_t)b * NB, ldbuf, Bm + (int64_t)(g0 + b) * NB * ldb, ldb,
                               Wt + (int64_t)b * NB * ldw, ldw, 1.0, 0.0, D.d_rect, cols, h->np);
        g.kfix0 = 0; g.kfix1 = 1;
        g.cj_max = col_tiles;
        GP_TRY(launch_gemm(h, false, true, EPI_STORE, g));
        const int rows_in = nblk - 1 - b;
        if (rows_in > 0) {
            // B[g0+b+1 .. g0+nblk-1] -= L[., g0+b] * W_b
            GemmArgs u = gemm_args(Ps, ldbuf, Wt + (int64_t)b * NB * ldw, ldw, Bm, ldb, -1.0, 1.0, D.d_rect,
                                   rows_in * cols, h->np);
            u.a_roff = g0 + b + 1; u.a_coff = g0 + b; u.c_roff = g0 + b + 1;
            u.kfix0 = 0; u.kfix1 = 1;
            u.cj_max = col_tiles;
            GP_TRY(launch_gemm(h, false, true, EPI_STORE, u));
        }
    }
    return GPIMHIP_OK;
}
// B[row0 .. row0 + nrows) -= L[., kb0 .. kb0 + kblk) * Wt  (block indices; abase: the buffer whose column 0 is block column kb0)
static int dist_solve_below(gpimhip_ctx* h, const double* abase, int64_t ldbuf, int kb0, int kblk, int row0, int nrows,
                            double* Bm, int64_t ldb, int cols, const double* Wt, int64_t ldw, int32_t col_tiles) {
    if (nrows <= 0) return GPIMHIP_OK;
    const DistPlan& D = h->dplan;
    const double* Ps = abase - (int64_t)kb0 * NB;
    // only the column tiles that are not structurally zero, in 8 x 8 patches, one patch per XCD at a time (until round 5:
    // the row-major list over ALL column tiles with the zero ones leaving at once -- three workgroups in four at the
    // middle of an inverse -- and a row of 100-500 tiles between two uses of a W panel: 37 TFLOP/s per launch)
    const int ncol = col_tiles > 0 ? std::min<int>(col_tiles, cols) : cols;
    GemmArgs u = gemm_args(Ps, ldbuf, Wt, ldw, Bm, ldb, -1.0, 1.0, D.d_rect, nrows * ncol, h->np);
    u.a_roff = row0; u.a_coff = kb0; u.c_roff = row0;
    u.kfix0 = 0; u.kfix1 = kblk;
    u.rect_rows = nrows; u.rect_cols = ncol;
    u.chunk = 64;
    return launch_gemm(h, false, true, EPI_STORE, u);
}
static int dist_solve_check(gpimhip_ctx* h, const double* buf, int32_t g0, const double* Bm, int64_t ldb, int64_t mpad,
                            const double* Wt, int64_t ldw) {
    if (!h || !buf || !Bm || !Wt || !dist_plan_ok(h) || g0 < 0 || g0 % OUTER_W || mpad < NB || mpad % NB || ldb < mpad ||
        ldw < mpad || g0 >= h->dplan.nb)
        return GPIMHIP_E_BADARG;
    return GPIMHIP_OK;
}
int gpimhip_dist_solve_update(gpimhip_handle h, const double* buf, int64_t ldbuf, int32_t panel_glob_blk0, double* Bm,
                              int64_t ldb, int64_t mpad, double* Wt, int64_t ldw, double* q, int32_t col_tiles) {
    FP64_ONLY(h);
    GP_TRY(dist_solve_check(h, buf, panel_glob_blk0, Bm, ldb, mpad, Wt, ldw));
    HIP_TRY(hipSetDevice(h->device));
    const DistPlan& D = h->dplan;
    const int nb = D.nb, g0 = panel_glob_blk0;
    h->nbatch = 1;
    const int cols = (int)(mpad / NB), nblk = std::min(OUTER_W, nb - g0);
    GP_TRY(dist_rect_ensure(h, cols));
    GP_TRY(dist_solve_panel(h, buf, ldbuf, g0, Bm, ldb, cols, Wt, ldw, col_tiles));
    GP_TRY(dist_solve_below(h, buf, ldbuf, g0, nblk, g0 + nblk, nb - g0 - nblk, Bm, ldb, cols, Wt, ldw, col_tiles));
    if (q) GP_TRY(launch_colsumsq_acc(h, Wt, ldw, nblk * NB, mpad, q));
    return GPIMHIP_OK;
}
// The same step for the panels of a PAIR held side by side: `wide` is a buffer of 1024 columns (leading dimension ldbuf
// >= 1024) with the pair's first panel in columns [0, 512) and the second in [512, 1024), both packed like the
// single-panel buffer (np rows + 128 rows of diagonal-block inverses); Wt2 holds 1024 rows.
//   second = 0: the pair's first panel (global block panel_glob_blk0): its own solves, W -> rows [0, 512) of Wt2, and the
//               update of the NEXT panel's block rows only;
//   second = 1: the pair's second panel (panel_glob_blk0 = the first panel's + 4): its own solves, W -> rows [512, 1024),
//               then every block row below it receives both panels at once (k-depth 1024).
int gpimhip_dist_solve_update2(gpimhip_handle h, const double* wide, int64_t ldbuf, int32_t panel_glob_blk0, double* Bm,
                               int64_t ldb, int64_t mpad, double* Wt2, int64_t ldw, double* q, int32_t col_tiles,
                               int32_t second) {
    FP64_ONLY(h);
    GP_TRY(dist_solve_check(h, wide, panel_glob_blk0, Bm, ldb, mpad, Wt2, ldw));
    if (ldbuf < 2 * OUTER_W * NB || (second && panel_glob_blk0 < OUTER_W)) return GPIMHIP_E_BADARG;
    HIP_TRY(hipSetDevice(h->device));
    const DistPlan& D = h->dplan;
    const int nb = D.nb, g0 = panel_glob_blk0;
    h->nbatch = 1;
    const int cols = (int)(mpad / NB), nblk = std::min(OUTER_W, nb - g0);
    GP_TRY(dist_rect_ensure(h, cols));
    const double* half = wide + (second ? OUTER_W * NB : 0);
    double* Wh = Wt2 + (second ? (int64_t)OUTER_W * NB * ldw : 0);
    GP_TRY(dist_solve_panel(h, half, ldbuf, g0, Bm, ldb, cols, Wh, ldw, col_tiles));
    if (!second)
        GP_TRY(dist_solve_below(h, half, ldbuf, g0, nblk, g0 + nblk, std::min(OUTER_W, nb - g0 - nblk), Bm, ldb, cols, Wh, ldw,
                                col_tiles));
    else
        GP_TRY(dist_solve_below(h, wide, ldbuf, g0 - OUTER_W, OUTER_W + nblk, g0 + nblk, nb - g0 - nblk, Bm, ldb, cols, Wt2,
                                ldw, col_tiles));
    if (q) GP_TRY(launch_colsumsq_acc(h, Wh, ldw, nblk * NB, mpad, q));
    return GPIMHIP_OK;
}

// ---- distributed training: covariance columns at u, K^-1 rows, gradient sums, finalize ----
int gpimhip_dist_kmat_cols(gpimhip_handle h, const gpimhip_model_t* m, const double* X, int64_t N, const double* u,
                           int64_t col0, int64_t ncols_pad, double* out, int64_t ld) {
    FP64_ONLY(h);
    if (!h || !X || !u || !out || N < 1 || col0 < 0 || ncols_pad < NB || ncols_pad % NB || ld < ncols_pad || !h->np)
        return GPIMHIP_E_BADARG;
    GP_TRY(check_model(m));
    HIP_TRY(hipSetDevice(h->device));
    h->nbatch = 1;
    GP_TRY(launch_theta(h, m, u));
    const int64_t M = std::max<int64_t>(0, std::min(ncols_pad, N - col0));
    GP_TRY(launch_kmat(h, m, X, N, X + std::min(col0, N - 1) * m->dim, M, h->theta, 0.0, 0, out, ld, h->np, ncols_pad, 0, 0, 0,
                       0, 0));
    return launch_add_diag_theta(h, out, ld, col0, M, std::min(ncols_pad, h->np - col0));
}

int gpimhip_dist_kinv_update_n(gpimhip_handle h, const double* xbuf, int64_t ldx, int32_t panel_glob_blk0, int32_t npanels,
                               const double* Xloc, int64_t ldloc, double* Kinv, int64_t ldk) {
    FP64_ONLY(h);
    if (!h || !xbuf || !Xloc || !Kinv || !dist_plan_ok(h) || panel_glob_blk0 < 0 || panel_glob_blk0 % OUTER_W ||
        panel_glob_blk0 >= h->dplan.nb || npanels < 1)
        return GPIMHIP_E_BADARG;
    HIP_TRY(hipSetDevice(h->device));
    h->nbatch = 1;
    const DistPlan& D = h->dplan;
    const int p0 = panel_glob_blk0 / OUTER_W, p1 = std::min<int>(p0 + npanels, (int)D.kinv_panel.size());
    // the tile lists of consecutive panels are adjacent in the plan (dist_plan_build): one launch for all of them
    PlanRange r = D.kinv_panel[p0];
    for (int p = p0 + 1; p < p1; ++p) r.n += D.kinv_panel[p].n;
    if (!r.n) return GPIMHIP_OK;
    // K^-1[ci, cj] = sum_{kb >= ci} X[kb, ci]^T X[kb, cj]: A = the broadcast column panels of X side by side (block column
    // ci - panel_glob_blk0 of xbuf), B = the owned columns
    GemmArgs g = gemm_args(xbuf, ldx, Xloc, ldloc, Kinv, ldk, 1.0, 0.0, D.d_tiles + r.off, r.n, h->np);
    g.a_coff = -panel_glob_blk0;
    g.krev = 1;
    g.chunk = 64;
    return launch_gemm(h, true, true, EPI_STORE, g);
}
int gpimhip_dist_kinv_update(gpimhip_handle h, const double* xbuf, int64_t ldx, int32_t panel_glob_blk0,
                             const double* Xloc, int64_t ldloc, double* Kinv, int64_t ldk) {
    return gpimhip_dist_kinv_update_n(h, xbuf, ldx, panel_glob_blk0, 1, Xloc, ldloc, Kinv, ldk);
}

int gpimhip_dist_grad_sums(gpimhip_handle h, const gpimhip_model_t* m, const double* X, int64_t N, const double* u,
                           const double* Kinv, int64_t ldk, const double* alpha, double* S_out) {
    FP64_ONLY(h);
    if (!h || !X || !u || !Kinv || !alpha || !S_out || !dist_plan_ok(h)) return GPIMHIP_E_BADARG;
    GP_TRY(check_model(m));
    HIP_TRY(hipSetDevice(h->device));
    h->nbatch = 1;
    const DistPlan& D = h->dplan;
    GP_TRY(launch_theta(h, m, u));
    if (D.grad_tiles.n)
        GP_TRY(launch_grad_reduce_tiles(h, m, Kinv, ldk, X, N, h->np, alpha, D.d_tiles + D.grad_tiles.off, D.grad_tiles.n,
                                        h->grad_part));
    return launch_sum7(h, h->grad_part, D.grad_tiles.n, S_out);
}

int gpimhip_dist_finalize(gpimhip_handle h, const gpimhip_model_t* m, int64_t N, double* u, const double* S, double quad,
                          double half_logdet, double lr, int32_t t, double* loss_out, double* grad_out, double* hist_row) {
    FP64_ONLY(h);
    if (!h || !u || !S || N < 1 || t < 0 || !h->np) return GPIMHIP_E_BADARG;
    GP_TRY(check_model(m));
    HIP_TRY(hipSetDevice(h->device));
    h->nbatch = 1;
    GP_TRY(launch_theta(h, m, u));
    AdamStep st;
    st.beta1 = 0.9; st.beta2 = 0.999; st.eps = 1e-8;
    st.lr_over_bc1 = t > 0 ? lr / (1.0 - pow(0.9, (double)t)) : 0.0;
    st.bc2_sqrt = t > 0 ? sqrt(1.0 - pow(0.999, (double)t)) : 1.0;
    if (t == 1) {
        HIP_TRY(hipMemsetAsync(h->adam_m, 0, MAXP * sizeof(double), h->stream));
        HIP_TRY(hipMemsetAsync(h->adam_v, 0, MAXP * sizeof(double), h->stream));
    }
    return launch_dist_finalize(h, m, N, S, quad, half_logdet, u, t > 0 ? 1 : 0, st, loss_out, grad_out, hist_row);
}

int gpimhip_dist_finalize_dev(gpimhip_handle h, const gpimhip_model_t* m, int64_t N, double* u, const double* red,
                              const double* quad, double lr, int32_t t, double* loss_out, double* grad_out,
                              double* hist_row) {
    FP64_ONLY(h);
    if (!h || !u || !red || !quad || N < 1 || t < 0) return GPIMHIP_E_BADARG;
    GP_TRY(check_model(m));
    HIP_TRY(hipSetDevice(h->device));
    h->nbatch = 1;
    // a rank that holds no block of the model (world size > number of reflection blocks) never built a workspace:
    // the parameters and the Adam state are all this call needs
    if (!h->np) GP_TRY(ws_ensure_b(h, 1, 1, 0, false));
    GP_TRY(launch_theta(h, m, u));
    AdamStep st;
    st.beta1 = 0.9; st.beta2 = 0.999; st.eps = 1e-8;
    st.lr_over_bc1 = t > 0 ? lr / (1.0 - pow(0.9, (double)t)) : 0.0;
    st.bc2_sqrt = t > 0 ? sqrt(1.0 - pow(0.999, (double)t)) : 1.0;
    if (t == 1) {
        HIP_TRY(hipMemsetAsync(h->adam_m, 0, MAXP * sizeof(double), h->stream));
        HIP_TRY(hipMemsetAsync(h->adam_v, 0, MAXP * sizeof(double), h->stream));
    }
    return launch_dist_finalize_dev(h, m, N, red, quad, u, t > 0 ? 1 : 0, st, loss_out, grad_out, hist_row);
}

int gpimhip_set_reflection(gpimhip_handle h, int32_t mask, const double* twoc, const double* wts, int64_t n_total,
                           int64_t var_count) {
    if (!h || mask < 0 || mask >= (1 << GPIMHIP_MAX_DIM) || (mask && !twoc) || n_total < 0 || var_count < 0) return GPIMHIP_E_BADARG;
    if (mask && h->fp32) {
        gpim_set_error("the symmetry-reduced model computes in double precision");
        return GPIMHIP_E_BADARG;
    }
    h->refl.mask = mask;
    for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) h->refl.twoc[k] = mask ? twoc[k] : 0.0;
    h->refl.wts = mask ? wts : nullptr;
    h->refl.n_total = mask ? n_total : 0;
    h->refl.var_count = mask ? var_count : 0;
    if (!mask) { h->refl.pb_off = 0; h->refl.pb_stride = 1; h->refl.nblocks_total = 0; h->refl.raw = 0; }
    return GPIMHIP_OK;
}

int gpimhip_set_reflection_shard(gpimhip_handle h, int32_t pb_off, int32_t pb_stride, int32_t nblocks_total, int32_t raw) {
    if (!h || pb_off < 0 || pb_stride < 1 || nblocks_total < 0 || nblocks_total > (1 << GPIMHIP_MAX_DIM)) return GPIMHIP_E_BADARG;
    h->refl.pb_off = pb_off;
    h->refl.pb_stride = pb_stride;
    h->refl.nblocks_total = nblocks_total;
    h->refl.raw = raw ? 1 : 0;
    return GPIMHIP_OK;
}

int gpimhip_refl_sums(gpimhip_handle h, const gpimhip_model_t* m, const double* X, const double* y, int64_t N, int32_t B,
                      const double* u, double* sums_out) {
    if (!h || !X || !y || !u || !sums_out || N < 1 || B < 1 || B > (1 << GPIMHIP_MAX_DIM) || !h->refl.mask) return GPIMHIP_E_BADARG;
    FP64_ONLY(h);
    GP_TRY(check_model(m));
    HIP_TRY(hipSetDevice(h->device));
    h->nbatch = B;
    HIP_TRY(hipMemsetAsync(h->info, 0, sizeof(int32_t), h->stream));
    GP_TRY(ws_ensure_padded(h, N));
    const int64_t np = h->np;
    GP_TRY(launch_pad_copy(h, y, N, h->ypad, np));
    GP_TRY(factor_at_u(h, m, X, 0, N, u));
    GP_TRY(launch_lauum(h, h->A, h->B, np, h->ld, rag_of(N, np)));
    GP_TRY(launch_grad_reduce_refl(h, m, h->B, h->ld, X, N, (int)(np / NB), h->alpha, 0));
    return launch_coupled_sums(h, np, sums_out);
}

int gpimhip_thin_batch(gpimhip_handle h, const double* vals, const int64_t* flat_idx, int32_t n, int32_t d,
                       const int64_t* shape, double dscale, int32_t max_out, int32_t* keep_out, int32_t* nkeep_out) {
    if (!h || !vals || !flat_idx || !shape || !keep_out || !nkeep_out || n < 1 || n > 1024 || d < 1 ||
        d > GPIMHIP_MAX_DIM || max_out < 1)
        return GPIMHIP_E_BADARG;
    HIP_TRY(hipSetDevice(h->device));
    return launch_thin_batch(h, vals, flat_idx, n, d, shape, dscale, max_out, keep_out, nkeep_out);
}

}  // extern "C"
