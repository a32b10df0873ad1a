// vfe.hip -- sparse (inducing-point) GP regression, variational free energy bound.
//
// Replaces pyro.contrib.gp.models.SparseGPRegression(approx="VFE") as built by
// reconstructor(sparse=True) (gpim/gpreg/gpr.py:145-155):
//   A = k(Xu,Xu) + jitter I = Luu Luu^T,  B = k(Xu,X),  W = Luu^-1 B   (Mu x N),  s = noise
//   Cc = I + W W^T / s = Lc Lc^T
//   loss = 1/2 [ N log 2pi + log|Cc| + N log s + y^T y / s - |Lc^-1 W y / s|^2 ]
//          + (N s2 - |W|_F^2) / (2 s)  + prior constant
// Trainable: the kernel parameters (as in the exact GP) AND the inducing inputs Xu.
// The gradient is analytic (no autograd):  with  beta = Cc^-1 W y / s,  r = y - W^T beta,
//   dF/dW = (1/s) [ Cc^-1 W - W - beta r^T ]          G_B' = Luu^-T (s dF/dW)       (dF/dB = G_B'/s)
//         G_B' is formed as  Q W - (Luu^-T beta) r^T  with the small matrix  Q = Luu^-T (Cc^-1 - I):  one Mu x Mu x N product
//         instead of three
//   dF/dA = 1/2 Luu^-T [ Cc^-1 + Cc - 2I + beta beta^T ] Luu^-1 = 1/2 G_A'
//   dF/ds = 1/2 [ -(Mu - tr Cc^-1)/s + N/s - y^T y/s^2 + 2 v^T beta/s - |W^T beta|^2/s^2 ]
//           - (N s2 - |W|^2)/(2 s^2),   v = W y / s
// and G_A', G_B' are contracted with dk/dtheta and dk/dXu by two fused reduction kernels.  All dense
// products run on the fp64 MFMA tile engine (gemm.hip) with triangular k-ranges; the two Cholesky
// factorisations and triangular inverses reuse the blocked drivers of the exact path.
//
// Lock-step batches (gpimhip_fit_vfe_batched / gpimhip_predict_vfe_batched): B models of equal N and Mu (the slices of
// config C5) advance together through every launch, blockIdx.y = model.  Mu x Mu matrices and vectors of the B models
// are stacked; the Mu x N matrices lie side by side as ONE mp x (B nq) matrix (model b = columns b nq ...), so that the
// k-chunks of P = W W^T of all models are again one linear batch dimension.  B = 1 is the single-model path.
#include <string.h>
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <map>
#include "kfun.hpp"
#include "theta.hpp"

typedef double d2 __attribute__((ext_vector_type(2)));

int launch_kmat(gpimhip_ctx* h, const gpimhip_model_t* m, const double* X, int64_t N, const double* Z,
                int64_t M, const ThetaDev* theta, double diag_add, int use_theta_diag, double* out,
                int64_t ld, int64_t rows_pad, int64_t cols_pad, int sym, int lower_only, int64_t x_bs,
                int64_t z_bs, int64_t out_bs);
int launch_potrf(gpimhip_ctx* h, double* A, int64_t np, int64_t ld, int32_t* info);
int launch_trtri(gpimhip_ctx* h, double* A, double* Tm, int64_t np, int64_t ld);
int launch_potrf_inv(gpimhip_ctx* h, double* A, double* Tm, int64_t np, int64_t ld, int32_t* info, int rag);
int launch_lauum(gpimhip_ctx* h, const double* A, double* B, int64_t np, int64_t ld, int rag);
int launch_trmv_lower(gpimhip_ctx* h, const double* L, int64_t ld, int64_t np, const double* y, double* z);
hipStream_t ensure_capture_stream(gpimhip_ctx* h);
void capture_lock(gpimhip_ctx* h);
void capture_unlock(gpimhip_ctx* h);
int launch_gemv_t(gpimhip_ctx* h, const double* A, int64_t ld, int64_t nrows, int64_t ncols, const double* x,
                  double* out, int tri, int64_t a_bs, int64_t x_bs, int64_t o_bs);
int launch_pad_copy(gpimhip_ctx* h, const double* src, int64_t n, double* dst, int64_t np);
int ws_ensure(gpimhip_ctx* h, int64_t N);
int vfe_finish_and_check(gpimhip_ctx* h);

// ------------------------------------------------------------------------------------------
// workspace + tile lists of the sparse path
// ------------------------------------------------------------------------------------------
struct VfeWs {
    int64_t mp = 0, nq = 0;       // padded Mu, padded N
    int B = 0;                    // models of a lock-step batch (every buffer below holds B of them)
    double *Vc = nullptr, *Cs = nullptr, *Mm = nullptr, *T1 = nullptr, *GA = nullptr;     // mp x mp
    double* Pp = nullptr;         // ksplit x mp x mp: partial sums of P = W W^T over k-chunks (vfe_forward)
    int ksplit = 1;               // number of k-chunks (a divisor of nbq)
    double *Bm = nullptr, *Wm = nullptr, *Y1 = nullptr;                                     // mp x nq
    double *yq = nullptr, *wtb = nullptr;                                                   // nq
    double *v = nullptr, *c1 = nullptr, *beta = nullptr, *bt = nullptr;                     // mp
    double *part_rect = nullptr, *part_sym = nullptr;    // per-tile theta sums [tiles][8]
    double *xu_rect = nullptr, *xu_sym = nullptr;        // [nbq][mp][4], [mb][mp][4]
    double *adam_m = nullptr, *adam_v = nullptr;         // P + Mu*d
    TileDesc* tiles = nullptr;
    int n_lowtri_rect = 0, off_lowtri_rect = 0;   // (ci, cj<nbq, kb in [0,ci])      W = Linv B
    int n_syrk = 0, off_syrk = 0;                 // (ci>=cj, kb in [0,nbq))         P = W W^T
    int n_sq_colge = 0, off_sq_colge = 0;         // (ci, cj<mb, kb in [cj,mb))      Mm Linv
    int n_sq_rowge = 0, off_sq_rowge = 0;         // (ci, cj<mb, kb in [ci,mb))      Linv^T T1
    int n_full_rect = 0, off_full_rect = 0;       // (ci, cj<nbq, kb in [0,mb))      Cc^-1 W
    // predict slab
    int64_t mc = 0;
    double *Ks = nullptr, *Ws = nullptr, *LW = nullptr;   // mp x mc
    TileDesc* ptiles = nullptr;
    int n_ptiles = 0;
};

// the sparse workspace hangs off the handle (gpimhip_ctx::vfe); the library keeps no global state

template <typename T>
static int valloc(T** p, int64_t n) {
    void* q = nullptr;
    if (hipMalloc(&q, std::max<int64_t>(n, 1) * sizeof(T)) != hipSuccess) {
        gpim_set_error("hipMalloc failed (sparse-GP workspace)");
        return GPIMHIP_E_NOMEM;
    }
    *p = (T*)q;
    return GPIMHIP_OK;
}
static void vfe_free(VfeWs& w) {
    void* ps[] = {w.Pp, w.Vc, w.Cs, w.Mm, w.T1, w.GA, w.Bm, w.Wm, w.Y1, w.bt, w.yq, w.wtb, w.v, w.c1, w.beta,
                  w.part_rect, w.part_sym, w.xu_rect, w.xu_sym, w.adam_m, w.adam_v, w.tiles, w.Ks, w.Ws, w.LW,
                  w.ptiles};
    for (void* p : ps)
        if (p) (void)hipFree(p);
    w = VfeWs();
}
void vfe_release(gpimhip_ctx* h) {
    VfeWs* w = static_cast<VfeWs*>(h->vfe);
    if (w) {
        vfe_free(*w);
        delete w;
        h->vfe = nullptr;
    }
}

static int vfe_ensure(gpimhip_ctx* h, int64_t Mu, int64_t N, int d, int P, int B, VfeWs** out) {
    if (!h->vfe) h->vfe = new VfeWs();
    VfeWs& w = *static_cast<VfeWs*>(h->vfe);
    const int64_t mp = pad_to(Mu, NB), nq = pad_to(N, NB);
    *out = &w;
    if (w.mp == mp && w.nq == nq && w.B == B) return GPIMHIP_OK;
    HIP_TRY(hipStreamSynchronize(h->stream));
    vfe_free(w);
    const int mb = (int)(mp / NB), nbq = (int)(nq / NB);
    GP_TRY(valloc(&w.Vc, B * mp * mp)); GP_TRY(valloc(&w.Cs, B * mp * mp)); GP_TRY(valloc(&w.Mm, B * mp * mp));
    GP_TRY(valloc(&w.T1, B * mp * mp)); GP_TRY(valloc(&w.GA, B * mp * mp));
    // P = W W^T has few output tiles (15 at Mu = 534) and a long k-range (N / 128 blocks): the k-range is cut into
    // `ksplit` equal chunks that run as the batch dimension of ONE launch (each into its own partial), summed in a fixed
    // order by vfe_cap_kernel.  302 -> 74 us per iteration at N = 6400 (config C5).
    w.ksplit = 1;
    for (int sdiv = 2; sdiv <= 16; ++sdiv)
        if (nbq % sdiv == 0 && nbq / sdiv >= 2) w.ksplit = sdiv;
    if (w.ksplit > 1) GP_TRY(valloc(&w.Pp, (int64_t)B * w.ksplit * mp * mp));
    GP_TRY(valloc(&w.Bm, B * mp * nq)); GP_TRY(valloc(&w.Wm, B * mp * nq));
    GP_TRY(valloc(&w.Y1, B * mp * nq)); GP_TRY(valloc(&w.bt, B * mp));
    GP_TRY(valloc(&w.yq, B * nq)); GP_TRY(valloc(&w.wtb, B * nq));
    GP_TRY(valloc(&w.v, B * mp)); GP_TRY(valloc(&w.c1, B * mp)); GP_TRY(valloc(&w.beta, B * mp));
    GP_TRY(valloc(&w.part_rect, (int64_t)B * mb * nbq * 8)); GP_TRY(valloc(&w.part_sym, (int64_t)B * mb * mb * 8));
    GP_TRY(valloc(&w.xu_rect, (int64_t)B * nbq * mp * 4)); GP_TRY(valloc(&w.xu_sym, (int64_t)B * mb * mp * 4));
    GP_TRY(valloc(&w.adam_m, B * (P + mp * GPIMHIP_MAX_DIM))); GP_TRY(valloc(&w.adam_v, B * (P + mp * GPIMHIP_MAX_DIM)));
    std::vector<TileDesc> tl;
    auto mark = [&](int& off, int& n, size_t s) { off = (int)s; n = (int)(tl.size() - s); };
    size_t s = tl.size();
    for (int ci = mb - 1; ci >= 0; --ci)
        for (int cj = 0; cj < nbq; ++cj) tl.push_back({ci, cj, 0, ci + 1});
    mark(w.off_lowtri_rect, w.n_lowtri_rect, s);
    s = tl.size();
    for (int ci = 0; ci < mb; ++ci)
        for (int cj = 0; cj <= ci; ++cj) tl.push_back({ci, cj, 0, nbq / w.ksplit});
    mark(w.off_syrk, w.n_syrk, s);
    s = tl.size();
    for (int cj = 0; cj < mb; ++cj)
        for (int ci = 0; ci < mb; ++ci) tl.push_back({ci, cj, cj, mb});
    mark(w.off_sq_colge, w.n_sq_colge, s);
    s = tl.size();
    for (int ci = 0; ci < mb; ++ci)
        for (int cj = 0; cj < mb; ++cj) tl.push_back({ci, cj, ci, mb});
    mark(w.off_sq_rowge, w.n_sq_rowge, s);
    s = tl.size();
    for (int ci = 0; ci < mb; ++ci)
        for (int cj = 0; cj < nbq; ++cj) tl.push_back({ci, cj, 0, mb});
    mark(w.off_full_rect, w.n_full_rect, s);
    GP_TRY(valloc(&w.tiles, (int64_t)tl.size()));
    HIP_TRY(hipMemcpyAsync(w.tiles, tl.data(), tl.size() * sizeof(TileDesc), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    w.mp = mp;
    w.nq = nq;
    w.B = B;
    (void)d;
    return GPIMHIP_OK;
}

static GemmArgs vg(const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc, double alpha,
                   double beta, const TileDesc* t, int n) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
    g.alpha = alpha; g.beta = beta; g.tiles = t; g.ntiles = n; g.chunk = 64;
    return g;
}
// ... with the per-model strides of a lock-step batch; the tile engine picks the launch shape (whole tiles, quadrants, 4 or
// 8 waves -- which differ in the order of their k-steps) from the tiles of ONE model, so that a model's bits do not depend
// on the batch it runs in (nmodels = B)
static GemmArgs vgb(int nmodels, const double* A, int64_t lda, int64_t sA, const double* B, int64_t ldb, int64_t sB, double* C, int64_t ldc,
                    int64_t sC, const TileDesc* t, int n) {
    GemmArgs g = vg(A, lda, B, ldb, C, ldc, 1.0, 0.0, t, n);
    g.sA = sA; g.sB = sB; g.sC = sC;
    g.shape_div = nmodels;
    return g;
}

// u -> theta for B models whose parameter vectors are ustride doubles apart ([u_theta | Xu] each)
__global__ void theta_kernel_strided(gpimhip_model_t m, const double* __restrict__ u, int64_t ustride, ThetaDev* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        ThetaDev t;
        theta_from_u(m, u + blockIdx.y * ustride, t);
        out[blockIdx.y] = t;
    }
}

// ------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------
// Cc = I + P/s on the lower triangle (into Pm) and as a full symmetric copy in Cs; P = the sum of `nsplit` partials
// Pp[c] (fixed order; nsplit == 1: P is in Pm already)
__global__ void vfe_cap_kernel(double* __restrict__ Pm, double* __restrict__ Cs, int64_t mp,
                               const ThetaDev* __restrict__ th, const double* __restrict__ Pp, int nsplit) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= mp * mp) return;
    Pm += blockIdx.y * mp * mp;
    Cs += blockIdx.y * mp * mp;
    if (Pp) Pp += (int64_t)blockIdx.y * nsplit * mp * mp;
    th += blockIdx.y;
    const int64_t i = idx / mp, j = idx % mp;
    if (j > i) return;
    double pv;
    if (nsplit > 1) {
        pv = Pp[idx];
        for (int c = 1; c < nsplit; ++c) pv += Pp[(int64_t)c * mp * mp + idx];
    } else {
        pv = Pm[idx];
    }
    const double c = ((i == j) ? 1.0 : 0.0) + pv / th->noise;
    Pm[i * mp + j] = c;
    Cs[i * mp + j] = c;
    Cs[j * mp + i] = c;
}

// out[i] = sum_j A[i][j] x[j] (j < ncols), one wave per row
__global__ __launch_bounds__(256) void gemv_n_kernel(const double* __restrict__ A, int64_t ld, int64_t nrows,
                                                     int64_t ncols, const double* __restrict__ x,
                                                     double* __restrict__ out, int64_t a_bs, int64_t x_bs, int64_t o_bs) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i = (int64_t)blockIdx.x * 4 + wave;
    if (i >= nrows) return;
    A += blockIdx.y * a_bs;
    x += blockIdx.y * x_bs;
    out += blockIdx.y * o_bs;
    const double* row = A + i * ld;
    double s = 0.0;
    for (int64_t j = lane * 2; j < ncols; j += 128) {
        const d2 a = *reinterpret_cast<const d2*>(row + j);
        const d2 v = *reinterpret_cast<const d2*>(x + j);
        s = fma(a[0], v[0], s);
        s = fma(a[1], v[1], s);
    }
    s += __shfl_xor(s, 32); s += __shfl_xor(s, 16); s += __shfl_xor(s, 8);
    s += __shfl_xor(s, 4); s += __shfl_xor(s, 2); s += __shfl_xor(s, 1);
    if (lane == 0) out[i] = s;
}

// v <- v / s
__global__ void vfe_scale_kernel(double* __restrict__ v, int64_t n, const ThetaDev* __restrict__ th) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    v += blockIdx.y * n;
    th += blockIdx.y;
    if (i < n) v[i] = v[i] / th->noise;
}

// G <- G - bt (y - W^T beta)^T      (the rank-one part of G_B' = Luu^-T (s dF/dW); bt = Luu^-T beta)
__global__ void vfe_gw_kernel(double* __restrict__ G, const double* __restrict__ bt, const double* __restrict__ yq,
                              const double* __restrict__ wtb, int64_t mp, int64_t nq, int64_t ld) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= mp * nq) return;
    const int64_t m = idx / nq, n = idx % nq, e = m * ld + blockIdx.y * nq + n;      // model b: columns b nq ...
    bt += blockIdx.y * mp;
    yq += blockIdx.y * nq;
    wtb += blockIdx.y * nq;
    G[e] = G[e] - bt[m] * (yq[n] - wtb[n]);
}

// Mm = Cc^-1 + Cc - 2I + beta beta^T   (full, symmetric; Vc holds the lower triangle of Cc^-1); Vs = Cc^-1 - I as a full matrix
__global__ void vfe_mmat_kernel(const double* __restrict__ Vc, const double* __restrict__ Cs,
                                const double* __restrict__ beta, double* __restrict__ Mm, double* __restrict__ Vs, int64_t mp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= mp * mp) return;
    Vc += blockIdx.y * mp * mp;
    Cs += blockIdx.y * mp * mp;
    Mm += blockIdx.y * mp * mp;
    Vs += blockIdx.y * mp * mp;
    beta += blockIdx.y * mp;
    const int64_t i = idx / mp, j = idx % mp;
    const double cinv = (j <= i) ? Vc[i * mp + j] : Vc[j * mp + i];
    Vs[idx] = cinv - ((i == j) ? 1.0 : 0.0);
    Mm[idx] = cinv + Cs[idx] - ((i == j) ? 2.0 : 0.0) + beta[i] * beta[j];
}

// ------------------------------------------------------------------------------------------
// gradient contractions.  G: (rows = inducing points) x (cols = Z points), Z = X (rect) or Xu (sym)
//   part[tile][0]   += G e                      (d/d s2)
//   part[tile][1+k] += G h (a_k - b_k)^2        (d/d l_k)
//   part[tile][6]   += G ga                     (d/d alpha, RQ)
//   xu[cj][m][k]    += G h (a_mk - b_nk)        (d/d Xu_mk, up to -s2/l_k)
// ------------------------------------------------------------------------------------------
template <int KIND>
__global__ __launch_bounds__(256) void vfe_grad_kernel(const double* __restrict__ G, int64_t ld,
                                                       const double* __restrict__ Xu, int64_t Mu,
                                                       const double* __restrict__ Z, int64_t Nz, int d,
                                                       const ThetaDev* __restrict__ th, int ntc,
                                                       double* __restrict__ part, double* __restrict__ xu_part,
                                                       int64_t mp, int64_t g_bs, int64_t xu_bs, int64_t z_bs,
                                                       int64_t part_bs, int64_t xup_bs) {
    __shared__ double xa[128][5];
    __shared__ double xz[128][5];
    __shared__ double red[4][8];
    __shared__ double rowacc[128][4][4];      // [row][column-lane-group tx&3][k]  (fixed-order combine below)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ci = blockIdx.x / ntc, cj = blockIdx.x % ntc;
    G += blockIdx.y * g_bs;
    Xu += blockIdx.y * xu_bs;
    Z += blockIdx.y * z_bs;
    part += blockIdx.y * part_bs;
    xu_part += blockIdx.y * xup_bs;
    const ThetaDev t = th[blockIdx.y];
    {
        const bool isrow = tid < 128;
        const int loc = tid & 127;
        const int64_t g = (int64_t)(isrow ? ci : cj) * 128 + loc;
        const double* src = isrow ? Xu : Z;
        const int64_t lim = isrow ? Mu : Nz;
        double s2 = 0.0;
        double (*dst)[5] = isrow ? xa : xz;
        for (int k = 0; k < GPIMHIP_MAX_DIM; ++k) {
            double a = 0.0;
            if (k < d && g < lim) a = src[g * d + k] / t.ls[k];
            dst[loc][k] = a;
            s2 += a * a;
        }
        dst[loc][4] = s2;
    }
    __syncthreads();
    double S[7] = {0, 0, 0, 0, 0, 0, 0};
    const int ty = tid >> 4, tx = tid & 15;
    for (int rr = 0; rr < 8; ++rr) {
        const int r = ty + 16 * rr;
        const int64_t gi = (int64_t)ci * 128 + r;
        const double a0 = xa[r][0], a1 = xa[r][1], a2 = xa[r][2], a3 = xa[r][3], an = xa[r][4];
        double x0 = 0, x1 = 0, x2 = 0, x3 = 0;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            const d2 gv = *reinterpret_cast<const d2*>(G + gi * ld + (int64_t)cj * 128 + tx * 2 + 32 * cc);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int c = tx * 2 + 32 * cc + e;
                const int64_t gj = (int64_t)cj * 128 + c;
                if (gi >= Mu || gj >= Nz) continue;
                const double w = gv[e];
                double dot = a0 * xz[c][0];
                dot = fma(a1, xz[c][1], dot);
                dot = fma(a2, xz[c][2], dot);
                dot = fma(a3, xz[c][3], dot);
                const double r2 = clamp0_nan((an - 2.0 * dot) + xz[c][4]);
                const KVal kv = kfun_grad<KIND>(r2, t.alpha);
                S[0] = fma(w, kv.e, S[0]);
                const double wh = w * kv.h;
                const double d0 = a0 - xz[c][0], d1 = a1 - xz[c][1], d2_ = a2 - xz[c][2], d3 = a3 - xz[c][3];
                S[1] = fma(wh, d0 * d0, S[1]);
                S[2] = fma(wh, d1 * d1, S[2]);
                S[3] = fma(wh, d2_ * d2_, S[3]);
                S[4] = fma(wh, d3 * d3, S[4]);
                if (KIND == GPIMHIP_KERNEL_RQ) S[6] = fma(w, kv.ga, S[6]);
                x0 = fma(wh, d0, x0);
                x1 = fma(wh, d1, x1);
                x2 = fma(wh, d2_, x2);
                x3 = fma(wh, d3, x3);
            }
        }
        // the 16 threads tx = 0..15 of this row: combine 4 neighbours by shuffle, keep 4 partials
        x0 += __shfl_xor(x0, 1); x0 += __shfl_xor(x0, 2);
        x1 += __shfl_xor(x1, 1); x1 += __shfl_xor(x1, 2);
        x2 += __shfl_xor(x2, 1); x2 += __shfl_xor(x2, 2);
        x3 += __shfl_xor(x3, 1); x3 += __shfl_xor(x3, 2);
        if ((tx & 3) == 0) {
            rowacc[r][tx >> 2][0] = x0;
            rowacc[r][tx >> 2][1] = x1;
            rowacc[r][tx >> 2][2] = x2;
            rowacc[r][tx >> 2][3] = x3;
        }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        double v = S[k];
        v += __shfl_xor(v, 32); v += __shfl_xor(v, 16); v += __shfl_xor(v, 8);
        v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (tid < 8) {
        double v = 0.0;
        if (tid < 7) v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
        part[(int64_t)blockIdx.x * 8 + tid] = v;
    }
    for (int e = tid; e < 128 * 4; e += 256) {
        const int r = e >> 2, k = e & 3;
        const double v = (rowacc[r][0][k] + rowacc[r][1][k]) + (rowacc[r][2][k] + rowacc[r][3][k]);
        xu_part[((int64_t)cj * mp + (int64_t)ci * 128 + r) * 4 + k] = v;
    }
}

// ------------------------------------------------------------------------------------------
// finalize: loss, gradient w.r.t. [u_theta | Xu], Adam on all of it, history rows
// ------------------------------------------------------------------------------------------
struct VfeFinalArgs {
    gpimhip_model_t m;
    int64_t N, Mu, mp, nq;
    int mb, nbq;
    const double *part_rect, *part_sym, *xu_rect, *xu_sym;
    const double *yq, *wtb, *v, *c1, *beta, *Cs, *Vc, *logdet_part;
    const ThetaDev* th;
    double* u;                      // [P + Mu*d]
    double *adam_m, *adam_v;
    int do_adam;
    int32_t* iter; const double* bc; int32_t T;
    double *hist_theta, *hist_xu, *loss_out, *grad_out;
    int32_t* info;                  // [0] factorisation status, [1] iterations completed at the first failure
    int64_t ustride, na;            // lock-step batch (blockIdx.x = model): doubles per model in u / in the Adam state
};

// sum over the 256 threads of the workgroup, the same value in every thread: a shuffle tree inside each wave, then the four
// wave sums in a fixed order (two barriers; the LDS tree it replaces took nine: 21 sums were 50 of the kernel's 57 us)
__device__ double vfe_block_sum(double v, double* red) {
    const int tid = threadIdx.x;
    v += __shfl_xor(v, 32); v += __shfl_xor(v, 16); v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4); v += __shfl_xor(v, 2); v += __shfl_xor(v, 1);
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    const double r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void vfe_finalize_kernel(VfeFinalArgs a) {
    __shared__ double red[256];
    __shared__ double R[8], Q[8];
    const int tid = threadIdx.x;
    const int d = a.m.dim;
    const int P = 2 + a.m.n_ls + (a.m.kernel == GPIMHIP_KERNEL_RQ ? 1 : 0);
    {   // this workgroup's model
        const int64_t b = blockIdx.x;
        a.part_rect += b * a.mb * a.nbq * 8; a.part_sym += b * a.mb * a.mb * 8;
        a.yq += b * a.nq; a.wtb += b * a.nq;
        a.v += b * a.mp; a.c1 += b * a.mp; a.beta += b * a.mp;
        a.Cs += b * a.mp * a.mp; a.Vc += b * a.mp * a.mp;
        a.logdet_part += b * a.mb; a.th += b;
        a.u += b * a.ustride; a.adam_m += b * a.na; a.adam_v += b * a.na;
        if (a.iter) {
            a.iter += b;
            if (a.hist_theta) a.hist_theta += b * a.T * P;
            if (a.loss_out) a.loss_out += b * a.T;
        } else if (a.loss_out) {
            a.loss_out += b;
        }
        if (a.grad_out) a.grad_out += b * a.ustride;
    }
    // a factorisation of this training loop failed: freeze u, Xu, the Adam state and the histories at
    // the failing iteration (the reference raises there, gpr.py:192) and record how far the loop got
    if (a.iter && *a.info != 0) {
        if (tid == 0) atomicMin(a.info + 1, *a.iter);
        return;
    }
    for (int k = 0; k < 7; ++k) {
        double v = 0.0;
        for (int q = tid; q < a.mb * a.nbq; q += 256) v += a.part_rect[(int64_t)q * 8 + k];
        v = vfe_block_sum(v, red);
        if (tid == 0) R[k] = v;
        v = 0.0;
        for (int q = tid; q < a.mb * a.mb; q += 256) v += a.part_sym[(int64_t)q * 8 + k];
        v = vfe_block_sum(v, red);
        if (tid == 0) Q[k] = v;
    }
    double yy = 0, c1c1 = 0, vb = 0, wb2 = 0, trC = 0, trCi = 0, lg = 0;
    for (int64_t i = tid; i < a.N; i += 256) { yy = fma(a.yq[i], a.yq[i], yy); wb2 = fma(a.wtb[i], a.wtb[i], wb2); }
    for (int64_t i = tid; i < a.Mu; i += 256) {
        c1c1 = fma(a.c1[i], a.c1[i], c1c1);
        vb = fma(a.v[i], a.beta[i], vb);
        trC += a.Cs[i * a.mp + i] - 1.0;
        trCi += a.Vc[i * a.mp + i];
    }
    for (int k = tid; k < a.mb; k += 256) lg += a.logdet_part[k];
    yy = vfe_block_sum(yy, red); wb2 = vfe_block_sum(wb2, red); c1c1 = vfe_block_sum(c1c1, red);
    vb = vfe_block_sum(vb, red); trC = vfe_block_sum(trC, red); trCi = vfe_block_sum(trCi, red);
    lg = vfe_block_sum(lg, red);
    const ThetaDev t = *a.th;
    const double s = t.noise, Nd = (double)a.N, Md = (double)a.Mu;
    const double w2 = s * trC;                        // |W|_F^2 = tr(W W^T) = s tr(Cc - I)
    const double tr_raw = Nd * t.var - w2;
    if (tid == 0) {
        const double trace_term = (tr_raw > 0.0 ? tr_raw : 0.0) / s;
        const double logdet = 2.0 * lg + Nd * log(s);
        const double mahal = yy / s - c1c1;
        double prior = log(a.m.amp_hi - a.m.amp_lo);
        for (int k = 0; k < a.m.n_ls; ++k) prior += log(a.m.ls_hi[k] - a.m.ls_lo[k]);
        const double loss = 0.5 * (Nd * 1.8378770664093453 + logdet + mahal) + 0.5 * trace_term + prior;
        // d/d theta
        double g[MAXP];
        const double gs2 = 0.5 * Q[0] + R[0] / s + Nd / (2.0 * s);
        g[0] = gs2 * t.dvar_du;
        if (a.m.n_ls == 1) {
            double acc = 0.0;
            for (int k = 0; k < d; ++k) acc += 0.5 * Q[1 + k] + R[1 + k] / s;
            g[1] = acc * t.var / t.ls[0] * t.dls_du[0];
        } else {
            for (int k = 0; k < d; ++k) g[1 + k] = (0.5 * Q[1 + k] + R[1 + k] / s) * t.var / t.ls[k] * t.dls_du[k];
        }
        const double gs = 0.5 * (-(Md - trCi) / s + Nd / s - yy / (s * s) + 2.0 * vb / s - wb2 / (s * s)) -
                          tr_raw / (2.0 * s * s);
        g[1 + a.m.n_ls] = gs * t.dnoise_du;
        if (a.m.kernel == GPIMHIP_KERNEL_RQ) g[2 + a.m.n_ls] = (0.5 * Q[6] + R[6] / s) * t.var * t.dalpha_du;
        AdamStep st;
        st.beta1 = 0.9; st.beta2 = 0.999; st.eps = 1e-8; st.lr_over_bc1 = 0.0; st.bc2_sqrt = 1.0;
        int it = 0;
        double* loss_out = a.loss_out;
        double* hist_row = nullptr;
        if (a.iter) {
            it = *a.iter;
            st.lr_over_bc1 = a.bc[it];
            st.bc2_sqrt = a.bc[a.T + it];
            if (loss_out) loss_out += it;
            if (a.hist_theta) hist_row = a.hist_theta + (int64_t)it * P;
            *a.iter = it + 1;
        }
        if (loss_out) *loss_out = loss;
        if (a.grad_out)
            for (int k = 0; k < P; ++k) a.grad_out[k] = g[k];
        if (a.do_adam) {
            for (int k = 0; k < P; ++k) {
                double mm = a.adam_m[k], vv = a.adam_v[k];
                mm = mm + (g[k] - mm) * (1.0 - st.beta1);
                vv = vv * st.beta2 + (1.0 - st.beta2) * g[k] * g[k];
                const double denom = sqrt(vv) / st.bc2_sqrt + st.eps;
                a.u[k] = a.u[k] + (-st.lr_over_bc1) * (mm / denom);
                a.adam_m[k] = mm;
                a.adam_v[k] = vv;
            }
            if (hist_row) {
                ThetaDev tn;
                theta_from_u(a.m, a.u, tn);
                hist_row[0] = tn.var;
                for (int k = 0; k < a.m.n_ls; ++k) hist_row[1 + k] = tn.ls[k];
                hist_row[1 + a.m.n_ls] = tn.noise;
                if (a.m.kernel == GPIMHIP_KERNEL_RQ) hist_row[2 + a.m.n_ls] = tn.alpha;
            }
        }
    }
}

// inducing inputs: g_xu[m][k] = -(s2 / l_k) * ( Xq[m][k] + Xr[m][k] / s ), Adam step and history row; one thread per
// coordinate (inside the single workgroup of vfe_finalize_kernel this loop was 100 of the kernel's 133 us at Mu = 534).
// Launched BEFORE vfe_finalize_kernel, which advances the iteration counter.
struct VfeXuArgs {
    int d, P, mb, nbq;
    int64_t Mu, mp;
    const double *xu_rect, *xu_sym;
    const ThetaDev* th;
    double* u;
    double *adam_m, *adam_v;
    int do_adam;
    const int32_t* iter; const double* bc; int32_t T;
    double *hist_xu, *grad_out;
    const int32_t* info;
    int64_t ustride, na;            // lock-step batch (blockIdx.y = model)
};
__global__ __launch_bounds__(256) void vfe_xu_kernel(VfeXuArgs a) {
    const int64_t e = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= a.Mu * a.d) return;
    if (a.iter && *a.info != 0) return;         // a factorisation of this loop failed: everything stays frozen
    {
        const int64_t b = blockIdx.y;
        a.xu_rect += b * a.nbq * a.mp * 4; a.xu_sym += b * a.mb * a.mp * 4;
        a.th += b; a.u += b * a.ustride; a.adam_m += b * a.na; a.adam_v += b * a.na;
        if (a.iter) a.iter += b;
        if (a.hist_xu) a.hist_xu += b * a.T * a.Mu * a.d;
        if (a.grad_out) a.grad_out += b * a.ustride;
    }
    const ThetaDev t = *a.th;
    const double s = t.noise;
    const int64_t mrow = e / a.d;
    const int k = (int)(e % a.d);
    double xr = 0.0, xq = 0.0;
#pragma unroll 10
    for (int cj = 0; cj < a.nbq; ++cj) xr += a.xu_rect[((int64_t)cj * a.mp + mrow) * 4 + k];
    for (int cj = 0; cj < a.mb; ++cj) xq += a.xu_sym[((int64_t)cj * a.mp + mrow) * 4 + k];
    const double gx = -(t.var / t.ls[k]) * (xq + xr / s);
    if (a.grad_out) a.grad_out[a.P + e] = gx;
    if (a.do_adam) {
        const double beta1 = 0.9, beta2 = 0.999, eps = 1e-8;
        int it = 0;
        double lr_over_bc1 = 0.0, bc2_sqrt = 1.0;
        if (a.iter) { it = *a.iter; lr_over_bc1 = a.bc[it]; bc2_sqrt = a.bc[a.T + it]; }
        double mm = a.adam_m[a.P + e], vv = a.adam_v[a.P + e];
        mm = mm + (gx - mm) * (1.0 - beta1);
        vv = vv * beta2 + (1.0 - beta2) * gx * gx;
        const double denom = sqrt(vv) / bc2_sqrt + eps;
        const double xn = a.u[a.P + e] + (-lr_over_bc1) * (mm / denom);
        a.u[a.P + e] = xn;
        a.adam_m[a.P + e] = mm;
        a.adam_v[a.P + e] = vv;
        if (a.hist_xu) a.hist_xu[(int64_t)it * a.Mu * a.d + e] = xn;
    }
}

// per test column: mean = sum_m c1[m] LW[m][j];  var = s2 + s - sum_m Ws^2 + sum_m LW^2
__global__ void vfe_predict_cols_kernel(const double* __restrict__ Ws, const double* __restrict__ LW, int64_t ld,
                                        const double* __restrict__ c1, int64_t mp, int64_t cnt,
                                        const ThetaDev* __restrict__ th, double* __restrict__ mean,
                                        double* __restrict__ var, int64_t w_bs, int64_t o_bs) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= cnt) return;
    Ws += blockIdx.y * w_bs;
    LW += blockIdx.y * w_bs;
    c1 += blockIdx.y * mp;
    th += blockIdx.y;
    mean += blockIdx.y * o_bs;
    var += blockIdx.y * o_bs;
    double mu = 0.0, q1 = 0.0, q2 = 0.0;
    for (int64_t m = 0; m < mp; ++m) {
        const double w = Ws[m * ld + j], l = LW[m * ld + j];
        mu = fma(c1[m], l, mu);
        q1 = fma(w, w, q1);
        q2 = fma(l, l, q2);
    }
    mean[j] = mu;
    var[j] = th->var + th->noise - q1 + q2;
}

// ------------------------------------------------------------------------------------------
// host driver
// ------------------------------------------------------------------------------------------
static int launch_grad(gpimhip_ctx* h, const gpimhip_model_t* m, const double* G, int64_t ld, int64_t g_bs, const double* Xu,
                       int64_t xu_bs, int64_t Mu, const double* Z, int64_t z_bs, int64_t Nz, int ntr, int ntc, double* part,
                       double* xu_part, int64_t mp, int B) {
    dim3 grid(ntr * ntc, B), block(256);
    const int64_t part_bs = (int64_t)ntr * ntc * 8, xup_bs = (int64_t)ntc * mp * 4;
#define VG_LAUNCH(KIND)                                                                                    \
    hipLaunchKernelGGL((vfe_grad_kernel<KIND>), grid, block, 0, h->stream, G, ld, Xu, Mu, Z, Nz, m->dim,   \
                       h->theta, ntc, part, xu_part, mp, g_bs, xu_bs, z_bs, part_bs, xup_bs)
    switch (m->kernel) {
        case GPIMHIP_KERNEL_RBF: VG_LAUNCH(GPIMHIP_KERNEL_RBF); break;
        case GPIMHIP_KERNEL_MATERN52: VG_LAUNCH(GPIMHIP_KERNEL_MATERN52); break;
        default: VG_LAUNCH(GPIMHIP_KERNEL_RQ); break;
    }
#undef VG_LAUNCH
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// The B models of a call: u holds B vectors [u_theta (P) | Xu (Mu x d)] one after the other, X advances by x_bs doubles per
// model (0: shared inputs).  h->nbatch == w.B throughout (vfe_prepare).
struct VfeProb { const double* X; int64_t x_bs; int64_t N, Mu; int P; int64_t ustride; };

// theta, Luu^-1 (h->A), W (w.Wm), Cc (w.Cs full), Lc^-1 (h->B), v, c1: shared by loss/grad and predict
static int vfe_forward(gpimhip_ctx* h, VfeWs& w, const gpimhip_model_t* m, const VfeProb& q, const double* u) {
    const int64_t mp = w.mp, nq = w.nq, mm = mp * mp;
    const int B = w.B;
    const int64_t ldw = (int64_t)B * nq;
    const double* Xu = u + q.P;
    hipLaunchKernelGGL(theta_kernel_strided, dim3(1, B), dim3(64), 0, h->stream, *m, u, q.ustride, h->theta);
    GP_TRY(launch_kmat(h, m, Xu, q.Mu, nullptr, q.Mu, h->theta, m->jitter, 0, h->A, mp, mp, mp, 1, 1, q.ustride, q.ustride, mm));
    GP_TRY(launch_potrf_inv(h, h->A, h->Tm, mp, mp, h->info, 0));
    GP_TRY(launch_kmat(h, m, Xu, q.Mu, q.X, q.N, h->theta, 0.0, 0, w.Bm, ldw, mp, nq, 0, 0, q.ustride, q.x_bs, nq));
    {   // W = Luu^-1 B
        GemmArgs g = vgb(B, h->A, mp, mm, w.Bm, ldw, nq, w.Wm, ldw, nq, w.tiles + w.off_lowtri_rect, w.n_lowtri_rect);
        GP_TRY(launch_gemm(h, false, true, EPI_STORE, g));
    }
    {   // P = W W^T (lower) -> h->B, then Cc = I + P/s.  k-chunk c of model b = "problem" b S + c of the launch: the
        // operands advance by one chunk of columns (model b starts at column b nq = b S chunks)
        const int S = w.ksplit;
        GemmArgs g = vgb(B, w.Wm, ldw, (nq / NB / S) * NB, w.Wm, ldw, (nq / NB / S) * NB, S > 1 ? w.Pp : h->B, mp, mm,
                         w.tiles + w.off_syrk, w.n_syrk);
        h->nbatch = B * S;
        const int rc = launch_gemm(h, false, false, EPI_STORE, g);
        h->nbatch = B;
        GP_TRY(rc);
        hipLaunchKernelGGL(vfe_cap_kernel, dim3((unsigned)((mm + 255) / 256), B), dim3(256), 0, h->stream, h->B, w.Cs,
                           mp, h->theta, (const double*)w.Pp, S);
    }
    GP_TRY(launch_potrf_inv(h, h->B, h->Tm, mp, mp, h->info, 0));   // h->B = Lc^-1, logdet_part <- log diag Lc
    // v = W y / s ; c1 = Lc^-1 v
    hipLaunchKernelGGL(gemv_n_kernel, dim3((unsigned)((mp + 3) / 4), B), dim3(256), 0, h->stream, w.Wm, ldw, mp, nq, w.yq, w.v,
                       nq, nq, mp);
    hipLaunchKernelGGL(vfe_scale_kernel, dim3((unsigned)((mp + 255) / 256), B), dim3(256), 0, h->stream, w.v, mp, h->theta);
    GP_TRY(launch_trmv_lower(h, h->B, mp, mp, w.v, w.c1));
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

struct VfeIter { int32_t* iter; const double* bc; int T; double* hist_theta; double* hist_xu; double* loss; };

static int vfe_loss_grad(gpimhip_ctx* h, VfeWs& w, const gpimhip_model_t* m, const VfeProb& q, double* u, int do_adam,
                         const VfeIter* it, double* loss_out, double* grad_out) {
    const int64_t mp = w.mp, nq = w.nq, mm = mp * mp;
    const int mb = (int)(mp / NB), nbq = (int)(nq / NB);
    const int B = w.B, P = q.P;
    const int64_t ldw = (int64_t)B * nq;
    const double* Xu = u + P;
    GP_TRY(vfe_forward(h, w, m, q, u));
    GP_TRY(launch_lauum(h, h->B, w.Vc, mp, mp, 0));              // Cc^-1 (lower)
    GP_TRY(launch_gemv_t(h, h->B, mp, mp, mp, w.c1, w.beta, 1, mm, mp, mp));           // beta = Lc^-T c1
    GP_TRY(launch_gemv_t(h, w.Wm, ldw, mp, nq, w.beta, w.wtb, 0, nq, mp, nq));         // W^T beta
    // Mm (for G_A') and Cc^-1 - I as a full matrix (in T1, which G_A' overwrites below)
    hipLaunchKernelGGL(vfe_mmat_kernel, dim3((unsigned)((mm + 255) / 256), B), dim3(256), 0, h->stream, w.Vc, w.Cs,
                       w.beta, w.Mm, w.T1, mp);
    {   // G_B' = Luu^-T (Cc^-1 W - W - beta r^T) = Q W - bt r^T,  Q = Luu^-T (Cc^-1 - I) (Mu x Mu, in GA until G_A' needs
        // it), bt = Luu^-T beta: ONE Mu x Mu x N product for what were three (Lc^-1 W, Lc^-T (.), Luu^-T (.)) -- 1250 instead
        // of 2250 tile k-blocks per model at config C5's size
        GemmArgs gq = vgb(B, h->A, mp, mm, w.T1, mp, mm, w.GA, mp, mm, w.tiles + w.off_sq_rowge, w.n_sq_rowge);
        GP_TRY(launch_gemm(h, true, true, EPI_STORE, gq));
        GP_TRY(launch_gemv_t(h, h->A, mp, mp, mp, w.beta, w.bt, 1, mm, mp, mp));
        GemmArgs g = vgb(B, w.GA, mp, mm, w.Wm, ldw, nq, w.Y1, ldw, nq, w.tiles + w.off_full_rect, w.n_full_rect);
        GP_TRY(launch_gemm(h, false, true, EPI_STORE, g));
    }
    hipLaunchKernelGGL(vfe_gw_kernel, dim3((unsigned)((mp * nq + 255) / 256), B), dim3(256), 0, h->stream, w.Y1, w.bt, w.yq,
                       w.wtb, mp, nq, ldw);
    {   // G_A' = Luu^-T Mm Luu^-1 (full)
        GemmArgs g1 = vgb(B, w.Mm, mp, mm, h->A, mp, mm, w.T1, mp, mm, w.tiles + w.off_sq_colge, w.n_sq_colge);
        GP_TRY(launch_gemm(h, false, true, EPI_STORE, g1));
        GemmArgs g2 = vgb(B, h->A, mp, mm, w.T1, mp, mm, w.GA, mp, mm, w.tiles + w.off_sq_rowge, w.n_sq_rowge);
        GP_TRY(launch_gemm(h, true, true, EPI_STORE, g2));
    }
    GP_TRY(launch_grad(h, m, w.Y1, ldw, nq, Xu, q.ustride, q.Mu, q.X, q.x_bs, q.N, mb, nbq, w.part_rect, w.xu_rect, mp, B));
    GP_TRY(launch_grad(h, m, w.GA, mp, mm, Xu, q.ustride, q.Mu, Xu, q.ustride, q.Mu, mb, mb, w.part_sym, w.xu_sym, mp, B));
    const int64_t na = P + mp * GPIMHIP_MAX_DIM;
    VfeFinalArgs a;
    memset(&a, 0, sizeof(a));
    a.m = *m; a.N = q.N; a.Mu = q.Mu; a.mp = mp; a.nq = nq; a.mb = mb; a.nbq = nbq;
    a.part_rect = w.part_rect; a.part_sym = w.part_sym; a.xu_rect = w.xu_rect; a.xu_sym = w.xu_sym;
    a.yq = w.yq; a.wtb = w.wtb; a.v = w.v; a.c1 = w.c1; a.beta = w.beta; a.Cs = w.Cs; a.Vc = w.Vc;
    a.logdet_part = h->logdet_part; a.th = h->theta; a.u = u; a.adam_m = w.adam_m; a.adam_v = w.adam_v;
    a.do_adam = do_adam;
    a.info = h->info;
    a.ustride = q.ustride; a.na = na;
    if (it) { a.iter = it->iter; a.bc = it->bc; a.T = it->T; a.hist_theta = it->hist_theta; a.hist_xu = it->hist_xu; a.loss_out = it->loss; }
    else { a.loss_out = loss_out; a.grad_out = grad_out; }
    VfeXuArgs x;
    memset(&x, 0, sizeof(x));
    x.d = m->dim; x.P = P; x.mb = mb; x.nbq = nbq; x.Mu = q.Mu; x.mp = mp;
    x.xu_rect = w.xu_rect; x.xu_sym = w.xu_sym; x.th = h->theta; x.u = u; x.adam_m = w.adam_m; x.adam_v = w.adam_v;
    x.do_adam = do_adam; x.iter = a.iter; x.bc = a.bc; x.T = a.T; x.hist_xu = a.hist_xu; x.grad_out = a.grad_out;
    x.info = h->info;
    x.ustride = q.ustride; x.na = na;
    hipLaunchKernelGGL(vfe_xu_kernel, dim3((unsigned)((q.Mu * m->dim + 255) / 256), B), dim3(256), 0, h->stream, x);
    hipLaunchKernelGGL(vfe_finalize_kernel, dim3(B), dim3(256), 0, h->stream, a);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

int check_model(const gpimhip_model_t* m);

// workspace for B models, h->nbatch = B until the call returns (VfeBatchScope), y (B x N) padded into w.yq
struct VfeBatchScope {
    gpimhip_ctx* h;
    explicit VfeBatchScope(gpimhip_ctx* h_) : h(h_) {}
    ~VfeBatchScope() { h->nbatch = 1; }
};
static int vfe_prepare(gpimhip_ctx* h, const gpimhip_model_t* m, const double* y, int64_t N, int64_t Mu, int P, int B,
                       VfeWs** w) {
    GP_TRY(check_model(m));
    if (Mu > N) {
        gpim_set_error("sparse GP: more inducing inputs than observations");
        return GPIMHIP_E_BADARG;
    }
    HIP_TRY(hipSetDevice(h->device));
    h->nbatch = B;
    GP_TRY(ws_ensure(h, Mu));                                  // mp x mp blocked-algorithm workspace
    GP_TRY(vfe_ensure(h, Mu, N, m->dim, P, B, w));
    HIP_TRY(hipMemsetAsync(h->info, 0, sizeof(int32_t), h->stream));
    GP_TRY(launch_pad_copy(h, y, N, (*w)->yq, (*w)->nq));
    return GPIMHIP_OK;
}

static int fit_vfe_impl(gpimhip_ctx* h, const gpimhip_model_t* m, const double* X, int64_t x_bs, const double* y, int64_t N,
                        int64_t Mu, int B, double* u_inout, double lr, int32_t T, double* hist_theta, double* hist_xu,
                        double* loss_out) {
    const int P = 2 + m->n_ls + (m->kernel == GPIMHIP_KERNEL_RQ ? 1 : 0);
    VfeWs* w;
    VfeBatchScope scope(h);
    GP_TRY(vfe_prepare(h, m, y, N, Mu, P, B, &w));
    const VfeProb q{X, x_bs, N, Mu, P, P + Mu * m->dim};
    HIP_TRY(hipMemsetAsync(h->info + 1, 0x7f, sizeof(int32_t), h->stream));
    h->fit_completed = T;
    if (T == 0) return GPIMHIP_OK;
    const int64_t na = (int64_t)B * (P + w->mp * GPIMHIP_MAX_DIM);
    HIP_TRY(hipMemsetAsync(w->adam_m, 0, na * sizeof(double), h->stream));
    HIP_TRY(hipMemsetAsync(w->adam_v, 0, na * sizeof(double), h->stream));
    HIP_TRY(hipMemsetAsync(h->iter, 0, B * sizeof(int32_t), h->stream));
    // Adam bias-correction table (same libm pow() values as every other path)
    if (h->bc_cap < 2 * (int64_t)T) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (h->bc) { (void)hipFree(h->bc); h->bytes -= h->bc_cap * (int64_t)sizeof(double); h->bc = nullptr; }
        void* qq = nullptr;
        if (hipMalloc(&qq, 2 * (size_t)T * sizeof(double)) != hipSuccess) return GPIMHIP_E_NOMEM;
        h->bc = (double*)qq;
        h->bc_cap = 2 * (int64_t)T;
        h->bytes += h->bc_cap * (int64_t)sizeof(double);
    }
    h->bc_host.resize(2 * (size_t)T);
    for (int t = 1; t <= T; ++t) {
        h->bc_host[t - 1] = lr / (1.0 - pow(0.9, (double)t));
        h->bc_host[T + t - 1] = sqrt(1.0 - pow(0.999, (double)t));
    }
    HIP_TRY(hipMemcpyAsync(h->bc, h->bc_host.data(), 2 * (size_t)T * sizeof(double), hipMemcpyHostToDevice, h->stream));
    VfeIter it{h->iter, h->bc, T, hist_theta, hist_xu, loss_out};
    // every iteration is the same ~90 launches (iteration index and bias corrections live on the
    // device): capture one into a hipGraph and replay it
    if (T >= 8 && !getenv("GPIMHIP_NO_GRAPH") && ensure_capture_stream(h)) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        hipStream_t main_s = h->stream;
        h->stream = h->capture_stream;
        capture_lock(h);
        hipError_t e = hipStreamBeginCapture(h->capture_stream, hipStreamCaptureModeRelaxed);
        int rc = GPIMHIP_OK;
        if (e == hipSuccess) {
            rc = vfe_loss_grad(h, *w, m, q, u_inout, 1, &it, nullptr, nullptr);
            e = hipStreamEndCapture(h->capture_stream, &graph);
        }
        capture_unlock(h);
        h->stream = main_s;
        if (rc != GPIMHIP_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
            for (int t = 0; t < T; ++t) HIP_TRY(hipGraphLaunch(exec, main_s));
            rc = vfe_finish_and_check(h);
            (void)hipGraphExecDestroy(exec);
            (void)hipGraphDestroy(graph);
            return rc;
        }
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
    }
    for (int t = 0; t < T; ++t) GP_TRY(vfe_loss_grad(h, *w, m, q, u_inout, 1, &it, nullptr, nullptr));
    return vfe_finish_and_check(h);
}

static int predict_vfe_impl(gpimhip_ctx* h, const gpimhip_model_t* m, const double* X, int64_t x_bs, const double* y, int64_t N,
                            int64_t Mu, int B, const double* u, const double* Xs, int64_t xs_bs, int64_t M, double* mean_out,
                            double* var_out) {
    const int P = 2 + m->n_ls + (m->kernel == GPIMHIP_KERNEL_RQ ? 1 : 0);
    VfeWs* wp;
    VfeBatchScope scope(h);
    GP_TRY(vfe_prepare(h, m, y, N, Mu, P, B, &wp));
    VfeWs& w = *wp;
    const VfeProb q{X, x_bs, N, Mu, P, P + Mu * m->dim};
    const int64_t mp = w.mp, mm = mp * mp;
    const int mb = (int)(mp / NB);
    GP_TRY(vfe_forward(h, w, m, q, u));
    // test points in slabs of mc columns per model (the slabs of the B models side by side, like the Mu x N matrices)
    int64_t mc = std::min<int64_t>(pad_to(M, NB), std::max<int64_t>(NB, ((int64_t)1 << 26) / mp / B / NB * NB));
    if (w.mc != mc) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (w.Ks) (void)hipFree(w.Ks);
        if (w.Ws) (void)hipFree(w.Ws);
        if (w.LW) (void)hipFree(w.LW);
        if (w.ptiles) (void)hipFree(w.ptiles);
        w.Ks = w.Ws = w.LW = nullptr; w.ptiles = nullptr; w.mc = 0;
        GP_TRY(valloc(&w.Ks, B * mp * mc)); GP_TRY(valloc(&w.Ws, B * mp * mc)); GP_TRY(valloc(&w.LW, B * mp * mc));
        std::vector<TileDesc> tl;
        for (int ci = mb - 1; ci >= 0; --ci)
            for (int cj = 0; cj < (int)(mc / NB); ++cj) tl.push_back({ci, cj, 0, ci + 1});
        w.n_ptiles = (int)tl.size();
        GP_TRY(valloc(&w.ptiles, (int64_t)tl.size()));
        HIP_TRY(hipMemcpyAsync(w.ptiles, tl.data(), tl.size() * sizeof(TileDesc), hipMemcpyHostToDevice, h->stream));
        HIP_TRY(hipStreamSynchronize(h->stream));
        w.mc = mc;
    }
    const double* Xu = u + P;
    const int64_t ldk = (int64_t)B * mc;
    for (int64_t m0 = 0; m0 < M; m0 += mc) {
        const int64_t cnt = std::min(mc, M - m0), cpad = pad_to(cnt, NB);
        GP_TRY(launch_kmat(h, m, Xu, Mu, Xs + m0 * m->dim, cnt, h->theta, 0.0, 0, w.Ks, ldk, mp, cpad, 0, 0, q.ustride, xs_bs, mc));
        GemmArgs g1 = vgb(B, h->A, mp, mm, w.Ks, ldk, mc, w.Ws, ldk, mc, w.ptiles, w.n_ptiles);     // Ws = Luu^-1 Kus
        GP_TRY(launch_gemm(h, false, true, EPI_STORE, g1));
        GemmArgs g2 = vgb(B, h->B, mp, mm, w.Ws, ldk, mc, w.LW, ldk, mc, w.ptiles, w.n_ptiles);     // Lc^-1 Ws
        GP_TRY(launch_gemm(h, false, true, EPI_STORE, g2));
        hipLaunchKernelGGL(vfe_predict_cols_kernel, dim3((unsigned)((cnt + 255) / 256), B), dim3(256), 0, h->stream, w.Ws,
                           w.LW, ldk, w.c1, mp, cnt, h->theta, mean_out + m0, var_out + m0, mc, M);
    }
    HIP_TRY(hipGetLastError());
    return vfe_finish_and_check(h);
}

extern "C" {

int gpimhip_vfe_nll_grad(gpimhip_handle h, const gpimhip_model_t* m, const double* X, const double* y, int64_t N,
                         int64_t Mu, const double* u, double* loss_out, double* grad_out) {
    FP64_ONLY(h);
    if (!h || !m || !X || !y || !u || N < 1 || Mu < 1) return GPIMHIP_E_BADARG;
    const int P = 2 + m->n_ls + (m->kernel == GPIMHIP_KERNEL_RQ ? 1 : 0);
    VfeWs* w;
    VfeBatchScope scope(h);
    GP_TRY(vfe_prepare(h, m, y, N, Mu, P, 1, &w));
    const VfeProb q{X, 0, N, Mu, P, P + Mu * m->dim};
    GP_TRY(vfe_loss_grad(h, *w, m, q, const_cast<double*>(u), 0, nullptr, loss_out, grad_out));
    return vfe_finish_and_check(h);
}

int gpimhip_fit_vfe(gpimhip_handle h, const gpimhip_model_t* m, const double* X, const double* y, int64_t N,
                    int64_t Mu, double* u_inout, double lr, int32_t T, double* hist_theta, double* hist_xu,
                    double* loss_out) {
    FP64_ONLY(h);
    if (!h || !m || !X || !y || !u_inout || N < 1 || Mu < 1 || T < 0) return GPIMHIP_E_BADARG;
    return fit_vfe_impl(h, m, X, 0, y, N, Mu, 1, u_inout, lr, T, hist_theta, hist_xu, loss_out);
}

int gpimhip_fit_vfe_batched(gpimhip_handle h, const gpimhip_model_t* m, const double* X, int64_t x_stride, const double* y,
                            int64_t N, int64_t Mu, int32_t B, double* u_inout, double lr, int32_t T, double* hist_theta,
                            double* hist_xu, double* loss_out) {
    FP64_ONLY(h);
    if (!h || !m || !X || !y || !u_inout || N < 1 || Mu < 1 || T < 0 || B < 1 || B > 4096 || x_stride < 0) return GPIMHIP_E_BADARG;
    return fit_vfe_impl(h, m, X, x_stride, y, N, Mu, B, u_inout, lr, T, hist_theta, hist_xu, loss_out);
}

int gpimhip_predict_vfe(gpimhip_handle h, const gpimhip_model_t* m, const double* X, const double* y, int64_t N,
                        int64_t Mu, const double* u, const double* Xs, int64_t M, double* mean_out,
                        double* var_out) {
    FP64_ONLY(h);
    if (!h || !m || !X || !y || !u || !Xs || N < 1 || Mu < 1 || M < 1 || !mean_out || !var_out) return GPIMHIP_E_BADARG;
    return predict_vfe_impl(h, m, X, 0, y, N, Mu, 1, u, Xs, 0, M, mean_out, var_out);
}

int gpimhip_predict_vfe_batched(gpimhip_handle h, const gpimhip_model_t* m, const double* X, int64_t x_stride, const double* y,
                                int64_t N, int64_t Mu, int32_t B, const double* u, const double* Xs, int64_t xs_stride,
                                int64_t M, double* mean_out, double* var_out) {
    FP64_ONLY(h);
    if (!h || !m || !X || !y || !u || !Xs || N < 1 || Mu < 1 || M < 1 || !mean_out || !var_out || B < 1 || B > 4096 ||
        x_stride < 0 || xs_stride < 0)
        return GPIMHIP_E_BADARG;
    return predict_vfe_impl(h, m, X, x_stride, y, N, Mu, B, u, Xs, xs_stride, M, mean_out, var_out);
}

}  // extern "C"
