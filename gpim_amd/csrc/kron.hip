// kron.hip -- exact GP on a FULLY OBSERVED regular grid through the Kronecker structure of the
// covariance (the role of the reference's structured-kernel reconstructor,
// gpim/gpreg/skgpr.py:399-448 -- there GPyTorch's interpolated SKI approximation, here exact).
//
// For the ARD (or isotropic) RBF kernel on a product grid X = c_1 x c_2 x ... x c_d
//     K = s2 * K_1 (x) K_2 (x) ... (x) K_d,     K_i[a,b] = exp(-((c_i[a] - c_i[b]) / l_i)^2 / 2),
// so with K_i = Q_i diag(lam_i) Q_i^T (n_i x n_i symmetric eigenproblems)
//     K + (jitter + noise) I = Q diag(D) Q^T,   Q = (x) Q_i,   D = s2 * prod_i lam_i[j_i] + jitter + noise.
// Everything the dense path gets from the N x N Cholesky factor and K^-1 -- the loss, the SAME reduced
// gradient sums S[] that csrc/engine.hip:grad_reduce_kernel produces (so the chain rule, the Adam step and
// the history row are the shared finalize_step of theta.hpp), the posterior mean and variance -- follows
// from mode products with the small matrices: O(sum n_i^3 + N sum n_i) instead of O(N^3) work and
// O(N + sum n_i^2) instead of O(N^2) memory.  Same model, same parameterisation, same results as
// gpimhip_fit_exact / gpimhip_predict_exact up to rounding (tests/test_gpu_kron.py checks against the
// dense oracle).
//
// Kernels (all HBM/latency-bound; the work is tiny next to the dense path's):
//   kron_axis_kernel     K_i, E_i = K_i o ((c_a - c_b)/l_i)^2 (= l_i dK_i/dl_i), identity for the rotations
//   kron_reduce_kernel   axes with centro-symmetric coordinates (uniform grids): K_i splits into a symmetric and a
//                        skew-symmetric eigen-problem of half the order
//   kron_eigh_kernel     one-sided (Hestenes) Jacobi, one 16-wave workgroup per eigen-problem, round-robin ordering
//   kron_assemble_kernel eigenvectors of K_i from the rotations of its eigen-problems
//   kron_modeprod_kernel out[p,a,q] = sum_b M[a,b] in[p,b,q]   (tensor-times-matrix along one mode)
//   kron_dvec_kernel     D, alpha~ = y~ / D, first reduction pass;  kron_quad_kernel second pass
//   kron_finalize_kernel sums -> finalize_step;  kron_cross_kernel / kron_var_kernel for the posterior
#include <math.h>
#include <string.h>
#include <algorithm>
#include "kfun.hpp"
#include "theta.hpp"
#include "blocklds.hpp"      // wave_sum: DPP / readlane reduction (no LDS crossbar round trips)

hipStream_t ensure_capture_stream(gpimhip_ctx* h);
void capture_lock(gpimhip_ctx* h);
void capture_unlock(gpimhip_ctx* h);
int ws_ensure(gpimhip_ctx* h, int64_t N);
int check_model(const gpimhip_model_t* m);
int upload_bc_table(gpimhip_ctx* h, double lr, int T);
int finish_and_check(gpimhip_ctx* h);

#define KMAXD GPIMHIP_MAX_DIM

struct KronDev {
    int d;
    int n[KMAXD];            // training grid
    int off[KMAXD];          // offset of axis i in the per-axis vectors (coords, lam, mdiag)
    int64_t moff[KMAXD];     // offset of axis i's n_i x n_i matrices
    int64_t N;
    const double* coords;    // [sum n_i]
    double *K, *E, *W, *Qt, *Mm, *Tt, *Vr;   // [sum n_i^2] each
    double *lam, *mdiag;     // [sum n_i]
    // eigen-problems: one per axis, or two (symmetric / skew-symmetric half) when the axis' coordinates
    // are centro-symmetric (any uniform grid): K_i then commutes with the exchange matrix and splits into
    // two problems of half the size -- an eighth of the Jacobi work each, half the round-robin steps
    int nprob;
    int pax[2 * KMAXD], ppart[2 * KMAXD], pn[2 * KMAXD];   // axis, part (0 whole, 1 symmetric, 2 skew), order
    int64_t poff[2 * KMAXD];                                // offset inside the axis' n_i^2 region (W, Vr, Tt)
    int plam[2 * KMAXD];                                    // offset of its eigenvalues / Qt rows inside the axis
};

struct KronWs {
    int d = 0;
    int n[KMAXD] = {0, 0, 0, 0};
    int64_t N = 0, sum_n = 0, sum_n2 = 0;
    double* arena = nullptr;
    int64_t arena_count = 0;
    KronDev dev;
    double *yt = nullptr, *at = nullptr, *t1 = nullptr, *t2 = nullptr, *Z = nullptr, *part = nullptr;
    int nblk = 0;
    // prediction
    int m[KMAXD] = {0, 0, 0, 0};
    int64_t M = 0, pmax = 0;
    double* parena = nullptr;
    int64_t parena_count = 0;
    double *Ks = nullptr, *Bs = nullptr, *p1 = nullptr, *p2 = nullptr, *tcoords = nullptr;
    int64_t koff[KMAXD] = {0, 0, 0, 0};
};

void kron_release(gpimhip_ctx* h) {
    KronWs* w = static_cast<KronWs*>(h->kron);
    if (!w) return;
    if (w->arena) { (void)hipFree(w->arena); h->bytes -= w->arena_count * (int64_t)sizeof(double); }
    if (w->parena) { (void)hipFree(w->parena); h->bytes -= w->parena_count * (int64_t)sizeof(double); }
    delete w;
    h->kron = nullptr;
}

static int kron_ensure(gpimhip_ctx* h, int d, const int32_t* n, KronWs** out) {
    if (!h->kron) h->kron = new KronWs();
    KronWs& w = *static_cast<KronWs*>(h->kron);
    *out = &w;
    bool same = w.d == d && w.arena;
    for (int i = 0; i < d && same; ++i) same = w.n[i] == n[i];
    if (same) return GPIMHIP_OK;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (w.arena) { (void)hipFree(w.arena); h->bytes -= w.arena_count * (int64_t)sizeof(double); w.arena = nullptr; }
    w.d = d;
    w.N = 1; w.sum_n = 0; w.sum_n2 = 0;
    for (int i = 0; i < KMAXD; ++i) w.n[i] = i < d ? n[i] : 1;
    for (int i = 0; i < d; ++i) { w.N *= n[i]; w.sum_n += n[i]; w.sum_n2 += (int64_t)n[i] * n[i]; }
    w.nblk = (int)std::min<int64_t>(1024, (w.N + 255) / 256);
    const int64_t count = 3 * w.sum_n + 7 * w.sum_n2 + (4 + d) * w.N + (int64_t)w.nblk * 16 + 64;
    void* q = nullptr;
    if (hipMalloc(&q, (size_t)count * sizeof(double)) != hipSuccess) {
        gpim_set_error("hipMalloc failed (structured-GP workspace)");
        return GPIMHIP_E_NOMEM;
    }
    w.arena = (double*)q;
    w.arena_count = count;
    h->bytes += count * (int64_t)sizeof(double);
    double* p = w.arena;
    auto take = [&](int64_t c) { double* r = p; p += (c + 1) / 2 * 2; return r; };    // keep 16-byte alignment
    KronDev& dv = w.dev;
    memset(&dv, 0, sizeof(dv));
    dv.d = d; dv.N = w.N;
    int off = 0; int64_t moff = 0;
    for (int i = 0; i < KMAXD; ++i) {
        dv.n[i] = w.n[i]; dv.off[i] = off; dv.moff[i] = moff;
        if (i < d) { off += n[i]; moff += (int64_t)n[i] * n[i]; }
    }
    dv.coords = take(w.sum_n); dv.lam = take(w.sum_n); dv.mdiag = take(w.sum_n);
    dv.K = take(w.sum_n2); dv.E = take(w.sum_n2); dv.W = take(w.sum_n2);
    dv.Qt = take(w.sum_n2); dv.Mm = take(w.sum_n2); dv.Tt = take(w.sum_n2); dv.Vr = take(w.sum_n2);
    w.yt = take(w.N); w.at = take(w.N); w.t1 = take(w.N); w.t2 = take(w.N);
    w.Z = take((int64_t)d * w.N);
    w.part = take((int64_t)w.nblk * 16);
    return GPIMHIP_OK;
}

// ------------------------------------------------------------------------------------------
// per-axis matrices
// ------------------------------------------------------------------------------------------
__global__ void kron_axis_kernel(KronDev dv, const ThetaDev* __restrict__ th) {
    const int ax = blockIdx.y;
    const int n = dv.n[ax];
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)n * n) return;
    const int a = (int)(e / n), b = (int)(e % n);
    const double* c = dv.coords + dv.off[ax];
    const double il = 1.0 / th->ls[ax];
    const double dl = c[a] * il - c[b] * il;
    const double r2 = dl * dl;
    const double k = exp(-0.5 * r2);
    const int64_t o = dv.moff[ax] + e;
    dv.K[o] = k;
    dv.E[o] = k * r2;
}

// Vr <- I for every eigen-problem (cold start of the rotations)
__global__ void kron_eye_kernel(KronDev dv) {
    const int pr = blockIdx.y;
    const int n = dv.pn[pr];
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)n * n) return;
    dv.Vr[dv.moff[dv.pax[pr]] + dv.poff[pr] + e] = (e / n == e % n) ? 1.0 : 0.0;
}

// The matrix of each eigen-problem (into Tt): K_i itself, or its restriction to the symmetric / skew-symmetric
// vectors.  With n = 2m (+1), basis (e_a +- e_{n-1-a}) / sqrt2 (and e_m):
//   symmetric  S[a][b] = K[a][b] + K[a][n-1-b],  S[a][m] = S[m][a] = sqrt2 K[a][m],  S[m][m] = K[m][m]
//   skew       D[a][b] = K[a][b] - K[a][n-1-b]
__global__ void kron_reduce_kernel(KronDev dv) {
    const int pr = blockIdx.y;
    const int ax = dv.pax[pr], part = dv.ppart[pr], np_ = dv.pn[pr], n = dv.n[ax], m = n / 2;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)np_ * np_) return;
    const int a = (int)(e / np_), b = (int)(e % np_);
    const double* K = dv.K + dv.moff[ax];
    double v;
    if (part == 0) v = K[(int64_t)a * n + b];
    else if (part == 2) v = K[(int64_t)a * n + b] - K[(int64_t)a * n + (n - 1 - b)];
    else if (a < m && b < m) v = K[(int64_t)a * n + b] + K[(int64_t)a * n + (n - 1 - b)];
    else if (a == m && b == m) v = K[(int64_t)m * n + m];
    else v = 1.4142135623730951 * K[(int64_t)(a < m ? a : b) * n + m];
    dv.Tt[dv.moff[ax] + dv.poff[pr] + e] = v;
}

// Qt rows (eigenvectors of K_i, length n) from the rotations of the eigen-problems
__global__ void kron_assemble_kernel(KronDev dv) {
    const int pr = blockIdx.y;
    const int ax = dv.pax[pr], part = dv.ppart[pr], np_ = dv.pn[pr], n = dv.n[ax], m = n / 2;
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)np_ * n) return;
    const int j = (int)(e / n), a = (int)(e % n);
    const double* V = dv.Vr + dv.moff[ax] + dv.poff[pr] + (int64_t)j * np_;
    double v;
    if (part == 0) v = V[a];
    else {
        const int lo = a < m ? a : n - 1 - a;            // mirrored index; a == m only when n is odd
        if ((n & 1) && a == m) v = (part == 1) ? V[m] : 0.0;
        else v = 0.7071067811865476 * V[lo] * ((part == 2 && a >= n - m) ? -1.0 : 1.0);
    }
    dv.Qt[dv.moff[ax] + (int64_t)(dv.plam[pr] + j) * n + a] = v;
}

__device__ __forceinline__ double kwave_sum(double v) { return wave_sum(v); }

// One-sided Jacobi on the rows of the symmetric positive semi-definite W (= its columns): plane rotations
// make the rows mutually orthogonal, W = V K; for symmetric K the accumulated V holds the eigenvectors
// (row j = eigenvector j).  Round-robin (circle) ordering: n/2 disjoint pairs per step, the pairs of a step
// dealt to the 16 waves; per step every wave first forms the rotations of all its pairs (independent
// loads, several pairs in flight), then applies them.
// Stopping rule per pair (i, j) with a = |w_i|^2, b = |w_j|^2, c = w_i . w_j:
//   |c| <= 1e-15 sqrt(a b)                          rows orthogonal to working precision, or
//   |c| <= 2e-15 lmax sqrt(a + b)                   the coupling (V K V^T)_ij ~ c / (lam_i + lam_j) is below
//                                                   ~10 eps * lmax: RBF Gram matrices have most of their
//                                                   spectrum below eps * lmax, and rows that are pure
//                                                   rounding noise never get orthogonal in the relative sense
// Eigenvalue j = (row j of W) . (row j of V) -- keeps the sign of numerically negative ones.
#define EIGH_WAVES 16

// 1/x and 1/sqrt(x) from the hardware seeds (v_rcp_f64 / v_rsq_f64) + two Newton steps: a tenth of the
// instructions of an IEEE divide / square root.  tools/jacobi_prof.hip: with 16 waves on 4 SIMDs the
// rotation arithmetic -- 3 divides and 4 square roots per pair, ~1500 issue cycles -- was what a step waited
// for, not the row loads.
__device__ __forceinline__ double fast_rcp(double x) {
    double y = __builtin_amdgcn_rcp(x);
    y = fma(y, fma(-x, y, 1.0), y);
    return fma(y, fma(-x, y, 1.0), y);
}
__device__ __forceinline__ double fast_rsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * fma(-0.5 * x * y, y, 1.5);
    return y * fma(-0.5 * x * y, y, 1.5);
}

// rotation that makes rows i, j orthogonal (a = |w_i|^2, b = |w_j|^2, c = w_i . w_j), or false when the pair
// passes the stopping rule above.  With d = b - a, e = 2c:  tan = sign(d e) |e| / (|d| + sqrt(d^2 + e^2)).
// The tests are on squares ((sqrt a + sqrt b)^2 <= 2 (a + b)), so nothing here divides or takes a root the
// slow way; cos = 1/sqrt(1 + tan^2) is accurate to rounding, which is what keeps V orthogonal.
__device__ __forceinline__ bool jacobi_rotation(double a, double b, double c, double lmax, double& cs, double& sn) {
    const double c2 = c * c;
    if (!(c2 > 1e-30 * (a * b) && c2 > (4e-30 * lmax * lmax) * (a + b))) return false;
    const double d = b - a, e = 2.0 * c;
    const double h2 = fma(d, d, e * e);
    const double h = h2 * fast_rsqrt(h2);
    double t = fabs(e) * fast_rcp(fabs(d) + h);
    t = ((d >= 0.0) == (e >= 0.0)) ? t : -t;
    cs = fast_rsqrt(fma(t, t, 1.0));
    sn = cs * t;
    return true;
}

// One step of the round-robin ordering for rows of at most 64 * NPL elements: every wave takes G of its
// pairs at a time, with the four rows of each pair (w_i, w_j, v_i, v_j) held in registers -- one memory
// round trip per G pairs instead of two per pair (the kernel is bound by the latency of these trips).
template <int NPL, int G>
__device__ __forceinline__ double jacobi_step_regs(double* W, double* V, int n, int half, int ring, int step, int wave,
                                                   int lane, double lmax) {
    double any = 0.0;          // largest |sin| applied
    for (int k0 = wave; k0 < half; k0 += EIGH_WAVES * G) {
        double wi[G][NPL], wj[G][NPL], vi[G][NPL], vj[G][NPL];
        int ri[G], rj[G];
        double a[G], b[G], c[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int k = k0 + g * EIGH_WAVES;
            const int i = (k == 0) ? ring : (step + k) % ring;
            const int j = (step + ring - k) % ring;
            const bool ok = k < half && i < n && j < n;
            ri[g] = ok ? i : -1;
            rj[g] = j;
            a[g] = b[g] = c[g] = 0.0;
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const int e = lane + 64 * q;
                const bool in = ok && e < n;
                wi[g][q] = in ? W[(int64_t)i * n + e] : 0.0;
                wj[g][q] = in ? W[(int64_t)j * n + e] : 0.0;
                vi[g][q] = in ? V[(int64_t)i * n + e] : 0.0;
                vj[g][q] = in ? V[(int64_t)j * n + e] : 0.0;
            }
        }
#pragma unroll
        for (int g = 0; g < G; ++g)
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                a[g] = fma(wi[g][q], wi[g][q], a[g]);
                b[g] = fma(wj[g][q], wj[g][q], b[g]);
                c[g] = fma(wi[g][q], wj[g][q], c[g]);
            }
#pragma unroll
        for (int g = 0; g < G; ++g) { a[g] = kwave_sum(a[g]); b[g] = kwave_sum(b[g]); c[g] = kwave_sum(c[g]); }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            double cs, sn;
            if (ri[g] < 0 || !jacobi_rotation(a[g], b[g], c[g], lmax, cs, sn)) continue;
            any = fmax(any, fabs(sn));
            const int i = ri[g], j = rj[g];
#pragma unroll
            for (int q = 0; q < NPL; ++q) {
                const int e = lane + 64 * q;
                if (e < n) {
                    W[(int64_t)i * n + e] = cs * wi[g][q] - sn * wj[g][q];
                    W[(int64_t)j * n + e] = sn * wi[g][q] + cs * wj[g][q];
                    V[(int64_t)i * n + e] = cs * vi[g][q] - sn * vj[g][q];
                    V[(int64_t)j * n + e] = sn * vi[g][q] + cs * vj[g][q];
                }
            }
        }
    }
    return any;
}

// any row length: rows streamed from memory twice (dot products, then the update)
__device__ __forceinline__ double jacobi_step_stream(double* W, double* V, int n, int half, int ring, int step,
                                                     int wave, int lane, double lmax) {
    double any = 0.0;
    for (int k = wave; k < half; k += EIGH_WAVES) {
        const int i = (k == 0) ? ring : (step + k) % ring;
        const int j = (step + ring - k) % ring;
        if (i >= n || j >= n) continue;
        double* wi = W + (int64_t)i * n;
        double* wj = W + (int64_t)j * n;
        double a = 0.0, b = 0.0, c = 0.0;
        for (int e = lane; e < n; e += 64) {
            const double x = wi[e], y = wj[e];
            a = fma(x, x, a); b = fma(y, y, b); c = fma(x, y, c);
        }
        a = kwave_sum(a); b = kwave_sum(b); c = kwave_sum(c);
        double cs, sn;
        if (!jacobi_rotation(a, b, c, lmax, cs, sn)) continue;
        any = fmax(any, fabs(sn));
        double* vi = V + (int64_t)i * n;
        double* vj = V + (int64_t)j * n;
        for (int e = lane; e < n; e += 64) {
            const double x = wi[e], y = wj[e], p = vi[e], q = vj[e];
            wi[e] = cs * x - sn * y;
            wj[e] = sn * x + cs * y;
            vi[e] = cs * p - sn * q;
            vj[e] = sn * p + cs * q;
        }
    }
    return any;
}

// NPL = row elements per lane held in registers (rows of at most 64 * NPL entries), G = pairs in flight
// per wave; NPL = 0: rows streamed from memory (any length).  One instantiation per size class, so the
// small ones do not inherit the register footprint of the large ones.
template <int NPL, int G>
__global__ __launch_bounds__(EIGH_WAVES * 64) void kron_eigh_kernel(KronDev dv) {
    __shared__ double s_max[EIGH_WAVES];
    const int pr = blockIdx.x;
    const int ax = dv.pax[pr];
    const int n = dv.pn[pr];
    double* W = dv.W + dv.moff[ax] + dv.poff[pr];
    double* V = dv.Vr + dv.moff[ax] + dv.poff[pr];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n2 = n + (n & 1), half = n2 / 2, ring = n2 - 1;
    // lmax: the largest absolute row sum of K (Gershgorin: an upper bound of the largest eigenvalue, tight
    // within a small factor for these non-negative matrices)
    {
        double mx = 0.0;
        for (int r = wave; r < n; r += EIGH_WAVES) {
            double a = 0.0;
            for (int e = lane; e < n; e += 64) a += fabs(W[(int64_t)r * n + e]);
            mx = fmax(mx, kwave_sum(a));
        }
        if (lane == 0) s_max[wave] = mx;
        __syncthreads();
    }
    double lmax = 0.0;
    for (int w = 0; w < EIGH_WAVES; ++w) lmax = fmax(lmax, s_max[w]);
    // Sweeps until the largest rotation of a sweep is below 1e-8: the method converges quadratically, so
    // what is left after such a sweep is at rounding level (no rotation-free sweep needed to confirm).  The
    // rotations continue from the V the caller left in Vr -- the identity, or the eigenvectors of the
    // previous Adam iteration, whose K differs by a small change of the lengthscale: W = V K then starts
    // almost orthogonal and two or three sweeps are enough instead of 8 ... 18.
    int nsweep = 0;
    for (int sweep = 0; sweep < 40; ++sweep) {
        ++nsweep;
        if (tid < EIGH_WAVES) s_max[tid] = 0.0;
        __syncthreads();
        double big = 0.0;
        for (int step = 0; step < ring; ++step) {
            double r;
            if (NPL > 0) r = jacobi_step_regs<(NPL > 0 ? NPL : 1), G>(W, V, n, half, ring, step, wave, lane, lmax);
            else r = jacobi_step_stream(W, V, n, half, ring, step, wave, lane, lmax);
            big = fmax(big, r);
            __syncthreads();
        }
        if (lane == 0) s_max[wave] = big;
        __syncthreads();
        double all = 0.0;
        for (int w = 0; w < EIGH_WAVES; ++w) all = fmax(all, s_max[w]);
        __syncthreads();
        if (all < 1e-8) break;
    }
#ifdef KRON_DEBUG
    if (tid == 0) printf("eigh axis %d part %d n %d: %d sweeps, lmax %.3f\n", ax, dv.ppart[pr], n, nsweep, lmax);
#endif
    double* lam = dv.lam + dv.off[ax] + dv.plam[pr];
    for (int j = wave; j < n; j += EIGH_WAVES) {
        double s = 0.0;
        for (int e = lane; e < n; e += 64) s = fma(W[(int64_t)j * n + e], V[(int64_t)j * n + e], s);
        s = kwave_sum(s);
        if (lane == 0) lam[j] = s;
    }
}

// ------------------------------------------------------------------------------------------
// tensor-times-matrix along one mode: out[p, a, q] = sum_b Mx(a, b) in[p, b, q]
//   Mx(a, b) = M[a * ldm + b]  (trans == 0)  or  M[b * ldm + a]  (trans != 0);  sq != 0 squares the entries
// ------------------------------------------------------------------------------------------
__global__ void kron_modeprod_kernel(const double* __restrict__ in, double* __restrict__ out,
                                     const double* __restrict__ M, int ma, int nbk, int ldm, int trans, int sq,
                                     int64_t pre, int64_t post) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= pre * ma * post) return;
    const int64_t q = idx % post, a = (idx / post) % ma, p = idx / (post * ma);
    const double* src = in + p * nbk * post + q;
    double s = 0.0;
    for (int b = 0; b < nbk; ++b) {
        double mv = trans ? M[(int64_t)b * ldm + a] : M[(int64_t)a * ldm + b];
        if (sq) mv *= mv;
        s = fma(mv, src[(int64_t)b * post], s);
    }
    out[idx] = s;
}

static int modeprod(gpimhip_ctx* h, const double* in, double* out, const double* M, int ma, int nbk, int ldm,
                    int trans, int sq, int64_t pre, int64_t post) {
    const int64_t total = pre * ma * post;
    hipLaunchKernelGGL(kron_modeprod_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, in, out, M,
                       ma, nbk, ldm, trans, sq, pre, post);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// applies, for every axis i, the matrix mats[i] (ma[i] x nb[i]) along mode i of `src` (shape nb[]);
// ping-pongs between bufa/bufb and returns the buffer that holds the result (shape ma[])
static int tensor_apply(gpimhip_ctx* h, int d, const int* nbv, const int* mav, const double* const* mats,
                        const int* ldm, int trans, int sq, const double* src, double* bufa, double* bufb,
                        double** result) {
    const double* cur = src;
    double* dst = bufa;
    int shape[KMAXD];
    for (int i = 0; i < d; ++i) shape[i] = nbv[i];
    for (int i = 0; i < d; ++i) {
        int64_t pre = 1, post = 1;
        for (int k = 0; k < i; ++k) pre *= shape[k];
        for (int k = i + 1; k < d; ++k) post *= shape[k];
        GP_TRY(modeprod(h, cur, dst, mats[i], mav[i], nbv[i], ldm[i], trans, sq, pre, post));
        shape[i] = mav[i];
        cur = dst;
        dst = (dst == bufa) ? bufb : bufa;
    }
    *result = const_cast<double*>(cur);
    return GPIMHIP_OK;
}

// ------------------------------------------------------------------------------------------
// reductions.  part[blk][0..15]: 0 lg = sum log D / 2, 1 q2 = sum yt^2/D, 2 sum lam/D, 3 sum lam yt^2/D^2,
// 4 sum 1/D, 5 sum yt^2/D^2, 6+i trace part of axis i, 10+i quadratic part of axis i
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void kron_block_reduce(double* vals, int nv, double* part) {
    __shared__ double red[4][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int k = 0; k < nv; ++k) {
        const double v = kwave_sum(vals[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < nv)
        part[(int64_t)blockIdx.x * 16 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void kron_dvec_kernel(KronDev dv, const ThetaDev* __restrict__ th,
                                                        const double* __restrict__ yt, double* __restrict__ at,
                                                        double* __restrict__ part) {
    const ThetaDev t = *th;
    double v[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < dv.N; j += (int64_t)gridDim.x * 256) {
        int idx[KMAXD];
        int64_t r = j;
        for (int i = dv.d - 1; i >= 0; --i) { idx[i] = (int)(r % dv.n[i]); r /= dv.n[i]; }
        double lam = t.var, l[KMAXD];
        for (int i = 0; i < dv.d; ++i) { l[i] = dv.lam[dv.off[i] + idx[i]]; lam *= l[i]; }
        const double D = lam + t.diag_add;
        const double y = yt[j], a = y / D;
        at[j] = a;
        v[0] += 0.5 * log(D);
        v[1] = fma(y, a, v[1]);
        v[2] += lam / D;
        v[3] = fma(lam * a, a, v[3]);
        v[4] += 1.0 / D;
        v[5] = fma(a, a, v[5]);
        for (int i = 0; i < dv.d; ++i) {
            double po = 1.0;
            for (int k = 0; k < dv.d; ++k) if (k != i) po *= l[k];
            v[6 + i] += dv.mdiag[dv.off[i] + idx[i]] * po / D;
        }
    }
    kron_block_reduce(v, 10, part);
}

// quadratic parts: sum_j at_j * Z_i[j] * prod_{k != i} lam_k[j_k]
__global__ __launch_bounds__(256) void kron_quad_kernel(KronDev dv, const double* __restrict__ at,
                                                        const double* __restrict__ Z, double* __restrict__ part) {
    double v[4] = {0, 0, 0, 0};
    for (int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x; j < dv.N; j += (int64_t)gridDim.x * 256) {
        int idx[KMAXD];
        int64_t r = j;
        for (int i = dv.d - 1; i >= 0; --i) { idx[i] = (int)(r % dv.n[i]); r /= dv.n[i]; }
        double l[KMAXD];
        for (int i = 0; i < dv.d; ++i) l[i] = dv.lam[dv.off[i] + idx[i]];
        const double a = at[j];
        for (int i = 0; i < dv.d; ++i) {
            double po = 1.0;
            for (int k = 0; k < dv.d; ++k) if (k != i) po *= l[k];
            v[i] = fma(a * po, Z[(int64_t)i * dv.N + j], v[i]);
        }
    }
    kron_block_reduce(v, 4, part + 10);
}

__global__ void kron_mdiag_kernel(KronDev dv) {
    const int ax = blockIdx.y, n = dv.n[ax];
    const int a = blockIdx.x * blockDim.x + threadIdx.x;
    if (a < n) dv.mdiag[dv.off[ax] + a] = dv.Mm[dv.moff[ax] + (int64_t)a * n + a];
}

struct KronIter { int32_t* iter; const double* bc; int32_t T; double* hist_base; double* loss_base; };

__global__ __launch_bounds__(256) void kron_finalize_kernel(gpimhip_model_t m, KronDev dv, int nblk,
                                                            const double* __restrict__ part,
                                                            const ThetaDev* __restrict__ th, double* __restrict__ u,
                                                            double* __restrict__ adam_m, double* __restrict__ adam_v,
                                                            int do_adam, double* __restrict__ loss_out,
                                                            double* __restrict__ grad_out, KronIter fi,
                                                            int32_t* __restrict__ info) {
    __shared__ double red[256];
    __shared__ double tot[16];
    const int tid = threadIdx.x;
    for (int k = 0; k < 14; ++k) {
        double v = 0.0;
        for (int b = tid; b < nblk; b += 256) v += part[(int64_t)b * 16 + k];
        red[tid] = v;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        if (tid == 0) tot[k] = red[0];
        __syncthreads();
    }
    if (tid != 0) return;
    const ThetaDev t = *th;
    double S[7] = {0, 0, 0, 0, 0, 0, 0};
    S[0] = (tot[2] - tot[3]) / t.var;
    S[5] = tot[4] - tot[5];
    for (int i = 0; i < dv.d; ++i) S[1 + i] = tot[6 + i] - tot[10 + i];
    // a structurally indefinite spectrum (D <= 0) cannot happen with jitter + noise > 0 unless the
    // eigenvalues are garbage (NaN inputs): flag it like a failed factorisation
    if (!(tot[0] == tot[0]) && *info == 0) *info = 1;
    AdamStep st;
    st.beta1 = 0.9; st.beta2 = 0.999; st.eps = 1e-8; st.lr_over_bc1 = 0.0; st.bc2_sqrt = 1.0;
    double* hist_row = nullptr;
    if (fi.iter) {
        const int it = *fi.iter;
        if (*info != 0) { atomicMin(info + 1, it); return; }
        const int P = 2 + m.n_ls;
        st.lr_over_bc1 = fi.bc[it];
        st.bc2_sqrt = fi.bc[fi.T + it];
        loss_out = fi.loss_base ? fi.loss_base + it : nullptr;
        hist_row = fi.hist_base ? fi.hist_base + (int64_t)it * P : nullptr;
        *fi.iter = it + 1;
    }
    finalize_step(m, dv.N, S, tot[1], tot[0], t, u, adam_m, adam_v, do_adam, st, loss_out, grad_out, hist_row,
                  prior_constant(m));
}

// ------------------------------------------------------------------------------------------
// posterior
// ------------------------------------------------------------------------------------------
// Ks_i[t, a] = exp(-((ct[t] - c[a]) / l_i)^2 / 2)
__global__ void kron_cross_kernel(KronDev dv, const ThetaDev* __restrict__ th, const double* __restrict__ tcoords,
                                  int toff, int mi, int ax, double* __restrict__ Ks) {
    const int n = dv.n[ax];
    const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= (int64_t)mi * n) return;
    const int t = (int)(e / n), a = (int)(e % n);
    const double il = 1.0 / th->ls[ax];
    const double dl = tcoords[toff + t] * il - dv.coords[dv.off[ax] + a] * il;
    Ks[e] = exp(-0.5 * dl * dl);
}
__global__ void kron_invd_kernel(KronDev dv, const ThetaDev* __restrict__ th, double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= dv.N) return;
    int64_t r = j;
    double lam = th->var;
    for (int i = dv.d - 1; i >= 0; --i) { lam *= dv.lam[dv.off[i] + (int)(r % dv.n[i])]; r /= dv.n[i]; }
    out[j] = 1.0 / (lam + th->diag_add);
}
// (mean_raw may alias mean_out: element j is read before it is written, by the same thread)
__global__ void kron_post_kernel(const double* mean_raw, const double* __restrict__ q, int64_t M,
                                 const ThetaDev* __restrict__ th, double* mean_out, double* __restrict__ var_out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= M) return;
    const double s2 = th->var;
    mean_out[j] = s2 * mean_raw[j];
    var_out[j] = clamp0_nan(s2 - s2 * s2 * q[j]) + th->noise;
}

// ------------------------------------------------------------------------------------------
// drivers
// ------------------------------------------------------------------------------------------
static int kron_cold_start(gpimhip_ctx* h, KronWs& w) {
    int nmax = 1;
    for (int pr = 0; pr < w.dev.nprob; ++pr) nmax = std::max(nmax, w.dev.pn[pr]);
    hipLaunchKernelGGL(kron_eye_kernel, dim3((unsigned)(((int64_t)nmax * nmax + 255) / 256), w.dev.nprob), dim3(256), 0,
                       h->stream, w.dev);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// Splits every axis whose coordinates are centro-symmetric (c_a + c_{n-1-a} constant: any uniform grid) into a
// symmetric and a skew-symmetric eigen-problem.  Needs the coordinates on the host: one small copy + sync per call.
static int kron_plan_problems(gpimhip_ctx* h, KronWs& w) {
    std::vector<double> c((size_t)w.sum_n);
    HIP_TRY(hipMemcpyAsync(c.data(), w.dev.coords, c.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    KronDev& dv = w.dev;
    dv.nprob = 0;
    const bool allow = true;
    for (int i = 0; i < w.d; ++i) {
        const int n = w.n[i];
        const double* ci = c.data() + dv.off[i];
        bool sym = allow && n >= 8;
        double scale = 1.0;
        for (int a = 0; a < n; ++a) scale = std::max(scale, fabs(ci[a]));
        for (int a = 0; a < n && sym; ++a)
            sym = fabs((ci[a] + ci[n - 1 - a]) - (ci[0] + ci[n - 1])) <= 1e-12 * scale;
        auto add = [&](int part, int order, int64_t off, int lam0) {
            const int pr = dv.nprob++;
            dv.pax[pr] = i; dv.ppart[pr] = part; dv.pn[pr] = order; dv.poff[pr] = off; dv.plam[pr] = lam0;
        };
        if (sym) {
            const int ns = (n + 1) / 2, nd = n / 2;
            add(1, ns, 0, 0);
            add(2, nd, (int64_t)ns * ns, ns);
        } else {
            add(0, n, 0, 0);
        }
    }
    return GPIMHIP_OK;
}

// K_i(u), its eigen-decomposition continued from the rotations in Vr (kron_cold_start() resets them), y~
static int kron_decompose(gpimhip_ctx* h, KronWs& w, const gpimhip_model_t* m, const double* u, const double* y) {
    const KronDev& dv = w.dev;
    GP_TRY(launch_theta(h, m, u));
    int nmax = 1, pmax = 1;
    for (int i = 0; i < w.d; ++i) nmax = std::max(nmax, w.n[i]);
    for (int pr = 0; pr < dv.nprob; ++pr) pmax = std::max(pmax, dv.pn[pr]);
    hipLaunchKernelGGL(kron_axis_kernel, dim3((unsigned)(((int64_t)nmax * nmax + 255) / 256), w.d), dim3(256), 0, h->stream,
                       dv, h->theta);
    hipLaunchKernelGGL(kron_reduce_kernel, dim3((unsigned)(((int64_t)pmax * pmax + 255) / 256), dv.nprob), dim3(256), 0,
                       h->stream, dv);
    for (int pr = 0; pr < dv.nprob; ++pr) {      // W = V * (matrix of the problem)
        const int64_t o = dv.moff[dv.pax[pr]] + dv.poff[pr];
        const int np_ = dv.pn[pr];
        GP_TRY(modeprod(h, dv.Tt + o, dv.W + o, dv.Vr + o, np_, np_, np_, 0, 0, 1, np_));
    }
    if (pmax <= 64) hipLaunchKernelGGL((kron_eigh_kernel<1, 4>), dim3(dv.nprob), dim3(EIGH_WAVES * 64), 0, h->stream, dv);
    else if (pmax <= 128) hipLaunchKernelGGL((kron_eigh_kernel<2, 4>), dim3(dv.nprob), dim3(EIGH_WAVES * 64), 0, h->stream, dv);
    else if (pmax <= 256) hipLaunchKernelGGL((kron_eigh_kernel<4, 2>), dim3(dv.nprob), dim3(EIGH_WAVES * 64), 0, h->stream, dv);
    else if (pmax <= 512) hipLaunchKernelGGL((kron_eigh_kernel<8, 1>), dim3(dv.nprob), dim3(EIGH_WAVES * 64), 0, h->stream, dv);
    else hipLaunchKernelGGL((kron_eigh_kernel<0, 1>), dim3(dv.nprob), dim3(EIGH_WAVES * 64), 0, h->stream, dv);
    hipLaunchKernelGGL(kron_assemble_kernel, dim3((unsigned)(((int64_t)pmax * nmax + 255) / 256), dv.nprob), dim3(256), 0,
                       h->stream, dv);
    HIP_TRY(hipGetLastError());
    // y~ = (Q_1^T (x) ... (x) Q_d^T) y
    const double* mats[KMAXD]; int ld[KMAXD];
    for (int i = 0; i < w.d; ++i) { mats[i] = dv.Qt + dv.moff[i]; ld[i] = w.n[i]; }
    double* res;
    GP_TRY(tensor_apply(h, w.d, w.n, w.n, mats, ld, 0, 0, y, w.t1, w.t2, &res));
    HIP_TRY(hipMemcpyAsync(w.yt, res, (size_t)w.N * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    return GPIMHIP_OK;
}

static int kron_loss_grad(gpimhip_ctx* h, KronWs& w, const gpimhip_model_t* m, const double* y, double* u, int do_adam,
                          double* loss_out, double* grad_out, const KronIter* it) {
    const KronDev& dv = w.dev;
    GP_TRY(kron_decompose(h, w, m, u, y));
    // M_i = Q_i^T E_i Q_i  (two mode products on the n_i x n_i "tensor") and its diagonal
    int nmax = 1;
    for (int i = 0; i < w.d; ++i) {
        const int n = w.n[i];
        nmax = std::max(nmax, n);
        GP_TRY(modeprod(h, dv.E + dv.moff[i], dv.Tt + dv.moff[i], dv.Qt + dv.moff[i], n, n, n, 0, 0, 1, n));
        GP_TRY(modeprod(h, dv.Tt + dv.moff[i], dv.Mm + dv.moff[i], dv.Qt + dv.moff[i], n, n, n, 0, 0, n, 1));
    }
    hipLaunchKernelGGL(kron_mdiag_kernel, dim3((nmax + 63) / 64, w.d), dim3(64), 0, h->stream, dv);
    hipLaunchKernelGGL(kron_dvec_kernel, dim3(w.nblk), dim3(256), 0, h->stream, dv, h->theta, w.yt, w.at, w.part);
    for (int i = 0; i < w.d; ++i) {
        int64_t pre = 1, post = 1;
        for (int k = 0; k < i; ++k) pre *= w.n[k];
        for (int k = i + 1; k < w.d; ++k) post *= w.n[k];
        GP_TRY(modeprod(h, w.at, w.Z + (int64_t)i * w.N, dv.Mm + dv.moff[i], w.n[i], w.n[i], w.n[i], 0, 0, pre, post));
    }
    hipLaunchKernelGGL(kron_quad_kernel, dim3(w.nblk), dim3(256), 0, h->stream, dv, w.at, w.Z, w.part);
    KronIter fi{nullptr, nullptr, 0, nullptr, nullptr};
    if (it) fi = *it;
    hipLaunchKernelGGL(kron_finalize_kernel, dim3(1), dim3(256), 0, h->stream, *m, dv, w.nblk, w.part, h->theta, u,
                       h->adam_m, h->adam_v, do_adam, loss_out, grad_out, fi, h->info);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

static int kron_prepare(gpimhip_ctx* h, const gpimhip_model_t* m, int32_t d, const int32_t* n, const double* axes,
                        KronWs** w) {
    GP_TRY(check_model(m));
    if (m->kernel != GPIMHIP_KERNEL_RBF) {
        gpim_set_error("structured (Kronecker) solver: only the RBF kernel factorises over the grid axes");
        return GPIMHIP_E_BADARG;
    }
    if (d != m->dim || !n || !axes) return GPIMHIP_E_BADARG;
    for (int i = 0; i < d; ++i)
        if (n[i] < 1 || n[i] > 4096) return GPIMHIP_E_BADARG;
    HIP_TRY(hipSetDevice(h->device));
    h->nbatch = 1;
    GP_TRY(ws_ensure(h, 1));                   // theta, Adam state, iteration counter
    GP_TRY(kron_ensure(h, d, n, w));
    HIP_TRY(hipMemcpyAsync(const_cast<double*>((*w)->dev.coords), axes, (size_t)(*w)->sum_n * sizeof(double),
                           hipMemcpyDeviceToDevice, h->stream));
    HIP_TRY(hipMemsetAsync(h->info, 0, sizeof(int32_t), h->stream));
    GP_TRY(kron_plan_problems(h, **w));
    return kron_cold_start(h, **w);
}

extern "C" {

int gpimhip_kron_nll_grad(gpimhip_handle h, const gpimhip_model_t* m, int32_t d, const int32_t* n, const double* axes,
                          const double* y, const double* u, double* loss_out, double* grad_out) {
    if (!h || !m || !y || !u) return GPIMHIP_E_BADARG;
    KronWs* w;
    GP_TRY(kron_prepare(h, m, d, n, axes, &w));
    GP_TRY(kron_loss_grad(h, *w, m, y, const_cast<double*>(u), 0, loss_out, grad_out, nullptr));
    return finish_and_check(h);
}

int gpimhip_fit_kron(gpimhip_handle h, const gpimhip_model_t* m, int32_t d, const int32_t* n, const double* axes,
                     const double* y, double* u_inout, double lr, int32_t T, double* hist_out, double* loss_out) {
    if (!h || !m || !y || !u_inout || T < 0) return GPIMHIP_E_BADARG;
    KronWs* w;
    GP_TRY(kron_prepare(h, m, d, n, axes, &w));
    HIP_TRY(hipMemsetAsync(h->info + 1, 0x7f, sizeof(int32_t), h->stream));
    h->fit_completed = T;
    if (T == 0) return GPIMHIP_OK;
    GP_TRY(upload_bc_table(h, lr, T));
    HIP_TRY(hipMemsetAsync(h->adam_m, 0, MAXP * sizeof(double), h->stream));
    HIP_TRY(hipMemsetAsync(h->adam_v, 0, MAXP * sizeof(double), h->stream));
    HIP_TRY(hipMemsetAsync(h->iter, 0, sizeof(int32_t), h->stream));
    KronIter it{h->iter, h->bc, T, hist_out, loss_out};
    // every iteration is the same ~4d + 8 small launches: capture one into a hipGraph and replay it
    if (T >= 8 && !getenv("GPIMHIP_NO_GRAPH") && ensure_capture_stream(h)) {
        hipGraph_t graph = nullptr;
        hipGraphExec_t exec = nullptr;
        hipStream_t main_s = h->stream;
        h->stream = h->capture_stream;
        capture_lock(h);
        hipError_t e = hipStreamBeginCapture(h->capture_stream, hipStreamCaptureModeRelaxed);
        int rc = GPIMHIP_OK;
        if (e == hipSuccess) {
            rc = kron_loss_grad(h, *w, m, y, u_inout, 1, nullptr, nullptr, &it);
            e = hipStreamEndCapture(h->capture_stream, &graph);
        }
        capture_unlock(h);
        h->stream = main_s;
        if (rc != GPIMHIP_OK) { if (graph) (void)hipGraphDestroy(graph); return rc; }
        if (e == hipSuccess && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) == hipSuccess) {
            hipError_t le = hipSuccess;
            for (int t = 0; t < T && le == hipSuccess; ++t) le = hipGraphLaunch(exec, main_s);
            rc = finish_and_check(h);
            (void)hipGraphExecDestroy(exec);
            (void)hipGraphDestroy(graph);
            HIP_TRY(le);
            return rc;
        }
        if (graph) (void)hipGraphDestroy(graph);
        (void)hipGetLastError();
    }
    for (int t = 0; t < T; ++t) GP_TRY(kron_loss_grad(h, *w, m, y, u_inout, 1, nullptr, nullptr, &it));
    return finish_and_check(h);
}

int gpimhip_predict_kron(gpimhip_handle h, const gpimhip_model_t* m, int32_t d, const int32_t* n, const double* axes,
                         const double* y, const double* u, const int32_t* n_test, const double* axes_test,
                         double* mean_out, double* var_out) {
    if (!h || !m || !y || !u || !n_test || !axes_test || !mean_out || !var_out) return GPIMHIP_E_BADARG;
    KronWs* wp;
    GP_TRY(kron_prepare(h, m, d, n, axes, &wp));
    KronWs& w = *wp;
    const KronDev& dv = w.dev;
    // prediction workspace
    int64_t M = 1, sum_m = 0, sum_mn = 0, pmax = w.N;
    {
        int64_t cur = w.N;
        for (int i = 0; i < d; ++i) {
            if (n_test[i] < 1) return GPIMHIP_E_BADARG;
            M *= n_test[i]; sum_m += n_test[i]; sum_mn += (int64_t)n_test[i] * w.n[i];
            cur = cur / w.n[i] * n_test[i];
            pmax = std::max(pmax, cur);
        }
    }
    bool same = w.parena && w.M == M && w.pmax == pmax;
    for (int i = 0; i < d && same; ++i) same = w.m[i] == n_test[i];
    if (!same) {
        HIP_TRY(hipStreamSynchronize(h->stream));
        if (w.parena) { (void)hipFree(w.parena); h->bytes -= w.parena_count * (int64_t)sizeof(double); w.parena = nullptr; }
        const int64_t count = sum_m + 2 * sum_mn + 2 * pmax + 64;
        void* q = nullptr;
        if (hipMalloc(&q, (size_t)count * sizeof(double)) != hipSuccess) {
            gpim_set_error("hipMalloc failed (structured-GP prediction workspace)");
            return GPIMHIP_E_NOMEM;
        }
        w.parena = (double*)q; w.parena_count = count; h->bytes += count * (int64_t)sizeof(double);
        double* p = w.parena;
        auto take = [&](int64_t c) { double* r = p; p += (c + 1) / 2 * 2; return r; };
        w.tcoords = take(sum_m); w.Ks = take(sum_mn); w.Bs = take(sum_mn); w.p1 = take(pmax); w.p2 = take(pmax);
        int64_t ko = 0;
        for (int i = 0; i < KMAXD; ++i) {
            w.m[i] = i < d ? n_test[i] : 1;
            w.koff[i] = ko;
            if (i < d) ko += (int64_t)n_test[i] * w.n[i];
        }
        w.M = M; w.pmax = pmax;
    }
    HIP_TRY(hipMemcpyAsync(w.tcoords, axes_test, (size_t)sum_m * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    GP_TRY(kron_decompose(h, w, m, u, y));
    hipLaunchKernelGGL(kron_dvec_kernel, dim3(w.nblk), dim3(256), 0, h->stream, dv, h->theta, w.yt, w.at, w.part);
    // cross-covariances per axis and B_i = Ks_i Q_i
    int toff = 0;
    for (int i = 0; i < d; ++i) {
        const int64_t cnt = (int64_t)w.m[i] * w.n[i];
        hipLaunchKernelGGL(kron_cross_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, h->stream, dv, h->theta,
                           w.tcoords, toff, w.m[i], i, w.Ks + w.koff[i]);
        GP_TRY(modeprod(h, w.Ks + w.koff[i], w.Bs + w.koff[i], dv.Qt + dv.moff[i], w.n[i], w.n[i], w.n[i], 0, 0, w.m[i], 1));
        toff += w.m[i];
    }
    HIP_TRY(hipGetLastError());
    // mean = s2 * ((x) B_i) alpha~        (B_i = Ks_i Q_i maps eigen-coordinates straight to test points)
    const double* mats[KMAXD]; int ld[KMAXD];
    for (int i = 0; i < d; ++i) { mats[i] = w.Bs + w.koff[i]; ld[i] = w.n[i]; }
    double* res;
    GP_TRY(tensor_apply(h, d, w.n, w.m, mats, ld, 0, 0, w.at, w.p1, w.p2, &res));
    HIP_TRY(hipMemcpyAsync(mean_out, res, (size_t)M * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    // var = s2 - s2^2 * ((x) B_i o B_i) (1 / D)
    hipLaunchKernelGGL(kron_invd_kernel, dim3((unsigned)((w.N + 255) / 256)), dim3(256), 0, h->stream, dv, h->theta, w.t1);
    GP_TRY(tensor_apply(h, d, w.n, w.m, mats, ld, 0, 1, w.t1, w.p1, w.p2, &res));
    hipLaunchKernelGGL(kron_post_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, h->stream, mean_out, res, M,
                       h->theta, mean_out, var_out);
    HIP_TRY(hipGetLastError());
    return finish_and_check(h);
}

}  // extern "C"
