// cholstep.hip -- the blocked Cholesky as a single in-order stream of two kinds of launches (SURVEY 8(a) row a6;
// replaces torch.linalg.cholesky at gpim/gpreg/gpr.py:192,248).
//
// The diagonal-block factorisation (potf2, one workgroup, 128 serial pivots, ~36 us) is the one step of the
// factorisation that cannot be spread over the chip.  Instead of hiding it behind a second stream, its launch
// HOSTS the trailing update: workgroup 0 of `chol_step_kernel` factors block j, every other workgroup of the same
// launch computes one tile of pending trailing-update work with the MFMA tile engine ("fillers").  To make enough
// filler work independent of block j the update order is left-looking inside a window of two outer panels and
// right-looking beyond it:
//
//   H_j  (chol_step_kernel)   potf2(j)  ||  column j, rows i > j:  A[i,j] -= sum_{c in [p0(j)-W, j)} L[i,c] L[j,c]^T
//                                       ||  a share of bulk(P-1): tiles (i,jj), jj >= p0(P)+W, -= L[i,P-1] L[jj,P-1]^T
//   F_j  (panel_solve_kernel) L[i,j] = A[i,j] * inv(L[j,j])^T for all rows i > j, one-shot: the whole strip and the
//                             inverse (in MFMA operand order, written by potf2) are fetched with every load in
//                             flight at once -- no k-loop of load/barrier/compute rounds
//   D_j  (tile engine)        the next diagonal tiles (jj,jj), j < jj < p1(j)+W:  A[jj,jj] -= L[jj,j] L[jj,j]^T
//
// P = outer panel of W = 4 block columns, p0/p1 its first / one-past-last column.  Per block column the chain is
// H_j -> F_j -> D_j; nothing else is ever on it.  Every tile receives each k-block exactly once: columns of the
// two most recent panels through the left-looking column updates (off-diagonal) or D (diagonal), older ones
// through bulk() -- see build_step_plan().
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <functional>
#include "potf2_body.hpp"
#include "gemm_body.hpp"

#define STEP_W 4

struct StepArgs {
    double* A; int64_t ld; int kblk; int nb;
    double* dinv_all; double* dinvB_all; double* logdet; int32_t* info;
    int col_off;                // global index of the sub-matrix's first column (non-PD reporting)
    GemmArgs g;                 // filler tiles (NT, alpha = -1, beta = 1)
};

// blockIdx.x == 0: potf2 of block kblk; 1..7: idle (keeps filler b on XCD b % 8, which the tile engine's
// XCD-aware list mapping assumes); >= 8: filler tile b - 8.  FTS: 128 = one workgroup per 128x128 tile,
// 64 = one per 64x64 quadrant (few tiles: spread each over four CUs).
// The hosted tiles run one workgroup per CU.  A ring of 3 - 6 LDS stages (the factorisation role's 134 KB are
// reserved anyway; gemm_tile_body<..., NSTG>, build with -DGPIMHIP_STEP_RING) that keeps the loads of several
// k-steps in flight was measured and does NOT help: potrf 1.91 vs 1.86 ms at N = 4224, 6.58 vs 6.39 at 8192,
// 32.7 vs 32.1 at 16384 -- what a lone 8-wave workgroup lacks is not load latency cover.
template <int FTM, int FTN>
__global__ __launch_bounds__(NTH, 4) void chol_step_kernel(StepArgs a) {
#ifdef GPIMHIP_STEP_RING
    constexpr int NSTG = (FTM == 128 && FTN == 128) ? 3 : (FTM == 128 ? 4 : 6);
#else
    constexpr int NSTG = 2;
#endif
    constexpr int SM = POTF2_SMEM_DOUBLES > gemm_smem_doubles<FTM, FTN, NSTG>() ? POTF2_SMEM_DOUBLES
                                                                                : gemm_smem_doubles<FTM, FTN, NSTG>();
    static_assert(SM * 8 <= 160 * 1024, "LDS");
    __shared__ __attribute__((aligned(16))) double smem[SM];
    const int b = blockIdx.x;
    if (b >= 8) {
        gemm_tile_body<false, false, EPI_STORE, 8, FTM, FTN, NSTG>(a.g, b - 8, (int)blockIdx.y, smem);
        return;
    }
    if (b != 0) return;
    potf2_body<double>(smem, (int)blockIdx.y, a.A, a.ld, a.kblk, a.dinv_all, a.dinvB_all, a.logdet, a.info, a.nb, a.col_off);
}

// Panel solve, one workgroup (4 waves) per 32-row strip of the block column below the diagonal block:
// S = P * Dinv^T, in place.  Wave w owns the 16-column tiles w and 7 - w of the strip (Dinv is lower triangular:
// tile t needs k-steps 0 .. 4t+3, so every wave runs 36 of them per 16 rows).
// ROWS = rows of one workgroup's strip: 32, or 16 (twice the workgroups, half the MFMA chain per wave: the kernel is
// latency-bound, and the inverse it re-reads per workgroup comes out of L2).  Measured: 16 and 64 are both slower than
// 32 (potrf at N = 4212: 1.694 / 1.647 / 1.783 ms, at 16384: 29.65 / 28.96 / 29.60); GPIMHIP_STRIP16=1 selects 16.
template <int ROWS>
__global__ __launch_bounds__(256) void panel_solve_kernel(double* __restrict__ A, int64_t ld, int kblk, int nb,
                                                          const double* __restrict__ dinvB_all) {
    constexpr int LDS_LD = 130;             // 130 % 32 == 2: the A-fragment reads below are bank-conflict free
    constexpr int MTS = ROWS / 16;
    __shared__ __attribute__((aligned(16))) double S[ROWS * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    A += (int64_t)blockIdx.y * nb * NB * ld;
    const d2* DB = reinterpret_cast<const d2*>(dinvB_all + ((int64_t)blockIdx.y * nb + kblk) * (NB * NB));
    double* P = A + ((int64_t)(kblk + 1) * NB + (int64_t)blockIdx.x * ROWS) * ld + (int64_t)kblk * NB;
    // the strip: one 1 KB row per wave-wide load, global -> LDS directly
#pragma unroll
    for (int i = 0; i < ROWS / 4; ++i) {
        const int row = wave * (ROWS / 4) + i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(P + (int64_t)row * ld + lane * 2),
                                         (__attribute__((address_space(3))) void*)(S + row * LDS_LD), 16, 0, 0);
    }
    // B fragments of both tiles: 18 sixteen-byte loads per lane, all in flight
    const int t0 = wave, t1 = 7 - wave, n0 = 2 * t0 + 2;
    d2 bf[18];
#pragma unroll
    for (int q = 0; q < 18; ++q) {
        const int t = q < n0 ? t0 : t1, s2 = q < n0 ? q : q - n0;
        bf[q] = DB[(t * 16 + s2) * 64 + lane];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const d4 zero = (d4){0.0, 0.0, 0.0, 0.0};
    d4 acc[2][MTS];
#pragma unroll
    for (int x = 0; x < 2; ++x)
#pragma unroll
        for (int mt = 0; mt < MTS; ++mt) acc[x][mt] = zero;
    const double* Sa = S + (lane & 15) * LDS_LD + (lane >> 4);
#pragma unroll
    for (int q = 0; q < 18; ++q) {
        const bool first = q < n0;          // wave-uniform
        const int s2 = first ? q : q - n0;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int s = 2 * s2 + e;
#pragma unroll
            for (int mt = 0; mt < MTS; ++mt) {
                const double av = Sa[mt * 16 * LDS_LD + 4 * s];
                if (first) acc[0][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bf[q][e], acc[0][mt], 0, 0, 0);
                else acc[1][mt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bf[q][e], acc[1][mt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int x = 0; x < 2; ++x) {
        const int t = x == 0 ? t0 : t1;
#pragma unroll
        for (int mt = 0; mt < MTS; ++mt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                P[(int64_t)(mt * 16 + (lane >> 4) + 4 * rg) * ld + t * 16 + (lane & 15)] = acc[x][mt][rg];
    }
}

// Diagonal tiles (jj,jj), jj = kblk+1 .. kblk+ntile, -= L[jj,kblk] L[jj,kblk]^T, one workgroup (4 waves, one
// 16x16 MFMA tile each) per 32x32 quadrant of the lower half; one-shot like the panel solve: both 32x128
// operand strips go straight to LDS, the C values to registers, all loads in flight together.
__global__ __launch_bounds__(256) void diag_update_kernel(double* __restrict__ A, int64_t ld, int kblk, int nb) {
    constexpr int LDS_LD = 130;
    __shared__ __attribute__((aligned(16))) double S[64 * LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    A += (int64_t)blockIdx.y * nb * NB * ld;
    const int jj = kblk + 1 + blockIdx.x / 10, q = blockIdx.x % 10;
    // q -> (a, b), a >= b, a, b in 0..3
    const int a = q < 1 ? 0 : q < 3 ? 1 : q < 6 ? 2 : 3, b = q - a * (a + 1) / 2;
    const double* Pa = A + ((int64_t)jj * NB + a * 32) * ld + (int64_t)kblk * NB;
    const double* Pb = A + ((int64_t)jj * NB + b * 32) * ld + (int64_t)kblk * NB;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = wave * 8 + i;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Pa + (int64_t)row * ld + lane * 2),
                                         (__attribute__((address_space(3))) void*)(S + row * LDS_LD), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Pb + (int64_t)row * ld + lane * 2),
                                         (__attribute__((address_space(3))) void*)(S + (32 + row) * LDS_LD), 16, 0, 0);
    }
    const int wm = wave >> 1, wn = wave & 1;
    double* C = A + ((int64_t)jj * NB + a * 32 + wm * 16 + (lane >> 4)) * ld + (int64_t)jj * NB + b * 32 + wn * 16 + (lane & 15);
    double cv[4];
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) cv[rg] = C[(int64_t)(4 * rg) * ld];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    d4 acc0 = (d4){0.0, 0.0, 0.0, 0.0}, acc1 = acc0;      // two chains: a lone dependent fp64 MFMA chain issues at half rate
    const double* Sa = S + (wm * 16 + (lane & 15)) * LDS_LD + (lane >> 4);
    const double* Sb = S + (32 + wn * 16 + (lane & 15)) * LDS_LD + (lane >> 4);
#pragma unroll
    for (int s = 0; s < 32; s += 2) {
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(Sa[4 * s], Sb[4 * s], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(Sa[4 * s + 4], Sb[4 * s + 4], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) C[(int64_t)(4 * rg) * ld] = cv[rg] - (acc0[rg] + acc1[rg]);
}

// ------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------
static void lower_patches(std::vector<TileDesc>& out, int lo, int hi, int kb0, int kb1) {
    for (int ig = lo / 8; ig <= (hi - 1) / 8; ++ig)
        for (int jg = lo / 8; jg <= ig; ++jg)
            for (int i = std::max(lo, ig * 8); i < std::min(hi, ig * 8 + 8); ++i)
                for (int j = std::max(lo, jg * 8); j < std::min(hi, jg * 8 + 8); ++j)
                    if (j <= i) out.push_back({i, j, kb0, kb1});
}

static bool pair_mode(int nb, bool fp32) {
    static const int v = getenv("GPIMHIP_PAIR") ? atoi(getenv("GPIMHIP_PAIR")) : -1;
    if (v >= 0) return v != 0;
    // float matrices (tools/r3_f32_sweep2.sh): pairs lose up to N = 16384 (17.70 vs 17.33 ms), draw at 20480 (30.6 / 30.8)
    if (fp32) return nb >= 160;
    return nb >= 112;     // N = 16384: 32.2 -> 31.4 ms, 20480: 57.9 -> 57.3; no gain at 8192 / 12288
}

// Share of the previous panel's bulk update hosted by one step launch: at most this many tiles; what is left runs
// as a plain tile-engine launch before the panel's first step.  Hosted tiles run one workgroup per CU (the 134 KB
// factorisation role sets the launch's LDS size), 7 % slower than in their own launch (two per CU): up to
// nb = 63 everything is hosted (the factorisation is bound by the chain of diagonal blocks, hosted tiles are
// free), beyond that 32 tiles per launch (N = 16384: 32.1 ms with 0 / 32 / 128, 34.0 with everything hosted;
// N = 10240: 10.6 / 10.2 / 10.8 / 10.6).
// Float matrices: a hosted tile takes half the matrix-core time while the factorisation role (double) lasts as long
// as ever, so more is hosted: everything up to nb = 87 (potrf at N = 6400 / 8192 / 10240: 2.65 / 4.04 / 6.23 ms; with
// 64 tiles per launch 2.72 / 4.19 / 6.42), 128 tiles per launch beyond (12288: 9.10 vs 9.25 (64) / 9.35 (all);
// 16384: 17.33 vs 17.68 / 18.58), tools/r3_f32_sweep2.sh.
static int fill_cap(int nb, bool fp32) {
    static const int v = getenv("GPIMHIP_FILL_CAP") ? atoi(getenv("GPIMHIP_FILL_CAP")) : -1;
    if (v >= 0) return v;
    if (pair_mode(nb, fp32)) return fp32 ? 64 : 0;   // fp64: hosted k-depth-1024 tiles would outlast the factorisation role by far
    if (fp32) return nb < 88 ? (1 << 30) : 128;
    return nb < 64 ? (1 << 30) : 32;
}

// ---- hosting plan for double-precision matrices of nb >= 64 block columns -------------------------------------
// Everything the trailing matrix receives from panels older than the window is HOSTED by the step launches (two
// workgroups per CU since the factorisation role fits 79 KB).  What a launch hosts is chosen so that it ends on a
// full round of the chip's 512 workgroup slots:
//   * tile (i, jj) keeps the first block column it has not yet received (`pend`); a flush applies [pend, p0) in ONE
//     tile operation, so a tile that is skipped for a panel comes back twice as deep (k = 1024: half the
//     read-modify-write passes over C, half the prologues -- what r3's "pair mode" did for whole panels);
//   * tiles of the NEXT panel's columns must be flushed during this panel (deadline), the others are optional;
//   * steady state: the two patch classes ((i / 8 + jj / 8) & 1) take turns, each flushed every other panel at depth
//     8 -- every panel carries the same amount of work -- and the holes of a launch's last round are filled with
//     whatever else is pending (list scheduling on 512 slots, the order the hardware dispatches workgroups in).
// Below 64 remaining block columns a panel's updates are less than one round per launch: the chain of diagonal
// blocks bounds the factorisation and everything pending is flushed at once, evenly over the panel's launches.
struct HostSim {
    std::vector<double> slot;       // min-heap of slot finish times
    double makespan = 0.0;
    explicit HostSim(int S) : slot(S, 0.0) {}
    static double cost(const TileDesc& t) { return (t.kb1 - t.kb0) + 0.5; }     // k-blocks + prologue / epilogue
    double peek() const { return slot.front(); }
    void add(double c) {
        std::pop_heap(slot.begin(), slot.end(), std::greater<double>());
        slot.back() += c;
        makespan = std::max(makespan, slot.back());
        std::push_heap(slot.begin(), slot.end(), std::greater<double>());
    }
};

static void step_plan_hosted(int nb, std::vector<TileDesc>& tl, StepPlan& P) {
    const int W = STEP_W, S = 512;
    auto mark = [&](size_t start) { return PlanRange{(int64_t)start, (int32_t)(tl.size() - start)}; };
    const int npanel = (nb + W - 1) / W;
    std::vector<int> pend((size_t)nb * nb, 0);        // first source block column tile (i, jj) has not received via bulk
    for (int p = 0; p < npanel; ++p) {
        const int p0 = p * W, p1 = std::min(p0 + W, nb), ncol = p1 - p0;
        const int rem = nb - p0 - W;
        std::vector<TileDesc> req, opt;
        if (p > 0 && rem > 0) {
            const int c0 = p0 + W, c1 = std::min(p0 + 2 * W, nb);
            for (int i = c0; i < nb; ++i)
                for (int jj = c0; jj < std::min(c1, i + 1); ++jj) req.push_back({i, jj, pend[(size_t)i * nb + jj], p0});
            if (c1 < nb) lower_patches(opt, c1, nb, 0, p0);
            for (auto& t : opt) t.kb0 = pend[(size_t)t.ci * nb + t.cj];
        }
        const bool bulk_regime = rem >= 64;
        if (bulk_regime) {
            // due first (deepest first), then the rest; both in patch order
            auto due = [&](const TileDesc& t) { return (t.kb1 - t.kb0) >= 2 * W || (((t.ci >> 3) + (t.cj >> 3)) & 1) == (p & 1); };
            std::stable_sort(opt.begin(), opt.end(), [&](const TileDesc& a, const TileDesc& b) {
                const int da = due(a), db = due(b);
                if (da != db) return da > db;
                return (a.kb1 - a.kb0) > (b.kb1 - b.kb0);
            });
        }
        double due_cost = 0.0;                 // cost of the due tiles not yet hosted
        size_t n_due = 0;
        if (bulk_regime)
            for (auto& t : opt)
                if ((t.kb1 - t.kb0) >= 2 * W || (((t.ci >> 3) + (t.cj >> 3)) & 1) == (p & 1)) { ++n_due; due_cost += HostSim::cost(t); }
        size_t rtaken = 0, otaken = 0;
        for (int j = p0; j < p1; ++j) {
            const int left = p1 - j;
            size_t s = tl.size();
            const int kb0 = std::max(0, p0 - W);
            if (j > kb0)
                for (int i = j + 1; i < nb; ++i) tl.push_back({i, j, kb0, j});
            const size_t nreq = (req.size() - rtaken + left - 1) / left;
            for (size_t q = 0; q < nreq; ++q) tl.push_back(req[rtaken++]);
            if (!bulk_regime) {
                const size_t nopt = (opt.size() - otaken + left - 1) / left;
                for (size_t q = 0; q < nopt; ++q) tl.push_back(opt[otaken++]);
            } else {
                HostSim sim(S);
                sim.add(1.0);                                               // the factorisation role
                std::stable_sort(tl.begin() + s, tl.end(), [](const TileDesc& a, const TileDesc& b) {
                    return (a.kb1 - a.kb0) > (b.kb1 - b.kb0); });
                for (size_t q = s; q < tl.size(); ++q) sim.add(HostSim::cost(tl[q]));
                // This launch's length: its share of the due work, rounded to whole deep tiles per slot (a slot runs
                // a sequence of tiles; 8.5 = one tile of depth 8) and never shorter than what it must host anyway.
                double base = 0.0;
                for (size_t q = s; q < tl.size(); ++q) base += HostSim::cost(tl[q]);
                const double per_slot = (base + due_cost / left) / S;
                const double target = std::max(sim.makespan, 8.5 * std::max(1.0, std::floor(per_slot / 8.5 + 0.5)));
                double got = 0.0;
                // due tiles first (the sorted list starts with them), then anything else that still ends in time
                while (otaken < opt.size() && sim.peek() + HostSim::cost(opt[otaken]) <= target + 0.25) {
                    const double c = HostSim::cost(opt[otaken]);
                    sim.add(c);
                    if (otaken < n_due) got += c;
                    tl.push_back(opt[otaken++]);
                }
                due_cost -= std::min(due_cost, got);
            }
            // dispatch order = list order: deepest tiles first
            std::stable_sort(tl.begin() + s, tl.end(), [](const TileDesc& a, const TileDesc& b) {
                return (a.kb1 - a.kb0) > (b.kb1 - b.kb0); });
            for (size_t q = s; q < tl.size(); ++q)
                if (tl[q].cj != j) pend[(size_t)tl[q].ci * nb + tl[q].cj] = tl[q].kb1;
            P.fill[j] = mark(s);
            s = tl.size();
            for (int jj = j + 1; jj < std::min(nb, p1 + W); ++jj) tl.push_back({jj, jj, j, j + 1});
            P.diag[j] = mark(s);
        }
    }
}

static int step_plan_build(gpimhip_ctx* h, int nb, StepPlan& P) {
    const bool fp32 = h->fp32 != 0;
    if (P.nb == nb && P.fp32 == (int)fp32) return GPIMHIP_OK;
    if (P.d_tiles) { (void)hipFree(P.d_tiles); P.d_tiles = nullptr; }
    const int W = STEP_W;
    std::vector<TileDesc> tl;
    auto mark = [&](size_t start) { return PlanRange{(int64_t)start, (int32_t)(tl.size() - start)}; };
    P.fill.assign(nb, {0, 0});
    P.diag.assign(nb, {0, 0});
    const int npanel = (nb + W - 1) / W;
    P.bulk_rest.assign(npanel, {0, 0});
    static const bool old_plan = getenv("GPIMHIP_OLD_PLAN") != nullptr;
    if (!fp32 && nb >= 64 && !old_plan) {
        step_plan_hosted(nb, tl, P);
    } else
    for (int p = 0; p < npanel; ++p) {
        const int p0 = p * W, p1 = std::min(p0 + W, nb), ncol = p1 - p0;
        // bulk(p-1): columns >= p0 + W, k-blocks = the columns of panel p-1.
        // Pair mode (large matrices): panels are applied two at a time to everything at least two panels to their
        // right -- k-depth 1024, half the passes over the trailing matrix -- and an even panel alone only to the one
        // destination panel that needs it before its partner is factored:
        //   start of an odd panel p  (p-1 even): panel p-1 -> destination panel p+1 only           (k-depth 512)
        //   start of an even panel p (p-1 odd) : panels p-2, p-1 -> every column >= p0 + W          (k-depth 1024)
        // Each destination panel Q still receives every source panel S <= Q-2 exactly once here and S >= Q-1 through
        // the left-looking column updates / diagonal updates below.
        std::vector<TileDesc> bulk;
        if (p > 0 && p0 + W < nb) {
            if (!pair_mode(nb, fp32)) {
                lower_patches(bulk, p0 + W, nb, p0 - W, p0);
            } else if ((p - 1) % 2 == 0) {
                const int c0 = p0 + W, c1 = std::min(p0 + 2 * W, nb);
                for (int ig = c0 / 8; ig <= (nb - 1) / 8; ++ig)
                    for (int i = std::max(c0, ig * 8); i < std::min(nb, ig * 8 + 8); ++i)
                        for (int j = c0; j < std::min(c1, i + 1); ++j) bulk.push_back({i, j, p0 - W, p0});
            } else {
                lower_patches(bulk, p0 + W, nb, p0 - 2 * W, p0);
            }
        }
        const int per = std::min<int>(fill_cap(nb, fp32), (int)((bulk.size() + ncol - 1) / ncol));
        size_t taken = 0;
        for (int j = p0; j < p1; ++j) {
            size_t s = tl.size();
            const int kb0 = std::max(0, p0 - W);
            if (j > kb0)
                for (int i = j + 1; i < nb; ++i) tl.push_back({i, j, kb0, j});
            for (int q = 0; q < per && taken < bulk.size(); ++q) tl.push_back(bulk[taken++]);
            P.fill[j] = mark(s);
            s = tl.size();
            for (int jj = j + 1; jj < std::min(nb, p1 + W); ++jj) tl.push_back({jj, jj, j, j + 1});
            P.diag[j] = mark(s);
        }
        size_t s = tl.size();
        while (taken < bulk.size()) tl.push_back(bulk[taken++]);
        P.bulk_rest[p] = mark(s);
    }
    P.n_tiles = (int64_t)tl.size();
    void* q = nullptr;
    HIP_TRY(hipMalloc(&q, std::max<size_t>(tl.size(), 1) * sizeof(TileDesc)));
    P.d_tiles = (TileDesc*)q;
    HIP_TRY(hipMemcpyAsync(P.d_tiles, tl.data(), tl.size() * sizeof(TileDesc), hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    P.nb = nb;
    P.fp32 = (int)fp32;
    return GPIMHIP_OK;
}

int step_plan_ensure(gpimhip_ctx* h, int nb) { return step_plan_build(h, nb, h->splan); }
int step_plan_ensure_tail(gpimhip_ctx* h, int nb) { return step_plan_build(h, nb, h->splan_tail); }

void step_plan_release(gpimhip_ctx* h) {
    for (StepPlan* P : {&h->splan, &h->splan_tail}) {
        if (P->d_tiles) (void)hipFree(P->d_tiles);
        P->d_tiles = nullptr;
        P->nb = 0;
    }
}

static GemmArgs nt_update(double* A, int64_t ld, const TileDesc* tiles, int n, int64_t rows) {
    GemmArgs g;
    memset(&g, 0, sizeof(g));
    g.A = A; g.lda = ld; g.B = A; g.ldb = ld; g.C = A; g.ldc = ld;
    g.alpha = -1.0; g.beta = 1.0; g.tiles = tiles; g.ntiles = n;
    g.sA = g.sB = g.sC = rows * ld;
    return g;
}

// lower Cholesky of the np x np matrix A (np = nb * 128), in place, on h->stream; h->dinv / h->dinvB / h->logdet_part
// receive the inverses of the diagonal blocks and the log-determinant partials like the in-order driver of api.hip.
// blk_off > 0: the trailing sub-matrix that starts at block (blk_off, blk_off) -- a Schur complement the caller has
// brought up to date with every column left of it (hybrid schedule of api.hip: look-ahead head, step-schedule tail).
int launch_potrf_steps_f32(gpimhip_ctx* h, double* A, int64_t np, int64_t ld, int32_t* info, int blk_off);   // cholstep32.hip
int launch_potrf_steps(gpimhip_ctx* h, double* A, int64_t np, int64_t ld, int32_t* info, int blk_off) {
    if (h->fp32) return launch_potrf_steps_f32(h, A, np, ld, info, blk_off);
    const int nb = (int)(np / NB) - blk_off, W = STEP_W;
    if (blk_off) GP_TRY(step_plan_ensure_tail(h, nb));
    else GP_TRY(step_plan_ensure(h, nb));
    const StepPlan& P = blk_off ? h->splan_tail : h->splan;
    A += (int64_t)blk_off * NB * (ld + 1);
    double* const dinv = h->dinv + (int64_t)blk_off * NB * NB;
    double* const dinvB = h->dinvB + (int64_t)blk_off * NB * NB;
    double* const logdet = h->logdet_part + blk_off;
    const int B = h->nbatch;
    static const int quad_max = getenv("GPIMHIP_FILL_QUAD_MAX") ? atoi(getenv("GPIMHIP_FILL_QUAD_MAX")) : 128;
    static const int half_max = getenv("GPIMHIP_FILL_HALF_MAX") ? atoi(getenv("GPIMHIP_FILL_HALF_MAX")) : 512;
    static const bool old_diag = getenv("GPIMHIP_OLD_DIAG") != nullptr;
    static const int host_max_batch = getenv("GPIMHIP_HOST_MAX_BATCH") ? atoi(getenv("GPIMHIP_HOST_MAX_BATCH")) : 4;
    static const int strip16_env = getenv("GPIMHIP_STRIP16") ? atoi(getenv("GPIMHIP_STRIP16")) : -1;
    const bool strip16 = strip16_env >= 0 ? strip16_env != 0 : false;
    for (int j = 0; j < nb; ++j) {
        if (j % W == 0 && P.bulk_rest[j / W].n) {
            GemmArgs g = nt_update(A, ld, P.d_tiles + P.bulk_rest[j / W].off, P.bulk_rest[j / W].n, h->np);
            GP_TRY(launch_gemm(h, false, false, EPI_STORE, g));
        }
        StepArgs a;
        a.A = A; a.ld = ld; a.kblk = j; a.nb = nb;
        a.dinv_all = dinv; a.dinvB_all = dinvB; a.logdet = logdet; a.info = info;
        a.col_off = blk_off * NB;
        a.g = nt_update(A, ld, P.d_tiles + P.fill[j].off, P.fill[j].n, h->np);
        const int nf = P.fill[j].n;
        // mixed k-ranges (column updates are up to 2W-1 blocks deep, bulk tiles W): deal the list to the XCDs in chunks
        static const int host_chunk = getenv("GPIMHIP_HOST_CHUNK") ? atoi(getenv("GPIMHIP_HOST_CHUNK")) : 0;
        a.g.chunk = host_chunk > 0 ? host_chunk : std::max(1, std::min(64, nf / 512));
        if (B > host_max_batch) {
            // large batches saturate the chip by themselves: the pending tiles run as their own launch (two to four
            // workgroups per CU) in front of a factorisation-only step launch.  Same tile operations in the same
            // order per output element as the hosted form, hence the same bits as a stand-alone problem.
            if (nf) GP_TRY(launch_gemm(h, false, false, EPI_STORE, a.g));
            hipLaunchKernelGGL((chol_step_kernel<64, 64>), dim3(1, B), dim3(NTH), 0, h->stream, a);
        } else if ((int64_t)nf * B <= quad_max)
            hipLaunchKernelGGL((chol_step_kernel<64, 64>), dim3(nf ? 8 + 4 * nf : 1, B), dim3(NTH), 0, h->stream, a);
        else if ((int64_t)nf * B <= half_max)
            hipLaunchKernelGGL((chol_step_kernel<128, 64>), dim3(8 + 2 * nf, B), dim3(NTH), 0, h->stream, a);
        else
            hipLaunchKernelGGL((chol_step_kernel<128, 128>), dim3(8 + nf, B), dim3(NTH), 0, h->stream, a);
        HIP_TRY(hipGetLastError());
        if (j + 1 < nb) {
            if (strip16)
                hipLaunchKernelGGL(panel_solve_kernel<16>, dim3(8 * (nb - j - 1), B), dim3(256), 0, h->stream, A, ld, j, nb,
                                   (const double*)dinvB);
            else
                hipLaunchKernelGGL(panel_solve_kernel<32>, dim3(4 * (nb - j - 1), B), dim3(256), 0, h->stream, A, ld, j, nb,
                                   (const double*)dinvB);
            HIP_TRY(hipGetLastError());
            if (old_diag) {
                GemmArgs g = nt_update(A, ld, P.d_tiles + P.diag[j].off, P.diag[j].n, h->np);
                GP_TRY(launch_gemm(h, false, false, EPI_STORE, g));
            } else {
                hipLaunchKernelGGL(diag_update_kernel, dim3(10 * P.diag[j].n, B), dim3(256), 0, h->stream, A, ld, j, nb);
                HIP_TRY(hipGetLastError());
            }
        }
    }
    return GPIMHIP_OK;
}

// The chain of one outer panel [p0, p1) whose columns have already received every update from the panels left of
// it (distributed factorisation: right-looking across panels, gpimhip_dist_panel_factor).  Same launches as above
// with the left-looking window restricted to the panel: colfill[j - p0] = tiles (i, j), i > j, k-blocks [p0, j).
// A is addressed with GLOBAL block indices (the caller shifts the base of a local column panel accordingly).
int launch_panel_chain(gpimhip_ctx* h, double* A, int64_t ld, int p0, int p1, int nb, const TileDesc* tiles,
                       const PlanRange* colfill, int32_t* info) {
    const int B = h->nbatch;
    for (int j = p0; j < p1; ++j) {
        StepArgs a;
        a.A = A; a.ld = ld; a.kblk = j; a.nb = nb;
        a.dinv_all = h->dinv; a.dinvB_all = h->dinvB; a.logdet = h->logdet_part; a.info = info;
        a.col_off = 0;
        const PlanRange f = colfill[j - p0];
        a.g = nt_update(A, ld, tiles + f.off, f.n, h->np);
        a.g.chunk = 1;
        if (f.n <= 128)
            hipLaunchKernelGGL((chol_step_kernel<64, 64>), dim3(f.n ? 8 + 4 * f.n : 1, B), dim3(NTH), 0, h->stream, a);
        else
            hipLaunchKernelGGL((chol_step_kernel<128, 64>), dim3(8 + 2 * f.n, B), dim3(NTH), 0, h->stream, a);
        HIP_TRY(hipGetLastError());
        if (j + 1 < nb) {
            hipLaunchKernelGGL(panel_solve_kernel<32>, dim3(4 * (nb - j - 1), B), dim3(256), 0, h->stream, A, ld, j, nb,
                               (const double*)h->dinvB);
            HIP_TRY(hipGetLastError());
        }
        if (j + 1 < p1) {
            hipLaunchKernelGGL(diag_update_kernel, dim3(10 * (p1 - j - 1), B), dim3(256), 0, h->stream, A, ld, j, nb);
            HIP_TRY(hipGetLastError());
        }
    }
    return GPIMHIP_OK;
}
