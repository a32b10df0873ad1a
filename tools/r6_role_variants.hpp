// r6_role_variants.hpp -- two schedules of the diagonal-block role that were built and measured in round 6 and NOT kept
// (docs/experiments.md, "round 6: the diagonal-block role"): development aid, not part of the product.  Included by
// gpim_amd/csrc/potf2_body.hpp only when FI_ROLE_VARIANTS is defined (tools/potf2_prof.hip -DFI_ROLE_VARIANTS
// -DFI_LOOKAHEAD=1 [-DFI_PIPE=1], tools/ab/build1.sh ... cholstep -DFI_ROLE_VARIANTS -DFI_LOOKAHEAD=1).
//
//   lds_factor_inv_la    one barrier per 16-column step, the workers' update one step behind wave 0's chain ("look-ahead")
//   fi_trail_pipelined   the trailing items of a wave software-pipelined (two register sets)
//
// Both give the bits of lds_factor_inv (tools/potf2_prof.hip prints a hash of L and of its inverse).  MI355X, cycles of the
// eight steps: lds_factor_inv 46.5K; look-ahead 46.3K (first version), 49.1K (tile counts evened out, solves before the
// inverse when X_q is ready); look-ahead + pipelined items 53.0K.  The role is bound by what its six worker waves get
// through -- ~850 cycles per tile and wave against 256 of MFMA issue -- not by where its barriers stand.
#pragma once

// The trailing items of one wave, software-pipelined (round 6): the operand fragments of item n + 1 are in flight while
// item n multiplies -- two register sets taking turns, no copies.  fi_trail2 alone is a chain of  LDS latency -> eight
// MFMAs -> drain -> stores  per item, and with two waves per SIMD the matrix pipe idles whenever both wait (measured: ~850
// cycles per tile and wave in the first step, 256 of them MFMA issue).  Needs ~100 more registers than the plain loop: used
// where the role's workgroup has a CU to itself (cholstep.hip: ALONE, 256 registers per lane).  An item with one tile
// runs its tile twice (the second result is dropped).  The same operations on every tile in the same order: the same bits.
struct FiItem {
    d4 a0, a1, b0, b1, c0, c1;
    int o0, o1;        // offsets of the result tiles in D
    bool two;
};
template <class Lay>
__device__ __forceinline__ void fi_item_load(FiItem& R, const double* D, int p, int it, int r, int kq, int rr) {
    const int rtA = it >> 9, ctA = (it >> 6) & 7;
    int rtB = (it >> 3) & 7, ctB = it & 7;
    R.two = rtB != 0;
    if (!R.two) { rtB = rtA; ctB = ctA; }
    R.b0 = fi_frag<Lay>(D + Lay::tile(rtA, p), r, kq);
    R.b1 = fi_frag<Lay>(D + Lay::tile(rtB, p), r, kq);
    R.a0 = fi_frag<Lay>(D + Lay::tile(ctA, p), rr, kq);
    R.a1 = fi_frag<Lay>(D + Lay::tile(ctB, p), rr, kq);
    R.o0 = Lay::tile(rtA, ctA);
    R.o1 = Lay::tile(rtB, ctB);
    R.c0 = fi_frag<Lay>(D + R.o0, r, kq);
    R.c1 = fi_frag<Lay>(D + R.o1, r, kq);
}
template <class Lay>
__device__ __forceinline__ void fi_item_run(FiItem& R, double* D, int r, int kq) {
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) {
        R.c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-R.a0[s4], R.b0[s4], R.c0, 0, 0, 0);
        R.c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-R.a1[s4], R.b1[s4], R.c1, 0, 0, 0);
    }
    fi_frag_store<Lay>(D + R.o0, r, kq, R.c0);
    if (R.two) fi_frag_store<Lay>(D + R.o1, r, kq, R.c1);
}
// up to four items (12 bits each from bit 12 of `plan`, count in bits 8-11); after_first(): called once the first item's
// results are in LDS
template <class Lay, typename F>
__device__ __forceinline__ void fi_trail_pipelined(double* D, int p, unsigned long long plan, int lane, F after_first) {
    const int r = lane & 15, kq = lane >> 4, rr = fi_rho(r);
    const int ntr = (int)(plan >> 8) & 15;
    FiItem S0, S1;
    auto item = [&](int n) { return (int)(plan >> (12 + 12 * n)) & 0xFFF; };
    if (ntr > 0) fi_item_load<Lay>(S0, D, p, item(0), r, kq, rr);
    if (ntr > 1) fi_item_load<Lay>(S1, D, p, item(1), r, kq, rr);
    if (ntr > 0) { fi_item_run<Lay>(S0, D, r, kq); after_first(); }
    if (ntr > 2) fi_item_load<Lay>(S0, D, p, item(2), r, kq, rr);
    if (ntr > 1) fi_item_run<Lay>(S1, D, r, kq);
    if (ntr > 3) fi_item_load<Lay>(S1, D, p, item(3), r, kq, rr);
    if (ntr > 2) fi_item_run<Lay>(S0, D, r, kq);
    if (ntr > 3) fi_item_run<Lay>(S1, D, r, kq);
}

// ------------------------------------------------------------------------------------------------------------------------
// Round 6: the same factorisation + inverse of a 128 x 128 block (npan = 8, LayTri) with ONE barrier per 16-column step.
//
// lds_factor_inv above runs a step as  solve phase | barrier | update phase | barrier:  wave 0's chain (solve its tile ->
// update the next diagonal tile -> factor it, 4.1K cycles) and the workers' update phase (3.4 - 4.5K) overlap, but the solve
// phase (0.8 - 1.5K with its barrier) lies in front of both -- tools/potf2_prof.hip, round 6: steps of 4.5 - 6.2K cycles,
// 46.8K for eight.  Here the workers' work is shifted by one step against the chain ("look-ahead"):
//
//   iteration q (between two barriers), q = 0 .. 8
//     wave 0   : factor diagonal tile q (its inverse X_q -> Xs[q & 1]; flag_x = q) -> solve tile (q+1, q) -> update diagonal
//                tile (q+1, q+1) with it                     (needs tiles (q+1, q), (q+1, q+1) updated with column q-1: flag_u)
//     workers  : store the tiles of row q-2 of the inverse they hold (count_k)
//                -> trailing update with column q-1 (first (q+1, q) and (q+1, q+1): flag_u = q)
//                -> row q-1 of the inverse (needs every worker's row q-2 in LDS: count_k)
//                -> solve their tiles (t, q), t >= q+2 -- each solved by the wave that has just updated it (needs X_q: flag_x)
//     wave 4   : exports block row q-1 of L, stores X(q-1, q-1) over L(q-1, q-1)
//
// so that an iteration costs max(chain, workers' update + solve) + one barrier.  The three flags live in LDS; a wave
// spins on one with s_sleep (bounded: a broken schedule yields a wrong block, which the tests see, never a hung GPU).
// Every producer runs without waiting for its consumer (no cycle).  Every tile goes through the same operations in the same
// order as in lds_factor_inv: the same bits.
// ------------------------------------------------------------------------------------------------------------------------
struct FiPlanLA {
    unsigned long long w[9][7];      // as FiPlan::w: inverse tiles of row q-1 (bits 0-7), trailing items of column q-1
    unsigned char s[9][7];           // bits 0-2 / 3-5: rows t of the tiles (t, q) this worker solves (0: none); bit 6: set flag_u after its first item
};
constexpr FiPlanLA make_fi_plan_la() {
    FiPlanLA P{};
    for (int q = 0; q <= 8; ++q) {
        int load[7] = {0, 0, 0, 0, 0, 0, 0}, ninv[7] = {0, 0, 0, 0, 0, 0, 0}, ntr[7] = {0, 0, 0, 0, 0, 0, 0}, nsol[7] = {0, 0, 0, 0, 0, 0, 0};
        unsigned long long word[7] = {0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF, 0xFF};
        unsigned char sol[7] = {0, 0, 0, 0, 0, 0, 0};
        const int nw = (q == 8) ? 7 : 6;                 // wave 0 (worker 6) joins for the last row of the inverse
        // load in units of one LDS-fed MFMA product: an inverse product 3, a trailing tile 4, a solve 4
        auto simd_load = [&](int w) { return (w == 6) ? load[6] : load[w % 3] + load[w % 3 + 3]; };
        auto least = [&](auto ok) {
            int best = -1;
            for (int w = 0; w < nw; ++w) {
                if (!ok(w)) continue;
                if (best < 0 || simd_load(w) < simd_load(best) || (simd_load(w) == simd_load(best) && load[w] < load[best])) best = w;
            }
            return best;
        };
        auto add_item = [&](int w, int rtA, int ctA, int rtB, int ctB) {
            word[w] |= (unsigned long long)((rtA << 9) | (ctA << 6) | (rtB << 3) | ctB) << (12 + 12 * ntr[w]);
            ++ntr[w];
            load[w] += (rtB ? 8 : 4);
        };
        // row r = q - 1 of the inverse: tiles j < r, cost (r - j) + 1 products
        if (q >= 2) {
            const int r = q - 1;
            for (int j = 0; j < r; ++j) {
                const int w = least([&](int v) { return ninv[v] < 2; });
                word[w] = (word[w] & ~(0xFull << (4 * ninv[w]))) | ((unsigned long long)j << (4 * ninv[w]));
                ++ninv[w];
                load[w] += 3 * ((r - j) + 1);
            }
        }
        if (q == 0) {
            for (int t = 2; t <= 7; ++t) sol[t - 2] = (unsigned char)t;          // column 0: nothing to update first
        } else if (q <= 7) {
            // trailing update with column q - 1: tiles (rt, ct), q <= ct <= rt <= 7, without (q, q) (wave 0's)
            if (q + 1 <= 7) {
                // what wave 0 waits for: (q+1, q) and (q+1, q+1), one shared-row pair, first item of its worker
                const int w = least([&](int) { return true; });
                add_item(w, q + 1, q, q + 1, q + 1);
                sol[w] |= 0x40;
            }
            // the tiles of column q below that, each with its right neighbour (same block row): solved by their worker
            for (int t = q + 2; t <= 7; ++t) {
                const int w = least([&](int v) { return nsol[v] < 2 && ntr[v] < 4; });
                add_item(w, t, q, t, q + 1);
                sol[w] |= (unsigned char)(t << (3 * nsol[w]));
                ++nsol[w];
                load[w] += 4;
            }
            // the rest, row-major: dealt one tile at a time to the least-loaded wave (tile counts differ by at most one per
            // wave; pairs of unequal count -- 2 + 2 + 2 against 2 + 2 items -- measured 3.2K against 5.0K cycles of trailing
            // work in one iteration), then paired inside a wave's list (neighbours mostly share their block row)
            int tiles[32] = {}, nt = 0, mine[7][16] = {}, nm[7] = {0, 0, 0, 0, 0, 0, 0};
            for (int rt = 7; rt >= q + 2; --rt)
                for (int ct = q + 2; ct <= rt; ++ct) tiles[nt++] = (rt << 3) | ct;
            // contiguous runs: how many tiles each wave takes
            int cnt[7] = {0, 0, 0, 0, 0, 0, 0}, wl[7] = {0, 0, 0, 0, 0, 0, 0};
            for (int w = 0; w < 6; ++w) wl[w] = load[w];
            for (int i = 0; i < nt; ++i) {
                int best = -1;
                for (int w = 0; w < 6; ++w) {
                    if (2 * ntr[w] + cnt[w] >= 8) continue;            // four items at most
                    if (best < 0) { best = w; continue; }
                    const int sb = wl[best % 3] + wl[best % 3 + 3], sw = wl[w % 3] + wl[w % 3 + 3];
                    if (wl[w] < wl[best] || (wl[w] == wl[best] && sw < sb)) best = w;
                }
                ++cnt[best];
                wl[best] += 4;
            }
            int at = 0;
            for (int w = 0; w < 6; ++w) {
                for (int i = 0; i < cnt[w]; ++i) mine[w][nm[w]++] = tiles[at + i];
                at += cnt[w];
                int i = 0;
                for (; i + 1 < nm[w]; i += 2) add_item(w, mine[w][i] >> 3, mine[w][i] & 7, mine[w][i + 1] >> 3, mine[w][i + 1] & 7);
                if (i < nm[w]) add_item(w, mine[w][i] >> 3, mine[w][i] & 7, 0, 0);
            }
        }
        for (int w = 0; w < 7; ++w) {
            P.w[q][w] = word[w] | ((unsigned long long)ntr[w] << 8);
            P.s[q][w] = sol[w];
        }
    }
    return P;
}
static __constant__ FiPlanLA c_fi_plan_la = make_fi_plan_la();

// The flags are LDS words.  A wave's LDS instructions execute in order, so a flag written after data is seen after the data,
// and data read after a flag is read after it: what is needed is that the COMPILER keeps the order (the asm statements) and
// that the reader has the flag's value before it branches (lgkmcnt).  No workgroup-scope fence: that would also wait for
// the wave's outstanding GLOBAL stores (the exported tiles of L), a round trip to memory on the critical path.
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ void fi_flag_set(int* f, int v) {
    asm volatile("" ::: "memory");
    *reinterpret_cast<volatile lds_int*>((lds_int*)f) = v;
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ bool fi_flag_ready(const int* f, int v) {
    const int got = *reinterpret_cast<const volatile lds_int*>((const lds_int*)f);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    return __builtin_amdgcn_readfirstlane(got) >= v;
}
__device__ __forceinline__ void fi_flag_wait(const int* f, int v) {
    int spins = 0;
    while (!fi_flag_ready(f, v)) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) break;                               // (never: see the header)
    }
}
__device__ __forceinline__ void fi_count_add(int* f) {
    asm volatile("" ::: "memory");
    __hip_atomic_fetch_add((lds_int*)f, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}

// flags: three ints of LDS, zero on entry ([0] flag_x + 1 = number of diagonal tiles factored, [1] flag_u, [2] count_k).
// The first diagonal tile has been factored by the caller (load_block_chol0).  Leaves D, invd and the sink as
// lds_factor_inv<Sink, true, LayTri>(D, invd, Xs, 8, ...) does.
template <typename Sink>
__device__ __forceinline__ void lds_factor_inv_la(double* D, double* invd, double* Xs, int* s_bad, int* flags, int tid, Sink sink) {
    typedef LayTri Lay;
    constexpr int npan = 8;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, kq = lane >> 4;
    const d4 zero = (d4){0.0, 0.0, 0.0, 0.0};
    const int wid = (wave == 0) ? 6 : (wave < 4) ? wave - 1 : wave - 2;      // worker number (wave 4: none)
    int* flag_x = flags;
    int* flag_u = flags + 1;
    int* count_k = flags + 2;
    d4 keep[2] = {zero, zero};       // tiles of a block row of X, carried into the next iteration
    unsigned long long held = 0xFF;
    if (tid == 0) fi_flag_set(flag_x, 1);                 // X_0 is there
    for (int q = 0; q <= npan; ++q) {
        __syncthreads();
        FSTAMP(8 * (q < 8 ? q : 7) + 0);
        if (wave == 4) {
            if (q >= 1) {
                // block row q-1 of L (final since iteration q-1) goes back to HBM; then X(q-1, q-1) takes the diagonal tile's place
                sink.row(q - 1, lane);
                const double* Xq = Xs + ((q - 1) & 1) * 16 * XS_LD;
                d4 xd;
#pragma unroll
                for (int g = 0; g < 4; ++g) xd[g] = Xq[r * XS_LD + kq + 4 * g];
                tile_write<Lay>(D + Lay::tile(q - 1, q - 1), xd, lane);
            }
            WSTAMP(q < 8 ? q : 7, 0);
            continue;
        }
        if (wave == 0 && q < npan) {
            // ---- the chain
            const double* Xp = Xs + (q & 1) * 16 * XS_LD;
            if (q >= 1) {
                const int bad = chol16_lp<Lay, false>(D + Lay::tile(q, q), invd + 16 * q, lane, Xs + (q & 1) * 16 * XS_LD, XS_LD);
                if (lane == 0 && bad && *s_bad == 0) *s_bad = 16 * q + bad;
                fi_flag_set(flag_x, q + 1);
            }
            FSTAMP(8 * q + 1);
            if (q + 1 < npan) {
                if (q >= 1) fi_flag_wait(flag_u, q);
                FSTAMP(8 * q + 2);
                double* C = D + Lay::tile(q + 1, q);
                const d4 a = fi_frag<Lay>(C, r, kq);
                const d4 x = fi_frag_xs(Xp, fi_rho(r), kq);
                d4 acc = zero;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s4], a[s4], acc, 0, 0, 0);
                fi_frag_store<Lay>(C, r, kq, acc);
                FSTAMP(8 * q + 3);
                double* Cd = D + Lay::tile(q + 1, q + 1);
                d4 c = tile_read<Lay>(Cd, lane);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-acc[s4], acc[s4], c, 0, 0, 0);
                tile_write<Lay>(Cd, c, lane);
            }
            FSTAMP(8 * q + 4);
            continue;
        }
        // ---- workers (wave 0 in the last iteration: row 7 of the inverse)
        const unsigned long long plan = c_fi_plan_la.w[q][wid];
        const int sol = c_fi_plan_la.s[q][wid];
        // the tiles of block row q-2 of X held since the previous iteration
#pragma unroll
        for (int cnt = 0; cnt < 2; ++cnt) {
            const int j = (int)(held >> (4 * cnt)) & 15;
            if (j != 15) fi_xstore<Lay>(D + Lay::tile(q - 2, j), keep[cnt], lane);
        }
        held = 0xFF;
        if (q >= 2 && wave != 0 && lane == 0) fi_count_add(count_k);
        // trailing update with column q-1
#if defined(FI_PIPE) && FI_PIPE
        fi_trail_pipelined<Lay>(D, q - 1, plan, lane, [&] { if (sol & 0x40) fi_flag_set(flag_u, q); });
#else
        {
            const int ntr = (int)(plan >> 8) & 15;
            for (int n = 0; n < ntr; ++n) {
                const int it = (int)(plan >> (12 + 12 * n)) & 0xFFF;
                const int rtA = it >> 9, ctA = (it >> 6) & 7, rtB = (it >> 3) & 7, ctB = it & 7;
                if (rtB == 0) fi_trail1<Lay>(D, q - 1, rtA, ctA, lane);
                else if (rtA == rtB) fi_trail2<Lay, true>(D, q - 1, rtA, ctA, rtB, ctB, lane);
                else fi_trail2<Lay, false>(D, q - 1, rtA, ctA, rtB, ctB, lane);
                if (n == 0 && (sol & 0x40)) fi_flag_set(flag_u, q);
            }
        }
#endif
        WSTAMP(q < 8 ? q : 7, 2);
        // the solves (below) as soon as X_q is there: before this wave's tiles of the inverse if wave 0 has finished its
        // 16x16 factorisation by now (the solved tiles are what the NEXT iteration starts from), after them otherwise
        const bool solve_first = (sol & 0x3F) && fi_flag_ready(flag_x, q + 1);
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
        if ((pass == 0) != solve_first) {
        // row q-1 of the inverse: this worker's tiles
        if ((plan & 15) != 15) {
            fi_flag_wait(count_k, 6 * (q - 1));
            held = plan & 0xFF;
            fi_inv_pair<Lay>(D, Xs + ((q - 1) & 1) * 16 * XS_LD, q - 1, (int)plan & 15, (int)(plan >> 4) & 15, lane, keep[0], keep[1]);
        }
        WSTAMP(q < 8 ? q : 7, 1);
        } else {
        // solve the tiles (t, q) this wave has just updated:  S = A X_q^T
        if (sol & 0x3F) {
            fi_flag_wait(flag_x, q + 1);
            const double* Xp = Xs + (q & 1) * 16 * XS_LD;
            const d4 x = fi_frag_xs(Xp, fi_rho(r), kq);
#pragma unroll
            for (int cnt = 0; cnt < 2; ++cnt) {
                const int t = (sol >> (3 * cnt)) & 7;
                if (t == 0) continue;
                double* C = D + Lay::tile(t, q);
                const d4 a = fi_frag<Lay>(C, r, kq);
                d4 acc = zero;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x[s4], a[s4], acc, 0, 0, 0);
                fi_frag_store<Lay>(C, r, kq, acc);
                sink.tile(t, q, acc, lane);
            }
        }
        WSTAMP(q < 8 ? q : 7, 3);
        }
        }
    }
    __syncthreads();
    // row 7 of the inverse (held in registers since the last iteration; wave 4 has stored X(7, 7) in it)
    if (wave != 4) {
#pragma unroll
        for (int cnt = 0; cnt < 2; ++cnt) {
            const int j = (int)(held >> (4 * cnt)) & 15;
            if (j != 15) fi_xstore<Lay>(D + Lay::tile(npan - 1, j), keep[cnt], lane);
        }
    }
    __syncthreads();
}
