// step_plan_check.hip -- host-only check of the hosting plan of cholstep.hip (no GPU needed): every tile (i, jj), i >= jj,
// must receive every source block column c < jj exactly once, in increasing order -- bulk flushes [kb0, kb1) first, then the
// left-looking column update of the window (off-diagonal) or the diagonal updates; prints the simulated rounds per launch.
//   hipcc --offload-arch=gfx950 -O2 -std=c++17 -o tools/step_plan_check tools/step_plan_check.hip && tools/step_plan_check 128
#include "../gpim_amd/csrc/cholstep.hip"
#include <stdio.h>
void gpim_set_error(const std::string&) {}
int launch_gemm(gpimhip_ctx*, bool, bool, int, const GemmArgs&) { return 0; }
int launch_potrf_steps_f32(gpimhip_ctx*, double*, int64_t, int64_t, int32_t*, int) { return 0; }
int main(int argc, char** argv) {
    const int nb = argc > 1 ? atoi(argv[1]) : 128, verbose = argc > 2 ? atoi(argv[2]) : 0;
    StepPlan P;
    std::vector<TileDesc> tl;
    P.fill.assign(nb, {0, 0});
    P.diag.assign(nb, {0, 0});
    step_plan_hosted(nb, tl, P);
    // cnt[(i * nb + jj) * nb + c] = how often tile (i, jj) has received source block column c
    std::vector<unsigned char> cnt((size_t)nb * nb * nb, 0);
    long bad = 0, nt = 0;
    double cost_total = 0, span_total = 0;
    for (int j = 0; j < nb; ++j) {
        std::vector<char> seen((size_t)nb * nb, 0);
        HostSim sim(512);
        sim.add(1.0);
        int dmax = 0, n8 = 0, n4 = 0;
        for (int q = 0; q < P.fill[j].n; ++q) {
            const TileDesc t = tl[P.fill[j].off + q];
            if (t.ci < t.cj || t.kb0 >= t.kb1 || t.kb1 > t.cj || t.kb1 > j || t.cj < j) { if (bad < 12) printf("bad tile step %d: (%d,%d) [%d,%d)\n", j, t.ci, t.cj, t.kb0, t.kb1); ++bad; continue; }
            if (seen[(size_t)t.ci * nb + t.cj]++) ++bad;                         // same output tile twice in one launch
            for (int c = t.kb0; c < t.kb1; ++c) cnt[((size_t)t.ci * nb + t.cj) * nb + c]++;
            sim.add(HostSim::cost(t));
            cost_total += HostSim::cost(t);
            dmax = std::max(dmax, t.kb1 - t.kb0);
            (t.kb1 - t.kb0 >= 8 ? n8 : n4)++;
            ++nt;
        }
        span_total += std::max(sim.makespan, 1.0);
        if (verbose) printf("step %3d: %5d tiles (%4d deep, %4d shallow, max depth %2d)  makespan %6.1f  fill %.2f\n", j, P.fill[j].n, n8, n4, dmax, sim.makespan, [&]{ double c = 0; for (int q = 0; q < P.fill[j].n; ++q) c += HostSim::cost(tl[P.fill[j].off + q]); return c / 512 / sim.makespan; }());
        for (int q = 0; q < P.diag[j].n; ++q) {
            const TileDesc t = tl[P.diag[j].off + q];
            if (t.ci != t.cj || t.kb0 != j || t.kb1 != j + 1 || t.ci <= j) { ++bad; continue; }
            cnt[((size_t)t.ci * nb + t.cj) * nb + j]++;
        }
    }
    for (int i = 0; i < nb; ++i)
        for (int jj = 0; jj <= i; ++jj)
            for (int c = 0; c < jj; ++c)
                if (cnt[((size_t)i * nb + jj) * nb + c] != 1) { if (bad < 12) printf("tile (%d,%d) source %d: %d times\n", i, jj, c, cnt[((size_t)i * nb + jj) * nb + c]); ++bad; }
    printf("nb %d: %ld hosted tiles, %ld errors; simulated span %.0f units vs work / 512 = %.0f (%.1f %%)\n", nb, nt, bad, span_total,
           cost_total / 512, 100.0 * cost_total / 512 / span_total);
    return bad != 0;
}
