// r6_plan_la_check.hip -- host-side check of the look-ahead plan of the diagonal-block role (blocklds.hpp: make_fi_plan_la):
// every trailing tile exactly once per step, the solver of a tile is the wave that updated it, what wave 0 waits for comes
// first, every tile of a row of the inverse exactly once.  hipcc tools/r6_plan_la_check.hip -o /tmp/plan_check && /tmp/plan_check
#include "../gpim_amd/csrc/potf2_body.hpp"
#include <stdio.h>
#include <string>
void gpim_set_error(const std::string&) {}
int main() {
    constexpr FiPlanLA P = make_fi_plan_la();
    int bad = 0;
    for (int q = 0; q <= 8; ++q) {
        int seen[8][8] = {};
        int inv[8] = {};
        int solved[8] = {};
        int nflag = 0, maxload = 0, minload = 1 << 30;
        for (int w = 0; w < 7; ++w) {
            const unsigned long long word = P.w[q][w];
            const int s = P.s[q][w];
            const int ntr = (int)(word >> 8) & 15;
            int load = 0;
            if (ntr > 4) { printf("q %d w %d: %d items\n", q, w, ntr); ++bad; }
            bool has[8][8] = {};
            for (int n = 0; n < ntr; ++n) {
                const int it = (int)(word >> (12 + 12 * n)) & 0xFFF;
                const int rtA = it >> 9, ctA = (it >> 6) & 7, rtB = (it >> 3) & 7, ctB = it & 7;
                ++seen[rtA][ctA]; has[rtA][ctA] = true; load += 4;
                if (rtB) { ++seen[rtB][ctB]; has[rtB][ctB] = true; load += 4; }
                if (n == 0 && (s & 0x40)) {
                    ++nflag;
                    if (!(rtA == q + 1 && ctA == q && rtB == q + 1 && ctB == q + 1)) { printf("q %d: flag item is (%d,%d),(%d,%d)\n", q, rtA, ctA, rtB, ctB); ++bad; }
                }
            }
            for (int c = 0; c < 2; ++c) {
                const int j = (int)(word >> (4 * c)) & 15;
                if (j != 15) { ++inv[j]; load += 3 * ((q - 1 - j) + 1); }
                const int t = (s >> (3 * c)) & 7;
                if (t) {
                    ++solved[t]; load += 4;
                    if (q >= 1 && !has[t][q]) { printf("q %d w %d solves (%d,%d) without having updated it\n", q, w, t, q); ++bad; }
                }
            }
            if (w < 6) { maxload = load > maxload ? load : maxload; minload = load < minload ? load : minload; }
            if (w == 6 && q < 8 && (ntr || (word & 0xFF) != 0xFF || (s & 0x3F))) { printf("q %d: wave 0 has worker items\n", q); ++bad; }
        }
        for (int rt = 0; rt < 8; ++rt)
            for (int ct = 0; ct <= rt; ++ct) {
                const bool want = q >= 1 && q <= 7 && ct >= q && !(rt == q && ct == q);
                if (seen[rt][ct] != (want ? 1 : 0)) { printf("q %d: tile (%d,%d) updated %d times, expected %d\n", q, rt, ct, seen[rt][ct], want); ++bad; }
            }
        for (int j = 0; j < 8; ++j) {
            const bool want = q >= 2 && j < q - 1;
            if (inv[j] != (want ? 1 : 0)) { printf("q %d: inverse tile %d assigned %d times\n", q, j, inv[j]); ++bad; }
        }
        for (int t = 0; t < 8; ++t) {
            const bool want = q <= 6 && t >= q + 2;
            if (solved[t] != (want ? 1 : 0)) { printf("q %d: tile (%d,%d) solved %d times\n", q, t, q, solved[t]); ++bad; }
        }
        if (nflag != ((q >= 1 && q <= 6) ? 1 : 0)) { printf("q %d: %d flag items\n", q, nflag); ++bad; }
        printf("iteration %d: worker load (MFMA-product units) min %d max %d\n", q, minload, maxload);
    }
    printf(bad ? "PLAN BAD (%d)\n" : "PLAN OK\n", bad);
    return bad != 0;
}
