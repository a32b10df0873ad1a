// r5_trail_probe.hip -- what does one trailing-tile update of lds_factor_inv cost per wave and per SIMD?  (development aid)
// Waves of one 512-thread workgroup update tiles of a LayTri block in LDS in a loop; variants of the loop body.
#include "../gpim_amd/csrc/blocklds.hpp"
#include <stdio.h>
#include <string>
void gpim_set_error(const std::string&) {}
typedef LayTri PL;
// the operands of one trailing tile: fragments of L(rt, p) and L(ct, p), the tile itself
template <class Lay>
struct FiTrailOps {
    d4 a, b, c;
    __device__ __forceinline__ void load(const double* D, int p, int rt, int ct, int lane) {
        const int r = lane & 15, kq = lane >> 4;
        a = fi_frag<Lay>(D + Lay::tile(rt, p), r, kq);
        b = fi_frag<Lay>(D + Lay::tile(ct, p), r, kq);
        c = tile_read<Lay>(D + Lay::tile(rt, ct), lane);
    }
};
// V: 0 = FiTrailOps loop as in lds_factor_inv, 1 = plain fi_trail1 per tile, 2 = as 0 without the write-back,
//    3 = MFMAs only (operands loaded once), 4 = as 0 but two tiles per iteration (two accumulator chains)
template <int V>
__global__ __launch_bounds__(512) void probe(long long* out, unsigned wavemask, int ntile) {
    __shared__ double D[PL::DOUBLES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int e = tid; e < PL::DOUBLES; e += 512) D[e] = 1e-3 * (e % 97);
    __syncthreads();
    long long t0 = clock64();
    if ((wavemask >> wave) & 1) {
        const int p = 0;
        if (V == 0 || V == 2) {
            FiTrailOps<PL> cur, nxt;
            int rt = 1 + (wave % 7), ct = 1;
            cur.load(D, p, rt, ct, lane);
            for (int n = 0; n < ntile; ++n) {
                int rt1 = 1 + ((rt + 2) % 7), ct1 = 1 + (n % rt1 == 0 ? 0 : (n % rt1));
                if (ct1 > rt1) ct1 = rt1;
                nxt = cur;
                if (n + 1 < ntile) nxt.load(D, p, rt1, ct1, lane);
                d4 c = cur.c;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-cur.a[s4], cur.b[s4], c, 0, 0, 0);
                if (V == 0) tile_write<PL>(D + PL::tile(rt, ct), c, lane);
                else asm volatile("" :: "v"(c));
                cur = nxt; rt = rt1; ct = ct1;
            }
        } else if (V == 1) {
            int rt = 1 + (wave % 7), ct = 1;
            for (int n = 0; n < ntile; ++n) {
                fi_trail1<PL>(D, p, rt, ct, lane);
                rt = 1 + ((rt + 2) % 7); ct = 1 + (n % rt); if (ct > rt) ct = rt;
            }
        } else if (V == 3) {
            FiTrailOps<PL> cur;
            cur.load(D, p, 1 + (wave % 7), 1, lane);
            d4 c = cur.c;
            for (int n = 0; n < ntile; ++n) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) c = __builtin_amdgcn_mfma_f64_16x16x4f64(-cur.a[s4], cur.b[s4], c, 0, 0, 0);
            }
            tile_write<PL>(D + PL::tile(1 + (wave % 7), 1), c, lane);
        } else if (V == 4) {
            int rt = 1 + (wave % 7);
            for (int n = 0; n < ntile; n += 2) {
                int ct0 = 1, ct1 = rt > 1 ? 2 : 1;
                FiTrailOps<PL> x, y;
                x.load(D, p, rt, ct0, lane);
                y.load(D, p, rt, ct1, lane);
                d4 c0 = x.c, c1 = y.c;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x.a[s4], x.b[s4], c0, 0, 0, 0);
                    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x.a[s4], y.b[s4], c1, 0, 0, 0);
                }
                tile_write<PL>(D + PL::tile(rt, ct0), c0, lane);
                tile_write<PL>(D + PL::tile(rt, ct1), c1, lane);
                rt = 1 + ((rt + 2) % 7);
            }
        }
    }
    if (((wavemask >> wave) & 1) && V == 5) {
        // two independent tiles per iteration
        int rt = 1 + (wave % 7);
        for (int n = 0; n < ntile; n += 2) {
            int rt2 = 1 + ((rt + 3) % 7);
            FiTrailOps<PL> x, y;
            x.load(D, 0, rt, 1, lane);
            y.load(D, 0, rt2, 1, lane);
            d4 c0 = x.c, c1 = y.c;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x.a[s4], x.b[s4], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-y.a[s4], y.b[s4], c1, 0, 0, 0);
            }
            tile_write<PL>(D + PL::tile(rt, 1), c0, lane);
            tile_write<PL>(D + PL::tile(rt2, 1), c1, lane);
            rt = 1 + ((rt + 2) % 7);
        }
    }
    if (((wavemask >> wave) & 1) && V == 6) {
        // 2 x 2 tiles per iteration: rows rt, rt2, columns 1, 2
        int rt = 2 + (wave % 6);
        const int r = lane & 15, kq = lane >> 4;
        for (int n = 0; n < ntile; n += 4) {
            int rt2 = 2 + ((rt + 3) % 6);
            const d4 a0 = fi_frag<PL>(D + PL::tile(rt, 0), r, kq), a1 = fi_frag<PL>(D + PL::tile(rt2, 0), r, kq);
            const d4 b0 = fi_frag<PL>(D + PL::tile(1, 0), r, kq), b1 = fi_frag<PL>(D + PL::tile(2, 0), r, kq);
            d4 c00 = tile_read<PL>(D + PL::tile(rt, 1), lane), c01 = tile_read<PL>(D + PL::tile(rt, 2), lane);
            d4 c10 = tile_read<PL>(D + PL::tile(rt2, 1), lane), c11 = tile_read<PL>(D + PL::tile(rt2, 2), lane);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                c00 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[s4], b0[s4], c00, 0, 0, 0);
                c01 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a0[s4], b1[s4], c01, 0, 0, 0);
                c10 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[s4], b0[s4], c10, 0, 0, 0);
                c11 = __builtin_amdgcn_mfma_f64_16x16x4f64(-a1[s4], b1[s4], c11, 0, 0, 0);
            }
            tile_write<PL>(D + PL::tile(rt, 1), c00, lane);
            tile_write<PL>(D + PL::tile(rt, 2), c01, lane);
            tile_write<PL>(D + PL::tile(rt2, 1), c10, lane);
            tile_write<PL>(D + PL::tile(rt2, 2), c11, lane);
            rt = 2 + ((rt + 1) % 6);
        }
    }
    if (((wavemask >> wave) & 1) && V == 7) {
        // one tile per half-iteration, ping-pong register sets (no copies), the other set's loads in flight during the MFMAs
        int rt = 1 + (wave % 7);
        FiTrailOps<PL> x, y;
        x.load(D, 0, rt, 1, lane);
        for (int n = 0; n < ntile; n += 2) {
            int rt2 = 1 + ((rt + 3) % 7), rt3 = 1 + ((rt + 2) % 7);
            y.load(D, 0, rt2, 1, lane);
            d4 c0 = x.c;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(-x.a[s4], x.b[s4], c0, 0, 0, 0);
            x.load(D, 0, rt3, 1, lane);
            d4 c1 = y.c;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(-y.a[s4], y.b[s4], c1, 0, 0, 0);
            tile_write<PL>(D + PL::tile(rt, 1), c0, lane);
            tile_write<PL>(D + PL::tile(rt2, 1), c1, lane);
            rt = rt3;
        }
    }
    if (((wavemask >> wave) & 1) && V == 8) {
        // software pipeline without copies: first MFMA of tile n+1 in front of the write-back of tile n, loads of tile n+2 behind it
        FiTrailOps<PL> s0, s1;
        int rt = 1 + (wave % 7), rtn;
        int w0 = rt, w1 = 0;
        s0.load(D, 0, rt, 1, lane);
        rt = 1 + ((rt + 2) % 7); w1 = rt;
        s1.load(D, 0, rt, 1, lane);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) s0.c = __builtin_amdgcn_mfma_f64_16x16x4f64(-s0.a[s4], s0.b[s4], s0.c, 0, 0, 0);
        for (int n = 2; n < ntile; n += 2) {
            s1.c = __builtin_amdgcn_mfma_f64_16x16x4f64(-s1.a[0], s1.b[0], s1.c, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            tile_write<PL>(D + PL::tile(w0, 1), s0.c, lane);
            rt = 1 + ((rt + 2) % 7); w0 = rt;
            s0.load(D, 0, rt, 1, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s4 = 1; s4 < 4; ++s4) s1.c = __builtin_amdgcn_mfma_f64_16x16x4f64(-s1.a[s4], s1.b[s4], s1.c, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            s0.c = __builtin_amdgcn_mfma_f64_16x16x4f64(-s0.a[0], s0.b[0], s0.c, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            tile_write<PL>(D + PL::tile(w1, 1), s1.c, lane);
            rt = 1 + ((rt + 2) % 7); w1 = rt;
            s1.load(D, 0, rt, 1, lane);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s4 = 1; s4 < 4; ++s4) s0.c = __builtin_amdgcn_mfma_f64_16x16x4f64(-s0.a[s4], s0.b[s4], s0.c, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        tile_write<PL>(D + PL::tile(w0, 1), s0.c, lane);
        asm volatile("" :: "v"(s1.c));
    }
    if (((wavemask >> wave) & 1) && V >= 9 && V <= 13) {
        // as 8, LDS instructions spread between the MFMAs: M1 | write prev | M2 | next c | M3 | next a0 b0 a1 b1 | M4 | next a2 b2 a3 b3
        FiTrailOps<PL> s0, s1;
        const int r = lane & 15, kq = lane >> 4;
        int rt = 1 + (wave % 7);
        int w0 = rt, w1 = 0;
        s0.load(D, 0, rt, 1, lane);
        rt = 1 + ((rt + 2) % 7); w1 = rt;
        s1.load(D, 0, rt, 1, lane);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) s0.c = __builtin_amdgcn_mfma_f64_16x16x4f64(-s0.a[s4], s0.b[s4], s0.c, 0, 0, 0);
#define SB __builtin_amdgcn_sched_barrier(0)
#define HALF(CUR, PRV, WPRV)                                                                                        \
        CUR.c = __builtin_amdgcn_mfma_f64_16x16x4f64(-CUR.a[0], CUR.b[0], CUR.c, 0, 0, 0); SB;                      \
        if (V != 10 && V != 13) tile_write<PL>(D + PL::tile(WPRV, 1), PRV.c, lane); else asm volatile("" :: "v"(PRV.c)); SB; \
        rt = 1 + ((rt + 2) % 7); WPRV = rt;                                                                         \
        CUR.c = __builtin_amdgcn_mfma_f64_16x16x4f64(-CUR.a[1], CUR.b[1], CUR.c, 0, 0, 0); SB;                      \
        if (V != 11 && V != 13) PRV.c = tile_read<PL>(D + PL::tile(rt, 1), lane); SB;                                      \
        CUR.c = __builtin_amdgcn_mfma_f64_16x16x4f64(-CUR.a[2], CUR.b[2], CUR.c, 0, 0, 0); SB;                      \
        if (V < 12) { PRV.a[0] = D[PL::tile(rt, 0) + PL::in(r, kq)]; PRV.b[0] = D[PL::tile(1, 0) + PL::in(r, kq)];                \
        PRV.a[1] = D[PL::tile(rt, 0) + PL::in(r, 4 + kq)]; PRV.b[1] = D[PL::tile(1, 0) + PL::in(r, 4 + kq)]; } SB;  \
        CUR.c = __builtin_amdgcn_mfma_f64_16x16x4f64(-CUR.a[3], CUR.b[3], CUR.c, 0, 0, 0); SB;                      \
        if (V < 12) { PRV.a[2] = D[PL::tile(rt, 0) + PL::in(r, 8 + kq)]; PRV.b[2] = D[PL::tile(1, 0) + PL::in(r, 8 + kq)];        \
        PRV.a[3] = D[PL::tile(rt, 0) + PL::in(r, 12 + kq)]; PRV.b[3] = D[PL::tile(1, 0) + PL::in(r, 12 + kq)]; } SB;
        for (int n = 2; n < ntile; n += 2) {
            HALF(s1, s0, w0)
            HALF(s0, s1, w1)
        }
        tile_write<PL>(D + PL::tile(w0, 1), s0.c, lane);
        asm volatile("" :: "v"(s1.c));
    }
    if (((wavemask >> wave) & 1) && V >= 20 && V <= 24) {
        // one accumulation chain, `ntile` 16-deep products: 20 = plain loop with 8 ds_read_b64, 21 = two register sets taking
        // turns, 22 = plain with 4 ds_read_b128 (fragment-major tiles), 23 = as 22 with two sets, 24 = two chains sharing A, b128
        const int r = lane & 15, kq = lane >> 4;
        d4 t = (d4){0.0, 0.0, 0.0, 0.0}, t2 = t;
        const d2* F = reinterpret_cast<const d2*>(D);
        auto ld64 = [&](int tile, d4& a, d4& b) { a = fi_frag<PL>(D + PL::tile(7, tile), r, kq); b = fi_fragT<PL>(D + PL::tile(tile, 0), r, kq); };
        auto ld128 = [&](int tile, d4& a, d4& b) {
            const d2 a0 = F[(PL::tile(7, tile) >> 1) + lane], a1 = F[(PL::tile(7, tile) >> 1) + 64 + lane];
            const d2 b0 = F[(PL::tile(tile, 0) >> 1) + lane], b1 = F[(PL::tile(tile, 0) >> 1) + 64 + lane];
            a = (d4){a0[0], a0[1], a1[0], a1[1]}; b = (d4){b0[0], b0[1], b1[0], b1[1]};
        };
        if (V == 20 || V == 22) {
            for (int n = 0; n < ntile; ++n) {
                d4 a, b;
                if (V == 20) ld64(n % 7, a, b); else ld128(n % 7, a, b);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) t = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s4], b[s4], t, 0, 0, 0);
            }
        } else if (V == 21 || V == 23) {
            d4 a0, b0, a1, b1;
            if (V == 21) ld64(0, a0, b0); else ld128(0, a0, b0);
            for (int n = 0; n < ntile; n += 2) {
                if (V == 21) ld64((n + 1) % 7, a1, b1); else ld128((n + 1) % 7, a1, b1);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) t = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[s4], b0[s4], t, 0, 0, 0);
                if (V == 21) ld64((n + 2) % 7, a0, b0); else ld128((n + 2) % 7, a0, b0);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) t = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[s4], b1[s4], t, 0, 0, 0);
            }
        } else {
            for (int n = 0; n < ntile; n += 2) {
                d4 a, b, b2, dummy;
                ld128(n % 7, a, b);
                ld128((n + 3) % 7, dummy, b2);
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    t = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s4], b[s4], t, 0, 0, 0);
                    t2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s4], b2[s4], t2, 0, 0, 0);
                }
            }
        }
        tile_write<PL>(D + PL::tile(7, 7), t + t2, lane);
    }
    long long t1 = clock64();
    if (lane == 0) out[wave] = t1 - t0;
    __syncthreads();
    if (tid == 0) out[8] = clock64() - t0;
}
template <int V>
void run(const char* name, unsigned mask, int ntile, long long* d) {
    long long h[9];
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe<V>, dim3(1), dim3(512), 0, 0, d, mask, ntile);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    }
    int nw = __builtin_popcount(mask);
    long long mx = 0;
    for (int w = 0; w < 8; ++w) if ((mask >> w) & 1) mx = h[w] > mx ? h[w] : mx;
    printf("%-44s waves 0x%02x: %6.0f cycles per tile per wave (slowest), %6.0f per tile overall\n", name, mask, (double)mx / ntile,
           (double)mx / (ntile * nw));
}
int main() {
    long long* d; hipMalloc(&d, 9 * 8);
    const int nt = 64;
    for (unsigned mask : {0x02u, 0x22u, 0xeeu}) {
        run<0>("prefetching loop (as shipped)", mask, nt, d);
        run<1>("plain load-mfma-store per tile", mask, nt, d);
        run<2>("prefetching loop, no write-back", mask, nt, d);
        run<3>("MFMAs only", mask, nt, d);
        run<4>("two tiles per iteration sharing the A fragments", mask, nt, d);
        run<5>("two independent tiles per iteration", mask, nt, d);
        run<6>("2 x 2 tiles per iteration", mask, nt, d);
        run<7>("ping-pong register sets", mask, nt, d);
        run<8>("pipelined, MFMA of n+1 before write of n", mask, nt, d);
        run<9>("pipelined, LDS instructions between the MFMAs", mask, nt, d);
        run<20>("chain: plain loop, 8 ds_read_b64 per product", mask, nt, d);
        run<21>("chain: two register sets, b64", mask, nt, d);
        run<22>("chain: plain loop, 4 ds_read_b128 per product", mask, nt, d);
        run<23>("chain: two register sets, b128", mask, nt, d);
        run<24>("two chains sharing A, b128 (3 reads per product)", mask, nt, d);
        run<10>("  ... without the write-back", mask, nt, d);
        run<11>("  ... without the loads of C", mask, nt, d);
        run<12>("  ... without the loads of A and B", mask, nt, d);
        run<13>("  ... without any LDS instruction", mask, nt, d);
    }
    return 0;
}
