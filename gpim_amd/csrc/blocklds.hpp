// blocklds.hpp -- device helpers for a (<=128)x(<=128) lower-triangular block resident in LDS:
// 16x16 register Cholesky / triangular inverse by one wave, MFMA tile access, and the panel-wise
// factorisation + recursive-doubling inverse built from them.  Shared by potf2.hip (diagonal steps of
// the blocked Cholesky) and smalln.hip (fused trainer for N <= 128).  512-thread workgroups.
#pragma once
#include "common.hpp"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define LDD 130   // row stride of the LDS block: 130 % 32 == 2 keeps MK fragment reads conflict free
#define NTH 512

__device__ __forceinline__ double bcast_lane(double v, int srclane) {
    int lo = __builtin_amdgcn_readlane(__double2loint(v), srclane);
    int hi = __builtin_amdgcn_readlane(__double2hiint(v), srclane);
    return __hiloint2double(hi, lo);
}

// 16x16 lower Cholesky by lanes 0..15 of one wave; D points at the (c0,c0) corner (stride LDD).
// Writes L16 (lower) back and 1/L_jj to invd[0..15].  Returns 0 or 1 + first bad local column.
__device__ __forceinline__ int chol16(double* D, double* invd_out, int lane) {
    const int r = lane & 15;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = D[r * LDD + c];
    int bad = 0;
    double myinv = 0.0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const double djj = bcast_lane(a[j], j);
        if (!(djj > 0.0) && bad == 0) bad = j + 1;
        // 1/sqrt(d): v_rsq_f64 seed + Newton on both l = sqrt(d) and 1/l (a third of the dependent
        // instruction chain of sqrt() followed by a divide)
        double inv = __builtin_amdgcn_rsq(djj);
        double ljj = djj * inv;
        ljj = fma(0.5 * inv, fma(-ljj, ljj, djj), ljj);
        inv = fma(inv, fma(-ljj, inv, 1.0), inv);
        ljj = fma(0.5 * inv, fma(-ljj, ljj, djj), ljj);
        inv = fma(inv, fma(-ljj, inv, 1.0), inv);
        if (r == j) myinv = inv;
        a[j] = (r == j) ? ljj : a[j] * inv;
#pragma unroll
        for (int c = j + 1; c < 16; ++c) {
            const double lcj = bcast_lane(a[j], c);
            a[c] = fma(-a[j], lcj, a[c]);
        }
    }
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (c <= r) D[r * LDD + c] = a[c];
        invd_out[r] = myinv;
    }
    return bad;
}

// inverse of a 16x16 lower-triangular block held row-per-lane: lane c builds column c of X = L^-1
//   X[r][c] = (delta_rc - sum_{k<r} L[r][k] X[k][c]) / L[r][r]
__device__ __forceinline__ void trinv16(const double* L, int ldl, const double* invd, double* out, int ldo,
                                        int lane) {
    const int r = lane & 15;
    double a[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) a[c] = (c <= r) ? L[r * ldl + c] : 0.0;
    const double myinv = invd[r];
    double xc[16];
#pragma unroll
    for (int rr = 0; rr < 16; ++rr) {
        double s = (rr == r) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < rr; ++k) {
            const double lrk = bcast_lane(a[k], rr);
            s = fma(-lrk, xc[k], s);
        }
        xc[rr] = s * bcast_lane(myinv, rr);
    }
    if (lane < 16) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) out[rr * ldo + r] = (rr >= r) ? xc[rr] : 0.0;
    }
}

__device__ __forceinline__ d4 tile_read(const double* C, int lane) {
    d4 v;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) v[rg] = C[((lane >> 4) + 4 * rg) * LDD + (lane & 15)];
    return v;
}
__device__ __forceinline__ void tile_write(double* C, d4 v, int lane) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) C[((lane >> 4) + 4 * rg) * LDD + (lane & 15)] = v[rg];
}

__device__ __forceinline__ void load_block(double* D, const double* __restrict__ Ablk, int64_t ld, int tid) {
    // 128x128 doubles = 8192 16-byte chunks, 16 per thread, all loads in flight before the first store
    d2 r[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = tid + NTH * i, row = e >> 6, c2 = (e & 63) * 2;
        r[i] = *reinterpret_cast<const d2*>(Ablk + (int64_t)row * ld + c2);
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int e = tid + NTH * i, row = e >> 6, c2 = (e & 63) * 2;
        *reinterpret_cast<d2*>(D + row * LDD + c2) = r[i];
    }
}


// Cholesky of the leading npan*16 rows/cols of the block in D (lower part), in place.
// invd[j] <- 1/L_jj.  *s_bad <- 1 + first non-positive pivot column (if any, first only).
__device__ inline void lds_factor(double* D, double* invd, int npan, int* s_bad, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int ns = npan * 16;
    // The 16x16 register Cholesky of the next diagonal tile is the serial part.  It is overlapped with
    // the trailing update: wave 0 updates tile (p+1,p+1) first and goes straight on to factor it
    // while waves 1..7 update the other tiles of the step.
    if (wave == 0) {
        const int bad = chol16(D, invd, lane);
        if (lane == 0 && bad && *s_bad == 0) *s_bad = bad;
    }
    __syncthreads();
    for (int p = 0; p < npan; ++p) {
        const int c0 = p * 16;
        // panel solve, one thread per row below the diagonal tile: x L16^T = a
        {
            const int row = c0 + 16 + tid;
            if (row < ns) {
                double x[16];
                double* px = D + row * LDD + c0;
#pragma unroll
                for (int c = 0; c < 16; c += 2) {
                    const d2 v = *reinterpret_cast<const d2*>(px + c);
                    x[c] = v[0];
                    x[c + 1] = v[1];
                }
                const double* Lp = D + c0 * LDD + c0;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    double s = x[c];
#pragma unroll
                    for (int k = 0; k < c; ++k) s = fma(-x[k], Lp[c * LDD + k], s);
                    x[c] = s * invd[c0 + c];
                }
#pragma unroll
                for (int c = 0; c < 16; c += 2) *reinterpret_cast<d2*>(px + c) = (d2){x[c], x[c + 1]};
            }
        }
        __syncthreads();
        // trailing update of the lower tiles (rt >= ct > p); tile q = 0 is (p+1, p+1)
        const int m = npan - 1 - p;
        const int ntile = m * (m + 1) / 2;
        const int nworkers = NTH / 64 - 1;
        for (int q = (wave == 0) ? 0 : wave; q < ntile; q += (wave == 0) ? ntile : nworkers) {
            int i = (int)((sqrtf(8.0f * q + 1.0f) - 1.0f) * 0.5f);
            while (i * (i + 1) / 2 > q) --i;
            while ((i + 1) * (i + 2) / 2 <= q) ++i;
            const int j = q - i * (i + 1) / 2;
            const int rt = p + 1 + i, ct = p + 1 + j;
            double* C = D + rt * 16 * LDD + ct * 16;
            d4 acc = tile_read(C, lane);
#pragma unroll
            for (int s = 0; s < 16; s += 4) {
                const double a = -D[(rt * 16 + (lane & 15)) * LDD + c0 + s + (lane >> 4)];
                const double b = D[(ct * 16 + (lane & 15)) * LDD + c0 + s + (lane >> 4)];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
            }
            tile_write(C, acc, lane);
        }
        if (wave == 0 && p + 1 < npan) {
            const int c1 = c0 + 16;
            const int bad = chol16(D + c1 * LDD + c1, invd + c1, lane);
            if (lane == 0 && bad && *s_bad == 0) *s_bad = c1 + bad;
        }
        __syncthreads();
    }
}

// In-place inverse of the lower-triangular block (leading npan*16 rows) by recursive doubling.
// Precondition: the 16x16 diagonal sub-blocks already hold their inverses with zeros above the
// diagonal.  Only tiles whose rows lie inside the leading npan*16 rows are touched.
__device__ inline void lds_invert_levels(double* D, int npan, int tid) {
    const int lane = tid & 63, wave = tid >> 6;
    const int ns = npan * 16;
    for (int half = 16; half < ns; half *= 2) {
        const int ht = half / 16;               // tiles per side of one sub-block
        const int tiles_per_pair = ht * ht;
        const int ntile = (NB / (2 * half)) * tiles_per_pair;   // 4, 8, 16  (<= 2 per wave)
        d4 keep[2];
        int ti2[2], tj2[2], r02[2], c02[2];
        bool on2[2];
#pragma unroll
        for (int cnt = 0; cnt < 2; ++cnt) {
            const int q = wave + 8 * cnt;
            const int pr = q / tiles_per_pair, w = q % tiles_per_pair;
            ti2[cnt] = w / ht;
            tj2[cnt] = w % ht;
            r02[cnt] = (2 * pr + 1) * half;
            c02[cnt] = 2 * pr * half;
            on2[cnt] = (q < ntile) && (r02[cnt] + ti2[cnt] * 16 < ns);
            keep[cnt] = (d4){0.0, 0.0, 0.0, 0.0};
        }
        // phase 1: T = L21 * X11 (X11 lower: k-tiles kt >= tj); results stay in registers until every
        // wave has finished reading L21, then overwrite it
        for (int kt = 0; kt < ht; ++kt)
#pragma unroll
            for (int sft = 0; sft < 16; sft += 4)
#pragma unroll
                for (int cnt = 0; cnt < 2; ++cnt) {
                    if (!on2[cnt] || kt < tj2[cnt]) continue;
                    const double a = D[(r02[cnt] + ti2[cnt] * 16 + (lane & 15)) * LDD + c02[cnt] + kt * 16 + sft + (lane >> 4)];
                    const double b = D[(c02[cnt] + kt * 16 + sft + (lane >> 4)) * LDD + c02[cnt] + tj2[cnt] * 16 + (lane & 15)];
                    keep[cnt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, keep[cnt], 0, 0, 0);
                }
        __syncthreads();
#pragma unroll
        for (int cnt = 0; cnt < 2; ++cnt)
            if (on2[cnt]) tile_write(D + (r02[cnt] + ti2[cnt] * 16) * LDD + c02[cnt] + tj2[cnt] * 16, keep[cnt], lane);
        __syncthreads();
        // phase 2: X21 = -X22 * T (X22 lower: k-tiles kt <= ti)
#pragma unroll
        for (int cnt = 0; cnt < 2; ++cnt) keep[cnt] = (d4){0.0, 0.0, 0.0, 0.0};
        for (int kt = 0; kt < ht; ++kt)
#pragma unroll
            for (int sft = 0; sft < 16; sft += 4)
#pragma unroll
                for (int cnt = 0; cnt < 2; ++cnt) {
                    if (!on2[cnt] || kt > ti2[cnt]) continue;
                    const double a = -D[(r02[cnt] + ti2[cnt] * 16 + (lane & 15)) * LDD + r02[cnt] + kt * 16 + sft + (lane >> 4)];
                    const double b = D[(r02[cnt] + kt * 16 + sft + (lane >> 4)) * LDD + c02[cnt] + tj2[cnt] * 16 + (lane & 15)];
                    keep[cnt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, keep[cnt], 0, 0, 0);
                }
        __syncthreads();
#pragma unroll
        for (int cnt = 0; cnt < 2; ++cnt)
            if (on2[cnt]) tile_write(D + (r02[cnt] + ti2[cnt] * 16) * LDD + c02[cnt] + tj2[cnt] * 16, keep[cnt], lane);
        __syncthreads();
    }
}
