// select.hip -- ranking kernels of the Bayesian-optimisation step for LARGE candidate grids
// (reference: gpim/gpbayes/boptim.py:303-376, acqfunc.py:59,88).
//
//   nanmax_two_stage     np.nanmax over n values with the whole chip (per-block partials, then one block)
//   radix top-k          descending top-k by (value, flat index) -- the order np.argsort(...)[::-1] gives, with
//                        NaN first when kept -- in 12 passes over the 64-bit order keys: eight 8-bit radix
//                        passes find the k-th largest key, four more the index cut among its ties; each pass
//                        is one multi-block histogram launch whose LAST block (ticket counter) picks the digit
//                        and resets the histogram, so the whole selection is 15 launches without a host
//                        round trip, O(M) work each, instead of k passes of one workgroup over all M.
//   thin_batch_kernel    boptimizer.update_points: greedy maximum + suppression of everything within
//                        `dscale` (cKDTree.query_ball_point semantics: distance <= r), on the <= 1024 ranked
//                        candidates, in one workgroup.
// The single-workgroup kernels of engine.hip stay in use for small grids (M <= 2048: fewer launches).
#include "common.hpp"

#define SEL_MAXK 1024

struct SelState {
    unsigned long long prefix;      // high bits of the k-th largest key found so far
    unsigned long long key_t;       // the k-th largest key (after pass 7)
    unsigned int idx_prefix;        // high bits of the index cut among key == key_t
    unsigned int idx_t;             // smallest index still selected among key == key_t
    long long remaining;            // how many of the current candidate set are still to be taken
    long long kk;                   // min(k, number of rankable entries)
    unsigned int done;              // ticket counter of the current pass
    unsigned int ncand;             // compaction cursor
    unsigned int hist[256];
};

struct SelCand { unsigned long long key; long long idx; };

__device__ __forceinline__ unsigned long long sel_order_key(double v, int keep_nan) {
    if (v != v) return keep_nan ? 0xFFFFFFFFFFFFFFFFull : 0ull;
    unsigned long long b = (unsigned long long)__double_as_longlong(v);
    b = (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
    if (b == 0ull) b = 1ull;                                      // 0 = excluded, all-ones = NaN
    if (b == 0xFFFFFFFFFFFFFFFFull) b = 0xFFFFFFFFFFFFFFFEull;
    return b;
}

// keys + number of rankable entries (-> st->remaining = st->kk = min(k, valid))
__global__ __launch_bounds__(256) void sel_keys_kernel(const double* __restrict__ x, long long M, int keep_nan, int k,
                                                       unsigned long long* __restrict__ keys, SelState* st) {
    __shared__ unsigned int s_cnt;
    if (threadIdx.x == 0) s_cnt = 0;
    __syncthreads();
    unsigned int c = 0;
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < M; j += (long long)gridDim.x * 256) {
        const unsigned long long kk = sel_order_key(x[j], keep_nan);
        keys[j] = kk;
        c += kk != 0ull;
    }
    atomicAdd(&s_cnt, c);
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd((unsigned long long*)&st->remaining, (unsigned long long)s_cnt);
        __threadfence();
        const unsigned int t = atomicAdd(&st->done, 1u);
        if (t == gridDim.x - 1) {
            __threadfence();
            const long long valid = atomicAdd((unsigned long long*)&st->remaining, 0ull);
            const long long kk = valid < k ? valid : k;
            st->kk = kk;
            st->remaining = kk;
            st->prefix = 0ull;
            st->idx_prefix = 0u;
            st->done = 0u;
            st->ncand = 0u;
            __threadfence();
        }
    }
}

// one radix pass.  phase 0 (pass = 0..7): digit = byte (7 - pass) of the key among keys whose higher bytes equal
// st->prefix.  phase 1 (pass = 0..3): digit = byte (3 - pass) of the index among entries with key == key_t whose
// higher index bytes equal st->idx_prefix.  The last block to finish walks the histogram from the top, fixes
// the digit that contains the remaining-th largest entry, and clears the histogram for the next pass.
__global__ __launch_bounds__(256) void sel_pass_kernel(const unsigned long long* __restrict__ keys, long long M,
                                                       int phase, int pass, SelState* st) {
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    const unsigned long long prefix = st->prefix, key_t = st->key_t;
    const unsigned int ipre = st->idx_prefix;
    const int shift = phase == 0 ? 8 * (7 - pass) : 8 * (3 - pass);
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < M; j += (long long)gridDim.x * 256) {
        const unsigned long long kk = keys[j];
        if (phase == 0) {
            if (kk == 0ull) continue;
            const bool match = pass == 0 || (kk >> (shift + 8)) == (prefix >> (shift + 8));
            if (match) atomicAdd(&h[(unsigned int)(kk >> shift) & 255u], 1u);
        } else {
            if (kk != key_t) continue;
            const unsigned int ix = (unsigned int)j;
            const bool match = pass == 0 || (ix >> (shift + 8)) == (ipre >> (shift + 8));
            if (match) atomicAdd(&h[(ix >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&st->hist[threadIdx.x], h[threadIdx.x]);
    __threadfence();
    __syncthreads();
    __shared__ unsigned int s_last;
    if (threadIdx.x == 0) s_last = atomicAdd(&st->done, 1u) == gridDim.x - 1;
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    // (one thread: 256 bins)
    if (threadIdx.x == 0) {
        long long rem = st->remaining;
        int digit = 0;
        if (rem > 0) {
            long long cum = 0;
            for (int b = 255; b >= 0; --b) {
                const long long c = (long long)atomicAdd(&st->hist[b], 0u);
                if (cum + c >= rem) { digit = b; break; }
                cum += c;
            }
            rem -= cum;
        }
        if (phase == 0) {
            st->prefix |= (unsigned long long)digit << shift;
            if (pass == 7) st->key_t = st->prefix;
        } else {
            st->idx_prefix |= (unsigned int)digit << shift;
            if (pass == 3) st->idx_t = st->idx_prefix;
        }
        st->remaining = rem;
        st->done = 0u;
        __threadfence();
    }
    __syncthreads();
    st->hist[threadIdx.x] = 0u;
}

// compaction of the kk selected entries (unordered)
__global__ __launch_bounds__(256) void sel_collect_kernel(const unsigned long long* __restrict__ keys, long long M,
                                                          SelState* st, SelCand* __restrict__ cand) {
    const long long kk = st->kk;
    if (kk <= 0) return;
    const unsigned long long key_t = st->key_t;
    const unsigned int idx_t = st->idx_t;
    for (long long j = (long long)blockIdx.x * 256 + threadIdx.x; j < M; j += (long long)gridDim.x * 256) {
        const unsigned long long k = keys[j];
        if (k == 0ull) continue;
        if (k > key_t || (k == key_t && (unsigned int)j >= idx_t)) {
            const unsigned int slot = atomicAdd(&st->ncand, 1u);
            if (slot < SEL_MAXK) { cand[slot].key = k; cand[slot].idx = j; }
        }
    }
}

// bitonic sort of the <= 1024 selected entries, descending by (key, index); writes the public outputs
__global__ __launch_bounds__(1024) void sel_sort_kernel(const double* __restrict__ x, SelState* st,
                                                        const SelCand* __restrict__ cand, int k,
                                                        double* __restrict__ vals, long long* __restrict__ idx,
                                                        long long* __restrict__ count) {
    __shared__ unsigned long long sk[SEL_MAXK];
    __shared__ long long si[SEL_MAXK];
    const int tid = threadIdx.x;
    const long long kk = st->kk;
    if (tid < kk) { sk[tid] = cand[tid].key; si[tid] = cand[tid].idx; }
    else { sk[tid] = 0ull; si[tid] = -1; }
    __syncthreads();
    for (int size = 2; size <= SEL_MAXK; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int partner = tid ^ stride;
            if (partner > tid) {
                const bool desc = (tid & size) == 0;
                const bool a_less = sk[tid] < sk[partner] || (sk[tid] == sk[partner] && si[tid] < si[partner]);
                if (a_less == desc) {
                    const unsigned long long tk = sk[tid]; sk[tid] = sk[partner]; sk[partner] = tk;
                    const long long ti = si[tid]; si[tid] = si[partner]; si[partner] = ti;
                }
            }
            __syncthreads();
        }
    }
    if (tid < k) {
        if (tid < kk) { vals[tid] = x[si[tid]]; idx[tid] = si[tid]; }
        else { vals[tid] = __builtin_nan(""); idx[tid] = -1; }
    }
    if (tid == 0) {
        *count = kk;
        st->remaining = 0;       // ready for the next call
        st->done = 0u;
    }
}

int launch_topk_radix(gpimhip_ctx* h, const double* x, int64_t M, int k, int keep_nan, double* vals, int64_t* idx,
                      int64_t* count) {
    SelState* st = reinterpret_cast<SelState*>(h->sel_scratch);
    SelCand* cand = reinterpret_cast<SelCand*>(h->sel_scratch + 4096);
    const unsigned nblk = (unsigned)std::min<int64_t>(1024, (M + 255) / 256);
    HIP_TRY(hipMemsetAsync(st, 0, sizeof(SelState), h->stream));
    hipLaunchKernelGGL(sel_keys_kernel, dim3(nblk), dim3(256), 0, h->stream, x, (long long)M, keep_nan, k, h->keys, st);
    for (int p = 0; p < 8; ++p)
        hipLaunchKernelGGL(sel_pass_kernel, dim3(nblk), dim3(256), 0, h->stream, h->keys, (long long)M, 0, p, st);
    for (int p = 0; p < 4; ++p)
        hipLaunchKernelGGL(sel_pass_kernel, dim3(nblk), dim3(256), 0, h->stream, h->keys, (long long)M, 1, p, st);
    hipLaunchKernelGGL(sel_collect_kernel, dim3(nblk), dim3(256), 0, h->stream, h->keys, (long long)M, st, cand);
    hipLaunchKernelGGL(sel_sort_kernel, dim3(1), dim3(1024), 0, h->stream, x, st, cand, k, vals, (long long*)idx,
                       (long long*)count);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}

// ------------------------------------------------------------------------------------------
// nanmax with the whole chip
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nanmax_partial_kernel(const double* __restrict__ x, long long n,
                                                             double* __restrict__ part) {
    __shared__ double red[256];
    double best = -INFINITY;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double v = x[i];
        if (v == v) best = fmax(best, v);
    }
    red[threadIdx.x] = best;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(1024) void nanmax_final_kernel(const double* __restrict__ part, int n,
                                                            double* __restrict__ out) {
    __shared__ double red[1024];
    red[threadIdx.x] = threadIdx.x < n ? part[threadIdx.x] : -INFINITY;
    __syncthreads();
    for (int s = 512; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + s]);
        __syncthreads();
    }
    // all-NaN input: np.nanmax returns NaN (with a warning)
    if (threadIdx.x == 0) out[0] = (red[0] == -INFINITY) ? __builtin_nan("") : red[0];
}
int launch_nanmax_two_stage(gpimhip_ctx* h, const double* x, int64_t n, double* out) {
    double* part = reinterpret_cast<double*>(h->sel_scratch + 4096 + SEL_MAXK * sizeof(SelCand));
    const int nblk = (int)std::min<int64_t>(1024, (n + 255) / 256);
    hipLaunchKernelGGL(nanmax_partial_kernel, dim3(nblk), dim3(256), 0, h->stream, x, (long long)n, part);
    hipLaunchKernelGGL(nanmax_final_kernel, dim3(1), dim3(1024), 0, h->stream, part, nblk, out);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
size_t sel_scratch_bytes() { return 4096 + SEL_MAXK * sizeof(SelCand) + 1024 * sizeof(double); }

// ------------------------------------------------------------------------------------------
// batch thinning (boptim.py:326-376): among n ranked candidates (values, d-dimensional grid indices) repeatedly
// take the largest remaining value and suppress every candidate within Euclidean distance <= dscale of it
// (itself included), until none is left or max_out are kept.  keep_out: kept candidate positions in order.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void thin_batch_kernel(const double* __restrict__ vals,
                                                          const long long* __restrict__ flat, int n, int d,
                                                          const long long* __restrict__ shape, double dscale,
                                                          int max_out, int* __restrict__ keep_out,
                                                          int* __restrict__ nkeep_out) {
    __shared__ double sv[SEL_MAXK];
    __shared__ int salive[SEL_MAXK];
    __shared__ double sc[SEL_MAXK][GPIMHIP_MAX_DIM];
    __shared__ double rv[1024];
    __shared__ int ri[1024];
    const int tid = threadIdx.x;
    if (tid < n) {
        sv[tid] = vals[tid];
        salive[tid] = 1;
        long long f = flat[tid];
        for (int k = d - 1; k >= 0; --k) { sc[tid][k] = (double)(f % shape[k]); f /= shape[k]; }
    }
    __syncthreads();
    int nkeep = 0;
    while (nkeep < max_out) {
        // arg-max over the remaining candidates; ties -> the earlier position (np.argmax)
        rv[tid] = (tid < n && salive[tid]) ? sv[tid] : -INFINITY;
        ri[tid] = (tid < n && salive[tid]) ? tid : 0x7fffffff;
        __syncthreads();
        for (int s = 512; s > 0; s >>= 1) {
            if (tid < s) {
                const bool take = rv[tid + s] > rv[tid] || (rv[tid + s] == rv[tid] && ri[tid + s] < ri[tid]);
                if (take) { rv[tid] = rv[tid + s]; ri[tid] = ri[tid + s]; }
            }
            __syncthreads();
        }
        const int cur = ri[0];
        __syncthreads();
        if (cur == 0x7fffffff) break;
        if (tid == 0) keep_out[nkeep] = cur;
        ++nkeep;
        if (tid < n && salive[tid]) {
            double r2 = 0.0;
            for (int k = 0; k < d; ++k) { const double dl = sc[tid][k] - sc[cur][k]; r2 = fma(dl, dl, r2); }
            // cKDTree.query_ball_point: points with distance <= r
            if (sqrt(r2) <= dscale) salive[tid] = 0;
        }
        __syncthreads();
    }
    if (tid == 0) *nkeep_out = nkeep;
}
int launch_thin_batch(gpimhip_ctx* h, const double* vals, const int64_t* flat, int n, int d, const int64_t* shape,
                      double dscale, int max_out, int32_t* keep_out, int32_t* nkeep_out) {
    hipLaunchKernelGGL(thin_batch_kernel, dim3(1), dim3(1024), 0, h->stream, vals, (const long long*)flat, n, d,
                       (const long long*)shape, dscale, max_out, keep_out, nkeep_out);
    HIP_TRY(hipGetLastError());
    return GPIMHIP_OK;
}
