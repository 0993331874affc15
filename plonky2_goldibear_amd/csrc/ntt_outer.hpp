// Transforms of more than 2^22 rows (the reference has no size cap below the field's two-adicity, field/src/fft.rs:168-205): one outer
// radix-R step, R = 2^K <= 16, around transforms of m = n / R rows - the passes themselves up to m = 2^22 (round 6; 2^20 in round 5),
// another outer step above that (2^27 rows and more: the sub-transform is its own caller).  With i = R i2 + i1 (decimation in time),
//     X[k2 + m k1] = sum_i1 w_R^(i1 k1) * ( g(k2)^i1 * Y_i1[k2] ),     Y_i1 = the size-m transform of the stride-R subsequence i1,
// g(k2) = w_n^(k2) (times the coset shift s_c for the LDE, whose sub-transforms run on the shift s_c^R).  So: de-interleave the R
// subsequences (one pass), run the sub-transforms on R times as many columns, and combine (one pass: twiddle, R-point DFT).  Two
// extra passes over the data per transform: sizes far outside BASELINE.json's configs, built for coverage of the reference's range.
// Written once against the field traits (GlF: canonical words and plain tables; BbF: Montgomery words and tables).
#pragma once
#include <algorithm>

#include "field_traits.hpp"

namespace gbk {
namespace outer {

static constexpr int THREADS = 256;

__device__ __forceinline__ u32 brev_bits(u32 x, u32 bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// dst[col][i1][i2] = src[col][R i2 + i1]
template <class F, u32 K>
__global__ __launch_bounds__(THREADS) void k_deinterleave(const typename F::T* __restrict__ src, typename F::T* __restrict__ dst, u32 log_m) {
    typedef typename F::T T;
    constexpr u32 R = 1u << K, CHUNK = R * sizeof(T) >= 16 ? 16 / sizeof(T) : R;
    struct alignas(CHUNK * sizeof(T)) Chunk { T v[CHUNK]; };
    const size_t g = (size_t)blockIdx.x * THREADS + threadIdx.x;   // (col, i2)
    const size_t col = g >> log_m, i2 = g & (((size_t)1 << log_m) - 1);
    const Chunk* s = reinterpret_cast<const Chunk*>(src + ((col << log_m) << K) + (i2 << K));
    T* d = dst + ((col << log_m) << K) + i2;
#pragma unroll
    for (u32 j = 0; j < R / CHUNK; j++) {
        const Chunk ch = s[j];
#pragma unroll
        for (u32 e = 0; e < CHUNK; e++) d[(size_t)(j * CHUNK + e) << log_m] = ch.v[e];
    }
}

// w^e from split tables (e = 1024 e_hi + e_lo), device form
template <class F>
__device__ __forceinline__ typename F::T tw_split(const typename F::T* __restrict__ hi, const typename F::T* __restrict__ lo, u32 e) {
    const typename F::T w = lo[e & 1023];
    return (e >> 10) ? F::mul(w, hi[e >> 10]) : w;
}

// R-point DFT (R = 2^K <= 16) of z[0..R) with the primitive R-th root `wr` (forward or inverse, as the caller passes it): X[k1] in
// z[k1].  K is a template parameter of everything here, so that the R values live in registers (as run-time bounds they lived in
// scratch memory and the combine pass ran at 2.5 TB/s).  R = 8 and 16 as the plain O(R^2) sum.
template <class F, u32 K>
__device__ __forceinline__ void dft_r(typename F::T (&z)[1u << K], typename F::T wr) {
    typedef typename F::T T;
    constexpr u32 R = 1u << K;
    if constexpr (K == 1) {
        const T a = z[0], b = z[1];
        z[0] = F::add(a, b);
        z[1] = F::sub(a, b);
    } else if constexpr (K > 2) {
        T wp[R], x[R];
        wp[0] = F::one();
#pragma unroll
        for (u32 j = 1; j < R; j++) wp[j] = F::mul(wp[j - 1], wr);
#pragma unroll
        for (u32 k1 = 0; k1 < R; k1++) {
            T acc = z[0];
#pragma unroll
            for (u32 i1 = 1; i1 < R; i1++) acc = F::add(acc, F::mul(z[i1], wp[(i1 * k1) & (R - 1)]));
            x[k1] = acc;
        }
#pragma unroll
        for (u32 k1 = 0; k1 < R; k1++) z[k1] = x[k1];
    } else {
        const T w4 = wr;
        const T s02 = F::add(z[0], z[2]), d02 = F::sub(z[0], z[2]), s13 = F::add(z[1], z[3]), d13 = F::mul(F::sub(z[1], z[3]), w4);
        z[0] = F::add(s02, s13);
        z[1] = F::add(d02, d13);
        z[2] = F::sub(s02, s13);
        z[3] = F::sub(d02, d13);
    }
}

// in place: data[col][k1][k2] <- r_inv * sum_i1 w_R^-(i1 k1) w_n^-(i1 k2) data[col][i1][k2]
template <class F, u32 K>
__global__ __launch_bounds__(THREADS) void k_intt_combine(typename F::T* __restrict__ data, u32 log_m, const typename F::T* __restrict__ tw_hi_inv,
                                                          const typename F::T* __restrict__ tw_lo_inv, typename F::T w4_inv, typename F::T r_inv) {
    typedef typename F::T T;
    constexpr u32 R = 1u << K;
    const size_t g = (size_t)blockIdx.x * THREADS + threadIdx.x;
    const size_t col = g >> log_m;
    const u32 k2 = (u32)(g & (((size_t)1 << log_m) - 1));
    T* p = data + ((col << log_m) << K) + k2;
    const T w = tw_split<F>(tw_hi_inv, tw_lo_inv, k2);   // w_n^-k2
    T z[R], f = r_inv;
#pragma unroll
    for (u32 i1 = 0; i1 < R; i1++) {
        z[i1] = F::mul(p[(size_t)i1 << log_m], f);
        if (i1 + 1 < R) f = F::mul(f, w);
    }
    dft_r<F, K>(z, w4_inv);
#pragma unroll
    for (u32 k1 = 0; k1 < R; k1++) p[(size_t)k1 << log_m] = z[k1];
}

// sub: [col][i1][c][q] (q = brev(k2): the sub-LDEs' leaf order) -> lde[col][c][q R + brev_K(k1)]
template <class F, u32 K>
__global__ __launch_bounds__(THREADS) void k_lde_combine(const typename F::T* __restrict__ sub, typename F::T* __restrict__ lde, u32 log_m, u32 rate_bits,
                                                         const typename F::T* __restrict__ tw_hi, const typename F::T* __restrict__ tw_lo,
                                                         const typename F::T* __restrict__ tw_top, const typename F::T* __restrict__ pow_lo, u32 nlo,
                                                         typename F::T w4) {
    typedef typename F::T T;
    constexpr u32 R = 1u << K;
    static_assert(THREADS == 256, "tw_top holds the factor of the 256 threads of a workgroup");
    const size_t g = (size_t)blockIdx.x * THREADS + threadIdx.x;   // (col, c, q)
    const u32 q = (u32)(g & (((size_t)1 << log_m) - 1));
    const size_t cc = g >> log_m;
    const u32 c = (u32)(cc & ((1u << rate_bits) - 1));
    const size_t col = cc >> rate_bits;
    // k2 = brev(q): the workgroup's 256 consecutive q are the TOP eight bits of k2 - taken from the split tables every lane of a load
    // hit a cache line of its own (64 lines per kilobyte of data streamed: 2.9 TB/s).  w_n^k2 = tw_top[q & 255] * w_n^(the low bits),
    // the second factor uniform over the workgroup, the first from 2 KB.
    const u32 rest = brev_bits(q >> 8, log_m - 8);
    const T gk = F::mul(F::mul(pow_lo[(size_t)c * nlo + 1], tw_split<F>(tw_hi, tw_lo, rest)), tw_top[threadIdx.x]);   // s_c w_n^k2
    T z[R], f = gk;
#pragma unroll
    for (u32 i1 = 0; i1 < R; i1++) {
        const T v = sub[(((((col << K) + i1) << rate_bits) + c) << log_m) + q];
        z[i1] = i1 ? F::mul(v, f) : v;
        if (i1 && i1 + 1 < R) f = F::mul(f, gk);
    }
    dft_r<F, K>(z, w4);
    // the R outputs of a thread are neighbours: stored as 16-byte words (one 8-byte store per element left every other sector of a
    // line to a second instruction - 2.9 TB/s)
    constexpr u32 CHUNK = R * sizeof(T) >= 16 ? 16 / sizeof(T) : R;
    struct alignas(CHUNK * sizeof(T)) Chunk { T v[CHUNK]; };
    T out[R];
#pragma unroll
    for (u32 k1 = 0; k1 < R; k1++) out[brev_bits(k1, K)] = z[k1];
    Chunk* o = reinterpret_cast<Chunk*>(lde + ((((col << rate_bits) + c) << log_m) << K) + ((size_t)q << K));
#pragma unroll
    for (u32 j = 0; j < R / CHUNK; j++) {
        Chunk ch;
#pragma unroll
        for (u32 e = 0; e < CHUNK; e++) ch.v[e] = out[j * CHUNK + e];
        o[j] = ch;
    }
}

// the kernels above by the run-time K (1 .. NTT_OUTER_MAX_BITS = 4)
template <class F, class... A>
void launch_deinterleave(u32 K, dim3 grid, hipStream_t st, A... a) {
    switch (K) {
        case 1: hipLaunchKernelGGL((k_deinterleave<F, 1>), grid, dim3(THREADS), 0, st, a...); break;
        case 2: hipLaunchKernelGGL((k_deinterleave<F, 2>), grid, dim3(THREADS), 0, st, a...); break;
        case 3: hipLaunchKernelGGL((k_deinterleave<F, 3>), grid, dim3(THREADS), 0, st, a...); break;
        default: hipLaunchKernelGGL((k_deinterleave<F, 4>), grid, dim3(THREADS), 0, st, a...); break;
    }
}
template <class F, class... A>
void launch_intt_combine(u32 K, dim3 grid, hipStream_t st, A... a) {
    switch (K) {
        case 1: hipLaunchKernelGGL((k_intt_combine<F, 1>), grid, dim3(THREADS), 0, st, a...); break;
        case 2: hipLaunchKernelGGL((k_intt_combine<F, 2>), grid, dim3(THREADS), 0, st, a...); break;
        case 3: hipLaunchKernelGGL((k_intt_combine<F, 3>), grid, dim3(THREADS), 0, st, a...); break;
        default: hipLaunchKernelGGL((k_intt_combine<F, 4>), grid, dim3(THREADS), 0, st, a...); break;
    }
}
template <class F, class... A>
void launch_lde_combine(u32 K, dim3 grid, hipStream_t st, A... a) {
    switch (K) {
        case 1: hipLaunchKernelGGL((k_lde_combine<F, 1>), grid, dim3(THREADS), 0, st, a...); break;
        case 2: hipLaunchKernelGGL((k_lde_combine<F, 2>), grid, dim3(THREADS), 0, st, a...); break;
        case 3: hipLaunchKernelGGL((k_lde_combine<F, 3>), grid, dim3(THREADS), 0, st, a...); break;
        default: hipLaunchKernelGGL((k_lde_combine<F, 4>), grid, dim3(THREADS), 0, st, a...); break;
    }
}

// ---- host side.  `Sub` = callables running the 2^(log_n - K)-row transforms: intt(src, dst, scratch, ncols) / lde(coeffs, out, ncols).
// values [ncols][n] -> coefficients in `coeffs`; scratch holds ncols * n elements; src may equal coeffs.
template <class F, class SubIntt>
void intt_columns(const typename F::T* src, typename F::T* coeffs, typename F::T* scratch, size_t ncols, u32 log_n, u32 K, const typename F::T* tw_hi_inv,
                  const typename F::T* tw_lo_inv, SubIntt sub_intt, hipStream_t st) {
    typedef typename F::T T;
    const u32 log_m = log_n - K;
    const size_t n = (size_t)1 << log_n;
    const u32 grid = (u32)((ncols << log_m) / THREADS);
    if (src != coeffs) {
        launch_deinterleave<F>(K, dim3(grid), st, src, coeffs, log_m);
    } else {
        launch_deinterleave<F>(K, dim3(grid), st, src, scratch, log_m);
        (void)hipMemcpyAsync(coeffs, scratch, ncols * n * sizeof(T), hipMemcpyDeviceToDevice, st);
    }
    sub_intt(coeffs, coeffs, scratch, ncols << K);
    const T w4_inv = F::inv(F::two_adic_generator(K < 2 ? 2 : K)), r_inv = F::inv(F::enc((u64)1 << K));   // the R-th root (R = 2: unused)
    launch_intt_combine<F>(K, dim3(grid), st, coeffs, log_m, tw_hi_inv, tw_lo_inv, w4_inv, r_inv);
}

// coefficients [ncols][n] -> lde [ncols][2^r][n] (leaf order); work holds work_elems elements (>= (1 + 2^r) n for one column)
template <class F, class SubLde>
void lde_columns(const typename F::T* coeffs, typename F::T* lde, size_t ncols, u32 log_n, u32 K, u32 rate_bits, const typename F::T* tw_hi, const typename F::T* tw_lo,
                 const typename F::T* tw_top, const typename F::T* pow_lo, typename F::T* work, size_t work_elems, SubLde sub_lde, hipStream_t st) {
    typedef typename F::T T;
    const u32 log_m = log_n - K;
    const size_t n = (size_t)1 << log_n, N = n << rate_bits;
    const size_t group = std::max<size_t>(1, work_elems / (n + N));
    const T w4 = F::two_adic_generator(K < 2 ? 2 : K);   // the primitive R-th root
    for (size_t c0 = 0; c0 < ncols; c0 += group) {
        const size_t g = std::min(group, ncols - c0);
        T* wc = work;            // [g R][m] de-interleaved coefficients
        T* wl = work + g * n;    // [g R][2^r][m] their LDEs
        launch_deinterleave<F>(K, dim3((u32)((g << log_m) / THREADS)), st, coeffs + c0 * n, wc, log_m);
        sub_lde(wc, wl, g << K);
        launch_lde_combine<F>(K, dim3((u32)(((g << rate_bits) << log_m) / THREADS)), st, (const T*)wl, lde + c0 * N, log_m, rate_bits, tw_hi, tw_lo, tw_top,
                              pow_lo, (u32)4096, w4);
    }
}

}  // namespace outer
}  // namespace gbk
