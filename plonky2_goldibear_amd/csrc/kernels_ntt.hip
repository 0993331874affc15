// Goldilocks NTT / LDE kernels for gfx950 (v1: LDS-tiled Cooley-Tukey passes, radix-2 layers in LDS).
//
// Replaces the per-column Rayon loops of PolynomialBatch::from_values / from_coeffs
// (fri/oracle.rs:76-80 "IFFT", :125-150 "FFT + blinding") and the transpose + bit-reverse that
// follow them (:108-109).  Data stays column-major [col][n]; every pass moves 2^12-element tiles
// HBM -> LDS -> HBM with >= 128-byte contiguous segments per row.
//
//  inverse (values -> coefficients, natural -> natural), n = 2^L, 12 < L <= 20:
//     index i = a*2^LL + b*2^LC + c   (LA + LB + LC = L, LL = LB + LC)
//     P1: DFT over a, twiddle w^-(k_a * l)                 tile 2^LA rows x 16 contiguous
//     P2: DFT over b, twiddle w_{2^LL}^-(c * k_b)           tile 16 k_a x 2^LB x 16 c   (LB > 0 only)
//     P3: DFT over c, * n^-1, transposed write to k = k_a + 2^LA k_b + 2^(LA+LB) k_c
//  LDE (coefficients -> leaf-order evaluations on the 2^r cosets 7 w_N^bitrev(c) H_n), L > 12:
//     PA: per coset: scale by s^(4096 a), DFT over a (LA = L-12 bits), twiddle w_n^(k_a l) s^l
//     PB: contiguous 4096-point DIF, natural -> bit-reversed = leaf order
//  L <= 12: one tile per column (inverse) / per (column, coset) (LDE).
#include <algorithm>
#include <cstdlib>

#include "kernels.hpp"
#include "gl_field.hpp"
#include "ntt_outer.hpp"

namespace gbk {

static constexpr int THREADS = 256;
static constexpr int TILE = 4096;

__device__ __forceinline__ u32 brev(u32 x, u32 bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// Pure size-2^m DFT along bits [p, p+m) of the LDS tile index, natural -> bit-reversed positions,
// all other index bits are batch.  tw = w_4096^(+-j) table (4096 entries).
__device__ __forceinline__ void lds_dft(u64* sh, u32 tile_elems, u32 p, u32 m, const u64* __restrict__ tw) {
    const u32 half = tile_elems >> 1;
    for (u32 l = 0; l < m; l++) {
        const u32 bitpos = p + m - 1 - l;
        const u32 hmask = (1u << (m - 1 - l)) - 1;
        const u32 lowmask = (1u << bitpos) - 1;
        for (u32 q = threadIdx.x; q < half; q += THREADS) {
            u32 e1 = ((q >> bitpos) << (bitpos + 1)) | (q & lowmask);
            u32 e2 = e1 | (1u << bitpos);
            u32 j = (e1 >> p) & hmask;
            u64 a = sh[e1], b = sh[e2];
            u64 d = gl::sub(a, b);
            if (j) d = gl::mul(d, tw[(j << l) << (12 - m)]);
            sh[e1] = gl::add(a, b);
            sh[e2] = d;
        }
        __syncthreads();
    }
}

// w_n^(+-e) from the split tables: e = 1024*e_hi + e_lo
__device__ __forceinline__ u64 tw_split(const u64* __restrict__ hi, const u64* __restrict__ lo, u32 e) {
    u32 eh = e >> 10, el = e & 1023;
    u64 w = lo[el];
    return eh ? gl::mul(w, hi[eh]) : w;
}

// ------------------------------------------------------------------ inverse NTT, multi-pass (12 < L <= 20)

struct InvGeom {
    u32 L, LA, LB, LC;
};

// P1: grid = ncols * 2^(LL-4)
__global__ __launch_bounds__(THREADS) void k_gl_intt_p1(const u64* __restrict__ src, u64* __restrict__ dst, InvGeom g,
                                                        const u64* __restrict__ tw4096, const u64* __restrict__ tw_hi,
                                                        const u64* __restrict__ tw_lo) {
    __shared__ u64 sh[TILE];
    const u32 LL = g.LB + g.LC;
    const u32 tiles_per_col = 1u << (LL - 4);
    const size_t col = blockIdx.x / tiles_per_col;
    const u32 tg = blockIdx.x % tiles_per_col;
    const size_t base = (col << g.L) + ((size_t)tg << 4);
    const u32 rows = 1u << g.LA;
    const u32 j = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    for (u32 a = r0; a < rows; a += 16) sh[a * 16 + j] = src[base + ((size_t)a << LL) + j];
    __syncthreads();
    lds_dft(sh, rows * 16, 4, g.LA, tw4096);
    const u32 l = (tg << 4) + j;
    for (u32 ka = r0; ka < rows; ka += 16) {
        u64 v = sh[brev(ka, g.LA) * 16 + j];
        u32 e = ka * l;
        if (e) v = gl::mul(v, tw_split(tw_hi, tw_lo, e));
        dst[base + ((size_t)ka << LL) + j] = v;
    }
}

// P2: grid = ncols * 2^(LA-4) * 2^(LC-4);  src layout [k_a][b][c], dst layout [k_b][k_a][c]
__global__ __launch_bounds__(THREADS) void k_gl_intt_p2(const u64* __restrict__ src, u64* __restrict__ dst, InvGeom g,
                                                        const u64* __restrict__ tw4096) {
    __shared__ u64 sh[TILE];
    const u32 LL = g.LB + g.LC;
    const u32 n_ga = 1u << (g.LA - 4), n_gc = 1u << (g.LC - 4);
    const size_t col = blockIdx.x / (n_ga * n_gc);
    const u32 rem = blockIdx.x % (n_ga * n_gc);
    const u32 ga = rem / n_gc, gc = rem % n_gc;
    const size_t cbase = col << g.L;
    const u32 nb = 1u << g.LB;
    const u32 jc = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    // LDS index t = (ia * nb + b) * 16 + jc ; rows = 16 * nb
    for (u32 r = r0; r < 16 * nb; r += 16) {
        u32 ia = r >> g.LB, b = r & (nb - 1);
        sh[r * 16 + jc] = src[cbase + ((size_t)(16 * ga + ia) << LL) + ((size_t)b << g.LC) + 16 * gc + jc];
    }
    __syncthreads();
    lds_dft(sh, 16 * nb * 16, 4, g.LB, tw4096);
    const u32 c = 16 * gc + jc;
    for (u32 r = r0; r < 16 * nb; r += 16) {
        u32 ia = r >> g.LB, kb = r & (nb - 1);
        u64 v = sh[(ia * nb + brev(kb, g.LB)) * 16 + jc];
        u32 e = c * kb;  // < 2^LL
        if (e) v = gl::mul(v, tw4096[e << (12 - LL)]);
        dst[cbase + ((size_t)kb << (g.LA + g.LC)) + ((size_t)(16 * ga + ia) << g.LC) + c] = v;
    }
}

// P3: grid = ncols * 2^LB * 2^(LA-4); src layout [k_b][k_a][c]; dst natural k = k_a + 2^LA k_b + 2^(LA+LB) k_c
__global__ __launch_bounds__(THREADS) void k_gl_intt_p3(const u64* __restrict__ src, u64* __restrict__ dst, InvGeom g,
                                                        const u64* __restrict__ tw4096, u64 n_inv) {
    __shared__ u64 sh[TILE];
    const u32 n_ga = 1u << (g.LA - 4), nb = 1u << g.LB, nc = 1u << g.LC;
    const size_t col = blockIdx.x / (nb * n_ga);
    const u32 rem = blockIdx.x % (nb * n_ga);
    const u32 kb = rem / n_ga, ga = rem % n_ga;
    const size_t cbase = col << g.L;
    const size_t sbase = cbase + ((size_t)kb << (g.LA + g.LC)) + ((size_t)(16 * ga) << g.LC);
    for (u32 t = threadIdx.x; t < 16 * nc; t += THREADS) sh[t] = src[sbase + t];  // 16 rows of 2^LC, contiguous
    __syncthreads();
    lds_dft(sh, 16 * nc, 0, g.LC, tw4096);
    const u32 ia = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    for (u32 kc = r0; kc < nc; kc += 16) {
        u64 v = gl::mul(sh[ia * nc + brev(kc, g.LC)], n_inv);
        dst[cbase + ((size_t)kc << (g.LA + g.LB)) + ((size_t)kb << g.LA) + 16 * ga + ia] = v;
    }
}

// single-tile transform for L <= 12: grid = ncols; natural -> natural, scaled by `scale`
__global__ __launch_bounds__(THREADS) void k_gl_ntt_small(const u64* __restrict__ src, u64* __restrict__ dst, u32 L,
                                                          const u64* __restrict__ tw4096, u64 scale) {
    __shared__ u64 sh[TILE];
    const u32 n = 1u << L;
    const size_t base = (size_t)blockIdx.x << L;
    for (u32 t = threadIdx.x; t < n; t += THREADS) sh[t] = src[base + t];
    __syncthreads();
    lds_dft(sh, n, 0, L, tw4096);
    for (u32 k = threadIdx.x; k < n; k += THREADS) {
        u64 v = sh[brev(k, L)];
        dst[base + k] = scale == 1 ? v : gl::mul(v, scale);
    }
}

// ------------------------------------------------------------------ LDE

// PA (L > 12): grid = ncols * 256 ; tile = 2^LA rows (a) x 16 contiguous l ; loops over the cosets.
__global__ __launch_bounds__(THREADS) void k_gl_lde_pa(const u64* __restrict__ coeffs, u64* __restrict__ lde, u32 L,
                                                       u32 rate_bits, const u64* __restrict__ tw4096,
                                                       const u64* __restrict__ tw_hi, const u64* __restrict__ tw_lo,
                                                       const u64* __restrict__ pow_lo, const u64* __restrict__ pow_hi) {
    __shared__ u64 sh[TILE];
    const u32 LA = L - 12;
    const u32 rows = 1u << LA;
    const size_t col = blockIdx.x >> 8;
    const u32 tg = blockIdx.x & 255;
    const u32 j = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    const u32 l = (tg << 4) + j;
    const size_t n = (size_t)1 << L;
    const u64* cin = coeffs + col * n + l;
    u64 orig[16];
#pragma unroll
    for (u32 it = 0; it < 16; it++) {
        u32 a = r0 + 16 * it;
        orig[it] = a < rows ? cin[(size_t)a << 12] : 0;
    }
    const u32 ncosets = 1u << rate_bits;
    for (u32 c = 0; c < ncosets; c++) {
        const u64* ph = pow_hi + (size_t)c * rows;
#pragma unroll
        for (u32 it = 0; it < 16; it++) {
            u32 a = r0 + 16 * it;
            if (a < rows) sh[a * 16 + j] = a ? gl::mul(orig[it], ph[a]) : orig[it];
        }
        __syncthreads();
        lds_dft(sh, rows * 16, 4, LA, tw4096);
        const u64 sl = pow_lo[(size_t)c * 4096 + l];
        u64* out = lde + (col << (L + rate_bits)) + (size_t)c * n + l;
#pragma unroll
        for (u32 it = 0; it < 16; it++) {
            u32 pa = r0 + 16 * it;
            if (pa < rows) {
                u32 e = brev(pa, LA) * l;
                u64 f = e ? gl::mul(sl, tw_split(tw_hi, tw_lo, e)) : sl;
                out[(size_t)pa << 12] = gl::mul(sh[pa * 16 + j], f);
            }
        }
        __syncthreads();
    }
}

// PB: contiguous tile of 2^LT = min(n, 4096) points, forward DIF natural -> bit-reversed, in place in
// `lde`.  grid = ncols * 2^r * (n / 2^LT).  FROM_COEFFS (L <= 12): read coeffs * s^l instead.
template <bool FROM_COEFFS>
__global__ __launch_bounds__(THREADS) void k_gl_lde_pb(const u64* __restrict__ coeffs, u64* __restrict__ lde, u32 L,
                                                       u32 rate_bits, const u64* __restrict__ tw4096,
                                                       const u64* __restrict__ pow_lo) {
    __shared__ u64 sh[TILE];
    const u32 LT = L < 12 ? L : 12;
    const u32 te = 1u << LT;
    const size_t tile = blockIdx.x;  // (col, coset, row) flattened == contiguous tiles of lde
    u64* p = lde + (tile << LT);
    if (FROM_COEFFS) {
        const size_t col = tile >> rate_bits;
        const u32 c = (u32)(tile & ((1u << rate_bits) - 1));
        const u64* cin = coeffs + (col << L);
        const u64* pl = pow_lo + ((size_t)c << LT);
        for (u32 t = threadIdx.x; t < te; t += THREADS) sh[t] = gl::mul(cin[t], pl[t]);
    } else {
        for (u32 t = threadIdx.x; t < te; t += THREADS) sh[t] = p[t];
    }
    __syncthreads();
    lds_dft(sh, te, 0, LT, tw4096);
    for (u32 t = threadIdx.x; t < te; t += THREADS) p[t] = sh[t];
}

// x -> x mod p for any u64 x (x < 2p: one conditional subtraction), 2 elements per thread
__global__ __launch_bounds__(256) void k_gl_canonicalize(u64* __restrict__ p, size_t count) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 2;
    if (i + 1 < count) {
        ulonglong2 v = *reinterpret_cast<ulonglong2*>(p + i);
        v.x = v.x >= gl::P ? v.x - gl::P : v.x;
        v.y = v.y >= gl::P ? v.y - gl::P : v.y;
        *reinterpret_cast<ulonglong2*>(p + i) = v;
    } else if (i < count) {
        p[i] = p[i] >= gl::P ? p[i] - gl::P : p[i];
    }
}

// ------------------------------------------------------------------ host launchers

// radix-16 register kernels (kernels_ntt16.hip); return false when the shape is not covered
bool gl_intt_columns_r16(const u64* src, u64* coeffs, u64* scratch, size_t ncols, const GlNttTables& t, hipStream_t stream);
bool gl_lde_pa_r16(const u64* coeffs, u64* lde, size_t ncols, const GlNttTables& t, const GlCosetTables& ct, hipStream_t stream);
void gl_lde_pb_r16(u64* lde, size_t ntiles, const GlNttTables& t, hipStream_t stream);

void gl_canonicalize(u64* p, size_t count, hipStream_t stream) {
    if (!count) return;
    hipLaunchKernelGGL(k_gl_canonicalize, dim3((u32)((count + 511) / 512)), dim3(256), 0, stream, p, count);
}

static InvGeom inv_geom(u32 L) {
    InvGeom g;
    g.L = L;
    if (L <= 16) {
        g.LC = 8;
        g.LA = L - 8;
        g.LB = 0;
    } else {
        g.LA = 8;
        g.LC = 8;
        g.LB = L - 16;
    }
    return g;
}

static void gl_intt_group(const u64* src, u64* coeffs, u64* scratch, size_t ncols, const GlNttTables& t, hipStream_t stream);

void gl_intt_columns(const u64* src, u64* coeffs, u64* scratch, size_t ncols, const GlNttTables& t, hipStream_t stream) {
    if (t.sub) {   // more than 2^22 rows: one outer radix step around the sub-transforms (ntt_outer.hpp)
        outer::intt_columns<GlF>(src, coeffs, scratch, ncols, t.log_n, t.outer_bits, t.tw_hi_inv, t.tw_lo_inv,
                                 [&](const u64* s, u64* d, u64* scr, size_t nc) { gl_intt_columns(s, d, scr, nc, *t.sub, stream); }, stream);
        return;
    }
    const size_t g = t.log_n > 20 ? INTT_GROUP >> (t.log_n - 20) : INTT_GROUP, n = (size_t)1 << t.log_n;   // the same bytes per group from 2^20 rows up
    if (t.log_n < 18 || ncols <= g) return gl_intt_group(src, coeffs, scratch, ncols, t, stream);
    for (size_t c0 = 0; c0 < ncols; c0 += g)   // the scratch block of one group is reused by the next: it never leaves the cache
        gl_intt_group(src + c0 * n, coeffs + c0 * n, scratch, std::min(g, ncols - c0), t, stream);
}

static void gl_intt_group(const u64* src, u64* coeffs, u64* scratch, size_t ncols, const GlNttTables& t, hipStream_t stream) {
    const u32 L = t.log_n;
    if (ncols == 0) return;
    if (L <= 12) {
        hipLaunchKernelGGL(k_gl_ntt_small, dim3((u32)ncols), dim3(THREADS), 0, stream, src, coeffs, L, t.tw4096_inv,
                           t.n_inv);
        return;
    }
    if (gl_intt_columns_r16(src, coeffs, scratch, ncols, t, stream)) return;
    InvGeom g = inv_geom(L);
    const u32 LL = g.LB + g.LC;
    u64* p1_dst = g.LB ? coeffs : scratch;
    hipLaunchKernelGGL(k_gl_intt_p1, dim3((u32)(ncols << (LL - 4))), dim3(THREADS), 0, stream, src, p1_dst, g,
                       t.tw4096_inv, t.tw_hi_inv, t.tw_lo_inv);
    if (g.LB) {
        hipLaunchKernelGGL(k_gl_intt_p2, dim3((u32)(ncols << (g.LA - 4 + g.LC - 4))), dim3(THREADS), 0, stream, coeffs,
                           scratch, g, t.tw4096_inv);
    }
    hipLaunchKernelGGL(k_gl_intt_p3, dim3((u32)(ncols << (g.LB + g.LA - 4))), dim3(THREADS), 0, stream, scratch, coeffs,
                       g, t.tw4096_inv, t.n_inv);
}

void gl_lde_columns(const u64* coeffs, u64* lde, size_t ncols, const GlNttTables& t, const GlCosetTables& ct,
                    hipStream_t stream) {
    const u32 L = t.log_n, r = ct.rate_bits;
    if (ncols == 0) return;
    if (t.sub) {
        outer::lde_columns<GlF>(coeffs, lde, ncols, L, t.outer_bits, r, t.tw_hi_fwd, t.tw_lo_fwd, t.tw_top_fwd, ct.pow_lo, (u64*)*ct.work, *ct.work_bytes / sizeof(u64),
                                [&](const u64* c, u64* o, size_t nc) { gl_lde_columns(c, o, nc, *t.sub, *ct.sub, stream); }, stream);
        return;
    }
    if (L <= 12) {
        hipLaunchKernelGGL(k_gl_lde_pb<true>, dim3((u32)(ncols << r)), dim3(THREADS), 0, stream, coeffs, lde, L, r,
                           t.tw4096_fwd, ct.pow_lo);
        return;
    }
    // both passes over all columns per launch: they are bound by VALU issue, not by HBM (column groups sized for the Infinity Cache
    // measured slower in every setting, HISTORY.md round 3)
    if (!gl_lde_pa_r16(coeffs, lde, ncols, t, ct, stream))
        hipLaunchKernelGGL(k_gl_lde_pa, dim3((u32)(ncols << 8)), dim3(THREADS), 0, stream, coeffs, lde, L, r, t.tw4096_fwd, t.tw_hi_fwd,
                           t.tw_lo_fwd, ct.pow_lo, ct.pow_hi);
    gl_lde_pb_r16(lde, ncols << (r + L - 12), t, stream);
}

}  // namespace gbk
