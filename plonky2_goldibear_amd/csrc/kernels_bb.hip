// BabyBear kernels for gfx950: LDS-tiled NTT / LDE passes (same pass structure and layouts as
// kernels_ntt.hip, 32-bit Montgomery arithmetic) and the Poseidon2-16 Merkle tree (H = 8).
//
// Replaces, for F = BabyBear, the same reference code as the Goldilocks kernels:
// fri/oracle.rs:68-150 (IFFT, FFT + blinding, transpose + bit-reverse folded into the leaf order),
// hash/merkle_tree.rs:86-181 with Poseidon2BabyBearHash (hash/poseidon2_babybear.rs:163-176).
// Device-resident element data is in Montgomery form; digests are canonical.
#include <algorithm>

#include "kernels.hpp"
#include "ntt_outer.hpp"
#include "poseidon2_bb.hpp"
#include "poseidon2_bb_coop.hpp"

namespace gbk {

static constexpr int THREADS = 256;
static constexpr int TILE = 4096;

__device__ __forceinline__ u32 brevb(u32 x, u32 bits) { return bits ? (__brev(x) >> (32 - bits)) : 0; }

// ------------------------------------------------------------------ NTT (see kernels_ntt.hip for the derivation)

__device__ __forceinline__ void bb_lds_dft(u32* sh, u32 tile_elems, u32 p, u32 m, const u32* __restrict__ tw) {
    const u32 half = tile_elems >> 1;
    for (u32 l = 0; l < m; l++) {
        const u32 bitpos = p + m - 1 - l;
        const u32 hmask = (1u << (m - 1 - l)) - 1;
        const u32 lowmask = (1u << bitpos) - 1;
        for (u32 q = threadIdx.x; q < half; q += THREADS) {
            u32 e1 = ((q >> bitpos) << (bitpos + 1)) | (q & lowmask);
            u32 e2 = e1 | (1u << bitpos);
            u32 j = (e1 >> p) & hmask;
            u32 a = sh[e1], b = sh[e2];
            u32 d = bb::sub(a, b);
            if (j) d = bb::mul(d, tw[(j << l) << (12 - m)]);
            sh[e1] = bb::add(a, b);
            sh[e2] = d;
        }
        __syncthreads();
    }
}
__device__ __forceinline__ u32 bb_tw_split(const u32* __restrict__ hi, const u32* __restrict__ lo, u32 e) {
    u32 eh = e >> 10, el = e & 1023;
    u32 w = lo[el];
    return eh ? bb::mul(w, hi[eh]) : w;
}

struct BbInvGeom {
    u32 L, LA, LB, LC;
};

__global__ __launch_bounds__(THREADS) void k_bb_intt_p1(const u32* __restrict__ src, u32* __restrict__ dst, BbInvGeom g,
                                                        const u32* __restrict__ tw4096, const u32* __restrict__ tw_hi,
                                                        const u32* __restrict__ tw_lo) {
    __shared__ u32 sh[TILE];
    const u32 LL = g.LB + g.LC;
    const u32 tiles_per_col = 1u << (LL - 4);
    const size_t col = blockIdx.x / tiles_per_col;
    const u32 tg = blockIdx.x % tiles_per_col;
    const size_t base = (col << g.L) + ((size_t)tg << 4);
    const u32 rows = 1u << g.LA;
    const u32 j = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    for (u32 a = r0; a < rows; a += 16) sh[a * 16 + j] = src[base + ((size_t)a << LL) + j];
    __syncthreads();
    bb_lds_dft(sh, rows * 16, 4, g.LA, tw4096);
    const u32 l = (tg << 4) + j;
    for (u32 ka = r0; ka < rows; ka += 16) {
        u32 v = sh[brevb(ka, g.LA) * 16 + j];
        u32 e = ka * l;
        if (e) v = bb::mul(v, bb_tw_split(tw_hi, tw_lo, e));
        dst[base + ((size_t)ka << LL) + j] = v;
    }
}
__global__ __launch_bounds__(THREADS) void k_bb_intt_p2(const u32* __restrict__ src, u32* __restrict__ dst, BbInvGeom g,
                                                        const u32* __restrict__ tw4096) {
    __shared__ u32 sh[TILE];
    const u32 LL = g.LB + g.LC;
    const u32 n_ga = 1u << (g.LA - 4), n_gc = 1u << (g.LC - 4);
    const size_t col = blockIdx.x / (n_ga * n_gc);
    const u32 rem = blockIdx.x % (n_ga * n_gc);
    const u32 ga = rem / n_gc, gc = rem % n_gc;
    const size_t cbase = col << g.L;
    const u32 nb = 1u << g.LB;
    const u32 jc = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    for (u32 r = r0; r < 16 * nb; r += 16) {
        u32 ia = r >> g.LB, b = r & (nb - 1);
        sh[r * 16 + jc] = src[cbase + ((size_t)(16 * ga + ia) << LL) + ((size_t)b << g.LC) + 16 * gc + jc];
    }
    __syncthreads();
    bb_lds_dft(sh, 16 * nb * 16, 4, g.LB, tw4096);
    const u32 c = 16 * gc + jc;
    for (u32 r = r0; r < 16 * nb; r += 16) {
        u32 ia = r >> g.LB, kb = r & (nb - 1);
        u32 v = sh[(ia * nb + brevb(kb, g.LB)) * 16 + jc];
        u32 e = c * kb;
        if (e) v = bb::mul(v, tw4096[e << (12 - LL)]);
        dst[cbase + ((size_t)kb << (g.LA + g.LC)) + ((size_t)(16 * ga + ia) << g.LC) + c] = v;
    }
}
__global__ __launch_bounds__(THREADS) void k_bb_intt_p3(const u32* __restrict__ src, u32* __restrict__ dst, BbInvGeom g,
                                                        const u32* __restrict__ tw4096, u32 n_inv) {
    __shared__ u32 sh[TILE];
    const u32 n_ga = 1u << (g.LA - 4), nb = 1u << g.LB, nc = 1u << g.LC;
    const size_t col = blockIdx.x / (nb * n_ga);
    const u32 rem = blockIdx.x % (nb * n_ga);
    const u32 kb = rem / n_ga, ga = rem % n_ga;
    const size_t cbase = col << g.L;
    const size_t sbase = cbase + ((size_t)kb << (g.LA + g.LC)) + ((size_t)(16 * ga) << g.LC);
    for (u32 t = threadIdx.x; t < 16 * nc; t += THREADS) sh[t] = src[sbase + t];
    __syncthreads();
    bb_lds_dft(sh, 16 * nc, 0, g.LC, tw4096);
    const u32 ia = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    for (u32 kc = r0; kc < nc; kc += 16) {
        u32 v = bb::mul(sh[ia * nc + brevb(kc, g.LC)], n_inv);
        dst[cbase + ((size_t)kc << (g.LA + g.LB)) + ((size_t)kb << g.LA) + 16 * ga + ia] = v;
    }
}
__global__ __launch_bounds__(THREADS) void k_bb_ntt_small(const u32* __restrict__ src, u32* __restrict__ dst, u32 L,
                                                          const u32* __restrict__ tw4096, u32 scale) {
    __shared__ u32 sh[TILE];
    const u32 n = 1u << L;
    const size_t base = (size_t)blockIdx.x << L;
    for (u32 t = threadIdx.x; t < n; t += THREADS) sh[t] = src[base + t];
    __syncthreads();
    bb_lds_dft(sh, n, 0, L, tw4096);
    for (u32 k = threadIdx.x; k < n; k += THREADS) dst[base + k] = bb::mul(sh[brevb(k, L)], scale);
}
__global__ __launch_bounds__(THREADS) void k_bb_lde_pa(const u32* __restrict__ coeffs, u32* __restrict__ lde, u32 L, u32 rate_bits,
                                                       const u32* __restrict__ tw4096, const u32* __restrict__ tw_hi,
                                                       const u32* __restrict__ tw_lo, const u32* __restrict__ pow_lo,
                                                       const u32* __restrict__ pow_hi) {
    __shared__ u32 sh[TILE];
    const u32 LA = L - 12;
    const u32 rows = 1u << LA;
    const size_t col = blockIdx.x >> 8;
    const u32 tg = blockIdx.x & 255;
    const u32 j = threadIdx.x & 15, r0 = threadIdx.x >> 4;
    const u32 l = (tg << 4) + j;
    const size_t n = (size_t)1 << L;
    const u32* cin = coeffs + col * n + l;
    u32 orig[16];
#pragma unroll
    for (u32 it = 0; it < 16; it++) {
        u32 a = r0 + 16 * it;
        orig[it] = a < rows ? cin[(size_t)a << 12] : 0;
    }
    const u32 ncosets = 1u << rate_bits;
    for (u32 c = 0; c < ncosets; c++) {
        const u32* ph = pow_hi + (size_t)c * rows;
#pragma unroll
        for (u32 it = 0; it < 16; it++) {
            u32 a = r0 + 16 * it;
            if (a < rows) sh[a * 16 + j] = a ? bb::mul(orig[it], ph[a]) : orig[it];
        }
        __syncthreads();
        bb_lds_dft(sh, rows * 16, 4, LA, tw4096);
        const u32 sl = pow_lo[(size_t)c * 4096 + l];
        u32* out = lde + (col << (L + rate_bits)) + (size_t)c * n + l;
#pragma unroll
        for (u32 it = 0; it < 16; it++) {
            u32 pa = r0 + 16 * it;
            if (pa < rows) {
                u32 e = brevb(pa, LA) * l;
                u32 f = e ? bb::mul(sl, bb_tw_split(tw_hi, tw_lo, e)) : sl;
                out[(size_t)pa << 12] = bb::mul(sh[pa * 16 + j], f);
            }
        }
        __syncthreads();
    }
}
template <bool FROM_COEFFS>
__global__ __launch_bounds__(THREADS) void k_bb_lde_pb(const u32* __restrict__ coeffs, u32* __restrict__ lde, u32 L, u32 rate_bits,
                                                       const u32* __restrict__ tw4096, const u32* __restrict__ pow_lo) {
    __shared__ u32 sh[TILE];
    const u32 LT = L < 12 ? L : 12;
    const u32 te = 1u << LT;
    const size_t tile = blockIdx.x;
    u32* p = lde + (tile << LT);
    if (FROM_COEFFS) {
        const size_t col = tile >> rate_bits;
        const u32 c = (u32)(tile & ((1u << rate_bits) - 1));
        const u32* cin = coeffs + (col << L);
        const u32* pl = pow_lo + ((size_t)c << LT);
        for (u32 t = threadIdx.x; t < te; t += THREADS) sh[t] = bb::mul(cin[t], pl[t]);
    } else {
        for (u32 t = threadIdx.x; t < te; t += THREADS) sh[t] = p[t];
    }
    __syncthreads();
    bb_lds_dft(sh, te, 0, LT, tw4096);
    for (u32 t = threadIdx.x; t < te; t += THREADS) p[t] = sh[t];
}

// ------------------------------------------------------------------ Merkle (Poseidon2, rate 8, digest 8 x u32 canonical)

__global__ __launch_bounds__(256) void k_bb_merkle_leaves(const u32* __restrict__ cols, size_t col_stride, u32 width, u64 num_leaves,
                                                          u32* __restrict__ out) {
    u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= num_leaves) return;
    u32 s[16];
#pragma unroll
    for (int i = 0; i < 16; i++) s[i] = 0;
    uint4* o = reinterpret_cast<uint4*>(out + 8 * j);
    if (width <= 8) {  // hash_or_noop (plonk/config.rs:70-84), NUM_HASH_OUT_ELTS = 8
        for (u32 c = 0; c < width; c++) s[c] = bb::from_mont(cols[(size_t)c * col_stride + j]);
        o[0] = make_uint4(s[0], s[1], s[2], s[3]);
        o[1] = make_uint4(s[4], s[5], s[6], s[7]);
        return;
    }
    for (u32 c0 = 0; c0 < width; c0 += 8) {  // one loop, one inlined copy of the permutation
        if (c0) {  // the capacity words go on at scale 1; the rate words are overwritten (partially in the last absorption)
#pragma unroll
            for (int i = 8; i < 16; i++) s[i] = poseidon2_bb::renorm_lazy(s[i]);
            if (c0 + 8 > width) {
#pragma unroll
                for (int i = 0; i < 8; i++) s[i] = poseidon2_bb::renorm_lazy(s[i]);
            }
        }
        if (c0 + 8 <= width) {
#pragma unroll
            for (int i = 0; i < 8; i++) s[i] = cols[(size_t)(c0 + i) * col_stride + j];
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (c0 + i < width) s[i] = cols[(size_t)(c0 + i) * col_stride + j];
        }
        poseidon2_bb::permute_scaled(s);
    }
    using poseidon2_bb::canonical_out;
    o[0] = make_uint4(canonical_out(s[0]), canonical_out(s[1]), canonical_out(s[2]), canonical_out(s[3]));
    o[1] = make_uint4(canonical_out(s[4]), canonical_out(s[5]), canonical_out(s[6]), canonical_out(s[7]));
}
// The leaf sponge in column SEGMENTS [c_begin, c_end) (see k_gl_merkle_leaves_seg): between segments the capacity words 8..15 - as
// the permutation left them: the next absorption's renorm brings them back to scale 1 - and, in front of a ragged last
// absorption, the rate words it leaves alone (`keep_from`..7) wait in `state` [8 + 8 - keep_from][num_leaves].
template <bool FIRST, bool LAST>
__global__ __launch_bounds__(256, 7) void k_bb_merkle_leaves_seg(const u32* __restrict__ cols, size_t col_stride, u32 c_begin, u32 c_end,
                                                                 u64 num_leaves, u32* __restrict__ state, u32 keep_from,
                                                                 u32* __restrict__ out) {
    u64 j = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= num_leaves) return;
    u32 s[16];
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = 0;
    // `state` holds only the rows that are used: [0, 8) the capacity words 8..15, then rate word i >= keep_from at row 8 + i - keep_from
#pragma unroll
    for (int i = 8; i < 16; i++) s[i] = FIRST ? 0 : state[(size_t)(i - 8) * num_leaves + j];
    if (LAST && !FIRST && c_end - c_begin < 8) {
        const u32 kf = c_end - c_begin;   // = the keep_from the segment before was given
#pragma unroll
        for (int i = 1; i < 8; i++)
            if ((u32)i >= kf) s[i] = state[(size_t)(8 + i - kf) * num_leaves + j];
    }
    for (u32 c0 = c_begin; c0 < c_end; c0 += 8) {
        if (c0) {  // the capacity words go on at scale 1; the rate words are overwritten (partially in the last absorption)
#pragma unroll
            for (int i = 8; i < 16; i++) s[i] = poseidon2_bb::renorm_lazy(s[i]);
            if (LAST && c0 + 8 > c_end) {
#pragma unroll
                for (int i = 0; i < 8; i++) s[i] = poseidon2_bb::renorm_lazy(s[i]);
            }
        }
        if (!LAST || c0 + 8 <= c_end) {
#pragma unroll
            for (int i = 0; i < 8; i++) s[i] = cols[(size_t)(c0 + i) * col_stride + j];
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (c0 + i < c_end) s[i] = cols[(size_t)(c0 + i) * col_stride + j];
        }
        poseidon2_bb::permute_scaled(s);
    }
    if (!LAST) {
#pragma unroll
        for (int i = 8; i < 16; i++) state[(size_t)(i - 8) * num_leaves + j] = s[i];
        if (keep_from < 8) {
#pragma unroll
            for (int i = 1; i < 8; i++)
                if ((u32)i >= keep_from) state[(size_t)(8 + i - keep_from) * num_leaves + j] = s[i];
        }
        return;
    }
    uint4* o = reinterpret_cast<uint4*>(out + 8 * j);
    using poseidon2_bb::canonical_out;
    o[0] = make_uint4(canonical_out(s[0]), canonical_out(s[1]), canonical_out(s[2]), canonical_out(s[3]));
    o[1] = make_uint4(canonical_out(s[4]), canonical_out(s[5]), canonical_out(s[6]), canonical_out(s[7]));
}
__global__ __launch_bounds__(256) void k_bb_merkle_level(const u32* __restrict__ in, u32* __restrict__ out, u64 num_out) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= num_out) return;
    const uint4* p = reinterpret_cast<const uint4*>(in + 16 * i);
    uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    u32 s[16] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y, d.z, d.w};
#pragma unroll
    for (int k = 0; k < 16; k++) s[k] = bb::to_mont(s[k]);
    poseidon2_bb::permute_scaled(s);
    using poseidon2_bb::canonical_out;
    uint4* o = reinterpret_cast<uint4*>(out + 8 * i);
    o[0] = make_uint4(canonical_out(s[0]), canonical_out(s[1]), canonical_out(s[2]), canonical_out(s[3]));
    o[1] = make_uint4(canonical_out(s[4]), canonical_out(s[5]), canonical_out(s[6]), canonical_out(s[7]));
}
// The same two kernels (and the FRI layer leaves) with one state per 16-lane row (poseidon2_bb_coop.hpp) for small trees.
__global__ __launch_bounds__(64) void k_bb_merkle_level_coop(const u32* __restrict__ in, u32* __restrict__ out, u64 num_out) {
    const u32 l = threadIdx.x & 15;
    const u64 node = (u64)blockIdx.x * 4 + (threadIdx.x >> 4);
    const bool valid = node < num_out;
    u32 x = bb::to_mont(in[16 * (valid ? node : 0) + l]);
    x = poseidon2_bb_coop::permute(x, l);
    if (valid && l < 8) out[8 * node + l] = bb::from_mont(x);
}
// width > 8 (narrower leaves are not hashed, plonk/config.rs:70-84); cols in Montgomery form
__global__ __launch_bounds__(64) void k_bb_merkle_leaves_coop(const u32* __restrict__ cols, size_t col_stride, u32 width,
                                                              u64 num_leaves, u32* __restrict__ out) {
    const u32 l = threadIdx.x & 15;
    const u64 leaf = (u64)blockIdx.x * 4 + (threadIdx.x >> 4);
    const bool valid = leaf < num_leaves;
    const u64 j = valid ? leaf : 0;
    u32 x = 0;
    for (u32 c0 = 0; c0 < width; c0 += 8) {
        if (l < 8 && c0 + l < width) x = cols[(size_t)(c0 + l) * col_stride + j];  // overwrite-mode absorption
        x = poseidon2_bb_coop::permute(x, l);
    }
    if (valid && l < 8) out[8 * leaf + l] = bb::from_mont(x);
}
// FRI layer leaves (fri/prover.rs:101-107), D = 4: vals = [4][len] coordinate columns (Montgomery), 4 * arity > 8 only
__global__ __launch_bounds__(64) void k_bb_fri_leaves_coop(const u32* __restrict__ vals, size_t len, u32 arity_bits, u64 num_leaves,
                                                           u32* __restrict__ out) {
    const u32 l = threadIdx.x & 15;
    const u64 leaf = (u64)blockIdx.x * 4 + (threadIdx.x >> 4);
    const bool valid = leaf < num_leaves;
    const u32* a = vals + ((valid ? leaf : 0) << arity_bits);
    const u32 arity = 1u << arity_bits;
    u32 x = 0;
    for (u32 k0 = 0; k0 < arity; k0 += 2) {   // two extension elements = eight base elements per absorption
        const u32 k = k0 + (l >> 2);
        if (l < 8 && k < arity) x = a[(size_t)(l & 3) * len + k];
        x = poseidon2_bb_coop::permute(x, l);
    }
    if (valid && l < 8) out[8 * leaf + l] = bb::from_mont(x);
}

__global__ __launch_bounds__(256) void k_bb_permute(const u32* __restrict__ in, u32* __restrict__ out, u64 count) {
    u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    u32 s[16];
#pragma unroll
    for (int e = 0; e < 16; e++) s[e] = bb::to_mont(in[16 * i + e]);
    poseidon2_bb::permute(s);
#pragma unroll
    for (int e = 0; e < 16; e++) out[16 * i + e] = bb::from_mont(s[e]);
}

// ------------------------------------------------------------------ conversions / gathers (u32)
__global__ void k_bb_to_mont(const u32* __restrict__ src, u32* __restrict__ dst, size_t n) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) dst[g] = bb::to_mont(src[g]);
}
__global__ void k_bb_from_mont(const u32* __restrict__ src, u32* __restrict__ dst, size_t n) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g < n) dst[g] = bb::from_mont(src[g]);
}
__global__ void k_bb_gather_row(const u32* __restrict__ cols, size_t col_stride, u32 width, u64 index, u32* dst) {
    u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < width) dst[c] = bb::from_mont(cols[(size_t)c * col_stride + index]);
}
__global__ void k_bb_bitrev_copy_to_mont(const u32* __restrict__ src, u32* __restrict__ dst, u32 bits, size_t total) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    size_t col = g >> bits;
    u64 j = g & (((u64)1 << bits) - 1);
    u64 r = bits ? (__brevll(j) >> (64 - bits)) : 0;
    dst[g] = bb::to_mont(src[(col << bits) + r]);
}
__global__ void k_bb_transpose_to_rows(const u32* __restrict__ cols, size_t col_stride, u32 width, u64 rows, u32* __restrict__ dst) {
    u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= rows * width) return;
    u64 r = g / width;
    u32 c = (u32)(g % width);
    dst[g] = bb::from_mont(cols[(size_t)c * col_stride + r]);
}

// ------------------------------------------------------------------ launchers
static inline u32 nblk(size_t n, u32 bs) { return (u32)((n + bs - 1) / bs); }

// radix-16 register kernels (kernels_bb16.hip); return false when the shape is not covered
bool bb_intt_columns_r16(const u32* src, u32* coeffs, u32* scratch, size_t ncols, const BbNttTables& t, hipStream_t stream,
                         u32* canonical_src = nullptr, size_t mont_cols = 0);
bool bb_lde_pa_r16(const u32* coeffs, u32* lde, size_t ncols, const BbNttTables& t, const BbCosetTables& ct, hipStream_t stream);

static void bb_intt_group(const u32* src, u32* coeffs, u32* scratch, size_t ncols, const BbNttTables& t, hipStream_t stream);
// values -> coefficients from CANONICAL values (kernels_bb16.hip, k_bb_intt16_p1<true>): no conversion pass; the first mont_cols
// columns of `vals` are left in Montgomery form, the rest canonical.  false: shape not covered (2^16..2^20 rows are), nothing done.
bool bb_intt_columns_canonical(u32* vals, u32* coeffs, u32* scratch, size_t ncols, size_t mont_cols, const BbNttTables& t, hipStream_t stream) {
    if (t.log_n < 16 || t.log_n > NTT_NATIVE_LOG) return false;
    const size_t g = t.log_n > 20 ? (2 * INTT_GROUP) >> (t.log_n - 20) : 2 * INTT_GROUP, n = (size_t)1 << t.log_n;
    if (t.log_n < 18 || ncols <= g) {   // the grouping rule of bb_intt_columns: groups from 2^18 rows
        bb_intt_columns_r16(nullptr, coeffs, scratch, ncols, t, stream, vals, mont_cols);
        return true;
    }
    for (size_t c0 = 0; c0 < ncols; c0 += g)
        bb_intt_columns_r16(nullptr, coeffs + c0 * n, scratch, std::min(g, ncols - c0), t, stream, vals + c0 * n, mont_cols > c0 ? mont_cols - c0 : 0);
    return true;
}
void bb_intt_columns(const u32* src, u32* coeffs, u32* scratch, size_t ncols, const BbNttTables& t, hipStream_t stream) {
    if (t.sub) {   // more than 2^22 rows: one outer radix step around the sub-transforms (ntt_outer.hpp)
        outer::intt_columns<BbF>(src, coeffs, scratch, ncols, t.log_n, t.outer_bits, t.tw_hi_inv, t.tw_lo_inv,
                                 [&](const u32* s, u32* d, u32* scr, size_t nc) { bb_intt_columns(s, d, scr, nc, *t.sub, stream); }, stream);
        return;
    }
    const size_t g = t.log_n > 20 ? (2 * INTT_GROUP) >> (t.log_n - 20) : 2 * INTT_GROUP, n = (size_t)1 << t.log_n;   // 4-byte words: twice Goldilocks' columns per group
    if (t.log_n < 18 || ncols <= g) return bb_intt_group(src, coeffs, scratch, ncols, t, stream);
    for (size_t c0 = 0; c0 < ncols; c0 += g)
        bb_intt_group(src + c0 * n, coeffs + c0 * n, scratch, std::min(g, ncols - c0), t, stream);
}
void bb_lde_pb_r16(u32* lde, size_t ntiles, const BbNttTables& t, hipStream_t stream);

static void bb_intt_group(const u32* src, u32* coeffs, u32* scratch, size_t ncols, const BbNttTables& t, hipStream_t stream) {
    const u32 L = t.log_n;
    if (!ncols) return;
    if (L <= 12) {
        hipLaunchKernelGGL(k_bb_ntt_small, dim3((u32)ncols), dim3(THREADS), 0, stream, src, coeffs, L, t.tw4096_inv, t.n_inv);
        return;
    }
    if (bb_intt_columns_r16(src, coeffs, scratch, ncols, t, stream)) return;
    BbInvGeom g{L, L <= 16 ? L - 8 : 8, L <= 16 ? 0 : L - 16, 8};
    const u32 LL = g.LB + g.LC;
    u32* p1_dst = g.LB ? coeffs : scratch;
    hipLaunchKernelGGL(k_bb_intt_p1, dim3((u32)(ncols << (LL - 4))), dim3(THREADS), 0, stream, src, p1_dst, g, t.tw4096_inv,
                       t.tw_hi_inv, t.tw_lo_inv);
    if (g.LB)
        hipLaunchKernelGGL(k_bb_intt_p2, dim3((u32)(ncols << (g.LA - 4 + g.LC - 4))), dim3(THREADS), 0, stream, coeffs, scratch,
                           g, t.tw4096_inv);
    hipLaunchKernelGGL(k_bb_intt_p3, dim3((u32)(ncols << (g.LB + g.LA - 4))), dim3(THREADS), 0, stream, scratch, coeffs, g,
                       t.tw4096_inv, t.n_inv);
}
void bb_lde_columns(const u32* coeffs, u32* lde, size_t ncols, const BbNttTables& t, const BbCosetTables& ct, hipStream_t stream) {
    const u32 L = t.log_n, r = ct.rate_bits;
    if (!ncols) return;
    if (t.sub) {
        outer::lde_columns<BbF>(coeffs, lde, ncols, L, t.outer_bits, r, t.tw_hi_fwd, t.tw_lo_fwd, t.tw_top_fwd, ct.pow_lo, (u32*)*ct.work, *ct.work_bytes / sizeof(u32),
                                [&](const u32* c, u32* o, size_t nc) { bb_lde_columns(c, o, nc, *t.sub, *ct.sub, stream); }, stream);
        return;
    }
    if (L <= 12) {
        hipLaunchKernelGGL(k_bb_lde_pb<true>, dim3((u32)(ncols << r)), dim3(THREADS), 0, stream, coeffs, lde, L, r, t.tw4096_fwd,
                           ct.pow_lo);
        return;
    }
    if (!bb_lde_pa_r16(coeffs, lde, ncols, t, ct, stream))
        hipLaunchKernelGGL(k_bb_lde_pa, dim3((u32)(ncols << 8)), dim3(THREADS), 0, stream, coeffs, lde, L, r, t.tw4096_fwd, t.tw_hi_fwd,
                           t.tw_lo_fwd, ct.pow_lo, ct.pow_hi);
    bb_lde_pb_r16(lde, ncols << (r + L - 12), t, stream);
}
// as for Goldilocks (kernels_merkle.hip): below this many states the lane-per-state kernels are latency-bound
static constexpr u64 BB_COOP_MAX_STATES = 16384;

void bb_merkle_leaves(const u32* cols, size_t col_stride, u32 width, u64 num_leaves, u32* out, hipStream_t stream) {
    if (width > 8 && num_leaves <= BB_COOP_MAX_STATES) {
        hipLaunchKernelGGL(k_bb_merkle_leaves_coop, dim3(nblk(num_leaves, 4)), dim3(64), 0, stream, cols, col_stride, width, num_leaves, out);
        return;
    }
    hipLaunchKernelGGL(k_bb_merkle_leaves, dim3(nblk(num_leaves, 256)), dim3(256), 0, stream, cols, col_stride, width, num_leaves, out);
}
void bb_merkle_leaves_segment(const u32* cols, size_t col_stride, u32 c_begin, u32 c_end, u64 num_leaves, u32* state, bool last,
                              u32 next_cols, u32* out, hipStream_t stream) {
    const dim3 grid(nblk(num_leaves, 256)), block(256);
    const u32 keep_from = next_cols < 8u ? next_cols : 8u;
    if (c_begin == 0 && !last)
        hipLaunchKernelGGL((k_bb_merkle_leaves_seg<true, false>), grid, block, 0, stream, cols, col_stride, c_begin, c_end, num_leaves, state, keep_from, out);
    else if (!last)
        hipLaunchKernelGGL((k_bb_merkle_leaves_seg<false, false>), grid, block, 0, stream, cols, col_stride, c_begin, c_end, num_leaves, state, keep_from, out);
    else
        hipLaunchKernelGGL((k_bb_merkle_leaves_seg<false, true>), grid, block, 0, stream, cols, col_stride, c_begin, c_end, num_leaves, state, keep_from, out);
}
bool bb_fri_leaves_coop(const u32* vals, size_t len, u32 arity_bits, u64 num_leaves, u32* out, hipStream_t stream) {
    if (num_leaves > BB_COOP_MAX_STATES || (4u << arity_bits) <= 8) return false;
    hipLaunchKernelGGL(k_bb_fri_leaves_coop, dim3(nblk(num_leaves, 4)), dim3(64), 0, stream, vals, len, arity_bits, num_leaves, out);
    return true;
}
void bb_merkle_level(const u32* in, u32* out, u64 num_out, hipStream_t stream) {
    if (num_out <= BB_COOP_MAX_STATES) {
        hipLaunchKernelGGL(k_bb_merkle_level_coop, dim3(nblk(num_out, 4)), dim3(64), 0, stream, in, out, num_out);
        return;
    }
    hipLaunchKernelGGL(k_bb_merkle_level, dim3(nblk(num_out, 256)), dim3(256), 0, stream, in, out, num_out);
}
void bb_poseidon2_permute(const u32* in, u32* out, u64 count, hipStream_t stream) {
    hipLaunchKernelGGL(k_bb_permute, dim3(nblk(count, 256)), dim3(256), 0, stream, in, out, count);
}
// any u32 word -> its residue below p, in place, 4 words per thread (GB_INPUT_P3_REPR: p3-monty-31 keeps its Montgomery words below
// p, but the words are the caller's memory: one out of range would otherwise go into the transforms as it is - ADVICE r5)
__global__ __launch_bounds__(256) void k_bb_reduce_words(u32* __restrict__ p, size_t count) {
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    auto red = [](u32 x) { x = x >= 2 * bb::P ? x - 2 * bb::P : x; return x >= bb::P ? x - bb::P : x; };
    if (i + 3 < count) {
        uint4 v = *reinterpret_cast<uint4*>(p + i);
        v.x = red(v.x); v.y = red(v.y); v.z = red(v.z); v.w = red(v.w);
        *reinterpret_cast<uint4*>(p + i) = v;
    } else {
        for (size_t k = i; k < count; k++) p[k] = red(p[k]);
    }
}
void bb_reduce_words(u32* p, size_t n, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_bb_reduce_words, dim3(nblk(n, 1024)), dim3(256), 0, stream, p, n);
}
void bb_to_mont(const u32* src, u32* dst, size_t n, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_bb_to_mont, dim3(nblk(n, 256)), dim3(256), 0, stream, src, dst, n);
}
void bb_from_mont(const u32* src, u32* dst, size_t n, hipStream_t stream) {
    if (n) hipLaunchKernelGGL(k_bb_from_mont, dim3(nblk(n, 256)), dim3(256), 0, stream, src, dst, n);
}
void bb_gather_row(const u32* cols, size_t col_stride, u32 width, u64 index, u32* dst, hipStream_t stream) {
    hipLaunchKernelGGL(k_bb_gather_row, dim3(nblk(width, 64)), dim3(64), 0, stream, cols, col_stride, width, index, dst);
}
void bb_bitrev_copy_to_mont(const u32* src, u32* dst, u32 bits, size_t ncols, hipStream_t stream) {
    size_t total = ncols << bits;
    if (total) hipLaunchKernelGGL(k_bb_bitrev_copy_to_mont, dim3(nblk(total, 256)), dim3(256), 0, stream, src, dst, bits, total);
}
void bb_transpose_to_rows(const u32* cols, size_t col_stride, u32 width, u64 rows, u32* dst, hipStream_t stream) {
    if (rows && width)
        hipLaunchKernelGGL(k_bb_transpose_to_rows, dim3(nblk(rows * width, 256)), dim3(256), 0, stream, cols, col_stride, width, rows, dst);
}

}  // namespace gbk
