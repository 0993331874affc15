// Goldilocks NTT passes, v2: radix-16 butterflies held in registers (gfx950).
//
// Why: tools/microbench_valu.hip shows the NTT is bound by integer VALU issue, not by HBM or LDS
// (a 64-bit mod-mul is ~17-26 VALU ops at ~4.2 cycles each).  The reference's generator makes every
// root of unity of order <= 64 a power of two (w_64 = 2^39, w_16 = 2^156 = -2^60, w_4 = 2^48; 2^96 = -1),
// so a 16-point DFT needs no general multiplication: its 17 non-trivial twiddles are shifts + one fold.
// A thread keeps 16 points in registers for four layers; general twiddles (tables) appear only between
// radix-16 stages, and LDS is touched once per stage instead of once per layer.
//
// Same passes, tiles and memory layouts as kernels_ntt.hip (whose LDS radix-2 passes remain for the inverse transform of
// 2^13..2^15 rows and for single-tile sizes):
//   inverse:  P1 (8 bits strided)  P2 (0..4 bits)  P3 (8 bits, transposed write)            2^16..2^20 rows
//   LDE:      PA (4..8 bits strided, per-coset loop)  PB (12 bits contiguous, natural -> leaf order)
#include "kernels.hpp"
#include "gl_field.hpp"

namespace gbk {

static constexpr int THREADS = 256;

__device__ __forceinline__ constexpr u32 brev4(u32 x) { return ((x & 1) << 3) | ((x & 2) << 1) | ((x & 4) >> 1) | ((x & 8) >> 3); }

// x * 2^K mod p, canonical in, canonical out, 0 <= K < 96.  With f = 2^32 (f^2 = f - 1, f^3 = -1 mod p) and
// y = x 2^(K mod 32) = y0 + y1 f + y2 f^2 (y2 < 2^(K mod 32)), multiplying on by f^(K / 32) only shuffles limbs:
//   K <  32:  y0 + y1 f + y2 (f - 1)            = lo64 + y2 EPS             one conditional subtraction of p
//   K <  64:  (y0 + y1) f - (y1 + y2)           = (u0 + cu) f - (y1 + y2 + cu),  u = y0 + y1 = u0 + cu f
//   K <  96:  (y0 - y2) f - (y0 + y1)
// In the last two the minuend is a multiple of f below p and the subtrahend is small, so one "+ p on borrow" leaves the
// canonical value: 10-12 VALU ops where shift + generic fold + canonicalisation took 17 and the K >= 64 cases a full
// multiplication by a 64-bit constant (four of the seventeen twiddles of a 16-point DFT).
template <int K>
__device__ __forceinline__ u64 mul_pow2(u64 x) {
    static_assert(K >= 0 && K < 96, "shift range");
    if constexpr (K == 0) return x;
    constexpr int k = K % 32, j = K / 32;
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32);
    const u32 y0 = k ? x0 << k : x0;
    const u32 y1 = k ? (x1 << k) | (x0 >> ((32 - k) & 31)) : x1;
    const u32 y2 = k ? x1 >> ((32 - k) & 31) : 0u;
    if constexpr (j == 0) {
        const u64 lo = (u64)y0 | ((u64)y1 << 32), he = ((u64)y2 << 32) - y2;  // y2 EPS < 2^63
        u64 t, t2;
        const bool c = __builtin_uaddll_overflow(lo, he, &t);
        const bool d = __builtin_uaddll_overflow(t, gl::EPS, &t2);            // t - p
        return (c | d) ? t2 : t;
    } else {
        u32 cu, cw, b0, B1, B2, kk;
        u32 hi, s0;   // minuend hi f, subtrahend s0 + cw f (cw: carry flag)
        if constexpr (j == 1) {
            const u32 u0 = __builtin_addc(y0, y1, 0u, &cu);
            hi = u0 + cu;                                   // no overflow: cu = 1 leaves u0 <= 2^32 - 2
            s0 = __builtin_addc(y1, y2, cu, &cw);
        } else {
            hi = y0;
            s0 = __builtin_addc(y0, y1, 0u, &cw);
            cw += y2;                                       // high word of the subtrahend: y2 + carry (< 2^31 + 1)
        }
        const u32 d0 = __builtin_subc(0u, s0, 0u, &b0);
        const u32 d1 = __builtin_subc(hi, cw, b0, &B1);
        (void)B2;
        const u32 m = 0u - B1;                              // borrowed: + p, i.e. - EPS (mod 2^64)
        const u32 e0 = __builtin_subc(d0, m, 0u, &kk);
        const u32 e1 = d1 - kk;
        return (u64)e0 | ((u64)e1 << 32);
    }
}

// (a - b) * w_16^(+-M): every such root is +-2^K (forward w_16 = 2^156; inverse w_16^-1 = 2^36)
template <bool INV, int M>
__device__ __forceinline__ u64 sub_twiddle(u64 a, u64 b) {
    constexpr int e = (INV ? 36 * M : 156 * M) % 192;
    if constexpr (e == 0) return gl::sub(a, b);
    else if constexpr (e < 96) return mul_pow2<e>(gl::sub(a, b));
    else return mul_pow2<e - 96>(gl::sub(b, a));  // 2^96 = -1
}

template <bool INV, int H, int J>
__device__ __forceinline__ void bfly(u64& a, u64& b) {  // DIF butterfly of half-size H at offset J: twiddle w_{2H}^J = w_16^(J * 8 / H)
    u64 s = gl::add(a, b);
    b = sub_twiddle<INV, J*(8 / H)>(a, b);
    a = s;
}

// 16-point DFT in registers, natural input order, output X[k] in slot brev4(k) (in-place DIF)
template <bool INV>
__device__ __forceinline__ void dft16(u64 (&x)[16]) {
#define B8(J) bfly<INV, 8, J>(x[J], x[J + 8]);
    B8(0) B8(1) B8(2) B8(3) B8(4) B8(5) B8(6) B8(7)
#undef B8
#define B4(O, J) bfly<INV, 4, J>(x[O + J], x[O + J + 4]);
    B4(0, 0) B4(0, 1) B4(0, 2) B4(0, 3) B4(8, 0) B4(8, 1) B4(8, 2) B4(8, 3)
#undef B4
#define B2(O, J) bfly<INV, 2, J>(x[O + J], x[O + J + 2]);
    B2(0, 0) B2(0, 1) B2(4, 0) B2(4, 1) B2(8, 0) B2(8, 1) B2(12, 0) B2(12, 1)
#undef B2
#define B1(O) bfly<INV, 1, 0>(x[O], x[O + 1]);
    B1(0) B1(2) B1(4) B1(6) B1(8) B1(10) B1(12) B1(14)
#undef B1
}

// 2-, 4- and 8-point DFTs of the same family (the tail stages of dft16 on the first 2^K registers): natural input order,
// output X[k] in slot brevK(k).  They carry the sizes between the multiples of four bits (2^17..2^19 rows).
template <bool INV, int K>
__device__ __forceinline__ void dft_small(u64 (&x)[16]) {
    static_assert(K >= 1 && K <= 3, "radix 2, 4 or 8");
    if constexpr (K == 3) {
        bfly<INV, 4, 0>(x[0], x[4]); bfly<INV, 4, 1>(x[1], x[5]); bfly<INV, 4, 2>(x[2], x[6]); bfly<INV, 4, 3>(x[3], x[7]);
    }
    if constexpr (K >= 2) {
#pragma unroll
        for (int o = 0; o < (1 << K); o += 4) {
            bfly<INV, 2, 0>(x[o], x[o + 2]);
            bfly<INV, 2, 1>(x[o + 1], x[o + 3]);
        }
    }
#pragma unroll
    for (int o = 0; o < (1 << K); o += 2) bfly<INV, 1, 0>(x[o], x[o + 1]);
}
__device__ __forceinline__ constexpr u32 brevk(u32 x, int k) {
    u32 r = 0;
    for (int i = 0; i < k; i++) r |= ((x >> i) & 1) << (k - 1 - i);
    return r;
}

// (a - b) * w_64^(+-M): w_64 = 2^39 (w_64^4 = w_16 = 2^156), w_64^-1 = 2^153 - shifts like the roots of the 16-point DFT
template <bool INV, int M>
__device__ __forceinline__ u64 sub_twiddle64(u64 a, u64 b) {
    constexpr int e = (INV ? 153 * M : 39 * M) % 192;
    if constexpr (e == 0) return gl::sub(a, b);
    else if constexpr (e < 96) return mul_pow2<e>(gl::sub(a, b));
    else return mul_pow2<e - 96>(gl::sub(b, a));  // 2^96 = -1
}
// one DIF layer of half-size H (32 or 16) on x[O .. O + 2H): twiddle w_{2H}^J = w_64^(J * 32 / H)
template <bool INV, int H, int O, int J = 0>
__device__ __forceinline__ void layer64(u64* x) {
    if constexpr (J < H) {
        const u64 s = gl::add(x[O + J], x[O + J + H]);
        x[O + J + H] = sub_twiddle64<INV, J*(32 / H)>(x[O + J], x[O + J + H]);
        x[O + J] = s;
        layer64<INV, H, O, J + 1>(x);
    }
}

// 32-point DFT in registers, natural input order, output X[k] in slot brev5(k): one DIF layer, then two 16-point blocks
template <bool INV>
__device__ __forceinline__ void dft32(u64 (&x)[32]) {
    layer64<INV, 16, 0>(x);
    dft16<INV>(*reinterpret_cast<u64(*)[16]>(&x[0]));
    dft16<INV>(*reinterpret_cast<u64(*)[16]>(&x[16]));
}

__device__ __forceinline__ u64 tw_split16(const u64* __restrict__ hi, const u64* __restrict__ lo, u32 e) {
    u32 eh = e >> 10, el = e & 1023;
    u64 w = lo[el];
    return eh ? gl::mul_mont_lazy(w, hi[eh]) : w;   // Montgomery-form tables: (lo R)(hi R) / R = lo hi R, any residue
}

// The 15 inter-stage twiddles w_4096^(k m), k = brev4(slot), as ONE batch of independent loads issued before the
// DFT that precedes their use.  (Loaded one by one behind `if (e)` each of them cost a full memory round trip:
// rocprofv3 showed the waves of these kernels parked in s_waitcnt for 46-66 % of their lifetime.)
__device__ __forceinline__ void load_tw16(u64 (&tw)[16], const u64* __restrict__ tw4096, u32 m) {
#pragma unroll
    for (u32 s = 1; s < 16; s++) tw[s] = tw4096[brev4(s) * m];
}

// ------------------------------------------------------------------ LDE pass B: 4096 contiguous points, 3 radix-16 stages
// grid = number of 4096-tiles of `lde` (in place).  natural -> bit-reversed.
// The inter-stage twiddles cost as much as the butterflies here (ablations in DESIGN.md: without their loads and products the pass
// runs 33 % faster, close to a plain copy): the first stage reads them from a copy of the table laid out [slot][tid] (coalesced),
// the second from a 256-entry table in LDS.
__global__ __launch_bounds__(THREADS) void k_gl_lde_pb16(u64* __restrict__ lde, const u64* __restrict__ tw4096) {
    __shared__ u64 sh[16 * 272];
    __shared__ u64 tw2[256];  // the stage-2 twiddles as the threads read them: tw2[s][d0] = w_256^(brev4(s) d0) - consecutive lanes,
                              // consecutive words (indexed by exponent, slots with brev4(s) = 4, 8, 12 were 2- and 4-way bank conflicts)
    u64* p = lde + ((size_t)blockIdx.x << 12);
    const u32 tid = threadIdx.x;
    tw2[tid] = tw4096[((brev4(tid >> 4) * (tid & 15)) & 255) * 16];
    u64 x[16];
    // stage 1: digit d2 (stride 256); this thread is (d1, d0) = tid
#pragma unroll
    for (u32 d = 0; d < 16; d++) x[d] = p[d * 256 + tid];
    u64 tw[16];
#pragma unroll
    for (u32 s = 1; s < 16; s++) tw[s] = tw4096[4096 + s * 256 + tid];  // w_4096^(k2 (16 d1 + d0)), k2 = brev4(s)
    dft16<false>(x);
#pragma unroll
    for (u32 s = 0; s < 16; s++) sh[s * 272 + tid] = s ? gl::mul_mont(x[s], tw[s]) : x[s];
    __syncthreads();
    // stage 2: digit d1; this thread is (k2 slot, d0)
    const u32 hi4 = tid >> 4, lo4 = tid & 15;
#pragma unroll
    for (u32 d = 0; d < 16; d++) x[d] = sh[hi4 * 272 + d * 16 + lo4];
    dft16<false>(x);
    __syncthreads();
#pragma unroll
    for (u32 s = 0; s < 16; s++)   // w_256^(k1 d0) from LDS; [k2 slot][d0][k1 slot], rows padded to 17
        sh[hi4 * 272 + lo4 * 17 + s] = s ? gl::mul_mont(x[s], tw2[s * 16 + lo4]) : x[s];
    __syncthreads();
    // stage 3: digit d0; this thread is (k2 slot, k1 slot)
#pragma unroll
    for (u32 d = 0; d < 16; d++) x[d] = sh[hi4 * 272 + d * 17 + lo4];
    dft16<false>(x);
    __syncthreads();
    // x[s] belongs at tile position tid * 16 + s: transpose through LDS for a coalesced store
#pragma unroll
    for (u32 s = 0; s < 16; s++) sh[tid * 17 + s] = x[s];
    __syncthreads();
#pragma unroll
    for (u32 it = 0; it < 16; it++) {
        const u32 q = it * 256 + tid;
        p[q] = sh[(q >> 4) * 17 + (q & 15)];
    }
}

// ------------------------------------------------------------------ LDE pass A
// LA = 8: grid = ncols * 256, tile 256 rows (a = 16 a1 + a0) x 16 columns.  Per coset: scale by s^(4096 a),
// two radix-16 stages over a, twiddle w_n^(k_a l) s^l, in-place DIF row order.
// (Its products keep round 3's five-mad form, mul_mont<true>: measured 3 % faster here than the four-mad form of gl_field.hpp, which
// wins everywhere else - same box, tools/ab_kernel_times.sh, profiles/r04_ab_kernel_times.txt.)
// 3 waves/SIMD (<= 168 VGPRs, no spills): the tile's 16 coefficients per thread stay in registers across the coset
// loop, so the coefficients cross HBM once (re-reading them per coset measured the same time but 7x the fetch bytes).
__global__ __launch_bounds__(THREADS, 3) void k_gl_lde_pa16x2(const u64* __restrict__ coeffs, u64* __restrict__ lde, u32 L, u32 rate_bits,
                                                           const u64* __restrict__ tw4096, const u64* __restrict__ tw_hi,
                                                           const u64* __restrict__ tw_lo, const u64* __restrict__ pow_lo,
                                                           const u64* __restrict__ pow_hi) {
    __shared__ u64 sh[16 * 272];
    __shared__ u64 tw256[256];  // w_256^m: the stage-1 twiddles, read from LDS at their use (keeps 30 VGPRs free)
    const size_t col = blockIdx.x >> 8;
    const u32 tg = blockIdx.x & 255;
    const u32 tid = threadIdx.x, hi4 = tid >> 4, j = tid & 15;
    const u32 l = (tg << 4) + j;
    const size_t n = (size_t)1 << L;
    const u64* cin = coeffs + col * n + l;
    tw256[tid] = tw4096[tid * 16];
    u64 orig[16];
#pragma unroll
    for (u32 a1 = 0; a1 < 16; a1++) orig[a1] = cin[(size_t)(a1 * 16 + hi4) << 12];  // stage-1 thread = (a0 = hi4, j)
    __syncthreads();  // tw256 visible
    const u64 ratio = tw_split16(tw_hi, tw_lo, 16 * l);
    const u64 f0 = tw_split16(tw_hi, tw_lo, brev4(hi4) * l);  // w_n^(k_a1 l)
    for (u32 c = 0; c < (1u << rate_bits); c++) {
        const u64* ph = pow_hi + (size_t)c * 256 + hi4;
        u64 x[16];
#pragma unroll
        for (u32 a1 = 0; a1 < 16; a1++) x[a1] = ph[a1 * 16];  // s_c^(4096 a): 16 independent loads (entry 0 is 1)
        const u64 sl = pow_lo[(size_t)c * 4096 + l];
#pragma unroll
        for (u32 a1 = 0; a1 < 16; a1++) x[a1] = gl::mul_mont<true>(orig[a1], x[a1]);
        dft16<false>(x);
#pragma unroll
        for (u32 s = 0; s < 16; s++) sh[s * 272 + tid] = s ? gl::mul_mont<true>(x[s], tw256[(brev4(s) * hi4) & 255]) : x[s];  // w_256^(k_a1 a0); [k_a1 slot][a0][j]
        __syncthreads();
        // stage 2 thread = (k_a1 slot = hi4, j): digit a0
#pragma unroll
        for (u32 a0 = 0; a0 < 16; a0++) x[a0] = sh[hi4 * 272 + a0 * 16 + j];
        dft16<false>(x);
        u64* out = lde + (col << (L + rate_bits)) + (size_t)c * n + l;
        // output twiddle s^l w_n^(k_a l), k_a = k_a1 + 16 k': a geometric progression in k' with ratio w_n^(16 l)
        u64 f = gl::mul_mont_lazy<true>(sl, f0);
#pragma unroll
        for (u32 k = 0; k < 16; k++) {  // k' = k: slot brev4(k), row position = brev8(k_a)
            out[(size_t)(hi4 * 16 + brev4(k)) << 12] = gl::mul_mont<true>(x[brev4(k)], f);
            if (k < 15) f = gl::mul_mont_lazy<true>(f, ratio);
        }
        __syncthreads();
    }
}

// LA = 8 + K, K = 1, 2 (2^21 and 2^22 rows, round 6; round 5 ran a de-interleave pass and a combine pass around the 2^20-row kernels):
// the (256 R)-point strided DFT over a' as TWO stages with 32 points per thread - 32 x 16 at 2^21 rows (256 threads: radix 32 over a1,
// then two radix-16 DFTs over a0 per thread), 32 x 32 at 2^22 rows (512 threads).  Every root of unity of order <= 64 is a shift, so
// the radix-32 DFT costs one butterfly layer more and NO general multiplication more: the multiplications per output are those of
// the 2^20-row pass (scale, inter-stage twiddle, output twiddle and its chain).  The price is registers - 32 coefficients + 32
// working values per thread = two waves per SIMD - and a tile of 64 / 128 KB of LDS: 1.20x / 1.39x the 2^20-row pass per element
// (1.24x / 1.47x before the coset factors s_c^(4096 a') went through LDS one coset ahead: the sixteen threads of a row share them).
// (The alternative measured on the same box, profiles/r06_large_sizes.txt: 16 points per thread, four waves per SIMD, a radix-R stage
// over the top digit as a butterfly ACROSS LANES - v_permlane16/32_swap - and one general twiddle more per output: 1.27x / 1.58x,
// its instruction count.  BabyBear's 2^22-row pass is that form, kernels_bb16.hip: its words are half as wide.)
template <int K>
__global__ __launch_bounds__(128 << K, 2) void k_gl_lde_pa32(const u64* __restrict__ coeffs, u64* __restrict__ lde, u32 rate_bits,
                                                             const u64* __restrict__ tw4096, const u64* __restrict__ tw_hi,
                                                             const u64* __restrict__ tw_lo, const u64* __restrict__ pow_lo,
                                                             const u64* __restrict__ pow_hi) {
    constexpr u32 L = 20 + K, A0 = 8u << K /* values of a0: 16 / 32 */, NT = 128u << K, SLOT = NT + 16, ROWS = 256u << K;
    __shared__ u64 sh[32 * SLOT];      // [k_a1 slot][a0][j], slots padded by 16 words
    __shared__ u64 twl[ROWS];          // w_ROWS^m: the inter-stage twiddles
    __shared__ u64 phs[2][ROWS];       // s_c^(4096 a') of this coset and of the next: the 16 threads of a row share them, so the
                                       // workgroup loads each once (two per thread, one coset ahead) instead of 32 per thread
    const size_t col = blockIdx.x >> 8;
    const u32 tg = blockIdx.x & 255;
    const u32 tid = threadIdx.x, hi = tid >> 4, j = tid & 15;   // stage 1: a0 = hi; stage 2: slot(s) from hi
    const u32 l = (tg << 4) + j;
    const size_t n = (size_t)1 << L;
    const u64* cin = coeffs + col * n + l;
    for (u32 i = tid; i < ROWS; i += NT) {
        twl[i] = tw4096[i << (4 - K)];
        phs[0][i] = pow_hi[i];
    }
    u64 orig[32];
#pragma unroll
    for (u32 a1 = 0; a1 < 32; a1++) orig[a1] = cin[(size_t)(a1 * A0 + hi) << 12];
    __syncthreads();  // twl, phs[0] visible
    const u64 ratio = tw_split16(tw_hi, tw_lo, 32 * l);   // w_n^(32 l)
    const u32 ncosets = 1u << rate_bits;
    u64 sl_next = pow_lo[l];
    for (u32 c = 0; c < ncosets; c++) {
        const u64 sl = sl_next;
        u64 nph[ROWS / NT];                                  // the next coset's factors, in flight across this coset's work
        if (c + 1 < ncosets) {
#pragma unroll
            for (u32 k = 0; k < ROWS / NT; k++) nph[k] = pow_hi[(size_t)(c + 1) * ROWS + tid + k * NT];
            sl_next = pow_lo[(size_t)(c + 1) * 4096 + l];
        }
        const u64* ph = phs[c & 1] + hi;
        u64 x[32];
#pragma unroll
        for (u32 a1 = 0; a1 < 32; a1++) x[a1] = ph[a1 * A0];  // s_c^(4096 a'), a' = a1 A0 + a0
#pragma unroll
        for (u32 a1 = 0; a1 < 32; a1++) x[a1] = gl::mul_mont<true>(orig[a1], x[a1]);
        dft32<false>(x);
#pragma unroll
        for (u32 s = 0; s < 32; s++) sh[s * SLOT + tid] = s ? gl::mul_mont<true>(x[s], twl[(brevk(s, 5) * hi) & (ROWS - 1)]) : x[s];  // w_ROWS^(k_a1 a0)
        __syncthreads();
        u64* out = lde + (col << (L + rate_bits)) + (size_t)c * n + l;
        if constexpr (K == 2) {   // stage 2 thread = (k_a1 slot = hi, j): radix 32 over a0
#pragma unroll
            for (u32 a0 = 0; a0 < 32; a0++) x[a0] = sh[hi * SLOT + a0 * 16 + j];
            dft32<false>(x);
            u64 f = gl::mul_mont_lazy<true>(sl, tw_split16(tw_hi, tw_lo, brevk(hi, 5) * l));   // s^l w_n^(k_a1 l)
#pragma unroll
            for (u32 k = 0; k < 32; k++) {  // k_a' = k_a1 + 32 k: row position brev10(k_a') = hi * 32 + brev5(k)
                out[(size_t)(hi * 32 + brevk(k, 5)) << 12] = gl::mul_mont<true>(x[brevk(k, 5)], f);
                if (k < 31) f = gl::mul_mont_lazy<true>(f, ratio);
            }
        } else {                  // stage 2 thread = (k_a1 slots hi and hi + 16, j): two radix-16 DFTs over a0
#pragma unroll
            for (u32 u = 0; u < 2; u++) {
                const u32 slot = hi + 16 * u;
                u64 y[16];
#pragma unroll
                for (u32 a0 = 0; a0 < 16; a0++) y[a0] = sh[slot * SLOT + a0 * 16 + j];
                dft16<false>(y);
                u64 f = gl::mul_mont_lazy<true>(sl, tw_split16(tw_hi, tw_lo, brevk(slot, 5) * l));
#pragma unroll
                for (u32 k = 0; k < 16; k++) {  // row position brev9(k_a1 + 32 k) = slot * 16 + brev4(k)
                    out[(size_t)(slot * 16 + brev4(k)) << 12] = gl::mul_mont<true>(y[brev4(k)], f);
                    if (k < 15) f = gl::mul_mont_lazy<true>(f, ratio);
                }
            }
        }
        if (c + 1 < ncosets) {
#pragma unroll
            for (u32 k = 0; k < ROWS / NT; k++) phs[(c + 1) & 1][tid + k * NT] = nph[k];
        }
        __syncthreads();
    }
}

// LA = 4 + K (L = 16 + K, K = 1..3): a = a1 2^K + a0.  grid = ncols * 2^(4+K); a block holds all 2^LA rows of 2^(8-K) consecutive
// columns l: stage 1 (thread = (a0, jl), radix 16 over a1), LDS, stage 2 (radix 2^K over a0; a thread takes 2^(4-K) columns so
// that it still moves 16 values).  Same conventions as k_gl_lde_pa16x2: coefficients read once, kept across the coset loop.
template <int K>
__global__ __launch_bounds__(THREADS, 3) void k_gl_lde_pa16xs(const u64* __restrict__ coeffs, u64* __restrict__ lde, u32 rate_bits,
                                                              const u64* __restrict__ tw4096, const u64* __restrict__ tw_hi,
                                                              const u64* __restrict__ tw_lo, const u64* __restrict__ pow_lo,
                                                              const u64* __restrict__ pow_hi) {
    constexpr u32 L = 16 + K, LA = 4 + K, R = 1u << K, M = 256u >> K;  // M columns per block
    __shared__ u64 sh[16 * (256 + 16)];                                  // [k_a1 slot][a0][jl], slots padded by 16 words
    const u32 blocks_per_col = 4096 / M;
    const size_t col = blockIdx.x / blocks_per_col;
    const u32 l0 = (blockIdx.x % blocks_per_col) * M;
    const u32 tid = threadIdx.x, a0 = tid / M, jl = tid % M;
    const size_t n = (size_t)1 << L;
    u64 orig[16];
    {
        const u64* cin = coeffs + col * n + l0 + jl;
#pragma unroll
        for (u32 a1 = 0; a1 < 16; a1++) orig[a1] = cin[(size_t)((a1 << K) + a0) << 12];
    }
    // stage-2 role: slot s2 = tid >> 4, columns q + 16 u (u < M / 16)
    const u32 s2 = tid >> 4, q = tid & 15;
    const u32 ka1 = brev4(s2);
    const u32 ncosets = 1u << rate_bits;
    for (u32 c = 0; c < ncosets; c++) {
        const u64* ph = pow_hi + ((size_t)c << LA) + a0;
        u64 x[16];
#pragma unroll
        for (u32 a1 = 0; a1 < 16; a1++) x[a1] = ph[a1 << K];  // s_c^(4096 a)
#pragma unroll
        for (u32 a1 = 0; a1 < 16; a1++) x[a1] = gl::mul_mont(orig[a1], x[a1]);
        dft16<false>(x);
#pragma unroll
        for (u32 s = 0; s < 16; s++)  // inter-stage twiddle w_{2^LA}^(k_a1 a0)
            sh[s * 272 + a0 * M + jl] = (s && a0) ? gl::mul_mont(x[s], tw4096[((brev4(s) * a0) << (12 - LA)) & 4095]) : x[s];
        __syncthreads();
#pragma unroll
        for (u32 u = 0; u < M / 16; u++) {
            const u32 jj = q + 16 * u, l = l0 + jj;
            u64 y[16];
#pragma unroll
            for (u32 b = 0; b < R; b++) y[b] = sh[s2 * 272 + b * M + jj];
            dft_small<false, K>(y);
            // output twiddle s^l w_n^(k_a l), k_a = k_a1 + 16 k': a geometric progression in k' with ratio w_n^(16 l)
            u64 f = gl::mul_mont_lazy(pow_lo[(size_t)c * 4096 + l], tw_split16(tw_hi, tw_lo, ka1 * l));
            const u64 ratio = tw_split16(tw_hi, tw_lo, 16 * l);
            u64* out = lde + (col << (L + rate_bits)) + (size_t)c * n + l;
#pragma unroll
            for (u32 k = 0; k < R; k++) {  // row position = brev_LA(k_a) = slot * 2^K + brevK(k')
                out[(size_t)(s2 * R + brevk(k, K)) << 12] = gl::mul_mont(y[brevk(k, K)], f);
                if (k + 1 < R) f = gl::mul_mont_lazy(f, ratio);
            }
        }
        __syncthreads();
    }
}

// LA = K in 1..3 (L = 12 + K): one radix-2^K stage over the 2^K rows, a thread per column l, no LDS.  grid = ncols * 16.
template <int K>
__global__ __launch_bounds__(THREADS) void k_gl_lde_pa_small(const u64* __restrict__ coeffs, u64* __restrict__ lde, u32 rate_bits,
                                                             const u64* __restrict__ tw_hi, const u64* __restrict__ tw_lo,
                                                             const u64* __restrict__ pow_lo, const u64* __restrict__ pow_hi) {
    constexpr u32 L = 12 + K, R = 1u << K;
    const size_t col = blockIdx.x >> 4;
    const u32 l = ((blockIdx.x & 15) << 8) + threadIdx.x;
    const size_t n = (size_t)1 << L;
    const u64* cin = coeffs + col * n + l;
    u64 orig[R];
#pragma unroll
    for (u32 a = 0; a < R; a++) orig[a] = cin[(size_t)a << 12];
    const u32 ncosets = 1u << rate_bits;
    const u64 ratio = tw_split16(tw_hi, tw_lo, l);  // w_n^l
    for (u32 c = 0; c < ncosets; c++) {
        const u64* ph = pow_hi + ((size_t)c << K);
        u64 x[16];
#pragma unroll
        for (u32 a = 0; a < R; a++) x[a] = a ? gl::mul_mont(orig[a], ph[a]) : orig[a];  // s_c^(4096 a)
        dft_small<false, K>(x);
        u64* out = lde + (col << (L + rate_bits)) + (size_t)c * n + l;
        u64 f = pow_lo[(size_t)c * 4096 + l];  // s^l w_n^(k l), k = 0..R-1
#pragma unroll
        for (u32 k = 0; k < R; k++) {
            out[(size_t)brevk(k, K) << 12] = gl::mul_mont(x[brevk(k, K)], f);
            if (k + 1 < R) f = gl::mul_mont_lazy(f, ratio);
        }
    }
}

// LA = 4 (L = 16): grid = ncols * 16, block = 16 rows x 256 contiguous columns, one radix-16 stage, no LDS.
__global__ __launch_bounds__(THREADS) void k_gl_lde_pa16x1(const u64* __restrict__ coeffs, u64* __restrict__ lde, u32 rate_bits,
                                                           const u64* __restrict__ tw_hi, const u64* __restrict__ tw_lo,
                                                           const u64* __restrict__ pow_lo, const u64* __restrict__ pow_hi) {
    constexpr u32 L = 16;
    const size_t col = blockIdx.x >> 4;
    const u32 l = ((blockIdx.x & 15) << 8) + threadIdx.x;
    const size_t n = (size_t)1 << L;
    const u64* cin = coeffs + col * n + l;
    u64 orig[16];
#pragma unroll
    for (u32 a = 0; a < 16; a++) orig[a] = cin[(size_t)a << 12];
    const u32 ncosets = 1u << rate_bits;
    const u64 ratio = tw_split16(tw_hi, tw_lo, l);
    for (u32 c = 0; c < ncosets; c++) {
        const u64* ph = pow_hi + (size_t)c * 16;
        u64 x[16];
#pragma unroll
        for (u32 a = 0; a < 16; a++) x[a] = a ? gl::mul_mont(orig[a], ph[a]) : orig[a];
        dft16<false>(x);
        u64* out = lde + (col << (L + rate_bits)) + (size_t)c * n + l;
        u64 f = pow_lo[(size_t)c * 4096 + l];  // s^l w_n^(k l), k = 0..15
#pragma unroll
        for (u32 k = 0; k < 16; k++) {
            out[(size_t)brev4(k) << 12] = gl::mul_mont(x[brev4(k)], f);
            if (k < 15) f = gl::mul_mont_lazy(f, ratio);
        }
    }
}

// ------------------------------------------------------------------ inverse NTT passes (LA = 8, LB in {0, 4}, LC = 8)
struct Inv16Geom {
    u32 L, LB;  // LA = LC = 8, LL = LB + 8
};

// P1: grid = ncols * 2^(LL-4); tile 256 rows (a) x 16 contiguous; rows written in natural k_a order
__global__ __launch_bounds__(THREADS) void k_gl_intt16_p1(const u64* __restrict__ src, u64* __restrict__ dst, Inv16Geom g,
                                                          const u64* __restrict__ tw4096, const u64* __restrict__ tw_hi,
                                                          const u64* __restrict__ tw_lo) {
    __shared__ u64 sh[16 * 272];
    const u32 LL = g.LB + 8;
    const u32 tiles_per_col = 1u << (LL - 4);
    const size_t col = blockIdx.x / tiles_per_col;
    const u32 tg = blockIdx.x % tiles_per_col;
    const size_t base = (col << g.L) + ((size_t)tg << 4);
    const u32 tid = threadIdx.x, hi4 = tid >> 4, j = tid & 15;
    u64 x[16];
#pragma unroll
    for (u32 a1 = 0; a1 < 16; a1++) x[a1] = src[base + ((size_t)(a1 * 16 + hi4) << LL) + j];
    u64 tw[16];
    load_tw16(tw, tw4096, hi4 * 16);
    const u32 l = (tg << 4) + j;
    const u32 ka1 = brev4(hi4);
    // output twiddle w_n^-(k_a l), k_a = k_a1 + 16 k: a geometric progression in k with ratio w_n^-(16 l)
    u64 f = tw_split16(tw_hi, tw_lo, ka1 * l);
    const u64 ratio = tw_split16(tw_hi, tw_lo, 16 * l);
    dft16<true>(x);
#pragma unroll
    for (u32 s = 0; s < 16; s++) sh[s * 272 + tid] = s ? gl::mul_mont(x[s], tw[s]) : x[s];
    __syncthreads();
#pragma unroll
    for (u32 a0 = 0; a0 < 16; a0++) x[a0] = sh[hi4 * 272 + a0 * 16 + j];
    dft16<true>(x);
#pragma unroll
    for (u32 k = 0; k < 16; k++) {
        dst[base + ((size_t)(ka1 + 16 * k) << LL) + j] = gl::mul_mont(x[brev4(k)], f);
        if (k < 15) f = gl::mul_mont_lazy(f, ratio);
    }
}

// P2 (LB = 4): grid = ncols * 16 * 16; tile 16 k_a x 16 b x 16 c; src [k_a][b][c] -> dst [k_b][k_a][c]
__global__ __launch_bounds__(THREADS) void k_gl_intt16_p2(const u64* __restrict__ src, u64* __restrict__ dst, u32 L,
                                                          const u64* __restrict__ tw4096) {
    const size_t col = blockIdx.x >> 8;
    const u32 ga = (blockIdx.x >> 4) & 15, gc = blockIdx.x & 15;
    const size_t cbase = col << L;
    const u32 ia = threadIdx.x >> 4, jc = threadIdx.x & 15;
    const u32 ka = 16 * ga + ia, c = 16 * gc + jc;
    u64 x[16];
#pragma unroll
    for (u32 b = 0; b < 16; b++) x[b] = src[cbase + ((size_t)ka << 12) + ((size_t)b << 8) + c];
    u64 tw[16];
    load_tw16(tw, tw4096, c);  // w_4096^-(c k_b)
    dft16<true>(x);
#pragma unroll
    for (u32 s = 0; s < 16; s++)
        dst[cbase + ((size_t)brev4(s) << 16) + ((size_t)ka << 8) + c] = s ? gl::mul_mont(x[s], tw[s]) : x[s];
}

// P2 for LB = K in 1..3 (L = 16 + K): the same pass with a radix-2^K DFT over b; src [k_a][b][c] -> dst [k_b][k_a][c]
template <int K>
__global__ __launch_bounds__(THREADS) void k_gl_intt16_p2s(const u64* __restrict__ src, u64* __restrict__ dst, const u64* __restrict__ tw4096) {
    constexpr u32 L = 16 + K, R = 1u << K;
    const size_t col = blockIdx.x >> 8;
    const u32 ga = (blockIdx.x >> 4) & 15, gc = blockIdx.x & 15;
    const size_t cbase = col << L;
    const u32 ka = 16 * ga + (threadIdx.x >> 4), c = 16 * gc + (threadIdx.x & 15);
    u64 x[16];
#pragma unroll
    for (u32 b = 0; b < R; b++) x[b] = src[cbase + ((size_t)ka << (8 + K)) + ((size_t)b << 8) + c];
    u64 tw[R];
#pragma unroll
    for (u32 s = 1; s < R; s++) tw[s] = tw4096[(brevk(s, K) * c) << (4 - K)];  // w_{2^(8+K)}^-(c k_b)
    dft_small<true, K>(x);
#pragma unroll
    for (u32 s = 0; s < R; s++)
        dst[cbase + ((size_t)brevk(s, K) << 16) + ((size_t)ka << 8) + c] = s ? gl::mul_mont(x[s], tw[s]) : x[s];
}

// P2 for LB = 5, 6 (2^21 and 2^22 rows, round 6): the middle pass as a radix-32 / radix-64 DFT over b in registers - every root of
// unity of order <= 64 is a power of two, so the whole DFT is shifts; one or two DIF layers bring it down to 16-point blocks.
// src [k_a][b][c] -> dst [k_b][k_a][c]; tw = w_{2^14}^-j (Montgomery form), j < 2^14.  X[k_b] ends up in slot brev_LB(k_b).
template <int LB>
__global__ __launch_bounds__(THREADS) void k_gl_intt16_p2w(const u64* __restrict__ src, u64* __restrict__ dst, const u64* __restrict__ tw16k) {
    constexpr u32 L = 16 + LB, R = 1u << LB;
    const size_t col = blockIdx.x >> 8;
    const u32 ga = (blockIdx.x >> 4) & 15, gc = blockIdx.x & 15;
    const size_t cbase = col << L;
    const u32 ka = 16 * ga + (threadIdx.x >> 4), c = 16 * gc + (threadIdx.x & 15);
    u64 x[R];
#pragma unroll
    for (u32 b = 0; b < R; b++) x[b] = src[cbase + ((size_t)ka << (8 + LB)) + ((size_t)b << 8) + c];
    if constexpr (LB == 6) {
        layer64<true, 32, 0>(x);
        layer64<true, 16, 0>(x);
        layer64<true, 16, 32>(x);
    } else {
        layer64<true, 16, 0>(x);
    }
#pragma unroll
    for (u32 o = 0; o < R; o += 16) dft16<true>(*reinterpret_cast<u64(*)[16]>(&x[o]));
#pragma unroll
    for (u32 s = 0; s < R; s++) {
        const u32 kb = brevk(s, LB);
        dst[cbase + ((size_t)kb << 16) + ((size_t)ka << 8) + c] = s ? gl::mul_mont(x[s], tw16k[(kb * c) << (6 - LB)]) : x[s];   // w_{2^(8+LB)}^-(c k_b)
    }
}

// P3: grid = ncols * 2^LB * 16; tile 16 k_a x 256 c (c = 16 c1 + c0); src [k_b][k_a][c];
// dst natural k = k_a + 256 k_b + 2^(8+LB) k_c, scaled by n^-1
__global__ __launch_bounds__(THREADS) void k_gl_intt16_p3(const u64* __restrict__ src, u64* __restrict__ dst, Inv16Geom g,
                                                          const u64* __restrict__ tw4096, u64 n_inv) {
    __shared__ u64 sh[16 * 272];
    const u32 nb = 1u << g.LB;
    const size_t col = blockIdx.x / (nb * 16);
    const u32 rem = blockIdx.x % (nb * 16);
    const u32 kb = rem >> 4, ga = rem & 15;
    const size_t cbase = col << g.L;
    const size_t sbase = cbase + ((size_t)kb << 16) + ((size_t)(16 * ga) << 8);
    const u32 tid = threadIdx.x, hi4 = tid >> 4, lo4 = tid & 15;
    u64 x[16];
    // stage 1 thread = (ia = hi4, c0 = lo4): digit c1
#pragma unroll
    for (u32 c1 = 0; c1 < 16; c1++) x[c1] = src[sbase + hi4 * 256 + c1 * 16 + lo4];
    u64 tw[16];
    load_tw16(tw, tw4096, lo4 * 16);  // w_256^-(k_c1 c0)
    dft16<true>(x);
#pragma unroll
    for (u32 s = 0; s < 16; s++) sh[s * 272 + lo4 * 17 + hi4] = s ? gl::mul_mont(x[s], tw[s]) : x[s];  // [k_c1 slot][c0][ia], rows padded to 17
    __syncthreads();
    // stage 2 thread = (k_c1 slot = hi4, ia = lo4): digit c0
#pragma unroll
    for (u32 c0 = 0; c0 < 16; c0++) x[c0] = sh[hi4 * 272 + c0 * 17 + lo4];
    dft16<true>(x);
    const u32 kc1 = brev4(hi4);
#pragma unroll
    for (u32 s = 0; s < 16; s++) {
        const u32 kc = kc1 + 16 * brev4(s);
        dst[cbase + ((size_t)kc << (8 + g.LB)) + ((size_t)kb << 8) + 16 * ga + lo4] = gl::mul_mont(x[s], n_inv);
    }
}

// ------------------------------------------------------------------ launchers (called from kernels_ntt.hip's dispatchers)
// Every kernel of this file takes the MONTGOMERY-form copies of the tables (GlNttTables::*_m, GlCosetTables::*_m): all of their
// general multiplications have a table value as one factor (gl::mul_mont).

bool gl_intt_columns_r16(const u64* src, u64* coeffs, u64* scratch, size_t ncols, const GlNttTables& t, hipStream_t stream) {
    const u32 L = t.log_n;
    if (L < 16 || L > 22) return false;
    Inv16Geom g{L, L - 16};
    const u32 LL = g.LB + 8;
    u64* p1_dst = g.LB ? coeffs : scratch;
    hipLaunchKernelGGL(k_gl_intt16_p1, dim3((u32)(ncols << (LL - 4))), dim3(THREADS), 0, stream, src, p1_dst, g, t.tw4096_inv_m,
                       t.tw_hi_inv_m, t.tw_lo_inv_m);
    const dim3 g2((u32)(ncols << 8));
    if (g.LB == 6) hipLaunchKernelGGL(k_gl_intt16_p2w<6>, g2, dim3(THREADS), 0, stream, coeffs, scratch, t.tw16k_inv_m);
    else if (g.LB == 5) hipLaunchKernelGGL(k_gl_intt16_p2w<5>, g2, dim3(THREADS), 0, stream, coeffs, scratch, t.tw16k_inv_m);
    else if (g.LB == 4) hipLaunchKernelGGL(k_gl_intt16_p2, g2, dim3(THREADS), 0, stream, coeffs, scratch, L, t.tw4096_inv_m);
    else if (g.LB == 3) hipLaunchKernelGGL(k_gl_intt16_p2s<3>, g2, dim3(THREADS), 0, stream, coeffs, scratch, t.tw4096_inv_m);
    else if (g.LB == 2) hipLaunchKernelGGL(k_gl_intt16_p2s<2>, g2, dim3(THREADS), 0, stream, coeffs, scratch, t.tw4096_inv_m);
    else if (g.LB == 1) hipLaunchKernelGGL(k_gl_intt16_p2s<1>, g2, dim3(THREADS), 0, stream, coeffs, scratch, t.tw4096_inv_m);
    hipLaunchKernelGGL(k_gl_intt16_p3, dim3((u32)(ncols << (g.LB + 4))), dim3(THREADS), 0, stream, scratch, coeffs, g,
                       t.tw4096_inv_m, t.n_inv_m);
    return true;
}

bool gl_lde_pa_r16(const u64* coeffs, u64* lde, size_t ncols, const GlNttTables& t, const GlCosetTables& ct, hipStream_t stream) {
    const u32 L = t.log_n;
    if (L == 20) {
        hipLaunchKernelGGL(k_gl_lde_pa16x2, dim3((u32)(ncols << 8)), dim3(THREADS), 0, stream, coeffs, lde, L, ct.rate_bits,
                           t.tw4096_fwd_m, t.tw_hi_fwd_m, t.tw_lo_fwd_m, ct.pow_lo_m, ct.pow_hi_m);
        return true;
    }
    if (L == 21 || L == 22) {
        if (L == 21)
            hipLaunchKernelGGL(k_gl_lde_pa32<1>, dim3((u32)(ncols << 8)), dim3(256), 0, stream, coeffs, lde, ct.rate_bits, t.tw4096_fwd_m,
                               t.tw_hi_fwd_m, t.tw_lo_fwd_m, ct.pow_lo_m, ct.pow_hi_m);
        else
            hipLaunchKernelGGL(k_gl_lde_pa32<2>, dim3((u32)(ncols << 8)), dim3(512), 0, stream, coeffs, lde, ct.rate_bits, t.tw4096_fwd_m,
                               t.tw_hi_fwd_m, t.tw_lo_fwd_m, ct.pow_lo_m, ct.pow_hi_m);
        return true;
    }
#define GB_PAS(KK)                                                                                                        \
    hipLaunchKernelGGL(k_gl_lde_pa16xs<KK>, dim3((u32)(ncols << (4 + KK))), dim3(THREADS), 0, stream, coeffs, lde, ct.rate_bits, \
                       t.tw4096_fwd_m, t.tw_hi_fwd_m, t.tw_lo_fwd_m, ct.pow_lo_m, ct.pow_hi_m)
    if (L >= 13 && L <= 15) {
        const dim3 grid((u32)(ncols << 4));
        if (L == 13) hipLaunchKernelGGL(k_gl_lde_pa_small<1>, grid, dim3(THREADS), 0, stream, coeffs, lde, ct.rate_bits, t.tw_hi_fwd_m, t.tw_lo_fwd_m, ct.pow_lo_m, ct.pow_hi_m);
        if (L == 14) hipLaunchKernelGGL(k_gl_lde_pa_small<2>, grid, dim3(THREADS), 0, stream, coeffs, lde, ct.rate_bits, t.tw_hi_fwd_m, t.tw_lo_fwd_m, ct.pow_lo_m, ct.pow_hi_m);
        if (L == 15) hipLaunchKernelGGL(k_gl_lde_pa_small<3>, grid, dim3(THREADS), 0, stream, coeffs, lde, ct.rate_bits, t.tw_hi_fwd_m, t.tw_lo_fwd_m, ct.pow_lo_m, ct.pow_hi_m);
        return true;
    }
    if (L == 17) { GB_PAS(1); return true; }
    if (L == 18) { GB_PAS(2); return true; }
    if (L == 19) { GB_PAS(3); return true; }
#undef GB_PAS
    if (L == 16) {
        hipLaunchKernelGGL(k_gl_lde_pa16x1, dim3((u32)(ncols << 4)), dim3(THREADS), 0, stream, coeffs, lde, ct.rate_bits,
                           t.tw_hi_fwd_m, t.tw_lo_fwd_m, ct.pow_lo_m, ct.pow_hi_m);
        return true;
    }
    return false;
}

void gl_lde_pb_r16(u64* lde, size_t ntiles, const GlNttTables& t, hipStream_t stream) {
    hipLaunchKernelGGL(k_gl_lde_pb16, dim3((u32)ntiles), dim3(THREADS), 0, stream, lde, t.tw4096_fwd_m);
}

}  // namespace gbk
