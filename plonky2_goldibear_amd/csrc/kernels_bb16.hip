// BabyBear NTT passes, v2: radix-16 butterflies held in registers (gfx950).
//
// Same passes, tiles, index math and memory layouts as kernels_ntt16.hip (Goldilocks); what differs is the
// arithmetic: BabyBear has no power-of-two roots of unity, so the 17 non-trivial twiddles inside a 16-point
// DFT are Montgomery multiplications by compile-time constants w_16^k.  The multiplication count per element
// and layer is therefore the same as radix-2 (1/2); what the register form buys is one LDS round trip per four
// layers instead of one per layer (v1, kernels_bb.hip: 34.8 ms of LDE per 2^20 proof, ~5x the VALU bound).
// kernels_bb.hip's v1 passes remain for the inverse transform of 2^13..2^15 rows and for single-tile sizes.
#include <algorithm>

#include "bb_field.hpp"
#include "kernels.hpp"

namespace gbk {

static constexpr int THREADS = 256;

__device__ __forceinline__ constexpr u32 brev4(u32 x) { return ((x & 1) << 3) | ((x & 2) << 1) | ((x & 4) >> 1) | ((x & 8) >> 3); }

namespace bbc {  // compile-time twiddles (canonical arithmetic, stored in Montgomery form)
constexpr u32 cmul(u32 a, u32 b) { return (u32)((u64)a * b % bb::P); }
constexpr u32 cpow(u32 b, u64 e) {
    u32 r = 1;
    while (e) {
        if (e & 1) r = cmul(r, b);
        b = cmul(b, b);
        e >>= 1;
    }
    return r;
}
constexpr u32 W16 = cpow(bb::TWO_ADIC_GEN_27, 1u << 23);  // primitive 16th root of unity (two_adic_generator(4))
constexpr u32 W16_INV = cpow(W16, 15);
constexpr u32 mont(u32 x) { return (u32)(((u64)x << 32) % bb::P); }
template <bool INV, int M>
constexpr u32 tw() { return mont(cpow(INV ? W16_INV : W16, M)); }
// the same twiddle as a signed word of magnitude <= p/2, for the signed Montgomery product (bb::mul_signed)
template <bool INV, int M>
constexpr int tw_centred() { return tw<INV, M>() > bb::P / 2 ? (int)tw<INV, M>() - (int)bb::P : (int)tw<INV, M>(); }
static_assert(cpow(W16, 16) == 1 && cpow(W16, 8) == bb::P - 1, "W16 must be a primitive 16th root of unity");
}  // namespace bbc

// (a - b) * w_16^(+-M)
template <bool INV, int M>
__device__ __forceinline__ u32 sub_twiddle(u32 a, u32 b) {
    if constexpr (M == 0) return bb::sub(a, b);
    else {   // a - b as a signed word in (-p, p) times a twiddle within +-p/2: the signed product lands within +-0.74 p and one
             // selection makes it canonical - six instructions where a - b + p and the unsigned product took seven
        const int r = bb::mul_signed((int)(a - b), bbc::tw_centred<INV, M>());
        const u32 e = (u32)r + bb::P;
        return e < (u32)r ? e : (u32)r;
    }
}

template <bool INV, int H, int J>
__device__ __forceinline__ void bfly(u32& a, u32& b) {  // DIF butterfly of half-size H at offset J: twiddle w_{2H}^J = w_16^(J * 8 / H)
    u32 s = bb::add(a, b);
    b = sub_twiddle<INV, J*(8 / H)>(a, b);
    a = s;
}

// 16-point DFT in registers, natural input order, output X[k] in slot brev4(k) (in-place DIF)
template <bool INV>
__device__ __forceinline__ void dft16(u32 (&x)[16]) {
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
// output X[k] in slot brevK(k) - for the sizes between the multiples of four bits (2^17..2^19 rows)
template <bool INV, int K>
__device__ __forceinline__ void dft_small(u32 (&x)[16]) {
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

namespace bbc {
constexpr u32 W64 = cpow(bb::TWO_ADIC_GEN_27, 1u << 21), W64_INV = cpow(W64, 63);
static_assert(cpow(W64, 4) == W16 && cpow(W64, 32) == bb::P - 1, "W64 must be the 64th root above W16");
template <bool INV, int M>
constexpr int tw64_centred() {
    const u32 t = mont(cpow(INV ? W64_INV : W64, M));
    return t > bb::P / 2 ? (int)t - (int)bb::P : (int)t;
}
}  // namespace bbc
template <bool INV, int H, int O, int J = 0>
__device__ __forceinline__ void layer64(u32* x) {   // one DIF layer of half-size H (32 or 16) on x[O .. O + 2H): twiddle w_64^(J * 32 / H)
    if constexpr (J < H) {
        const u32 a = x[O + J], b = x[O + J + H];
        x[O + J] = bb::add(a, b);
        if constexpr (J == 0) x[O + J + H] = bb::sub(a, b);
        else {
            const int r = bb::mul_signed((int)(a - b), bbc::tw64_centred<INV, J*(32 / H)>());
            const u32 e = (u32)r + bb::P;
            x[O + J + H] = e < (u32)r ? e : (u32)r;
        }
        layer64<INV, H, O, J + 1>(x);
    }
}
// 32-point DFT in registers, natural input order, output X[k] in slot brev5(k): one DIF layer, then two 16-point blocks
template <bool INV>
__device__ __forceinline__ void dft32(u32 (&x)[32]) {
    layer64<INV, 16, 0>(x);
    dft16<INV>(*reinterpret_cast<u32(*)[16]>(&x[0]));
    dft16<INV>(*reinterpret_cast<u32(*)[16]>(&x[16]));
}

__device__ __forceinline__ u32 bb_tw_split16(const u32* __restrict__ hi, const u32* __restrict__ lo, u32 e) {
    u32 eh = e >> 10, el = e & 1023;
    u32 w = lo[el];
    return eh ? bb::mul(w, hi[eh]) : w;
}

// The 15 inter-stage twiddles w_4096^(k m), k = brev4(slot), as ONE batch of independent loads issued before the
// DFT that precedes their use.  (Loaded one by one behind `if (e)` each of them cost a full memory round trip:
// rocprofv3 showed the waves of these kernels parked in s_waitcnt for 46-66 % of their lifetime.)
__device__ __forceinline__ void load_tw16(u32 (&tw)[16], const u32* __restrict__ tw4096, u32 m) {
#pragma unroll
    for (u32 s = 1; s < 16; s++) tw[s] = tw4096[brev4(s) * m];
}

// ------------------------------------------------------------------ LDE pass B: 4096 contiguous points, 3 radix-16 stages
// grid = number of 4096-tiles of `lde` (in place).  natural -> bit-reversed.
__global__ __launch_bounds__(THREADS) void k_bb_lde_pb16(u32* __restrict__ lde, const u32* __restrict__ tw4096, u32 ntiles) {
    __shared__ u32 sh[16 * 272];
    const u32 tid = threadIdx.x;
    const u32 hi4 = tid >> 4, lo4 = tid & 15;
    u32 tw1[16], tw2[16];
    load_tw16(tw1, tw4096, tid);       // w_4096^(k2 (16 d1 + d0))
    load_tw16(tw2, tw4096, lo4 * 16);  // w_256^(k1 d0)
    u32 x[16], nx[16];
    u32 tile = blockIdx.x;
    if (tile < ntiles) {
        const u32* p = lde + ((size_t)tile << 12);
#pragma unroll
        for (u32 d = 0; d < 16; d++) nx[d] = p[d * 256 + tid];
    }
    // persistent workgroups: the next tile's 16 loads are in flight while this tile is transformed
    for (; tile < ntiles; tile += gridDim.x) {
        u32* p = lde + ((size_t)tile << 12);
#pragma unroll
        for (u32 d = 0; d < 16; d++) x[d] = nx[d];
        const u32 next = tile + gridDim.x;
        if (next < ntiles) {
            const u32* q = lde + ((size_t)next << 12);
#pragma unroll
            for (u32 d = 0; d < 16; d++) nx[d] = q[d * 256 + tid];
        }
        // stage 1: digit d2 (stride 256); this thread is (d1, d0) = tid
        dft16<false>(x);
#pragma unroll
        for (u32 s = 0; s < 16; s++) sh[s * 272 + tid] = s ? bb::mul(x[s], tw1[s]) : x[s];
        __syncthreads();
        // stage 2: digit d1; this thread is (k2 slot, d0)
#pragma unroll
        for (u32 d = 0; d < 16; d++) x[d] = sh[hi4 * 272 + d * 16 + lo4];
        dft16<false>(x);
        __syncthreads();
#pragma unroll
        for (u32 s = 0; s < 16; s++) sh[hi4 * 272 + lo4 * 17 + s] = s ? bb::mul(x[s], tw2[s]) : x[s];  // [k2 slot][d0][k1 slot], rows padded to 17
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
        __syncthreads();
    }
}

// ------------------------------------------------------------------ LDE pass A
// LA = 8: tile 256 rows (a = 16 a1 + a0) x JW columns, JW = 32 so that a row segment is a full 128-byte line
// (16 x 4 B = 64 B segments left the stores of v2's first cut at 0.8 TB/s).  grid = ncols * 4096 / JW, 16 * JW threads.
// Per coset: scale by s^(4096 a), two radix-16 stages over a, twiddle w_n^(k_a l) s^l, in-place DIF row order.
template <u32 JW>
__global__ __launch_bounds__(16 * JW) void k_bb_lde_pa16x2(const u32* __restrict__ coeffs, u32* __restrict__ lde, u32 L, u32 rate_bits,
                                                            const u32* __restrict__ tw4096, const u32* __restrict__ tw_hi,
                                                            const u32* __restrict__ tw_lo, const u32* __restrict__ pow_lo,
                                                            const u32* __restrict__ pow_hi) {
    constexpr u32 ROW = 16 * JW;  // one k_a1 slot: [a0][j]; 4-byte words, consecutive j -> consecutive banks
    __shared__ u32 sh[16 * ROW];
    constexpr u32 TPC = 4096 / JW;  // tiles per column
    const size_t col = blockIdx.x / TPC;
    const u32 tg = blockIdx.x % TPC;
    const u32 tid = threadIdx.x, hi4 = tid / JW, j = tid % JW;
    const u32 l = tg * JW + j;
    const size_t n = (size_t)1 << L;
    const u32* cin = coeffs + col * n + l;
    u32 orig[16];
#pragma unroll
    for (u32 a1 = 0; a1 < 16; a1++) orig[a1] = cin[(size_t)(a1 * 16 + hi4) << 12];  // stage-1 thread = (a0 = hi4, j)
    const u32 ratio = bb_tw_split16(tw_hi, tw_lo, 16 * l);
    const u32 f0 = bb_tw_split16(tw_hi, tw_lo, brev4(hi4) * l);  // w_n^(k_a1 l)
    u32 tw[16];
    load_tw16(tw, tw4096, hi4 * 16);  // w_256^(k_a1 a0): the same for every coset
    for (u32 c = 0; c < (1u << rate_bits); c++) {
        const u32* ph = pow_hi + (size_t)c * 256 + hi4;
        u32 x[16];
#pragma unroll
        for (u32 a1 = 0; a1 < 16; a1++) x[a1] = ph[a1 * 16];  // s_c^(4096 a): 16 independent loads (entry 0 is 1)
        const u32 sl = pow_lo[(size_t)c * 4096 + l];
#pragma unroll
        for (u32 a1 = 0; a1 < 16; a1++) x[a1] = bb::mul(orig[a1], x[a1]);
        dft16<false>(x);
#pragma unroll
        for (u32 s = 0; s < 16; s++) sh[s * ROW + tid] = s ? bb::mul(x[s], tw[s]) : x[s];  // [k_a1 slot][a0][j]
        __syncthreads();
        // stage 2 thread = (k_a1 slot = hi4, j): digit a0
#pragma unroll
        for (u32 a0 = 0; a0 < 16; a0++) x[a0] = sh[hi4 * ROW + a0 * JW + j];
        dft16<false>(x);
        u32* out = lde + (col << (L + rate_bits)) + (size_t)c * n + l;
        // output twiddle s^l w_n^(k_a l), k_a = k_a1 + 16 k': a geometric progression in k' with ratio w_n^(16 l)
        u32 f = bb::mul(sl, f0);
#pragma unroll
        for (u32 k = 0; k < 16; k++) {  // k' = k: slot brev4(k), row position = brev8(k_a)
            out[(size_t)(hi4 * 16 + brev4(k)) << 12] = bb::mul(x[brev4(k)], f);
            if (k < 15) f = bb::mul(f, ratio);
        }
        __syncthreads();
    }
}

// LA = 10 (2^22 rows, round 6; K = 2 is what the library launches): the strided pass as a 1024-point DFT over a' = 256 i1 + 16 a1 + a0
// in THREE DIF stages - radix 4 over the top digit i1, twiddle w_1024^(k1 a), then the two radix-16 stages of k_bb_lde_pa16x2 once
// per k1.  Output digit k1 is the LEAST significant one of k_a' = k1 + 4 k_a, so the rows of (c, k1) land in block brev_2(k1) of coset
// block c: the 2^(r+2) blocks of 2^20 leaves ARE the cosets of a rate-2^(r+2) LDE of 2^20 rows, and everything after stage 0 is the
// 2^20-row pass on that finer coset c' = 4 c + brev_2(k1) - output twiddle (s_c w_n^k1)^l w_m^(k_a l) from the tables of m = 2^20 rows
// and of the rate r + 2.  One general multiplication per output more than at 2^20 rows, no extra pass over the data.
// 1024 threads: lanes 0..31 are the 32 columns of a 128-byte row segment, lane bit 5 is digit bit b of i1, the sixteen waves are a0.
// A thread holds i1 = b and b + 2 (32 words), does their butterfly itself and the second one ACROSS the wave's halves
// (v_permlane32_swap hands every lane both operands: no LDS), and goes through the radix-16 stages twice (k1 = 2 b and 2 b + 1).
// Measured against the two-stage form below (k_bb_lde_pa32<2>, same box, profiles/r06_large_sizes.txt): 13.0 against 14.8 ms per 167
// columns - the other three (field, size) pairs go the other way.
__device__ __forceinline__ void lane_pair32(u32 x, u32& lower, u32& upper) {
    const auto r = __builtin_amdgcn_permlane32_swap(x, x, false, false);   // upper half of the first operand <-> lower half of the second
    lower = r[0];   // (the builtin, not inline asm: the swap needs wait states after a vector write of its operands, which the compiler's
    upper = r[1];   //  hazard recognizer inserts for its own instructions only - the asm form returned stale halves on the GPU)
}
template <int K>
__global__ __launch_bounds__(1024) void k_bb_lde_pa16x2w(const u32* __restrict__ coeffs, u32* __restrict__ lde, u32 rate_bits,
                                                          const u32* __restrict__ tw4096, const u32* __restrict__ tw_hi,
                                                          const u32* __restrict__ tw_lo, const u32* __restrict__ pow_lo,
                                                          const u32* __restrict__ pow_hi) {
    constexpr u32 R = 1u << K, L = 20 + K, W = 64, SLOT = 16 * W;   // LDS: [k_a1 slot][a0][b][j]
    __shared__ u32 sh[16 * SLOT];
    __shared__ u32 tw0[R * 256];   // the stage-0 twiddles w_{256R}^(k a), [k][a]: in LDS so that they are not 32 more live registers
    const size_t col = blockIdx.x >> 7;
    const u32 tg = blockIdx.x & 127;
    const u32 tid = threadIdx.x, hi4 = tid >> 6, jl = tid & 63, b = jl >> 5, j = jl & 31;
    if (tid < R * 256) tw0[tid] = tw4096[((tid >> 8) * (tid & 255)) << (4 - K)];
    const u32 l = tg * 32 + j;
    const size_t n = (size_t)1 << L;
    const u32* cin = coeffs + col * n + l;
    constexpr u32 H = K == 1 ? 1 : 2;   // digits i1 a thread holds: b (+ 2)
    u32 orig[H][16];
#pragma unroll
    for (u32 h = 0; h < H; h++)
#pragma unroll
        for (u32 a1 = 0; a1 < 16; a1++) orig[h][a1] = cin[(size_t)((b + 2 * h) * 256 + a1 * 16 + hi4) << 12];
    const u32 ratio = bb_tw_split16(tw_hi, tw_lo, 16 * l);          // tables of m = 2^20 rows
    const u32 f0 = bb_tw_split16(tw_hi, tw_lo, brev4(hi4) * l);
    u32 tw[16];
    load_tw16(tw, tw4096, hi4 * 16);  // w_256^(k_a1 a0)
    // lane mask of the digit bit, opaque to the optimiser: written as selects on a lane-varying bool the butterflies below made the
    // compiler spill registers (it unswitched around them); as bit blends (v_bfi) they cost the same and spill nothing
    u32 m5 = b ? ~0u : 0u;
    asm("" : "+v"(m5));
    __syncthreads();                  // tw0 visible
    for (u32 c = 0; c < (1u << rate_bits); c++) {
        const u32* ph = pow_hi + (size_t)c * (256 * R) + hi4;
        u32 y[H][16];
#pragma unroll
        for (u32 h = 0; h < H; h++)
#pragma unroll
            for (u32 a1 = 0; a1 < 16; a1++) y[h][a1] = bb::mul(orig[h][a1], ph[(b + 2 * h) * 256 + a1 * 16]);  // s_c^(4096 a')
#pragma unroll
        for (u32 a1 = 0; a1 < 16; a1++) {
            u32 p, q;
            if constexpr (K == 2) {   // (x_b, x_(b+2)) in the thread: sum for the even k1, difference (times w_4 where b = 1) for the odd ones
                const u32 sm = bb::add(y[0][a1], y[1][a1]), d = bb::sub(y[0][a1], y[1][a1]);
                const u32 dw = bb::mul(d, bbc::tw<false, 4>());
                y[0][a1] = sm;
                y[1][a1] = (d & ~m5) | (dw & m5);
                lane_pair32(y[1][a1], p, q);
                const u32 s1 = bb::add(p, q), d1 = bb::sub(p, q);
                y[1][a1] = (s1 & ~m5) | (d1 & m5);          // k1 = 2 b + 1
            }
            lane_pair32(y[0][a1], p, q);
            const u32 s0 = bb::add(p, q), d0 = bb::sub(p, q);
            y[0][a1] = (s0 & ~m5) | (d0 & m5);              // k1 = b (K = 1) / 2 b (K = 2)
        }
#pragma nounroll
        for (u32 h = 0; h < H; h++) {   // (a real loop: unrolled, the two passes' addresses and twiddles were live together and spilled)
            const u32 k1 = K == 1 ? b : 2 * b + h;
            const u32 cf = c * R + (K == 1 ? b : b + 2 * h);   // brev_K(k1)
            u32 x[16];
#pragma unroll
            for (u32 a1 = 0; a1 < 16; a1++) x[a1] = bb::mul(y[0][a1], tw0[k1 * 256 + a1 * 16 + hi4]);   // w_{256R}^(k1 a) (k1 = 0: 1)
            if constexpr (K == 2) {
#pragma unroll
                for (u32 a1 = 0; a1 < 16; a1++) y[0][a1] = y[1][a1];
            }
            const u32 sl = pow_lo[(size_t)cf * 4096 + l];
            dft16<false>(x);
#pragma unroll
            for (u32 s = 0; s < 16; s++) sh[s * SLOT + tid] = s ? bb::mul(x[s], tw[s]) : x[s];  // [k_a1 slot][a0][b][j]
            __syncthreads();
#pragma unroll
            for (u32 a0 = 0; a0 < 16; a0++) x[a0] = sh[hi4 * SLOT + a0 * W + jl];
            dft16<false>(x);
            u32* out = lde + (col << (L + rate_bits)) + ((size_t)cf << 20) + l;
            u32 f = bb::mul(sl, f0);
#pragma unroll
            for (u32 k = 0; k < 16; k++) {
                out[(size_t)(hi4 * 16 + brev4(k)) << 12] = bb::mul(x[brev4(k)], f);
                if (k < 15) f = bb::mul(f, ratio);
            }
            __syncthreads();
        }
    }
}

// LA = 8 + K (2^21 rows: K = 1 is what the library launches; see k_gl_lde_pa32): two stages with 32 points per thread - 32 x 16 at
// 2^21 rows (512 threads: 16 a0 x 32 columns), 32 x 32 at 2^22 rows (1024 threads); the multiplications per output are those of the
// 2^20-row pass plus the fifteen constant twiddles of the extra butterfly layer per 32 points.
template <int K>
__global__ __launch_bounds__(256 << K, 4) void k_bb_lde_pa32(const u32* __restrict__ coeffs, u32* __restrict__ lde, u32 rate_bits,
                                                          const u32* __restrict__ tw4096, const u32* __restrict__ tw_hi,
                                                          const u32* __restrict__ tw_lo, const u32* __restrict__ pow_lo,
                                                          const u32* __restrict__ pow_hi) {
    constexpr u32 L = 20 + K, A0 = 8u << K, NT = 256u << K, SLOT = NT, ROWS = 256u << K, JW = 32;
    __shared__ u32 sh[32 * SLOT];      // [k_a1 slot][a0][j]
    __shared__ u32 twl[ROWS];          // w_ROWS^m: the inter-stage twiddles
    __shared__ u32 phs[2][ROWS];       // s_c^(4096 a') of this coset and of the next (see k_gl_lde_pa32)
    const size_t col = blockIdx.x >> 7;
    const u32 tg = blockIdx.x & 127;
    const u32 tid = threadIdx.x, hi = tid >> 5, j = tid & 31;
    const u32 l = tg * JW + j;
    const size_t n = (size_t)1 << L;
    const u32* cin = coeffs + col * n + l;
    for (u32 i = tid; i < ROWS; i += NT) {
        twl[i] = tw4096[i << (4 - K)];
        phs[0][i] = pow_hi[i];
    }
    u32 orig[32];
#pragma unroll
    for (u32 a1 = 0; a1 < 32; a1++) orig[a1] = cin[(size_t)(a1 * A0 + hi) << 12];
    __syncthreads();
    const u32 ratio = bb_tw_split16(tw_hi, tw_lo, 32 * l);   // w_n^(32 l)
    const u32 ncosets = 1u << rate_bits;
    u32 sl_next = pow_lo[l];
    for (u32 c = 0; c < ncosets; c++) {
        const u32 sl = sl_next;
        u32 nph = 0;                                             // ROWS == NT: one factor of the next coset per thread, in flight
        if (c + 1 < ncosets) {
            nph = pow_hi[(size_t)(c + 1) * ROWS + tid];
            sl_next = pow_lo[(size_t)(c + 1) * 4096 + l];
        }
        const u32* ph = phs[c & 1] + hi;
        u32 x[32];
#pragma unroll
        for (u32 a1 = 0; a1 < 32; a1++) x[a1] = bb::mul(orig[a1], ph[a1 * A0]);  // s_c^(4096 a'), a' = a1 A0 + a0
        dft32<false>(x);
#pragma unroll
        for (u32 s = 0; s < 32; s++) sh[s * SLOT + tid] = s ? bb::mul(x[s], twl[(brevk(s, 5) * hi) & (ROWS - 1)]) : x[s];
        __syncthreads();
        u32* out = lde + (col << (L + rate_bits)) + (size_t)c * n + l;
        if constexpr (K == 2) {
#pragma unroll
            for (u32 a0 = 0; a0 < 32; a0++) x[a0] = sh[hi * SLOT + a0 * JW + j];
            dft32<false>(x);
            u32 f = bb::mul(sl, bb_tw_split16(tw_hi, tw_lo, brevk(hi, 5) * l));
#pragma unroll
            for (u32 k = 0; k < 32; k++) {
                out[(size_t)(hi * 32 + brevk(k, 5)) << 12] = bb::mul(x[brevk(k, 5)], f);
                if (k < 31) f = bb::mul(f, ratio);
            }
        } else {
#pragma unroll
            for (u32 u = 0; u < 2; u++) {
                const u32 slot = hi + 16 * u;
                u32 y[16];
#pragma unroll
                for (u32 a0 = 0; a0 < 16; a0++) y[a0] = sh[slot * SLOT + a0 * JW + j];
                dft16<false>(y);
                u32 f = bb::mul(sl, bb_tw_split16(tw_hi, tw_lo, brevk(slot, 5) * l));
#pragma unroll
                for (u32 k = 0; k < 16; k++) {
                    out[(size_t)(slot * 16 + brev4(k)) << 12] = bb::mul(y[brev4(k)], f);
                    if (k < 15) f = bb::mul(f, ratio);
                }
            }
        }
        if (c + 1 < ncosets) phs[(c + 1) & 1][tid] = nph;
        __syncthreads();
    }
}

// LA = 4 + K (L = 16 + K, K = 1..3): a = a1 2^K + a0.  grid = ncols * 2^(4+K); a block holds all 2^LA rows of 2^(8-K) consecutive
// columns l (row segments of 128 bytes and more): stage 1 (thread = (a0, jl), radix 16 over a1), LDS, stage 2 (radix 2^K over a0;
// a thread takes 2^(4-K) columns so that it still moves 16 values).
template <int K>
__global__ __launch_bounds__(THREADS) void k_bb_lde_pa16xs(const u32* __restrict__ coeffs, u32* __restrict__ lde, u32 rate_bits,
                                                           const u32* __restrict__ tw4096, const u32* __restrict__ tw_hi,
                                                           const u32* __restrict__ tw_lo, const u32* __restrict__ pow_lo,
                                                           const u32* __restrict__ pow_hi) {
    constexpr u32 L = 16 + K, LA = 4 + K, R = 1u << K, M = 256u >> K;  // M columns per block
    __shared__ u32 sh[16 * 256];                                          // [k_a1 slot][a0][jl]
    const u32 blocks_per_col = 4096 / M;
    const size_t col = blockIdx.x / blocks_per_col;
    const u32 l0 = (blockIdx.x % blocks_per_col) * M;
    const u32 tid = threadIdx.x, a0 = tid / M, jl = tid % M;
    const size_t n = (size_t)1 << L;
    u32 orig[16];
    {
        const u32* cin = coeffs + col * n + l0 + jl;
#pragma unroll
        for (u32 a1 = 0; a1 < 16; a1++) orig[a1] = cin[(size_t)((a1 << K) + a0) << 12];
    }
    u32 tw[16];
#pragma unroll
    for (u32 s = 1; s < 16; s++) tw[s] = tw4096[((brev4(s) * a0) << (12 - LA)) & 4095];  // w_{2^LA}^(k_a1 a0): the same for every coset
    const u32 s2 = tid >> 4, q = tid & 15;  // stage-2 role: slot s2, columns q + 16 u
    const u32 ka1 = brev4(s2);
    const u32 ncosets = 1u << rate_bits;
    for (u32 c = 0; c < ncosets; c++) {
        const u32* ph = pow_hi + ((size_t)c << LA) + a0;
        u32 x[16];
#pragma unroll
        for (u32 a1 = 0; a1 < 16; a1++) x[a1] = ph[a1 << K];  // s_c^(4096 a)
#pragma unroll
        for (u32 a1 = 0; a1 < 16; a1++) x[a1] = bb::mul(orig[a1], x[a1]);
        dft16<false>(x);
#pragma unroll
        for (u32 s = 0; s < 16; s++) sh[s * 256 + tid] = s ? bb::mul(x[s], tw[s]) : x[s];
        __syncthreads();
#pragma unroll
        for (u32 u = 0; u < M / 16; u++) {
            const u32 jj = q + 16 * u, l = l0 + jj;
            u32 y[16];
#pragma unroll
            for (u32 b = 0; b < R; b++) y[b] = sh[s2 * 256 + b * M + jj];
            dft_small<false, K>(y);
            u32 f = bb::mul(pow_lo[(size_t)c * 4096 + l], bb_tw_split16(tw_hi, tw_lo, ka1 * l));
            const u32 ratio = bb_tw_split16(tw_hi, tw_lo, 16 * l);
            u32* out = lde + (col << (L + rate_bits)) + (size_t)c * n + l;
#pragma unroll
            for (u32 k = 0; k < R; k++) {  // row position = brev_LA(k_a) = slot * 2^K + brevK(k')
                out[(size_t)(s2 * R + brevk(k, K)) << 12] = bb::mul(y[brevk(k, K)], f);
                if (k + 1 < R) f = bb::mul(f, ratio);
            }
        }
        __syncthreads();
    }
}

// LA = K in 1..3 (L = 12 + K): one radix-2^K stage over the 2^K rows, a thread per column l, no LDS.  grid = ncols * 16.
template <int K>
__global__ __launch_bounds__(THREADS) void k_bb_lde_pa_small(const u32* __restrict__ coeffs, u32* __restrict__ lde, u32 rate_bits,
                                                             const u32* __restrict__ tw_hi, const u32* __restrict__ tw_lo,
                                                             const u32* __restrict__ pow_lo, const u32* __restrict__ pow_hi) {
    constexpr u32 L = 12 + K, R = 1u << K;
    const size_t col = blockIdx.x >> 4;
    const u32 l = ((blockIdx.x & 15) << 8) + threadIdx.x;
    const size_t n = (size_t)1 << L;
    const u32* cin = coeffs + col * n + l;
    u32 orig[R];
#pragma unroll
    for (u32 a = 0; a < R; a++) orig[a] = cin[(size_t)a << 12];
    const u32 ncosets = 1u << rate_bits;
    const u32 ratio = bb_tw_split16(tw_hi, tw_lo, l);  // w_n^l
    for (u32 c = 0; c < ncosets; c++) {
        const u32* ph = pow_hi + ((size_t)c << K);
        u32 x[16];
#pragma unroll
        for (u32 a = 0; a < R; a++) x[a] = a ? bb::mul(orig[a], ph[a]) : orig[a];  // s_c^(4096 a)
        dft_small<false, K>(x);
        u32* out = lde + (col << (L + rate_bits)) + (size_t)c * n + l;
        u32 f = pow_lo[(size_t)c * 4096 + l];  // s^l w_n^(k l), k = 0..R-1
#pragma unroll
        for (u32 k = 0; k < R; k++) {
            out[(size_t)brevk(k, K) << 12] = bb::mul(x[brevk(k, K)], f);
            if (k + 1 < R) f = bb::mul(f, ratio);
        }
    }
}

// LA = 4 (L = 16): grid = ncols * 16, block = 16 rows x 256 contiguous columns, one radix-16 stage, no LDS.
__global__ __launch_bounds__(THREADS) void k_bb_lde_pa16x1(const u32* __restrict__ coeffs, u32* __restrict__ lde, u32 rate_bits,
                                                           const u32* __restrict__ tw_hi, const u32* __restrict__ tw_lo,
                                                           const u32* __restrict__ pow_lo, const u32* __restrict__ pow_hi) {
    constexpr u32 L = 16;
    const size_t col = blockIdx.x >> 4;
    const u32 l = ((blockIdx.x & 15) << 8) + threadIdx.x;
    const size_t n = (size_t)1 << L;
    const u32* cin = coeffs + col * n + l;
    u32 orig[16];
#pragma unroll
    for (u32 a = 0; a < 16; a++) orig[a] = cin[(size_t)a << 12];
    const u32 ncosets = 1u << rate_bits;
    const u32 ratio = bb_tw_split16(tw_hi, tw_lo, l);
    for (u32 c = 0; c < ncosets; c++) {
        const u32* ph = pow_hi + (size_t)c * 16;
        u32 x[16];
#pragma unroll
        for (u32 a = 0; a < 16; a++) x[a] = a ? bb::mul(orig[a], ph[a]) : orig[a];
        dft16<false>(x);
        u32* out = lde + (col << (L + rate_bits)) + (size_t)c * n + l;
        u32 f = pow_lo[(size_t)c * 4096 + l];  // s^l w_n^(k l), k = 0..15
#pragma unroll
        for (u32 k = 0; k < 16; k++) {
            out[(size_t)brev4(k) << 12] = bb::mul(x[brev4(k)], f);
            if (k < 15) f = bb::mul(f, ratio);
        }
    }
}

// ------------------------------------------------------------------ inverse NTT passes (LA = 8, LB in {0, 4}, LC = 8)
struct BbInv16Geom {
    u32 L, LB;  // LA = LC = 8, LL = LB + 8
};

// P1: grid = ncols * 2^(LL-4); tile 256 rows (a) x 16 contiguous; rows written in natural k_a order.
// WB (round 4): the input is CANONICAL - the transform is linear and every twiddle product multiplies by the twiddle's value
// (x (w R) / R), so canonical words go through it unchanged in scale and P3's last factor n^-1 R^2 instead of n^-1 R brings the
// coefficients out in Montgomery form: no conversion pass (k_bb_to_mont: a read and a write of the whole witness).  The columns
// somebody reads as VALUES afterwards - the routed wires, for the permutation argument - are written back in Montgomery form
// here, each element by the one thread that has just read it (col < mont_cols).
template <bool WB>
__global__ __launch_bounds__(THREADS) void k_bb_intt16_p1(const u32* src, u32* __restrict__ dst, BbInv16Geom g,
                                                          const u32* __restrict__ tw4096, const u32* __restrict__ tw_hi,
                                                          const u32* __restrict__ tw_lo, u32* src_mont, u32 mont_cols) {
    __shared__ u32 sh[16 * 272];
    const u32 LL = g.LB + 8;
    const u32 tiles_per_col = 1u << (LL - 4);
    const size_t col = blockIdx.x / tiles_per_col;
    const u32 tg = blockIdx.x % tiles_per_col;
    const size_t base = (col << g.L) + ((size_t)tg << 4);
    const u32 tid = threadIdx.x, hi4 = tid >> 4, j = tid & 15;
    u32 x[16];
#pragma unroll
    for (u32 a1 = 0; a1 < 16; a1++) x[a1] = src[base + ((size_t)(a1 * 16 + hi4) << LL) + j];
    if (WB && col < mont_cols) {
#pragma unroll
        for (u32 a1 = 0; a1 < 16; a1++) src_mont[base + ((size_t)(a1 * 16 + hi4) << LL) + j] = bb::to_mont(x[a1]);
    }
    u32 tw[16];
    load_tw16(tw, tw4096, hi4 * 16);
    const u32 l = (tg << 4) + j;
    const u32 ka1 = brev4(hi4);
    // output twiddle w_n^-(k_a l), k_a = k_a1 + 16 k: a geometric progression in k with ratio w_n^-(16 l)
    u32 f = bb_tw_split16(tw_hi, tw_lo, ka1 * l);
    const u32 ratio = bb_tw_split16(tw_hi, tw_lo, 16 * l);
    dft16<true>(x);
#pragma unroll
    for (u32 s = 0; s < 16; s++) sh[s * 272 + tid] = s ? bb::mul(x[s], tw[s]) : x[s];
    __syncthreads();
#pragma unroll
    for (u32 a0 = 0; a0 < 16; a0++) x[a0] = sh[hi4 * 272 + a0 * 16 + j];
    dft16<true>(x);
#pragma unroll
    for (u32 k = 0; k < 16; k++) {
        dst[base + ((size_t)(ka1 + 16 * k) << LL) + j] = bb::mul(x[brev4(k)], f);
        if (k < 15) f = bb::mul(f, ratio);
    }
}

// P2 (LB = 4): grid = ncols * 16 * 16; tile 16 k_a x 16 b x 16 c; src [k_a][b][c] -> dst [k_b][k_a][c]
__global__ __launch_bounds__(THREADS) void k_bb_intt16_p2(const u32* __restrict__ src, u32* __restrict__ dst, u32 L,
                                                          const u32* __restrict__ tw4096) {
    const size_t col = blockIdx.x >> 8;
    const u32 ga = (blockIdx.x >> 4) & 15, gc = blockIdx.x & 15;
    const size_t cbase = col << L;
    const u32 ia = threadIdx.x >> 4, jc = threadIdx.x & 15;
    const u32 ka = 16 * ga + ia, c = 16 * gc + jc;
    u32 x[16];
#pragma unroll
    for (u32 b = 0; b < 16; b++) x[b] = src[cbase + ((size_t)ka << 12) + ((size_t)b << 8) + c];
    u32 tw[16];
    load_tw16(tw, tw4096, c);  // w_4096^-(c k_b)
    dft16<true>(x);
#pragma unroll
    for (u32 s = 0; s < 16; s++)
        dst[cbase + ((size_t)brev4(s) << 16) + ((size_t)ka << 8) + c] = s ? bb::mul(x[s], tw[s]) : x[s];
}

// P2 for LB = K in 1..3 (L = 16 + K): the same pass with a radix-2^K DFT over b; src [k_a][b][c] -> dst [k_b][k_a][c]
template <int K>
__global__ __launch_bounds__(THREADS) void k_bb_intt16_p2s(const u32* __restrict__ src, u32* __restrict__ dst, const u32* __restrict__ tw4096) {
    constexpr u32 L = 16 + K, R = 1u << K;
    const size_t col = blockIdx.x >> 8;
    const u32 ga = (blockIdx.x >> 4) & 15, gc = blockIdx.x & 15;
    const size_t cbase = col << L;
    const u32 ka = 16 * ga + (threadIdx.x >> 4), c = 16 * gc + (threadIdx.x & 15);
    u32 x[16];
#pragma unroll
    for (u32 b = 0; b < R; b++) x[b] = src[cbase + ((size_t)ka << (8 + K)) + ((size_t)b << 8) + c];
    u32 tw[R];
#pragma unroll
    for (u32 s = 1; s < R; s++) tw[s] = tw4096[(brevk(s, K) * c) << (4 - K)];  // w_{2^(8+K)}^-(c k_b)
    dft_small<true, K>(x);
#pragma unroll
    for (u32 s = 0; s < R; s++)
        dst[cbase + ((size_t)brevk(s, K) << 16) + ((size_t)ka << 8) + c] = s ? bb::mul(x[s], tw[s]) : x[s];
}

// P2 for LB = 5, 6 (2^21 and 2^22 rows, round 6): the middle pass as a radix-32 / radix-64 DFT over b in registers - one or two DIF
// layers with compile-time twiddles w_64^-j, then 16-point blocks.  src [k_a][b][c] -> dst [k_b][k_a][c]; tw16k = w_{2^14}^-j, j < 2^14.
template <int LB>
__global__ __launch_bounds__(THREADS) void k_bb_intt16_p2w(const u32* __restrict__ src, u32* __restrict__ dst, const u32* __restrict__ tw16k) {
    constexpr u32 L = 16 + LB, R = 1u << LB;
    const size_t col = blockIdx.x >> 8;
    const u32 ga = (blockIdx.x >> 4) & 15, gc = blockIdx.x & 15;
    const size_t cbase = col << L;
    const u32 ka = 16 * ga + (threadIdx.x >> 4), c = 16 * gc + (threadIdx.x & 15);
    u32 x[R];
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
    for (u32 o = 0; o < R; o += 16) dft16<true>(*reinterpret_cast<u32(*)[16]>(&x[o]));
#pragma unroll
    for (u32 s = 0; s < R; s++) {
        const u32 kb = brevk(s, LB);
        dst[cbase + ((size_t)kb << 16) + ((size_t)ka << 8) + c] = s ? bb::mul(x[s], tw16k[(kb * c) << (6 - LB)]) : x[s];   // w_{2^(8+LB)}^-(c k_b)
    }
}

// P3: grid = ncols * 2^LB * 16; tile 16 k_a x 256 c (c = 16 c1 + c0); src [k_b][k_a][c];
// dst natural k = k_a + 256 k_b + 2^(8+LB) k_c, scaled by n^-1
__global__ __launch_bounds__(THREADS) void k_bb_intt16_p3(const u32* __restrict__ src, u32* __restrict__ dst, BbInv16Geom g,
                                                          const u32* __restrict__ tw4096, u32 n_inv) {
    __shared__ u32 sh[16 * 272];
    const u32 nb = 1u << g.LB;
    const size_t col = blockIdx.x / (nb * 16);
    const u32 rem = blockIdx.x % (nb * 16);
    const u32 kb = rem >> 4, ga = rem & 15;
    const size_t cbase = col << g.L;
    const size_t sbase = cbase + ((size_t)kb << 16) + ((size_t)(16 * ga) << 8);
    const u32 tid = threadIdx.x, hi4 = tid >> 4, lo4 = tid & 15;
    u32 x[16];
    // stage 1 thread = (ia = hi4, c0 = lo4): digit c1
#pragma unroll
    for (u32 c1 = 0; c1 < 16; c1++) x[c1] = src[sbase + hi4 * 256 + c1 * 16 + lo4];
    u32 tw[16];
    load_tw16(tw, tw4096, lo4 * 16);  // w_256^-(k_c1 c0)
    dft16<true>(x);
#pragma unroll
    for (u32 s = 0; s < 16; s++) sh[s * 272 + lo4 * 17 + hi4] = s ? bb::mul(x[s], tw[s]) : x[s];  // [k_c1 slot][c0][ia], rows padded to 17
    __syncthreads();
    // stage 2 thread = (k_c1 slot = hi4, ia = lo4): digit c0
#pragma unroll
    for (u32 c0 = 0; c0 < 16; c0++) x[c0] = sh[hi4 * 272 + c0 * 17 + lo4];
    dft16<true>(x);
    const u32 kc1 = brev4(hi4);
#pragma unroll
    for (u32 s = 0; s < 16; s++) {
        const u32 kc = kc1 + 16 * brev4(s);
        dst[cbase + ((size_t)kc << (8 + g.LB)) + ((size_t)kb << 8) + 16 * ga + lo4] = bb::mul(x[s], n_inv);
    }
}

// ------------------------------------------------------------------ launchers (called from kernels_ntt.hip's dispatchers)

// canonical_src != nullptr: the input is canonical (= canonical_src, writable), its first mont_cols columns are overwritten with their
// Montgomery form, the coefficients come out in Montgomery form like those of a Montgomery-form input
bool bb_intt_columns_r16(const u32* src, u32* coeffs, u32* scratch, size_t ncols, const BbNttTables& t, hipStream_t stream,
                         u32* canonical_src, size_t mont_cols) {
    const u32 L = t.log_n;
    if (L < 16 || L > 22) return false;
    BbInv16Geom g{L, L - 16};
    const u32 LL = g.LB + 8;
    u32* p1_dst = g.LB ? coeffs : scratch;
    const u32 n_inv = canonical_src ? bb::to_mont(t.n_inv) : t.n_inv;   // n^-1 R^2 : n^-1 R
    if (canonical_src)
        hipLaunchKernelGGL(k_bb_intt16_p1<true>, dim3((u32)(ncols << (LL - 4))), dim3(THREADS), 0, stream, canonical_src, p1_dst, g, t.tw4096_inv,
                           t.tw_hi_inv, t.tw_lo_inv, canonical_src, (u32)std::min<size_t>(mont_cols, ncols));
    else
        hipLaunchKernelGGL(k_bb_intt16_p1<false>, dim3((u32)(ncols << (LL - 4))), dim3(THREADS), 0, stream, src, p1_dst, g, t.tw4096_inv,
                           t.tw_hi_inv, t.tw_lo_inv, (u32*)nullptr, 0u);
    const dim3 g2((u32)(ncols << 8));
    if (g.LB == 6) hipLaunchKernelGGL(k_bb_intt16_p2w<6>, g2, dim3(THREADS), 0, stream, coeffs, scratch, t.tw16k_inv);
    else if (g.LB == 5) hipLaunchKernelGGL(k_bb_intt16_p2w<5>, g2, dim3(THREADS), 0, stream, coeffs, scratch, t.tw16k_inv);
    else if (g.LB == 4) hipLaunchKernelGGL(k_bb_intt16_p2, g2, dim3(THREADS), 0, stream, coeffs, scratch, L, t.tw4096_inv);
    else if (g.LB == 3) hipLaunchKernelGGL(k_bb_intt16_p2s<3>, g2, dim3(THREADS), 0, stream, coeffs, scratch, t.tw4096_inv);
    else if (g.LB == 2) hipLaunchKernelGGL(k_bb_intt16_p2s<2>, g2, dim3(THREADS), 0, stream, coeffs, scratch, t.tw4096_inv);
    else if (g.LB == 1) hipLaunchKernelGGL(k_bb_intt16_p2s<1>, g2, dim3(THREADS), 0, stream, coeffs, scratch, t.tw4096_inv);
    hipLaunchKernelGGL(k_bb_intt16_p3, dim3((u32)(ncols << (g.LB + 4))), dim3(THREADS), 0, stream, scratch, coeffs, g,
                       t.tw4096_inv, n_inv);
    return true;
}

bool bb_lde_pa_r16(const u32* coeffs, u32* lde, size_t ncols, const BbNttTables& t, const BbCosetTables& ct, hipStream_t stream) {
    const u32 L = t.log_n;
    if (L == 20) {
        hipLaunchKernelGGL(k_bb_lde_pa16x2<32>, dim3((u32)(ncols << 7)), dim3(512), 0, stream, coeffs, lde, L, ct.rate_bits,
                           t.tw4096_fwd, t.tw_hi_fwd, t.tw_lo_fwd, ct.pow_lo, ct.pow_hi);
        return true;
    }
    if (L == 21 || L == 22) {
        if (L == 21) {
            hipLaunchKernelGGL(k_bb_lde_pa32<1>, dim3((u32)(ncols << 7)), dim3(512), 0, stream, coeffs, lde, ct.rate_bits, t.tw4096_fwd,
                               t.tw_hi_fwd, t.tw_lo_fwd, ct.pow_lo, ct.pow_hi);
        } else {   // twiddles of the 2^20-row transform (t.wide), cosets of the rate r + 2 (ct.fine), s_c^(4096 a') of this size
            if (!t.wide || !ct.fine) return false;
            hipLaunchKernelGGL(k_bb_lde_pa16x2w<2>, dim3((u32)(ncols << 7)), dim3(1024), 0, stream, coeffs, lde, ct.rate_bits, t.tw4096_fwd,
                               t.wide->tw_hi_fwd, t.wide->tw_lo_fwd, ct.fine->pow_lo, ct.pow_hi);
        }
        return true;
    }
#define GB_PAS(KK)                                                                                                        \
    hipLaunchKernelGGL(k_bb_lde_pa16xs<KK>, dim3((u32)(ncols << (4 + KK))), dim3(THREADS), 0, stream, coeffs, lde, ct.rate_bits, \
                       t.tw4096_fwd, t.tw_hi_fwd, t.tw_lo_fwd, ct.pow_lo, ct.pow_hi)
    if (L >= 13 && L <= 15) {
        const dim3 grid((u32)(ncols << 4));
        if (L == 13) hipLaunchKernelGGL(k_bb_lde_pa_small<1>, grid, dim3(THREADS), 0, stream, coeffs, lde, ct.rate_bits, t.tw_hi_fwd, t.tw_lo_fwd, ct.pow_lo, ct.pow_hi);
        if (L == 14) hipLaunchKernelGGL(k_bb_lde_pa_small<2>, grid, dim3(THREADS), 0, stream, coeffs, lde, ct.rate_bits, t.tw_hi_fwd, t.tw_lo_fwd, ct.pow_lo, ct.pow_hi);
        if (L == 15) hipLaunchKernelGGL(k_bb_lde_pa_small<3>, grid, dim3(THREADS), 0, stream, coeffs, lde, ct.rate_bits, t.tw_hi_fwd, t.tw_lo_fwd, ct.pow_lo, ct.pow_hi);
        return true;
    }
    if (L == 17) { GB_PAS(1); return true; }
    if (L == 18) { GB_PAS(2); return true; }
    if (L == 19) { GB_PAS(3); return true; }
#undef GB_PAS
    if (L == 16) {
        hipLaunchKernelGGL(k_bb_lde_pa16x1, dim3((u32)(ncols << 4)), dim3(THREADS), 0, stream, coeffs, lde, ct.rate_bits,
                           t.tw_hi_fwd, t.tw_lo_fwd, ct.pow_lo, ct.pow_hi);
        return true;
    }
    return false;
}

void bb_lde_pb_r16(u32* lde, size_t ntiles, const BbNttTables& t, hipStream_t stream) {
    const u32 grid = (u32)std::min<size_t>(ntiles, 256 * 7);  // persistent: 7 workgroups per CU (65 VGPRs, 17 KB LDS each)
    hipLaunchKernelGGL(k_bb_lde_pb16, dim3(grid), dim3(THREADS), 0, stream, lde, t.tw4096_fwd, (u32)ntiles);
}

}  // namespace gbk
