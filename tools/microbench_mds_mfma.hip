// Microbenchmark (go / no-go): the Poseidon-12 MDS layer for one state per lane, as today's VALU code
// (24 v_mad_u64_u32 per output word on 32-bit halves) against byte planes on the matrix pipe: eight
// v_mfma_i32_32x32x32_i8 per layer (one per byte plane of the 64-bit words) with the constant 12 x 12 matrix as a
// block-diagonal A operand, so that every lane gets its OWN state's twelve plane sums back (no cross-lane traffic):
//   lane l = (r = l & 31, h = l >> 5); B: lane l supplies k = 16 h + j (j < 16) for column r  -> byte p of word j of ITS state
//   A: row m = (q & 3) + 8 (q >> 2) + 4 h' carries M[q][.] in k-block h' only      (q < 12: output word, h' in {0, 1})
//   D: lane l register q = D[(q & 3) + 8 (q >> 2) + 4 h][r] = sum_i M[q][i] byte_p(word i of state l)
// Bytes are fed as b ^ 0x80 (signed b - 128); the offset is a per-row constant.  Recombination: v_mad_i64_i32 by 2^(8 (p & 3)).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iplonky2_goldibear_amd/csrc tools/microbench_mds_mfma.hip -o tools/bin/mbmds && tools/bin/mbmds
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "poseidon_gl.hpp"

typedef unsigned long long u64;
typedef unsigned int u32;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

static constexpr u64 P = 0xFFFFFFFF00000001ULL;
__host__ __device__ constexpr u32 circ_at(int i) {
    constexpr u32 c[12] = {GL_POSEIDON_MDS_CIRC_LIST};
    return c[i];
}
__host__ __device__ constexpr u32 mds_entry(int r, int c) { return circ_at(((c - r) % 12 + 12) % 12) + (r == 0 && c == 0 ? 8u : 0u); }

// ---------------------------------------------------------------- VALU layer (poseidon_gl::mds_layer) + a light nonlinearity
template <int SBOX>
__device__ __forceinline__ void between(u64 (&s)[12]) {
    if (SBOX) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = poseidon_gl::sbox(s[i]);
    }
}

template <int SBOX>
__global__ __launch_bounds__(256, 6) void k_valu(const u64* __restrict__ in, u64* __restrict__ out, int layers) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = in[i * (size_t)gridDim.x * 256 + t];
    for (int it = 0; it < layers; it++) {
        between<SBOX>(s);
        poseidon_gl::mds_layer(s, poseidon_gl::ZERO_RC);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) out[i * (size_t)gridDim.x * 256 + t] = gl::canon(s[i]);
}

// ---------------------------------------------------------------- MFMA layer
__device__ __forceinline__ u32 perm(u32 hi, u32 lo, u32 sel) { return __builtin_amdgcn_perm(hi, lo, sel); }

// 4 x 4 byte transpose: w[0..3] -> t[p] = [w0.b_p, w1.b_p, w2.b_p, w3.b_p]
__device__ __forceinline__ void transpose4(u32 w0, u32 w1, u32 w2, u32 w3, u32 (&t)[4]) {
    const u32 a_lo = perm(w1, w0, 0x05010400u);  // [w0.b0, w1.b0, w0.b1, w1.b1]
    const u32 a_hi = perm(w1, w0, 0x07030602u);  // [w0.b2, w1.b2, w0.b3, w1.b3]
    const u32 b_lo = perm(w3, w2, 0x05010400u);
    const u32 b_hi = perm(w3, w2, 0x07030602u);
    t[0] = perm(b_lo, a_lo, 0x05040100u);        // [a.b0, a.b1, b.b0, b.b1]
    t[1] = perm(b_lo, a_lo, 0x07060302u);
    t[2] = perm(b_hi, a_hi, 0x05040100u);
    t[3] = perm(b_hi, a_hi, 0x07060302u);
}

// a * b + c, signed 32 x 32 + 64 in one v_mad_i64_i32 (the compiler expands the C expression into shift + sign-extend + add)
__device__ __forceinline__ long long mad_i64(int a, int b, long long c) {
    long long d;
    u64 carry_unused;
    asm("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry_unused) : "v"(a), "s"(b), "v"(c));
    return d;
}

struct MdsConsts {
    u64 k[12];  // what the signed-byte offset and the accumulator biases add up to, mod p (to be subtracted)
};
static constexpr long long BIAS = 1ll << 41;   // keeps the signed plane sums' accumulators non-negative

__device__ __forceinline__ void mds_layer_mfma(u64 (&s)[12], const v4i amat, const u64* __restrict__ kfix) {
    u32 w[24];
#pragma unroll
    for (int i = 0; i < 12; i++) {
        w[i] = (u32)s[i] ^ 0x80808080u;
        w[12 + i] = (u32)(s[i] >> 32) ^ 0x80808080u;
    }
    u32 pl[8][4];  // pl[p][g]: bytes p of words 4g .. 4g+3; pl[p][3] = unused K lanes (A is zero there)
#pragma unroll
    for (int half = 0; half < 2; half++)
#pragma unroll
        for (int g = 0; g < 3; g++) {
            u32 t[4];
            transpose4(w[12 * half + 4 * g], w[12 * half + 4 * g + 1], w[12 * half + 4 * g + 2], w[12 * half + 4 * g + 3], t);
#pragma unroll
            for (int p = 0; p < 4; p++) pl[4 * half + p][g] = t[p];
        }
    long long lo[12], hi[12];
#pragma unroll
    for (int q = 0; q < 12; q++) {
        lo[q] = BIAS + (long long)(u32)kfix[q];
        hi[q] = BIAS + (long long)(kfix[q] >> 32);
    }
    const v16i zero = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // planes two at a time: t = d_p + 2^8 d_(p+1) in 32 bits (|t| < 2^25), then one signed 32 x 32 + 64 mad per pair and word
#pragma unroll
    for (int pp = 0; pp < 4; pp++) {
        v4i b0, b1;
        b0[0] = (int)pl[2 * pp][0]; b0[1] = (int)pl[2 * pp][1]; b0[2] = (int)pl[2 * pp][2]; b0[3] = b0[0];       // k = 12..15 meet zeros in A
        b1[0] = (int)pl[2 * pp + 1][0]; b1[1] = (int)pl[2 * pp + 1][1]; b1[2] = (int)pl[2 * pp + 1][2]; b1[3] = b1[0];
        const v16i d0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(amat, b0, zero, 0, 0, 0);
        const v16i d1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(amat, b1, zero, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 12; q++) {
            const int t = (int)(((u32)d1[q] << 8) + (u32)d0[q]);
            if (pp < 2) lo[q] = mad_i64(t, pp & 1 ? 65536 : 1, lo[q]);
            else hi[q] = mad_i64(t, pp & 1 ? 65536 : 1, hi[q]);
        }
    }
#pragma unroll
    for (int q = 0; q < 12; q++) s[q] = poseidon_gl::fold_halves((u64)lo[q], (u64)hi[q]);
}

// the constant A operand of this lane: row r = lane & 31 belongs to k-block h' = (r >> 2) & 1 and output word q = (r & 3) + 4 (r >> 3)
__device__ __forceinline__ v4i build_amat() {
    const u32 l = threadIdx.x & 63, r = l & 31, h = l >> 5;
    const u32 hp = (r >> 2) & 1, q = (r & 3) + 4 * (r >> 3);
    v4i a = {0, 0, 0, 0};
    if (h == hp && q < 12) {
        u32 bytes[16];
#pragma unroll
        for (int i = 0; i < 16; i++) bytes[i] = i < 12 ? mds_entry((int)q, i) : 0u;
#pragma unroll
        for (int g = 0; g < 4; g++) a[g] = (int)(bytes[4 * g] | (bytes[4 * g + 1] << 8) | (bytes[4 * g + 2] << 16) | (bytes[4 * g + 3] << 24));
    }
    return a;
}

template <int SBOX, int OCC>
__global__ __launch_bounds__(256, OCC) void k_mfma(const u64* __restrict__ in, u64* __restrict__ out, int layers, MdsConsts mc) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    const v4i amat = build_amat();
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = in[i * (size_t)gridDim.x * 256 + t];
    for (int it = 0; it < layers; it++) {
        between<SBOX>(s);
        mds_layer_mfma(s, amat, mc.k);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) out[i * (size_t)gridDim.x * 256 + t] = gl::canon(s[i]);
}

static u64 mulmod(u64 a, u64 b) { return (u64)((unsigned __int128)a * b % P); }
static u64 submod(u64 a, u64 b) { return a >= b ? a - b : a + (P - b); }

int main() {
    const int blocks = 256 * 16, layers = 64;
    const size_t nstates = (size_t)blocks * 256, count = nstates * 12;
    std::vector<u64> h(count);
    u64 x = 0x9E3779B97F4A7C15ULL;
    for (auto& v : h) {
        x += 0x9E3779B97F4A7C15ULL;
        u64 z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        v = (z ^ (z >> 31)) % P;
    }
    // the first states carry edge bytes: 0x00 / 0x7f / 0x80 / 0xff in every position
    const u64 edge[4] = {0, 0x7f7f7f7f7f7f7f7fULL, 0x8080808080808080ULL, 0xFFFFFFFF00000000ULL};
    for (int e = 0; e < 4; e++)
        for (int i = 0; i < 12; i++) h[i * nstates + e] = edge[(e + i) % 4];
    MdsConsts mc;
    for (int q = 0; q < 12; q++) {
        u64 rowsum = 0;
        for (int i = 0; i < 12; i++) rowsum += mds_entry(q, i);
        // true sum = signed sum + 128 rowsum per plane; accumulators start at BIAS (lo) and BIAS (hi, weight 2^32):
        // value = acc_lo + 2^32 acc_hi + 128 rowsum 0x0101..01 - BIAS - 2^32 BIAS.  kfix = that correction mod p, added at the start.
        u64 corr = mulmod(mulmod(128, rowsum % P), 0x0101010101010101ULL % P);
        corr = submod(corr, (u64)BIAS % P);
        corr = submod(corr, mulmod((u64)BIAS % P, (1ULL << 32) % P));
        mc.k[q] = corr;
    }
    u64 *din, *d1, *d2;
    CHECK(hipMalloc(&din, count * 8));
    CHECK(hipMalloc(&d1, count * 8));
    CHECK(hipMalloc(&d2, count * 8));
    CHECK(hipMemcpy(din, h.data(), count * 8, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    auto timeit = [&](auto launch, const char* name, double* ms_out) {
        launch();
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 5; i++) launch();
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        *ms_out = ms / 5;
        printf("%-28s %8.3f ms   %7.1f ns per wave-layer\n", name, ms / 5, ms / 5 * 1e6 / ((double)nstates / 64 * layers));
    };
    double tv, tm, tvs, tms;
    timeit([&] { hipLaunchKernelGGL(k_valu<0>, dim3(blocks), dim3(256), 0, 0, din, d1, layers); }, "MDS only, VALU", &tv);
    double dummy;
    timeit([&] { hipLaunchKernelGGL((k_mfma<0, 6>), dim3(blocks), dim3(256), 0, 0, din, d2, layers, mc); }, "MDS only, MFMA planes occ6", &dummy);
    timeit([&] { hipLaunchKernelGGL((k_mfma<0, 5>), dim3(blocks), dim3(256), 0, 0, din, d2, layers, mc); }, "MDS only, MFMA planes occ5", &dummy);
    timeit([&] { hipLaunchKernelGGL((k_mfma<0, 4>), dim3(blocks), dim3(256), 0, 0, din, d2, layers, mc); }, "MDS only, MFMA planes occ4", &tm);
    std::vector<u64> r1(count), r2(count);
    CHECK(hipMemcpy(r1.data(), d1, count * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(r2.data(), d2, count * 8, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (size_t i = 0; i < count; i++) bad += r1[i] != r2[i];
    printf("MDS only: %zu of %zu words differ\n", bad, count);
    timeit([&] { hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(256), 0, 0, din, d1, layers); }, "full round, VALU", &tvs);
    timeit([&] { hipLaunchKernelGGL((k_mfma<1, 6>), dim3(blocks), dim3(256), 0, 0, din, d2, layers, mc); }, "full round, MFMA planes occ6", &dummy);
    timeit([&] { hipLaunchKernelGGL((k_mfma<1, 5>), dim3(blocks), dim3(256), 0, 0, din, d2, layers, mc); }, "full round, MFMA planes occ5", &dummy);
    timeit([&] { hipLaunchKernelGGL((k_mfma<1, 4>), dim3(blocks), dim3(256), 0, 0, din, d2, layers, mc); }, "full round, MFMA planes occ4", &tms);
    CHECK(hipMemcpy(r1.data(), d1, count * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(r2.data(), d2, count * 8, hipMemcpyDeviceToHost));
    size_t bad2 = 0;
    for (size_t i = 0; i < count; i++) bad2 += r1[i] != r2[i];
    printf("full round: %zu of %zu words differ\n", bad2, count);
    printf("{\"mds_valu_ms\": %.4f, \"mds_mfma_ms\": %.4f, \"round_valu_ms\": %.4f, \"round_mfma_ms\": %.4f, \"mismatch\": %zu}\n", tv, tm, tvs,
           tms, bad + bad2);
    return bad + bad2 ? 2 : 0;
}
