// Microbenchmark (go / no-go for the grouped partial rounds, csrc/poseidon_gl_grouped.hpp): the lane-per-state Poseidon-12
// permutation with all 30 MDS layers as single MFMA layers (permute_mont_mfma_naive, the round-3 product) against the same
// permutation with the 22 partial rounds in groups of GB_POSEIDON_GROUP.  Checks: grouped == naive on every state, and both
// == the host's defining permutation (poseidon_gl_host.hpp) on a sample; then times `reps` chained permutations per lane.
//
//   for g in 2 3 4 5; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGB_POSEIDON_GROUP=$g -Iplonky2_goldibear_amd/csrc \
//       tools/microbench_poseidon_groups.hip -o tools/bin/mbgrp$g; done;  tools/bin/mbgrp4 [log2 states] [reps]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "poseidon_gl_grouped.hpp"
#include "poseidon_gl_host.hpp"

typedef unsigned long long u64;
typedef unsigned int u32;
#ifndef MB_OCC
#define MB_OCC 4
#endif

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

template <bool GROUPED>
__global__ __launch_bounds__(256, MB_OCC) void k_perm(const u64* __restrict__ in, u64* __restrict__ out, size_t n, int reps) {
    __shared__ poseidon_gl::v4i ops_lds[poseidon_gl::GROUP_LDS_V4 > 0 ? poseidon_gl::GROUP_LDS_V4 : 1];
    if (GROUPED) poseidon_gl::group_ops_init(ops_lds);
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;   // n is a multiple of 256
    const poseidon_gl::MdsOperand amat = poseidon_gl::mds_mfma_matrix();
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = poseidon_gl::to_mont(in[i * n + t]);
    for (int it = 0; it < reps; it++) {
        if (GROUPED) poseidon_gl::permute_mont_mfma_grouped(s, amat, ops_lds + (threadIdx.x & 63));
        else poseidon_gl::permute_mont_mfma_naive(s, amat);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) out[i * n + t] = poseidon_gl::from_mont(s[i]);
}

static u64 splitmix(u64& x) {
    u64 z = (x += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    const int logn = argc > 1 ? atoi(argv[1]) : 21;
    const int reps = argc > 2 ? atoi(argv[2]) : 8;
    const size_t n = (size_t)1 << logn;
    std::vector<u64> h(12 * n);
    u64 seed = 0xC0FFEE;
    const u64 P = 0xFFFFFFFF00000001ULL;
    const u64 edge[8] = {0, 1, P - 1, P - 2, 0x8080808080808080ULL % P, 0x7F7F7F7F7F7F7F7FULL, 0xFFFFFFFFULL, 0xFFFFFFFF00000000ULL};
    for (size_t i = 0; i < 12 * n; i++) {
        u64 v = splitmix(seed);
        h[i] = (v & 0xF) == 0 ? edge[(v >> 4) & 7] : v % P;
    }
    u64 *din, *d0, *d1;
    CHECK(hipMalloc(&din, 12 * n * 8));
    CHECK(hipMalloc(&d0, 12 * n * 8));
    CHECK(hipMalloc(&d1, 12 * n * 8));
    CHECK(hipMemcpy(din, h.data(), 12 * n * 8, hipMemcpyHostToDevice));
    // correctness, one permutation
    hipLaunchKernelGGL(k_perm<false>, dim3(n / 256), dim3(256), 0, 0, din, d0, n, 1);
    hipLaunchKernelGGL(k_perm<true>, dim3(n / 256), dim3(256), 0, 0, din, d1, n, 1);
    CHECK(hipDeviceSynchronize());
    std::vector<u64> o0(12 * n), o1(12 * n);
    CHECK(hipMemcpy(o0.data(), d0, 12 * n * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(o1.data(), d1, 12 * n * 8, hipMemcpyDeviceToHost));
    size_t bad = 0, bad_host = 0;
    for (size_t i = 0; i < 12 * n; i++) bad += o0[i] != o1[i];
    const size_t sample = n < 65536 ? n : 65536;
    for (size_t t = 0; t < sample; t++) {
        u64 st[12];
        for (int i = 0; i < 12; i++) st[i] = h[i * n + t];
        poseidon_gl_host::permute(st);
        for (int i = 0; i < 12; i++) bad_host += st[i] != o1[i * n + t];
    }
    printf("G=%d states=2^%d: grouped vs naive mismatching words=%zu, grouped vs host (first %zu states) mismatching words=%zu\n",
           poseidon_gl::GROUP_G, logn, bad, sample, bad_host);
    // timing
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int variant = 0; variant < 2; variant++) {
        float best = 1e30f;
        for (int run = 0; run < 4; run++) {
            CHECK(hipEventRecord(e0));
            if (variant == 0) hipLaunchKernelGGL(k_perm<false>, dim3(n / 256), dim3(256), 0, 0, din, d0, n, reps);
            else hipLaunchKernelGGL(k_perm<true>, dim3(n / 256), dim3(256), 0, 0, din, d1, n, reps);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        printf("  %-8s %8.3f ms  %6.3f G perm/s\n", variant ? "grouped" : "naive", best, (double)n * reps / best / 1e6);
    }
    CHECK(hipMemcpy(o0.data(), d0, 12 * n * 8, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(o1.data(), d1, 12 * n * 8, hipMemcpyDeviceToHost));
    bad = 0;
    for (size_t i = 0; i < 12 * n; i++) bad += o0[i] != o1[i];
    printf("  after %d chained permutations: mismatching words=%zu\n", reps, bad);
    return bad || bad_host ? 1 : 0;
}
