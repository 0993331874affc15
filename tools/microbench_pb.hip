// Where does the contiguous LDE pass (k_gl_lde_pb16) spend its time?  Variants of the same kernel:
//   0 full   1 no arithmetic (loads, LDS exchanges, barriers, stores only)   2 arithmetic only (no LDS, no barriers)
//   3 full without the global twiddle loads (twiddle = a register constant)
// Build on the GPU box:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I plonky2_goldibear_amd/csrc -o /tmp/mbpb tools/microbench_pb.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "gl_field.hpp"
typedef unsigned long long u64;
typedef unsigned int u32;
#define NTT16_NO_KERNELS
namespace gbk {
__device__ __forceinline__ constexpr u32 brev4(u32 x) { return ((x & 1) << 3) | ((x & 2) << 1) | ((x & 4) >> 1) | ((x & 8) >> 3); }
template <int K>
__device__ __forceinline__ u64 mul_pow2(u64 x) {
    if constexpr (K == 0) return x;
    else if constexpr (K < 64) { const u64 lo = x << K, hi = x >> (64 - K); return gl::canon(gl::fold128((u32)lo, (u32)(lo >> 32), (u32)hi, (u32)(hi >> 32))); }
    else { constexpr u64 c = ((u64)1 << (K - 32)) - ((u64)1 << (K - 64)); return gl::mul(x, c); }
}
template <bool INV, int M>
__device__ __forceinline__ u64 sub_twiddle(u64 a, u64 b) {
    constexpr int e = (INV ? 36 * M : 156 * M) % 192;
    if constexpr (e == 0) return gl::sub(a, b);
    else if constexpr (e < 96) return mul_pow2<e>(gl::sub(a, b));
    else return mul_pow2<e - 96>(gl::sub(b, a));
}
template <bool INV, int H, int J>
__device__ __forceinline__ void bfly(u64& a, u64& b) { u64 s = gl::add(a, b); b = sub_twiddle<INV, J*(8 / H)>(a, b); a = s; }
template <bool INV>
__device__ __forceinline__ void dft16(u64 (&x)[16]) {
#define B8(J) bfly<INV, 8, J>(x[J], x[J + 8]);
    B8(0) B8(1) B8(2) B8(3) B8(4) B8(5) B8(6) B8(7)
#define B4(O, J) bfly<INV, 4, J>(x[O + J], x[O + J + 4]);
    B4(0, 0) B4(0, 1) B4(0, 2) B4(0, 3) B4(8, 0) B4(8, 1) B4(8, 2) B4(8, 3)
#define B2(O, J) bfly<INV, 2, J>(x[O + J], x[O + J + 2]);
    B2(0, 0) B2(0, 1) B2(4, 0) B2(4, 1) B2(8, 0) B2(8, 1) B2(12, 0) B2(12, 1)
#define B1(O) bfly<INV, 1, 0>(x[O], x[O + 1]);
    B1(0) B1(2) B1(4) B1(6) B1(8) B1(10) B1(12) B1(14)
}
#ifndef PB_WAVES
#define PB_WAVES 1
#endif
#ifndef PB_LDS
#define PB_LDS (16 * 272)
#endif
template <int V>
__global__ __launch_bounds__(256, PB_WAVES) void k_pb(u64* __restrict__ lde, const u64* __restrict__ tw4096) {
    __shared__ u64 sh[PB_LDS];
    u64* p = lde + ((size_t)blockIdx.x << 12);
    const u32 tid = threadIdx.x;
    u64 x[16], tw[16];
#pragma unroll
    for (u32 d = 0; d < 16; d++) x[d] = p[d * 256 + tid];
#pragma unroll
    for (u32 s = 1; s < 16; s++) tw[s] = V == 3 ? (u64)(s * 77 + tid) : tw4096[brev4(s) * tid];
    if (V != 1) dft16<false>(x);
#pragma unroll
    for (u32 s = 0; s < 16; s++) { u64 v = (s && V != 1) ? gl::mul(x[s], tw[s]) : x[s]; if (V != 2) sh[s * 272 + tid] = v; else x[s] = v; }
    const u32 hi4 = tid >> 4, lo4 = tid & 15;
    if (V != 2) {
        __syncthreads();
#pragma unroll
        for (u32 d = 0; d < 16; d++) x[d] = sh[hi4 * 272 + d * 16 + lo4];
    }
#pragma unroll
    for (u32 s = 1; s < 16; s++) tw[s] = V == 3 ? (u64)(s * 79 + tid) : tw4096[brev4(s) * lo4 * 16];
    if (V != 1) dft16<false>(x);
    if (V != 2) __syncthreads();
#pragma unroll
    for (u32 s = 0; s < 16; s++) { u64 v = (s && V != 1) ? gl::mul(x[s], tw[s]) : x[s]; if (V != 2) sh[hi4 * 272 + lo4 * 17 + s] = v; else x[s] = v; }
    if (V != 2) {
        __syncthreads();
#pragma unroll
        for (u32 d = 0; d < 16; d++) x[d] = sh[hi4 * 272 + d * 17 + lo4];
    }
    if (V != 1) dft16<false>(x);
    if (V != 2) {
        __syncthreads();
#pragma unroll
        for (u32 s = 0; s < 16; s++) sh[tid * 17 + s] = x[s];
        __syncthreads();
#pragma unroll
        for (u32 it = 0; it < 16; it++) { const u32 q = it * 256 + tid; p[q] = sh[(q >> 4) * 17 + (q & 15)]; }
    } else {
#pragma unroll
        for (u32 it = 0; it < 16; it++) p[it * 256 + tid] = x[it];
    }
}
}  // namespace gbk
template <int V>
void run(const char* name, u64* d, u64* tw, size_t tiles) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(gbk::k_pb<V>, dim3(tiles), dim3(256), 0, 0, d, tw);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL(gbk::k_pb<V>, dim3(tiles), dim3(256), 0, 0, d, tw);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("%-28s %7.3f ms  (%.1f us per 2^23-element column, %.2f TB/s r+w)\n", name, ms, ms * 1e3 / (tiles / 2048.0), tiles * 4096.0 * 16 / ms / 1e9);
}
int main() {
    const size_t cols = 135, tiles = cols * 2048;
    u64 *d, *tw;
    hipMalloc(&d, tiles * 4096 * 8); hipMalloc(&tw, 4096 * 8);
    hipMemset(d, 1, tiles * 4096 * 8); hipMemset(tw, 3, 4096 * 8);
    run<0>("full", d, tw, tiles);
    run<1>("no arithmetic", d, tw, tiles);
    run<2>("arithmetic only (no LDS)", d, tw, tiles);
    run<3>("full, no twiddle loads", d, tw, tiles);
    return 0;
}
