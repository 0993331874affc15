// LAB builds of the Poseidon-12 kernels - never the product.  The ONE place where a preprocessor switch changes the hash kernels:
//   -DGB_LAB    tools/probe_leaves.py: s_memtime stamps of a few waves at the segment boundaries of the permutation (the attribution
//               of HISTORY.md, round 5); MdsOperand grows a trace pointer, the library an export gb_probe_setup
// Without GB_LAB everything here is empty.
// Included by poseidon_gl.hpp at file scope.
#pragma once
#ifdef GB_LAB
#define GB_LAB_PROBE_FIELDS        \
    mutable ulonglong2* trace;     \
    mutable unsigned pidx;
// every `wave_step`-th wave of a kernel that calls GB_LAB_PROBE_INIT writes a (s_memtime, site) pair at each probe site into its
// slice of a trace buffer (tools/probe_leaves.py sets it up through gb_probe_setup, an export of lab builds only)
struct GbLabProbeCfg {
    ulonglong2* buf;
    unsigned max_per_wave, wave_step, nslots;
};
static __device__ GbLabProbeCfg gb_probe_cfg;   // (one per translation unit; the traced kernels live in kernels_merkle.hip, which defines the setter)
#define GB_LAB_DEFINE_PROBE_SETUP                                                                                      \
    extern "C" int gb_probe_setup(void* dev_buf, unsigned max_per_wave, unsigned wave_step, unsigned nslots) {         \
        GbLabProbeCfg c{static_cast<ulonglong2*>(dev_buf), max_per_wave, wave_step, nslots};                           \
        return (int)hipMemcpyToSymbol(HIP_SYMBOL(gb_probe_cfg), &c, sizeof c);                                         \
    }
#define GB_LAB_PROBE_INIT(amat)                                                                                              \
    do {                                                                                                                     \
        const unsigned wave = __builtin_amdgcn_readfirstlane(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6));          \
        const GbLabProbeCfg pc = gb_probe_cfg;                                                                               \
        const bool traced = pc.buf && pc.wave_step && wave % pc.wave_step == 0 && wave / pc.wave_step < pc.nslots;           \
        (amat).trace = traced ? pc.buf + (size_t)(wave / pc.wave_step) * pc.max_per_wave : nullptr;                          \
        (amat).pidx = 0;                                                                                                     \
    } while (0)
namespace poseidon_gl {
typedef unsigned long long u64;
typedef unsigned int u32;
template <int N>
__device__ __forceinline__ void probe_pin(u64 (&s)[N]) {
    if constexpr (N == 12)
        asm volatile("" : "+v"(s[0]), "+v"(s[1]), "+v"(s[2]), "+v"(s[3]), "+v"(s[4]), "+v"(s[5]), "+v"(s[6]), "+v"(s[7]), "+v"(s[8]), "+v"(s[9]),
                     "+v"(s[10]), "+v"(s[11]));
    else
#pragma unroll
        for (int i = 0; i < N; i++) asm volatile("" : "+v"(s[i]));
}
template <int N>
__device__ __forceinline__ void probe_pin(long long (&s)[N]) {
#pragma unroll
    for (int i = 0; i < N; i++) asm volatile("" : "+v"(s[i]));
}
template <int A, int B>
__device__ __forceinline__ void probe_pin(u32 (&p)[A][B]) {
#pragma unroll
    for (int i = 0; i < A; i++)
#pragma unroll
        for (int j = 0; j < B; j++) asm volatile("" : "+v"(p[i][j]));
}
template <class X, class... Rest>
__device__ __forceinline__ void probe_pin_all(X& x, Rest&... rest) {
    probe_pin(x);
    if constexpr (sizeof...(rest) > 0) probe_pin_all(rest...);
}
template <int ID, class M>
__device__ __forceinline__ void probe_stamp(const M& m) {
    asm volatile("; GB_PROBE_SITE %0" ::"n"(ID));
    if (m.trace) {
        const u64 t = __builtin_amdgcn_s_memtime();
        if ((threadIdx.x & 63) == 0) m.trace[m.pidx] = make_ulonglong2(t, (u64)ID);
        m.pidx++;
    }
}
}  // namespace poseidon_gl
// wave time stamp (s_memtime, shader clock) + site ID into the wave's trace.  The named values pass through an empty asm as in-out
// operands, so that the vector work in front of the site is in front of it in the instruction stream and the work behind it behind
#define GB_PROBE_AT(amat, ID, ...)                   \
    do {                                             \
        poseidon_gl::probe_pin_all(__VA_ARGS__);     \
        poseidon_gl::probe_stamp<ID>(amat);          \
    } while (0)
#else
#define GB_LAB_PROBE_FIELDS
#define GB_LAB_DEFINE_PROBE_SETUP
#define GB_LAB_PROBE_INIT(amat) do {} while (0)
#define GB_PROBE_AT(amat, ID, ...) do {} while (0)
#endif
