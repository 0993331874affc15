// Poseidon-12 Merkle-tree kernels for gfx950: replaces MerkleTree::new's per-leaf sponge and
// recursive fill_subtree (hash/merkle_tree.rs:86-181) with a lane-per-leaf / lane-per-node grid.
//
// Input is the column-major LDE matrix (column stride = number of leaves), so a wave's 64 lanes
// read 64 consecutive leaves of one column: fully coalesced, no transpose.
// Digests are kept LEVEL-MAJOR on the device: level 0 = leaf digests, level k = N >> k nodes,
// 4 x u64 per digest.  The reference's interleaved layout is produced on request only.
#include "kernels.hpp"
#include "poseidon_gl_grouped.hpp"
#include "poseidon_gl_coop.hpp"

namespace gbk {

using poseidon_gl::from_mont;
using poseidon_gl::mds_mfma_matrix;
using poseidon_gl::to_mont;
// The lane-per-state kernels run the full rounds' MDS layers on the matrix pipe (poseidon_gl.hpp, mds_layer_mfma): an MFMA is a
// wave-wide instruction, so no lane leaves early - lanes past the end work on a clamped index and skip the store.  Waves per SIMD:
// the MFMA form holds 16-register result tiles next to the state - four (measured against three and five, HISTORY.md rounds 3-5).
static constexpr int POSEIDON_OCC = 4;
// Round 4: the 22 partial rounds run in GROUPS of four (poseidon_gl_grouped.hpp): one cut into byte planes and one recombination
// per group instead of per round, the group's matrix M^4 cut into four byte planes on the matrix pipe, the words the group's later
// s-boxes see as VALU dot products - 11.3 k VALU instructions per permutation where 30 single layers took 14.0 k.
// The group operands live in LDS: every kernel fills its workgroup's table first (GB_POSEIDON_OPS, one barrier).
#define GB_POSEIDON_OPS()                                            \
    __shared__ poseidon_gl::v4i gops_lds[poseidon_gl::GROUP_LDS_V4]; \
    poseidon_gl::group_ops_init(gops_lds);                           \
    const poseidon_gl::v4i* gops = gops_lds + (threadIdx.x & 63)
#define permute_mont_mfma(s, amat, cap_only, zero_cap) poseidon_gl::permute_mont_mfma_grouped(s, amat, gops, cap_only, zero_cap)
// The sponge state of these kernels is kept in the permutation's Montgomery form (poseidon_gl.hpp): absorbed words go through
// to_mont, the digest through from_mont (canonical); the capacity words never leave that form between absorptions.

GB_LAB_DEFINE_PROBE_SETUP   // (lab builds: the attribution probes' setter, poseidon_gl_lab.hpp; nothing in the product)

// hash/hashing.rs:100-123 (overwrite-mode sponge, rate 8) + plonk/config.rs:70-84 (hash_or_noop)
__global__ __launch_bounds__(256, POSEIDON_OCC) void k_gl_merkle_leaves(const u64* __restrict__ cols, size_t col_stride, u32 width,
                                                                        u64 num_leaves, u64* __restrict__ out) {
    const u64 j0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = j0 < num_leaves;
    const u64 j = live ? j0 : num_leaves - 1;
    poseidon_gl::MdsOperand amat = mds_mfma_matrix();
    GB_LAB_PROBE_INIT(amat);   // (lab builds: tools/probe_leaves.py; nothing in the product)
    GB_POSEIDON_OPS();
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = 0;
    GB_PROBE_AT(amat, 1, s);   // kernel: start (operand table filled)
    if (width <= 4) {
        for (u32 c = 0; c < width; c++) s[c] = cols[(size_t)c * col_stride + j];
    } else {
        // one loop, one inlined copy of the permutation (two copies are 72 KB of code against a 64 KB instruction cache)
        for (u32 c0 = 0; c0 < width; c0 += 8) {
            if (c0 + 8 <= width) {
#pragma unroll
                for (int i = 0; i < 8; i++) s[i] = to_mont(cols[(size_t)(c0 + i) * col_stride + j]);
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++)
                    if (c0 + i < width) s[i] = to_mont(cols[(size_t)(c0 + i) * col_stride + j]);
            }
            GB_PROBE_AT(amat, 2, s);   // absorption: 8 column loads + to_mont
            // (a full absorption follows: words 0..7 of this permutation's output will be overwritten - only the capacity is produced)
            permute_mont_mfma(s, amat, c0 + 16 <= width, c0 == 0);  // the state stays a lazy Montgomery-form residue between absorptions; the first absorption meets a zero capacity
            GB_PROBE_AT(amat, 3, s);   // permutation: the last layer
        }
#pragma unroll
        for (int i = 0; i < 4; i++) s[i] = from_mont(s[i]);
    }
    if (!live) return;
    ulonglong2* o = reinterpret_cast<ulonglong2*>(out + 4 * j);
    o[0] = make_ulonglong2(s[0], s[1]);
    o[1] = make_ulonglong2(s[2], s[3]);
}

// The sponge of a leaf run in SEGMENTS of columns [c_begin, c_end): a commitment whose columns arrive over PCIe hashes the columns
// it already has while the rest is still in flight (api.hip commit()).  Between segments the words that the next absorption does
// not overwrite wait in `state` ([4 + 8 - keep_from][num_leaves], lazy residues): the capacity words 8..11 always, and the rate words the ragged
// last absorption leaves alone (`keep_from` .. 7) when the following segment starts with it.  FIRST: fresh sponge; LAST: the
// digest goes to `out`.  Every segment but the last absorbs whole groups of 8 columns.
template <bool FIRST, bool LAST>
__global__ __launch_bounds__(256, POSEIDON_OCC) void k_gl_merkle_leaves_seg(const u64* __restrict__ cols, size_t col_stride, u32 c_begin,
                                                                            u32 c_end, u64 num_leaves, u64* __restrict__ state,
                                                                            u32 keep_from, u64* __restrict__ out) {
    const u64 j0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = j0 < num_leaves;
    const u64 j = live ? j0 : num_leaves - 1;
    const poseidon_gl::MdsOperand amat = mds_mfma_matrix();
    GB_POSEIDON_OPS();
    u64 s[12];
#pragma unroll
    for (int i = 0; i < 8; i++) s[i] = 0;
    // `state` holds only the rows that are used: [0, 4) the capacity words 8..11, then rate word i >= keep_from at row 4 + i - keep_from
#pragma unroll
    for (int i = 8; i < 12; i++) s[i] = FIRST ? 0 : state[(size_t)(i - 8) * num_leaves + j];
    if (LAST && !FIRST && c_end - c_begin < 8) {  // this segment is the ragged absorption alone: the rate words it leaves alone
        const u32 kf = c_end - c_begin;           // = the keep_from the segment before was given
#pragma unroll
        for (int i = 1; i < 8; i++)
            if ((u32)i >= kf) s[i] = state[(size_t)(4 + i - kf) * num_leaves + j];
    }
    for (u32 c0 = c_begin; c0 < c_end; c0 += 8) {
        if (!LAST || c0 + 8 <= c_end) {
#pragma unroll
            for (int i = 0; i < 8; i++) s[i] = to_mont(cols[(size_t)(c0 + i) * col_stride + j]);
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (c0 + i < c_end) s[i] = to_mont(cols[(size_t)(c0 + i) * col_stride + j]);
        }
        // is the absorption that follows (in this segment or at the head of the next) a full one?  then only the capacity matters
        const bool next_full = LAST ? c0 + 16 <= c_end : (c0 + 8 < c_end || keep_from == 8);
        permute_mont_mfma(s, amat, next_full, FIRST && c0 == c_begin);
    }
    if (!live) return;
    if (!LAST) {
#pragma unroll
        for (int i = 8; i < 12; i++) state[(size_t)(i - 8) * num_leaves + j] = s[i];
        if (keep_from < 8) {
#pragma unroll
            for (int i = 1; i < 8; i++)
                if ((u32)i >= keep_from) state[(size_t)(4 + i - keep_from) * num_leaves + j] = s[i];
        }
        return;
    }
    ulonglong2* o = reinterpret_cast<ulonglong2*>(out + 4 * j);
    o[0] = make_ulonglong2(from_mont(s[0]), from_mont(s[1]));
    o[1] = make_ulonglong2(from_mont(s[2]), from_mont(s[3]));
}

// hash/hashing.rs:76-96 compress / Hasher::two_to_one
__global__ __launch_bounds__(256, POSEIDON_OCC) void k_gl_merkle_level(const u64* __restrict__ in, u64* __restrict__ out, u64 num_out) {
    const u64 i0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i0 < num_out;
    const u64 i = live ? i0 : num_out - 1;
    const poseidon_gl::MdsOperand amat = mds_mfma_matrix();
    GB_POSEIDON_OPS();
    const ulonglong2* p = reinterpret_cast<const ulonglong2*>(in + 8 * i);
    ulonglong2 a = p[0], b = p[1], c = p[2], d = p[3];
    u64 s[12] = {to_mont(a.x), to_mont(a.y), to_mont(b.x), to_mont(b.y), to_mont(c.x), to_mont(c.y), to_mont(d.x), to_mont(d.y),
                 0, 0, 0, 0};
    permute_mont_mfma(s, amat, false, true);   // two_to_one: zero capacity
    if (!live) return;
    ulonglong2* o = reinterpret_cast<ulonglong2*>(out + 4 * i);
    o[0] = make_ulonglong2(from_mont(s[0]), from_mont(s[1]));
    o[1] = make_ulonglong2(from_mont(s[2]), from_mont(s[3]));
}

// The same two kernels with one state per 16-lane row (poseidon_gl_coop.hpp), for trees too small to fill the machine with
// one permutation per lane.  Blocks are one wave = four states; nothing returns before the last __syncthreads().
__global__ __launch_bounds__(64) void k_gl_merkle_level_coop(const u64* __restrict__ in, u64* __restrict__ out, u64 num_out) {
    __shared__ u64 sh[64];
    const u32 l = threadIdx.x & 15, row = threadIdx.x >> 4;
    const u64 node = (u64)blockIdx.x * 4 + row;
    const bool valid = node < num_out;
    u64 x = l < 8 ? in[8 * (valid ? node : 0) + l] : 0;
    x = poseidon_gl_coop::permute(x, l, sh + 16 * row);
    if (valid && l < 4) out[4 * node + l] = gl::canon(x);
}
// width > 4 (narrower leaves are not hashed, plonk/config.rs:70-84)
__global__ __launch_bounds__(64) void k_gl_merkle_leaves_coop(const u64* __restrict__ cols, size_t col_stride, u32 width,
                                                              u64 num_leaves, u64* __restrict__ out) {
    __shared__ u64 sh[64];
    const u32 l = threadIdx.x & 15, row = threadIdx.x >> 4;
    const u64 leaf = (u64)blockIdx.x * 4 + row;
    const bool valid = leaf < num_leaves;
    const u64 j = valid ? leaf : 0;
    u64 x = 0;
    for (u32 c0 = 0; c0 < width; c0 += 8) {
        if (l < 8 && c0 + l < width) x = cols[(size_t)(c0 + l) * col_stride + j];  // overwrite-mode absorption
        x = poseidon_gl_coop::permute(x, l, sh + 16 * row);
    }
    if (valid && l < 4) out[4 * leaf + l] = gl::canon(x);
}

// FRI layer leaves (fri/prover.rs:101-107): leaf m = the 2^arity_bits extension values of coset m, flattened (D = 2);
// vals = [2][len] coordinate columns in leaf order.  2 * arity > 4 only (narrower leaves are not hashed).
__global__ __launch_bounds__(64) void k_gl_fri_leaves_coop(const u64* __restrict__ vals, size_t len, u32 arity_bits, u64 num_leaves,
                                                           u64* __restrict__ out) {
    __shared__ u64 sh[64];
    const u32 l = threadIdx.x & 15, row = threadIdx.x >> 4;
    const u64 leaf = (u64)blockIdx.x * 4 + row;
    const bool valid = leaf < num_leaves;
    const u64* a = vals + ((valid ? leaf : 0) << arity_bits);
    const u32 arity = 1u << arity_bits;
    u64 x = 0;
    for (u32 k0 = 0; k0 < arity; k0 += 4) {   // four extension elements = eight base elements per absorption
        const u32 k = k0 + (l >> 1);
        if (l < 8 && k < arity) x = a[(size_t)(l & 1) * len + k];
        x = poseidon_gl_coop::permute(x, l, sh + 16 * row);
    }
    if (valid && l < 4) out[4 * leaf + l] = gl::canon(x);
}

// Level k node t (t < N>>k) sits, in the reference layout, inside subtree s = t >> (layers-k) at
// pair p = (t & mask) >> 1 of layer k:  2*((p << (k+1)) + (1<<k) - 1) + (t&1)   (merkle_tree.rs:200-217)
__global__ void k_gl_digests_to_reference(const u64* __restrict__ levels, u64* __restrict__ out, u32 log_leaves,
                                          u32 cap_height) {
    u32 layers = log_leaves - cap_height;
    u64 total = 2 * (((u64)1 << log_leaves) - ((u64)1 << cap_height));
    u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    // find level: level k occupies [off_k, off_k + N>>k), off_k = 2N - (2N >> k)
    u64 N = (u64)1 << log_leaves;
    u32 k = 0;
    u64 off = 0;
    while (g >= off + (N >> k)) {
        off += N >> k;
        k++;
    }
    u64 t = g - off;
    u64 per_tree = (u64)1 << (layers - k);
    u64 tree = t / per_tree, w = t % per_tree;
    u64 p = w >> 1;
    u64 tree_len = total >> cap_height;
    u64 dst = tree * tree_len + 2 * ((p << (k + 1)) + ((u64)1 << k) - 1) + (w & 1);
#pragma unroll
    for (int e = 0; e < 4; e++) out[4 * dst + e] = levels[4 * g + e];
}

__global__ void k_gl_gather_row(const u64* __restrict__ cols, size_t col_stride, u32 width, u64 index, u64* dst) {
    u32 c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c < width) dst[c] = cols[(size_t)c * col_stride + index];
}

__global__ void k_gl_gather_siblings(const u64* __restrict__ levels, u32 log_leaves, u32 cap_height, u64 leaf, u64* dst) {
    u32 i = threadIdx.x >> 2, e = threadIdx.x & 3;
    if (i >= log_leaves - cap_height) return;
    u64 N = (u64)1 << log_leaves;
    u64 off = 2 * N - ((2 * N) >> i);
    dst[4 * i + e] = levels[4 * (off + ((leaf >> i) ^ 1)) + e];
}

__global__ void k_u64_bitrev_copy(const u64* __restrict__ src, u64* __restrict__ dst, u32 bits, size_t total) {
    size_t g = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    size_t col = g >> bits;
    u64 j = g & (((u64)1 << bits) - 1);
    u64 r = bits ? (__brevll(j) >> (64 - bits)) : 0;
    dst[g] = src[(col << bits) + r];
}

__global__ void k_u64_transpose_to_rows(const u64* __restrict__ cols, size_t col_stride, u32 width, u64 rows,
                                        u64* __restrict__ dst) {
    u64 g = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= rows * width) return;
    u64 r = g / width;
    u32 c = (u32)(g % width);
    dst[g] = cols[(size_t)c * col_stride + r];
}

// raw permutation, canonical in and out, through the same MFMA form the tree kernels use (gb_permute: the reference's KATs)
__global__ __launch_bounds__(256, POSEIDON_OCC) void k_gl_poseidon_permute(const u64* __restrict__ in, u64* __restrict__ out, u64 count) {
    const u64 i0 = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i0 < count;
    const u64 i = live ? i0 : count - 1;
    const poseidon_gl::MdsOperand amat = mds_mfma_matrix();
    GB_POSEIDON_OPS();
    u64 s[12];
#pragma unroll
    for (int e = 0; e < 12; e++) s[e] = to_mont(in[12 * i + e]);
    permute_mont_mfma(s, amat, false, false);
    if (!live) return;
#pragma unroll
    for (int e = 0; e < 12; e++) out[12 * i + e] = from_mont(s[e]);
}

static inline u32 blocks_for(u64 n, u32 bs) { return (u32)((n + bs - 1) / bs); }

// Below this many states a lone wave's ~56 us per lane-per-state permutation dominates; the cooperative form does ~3.6x the
// instructions in total but its dependent chain is ~4x shorter (crossover between 2^14 and 2^15 states on 1024 SIMDs).
static constexpr u64 COOP_MAX_STATES = 16384;  // measured: 2048 / 8192 / 16384 / 32768 -> 5.41 / 5.14 / 5.00 / 5.50 ms per 2^12-row proof;
                                               // separate thresholds for levels and leaves (same-box sweeps, both fields) are within noise of this

void gl_merkle_leaves(const u64* cols, size_t col_stride, u32 width, u64 num_leaves, u64* out, hipStream_t stream) {
    if (width > 4 && num_leaves <= COOP_MAX_STATES) {
        hipLaunchKernelGGL(k_gl_merkle_leaves_coop, dim3(blocks_for(num_leaves, 4)), dim3(64), 0, stream, cols, col_stride, width,
                           num_leaves, out);
        return;
    }
    hipLaunchKernelGGL(k_gl_merkle_leaves, dim3(blocks_for(num_leaves, 256)), dim3(256), 0, stream, cols, col_stride,
                       width, num_leaves, out);
}
// one segment of the leaf sponges (k_gl_merkle_leaves_seg): columns [c_begin, c_end), c_begin a multiple of 8; `last`: c_end is the
// leaf width and the digests are written; next_cols: how many columns remain after this segment (the following segment's first
// absorption takes min(8, next_cols) of them)
void gl_merkle_leaves_segment(const u64* cols, size_t col_stride, u32 c_begin, u32 c_end, u64 num_leaves, u64* state, bool last,
                              u32 next_cols, u64* out, hipStream_t stream) {
    const dim3 grid(blocks_for(num_leaves, 256)), block(256);
    const u32 keep_from = next_cols < 8u ? next_cols : 8u;
    if (c_begin == 0 && !last)
        hipLaunchKernelGGL((k_gl_merkle_leaves_seg<true, false>), grid, block, 0, stream, cols, col_stride, c_begin, c_end, num_leaves, state, keep_from, out);
    else if (!last)
        hipLaunchKernelGGL((k_gl_merkle_leaves_seg<false, false>), grid, block, 0, stream, cols, col_stride, c_begin, c_end, num_leaves, state, keep_from, out);
    else
        hipLaunchKernelGGL((k_gl_merkle_leaves_seg<false, true>), grid, block, 0, stream, cols, col_stride, c_begin, c_end, num_leaves, state, keep_from, out);
}
bool gl_fri_leaves_coop(const u64* vals, size_t len, u32 arity_bits, u64 num_leaves, u64* out, hipStream_t stream) {
    if (num_leaves > COOP_MAX_STATES || (2u << arity_bits) <= 4) return false;
    hipLaunchKernelGGL(k_gl_fri_leaves_coop, dim3(blocks_for(num_leaves, 4)), dim3(64), 0, stream, vals, len, arity_bits, num_leaves, out);
    return true;
}
void gl_merkle_level(const u64* in, u64* out, u64 num_out, hipStream_t stream) {
    if (num_out <= COOP_MAX_STATES) {
        hipLaunchKernelGGL(k_gl_merkle_level_coop, dim3(blocks_for(num_out, 4)), dim3(64), 0, stream, in, out, num_out);
        return;
    }
    hipLaunchKernelGGL(k_gl_merkle_level, dim3(blocks_for(num_out, 256)), dim3(256), 0, stream, in, out, num_out);
}
void gl_digests_to_reference_layout(const u64* levels, u64* out, u32 log_leaves, u32 cap_height, hipStream_t stream) {
    u64 total = 2 * (((u64)1 << log_leaves) - ((u64)1 << cap_height));
    if (!total) return;
    hipLaunchKernelGGL(k_gl_digests_to_reference, dim3(blocks_for(total, 256)), dim3(256), 0, stream, levels, out,
                       log_leaves, cap_height);
}
void gl_gather_row(const u64* cols, size_t col_stride, u32 width, u64 index, u64* dst, hipStream_t stream) {
    hipLaunchKernelGGL(k_gl_gather_row, dim3(blocks_for(width, 64)), dim3(64), 0, stream, cols, col_stride, width, index,
                       dst);
}
void gl_gather_siblings(const u64* levels, u32 log_leaves, u32 cap_height, u64 leaf, u64* dst, hipStream_t stream) {
    if (log_leaves == cap_height) return;
    hipLaunchKernelGGL(k_gl_gather_siblings, dim3(1), dim3(256), 0, stream, levels, log_leaves, cap_height, leaf, dst);
}
void u64_bitrev_copy(const u64* src, u64* dst, u32 bits, size_t ncols, hipStream_t stream) {
    size_t total = ncols << bits;
    if (!total) return;
    hipLaunchKernelGGL(k_u64_bitrev_copy, dim3(blocks_for(total, 256)), dim3(256), 0, stream, src, dst, bits, total);
}
void u64_transpose_to_rows(const u64* cols, size_t col_stride, u32 width, u64 rows, u64* dst, hipStream_t stream) {
    if (!rows || !width) return;
    hipLaunchKernelGGL(k_u64_transpose_to_rows, dim3(blocks_for(rows * width, 256)), dim3(256), 0, stream, cols,
                       col_stride, width, rows, dst);
}
void gl_poseidon_permute(const u64* in, u64* out, u64 count, hipStream_t stream) {
    hipLaunchKernelGGL(k_gl_poseidon_permute, dim3(blocks_for(count, 256)), dim3(256), 0, stream, in, out, count);
}

}  // namespace gbk
