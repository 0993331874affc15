// Poseidon-12's partial rounds in GROUPS on the matrix pipe (gfx950), one permutation per lane.
//
// A partial round (hash/poseidon_goldilocks.rs:927-948) is non-linear in word 0 only, so G consecutive rounds are two constant
// integer matrices applied to the byte planes of the state: one that yields the words the G - 1 later s-boxes will see
// (phase A: rows (M^j)_0, j < G), and - once those s-boxes have run one after the other - one that yields the state G rounds
// on (phase B: M^G on the state, M^(G-j) e_0 on d_j = u_j^7 - u_j, which ride in the four spare K slots of the B operand).
// What mds_layer_mfma pays in VALU instructions per LAYER - 24 v_xor + 48 v_perm_b32 to cut the state into byte planes, 96 to
// recombine the plane sums, 24 to fold - is paid once per GROUP; the price is more MFMAs (M^G has G byte planes), on a pipe
// that was 14 % busy.  tools/gen_poseidon_groups.py builds the operands, the MFMA schedules and the accumulator start values
// (csrc/poseidon_gl_groups.h) and checks the construction against the defining permutation with an exact integer model.
//
// Operands live in LDS ([operand][lane], 16 bytes per lane: one conflict-free ds_read_b128 each); every lane of a wave must
// execute these functions (MFMA), whatever its data.
#pragma once
#include "poseidon_gl.hpp"
#include "poseidon_gl_groups.h"

namespace poseidon_gl {

namespace grp = poseidon_gl_groups;

// The plan: five groups of four partial rounds and one of two (G = 2 ... 4 with and without a remainder group, and phase A on the
// matrix pipe as well, were measured in round 4: HISTORY.md, profiles/r04_poseidon_group_ablations.txt).
static constexpr int GROUP_G = 4, GROUP_N = 5, GROUP_G2 = 2;
static_assert(GROUP_G * GROUP_N + GROUP_G2 == N_PARTIAL, "partial-round plan");
// operands a workgroup keeps in LDS: the phase-B planes of Shape<G>::ops() (the G - 1 phase-A planes in front of them are not loaded:
// the words the later s-boxes see are VALU dot products - row 0 of M^j, j <= 3, times twelve 32-bit halves fits 64 bits)
template <int G>
struct GroupOps {
    static constexpr int skip = G - 1;
    static constexpr int count = G;
};
static constexpr int GROUP_OPS_MAIN = GroupOps<GROUP_G>::count, GROUP_OPS_REM = GroupOps<GROUP_G2>::count;
static constexpr int GROUP_OPS_TOTAL = GROUP_OPS_MAIN + GROUP_OPS_REM;
static constexpr int GROUP_LDS_V4 = GROUP_OPS_TOTAL * 64;   // v4i entries a workgroup needs

// This lane's share of a constant 16 x 16 operand (rows = output registers q, columns = K slots): A row m = lane & 31 is output
// register q = (m & 3) + 4 (m >> 3) of the lanes with h = (m >> 2) & 1, and carries its matrix row in that k-block only (the
// block-diagonal form of mds_mfma_matrix, with all 16 registers and all 16 slots in use).
__device__ __forceinline__ v4i group_operand_share(const uint32_t (&op)[16][4], u32 lane) {
    const u32 r = lane & 31, h = lane >> 5;
    const u32 hp = (r >> 2) & 1, q = (r & 3) + 4 * (r >> 3);
    v4i a = {0, 0, 0, 0};
    if (h == hp) {
        a[0] = (int)op[q][0];
        a[1] = (int)op[q][1];
        a[2] = (int)op[q][2];
        a[3] = (int)op[q][3];
    }
    return a;
}
template <int G>
__device__ __forceinline__ void group_fill_ops(v4i* lds) {
    if constexpr (G >= 2) {
        for (u32 e = threadIdx.x; e < (u32)GroupOps<G>::count * 64; e += blockDim.x)
            lds[e] = group_operand_share(grp::Shape<G>::ops()[GroupOps<G>::skip + (e >> 6)], e & 63);
    }
}
// Fill the workgroup's operand table (call once, all threads; ends in a barrier).
__device__ __forceinline__ void group_ops_init(v4i* lds) {
    group_fill_ops<GROUP_G>(lds);
    group_fill_ops<GROUP_G2>(lds + GROUP_OPS_MAIN * 64);
    __syncthreads();
}

// (lo, hi) -> lo + 2^32 hi as a lazy residue, the carry fix on a rare wave-uniform path (see mds_layer_mfma)
template <int N>
__device__ __forceinline__ void fold_rows_rare_carry(u64 (&s)[N], const long long (&lo)[N], const long long (&hi)[N]) {
    u64 any_carry = 0;
#pragma unroll
    for (int q = 0; q < N; q++) {
        const u64 sl = (u64)lo[q], sh = (u64)hi[q];
        const u64 t = sl + (u64)(u32)(sh >> 32) * EPS;
        u32 r1;
        u64 carry_lanes;
        asm("v_add_co_u32 %0, %1, %2, %3" : "=v"(r1), "=s"(carry_lanes) : "v"((u32)(t >> 32)), "v"((u32)sh));
        any_carry |= carry_lanes;
        s[q] = (u64)(u32)t | ((u64)r1 << 32);
    }
    if (__builtin_expect(any_carry != 0, 0)) {
#pragma unroll
        for (int q = 0; q < N; q++) {
            const bool wrapped = (u32)(s[q] >> 32) < (u32)hi[q];
            s[q] += wrapped ? EPS : 0;
        }
    }
}

// a - b for lazy residues (any u64 in, any u64 out): every borrow is worth -EPS, and the second fix cannot borrow again
__device__ __forceinline__ u64 sub_lazy(u64 a, u64 b) {
    u64 d;
    const bool b1 = __builtin_usubll_overflow(a, b, &d);
    u64 e;
    const bool b2 = __builtin_usubll_overflow(d, b1 ? EPS : 0, &e);
    return e - (b2 ? EPS : 0);
}

// One output plane: the MFMAs of its schedule chained through the accumulator operand.  (Template recursion, not a loop: the
// schedule entries must be constants where the plane arrays are indexed, or the arrays end up in scratch memory.)
template <int G, bool PHASE_B, int PLANE, int E>
__device__ __forceinline__ void group_chain(v16i& d, const u32 (&pl)[8][4], const u32 (&cpl)[8][4], const v4i* __restrict__ ops) {
    using S = grp::Shape<G>;
    constexpr int n = PHASE_B ? S::LEN_B[PLANE] : S::LEN_A[PLANE];
    if constexpr (E < n) {
        constexpr grp::Mfma m = PHASE_B ? S::SCHED_B[PLANE][E] : S::SCHED_A[PLANE][E];
        const v4i a = ops[m.k * 64];
        v4i b;
        if constexpr (m.comp != 0) {
            b[0] = (int)cpl[m.p][0]; b[1] = (int)cpl[m.p][1]; b[2] = (int)cpl[m.p][2]; b[3] = (int)cpl[m.p][3];
        } else {
            b[0] = (int)pl[m.p][0]; b[1] = (int)pl[m.p][1]; b[2] = (int)pl[m.p][2]; b[3] = (int)pl[m.p][3];
        }
        d = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, d, 0, 0, 0);
        group_chain<G, PHASE_B, PLANE, E + 1>(d, pl, cpl, ops);
    }
}
template <int G, bool PHASE_B, int PLANE>
__device__ __forceinline__ v16i group_plane(const u32 (&pl)[8][4], const u32 (&cpl)[8][4], const v4i* __restrict__ ops) {
    v16i d = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    group_chain<G, PHASE_B, PLANE, 0>(d, pl, cpl, ops);
    return d;
}


// G partial rounds starting with round r0 (uniform): the state comes in with round r0's constants added and leaves with those
// of round r0 + G added.  `ops` = this lane's column of the workgroup's operand table, at the group shape's first operand.
template <int G>
__device__ __forceinline__ void partial_group(u64 (&s)[12], const MdsOperand& amat, const v4i* __restrict__ ops, int r0) {
    using S = grp::Shape<G>;
    const grp::GroupInit& init = S::init();
    const int gi = r0 - HALF_FULL;
    s[0] = sbox(s[0]);
    GB_PROBE_AT(amat, 30, s);   // group: the first s-box
    // byte planes of the state: pl[p][w] = byte p of words 4w .. 4w+3 (signed: ^ 0x80); dword 3 = the d_j, not known yet
    u32 pl[8][4], cpl[8][4];
    {
        u32 w[24];
#pragma unroll
        for (int i = 0; i < 12; i++) {
            w[i] = (u32)s[i] ^ 0x80808080u;
            w[12 + i] = (u32)(s[i] >> 32) ^ 0x80808080u;
        }
#pragma unroll
        for (int half = 0; half < 2; half++)
#pragma unroll
            for (int g = 0; g < 3; g++) {
                u32 t[4];
                byte_transpose4(w[12 * half + 4 * g], w[12 * half + 4 * g + 1], w[12 * half + 4 * g + 2], w[12 * half + 4 * g + 3], t);
#pragma unroll
                for (int p = 0; p < 4; p++) pl[4 * half + p][g] = t[p];
            }
#pragma unroll
        for (int p = 0; p < 8; p++) pl[p][3] = 0x80808080u;   // phase A's operands are zero in these slots; a defined register all the same
    }
#pragma unroll
    for (int p = 0; p < 8; p++)
#pragma unroll
        for (int w = 0; w < 4; w++) cpl[p][w] = ~pl[p][w];   // only the planes a schedule complements are ever materialised
    GB_PROBE_AT(amat, 31, pl);   // group: byte planes of the state (cut + complements)
    // ---- phase A: the words u_1 .. u_(G-1) before the d_i terms
    u64 ulo[G - 1], uhi[G - 1];
    // on the VALU: row 0 of M^j has entries below 2^(8j - 3), so the two 32-bit halves of the twelve words accumulate unreduced in
    // 64 bits - 24 v_mad_u64_u32 per word, against 8 j MFMAs whose pipe time the kernel does not hide (profiles/r04_poseidon_groups.txt)
#pragma unroll
    for (int j = 1; j < G; j++) {
        const u64 k = grp::KU[gi][j - 1];
        u64 lo = (u32)k, hi = k >> 32;
#pragma unroll
        for (int i = 0; i < 12; i++) {
            lo += (u64)(u32)s[i] * grp::ROW0[j][i];
            hi += (u64)(u32)(s[i] >> 32) * grp::ROW0[j][i];
        }
        ulo[j - 1] = lo;
        uhi[j - 1] = hi;
    }
    GB_PROBE_AT(amat, 32, ulo, uhi);   // group: phase A dot products
    // ---- the s-boxes of rounds r0 + 1 .. r0 + G - 1, one after the other
    u64 dl[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 1; j < G; j++) {
        u64 lo = ulo[j - 1], hi = uhi[j - 1];
#pragma unroll
        for (int i = 1; i < j; i++) {   // d_i (M^(j-i))_00, unreduced: < 2^32 * 2^29 on top of < 2^46
            lo += (u64)(u32)dl[i - 1] * S::TRI[j - i];
            hi += (u64)(u32)(dl[i - 1] >> 32) * S::TRI[j - i];
        }
        const u64 u = fold_halves(lo, hi);
        dl[j - 1] = sub_lazy(sbox(u), u);
    }
    GB_PROBE_AT(amat, 33, dl);   // group: the G - 1 dependent s-boxes
    // ---- the d_j's byte planes into dword 3 of the B operands
    if constexpr (G == 2) {   // one word: byte p of it in slot 12, the slots beside it meet zeros
#pragma unroll
        for (int half = 0; half < 2; half++) {
            const u32 w0 = (u32)(dl[0] >> (32 * half)) ^ 0x80808080u;
#pragma unroll
            for (int p = 0; p < 4; p++) pl[4 * half + p][3] = w0 >> (8 * p);
        }
    } else {
#pragma unroll
        for (int half = 0; half < 2; half++) {
            u32 t[4];
            byte_transpose4((u32)(dl[0] >> (32 * half)) ^ 0x80808080u, (u32)(dl[1] >> (32 * half)) ^ 0x80808080u,
                            (u32)(dl[2] >> (32 * half)) ^ 0x80808080u, (u32)(dl[3] >> (32 * half)) ^ 0x80808080u, t);
#pragma unroll
            for (int p = 0; p < 4; p++) pl[4 * half + p][3] = t[p];
        }
    }
#pragma unroll
    for (int p = 0; p < 8; p++) cpl[p][3] = ~pl[p][3];
    GB_PROBE_AT(amat, 34, pl);   // group: the d_j planes
    // ---- phase B: the state G rounds on
    long long lo[12], hi[12];
    {
        const v16i d0 = group_plane<G, true, 0>(pl, cpl, ops), d1 = group_plane<G, true, 1>(pl, cpl, ops);
#pragma unroll
        for (int q = 0; q < 12; q++) lo[q] = mad_i64_start((int)(((u32)d1[q] << 8) + (u32)d0[q]), amat, init.olo[gi][q]);
    }
    {
        const v16i d0 = group_plane<G, true, 2>(pl, cpl, ops), d1 = group_plane<G, true, 3>(pl, cpl, ops);
#pragma unroll
        for (int q = 0; q < 12; q++) lo[q] = mad_i64((int)(((u32)d1[q] << 8) + (u32)d0[q]), amat, lo[q]);
    }
    {
        const v16i d0 = group_plane<G, true, 4>(pl, cpl, ops), d1 = group_plane<G, true, 5>(pl, cpl, ops);
#pragma unroll
        for (int q = 0; q < 12; q++) hi[q] = mad_i64_start((int)(((u32)d1[q] << 8) + (u32)d0[q]), amat, init.ohi[gi][q]);
    }
    {
        const v16i d0 = group_plane<G, true, 6>(pl, cpl, ops), d1 = group_plane<G, true, 7>(pl, cpl, ops);
#pragma unroll
        for (int q = 0; q < 12; q++) hi[q] = mad_i64((int)(((u32)d1[q] << 8) + (u32)d0[q]), amat, hi[q]);
    }
    GB_PROBE_AT(amat, 35, lo, hi);   // group: phase B MFMA chains + recombination
    fold_rows_rare_carry(s, lo, hi);
    GB_PROBE_AT(amat, 36, s);   // group: fold
}

// The permutation with every MDS layer on the matrix pipe and the partial rounds in groups; same contract as
// permute_mont_mfma_naive (Montgomery-form lazy residues in and out).  `amat` = mds_mfma_matrix(), `ops` = the workgroup's operand
// table (group_ops_init) offset by this thread's lane.  capacity_only: see mds_layer_mfma<Q0>.  zero_capacity (uniform): words
// 8..11 of the state are zero on entry - their first s-boxes are the constants ZERO_CAP_SBOX_T and are not computed.
__device__ __forceinline__ void permute_mont_mfma_grouped(u64 (&s)[12], const MdsOperand& amat, const v4i* __restrict__ ops, bool capacity_only = false,
                                                          bool zero_capacity = false) {
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = add_rc(s[i], GB_RC[i]);
    GB_PROBE_AT(amat, 10, s);   // permutation: first constants
    for (int r = 0; r < HALF_FULL; r++) {
#pragma unroll
        for (int i = 0; i < 8; i++) s[i] = sbox(s[i]);
        if (r == 0 && zero_capacity) {
#pragma unroll
            for (int i = 0; i < 4; i++) s[8 + i] = ZERO_CAP_SBOX_T.v[i];
        } else {
#pragma unroll
            for (int i = 8; i < 12; i++) s[i] = sbox(s[i]);
        }
        GB_PROBE_AT(amat, 11, s);   // full round: 12 s-boxes
        mds_layer_mfma(s, amat, r + 1);
        GB_PROBE_AT(amat, 22, s);   // layer: fold
    }
    for (int gidx = 0; gidx < GROUP_N; gidx++) partial_group<GROUP_G>(s, amat, ops, HALF_FULL + GROUP_G * gidx);
    if constexpr (GROUP_G2 >= 2) partial_group<GROUP_G2>(s, amat, ops + GROUP_OPS_MAIN * 64, HALF_FULL + GROUP_G * GROUP_N);
    for (int r = HALF_FULL + N_PARTIAL; r + 1 < 2 * HALF_FULL + N_PARTIAL; r++) {
#pragma unroll
        for (int i = 0; i < 12; i++) s[i] = sbox(s[i]);
        GB_PROBE_AT(amat, 11, s);
        mds_layer_mfma(s, amat, r + 1);
        GB_PROBE_AT(amat, 22, s);
    }
#pragma unroll
    for (int i = 0; i < 12; i++) s[i] = sbox(s[i]);
    GB_PROBE_AT(amat, 11, s);
    // capacity_only (uniform): a full absorption follows and overwrites words 0..7 - only words 8..11 of the output are produced
    if (capacity_only) mds_layer_mfma<8>(s, amat, MFMA_NO_RC);
    else mds_layer_mfma<0>(s, amat, MFMA_NO_RC);
}

}  // namespace poseidon_gl
