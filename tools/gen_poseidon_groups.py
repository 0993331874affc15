#!/usr/bin/env python3
"""Generate plonky2_goldibear_amd/csrc/poseidon_gl_groups.h: the constant operands, MFMA schedules and start values of the
GROUPED partial rounds of Poseidon-12 (csrc/poseidon_gl.hpp, partial_group<G>), and check the construction with an exact integer
model of the device pipeline against the defining (naive) permutation.

The reference's partial round (hash/poseidon_goldilocks.rs:927-948, output-identical to the fast form, :1196-1198) is
    s <- s + rc_r;  s_0 <- s_0^7;  s <- M s
with M = circ(MDS_MATRIX_CIRC) + diag(MDS_MATRIX_DIAG) (:301-302, :547-557).  Only word 0 is non-linear, so G consecutive rounds
starting from y (= s with word 0 already through the s-box) are
    u_j = (M^j y)_0 + K_j + sum_{i<j} d_i (M^(j-i))_00        the word the j-th s-box sees            (j = 1 .. G-1)
    d_j = u_j^7 - u_j
    out = M^G y + sum_j d_j M^(G-j) e_0 + Kout                the state G rounds later
and on the matrix pipe both are products of CONSTANT integer matrices with the byte planes of y (and of d_1 .. d_(G-1)), whose
VALU cost - cutting the state into byte planes, recombining the plane sums - is paid once per group instead of once per round.
M^G has entries up to 2^(8G - 3.5): it is cut into G signed byte planes A_k (balanced digits), data plane p times matrix plane k
lands in output plane p + k, MFMAs into the same output plane chain through the accumulator operand, and output planes 8 and up
wrap with 2^64 = 2^32 - 1 (mod p): one MFMA into plane P - 4 and one, on the complemented bytes, into plane P - 8.

Everything here is integer arithmetic on numbers the product already holds (csrc/poseidon_constants.h); nothing is read from the
reference or from oracle/.

  python3 tools/gen_poseidon_groups.py            # check the model, write the header
  python3 tools/gen_poseidon_groups.py --check    # check only (tests/test_poseidon_groups_model.py)
"""
import os
import random
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HDR = os.path.join(ROOT, "plonky2_goldibear_amd", "csrc", "poseidon_constants.h")
OUT = os.path.join(ROOT, "plonky2_goldibear_amd", "csrc", "poseidon_gl_groups.h")

P = 0xFFFFFFFF00000001
R = (1 << 64) % P            # Montgomery radix of the device's state words
N_FULL_HALF, N_PARTIAL = 4, 22
GROUP_SIZES = (2, 4)   # group shapes the header carries: the product's plan is five groups of four and one of two (3 and 5 were measured in round 4)
MAXD = 4                     # K slots 12..15 of the B operand hold d_1 .. d_4


def grab(name):
    text = open(HDR).read()
    m = re.search(r"#define " + name + r"_LIST \\\n((?:.*\\\n)*.*)\n", text)
    return [int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\b\d+\b", re.sub(r"ULL|u\b", "", m.group(1)))]


CIRC = grab("GL_POSEIDON_MDS_CIRC")
DIAG = grab("GL_POSEIDON_MDS_DIAG")
RC = grab("GL_POSEIDON_ALL_ROUND_CONSTANTS")
M1 = [[CIRC[(i - q) % 12] + (DIAG[q] if i == q else 0) for i in range(12)] for q in range(12)]


def matmul(a, b):
    return [[sum(a[i][k] * b[k][j] for k in range(12)) for j in range(12)] for i in range(12)]


def mpow(e):
    r = [[int(i == j) for j in range(12)] for i in range(12)]
    for _ in range(e):
        r = matmul(r, M1)
    return r


def digits(v, n):
    """balanced base-256 digits, little endian, each in [-128, 127]"""
    d = []
    for _ in range(n):
        r = ((v + 128) % 256) - 128
        d.append(r)
        v = (v - r) // 256
    assert v == 0
    return d


# ------------------------------------------------------------------ the defining permutation (plain residues)
def sbox(x):
    return pow(x, 7, P)


def permute_naive(s):
    s = list(s)
    for r in range(2 * N_FULL_HALF + N_PARTIAL):
        s = [(s[i] + RC[12 * r + i]) % P for i in range(12)]
        if r < N_FULL_HALF or r >= N_FULL_HALF + N_PARTIAL:
            s = [sbox(x) for x in s]
        else:
            s[0] = sbox(s[0])
        s = [sum(M1[q][i] * s[i] for i in range(12)) % P for q in range(12)]
    return s


# ------------------------------------------------------------------ a group shape: operands and schedules
class Shape:
    """Operand tables and MFMA schedules of a group of g partial rounds.  An operand is a 16 x 16 signed-byte matrix
    A[q][slot]: q = the output register of the lane's D tile, slot = the K slot of the lane's B operand (0..11 the state words,
    12..15 d_1..d_4).  A schedule is, per output plane P, the list of (data plane p, operand, complemented) MFMAs chained into it."""

    def __init__(self, g):
        self.g = g
        pw = [mpow(e) for e in range(g + 1)]
        # phase A: row j-1 = (M^j)_0 . y  for j = 1 .. g-1; planes k = 0 .. g-2
        self.nA = max(g - 1, 1) if g > 1 else 0
        self.opsA = []
        for k in range(g - 1):
            a = [[0] * 16 for _ in range(16)]
            for j in range(1, g):
                for i in range(12):
                    dg = digits(pw[j][0][i], g - 1)
                    a[j - 1][i] = dg[k]
            self.opsA.append(a)
        # phase B: rows 0..11 = M^g y + sum_j d_j M^(g-j) e_0; planes k = 0 .. g-1
        self.opsB = []
        for k in range(g):
            a = [[0] * 16 for _ in range(16)]
            for q in range(12):
                for i in range(12):
                    a[q][i] = digits(pw[g][q][i], g)[k]
                for j in range(1, g):
                    a[q][12 + j - 1] = digits(pw[g - j][q][0], g)[k]
            self.opsB.append(a)
        self.schedA = self.schedule(g - 1)
        self.schedB = self.schedule(g)
        self.tri = [pw[k][0][0] for k in range(g)]   # tri[k] = (M^k)_00, k >= 1 used
        self.pw = pw

    @staticmethod
    def schedule(nplanes):
        sched = [[] for _ in range(8)]
        for k in range(nplanes):
            for p in range(8):
                if p + k < 8:
                    sched[p + k].append((p, k, 0))
                else:                       # 2^(8 (p + k)) = 2^(8 (p + k - 8)) (2^32 - 1)
                    w = p + k - 8
                    sched[w + 4].append((p, k, 0))
                    sched[w].append((p, k, 1))
        return sched

    def n_mfma(self):
        return sum(len(x) for x in self.schedA), sum(len(x) for x in self.schedB)

    def bound(self, ops, sched):
        """max |lo raw|, |hi raw| over all inputs (every signed byte within [-128, 127])"""
        worst = 0
        for half in range(2):
            for q in range(16):
                tot = 0
                for pp in range(4):
                    P8 = 4 * half + pp
                    tot += (sum(sum(abs(x) for x in ops[k][q]) * 128 for (_, k, _) in sched[P8])) << (8 * pp)
                worst = max(worst, tot)
        return worst


def signed_bytes(words16, p, comp):
    out = []
    for w in words16:
        b = (w >> (8 * p)) & 0xFF
        v = (b ^ (0x7F if comp else 0x80))
        out.append(v - 256 if v >= 128 else v)
    return out


def run_phase(ops, sched, words16):
    """the MFMAs and the recombination, exactly: returns (lo_raw[16], hi_raw[16]) as Python integers"""
    lo, hi = [0] * 16, [0] * 16
    for P8 in range(8):
        d = [0] * 16
        for (p, k, comp) in sched[P8]:
            b = signed_bytes(words16, p, comp)
            for q in range(16):
                d[q] += sum(ops[k][q][j] * b[j] for j in range(16))
        for q in range(16):
            assert -(1 << 31) <= d[q] < (1 << 31)
            if P8 < 4:
                lo[q] += d[q] << (8 * P8)
            else:
                hi[q] += d[q] << (8 * (P8 - 4))
    return lo, hi


def offsets(ops, sched):
    """what the pipeline returns for the all-zero input (the signed-byte offsets and the complement's -1s), mod p"""
    lo, hi = run_phase(ops, sched, [0] * 16)
    return [(lo[q] + (hi[q] << 32)) % P for q in range(16)]


class Group:
    """One group of the permutation: shape + the constants that depend on its first round r0."""

    def __init__(self, shape, r0):
        g = shape.g
        self.shape, self.r0 = shape, r0
        rcR = lambda r, i: RC[12 * r + i] * R % P
        pw = shape.pw
        # K_j[q] = (sum_{i=1..j} M^(j-i) rcR_(r0+i))_q
        def ksum(j):
            v = [0] * 12
            for i in range(1, j + 1):
                for q in range(12):
                    v[q] = (v[q] + sum(pw[j - i][q][c] * rcR(r0 + i, c) for c in range(12))) % P
            return v
        self.Ku = [ksum(j)[0] for j in range(1, g)]
        self.Kout = ksum(g)
        offA = offsets(shape.opsA, shape.schedA) if g > 1 else [0] * 16
        offB = offsets(shape.opsB, shape.schedB)
        self.cu = [(self.Ku[j - 1] - offA[j - 1]) % P for j in range(1, g)]
        self.cout = [(self.Kout[q] - offB[q]) % P for q in range(12)]


def run_group(grp, s, valu_phase_a=True):
    """s: Montgomery-form state entering round r0 (constants of r0 added) -> state entering round r0 + g (its constants added).
    valu_phase_a: the words u_j as the device's default computes them (64-bit unreduced sums of 32-bit halves times row 0 of M^j,
    csrc/poseidon_gl_grouped.hpp) instead of through the phase-A operands on the matrix pipe."""
    sh = grp.shape
    g = sh.g
    y = list(s)
    y[0] = sbox_mont(y[0])
    words = y + [0] * 4
    d = []
    if g > 1:
        if valu_phase_a:
            assert g <= 4
            lo, hi = [], []
            for j in range(1, g):
                k = grp.Ku[j - 1]
                l, h = k & 0xFFFFFFFF, k >> 32
                for i in range(12):
                    l += (y[i] & 0xFFFFFFFF) * sh.pw[j][0][i]
                    h += (y[i] >> 32) * sh.pw[j][0][i]
                assert l < (1 << 62) and h < (1 << 62)     # fold_halves takes sums below 2^63 after the d_i terms
                lo.append(l); hi.append(h)
            cu = [0] * (g - 1)
        else:
            lo, hi = run_phase(sh.opsA, sh.schedA, words)
            cu = grp.cu
        for j in range(1, g):
            l, h = lo[j - 1], hi[j - 1]
            for i in range(1, j):
                l += (d[i - 1] & 0xFFFFFFFF) * sh.tri[j - i]
                h += (d[i - 1] >> 32) * sh.tri[j - i]
            if valu_phase_a:
                assert l < (1 << 63) and h < (1 << 63)
            u = (l + (h << 32) + cu[j - 1]) % P
            dj = (sbox_mont(u) - u) % P
            # the device keeps lazy residues: any u64 congruent to the value; model that with a random representative
            d.append(dj + (P if dj < (1 << 64) - P and random.random() < 0.5 else 0))
    words = y + d + [0] * (4 - len(d))
    lo, hi = run_phase(sh.opsB, sh.schedB, words)
    return [(lo[q] + (hi[q] << 32) + grp.cout[q]) % P for q in range(12)]


RINV = pow(R, -1, P)


def sbox_mont(x):   # x R -> x^7 R
    return pow(x * RINV % P, 7, P) * R % P


def permute_grouped(s, sizes, valu_phase_a=False):
    """the device's permutation: Montgomery form, full rounds as single layers, partial rounds in groups of `sizes`"""
    assert sum(sizes) == N_PARTIAL
    s = [(x * R + RC[i] * R) % P for i, x in enumerate(s)]
    lazy = lambda v: v + (P if v < (1 << 64) - P and random.random() < 0.5 else 0)
    r = 0
    def full(s, r):
        s = [sbox_mont(x) for x in s]
        nxt = [RC[12 * (r + 1) + i] * R % P if r + 1 < 30 else 0 for i in range(12)]
        return [(sum(M1[q][i] * s[i] for i in range(12)) + nxt[q]) % P for q in range(12)]
    for _ in range(N_FULL_HALF):
        s = full(s, r); r += 1
    for g in sizes:
        s = [lazy(v) for v in run_group(Group(SHAPES[g], r), [lazy(v) for v in s], valu_phase_a and g <= 4)]
        r += g
    for _ in range(N_FULL_HALF):
        s = full(s, r); r += 1
    return [x * RINV % P for x in s]


SHAPES = {g: Shape(g) for g in (1,) + GROUP_SIZES}


def check(n=6):
    rnd = random.Random(1234)
    states = [[0] * 12, [P - 1] * 12, [rnd.randrange(P) for _ in range(12)]]
    states += [[rnd.choice((0, 1, P - 1, 0x80808080_80808080 % P, 0x7F7F7F7F_7F7F7F7F, rnd.randrange(P))) for _ in range(12)] for _ in range(n)]
    plans = [(2,) * 11, (4,) * 5 + (2,), (4,) * 4 + (2,) * 3, (1,) * 22]
    for st in states:
        want = permute_naive(st)
        for sizes in plans:
            for valu in (False, True):      # phase A on the matrix pipe / as VALU dot products (the product; groups of <= 4)
                got = permute_grouped(st, sizes, valu)
                assert got == want, (sizes, valu, st)
    return len(states) * len(plans)


# ------------------------------------------------------------------ header
def pack_operand(a):
    """[16 rows][4 dwords]: dword w of row q = slots 4w .. 4w+3 as bytes (little endian), two's complement"""
    rows = []
    for q in range(16):
        dws = []
        for w in range(4):
            v = 0
            for e in range(4):
                v |= (a[q][4 * w + e] & 0xFF) << (8 * e)
            dws.append(v)
        rows.append(dws)
    return rows


def group_init(g, bias_bits):
    """accumulator start values per first round r0 = 4 .. 25: BIAS + the 32-bit halves of (K - offsets - BIAS - 2^32 BIAS) mod p"""
    sh = SHAPES[g]
    bias = 1 << bias_bits
    half = lambda c: (bias + (c & 0xFFFFFFFF), bias + (c >> 32))
    ulo, uhi, olo, ohi = [], [], [], []
    for r0 in range(N_FULL_HALF, N_FULL_HALF + N_PARTIAL):
        if r0 + g > N_FULL_HALF + N_PARTIAL:
            ulo.append([0] * 4); uhi.append([0] * 4); olo.append([0] * 12); ohi.append([0] * 12)
            continue
        grp = Group(sh, r0)
        fix = lambda c: (c - bias - (bias << 32)) % P
        u = [half(fix(c)) for c in grp.cu] + [(0, 0)] * (4 - len(grp.cu))
        w = [half(fix(c)) for c in grp.cout]
        ulo.append([x[0] for x in u]); uhi.append([x[1] for x in u])
        olo.append([x[0] for x in w]); ohi.append([x[1] for x in w])
    return ulo, uhi, olo, ohi


def emit():
    o = []
    o.append("// GENERATED by tools/gen_poseidon_groups.py - do not edit.  Grouped partial rounds of Poseidon-12 on the matrix pipe:")
    o.append("// constant operands (powers of the MDS matrix of hash/poseidon_goldilocks.rs:301-302 cut into signed byte planes), MFMA")
    o.append("// schedules and accumulator start values; see the generator's docstring and csrc/poseidon_gl_grouped.hpp.")
    o.append("#pragma once")
    o.append("#include <stdint.h>")
    o.append("#if defined(__HIPCC__)   // (the one host / device split: the header is also read by host-only tools)")
    o.append("#define GB_GROUPS_DEVICE __device__")
    o.append("#define GB_GROUPS_ACCESSORS(OPS, INIT)                                                    \\")
    o.append("    static __device__ __forceinline__ const uint32_t (*ops())[16][4] { return OPS; }   \\")
    o.append("    static __device__ __forceinline__ const GroupInit& init() { return INIT; }")
    o.append("#else")
    o.append("#define GB_GROUPS_DEVICE")
    o.append("#define GB_GROUPS_ACCESSORS(OPS, INIT)")
    o.append("#endif")
    o.append("namespace poseidon_gl_groups {")
    o.append("struct Mfma { unsigned char p, k, comp; };   // data plane, operand (matrix plane), complemented bytes")
    o.append("// accumulator start values of a group whose first round is r0 = 4 + index: the u rows (phase A) and the state rows (phase B)")
    o.append("struct GroupInit { uint64_t ulo[22][4], uhi[22][4], olo[22][12], ohi[22][12]; };")
    o.append("template <int G> struct Shape;")
    # phase A on the VALU: u_j = sum_i ROW0[j][i] y_i + KU[r0 - 4][j - 1] + sum_{i<j} d_i TRI[j - i]
    pw = [mpow(e) for e in range(5)]
    o.append("static constexpr uint32_t ROW0[5][12] = {   // row 0 of M^j, exact integers (j <= 4: below 2^29)")
    for j in range(5):
        assert max(pw[j][0]) < (1 << 29)
        o.append("    {" + ", ".join("%du" % v for v in pw[j][0]) + "},")
    o.append("};")
    o.append("// K_j of a group whose first round is r0 = 4 + index: (sum_{i=1..j} M^(j-i) rc_(r0+i) R)_0 mod p, j = 1 .. 4")
    o.append("GB_GROUPS_DEVICE static const uint64_t KU[22][4] = {")
    for r0 in range(N_FULL_HALF, N_FULL_HALF + N_PARTIAL):
        jmax = min(4, N_FULL_HALF + N_PARTIAL - r0)   # rounds r0+1 .. r0+j must be partial rounds' constants or the next full round's
        ks = Group(Shape(jmax + 1), r0).Ku if jmax >= 1 else []
        ks = list(ks) + [0] * (4 - len(ks))
        o.append("    {" + ", ".join("0x%xull" % v for v in ks) + "},")
    o.append("};")
    fmt64 = lambda rows: ", ".join("{" + ", ".join("0x%xull" % v for v in r) + "}" for r in rows)
    for g in GROUP_SIZES:
        sh = SHAPES[g]
        nA, nB = sh.n_mfma()
        bA = sh.bound(sh.opsA, sh.schedA)
        bB = sh.bound(sh.opsB, sh.schedB)
        bias_bits = max(bA, bB).bit_length() + 1
        # the recombination adds plane pairs in 32 bits: |d_P + 256 d_(P+1)| must stay below 2^31
        for ops, sched in ((sh.opsA, sh.schedA), (sh.opsB, sh.schedB)):
            for P8 in range(8):
                for q in range(16):
                    assert sum(sum(abs(x) for x in ops[k][q]) * 128 for (_, k, _) in sched[P8]) * 257 < (1 << 31)
        ops = sh.opsA + sh.opsB
        o.append("GB_GROUPS_DEVICE static const uint32_t OPS_%d[%d][16][4] = {   // phase A planes 0..%d, then phase B planes 0..%d" % (g, len(ops), g - 2, g - 1))
        for a in ops:
            o.append("    {" + ", ".join("{" + ", ".join("0x%08xu" % v for v in row) + "}" for row in pack_operand(a)) + "},")
        o.append("};")
        ulo, uhi, olo, ohi = group_init(g, bias_bits)
        o.append("GB_GROUPS_DEVICE static const GroupInit INIT_%d = {" % g)
        for t in (ulo, uhi, olo, ohi):
            o.append("    {" + fmt64(t) + "},")
        o.append("};")
        o.append("template <> struct Shape<%d> {" % g)
        o.append("    static constexpr int G = %d, NA = %d, NB = %d, MFMA_A = %d, MFMA_B = %d, BIAS_BITS = %d;" % (g, g - 1, g, nA, nB, bias_bits))
        for name, sched in (("SCHED_A", sh.schedA), ("SCHED_B", sh.schedB)):
            mx = max(len(x) for x in sched)
            o.append("    static constexpr int LEN_%s[8] = {%s};" % (name[-1], ", ".join(str(len(x)) for x in sched)))
            o.append("    static constexpr Mfma %s[8][%d] = {" % (name, mx))
            for x in sched:
                ent = ["{%d, %d, %d}" % e for e in x] + ["{0, 0, 0}"] * (mx - len(x))
                o.append("        {" + ", ".join(ent) + "},")
            o.append("    };")
        o.append("    static constexpr uint32_t TRI[%d] = {%s};   // (M^k)_00" % (g, ", ".join(str(t) + "u" for t in sh.tri)))
        o.append("    GB_GROUPS_ACCESSORS(OPS_%d, INIT_%d)" % (g, g))
        o.append("};")
    o.append("}  // namespace poseidon_gl_groups")
    return "\n".join(o) + "\n"


if __name__ == "__main__":
    n = check()
    print("model == naive permutation on %d (state, plan) pairs" % n)
    for g in GROUP_SIZES:
        sh = SHAPES[g]
        print("G=%d  MFMAs phase A/B = %s  |raw| < 2^%d / 2^%d" % (g, sh.n_mfma(), sh.bound(sh.opsA, sh.schedA).bit_length(), sh.bound(sh.opsB, sh.schedB).bit_length()))
    if "--check" not in sys.argv:
        with open(OUT, "w") as f:
            f.write(emit())
        print("wrote", OUT)
