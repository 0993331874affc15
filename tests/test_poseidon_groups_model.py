"""The grouped partial rounds of Poseidon-12 (csrc/poseidon_gl_grouped.hpp): the generator's exact integer model of the device
pipeline - byte planes, signed operands, complemented planes, wrapped output planes, start values - equals the defining
permutation for every group shape, and the committed header is what the generator emits.  No GPU needed."""
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_poseidon_groups as G  # noqa: E402


def test_model_equals_defining_permutation():
    assert G.check(n=4) >= 28      # (state, plan) pairs: plans 11 x 2, 5 x 4 + 2 (the product's), 4 x 4 + 3 x 2, 22 x 1


def test_defining_permutation_is_the_references():
    # hash/poseidon_goldilocks.rs:1169-1190 (test vectors; tests/golden/reference_kats.json holds all four)
    import json
    kats = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_kats.json")))["poseidon12"]
    for kat in kats:
        inp = [int(x, 0) if isinstance(x, str) else int(x) for x in kat["input"]]
        out = [int(x, 0) if isinstance(x, str) else int(x) for x in kat["output"]]
        assert G.permute_naive(inp) == out


def test_committed_header_is_current():
    want = G.emit()
    have = open(G.OUT).read()
    assert want == have, "run python3 tools/gen_poseidon_groups.py"


def test_group_algebra_on_edge_words():
    rnd = random.Random(7)
    P = G.P
    edge = [0, 1, P - 1, 0x8080808080808080 % P, 0x7F7F7F7F7F7F7F7F, 0xFFFFFFFF, 0xFFFFFFFF00000000]
    for g in G.GROUP_SIZES:
        for r0 in (4, 26 - g):
            s = [rnd.choice(edge) for _ in range(12)]
            grp = G.Group(G.SHAPES[g], r0)
            got = G.run_group(grp, s, valu_phase_a=g <= 4)
            assert got == G.run_group(grp, s, valu_phase_a=False)
            # the same G rounds one at a time (Montgomery-form state, constants times R)
            want = list(s)
            for r in range(r0, r0 + g):
                want[0] = G.sbox_mont(want[0])
                want = [(sum(G.M1[q][i] * want[i] for i in range(12)) + G.RC[12 * (r + 1) + q] * G.R) % P for q in range(12)]
            assert got == want, (g, r0)
