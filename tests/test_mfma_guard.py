"""Static check of the compiled kernels for the matrix-pipe hazards gfx950 leaves to software (tools/mfma_guard.py).

Root cause of round 3's "undefined MFMA operand" incident and of round 4's first grouped Poseidon kernels (HISTORY.md): the
compiler's hazard recognizer pads VALU reads / writes of an in-flight MFMA destination tile with s_nop, but not those made by
INLINE ASM - and the byte-plane recombination was asm v_mad_i64_i32 whose results the allocator could park in tile registers the
kernel never reads.  The product writes the recombination in C (compiler-selected v_mad_i64_i32, padded like any other
instruction); this test walks the assembly of every kernel that issues MFMAs, and checks the walker itself on hand-written
sequences with the hazards in them (reads and writes too soon, undefined sources, accumulation registers, loop back edges)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import mfma_guard  # noqa: E402

CSRC = os.path.join(ROOT, "plonky2_goldibear_amd", "csrc")
pytestmark = pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc (cross-compiles without a GPU)")


def findings(src, flags=()):
    n, out = mfma_guard.check_text(mfma_guard.compile_to_asm(src, list(flags)))
    assert n > 0, "no kernel with MFMAs found in " + src
    return out


@pytest.mark.parametrize("flags", [(), ("-DGB_LAB",)], ids=["product", "attribution-build"])
def test_merkle_kernels_are_clean(flags):
    assert findings(os.path.join(CSRC, "kernels_merkle.hip"), flags) == []


def test_prover_kernels_are_clean():
    assert findings(os.path.join(CSRC, "kernels_prover.hip")) == []


def test_guard_reads_registers_and_wait_states():
    text = """
k:
	v_mfma_i32_32x32x32_i8 v[0:15], v[20:23], v[24:27], 0
	s_nop 9
	v_add_u32_e32 v40, v3, v41
	v_mfma_i32_32x32x32_i8 v[0:15], v[20:23], v[24:27], v[0:15]
	v_mov_b32_e32 v14, v41
	s_nop 11
	v_add_u32_e32 v40, v15, v41
	v_mfma_i32_32x32x32_i8 v[16:31], v[20:23], v[60:63], 0
	s_endpgm
"""
    n, out = mfma_guard.check_text(text)
    kinds = sorted(f[0] for f in out)
    assert n == 1 and kinds == ["RAW", "WAW", "undefined"], out
    raw = [f for f in out if f[0] == "RAW"][0]
    assert raw[3] == 10   # s_nop 9 = ten wait states: two short of what a read needs


def test_guard_follows_tiles_in_accumulation_registers():
    """the FRI-leaf / proof-of-work kernels keep their MFMA tiles in a[..]; the walk follows those registers too - a read
    (v_accvgpr_read) or write (v_accvgpr_write) of a tile register too soon after the MFMA is reported like a v-register one"""
    text = """
k:
	v_mov_b32_e32 v20, 0
	v_mov_b32_e32 v21, 0
	v_mov_b32_e32 v22, 0
	v_mov_b32_e32 v23, 0
	v_mfma_i32_32x32x32_i8 a[0:15], v[20:23], v[20:23], 0
	s_nop 5
	v_accvgpr_read_b32 v40, a3
	v_accvgpr_write_b32 a14, v20
	s_nop 15
	v_accvgpr_read_b32 v41, a4
	s_endpgm
"""
    n, out = mfma_guard.check_text(text)
    kinds = sorted(f[0] for f in out)
    assert n == 1 and kinds == ["RAW", "WAW"], out
    assert mfma_guard.regs_of("a[2:3]") == [514, 515] and mfma_guard.regs_of("v7") == [7]


def test_guard_follows_loop_back_edges():
    text = """
k:
	v_mov_b32_e32 v20, 0
.LBB0_1:
	v_add_u32_e32 v40, v3, v41
	s_nop 15
	v_mfma_i32_32x32x32_i8 v[0:15], v[20:23], v[20:23], 0
	s_cbranch_scc1 .LBB0_1
	s_nop 15
	s_endpgm
"""
    n, out = mfma_guard.check_text(text)
    assert [f[0] for f in out if f[0] != "undefined"] == ["RAW"], out      # the read at the top of the loop, one wait state after the MFMA at its bottom
