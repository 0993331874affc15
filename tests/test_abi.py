"""The C-ABI library loads and exports every symbol include/goldibear_gpu.h declares (no GPU needed)."""
import os
import re

from plonky2_goldibear_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "goldibear_gpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return set(re.findall(r"\b(gb_[a-z0-9_]+)\s*\(", src))


def test_header_symbols_are_exported_and_bound():
    lib = native.load()
    names = _declared()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), "missing export: " + n
    assert names == set(native.SIGNATURES), names ^ set(native.SIGNATURES)


def test_null_arguments_fail_cleanly_without_gpu():
    lib = native.load()
    assert lib.gb_commit_values(None, 0, None, 1, 4, 3, 4, None, 0, None) == native.GB_ERR_INVALID
    assert lib.gb_batch_free(None) == native.GB_OK
    assert lib.gb_ctx_destroy(None) == native.GB_OK
    assert b"null" in lib.gb_last_error(None)


def test_test_hooks_live_in_their_own_header():
    """the fault-injection export is declared in include/goldibear_gpu_test_hooks.h only: the product header carries no test hook
    (VERDICT r4 item 8), no flag bit for one, and the library exports what the hooks header declares"""
    hooks = open(os.path.join(ROOT, "include", "goldibear_gpu_test_hooks.h")).read()
    hooks = re.sub(r"/\*.*?\*/", "", hooks, flags=re.S)
    names = set(re.findall(r"\b(gb_[a-z0-9_]+)\s*\(", hooks))
    assert names == set(native.TEST_HOOK_SIGNATURES) == {"gb_test_arm_perm_arg_failure"}
    product = open(os.path.join(ROOT, "include", "goldibear_gpu.h")).read()
    assert "gb_test_" not in product and "GB_PROVE_FAIL_PERM_ARG" not in product
    lib = native.load()
    for n in names:
        assert hasattr(lib, n)
    assert not (names & _declared())
    # no environment switches inside the library: options go through gb_ctx_set_option
    csrc = os.path.join(ROOT, "plonky2_goldibear_amd", "csrc")
    for f in os.listdir(csrc):
        assert "getenv" not in open(os.path.join(csrc, f)).read(), f


def test_column_pointer_entry_points_reject_nulls_without_gpu():
    import ctypes as C
    lib = native.load()
    h = C.c_void_p()
    assert lib.gb_commit_values_cols(None, 0, None, 1, 4, 3, 4, None, 0, C.byref(h)) == native.GB_ERR_INVALID
    n = C.c_size_t()
    assert lib.gb_prove_cols(None, None, 0, None, 0, None, 0, C.byref(n)) == native.GB_ERR_INVALID
    assert lib.gb_host_alloc(None, 16, C.byref(h)) == native.GB_ERR_INVALID
    assert lib.gb_ctx_set_option(None, b"copy_threads", 2) == native.GB_ERR_INVALID


def test_bench_golden_hashes_are_committed():
    """tests/golden/bench_proof_sha256.json (tests/golden/make_bench_proof_golden.py): the oracle prover's proof hashes that bench.py
    compares its own proofs with - both fields at the benchmark's size"""
    import json
    g = json.load(open(os.path.join(ROOT, "tests", "golden", "bench_proof_sha256.json")))
    for key, ch, plen in (("goldilocks_2p20", 3, 198432), ("babybear_2p20", 10, 180776)):
        e = g[key]
        assert e["log_n"] == 20 and e["num_challenges"] == ch and e["proof_len"] == plen and len(e["sha256"]) == 64


def test_binding_asks_for_eight_hardware_queues_unless_the_host_chose():
    """plonky2_goldibear_amd/native.py: several contexts proving at once overlap only as far as the HIP runtime has hardware queues
    (GPU_MAX_HW_QUEUES, 4 by default, read when the runtime initialises) - the binding sets 8 at import unless the variable is set
    (include/goldibear_gpu.h at gb_ctx_create; tools/bench_recursion_shape.py --inflight 6: 575 -> 839 proofs/s)"""
    import subprocess
    import sys
    code = "import os; import plonky2_goldibear_amd.native; print(os.environ['GPU_MAX_HW_QUEUES'])"
    for preset, want in ((None, "8"), ("4", "4")):
        env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
        if preset:
            env["GPU_MAX_HW_QUEUES"] = preset
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=env, timeout=120)
        assert out.returncode == 0, out.stderr
        assert out.stdout.strip() == want
