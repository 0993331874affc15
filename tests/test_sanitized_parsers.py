"""The host-side parsers of untrusted bytes (csrc/verifier_host.inc, csrc/compress_host.inc) under AddressSanitizer +
UndefinedBehaviorSanitizer (sanitizer row): a host-only build of the library's host code
(tests/sanitize/build.py) is driven by a mutation loop (tests/sanitize/fuzz_host_parsers.cpp) over
  * the reference's own serialized recursion proof (RECURSIVE_VERIFIER_GL, zero-knowledge, twelve gates) and its compressed form,
  * a CPU-oracle proof of the dummy circuit for each field,
truncated, bit-flipped, spliced and extended: gb_verify / gb_proof_compress / gb_proof_decompress / gb_verify_compressed may
only answer GB_OK / GB_ERR_INVALID / GB_ERR_VERIFY (or GB_ERR_BUFFER_TOO_SMALL) - never crash, never trip a sanitizer.
CPU only (GPU sanitizers are not available on this pool); no GPU is touched."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

from oracle import plonk_dummy as D
from oracle import verifier as V
from oracle.fields import BB, GL

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "sanitize"))


@pytest.fixture(scope="module")
def harness():
    if not os.path.exists("/opt/rocm/lib/llvm/bin/clang++"):
        pytest.skip("no ROCm clang++ to build the sanitized harness with")
    import build as SB
    return SB.build()


def _case(path, cfg_words, gates, k_is, cap, digest, proof, dtype):
    gates = [tuple(g) + (0,) * (7 - len(g)) for g in gates]
    with open(path, "wb") as f:
        f.write(struct.pack("<18I", *cfg_words))
        f.write(struct.pack("<I", len(gates)))
        for g in gates:
            f.write(struct.pack("<7I", *g))
        f.write(np.ascontiguousarray(k_is, dtype=dtype).tobytes())
        f.write(np.ascontiguousarray(cap, dtype=dtype).tobytes())
        f.write(np.ascontiguousarray(digest, dtype=dtype).tobytes())
        f.write(struct.pack("<Q", len(proof)))
        f.write(proof)


def _run_alloc_fail(harness, case, points):
    env = dict(os.environ)
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:allocator_may_return_null=1"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([harness, case, "alloc-fail", str(points)], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-4000:]
    assert "alloc-fail ok" in out.stdout
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-4000:]
    return out.stdout


def _run(harness, case, iterations, seed):
    env = dict(os.environ)
    env["ASAN_OPTIONS"] = "detect_leaks=0:abort_on_error=0:allocator_may_return_null=1"
    env["UBSAN_OPTIONS"] = "print_stacktrace=1:halt_on_error=1"
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([harness, case, str(iterations), str(seed)], capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0, (out.stdout + out.stderr)[-4000:]
    assert "fuzz ok" in out.stdout
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, out.stderr[-4000:]
    return out.stdout


def test_reference_recursion_proof_mutations(harness, golden_dir, tmp_path):
    rd = lambda n: open(os.path.join(golden_dir, n), "rb").read()
    common = rd("recursive_verifier_gl_common_data.bin")
    cd = V.read_common_data(common)
    vd = V.read_verifier_data(rd("recursive_verifier_gl_verifier_data.bin"))
    gates = V.read_gates(common, cd)
    cfg, fc = cd["config"], cd["config"]["fri_config"]
    nsel = len(cd["selectors_info"]["groups"])
    words = [0, cd["fri_params"]["degree_bits"], cfg["num_wires"], cfg["num_routed_wires"], cd["num_constants"] - nsel,
             cfg["num_challenges"], cd["quotient_degree_factor"], fc["rate_bits"], fc["cap_height"], fc["proof_of_work_bits"],
             fc["num_query_rounds"], 4, 5, nsel, 0, 0, 1 if cd["fri_params"]["hiding"] else 0, cd["num_public_inputs"]]
    case = str(tmp_path / "ref.case")
    _case(case, words, gates, cd["k_is"], vd["constants_sigmas_cap"], vd["circuit_digest"], rd("recursive_verifier_gl_proof.bin"),
          np.uint64)
    line = _run(harness, case, 120, 0xC0FFEE)
    # the loop must have reached both failure classes, not bounced off the first length check every time
    counts = [int(x) for x in line.split("verify ok/invalid/verify = ")[1].split(";")[0].split("/")]
    assert counts[1] > 0 and counts[2] > 0, line
    # "never unwinds" under a failing allocator (round 6): gb_verifier_create / gb_verify / gb_proof_* / gb_verify_compressed with
    # the N-th operator new throwing, N swept over each call's allocations - every interrupted call answers GB_ERR_OOM
    out = _run_alloc_fail(harness, case, 60)
    assert int(out.split("alloc-fail ok: ")[1].split()[0]) >= 100, out


@pytest.mark.parametrize("F", [GL, BB], ids=["goldilocks", "babybear"])
def test_dummy_circuit_proof_mutations(harness, tmp_path, F):
    circ = D.DummyCircuit(5, F=F)
    proof, _ = D.prove_cpu(circ, circ.witness(seed=3))
    assert D.verify(circ, proof)
    c = circ.cfg
    words = [0 if F is GL else 1, circ.degree_bits, c.num_wires, c.num_routed_wires, c.num_constants, c.num_challenges,
             c.max_quotient_degree_factor, c.rate_bits, c.cap_height, c.proof_of_work_bits, c.num_query_rounds, c.arity_bits,
             c.final_poly_bits, circ.num_selectors, 0, 0, 0, 0]
    case = str(tmp_path / "dummy.case")
    _case(case, words, circ.gate_table, circ.k_is, circ.constants_sigmas_cap, circ.circuit_digest, proof, F.dtype)
    _run(harness, case, 300, 7 + F.D)
    _run_alloc_fail(harness, case, 40)


@pytest.mark.parametrize("F,lg,kw,zk", [
    (GL, 6, dict(rate_bits=6, cap_height=0, arity_bits=6, final_poly_bits=0, num_query_rounds=6), False),
    (BB, 4, dict(rate_bits=5, cap_height=2, num_challenges=11, num_query_rounds=5), True),
], ids=["goldilocks-rate6-arity64", "babybear-rate5-11challenges-zk"])
def test_other_configurations_mutations(harness, tmp_path, F, lg, kw, zk):
    """rate_bits above the quotient degree, a FRI arity of 2^6, cap_height 0, eleven challenges, salted leaves: the parsers' length
    arithmetic depends on all of them.  The harness also refuses any accepted mutant that is not the identity."""
    cfg = D.CircuitConfig(**kw) if F is GL else D.CircuitConfig.babybear(**kw)
    circ = D.DummyCircuit(lg, cfg, F=F)
    salts = None
    if zk:
        circ.zero_knowledge = True
        salts = F.fill(77, 12 * (circ.n << cfg.rate_bits)).reshape(3, 4, -1)
    for attempt in range(6):
        try:
            proof, _ = D.prove_cpu(circ, circ.witness(seed=3 + attempt), salts=salts)
            break
        except RuntimeError as e:
            assert "rc=1" in str(e)
    assert D.verify(circ, proof)
    c = circ.cfg
    words = [0 if F is GL else 1, circ.degree_bits, c.num_wires, c.num_routed_wires, c.num_constants, c.num_challenges,
             c.max_quotient_degree_factor, c.rate_bits, c.cap_height, c.proof_of_work_bits, c.num_query_rounds, c.arity_bits,
             c.final_poly_bits, circ.num_selectors, 0, 0, 1 if zk else 0, 0]
    case = str(tmp_path / "other.case")
    _case(case, words, circ.gate_table, circ.k_is, circ.constants_sigmas_cap, circ.circuit_digest, proof, F.dtype)
    _run(harness, case, 200, 31 + lg)
