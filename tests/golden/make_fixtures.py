#!/usr/bin/env python3
"""Extract golden DATA (never source text) from the reference's own test fixtures.

Run in the build container only (needs /root/reference); the resulting small binary / json
files are committed under tests/golden/ and are what travels to the GPU box.

Sources (all read as text, nothing is imported or executed):
  plonky2/src/recursion/regression_test_data.rs:5,62,93   three byte arrays
  plonky2/src/hash/poseidon_goldilocks.rs:1169-1190        4 Poseidon-12 known-answer vectors
  plonky2/src/recursion/recursive_verifier.rs:427-477      3 circuit_digest known answers
  plonky2/src/util/mod.rs:65-83                            256-entry bit-reverse table
  field/src/fft.rs:227-229                                 deterministic fft test input rule
"""
import json
import os
import re
import sys

REF = os.environ.get("GB_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def read(rel):
    with open(os.path.join(REF, rel)) as f:
        return f.read()


def byte_arrays():
    src = read("plonky2/src/recursion/regression_test_data.rs")
    pat = re.compile(r"static\s+(\w+)\s*:\s*\[u8;\s*(\d+)\]\s*=\s*\[(.*?)\];", re.S)
    found = {}
    for name, n, body in pat.findall(src):
        vals = [int(x) for x in re.findall(r"\d+", body)]
        assert len(vals) == int(n), (name, len(vals), n)
        found[name] = bytes(vals)
    return found


def poseidon_kats():
    src = read("plonky2/src/hash/poseidon_goldilocks.rs")
    start = src.index("let test_vectors12")
    end = src.index("check_test_vectors(test_vectors12)")
    body = src[start:end]
    neg_one = (1 << 64) - (1 << 32)  # ORDER - 1
    body = body.replace("neg_one", str(neg_one))
    nums = [int(x, 0) for x in re.findall(r"0x[0-9a-fA-F]+|\b\d+\b", body.split("= vec![", 1)[1])]
    assert len(nums) == 4 * 24, len(nums)
    vecs = []
    for k in range(4):
        chunk = nums[24 * k:24 * (k + 1)]
        vecs.append({"input": chunk[:12], "output": chunk[12:]})
    return vecs


def poseidon2_r0_babybear_kat():
    """hash/poseidon2_risc0_babybear.rs:321-342 test_against_r0_values: the one BabyBear known-answer test in the reference"""
    src = read("plonky2/src/hash/poseidon2_risc0_babybear.rs")
    body = src[src.index("fn test_against_r0_values"):src.index("poseidon2_r0.permute_mut(input)")]
    nums = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]{8}", body)]
    assert len(nums) == 48, len(nums)
    return {"input": nums[:24], "output": nums[24:]}


def digest_kats():
    src = read("plonky2/src/recursion/recursive_verifier.rs")
    start = src.index("fn test_recursive_recursive_verifier_gl")
    body = src[start:start + 6000]
    out = []
    for m in re.finditer(r"degree_bits\(\),\s*(\d+)\);\s*assert_eq!\(\s*vd\.circuit_digest\.elements,\s*\[(.*?)\]", body, re.S):
        out.append({"degree_bits": int(m.group(1)), "digest": [int(x) for x in re.findall(r"\d+", m.group(2))]})
    assert len(out) == 3, out
    return out


def bitrev_table():
    src = read("plonky2/src/util/mod.rs")
    start = src.index("let output256")
    end = src.index("assert_eq!(reverse_index_bits(&input256[..]), output256)")
    vals = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]{2}", src[start:end])]
    assert len(vals) == 256
    return vals


def main():
    arrays = byte_arrays()
    want = {
        "RECURSIVE_VERIFIER_GL_COMMON_DATA": "recursive_verifier_gl_common_data.bin",
        "RECURSIVE_VERIFIER_GL_VERIFIER_DATA": "recursive_verifier_gl_verifier_data.bin",
        "RECURSIVE_VERIFIER_GL_PROOF": "recursive_verifier_gl_proof.bin",
    }
    for k, fn in want.items():
        with open(os.path.join(OUT, fn), "wb") as f:
            f.write(arrays[k])
        print(fn, len(arrays[k]))
    kats = {
        "poseidon12": poseidon_kats(),
        "circuit_digest_gl": digest_kats(),
        "poseidon2_r0_babybear": poseidon2_r0_babybear_kat(),
        "reverse_index_bits_256": bitrev_table(),
    }
    with open(os.path.join(OUT, "reference_kats.json"), "w") as f:
        json.dump(kats, f, indent=1)
    print("reference_kats.json ok")


if __name__ == "__main__":
    sys.exit(main())
