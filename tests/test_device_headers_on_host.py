"""The device-side permutation header compiled for the CPU (g++, tests/host_shim) and compared with the host mirror on random and
extreme states: the signed / lazy / offset bookkeeping of csrc/poseidon2_bb.hpp is integer arithmetic that does not need a GPU
to be checked, and its failures would be data dependent."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_poseidon2_bb_lane_permutation_matches_host_mirror(tmp_path):
    exe = tmp_path / "poseidon2_bb_lane"
    shim = os.path.join(ROOT, "tests", "host_shim")
    cmd = ["g++", "-O2", "-std=c++17", "-include", os.path.join(shim, "shim.h"), "-I", shim,
           "-I", os.path.join(ROOT, "plonky2_goldibear_amd", "csrc"), "-o", str(exe), os.path.join(shim, "poseidon2_bb_lane.cpp")]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    out = subprocess.run([str(exe), "200000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches=0" in out.stdout


def test_bb_wide_accumulators_match_montgomery_sums(tmp_path):
    exe = tmp_path / "bb_wide_acc"
    shim = os.path.join(ROOT, "tests", "host_shim")
    clang = "/opt/rocm/lib/llvm/bin/clang++"   # field_traits.hpp pulls in gl_field.hpp, whose limb code uses clang's __builtin_addc
    if not os.path.exists(clang):
        pytest.skip("needs the ROCm clang++ as the host compiler")
    cmd = [clang, "-O2", "-std=c++17", "-include", os.path.join(shim, "shim.h"), "-I", shim,
           "-I", os.path.join(ROOT, "plonky2_goldibear_amd", "csrc"), "-o", str(exe), os.path.join(shim, "bb_wide_acc.cpp")]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    out = subprocess.run([str(exe), "3000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches=0" in out.stdout


def test_challenge_slices_cover_every_count(tmp_path):
    """csrc/challenge_slices.hpp: how a num_challenges without a compiled kernel width is split into launches"""
    exe = tmp_path / "challenge_slices"
    shim = os.path.join(ROOT, "tests", "host_shim")
    cmd = ["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "plonky2_goldibear_amd", "csrc"), "-o", str(exe),
           os.path.join(shim, "challenge_slices.cpp")]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches=0" in out.stdout


def test_host_transcript_permutation_matches_the_defining_form(tmp_path):
    """csrc/poseidon_gl_host.hpp (the Fiat-Shamir transcript's Poseidon-12: fast partial rounds with the sparse v / w_hat / M_init
    constants, MDS rows in 128-bit accumulators) == the naive 30-round definition written from the round constants and the MDS matrix
    alone, canonical outputs, on random and extreme states"""
    exe = tmp_path / "poseidon_gl_host_forms"
    shim = os.path.join(ROOT, "tests", "host_shim")
    clang = "/opt/rocm/lib/llvm/bin/clang++"   # gl_field.hpp's limb code uses clang's __builtin_addc
    if not os.path.exists(clang):
        pytest.skip("needs the ROCm clang++ as the host compiler")
    cmd = [clang, "-O2", "-std=c++17", "-include", os.path.join(shim, "shim.h"), "-I", shim,
           "-I", os.path.join(ROOT, "plonky2_goldibear_amd", "csrc"), "-o", str(exe), os.path.join(shim, "poseidon_gl_host_forms.cpp")]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    out = subprocess.run([str(exe), "20000"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "mismatches=0" in out.stdout
