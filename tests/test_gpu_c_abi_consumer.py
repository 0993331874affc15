"""The C ABI used the way a Rust FFI shim would use it: a C++ program that includes only include/goldibear_gpu.h and links
libgoldibear_gpu.so (no Python, no torch) builds the dummy circuit, commits, proves and verifies.  -m gpu only."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("degree_bits", [5, 10])
def test_compiled_consumer(tmp_path, degree_bits):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++ on this box")
    libdir = os.path.join(ROOT, "plonky2_goldibear_amd", "lib")
    exe = str(tmp_path / "c_abi_consumer")
    subprocess.check_call([gxx, "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi", "c_abi_consumer.cpp"), "-L", libdir, "-lgoldibear_gpu",
                           "-Wl,-rpath," + libdir, "-Wl,--allow-shlib-undefined", "-o", exe])
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = "/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    out = subprocess.run([exe, str(degree_bits)], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "c_abi_consumer ok" in out.stdout
