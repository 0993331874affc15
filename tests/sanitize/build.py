"""HOST-ONLY build of the library's translation units with AddressSanitizer + UndefinedBehaviorSanitizer, linked with the
mutation harness fuzz_host_parsers.cpp (CPU only: GPU sanitizers are not available on this pool).

csrc/api.hip - all of the library's host code, including the parsers under test (verifier_host.inc, compress_host.inc); it
holds no kernels - is compiled host-only with the sanitizers and linked with the library's ordinary objects of the kernel
units (plonky2_goldibear_amd/build/*.o, never called here: the harness only uses gb_verifier_create / gb_verify /
gb_proof_*, which touch no device).  -O0 keeps the sanitizer compile at a few seconds.
Output: tests/sanitize/_build/fuzz_host_parsers."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "plonky2_goldibear_amd", "csrc")
OUT = os.path.join(HERE, "_build")
EXE = os.path.join(OUT, "fuzz_host_parsers")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SAN = ["-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer"]
COMMON = ["-std=c++17", "-O0", "-g1", "-fPIC", "-w", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include")]


def _inputs():
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC))
    return srcs + [os.path.join(HERE, "fuzz_host_parsers.cpp"), os.path.join(ROOT, "include", "goldibear_gpu.h"), __file__]


def build(force=False):
    if not os.path.exists(CLANG):
        raise RuntimeError("no ROCm clang++ at " + CLANG)
    if not force and os.path.exists(EXE) and all(os.path.getmtime(p) < os.path.getmtime(EXE) for p in _inputs()):
        return EXE
    os.makedirs(OUT, exist_ok=True)
    sys.path.insert(0, ROOT)
    from plonky2_goldibear_amd import build as B
    B.build_library()   # the kernel units' objects (and a fresh library)
    objs = [os.path.join(B.OBJDIR, os.path.basename(src) + ".o") for src in B.sources() if not src.endswith("api.hip")]
    api = os.path.join(OUT, "api.hip.o")
    subprocess.check_call([CLANG, "-x", "hip", "--cuda-host-only", "--offload-arch=gfx950"] + COMMON + SAN +
                          ["-c", os.path.join(CSRC, "api.hip"), "-o", api])
    objs.append(api)
    subprocess.check_call([CLANG] + COMMON + SAN + [os.path.join(HERE, "fuzz_host_parsers.cpp")] + objs +
                          ["-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath,/opt/rocm/lib", "-o", EXE])
    return EXE


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
