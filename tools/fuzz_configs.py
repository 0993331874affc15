"""More seeds of the GPU suite's seeded fuzz tests, every result against the CPU oracle as in the tests themselves:
  python tools/fuzz_configs.py 48 348                  # tests/test_gpu_config_fuzz.py: configurations of prove(), proof bytes
  python tools/fuzz_configs.py commit 60 460           # tests/test_gpu_commit_fuzz.py: batch shapes, every coefficient / leaf / digest
  python tools/fuzz_configs.py circuit 24 174          # tests/test_gpu_circuit_fuzz.py: random circuits, proof bytes
usage (GPU box): ... > gpurun_out/config_fuzz.txt"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_config_fuzz as T  # noqa: E402
from plonky2_goldibear_amd import GpuContext  # noqa: E402


def other_family(kind, first, last):
    mod = __import__("test_gpu_commit_fuzz" if kind == "commit" else "test_gpu_circuit_fuzz")
    fn = mod.test_random_commit_shape if kind == "commit" else mod.test_random_circuit
    ctx = GpuContext(0)
    t0, bad = time.time(), 0
    for seed in range(first, last):
        try:
            fn(ctx, seed)
        except Exception as e:   # keep going: the point is the list
            bad += 1
            print("FAIL %s seed %d: %s" % (kind, seed, repr(e)[:300]), flush=True)
        if (seed - first) % 50 == 49:
            print("... %d, %d failures, %.0f s" % (seed - first + 1, bad, time.time() - t0), flush=True)
    print("%s fuzz: %d seeds (%d..%d), %d failures, %.0f s" % (kind, last - first, first, last - 1, bad, time.time() - t0))
    ctx.close()
    return 1 if bad else 0


def main():
    if sys.argv[1] in ("commit", "circuit"):
        return other_family(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
    first, last = int(sys.argv[1]), int(sys.argv[2])
    ctx = GpuContext(0)
    t0, bad = time.time(), []
    shapes = {}
    for seed in range(first, last):
        F, tag, lg, cfg, bits, zk = T.draw(seed)
        key = (F.name, "rate %d" % cfg.rate_bits)
        shapes[key] = shapes.get(key, 0) + 1
        try:
            T.test_random_configuration.__wrapped__(ctx, seed) if hasattr(T.test_random_configuration, "__wrapped__") else T.test_random_configuration(ctx, seed)
        except Exception as e:   # keep going: the point is the list
            bad.append((seed, repr(e)[:300]))
            print("FAIL seed %d: %s" % (seed, repr(e)[:300]), flush=True)
        if (seed - first) % 25 == 24:
            print("... %d configurations, %d failures, %.0f s" % (seed - first + 1, len(bad), time.time() - t0), flush=True)
    print("%d configurations (seeds %d..%d), %d failures, %.0f s" % (last - first, first, last - 1, len(bad), time.time() - t0))
    print("by field and rate:", ", ".join("%s %s: %d" % (k[0], k[1], v) for k, v in sorted(shapes.items())))
    ctx.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
