"""More of tests/test_gpu_config_fuzz.py: seeds FIRST .. LAST of the same draw(), every proof's bytes against the CPU oracle prover,
gb_verify and the oracle verifier.  usage (GPU box): python tools/fuzz_configs.py 48 348 > gpurun_out/config_fuzz.txt"""
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_config_fuzz as T  # noqa: E402
from plonky2_goldibear_amd import GpuContext  # noqa: E402


def main():
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
