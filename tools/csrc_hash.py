#!/usr/bin/env python3
"""sha256 over the device/host sources of the library (csrc/* and the public header): the figures under profiles/ carry the hash of
the code they were measured on, and bench.py marks a quoted figure "stale" when the tree has moved on since."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def csrc_sha16():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "plonky2_goldibear_amd", "csrc")
    paths = sorted(os.path.join(d, f) for f in os.listdir(d) if f.endswith((".hip", ".hpp", ".h", ".inc")))
    paths.append(os.path.join(ROOT, "include", "goldibear_gpu.h"))
    for p in paths:
        h.update(os.path.basename(p).encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()[:16]


def measured_sha16():
    """the hash recorded on the GPU box when the counters were collected (tools/make_profiles.sh writes it next to them);
    the current tree's when there is no record"""
    p = os.path.join(ROOT, "gpurun_out", "profile_csrc_sha16.txt")
    return open(p).read().strip() if os.path.exists(p) else csrc_sha16()


if __name__ == "__main__":
    print(csrc_sha16())
