#!/usr/bin/env python3
"""Issue slots of a lone wave: per loop of one kernel's gfx950 assembly, the instructions and the wait states the hazard recognizer
had to insert (s_nop N = N + 1 slots; on gfx950 two of them between a VALU instruction that writes a carry - SGPR pair or VCC - and
the VALU instruction that reads it).  In a throughput kernel other waves issue into those slots; a latency-bound kernel that runs
one wave per SIMD (the cooperative permutation of the small tree levels) pays for them.

  python tools/isa_slots.py plonky2_goldibear_amd/csrc/kernels_merkle.hip _ZN3gbk22k_gl_merkle_level_coopEPKyPyy [trip counts ...]"""
import os
import re
import subprocess
import sys
import tempfile


def main():
    src, symbol = sys.argv[1], sys.argv[2]
    trips = [int(a) for a in sys.argv[3:]]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(root, "plonky2_goldibear_amd", "csrc"),
                               "-I" + os.path.join(root, "include"), "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
        text = open(out).read()
    m = re.search(r"^%s:.*?s_endpgm" % re.escape(symbol), text, re.S | re.M)
    if not m:
        raise SystemExit("kernel %s not found" % symbol)
    body = m.group(0).splitlines()

    def count(seg):
        ins = [x.split(";")[0].split() for x in seg if x.strip() and not x.strip().startswith((".", ";"))]
        ins = [x for x in ins if x and not x[0].endswith(":")]
        nops = sum(int(x[1]) + 1 for x in ins if x[0] == "s_nop")
        real = sum(1 for x in ins if x[0] != "s_nop")
        carries = sum(1 for x in ins if re.match(r"v_(addc|subb|subbrev)_co_u32|v_cndmask_b32", x[0]))
        return real, nops, carries

    labels = {}
    for i, l in enumerate(body):
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            labels[mm.group(1)] = i
    loops = []
    for i, l in enumerate(body):
        mm = re.search(r"s_cbranch_\w+ (\.LBB\d+_\d+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            loops.append((labels[mm.group(1)], i, mm.group(1)))
    print("%s: %d lines of assembly" % (symbol, len(body)))
    total_real = total_nop = 0
    for k, (a, b, name) in enumerate(loops):
        real, nops, carries = count(body[a:b + 1])
        t = trips[k] if k < len(trips) else 1
        total_real += real * t
        total_nop += nops * t
        print("  loop %-10s x %3d: %4d instructions + %3d wait states = %4d slots per trip (%d carry / select consumers)" % (name, t, real, nops, real + nops, carries))
    real, nops, _ = count(body)
    inloop_real = sum(count(body[a:b + 1])[0] for a, b, _ in loops)
    inloop_nop = sum(count(body[a:b + 1])[1] for a, b, _ in loops)
    total_real += real - inloop_real
    total_nop += nops - inloop_nop
    print("  whole kernel with these trip counts: %d instructions + %d wait states = %d slots; wait states %.0f %%; at 4 cycles a slot and 2.4 GHz: %.1f us" % (
        total_real, total_nop, total_real + total_nop, 100.0 * total_nop / (total_real + total_nop), (total_real + total_nop) * 4 / 2.4e3))


if __name__ == "__main__":
    main()
