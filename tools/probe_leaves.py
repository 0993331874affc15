#!/usr/bin/env python3
"""Where the wave cycles of k_gl_merkle_leaves go: s_memtime segment probes (rocprofv3's PC sampling and thread
trace are not available on this pool - "configuration is not supported on any of the agents", no trace decoder library).

The attribution build of the library (tools/build_variant.sh probe "-DGB_LAB" kernels_merkle.hip; never the product) stamps the
shader clock (s_memtime) and a site ID at every segment boundary of the permutation in a few hundred waves spread over the grid.
This script runs the wires commitment (135 x 2^20 Goldilocks, input resident in HBM) on that library, reads the trace back and prints,
per segment: occurrences per permutation, wave cycles per occurrence, share of the wave's lifetime - next to the segment's STATIC
instruction mix from the same build's assembly, priced with the issue-cost model of DESIGN.md section 4 (tools/isa_mix.py).  A wave
shares its SIMD with three others, so "wave cycles per model issue cycle" is ~4 / efficiency for a segment that issues as the model
says and larger where the wave waits for something the other waves do not fill.

  tools/build_variant.sh probe "-DGB_LAB" kernels_merkle.hip
  gpurun -- 'python3 tools/probe_leaves.py > gpurun_out/probe_leaves.txt'
"""
import collections
import ctypes as C
import os
import re
import shutil
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
VARIANT = os.path.join(ROOT, "tools", "bin", "libs", "probe.so")
SYMBOL = "_ZN3gbk18k_gl_merkle_leavesEPKymjyPy"

NAMES = {
    (1, 2): "absorb: 8 column loads + to_mont (first)", (3, 2): "absorb: 8 column loads + to_mont",
    (2, 10): "first round constants", (10, 11): "full round: 12 s-boxes", (22, 11): "full round: 12 s-boxes",
    (36, 11): "full round: 12 s-boxes (after the last group)", (11, 20): "layer: cut into byte planes",
    (20, 21): "layer: 8 MFMAs + recombination", (21, 22): "layer: fold", (20, 3): "last layer (capacity only): MFMAs + recombination + fold",
    (21, 3): "last layer: fold", (22, 30): "group: first s-box", (36, 30): "group: first s-box", (30, 31): "group: cut into byte planes + complements",
    (31, 32): "group: phase A (VALU dot products)", (32, 33): "group: the dependent s-boxes", (33, 34): "group: d_j byte planes",
    (34, 35): "group: phase B MFMA chains + recombination", (35, 36): "group: fold",
}


def static_segments(flags=("-DGB_LAB",)):
    """instruction mix between consecutive probe sites of the kernel, in assembly order: {(id_a, id_b): [mix, ...]}"""
    import isa_mix
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "plonky2_goldibear_amd", "csrc"),
                               "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out,
                               os.path.join(ROOT, "plonky2_goldibear_amd", "csrc", "kernels_merkle.hip"), *flags], stderr=subprocess.DEVNULL)
        text = open(out).read()
    body = re.search(r"^%s:.*?s_endpgm" % re.escape(SYMBOL), text, re.S | re.M).group(0).splitlines()
    segs, cur, last = collections.defaultdict(list), None, None
    for line in body:
        m = re.search(r"; GB_PROBE_SITE (\d+)", line)
        if m:
            sid = int(m.group(1))
            if cur is not None:
                segs[(last, sid)].append(cur)
            cur, last = collections.Counter(), sid
            continue
        t = line.split(";")[0].split()
        if cur is None or not t:
            continue
        op = t[0]
        if op == "s_nop":
            cur["nop_ws"] += int(t[1], 0) + 1
            cur["s_nop"] += 1
        elif op.startswith("s_waitcnt"):
            cur["s_waitcnt"] += 1
        elif op.startswith("s_load") or op.startswith("s_buffer_load"):
            cur["smem"] += 1
        elif op.startswith("ds_"):
            cur["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_")):
            cur["vmem"] += 1
        elif op.startswith("s_"):
            cur["salu"] += 1
        elif op.startswith("v_mfma"):
            cur["mfma"] += 1
        elif op.startswith(("v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_hi_i32")):
            cur["mad"] += 1
        elif op.startswith("v_"):
            cur["plain32" if any(op.startswith(p) for p in isa_mix.PLAIN32) else "other"] += 1
    return segs


def model_cycles(mix):
    import isa_mix
    return sum(mix.get(k, 0) * isa_mix.COST[k] for k in ("mad", "plain32", "other", "mfma"))


def main():
    if not os.path.exists(VARIANT):
        raise SystemExit("build the attribution library first: tools/build_variant.sh probe \"-DGB_LAB\" kernels_merkle.hip")
    from plonky2_goldibear_amd import build as B
    product = B.LIB
    keep = product + ".product"
    shutil.copy(product, keep)
    try:
        shutil.copy(VARIANT, product)
        run()
    finally:
        shutil.move(keep, product)


def run():
    import torch
    from plonky2_goldibear_amd import GpuContext, PolynomialBatch, native
    lib = native.load()
    lib.gb_probe_setup.restype, lib.gb_probe_setup.argtypes = C.c_int, [C.c_void_p, C.c_uint, C.c_uint, C.c_uint]
    log_n, ncols, nslots, per = 20, 135, 512, 2048
    n_waves = (8 << log_n) // 64
    step = n_waves // nslots
    ctx = GpuContext(0)
    rng = np.random.default_rng(1)
    vals = (rng.integers(0, 1 << 63, (ncols, 1 << log_n), dtype=np.uint64) % np.uint64(0xFFFFFFFF00000001))
    dev = torch.from_numpy(vals.view(np.int64)).cuda()
    trace = torch.zeros(nslots * per * 2, dtype=torch.int64, device="cuda")
    b = PolynomialBatch.from_values(ctx, dev, 3, 4)   # warm-up without probes (tables, pool)
    b.free()
    torch.cuda.synchronize()
    assert lib.gb_probe_setup(trace.data_ptr(), per, step, nslots) == 0
    ctx.set_profiling(True)
    ctx.scope_reset()
    b = PolynomialBatch.from_values(ctx, dev, 3, 4)
    ctx.synchronize()
    ms = ctx.scope_ms("hash leaves")[0]
    assert lib.gb_probe_setup(None, 0, 0, 0) == 0
    t = trace.cpu().numpy().view(np.uint64).reshape(nslots, per, 2)
    b.free()
    perms = -(-ncols // 8)
    seg_cyc, seg_cnt, life = collections.Counter(), collections.Counter(), []
    first_ts, last_ts = [], []
    for w in range(nslots):
        ids = t[w, :, 1]
        k = int(np.count_nonzero(ids))
        if k < 10:
            continue
        ts, ids = t[w, :k, 0].astype(np.int64), ids[:k].astype(np.int64)
        d = np.diff(ts)
        for a, bb, c in zip(ids[:-1], ids[1:], d):
            seg_cyc[(int(a), int(bb))] += int(c)
            seg_cnt[(int(a), int(bb))] += 1
        life.append(int(ts[-1] - ts[0]))
        first_ts.append(int(ts[0]))
        last_ts.append(int(ts[-1]))
    nw = len(life)
    total = sum(seg_cyc.values())
    stat = static_segments()
    print("k_gl_merkle_leaves (attribution build, -DGB_LAB), wires commitment 135 x 2^20 Goldilocks: 'hash leaves' %.2f ms with probes" % ms)
    print("%d traced waves, %.0f wave cycles (s_memtime) from first to last probe on average, %d permutations per wave" % (nw, np.mean(life), perms))
    # Shader clock under this load.  s_memtime counters of different XCDs are not synchronised, so stamps of different waves are not
    # compared; instead: the grid's n_waves run in 4096 wave slots (256 CUs x 16 waves: 128 VGPRs, 4 waves per SIMD), back to back, so a
    # slot is busy for (n_waves / 4096) wave lifetimes during the kernel.  First-to-last probe leaves out a wave's prologue and epilogue
    # (the operand table, the store: ~1 %), so this reads slightly low.
    clock_hz = (n_waves / 4096.0) * float(np.mean(life)) / (ms * 1e-3)
    print("shader clock under this load: ~%.2f GHz (%d waves per slot x the mean wave lifetime / the kernel's 'hash leaves' scope; the guide's nominal "
          "clock is 2.4 GHz)" % (clock_hz / 1e9, n_waves // 4096))
    print()
    print("%-58s %6s %9s %7s | %5s %5s %5s %4s %5s %5s %4s %8s %6s" % ("segment (site a -> site b)", "n/perm", "cyc/occ", "share", "mad", "pl32", "other",
                                                                        "mfma", "nopws", "salu", "lds", "model", "cyc/m"))
    rows = []
    for key, cyc in seg_cyc.items():
        cnt = seg_cnt[key]
        mixes = stat.get(key, [])
        mix = collections.Counter()
        for m in mixes:
            mix.update(m)
        for k2 in mix:
            mix[k2] /= max(1, len(mixes))
        rows.append((cyc / total, key, cnt / nw / perms, cyc / cnt, mix))
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0.0])
    for share, key, npp, cpo, mix in sorted(rows, reverse=True):
        name = NAMES.get(key, "(%d -> %d)" % key)
        mc = model_cycles(mix)
        print("%-58s %6.2f %9.0f %6.1f%% | %5.0f %5.0f %5.0f %4.0f %5.0f %5.0f %4.0f %8.0f %6.2f" % (
            name[:58], npp, cpo, 100 * share, mix["mad"], mix["plain32"], mix["other"], mix["mfma"], mix["nop_ws"], mix["salu"], mix["lds"],
            mc, cpo / mc if mc else float("nan")))
        cls = name.split(":")[0]
        agg[cls][0] += share
        agg[cls][1] += npp * cpo
        agg[cls][2] += npp * mc
    print()
    print("%-28s %7s %14s %14s %8s" % ("class", "share", "cycles/perm", "model/perm", "cyc/m"))
    tot_c = tot_m = 0.0
    for cls, (share, c, m) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
        print("%-28s %6.1f%% %14.0f %14.0f %8.2f" % (cls, 100 * share, c, m, c / m if m else float("nan")))
        tot_c, tot_m = tot_c + c, tot_m + m
    print("%-28s %6.1f%% %14.0f %14.0f %8.2f" % ("all", 100.0, tot_c, tot_m, tot_c / tot_m))
    print()
    out_json = os.environ.get("GB_PROBE_JSON")
    if out_json:
        import json
        from csrc_hash import measured_sha16
        json.dump({"kernel": "gbk::k_gl_merkle_leaves (attribution build, -DGB_LAB)", "hash_leaves_ms_with_probes": ms, "traced_waves": nw,
                   "wave_cycles_per_permutation": tot_c, "model_issue_cycles_per_permutation": tot_m,
                   "simd_issue_utilisation_vs_model": 4.0 * tot_m / tot_c, "shader_clock_hz_under_load": clock_hz,
                   "classes": {cls: {"share": v[0], "wave_cycles_per_permutation": v[1], "model_cycles_per_permutation": v[2]} for cls, v in agg.items()},
                   "csrc_sha16": measured_sha16()}, open(out_json, "w"), indent=1)
    print("cyc/m = wave cycles per model issue cycle: four waves share a SIMD, so a segment that issues exactly as the cost model says reads 4.0 / (SIMD issue "
          "utilisation); a larger figure is a segment in which the wave waits for something its three neighbours do not fill")


if __name__ == "__main__":
    main()
