#!/usr/bin/env python3
"""Static guard for the matrix-pipe hazards of gfx950 that no hardware interlock covers (tests/test_mfma_guard.py).

gfx950 does not interlock a VALU / LDS / memory instruction against a v_mfma_* whose destination tile is still in flight: the
compiler pads with s_nop from a table.  tools/microbench_mfma_hazard.hip measured what the hardware needs for
v_mfma_i32_32x32x32_i8 (profiles/r04_mfma_hazard.txt): a READ of a tile register is stale up to 11 wait states after the MFMA
(12 are enough), a WRITE to a tile register is overwritten by the pipe's own write-back up to 8 wait states after it (9 are
enough); a write to a SOURCE register is safe at once.  This script walks the assembly of every kernel (hipcc -S) in program
order and reports, per MFMA destination tile,

  * RAW: a non-MFMA instruction reading a tile register fewer than RAW_MIN wait states after the MFMA,
  * WAW: a non-MFMA instruction writing a tile register fewer than WAW_MIN wait states after it - the allocator does this when
         it parks another value in a DEAD register of the tile (outputs the kernel never reads),
  * partial: an MFMA whose accumulator operand overlaps a tile in flight without being that tile,
  * undefined: an MFMA source register that no instruction of the kernel ever writes.
Accumulation registers (a[..] tiles, v_accvgpr_read / v_accvgpr_write: the FRI-leaf and proof-of-work kernels keep their tiles
there) are followed like architectural ones, in a register namespace of their own (a<N> = register 512 + N).

Wait states are counted as the hazard recognizer counts them: one per instruction, s_nop N = N + 1.  Program order is walked once (a
forward branch does not reset the count: conservative), and every backward branch is followed once more - the top of its loop is
scanned with the tiles that were in flight at the bottom.

  python3 tools/mfma_guard.py file.s [...]        # or: --lib (compiles every csrc/*.hip to assembly first)
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

RAW_MIN = 12
WAW_MIN = 9
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REG = re.compile(r"\b([va])\[(\d+):(\d+)\]|\b([va])(\d+)\b")
AGPR_BASE = 512   # a<N> is register 512 + N of the walk: its own namespace beside v0..v511


def regs_of(operand):
    out = []
    for m in REG.finditer(operand):
        if m.group(5) is not None:
            out.append(int(m.group(5)) + (AGPR_BASE if m.group(4) == "a" else 0))
        else:
            base = AGPR_BASE if m.group(1) == "a" else 0
            out.extend(range(base + int(m.group(2)), base + int(m.group(3)) + 1))
    return out


def split_operands(rest):
    ops, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        elif ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            ops.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        ops.append(cur.strip())
    return ops


NO_VDST = ("v_cmp", "v_cmpx", "v_readlane", "v_readfirstlane", "v_nop", "global_store", "scratch_store", "buffer_store", "ds_write",
           "flat_store", "ds_store", "global_atomic", "buffer_atomic", "s_", "ds_bpermute_dummy")


def parse(line):
    """-> (opcode, writes [vgpr], reads [vgpr]) or None"""
    code = line.split(";")[0].strip()
    if not code or code.startswith(".") or code.endswith(":"):
        return None
    parts = code.split(None, 1)
    op = parts[0]
    ops = split_operands(parts[1]) if len(parts) > 1 else []
    if op.startswith("s_") or not ops:
        return op, [], [r for o in ops for r in regs_of(o)]
    has_dst = not op.startswith(NO_VDST)
    if op.startswith(("global_atomic", "buffer_atomic", "ds_")) and not op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle")):
        has_dst = False
    if has_dst:
        return op, regs_of(ops[0]), [r for o in ops[1:] for r in regs_of(o)]
    return op, [], [r for o in ops for r in regs_of(o)]


def kernels(text):
    for m in re.finditer(r"^([A-Za-z_][\w$.]*):[^\n]*\n(.*?)^\s*s_endpgm", text, re.S | re.M):
        body = m.group(2)
        if ".amdhsa_kernel" in body:
            continue
        yield m.group(1), body


def check_kernel(name, body):
    findings = []
    written = set()
    mfma_sources = []
    lines = body.splitlines()
    labels = {}
    for ln, line in enumerate(lines):
        m = re.match(r"^(\.?[A-Za-z_][\w$.]*):", line.strip())
        if m:
            labels[m.group(1)] = ln

    def scan(start, inflight, limit=None, collect=True):
        """walk from line `start` with the tiles in `inflight` ([regs, wait states, MFMA text]); limit: stop once every tile is safe"""
        back_edges = []
        for ln in range(start, len(lines)):
            line = lines[ln]
            p = parse(line)
            if p is None:
                continue
            op, wr, rd = p
            ws = 1
            if op == "s_nop":
                ws = int(line.split(";")[0].split()[1], 0) + 1
            if op.startswith("v_mfma") or op.startswith("v_smfmac"):
                ops = split_operands(line.split(";")[0].strip().split(None, 1)[1])
                dst = set(regs_of(ops[0]))
                a, b = set(regs_of(ops[1])), set(regs_of(ops[2]))
                c = set(regs_of(ops[3])) if len(ops) > 3 else set()
                if collect:
                    mfma_sources.append((ln, line.strip(), a | b | c))
                for t in inflight:
                    if t[1] < RAW_MIN and c and (c & t[0]) and c != t[0]:
                        findings.append(("partial", name, ln, t[1], line.strip(), t[2]))
                # a new MFMA into the same or an overlapping tile supersedes the old entry for the registers it covers
                for t in inflight:
                    t[0] -= dst
                inflight = [t for t in inflight if t[0]]
                for t in inflight:
                    t[1] += ws
                inflight.append([set(dst), 0, line.strip()])
                if collect:
                    written.update(dst)
                continue
            for t in inflight:
                if t[0] & set(rd) and t[1] < RAW_MIN:
                    findings.append(("RAW", name, ln, t[1], line.strip(), t[2]))
                if t[0] & set(wr) and t[1] < WAW_MIN:
                    findings.append(("WAW", name, ln, t[1], line.strip(), t[2]))
            if collect:
                written.update(wr)
            if op.startswith("s_cbranch") or op == "s_branch":
                target = line.split(";")[0].split()[1]
                if collect and target in labels and labels[target] <= ln and inflight:
                    back_edges.append((labels[target], [[set(t[0]), t[1] + ws, t[2]] for t in inflight]))
            for t in inflight:
                t[1] += ws
            inflight = [t for t in inflight if t[1] < max(RAW_MIN, WAW_MIN)]
            if limit is not None and not inflight:
                break
        return back_edges

    # program order first; then every loop back edge: the top of the loop with the tiles that were in flight at its bottom
    for target, state in scan(0, []):
        scan(target, state, limit=True, collect=False)
    seen, unique = set(), []
    for f in findings:
        if f not in seen:
            seen.add(f)
            unique.append(f)
    findings = unique
    for ln, text, src in mfma_sources:
        undefined = sorted(r for r in src if r not in written)
        if undefined:
            findings.append(("undefined", name, ln, 0, text, "v%s never written in this kernel" % undefined))
    return findings


def check_text(text):
    out = []
    n = 0
    for name, body in kernels(text):
        if "v_mfma" not in body and "v_smfmac" not in body:
            continue
        n += 1
        out.extend(check_kernel(name, body))
    return n, out


def compile_to_asm(src, flags=()):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "plonky2_goldibear_amd", "csrc"),
                               "-I" + os.path.join(ROOT, "include"), "-S", "--cuda-device-only", "-o", out, src, *flags], stderr=subprocess.DEVNULL)
        return open(out).read()


def main(argv):
    files = [a for a in argv if not a.startswith("-")]
    texts = []
    if "--lib" in argv:
        for src in sorted(glob.glob(os.path.join(ROOT, "plonky2_goldibear_amd", "csrc", "*.hip"))):
            texts.append((os.path.basename(src), compile_to_asm(src, [a for a in argv if a.startswith("-D")])))
    for f in files:
        texts.append((f, open(f).read()))
    bad = 0
    for label, text in texts:
        n, findings = check_text(text)
        print("%s: %d kernels with MFMAs, %d findings" % (label, n, len(findings)))
        for kind, name, ln, ws, line, ref in findings:
            print("  %-9s %s line %d, %d wait states after [%s]: %s" % (kind, name, ln, ws, ref, line))
        bad += len(findings)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
