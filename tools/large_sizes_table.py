#!/usr/bin/env python3
"""profiles/rNN_large_sizes.txt from the raw output of tools/large_sizes.sh (gpurun_out/large_sizes.txt): per library variant, field
and size the NTT scopes of a wires-sized from_values batch, their ratio to 2^(L-20) times the product's 2^20-row batch of the same
call (VERDICT r5 item 1 asks for <= 1.15 at 2^21 and 2^22 rows), and the per-kernel times of the trace.
  python tools/large_sizes_table.py gpurun_out/large_sizes.txt profiles/r06_large_sizes.txt"""
import re
import sys

HEAD = re.compile(r"== (\S+) (\S+) 2\^(\d+) x (\d+) cols: IFFT ([\d.]+) ms\s+FFT\+blinding ([\d.]+) ms\s+Merkle ([\d.]+) ms\s+step ([\d.]+) ms")


def main():
    src, dst = sys.argv[1], sys.argv[2]
    blocks, cur = [], None
    for line in open(src):
        m = HEAD.match(line)
        if m:
            cur = {"v": m[1], "f": m[2], "L": int(m[3]), "cols": int(m[4]), "ifft": float(m[5]), "fft": float(m[6]), "merkle": float(m[7]),
                   "step": float(m[8]), "kernels": []}
            blocks.append(cur)
        elif cur is not None and line.startswith("   "):
            cur["kernels"].append(line.rstrip())
        elif line.startswith("FAILED"):
            blocks.append({"failed": line.strip()})
            cur = None
    base = {b["f"]: b["ifft"] + b["fft"] for b in blocks if "v" in b and b["v"] == "product" and b["L"] == 20}
    out = ["# from_values of a wires-sized batch at 2^20 .. 2^23 rows, input resident in HBM (bench.py --workload commit --steps 3), one MI355X, one gpurun call,",
           "# every block under rocprofv3 --kernel-trace.  regenerate: gpurun -- 'GB_LS_SIZES=\"20 21 22 23\" bash tools/large_sizes.sh product r05' ; python tools/large_sizes_table.py ...",
           "# product = this tree's library; r05 = tools/bin/libs/r05.so, the round-5 sources (outer radix step around the 2^20-row passes above 2^20 rows).",
           "# ratio = NTT scopes (IFFT + FFT + blinding) / (2^(L-20) x the product's 2^20-row batch of the same field): 1.000 = a 2^L-row column costs 2^(L-20) columns of 2^20 rows.",
           "# 2^21 / 2^22 rows run natively (radix-32 / radix-64 middle pass, 32-point strided LDE pass); 2^23 rows = one outer radix-2 step around two 2^22-row halves.", ""]
    out.append("%-9s %-10s %5s %9s %12s %9s %8s %9s %9s" % ("variant", "field", "rows", "IFFT ms", "FFT+blind ms", "NTT ms", "ratio", "Merkle ms", "step ms"))
    for b in blocks:
        if "failed" in b:
            out.append(b["failed"])
            continue
        ntt = b["ifft"] + b["fft"]
        b["ratio"] = ntt / (base[b["f"]] * (1 << (b["L"] - 20)))
        out.append("%-9s %-10s  2^%2d %9.2f %12.2f %9.2f %8.3f %9.2f %9.2f" % (b["v"], b["f"], b["L"], b["ifft"], b["fft"], ntt, b["ratio"], b["merkle"], b["step"]))
    out.append("")
    for b in blocks:
        if "failed" in b:
            continue
        out.append("== %-8s %-10s 2^%d x %d cols: NTT %.2f ms = %.3f x" % (b["v"], b["f"], b["L"], b["cols"], b["ifft"] + b["fft"], b["ratio"]))
        out.extend(b["kernels"])
    open(dst, "w").write("\n".join(out) + "\n")
    print("\n".join(out[:6 + 1 + len(blocks)]))


if __name__ == "__main__":
    main()
