#!/bin/bash
# Multi-rank rehearsal on the ONE GPU of a test box: real kernels, real circuits (2^20 rows), one process and one
# context per rank, gloo control plane (RCCL refuses ranks that share a device - the precondition check says so), per-rank CPU
# binding, host footprint of every rank in the line.  The pool allows at most 6 processes on the card at once, so 5 ranks (the launcher process counts as a sixth), not 8.
# Then N = 1 twice on the same box: bare `python bench.py` and through the launcher the driver uses for N > 1.
#   gpurun --timeout 1100 -- 'bash tools/rehearse_ranks.sh'
OUT=gpurun_out/rehearsal
rm -rf $OUT && mkdir -p $OUT
N=${GB_REHEARSE_RANKS:-5}
ARGS="--steps 4 --warmup 1 --no-babybear --no-inflight2 --no-cpu-baseline --no-resident"
echo "== $N ranks sharing one MI355X: GB_BENCH_SHARE_DEVICE=1 python3 bench.py --gpus $N $ARGS" | tee $OUT/summary.txt
( time GB_BENCH_SHARE_DEVICE=1 timeout -k 10 600 python3 bench.py --gpus $N $ARGS > $OUT/ranks.json 2> $OUT/ranks.err ) 2>> $OUT/summary.txt
echo "exit code $?" | tee -a $OUT/summary.txt
grep -E "rank-local|RCCL control plane" $OUT/ranks.err | tee -a $OUT/summary.txt
python3 - <<PY | tee -a $OUT/summary.txt
import json
j = json.loads([l for l in open("$OUT/ranks.json") if l.startswith("{")][-1])
print("n_gpus %d  value %.3f proofs/s aggregate  ms_per_step %.1f  value_vec_of_vecs %.3f  verified %s (%s)  golden %s" % (
    j["n_gpus"], j["value"], j["ms_per_step"], j.get("value_vec_of_vecs", 0), j["verified"], j["verified_witnesses"], j["proof_sha256_matches_golden"]))
print("control_plane", json.dumps(j["control_plane"]))
print("affinity.ranks", json.dumps(j["affinity"]["ranks"]))
for r, f in enumerate(j["ranks_host"]):
    print("rank %d host: rss %.0f MB (peak %.0f), pinned witness %.0f MB, circuit columns built in %.1f s, gb_circuit_create %.1f s" % (
        r, f["rss_mb"], f["rss_peak_mb"], f["pinned_witness_mb"], f["circuit_columns_build_s"], f["circuit_create_s"]))
PY
for mode in bare launcher bare launcher; do
    if [ $mode = bare ]; then
        timeout -k 10 300 python3 bench.py --gpus 1 --steps 10 --warmup 3 $ARGS > $OUT/n1_$mode.json 2> $OUT/n1_$mode.err
    else
        timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 $ARGS > $OUT/n1_$mode.json 2> $OUT/n1_$mode.err
    fi
    python3 - <<PY | tee -a $OUT/summary.txt
import json
j = json.loads([l for l in open("$OUT/n1_$mode.json") if l.startswith("{")][-1])
print("N = 1 %-8s value %.3f proofs/s  ms_per_step %.2f  vec_of_vecs %.3f" % ("$mode", j["value"], j["ms_per_step"], j.get("value_vec_of_vecs", 0)))
PY
done
# the branch in which RCCL DID come up, on real hardware: one rank under the launcher with the control plane forced on
GB_BENCH_FORCE_CONTROL_PLANE=1 timeout -k 10 300 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 1 --steps 6 --warmup 2 $ARGS > $OUT/n1_rccl.json 2> $OUT/n1_rccl.err
echo "forced control plane, one rank: exit code $?" | tee -a $OUT/summary.txt
python3 - <<PY | tee -a $OUT/summary.txt
import json
j = json.loads([l for l in open("$OUT/n1_rccl.json") if l.startswith("{")][-1])
print("N = 1 with an RCCL control plane: value %.3f proofs/s  control_plane %s" % (j["value"], json.dumps(j["control_plane"])))
PY
