#!/bin/bash
# PC sampling (rocprofv3, stochastic, hardware-based on gfx950) of the commit workload: where the wave cycles of
# k_gl_merkle_leaves go, per instruction, with the stall reason of every sample.
#   gpurun -- 'bash tools/pc_sample_leaves.sh goldilocks'
field=${1:-goldilocks}
cols=$([ "$field" = babybear ] && echo 167 || echo 135)
out=gpurun_out/pcs_$field
rm -rf $out && mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout -k 10 400 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method ${GB_PCS_METHOD:-stochastic} --pc-sampling-unit ${GB_PCS_UNIT:-cycles} \
    --pc-sampling-interval ${GB_PCS_INTERVAL:-1048576} --kernel-trace -d $out -o pcs --output-format csv -- \
    python3 bench.py --workload commit --field $field --cols $cols --steps 3 --warmup 1 --no-cpu-baseline > $out/run.log 2>&1
echo "rocprofv3 rc=$?" >> $out/run.log
tail -5 $out/run.log
ls -la $out $out/* | head -30
for f in $(find $out -name "*pc_sampling*.csv"); do echo "== $f"; head -5 $f; wc -l $f; done
