#!/bin/bash
# The measurements behind profiles/rNN_recursion_shape.txt, one gpurun call:
#   gpurun -- 'bash tools/recursion_profile.sh'   then, locally:   python tools/recursion_profile.py gpurun_out/rs profiles/r06_recursion_shape.txt
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-$OLDPWD}"
D=gpurun_out/rs
rm -rf $D && mkdir -p $D
python tools/bench_recursion_shape.py 12 13 14 > $D/gl.log 2>&1 &&
python tools/bench_recursion_shape.py --babybear 12 13 14 > $D/bb.log 2>&1 &&
python tools/bench_recursion_shape.py --high-rate 12 > $D/gl_high_rate.log 2>&1 &&
GPU_MAX_HW_QUEUES=4 python tools/bench_recursion_shape.py --inflight 6 12 > $D/gl_inflight6_q4.log 2>&1 &&
python tools/bench_recursion_shape.py --inflight 2 12 > $D/gl_inflight2_q8.log 2>&1 &&
python tools/bench_recursion_shape.py --inflight 4 12 > $D/gl_inflight4_q8.log 2>&1 &&
python tools/bench_recursion_shape.py --inflight 6 12 > $D/gl_inflight6_q8.log 2>&1 &&
rocprofv3 --kernel-trace -d $D/tr -o p -- python3 tools/bench_recursion_shape.py --no-scopes 12 > $D/trace.log 2>&1 &&
python tools/trace_gaps.py $D/tr/p_results.db > $D/gaps.txt 2>&1
rc=$?
rm -rf $D/tr
exit $rc
