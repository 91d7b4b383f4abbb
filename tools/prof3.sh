#!/bin/bash
# rocprofv3 kernel stats of 10 cfg-2 training steps, 10 cfg-3 (B=32) steps and 40 denoise steps -> gpurun_out/<tag>/
tag=${1:-cur}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/p_t -o t -- python $GRAFT_REPO_ROOT/tools/train_steps.py 10 > /dev/null 2>&1
cp /tmp/p_t/*kernel_stats.csv $OUT/train_step_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/p_c -o c -- python $GRAFT_REPO_ROOT/tools/train_steps_cfg3.py 10 32 > /dev/null 2>&1
cp /tmp/p_c/*kernel_stats.csv $OUT/cfg3_b32_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/p_s -o s -- python $GRAFT_REPO_ROOT/tools/sample_steps.py 40 > /dev/null 2>&1
cp /tmp/p_s/*kernel_stats.csv $OUT/denoise_kernel_stats.csv
ls -la $OUT
