#!/bin/bash
# rocprofv3 kernel stats of 10 bare cfg-2 training steps -> gpurun_out/<tag>/train_step_kernel_stats.csv, and a plain bench line
tag=${1:-cur}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $OUT
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_t_$tag -o t -- python $GRAFT_REPO_ROOT/tools/train_steps.py 10 > /dev/null 2>&1
cp /tmp/prof_t_$tag/*kernel_stats.csv $OUT/train_step_kernel_stats.csv
cd $GRAFT_REPO_ROOT
python bench.py --no-extras --no-cpu-baseline > $OUT/bench_short.json 2> /dev/null
tail -c 600 $OUT/bench_short.json
