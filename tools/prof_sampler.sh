#!/bin/bash
# rocprofv3 kernel stats of the hipGraph denoise step (B=64) with the GroupNorm fusion off / on:  tools/prof_sampler.sh <tag>
tag=${1:-r03}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $OUT
cd /tmp
for f in ${FUSE_MODES:-0 2}; do
  rm -rf /tmp/prof_s$f
  MI_DDPM_FUSE_GN=$f timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_s$f -o s -- python $GRAFT_REPO_ROOT/tools/sample_steps.py 40 > $OUT/denoise_fuse$f.txt 2>/dev/null
  cp /tmp/prof_s$f/*kernel_stats.csv $OUT/denoise_fuse${f}_kernel_stats.csv
done
