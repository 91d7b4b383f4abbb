#!/bin/bash
# Everything profiles/ needs for one round, run on the GPU box:  tools/profile_round.sh <tag>
#  1. rocprofv3 --kernel-trace --stats of the default bench.py command (+ its JSON line)
#  2. the same over 10 bare training steps (per-kernel ms/step), and over cfg 3 at B=32
#  3. PMC passes (separate runs, never combined with trace domains other than --kernel-trace): HBM traffic (FETCH_SIZE /
#     WRITE_SIZE) of the dominant kernels on their cfg-2 shapes and the SQ / TCC sets of tools/pmc.sh
tag=${1:-r05}
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $OUT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_b -o b -- python $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $OUT/bench_stdout.txt 2>/dev/null
cp /tmp/prof_b/*kernel_stats.csv $OUT/bench_kernel_stats.csv
tail -1 $OUT/bench_stdout.txt > $OUT/bench_under_rocprof.json
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_t -o t -- python $GRAFT_REPO_ROOT/tools/train_steps.py 10 > /dev/null 2>&1
cp /tmp/prof_t/*kernel_stats.csv $OUT/train_step_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_c -o c -- python $GRAFT_REPO_ROOT/tools/train_steps_cfg3.py 10 32 > /dev/null 2>&1
cp /tmp/prof_c/*kernel_stats.csv $OUT/cfg3_b32_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --stats -f csv -d /tmp/prof_s -o s -- python $GRAFT_REPO_ROOT/tools/sample_steps.py 40 > /dev/null 2>&1
cp /tmp/prof_s/*kernel_stats.csv $OUT/denoise_kernel_stats.csv
cd $GRAFT_REPO_ROOT
for cfg in "pw 128 8 512 512 bf16" "pw 128 32 128 128 bf16" "pw 128 16 256 256 bf16" "fusedpw 128 32 128 128 bf16" "fusedpw 128 32 128 128 fp32" "gtdown 128 32 128 128 bf16" "gtup 128 16 128 128 bf16" "halo 128 32 128 128 fp32" "wgrad 128 32 128 128 bf16" "wgrad 128 8 512 512 bf16" "wgradq 128 0 0 0 bf16" "fused 128 32 128 128 bf16" "fused 128 32 128 128 fp32" "gn 128 32 128 128 bf16" "ln 128 32 128 128 bf16" "attn 128 32 128 128 bf16" "c1x1 128 32 128 384 bf16"; do
  name=$(echo $cfg | tr ' ' '_')
  timeout 400 bash tools/pmc_traffic.sh ${tag}_$name $cfg > $OUT/pmc_traffic_$name.txt 2>&1
done
for cfg in "pw 128 8 512 512 bf16" "pw 128 32 128 128 bf16" "wgradq 128 0 0 0 bf16"; do
  name=$(echo $cfg | tr ' ' '_')
  timeout 400 bash tools/pmc.sh ${tag}_$name $cfg > $OUT/pmc_sq_$name.txt 2>&1
done
python bench.py > $OUT/bench.json 2> /dev/null
python bench.py --cfg 3 --batch 32 --graph 1 --no-extras > $OUT/bench_cfg3_b32_graph.json 2> /dev/null
python bench.py --cfg 3 --batch 32 --graph 0 --no-extras > $OUT/bench_cfg3_b32_eager.json 2> /dev/null
# the RCCL path on one rank (init, broadcast, bucketed all-reduce on RCCL's stream, barrier-bracketed timing), eager and as the segmented graph
for g in 0 1; do MI_BENCH_FORCE_DIST=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2951$g bench.py --gpus 1 --steps 30 --warmup 5 --no-extras --graph $g 2> /dev/null | tail -1 > $OUT/bench_rccl1_graph$g.json; done
ls -la $OUT
