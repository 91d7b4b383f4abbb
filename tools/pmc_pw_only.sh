#!/bin/bash
tag=r05; export TMPDIR=/tmp; OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag; mkdir -p $OUT; cd $GRAFT_REPO_ROOT
for cfg in "pw 128 8 512 512 bf16" "pw 128 32 128 128 bf16" "pw 128 16 256 256 bf16" "fusedpw 128 32 128 128 bf16" "fusedpw 128 32 128 128 fp32" "gtdown 128 32 128 128 bf16" "gtup 128 16 128 128 bf16" "c1x1 128 32 128 384 bf16"; do
  name=$(echo $cfg | tr ' ' '_')
  timeout 400 bash tools/pmc_traffic.sh ${tag}_$name $cfg > $OUT/pmc_traffic_$name.txt 2>&1
done
python -m pytest tests/test_kernels_gpu.py -q -k "pw or fused or sums or tap_gather" 2>&1 | grep -E "passed|failed" | tail -2
