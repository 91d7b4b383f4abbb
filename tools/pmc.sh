#!/bin/bash
# usage: tools/pmc.sh <tag> <bench_one args...>   -- runs several rocprofv3 --pmc passes (GPU box)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 90 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmc_$tag/s$i -o p -f csv -- python tools/bench_one.py "$@" > /dev/null 2>&1
done
python - <<PY
import csv, collections, glob
for d in sorted(glob.glob('gpurun_out/pmc_$tag/s*')):
    rows=list(csv.DictReader(open(d+'/p_counter_collection.csv')))
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); names={}
    for r in rows:
        agg[(r['Kernel_Name'][:60], r['Counter_Name'])][r['Dispatch_Id']] += float(r['Counter_Value'])
    for (k,c),v in sorted(agg.items()):
        if 'at::' in k or 'rocclr' in k: continue
        if c in ('SQ_BUSY_CYCLES',): continue
        vals=list(v.values())
        print(k[:48], c, round(vals[-1]))
    for r in csv.DictReader(open(d+'/p_kernel_trace.csv')):
        if 'at::' in r['Kernel_Name'] or 'rocclr' in r['Kernel_Name']: continue
        print('   dur_us', (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3); break
PY
