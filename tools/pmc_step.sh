#!/bin/bash
# Per-kernel PMC counters over whole training steps (GPU box): tools/pmc_step.sh <tag> "<COUNTER ...>" [steps]
# One rocprofv3 --pmc pass (kernel-trace only) over tools/train_steps.py; prints per kernel symbol the counter sums per step.
tag=$1; set=$2; steps=${3:-3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/pmcstep_$tag -o p -f csv -- python tools/train_steps.py $steps > /dev/null 2>&1
python - <<PY
import csv, collections, re
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
seen=set()
for r in csv.DictReader(open('gpurun_out/pmcstep_$tag/p_counter_collection.csv')):
    k=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name'])
    k=re.sub(r'^void ','',k).split('(')[0][:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if (r['Dispatch_Id'],) not in seen: seen.add((r['Dispatch_Id'],)); cnt[k]+=1
dur=collections.defaultdict(float)
for r in csv.DictReader(open('gpurun_out/pmcstep_$tag/p_kernel_trace.csv')):
    k=re.sub(r'\(anonymous namespace\)::','',r['Kernel_Name'])
    k=re.sub(r'^void ','',k).split('(')[0][:60]
    dur[k]+=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
names=sorted({c for v in agg.values() for c in v})
print('kernel'.ljust(62), 'launches', 'total_us', *names)
for k,v in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
    print(k.ljust(62), cnt[k], round(dur[k],1), *[int(v.get(c,0)) for c in names])
PY
