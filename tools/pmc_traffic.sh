#!/bin/bash
# HBM traffic of one kernel configuration: separate FETCH_SIZE / WRITE_SIZE passes (GPU box).
# usage: tools/pmc_traffic.sh <tag> <bench_one args...>
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 90 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/traffic_$tag/$c -o p -f csv -- python tools/bench_one.py "$@" > /dev/null 2>&1
done
python - <<PY
import csv, collections, glob
for d in sorted(glob.glob('gpurun_out/traffic_$tag/*')):
    agg=collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(d+'/p_counter_collection.csv')):
        if 'at::' in r['Kernel_Name'] or 'rocclr' in r['Kernel_Name']: continue
        agg[(r['Kernel_Name'][:50], r['Counter_Name'])][r['Dispatch_Id']] += float(r['Counter_Value'])
    for (k,c),v in agg.items():
        print(k, c, 'per-dispatch (counter units, KB):', round(list(v.values())[-1]))
    if d.endswith('FETCH_SIZE'):       # kernel durations of the same (counter) pass: last dispatch of each kernel
        last={}
        for r in csv.DictReader(open(d+'/p_kernel_trace.csv')):
            if 'at::' in r['Kernel_Name'] or 'rocclr' in r['Kernel_Name']: continue
            last[r['Kernel_Name'][:50]]=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
        for k,v in last.items(): print(k, 'DURATION_US', v)
PY
