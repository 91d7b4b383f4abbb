#!/usr/bin/env python3
"""Idle time inside the replayed training step: reads a rocprofv3 --kernel-trace CSV (start / end timestamps per dispatch) and reports,
for the steady-state steps, busy time (union of kernel intervals), gaps and the largest gaps with the kernels on either side."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda e: e[0])
# the timed, replayed steps: optimizer launches 12 .. 40 of a `bench.py --steps 40 --warmup 5` run (3 eager + capture + warm-up come first,
# the sampler and the per-kernel probe passes after)
ad = [i for i, e in enumerate(ev) if "adam_dev_kernel" in e[2] or "adam_kernel" in e[2]]
lo, hi = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (12, 40)
ev = ev[ad[lo] + 1:ad[hi] + 1]
span = ev[-1][1] - ev[0][0]
busy, cur_end, gaps = 0, ev[0][0], []
prev = None
for s, e, n in ev:
    if s > cur_end:
        gaps.append((s - cur_end, prev, n)); busy += e - s; cur_end = e
    else:
        if e > cur_end: busy += e - cur_end; cur_end = e
    if prev is None or e >= cur_end: prev = n
nadam = sum(1 for e in ev if "adam_dev_kernel" in e[2] or "adam_kernel" in e[2])
print(f"{len(ev)} dispatches, {nadam} steps, span {span/1e3:.0f} us, busy {busy/1e3:.0f} us ({100*busy/span:.1f} %), idle per step {(span-busy)/1e3/max(nadam,1):.1f} us")
gaps.sort(key=lambda g: -g[0])
from collections import Counter
c = Counter()
for g, a, b in gaps: c[(a[:50] if a else None, b[:50])] += g
for (a, b), g in c.most_common(15):
    print(f"  {g/1e3/max(nadam,1):7.2f} us/step   {a}  ->  {b}")
