#!/usr/bin/env python3
"""Registers / scratch / occupancy per kernel from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
    hipcc ... -Rpass-analysis=kernel-resource-usage -c file.hip -o file.o 2> res.txt;  python tools/kernel_resources.py res.txt [filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    name = b.split()[0]
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r"\(anonymous namespace\)::", "", dem).replace("void ", "")
    dem = re.sub(r"\(.*\)$", "", dem)
    if flt and flt not in dem:
        continue
    g = lambda k: (re.search(k + r": (\d+)", b) or [None, "?"])[1]
    scr, occ, lds = g(r"ScratchSize \[bytes/lane\]"), g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")
    print(f"{dem[:100]:100s} V {g('VGPRs'):>4} A {g('AGPRs'):>4} scratch {scr:>5} occ {occ} lds {lds}")
