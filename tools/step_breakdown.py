#!/usr/bin/env python3
"""Group a rocprofv3 kernel_stats.csv of tools/train_steps.py N into ms/step per kernel family."""
import csv, sys
path, steps = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
rows = list(csv.DictReader(open(path)))


def fam(n):
    if 'conv3x3_halo_kernel' in n:
        return 'halo 3x3 fwd+dgrad' if n.split('<')[1].split(',')[2].strip() == '3' else 'halo 1x1 fwd+dgrad'
    if 'wgrad3x3_kernel' in n:
        return 'wgrad3x3 (3x3)' if n.split('<')[1].split(',')[1].strip() == '3' else 'wgrad3x3 (1x1)'
    for key, name in (('wgrad_reduce', 'wgrad reduce'), ('igemm_fast', 'stride-2/transposed conv'), ('wgrad_fast', 'stride-2/transposed wgrad'),
                      ('igemm_kernel', 'generic fp32 GEMMs'), ('wgrad_kernel', 'generic fp32 GEMMs'), ('gn_mish_fwd', 'GN+Mish fwd'),
                      ('gn_mish_bwd', 'GN+Mish bwd'), ('linattn_fwd', 'linattn fwd'), ('linattn_bwd', 'linattn bwd'), ('chan_ln', 'LayerNorm'),
                      ('small_c', '3-channel ends'), ('partial_sum', '3-channel ends'), ('adam', 'Adam'), ('pack_weights', 'weight pack'),
                      ('colsum', 'colsum')):
        if key in n:
            return name
    return 'memset/fill' if 'fill' in n.lower() else 'other'


grp = {}
for r in rows:
    k = fam(r['Name']); grp[k] = grp.get(k, 0) + float(r['TotalDurationNs']) / 1e6 / steps
print(f"total {sum(grp.values()):.3f} ms/step")
for k, v in sorted(grp.items(), key=lambda kv: -kv[1]):
    print(f"{v:6.3f}  {k}")
