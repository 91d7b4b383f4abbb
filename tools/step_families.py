import csv,sys,re,collections
rows=list(csv.DictReader(open(sys.argv[1])))
steps=float(sys.argv[2]) if len(sys.argv)>2 else 10
fam=collections.defaultdict(float)
def f(n):
    if 'conv_pw_kernel' in n: return '3x3 conv_pw'
    if 'conv3x3_halo' in n: return '3x3 halo'
    if 'wgrad_tr' in n: return 'wgrad3x3'
    if 'conv1x1_pw' in n: return '1x1 conv'
    if 'wgrad1x1' in n: return 'wgrad1x1'
    if 'gn_mish_fwd' in n: return 'gn fwd'
    if 'gn_mish_bwd' in n: return 'gn bwd'
    if 'chan_ln' in n: return 'chan_ln'
    if 'linattn' in n: return 'linattn'
    if 'igemm' in n or 'conv_gt' in n: return 'igemm+gt (s2/convT)'
    if 'wgrad_s2' in n: return 'wgrad s2'
    if 'adam' in n: return 'adam'
    if 'colsum' in n or 'f32_to_bf16' in n: return 'colsum/cvt'
    if 'small_c' in n: return '3ch ends'
    if 'pack' in n: return 'pack'
    if 'small_gemm' in n: return 'time mlp'
    if 'wgrad' in n: return 'other wgrad'
    return 'rest'
for r in rows: fam[f(r['Name'])]+=float(r['TotalDurationNs'])/steps/1e6
for k,v in sorted(fam.items(),key=lambda x:-x[1]): print(f'{k:20s} {v:.3f}')
print('total',sum(fam.values()))
