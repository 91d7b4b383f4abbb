import os, sys
sys.argv=[sys.argv[0],'100']
ROOT='/root/repo'
sys.path[:0]=[ROOT, ROOT+'/image-generation-models_amd']
from src.ops import functional as K
t=int(os.environ.get('PW_TILE','0'))
if t: K.load_library().mi_debug_conv_pw_tile(t)
exec(open(ROOT+'/tools/sample_steps.py').read())
