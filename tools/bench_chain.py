import sys, types, time, numpy as np, torch
sys.path.insert(0,'.'); sys.path.insert(0,'tests/golden')
from fixtures import load_golden
import quantised_bayesian_nets_amd as q
from quantised_bayesian_nets_amd.layers import MCQTensor
from quantised_bayesian_nets_amd.models import run_identity_chain
g=load_golden('resnet_bbb_a7w8.npz')
args=types.SimpleNamespace(activation_precision=7, weight_precision=8)
m=q.ModelFactory.get_model('conv_resnet_bbb',[1,3,32,32],10,True,args).load_reference_state(g['state'])
S,B=100,256
which = sys.argv[1] if len(sys.argv)>1 else 'c1'
reps = int(sys.argv[2]) if len(sys.argv)>2 else 5
def rnd(shape): return torch.randint(0,128,shape,dtype=torch.uint8,device='cuda')
cases = {'c1': (list(m.layers[3]), (S,B,32,32,24)), 'c1a': ([m.layers[3][0]], (S,B,32,32,24)), 'c2': ([m.layers[4][1]], (S,B,16,16,48)),
         'c3': ([m.layers[5][1]], (S,B,8,8,96)), 'c4': ([m.layers[6][1]], (S,B,4,4,192))}
blocks, xs = cases[which]
x = MCQTensor(rnd(xs), 0.05, 60)
with q.mc_context(S, 3, 0):
    for i in range(reps+2):
        if i==2: torch.cuda.synchronize(); t=time.perf_counter()
        y = run_identity_chain(blocks, x)
    torch.cuda.synchronize(); print(which, (time.perf_counter()-t)/reps*1e3, 'ms per launch (incl. weight sampling)')
