import sys, os, types, time, ctypes as C, numpy as np, torch
os.environ["QBNN_LIB_OVERRIDE"] = os.path.abspath(os.environ.get("QBNN_STAMP_LIB", "tools/_build/libqbnn_STAMP%s.so" % os.environ.get("STAMPW","0")))
sys.path.insert(0,'.'); sys.path.insert(0,'tests/golden')
from fixtures import load_golden
import quantised_bayesian_nets_amd as q
from quantised_bayesian_nets_amd import _lib
from quantised_bayesian_nets_amd.layers import MCQTensor
from quantised_bayesian_nets_amd.models import run_identity_chain
g=load_golden('resnet_bbb_a7w8.npz')
args=types.SimpleNamespace(activation_precision=7, weight_precision=8)
m=q.ModelFactory.get_model('conv_resnet_bbb',[1,3,32,32],10,True,args).load_reference_state(g['state'])
S,B=100,256
which = sys.argv[1]
def rnd(shape): return torch.randint(0,128,shape,dtype=torch.uint8,device='cuda')
cases = {'c3': ([m.layers[5][1]], (S,B,8,8,96), 8), 'c4': ([m.layers[6][1]], (S,B,4,4,192), 16)}
blocks, xs, G = cases[which]
x = MCQTensor(rnd(xs), 0.05, 60)
dbg = torch.zeros(64, dtype=torch.int64, device='cuda')
L=_lib.lib(); L.qbnn_debug_stamp_buffer.argtypes=[C.c_void_p]
with q.mc_context(S, 3, 0):
    run_identity_chain(blocks, x); torch.cuda.synchronize()
    L.qbnn_debug_stamp_buffer(C.c_void_p(dbg.data_ptr()))
    run_identity_chain(blocks, x); torch.cuda.synchronize()
d = dbg.cpu().numpy().reshape(8,8).astype(np.float64)
names=['M_a (slabs)','barrier','E_a','M_b (slabs)','barrier','E_b','barrier','copy-out + next tile']
n_items = S*B/G
print(which, 'cycles per item (s_memtime ticks = 100 MHz? see ratio), waves 0 / 3 / 7')
for i,n in enumerate(names): print('%-22s %9.0f %9.0f %9.0f' % (n, d[0,i]/n_items, d[3,i]/n_items, d[7,i]/n_items))
print('total', d[0].sum()/n_items)
buf = (C.c_ulonglong*4)()
L.qbnn_debug_read_inner(buf)
v = np.array(list(buf), dtype=np.float64)
nconv = 2*n_items
print('inner (wave %s): per conv: slab-barrier wait %.0f   K loops %.0f cycles' % (os.environ.get("STAMPW","0"), v[0]/nconv, v[1]/nconv))
