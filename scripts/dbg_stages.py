import faulthandler, sys, time, os
faulthandler.dump_traceback_later(100, repeat=True, file=sys.stderr)
sys.path.insert(0, os.getcwd())
import torch
from pointnet_refine_amd.model import LineRefineNet
from pointnet_refine_amd.synth import synthetic_batch
from pointnet_refine_amd.train_step import TrainStep
def log(*a):
    print(f"[{time.time()-T0:7.2f}]", *a, flush=True)
T0=time.time()
dev=torch.device('cuda',0)
m=LineRefineNet().to(dev).train(); log('model')
ctx,noisy,target=synthetic_batch(64,1024,dev); torch.cuda.synchronize(); log('data')
mem=m.encode_context(ctx); torch.cuda.synchronize(); log('encode_context')
tgt=m.encode_line(noisy); torch.cuda.synchronize(); log('encode_line')
pm=m.pos_emb(ctx[:,:,:3]); torch.cuda.synchronize(); log('pos_emb')
out=m.decode(ctx,noisy,mem,tgt); torch.cuda.synchronize(); log('decode')
loss=(out-target.unsqueeze(0)).abs().mean(); loss.backward(); torch.cuda.synchronize(); log('backward', float(loss))
opt=torch.optim.Adam(m.parameters(),lr=1e-3)
st=TrainStep(m,opt,decoder_chunk=32)
for i in range(3):
    l=st(ctx,noisy,target); torch.cuda.synchronize(); log('step',i,float(l))
