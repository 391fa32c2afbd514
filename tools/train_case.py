import os, sys
R_="/root/repo" if os.path.isdir("/root/repo/tools") else os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0,R_); sys.path.insert(0,R_+"/oracle"); sys.path.insert(0,R_+"/tests")
import numpy as np, torch
from types import SimpleNamespace
import gat_oracle as go
from neural_spectral_codec_amd.gnn.model import SpectralGNN
from neural_spectral_codec_amd.gnn.trainer import TripletLoss
def rel(a,b): return ((a-b).abs().max()/(b.abs().max()+1e-12)).item()
L=int(sys.argv[1]) if len(sys.argv)>1 else 4
hidden=int(sys.argv[2]) if len(sys.argv)>2 else 256
n=int(sys.argv[3]) if len(sys.argv)>3 else 668
Din=int(sys.argv[4]) if len(sys.argv)>4 else 800
Dout=int(sys.argv[5]) if len(sys.argv)>5 else 800
seed=int(sys.argv[6]) if len(sys.argv)>6 else 0
torch.manual_seed(seed)
m=SpectralGNN(input_dim=Din,hidden_dim=hidden,output_dim=Dout,n_layers=L,dropout=0.0,residual=True,edge_dim=2)
go.randomize_bn_stats(m,1)
with torch.no_grad():
    for c in m.convs: c.bias.normal_(0,0.1)
m=m.to("cuda")
rng=np.random.default_rng(77)
i=np.arange(n-1)
src=np.concatenate([i,i+1,i[:-1],i[:-1]+2]); dst=np.concatenate([i+1,i,i[:-1]+2,i[:-1]])
ei=torch.from_numpy(np.stack([src,dst]).astype(np.int64))
x=torch.rand((n,Din))**4; x=x/x.sum(1,keepdim=True)
ea=torch.rand((ei.shape[1],2))
g=SimpleNamespace(x=x.cuda(),edge_index=ei.cuda(),edge_attr=ea.cuda(),num_nodes=n)
T=300
trip=np.stack([rng.integers(0,n,T) for _ in range(3)],1); tt=torch.from_numpy(trip)
Rm=torch.randn(n,Dout,generator=torch.Generator().manual_seed(0))*1e-3
f=lambda e_: go.triplet_loss_reference(e_,tt[:,0],tt[:,1],tt[:,2],0.1)+(e_*Rm.to(e_.dtype)).sum()
_,g32,gx32,l32=go.reference_gradients(m,g,f)
_,g64,gx64,l64=go.reference_gradients(m,g,f,dtype=torch.float64)
m.train(); g.x.requires_grad_(True)
emb=m(g); loss=TripletLoss(margin=0.1).forward_indexed(emb,trip[:,0],trip[:,1],trip[:,2])+(emb*Rm.cuda()).sum(); loss.backward()
params=dict(m.named_parameters())
print("loss",loss.item(),l32.item(),l64.item())
for k,ref in g64.items():
    if k not in params: continue
    got=params[k].grad.detach().cpu().double().reshape(ref.shape)
    if rel(got,ref) > 3*rel(g32[k].double(),ref) and ref.abs().max().item() > 1e-9: print(f"{k:34s} gpu {rel(got,ref):.2e}  f32ref {rel(g32[k].double(),ref):.2e}  |ref| {ref.abs().max().item():.2e}")
print(sys.argv[1:], "gx", rel(g.x.grad.cpu().double(),gx64), rel(gx32.double(),gx64))
# where the worst discrepancy sits: a single ReLU decided the other way by two float32 evaluations shows up as ONE output channel
k="convs.1.lin_src.weight" if L>1 else "convs.0.lin_src.weight"
got=params[k].grad.detach().cpu().double(); ref=g64[k]
d=(got-ref).abs()
rows=d.max(1).values; top=torch.topk(rows,4)
print("worst rows of", k, [(int(i), float(v)) for v,i in zip(top.values, top.indices)], "median row err", float(rows.median()))
kb="batch_norms.1.bias" if L>1 else "batch_norms.0.bias"
db=(params[kb].grad.detach().cpu().double()-g64[kb]).abs(); tb=torch.topk(db,4)
print("worst channels of", kb, [(int(i), float(v)) for v,i in zip(tb.values, tb.indices)], "median", float(db.median()))
