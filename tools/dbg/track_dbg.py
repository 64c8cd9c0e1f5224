import sys; sys.path.insert(0, '.')
import numpy as np, torch
from vidseg_diffusion_amd import synthetic, analysis as A, _lib
from vidseg_diffusion_amd._lib import call, ptr, stream
from oracle import analysis as O
F,h,w,C = 6,12,10,64; N=h*w
blocks,_ = synthetic.attention_q_dumps(F,h,w,C,num_blocks=1,seed=31)
dev = torch.device('cuda:0')
cond = torch.from_numpy(blocks[0][F:]).to(dev)
nb = N//500+1
normed = torch.empty((nb,F*N,C),dtype=torch.float16,device=dev)
call("vidseg_track_normalize", ptr(cond), F*N, C, nb, ptr(normed), stream())
ref = O._l2norm_rows16(blocks[0][F:].reshape(F*N,C))
print("normed v0 eq", np.array_equal(normed[0].cpu().numpy(), ref))
th,tw = O.dense_tracking(blocks[0],F,h,w); oidx = th*w+tw
idx,ties = A.dense_tracking(cond,F,h,w)
g = idx.cpu().numpy()
for f in range(F): print(f, np.mean(g[f]==oidx[f]))
print("ties", ties.item())
# blend map for f=0 with cur=arange
f=2
cur = torch.from_numpy(oidx[f].astype(np.int32)).to(dev)
blend = torch.empty((N,N),dtype=torch.float16,device=dev); nxt = torch.empty(N,dtype=torch.int32,device=dev); t=torch.zeros(1,dtype=torch.int32,device=dev)
call("vidseg_track_step", ptr(normed), F, N, w, C, f, 500, ptr(cur), 1, ptr(blend), ptr(nxt), ptr(t), stream())
s = ref.reshape(F,N,C)[f][oidx[f]]; trg = ref.reshape(F,N,C)[f+1]; aux = ref.reshape(F,N,C)[0]
cos = (s.astype(np.float64)@trg.astype(np.float64).T).astype(np.float16); ca=(s.astype(np.float64)@aux.astype(np.float64).T).astype(np.float16)
bl = (O._half_mul(f/(f+1),cos).astype(np.float32)+O._half_mul(1/(f+1),ca).astype(np.float32)).astype(np.float16)
b = blend.cpu().numpy()
print("blend eq", np.array_equal(b,bl), np.mean(b!=bl), np.abs(b.astype(np.float32)-bl.astype(np.float32)).max())
print("next eq", np.mean(nxt.cpu().numpy()==oidx[f+1]))
bad = np.nonzero(nxt.cpu().numpy()!=oidx[f+1])[0]
for i in bad[:6]:
    row = bl[i]; print(i, "gpu", nxt[i].item(), "or", oidx[f+1][i], "nmax", (row==row.max()).sum(), np.nonzero(row==row.max())[0])
