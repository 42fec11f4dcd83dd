#!/usr/bin/env python
"""CPU numerics study (no GPU): how accurate is the IAF stack when each fp32 conv is emulated with products of bf16
splits accumulated in fp32 (what a bf16-MFMA kernel would compute)?  Max abs error vs fp64 of (m_raw, s_raw, z_new)."""
import sys, numpy as np, torch
import os; ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,'tests','golden'))
import golden_inputs as gi
from oracle import iaf_oracle as O
import torch.nn.functional as F
torch.manual_seed(0)
B,nz,nh,d,H=8,32,160,2,16
rng=np.random.RandomState(0)
params=gi.ar_multiconv2d_params(rng,nz,[nh]*d,[nz,nz])
z=rng.standard_normal((B,nz,H,H)); ctx=rng.standard_normal((B,nh,H,H))
def split(t, n):
    parts=[]; r=t.clone()
    for _ in range(n):
        p=r.to(torch.bfloat16).to(torch.float32); parts.append(p); r=r-p
    return parts
def conv_split(x, w, nparts, terms):
    # x,w fp32 tensors; products of bf16 parts computed exactly in fp32-accumulate emulation (use fp64 conv of bf16-valued operands then round to fp32 at the end)
    xs=split(x,nparts); ws=split(w,nparts)
    acc=torch.zeros(x.shape[0], w.shape[0], x.shape[2], x.shape[3], dtype=torch.float64)
    for (i,j) in terms:
        acc+=F.conv2d(xs[i].double(), ws[j].double(), padding=1)
    return acc.float()
def weights(V,g,zerodiag):
    kh,kw,ni,no=V.shape
    mask=O.get_conv_ar_mask(kh,kw,ni,no,zerodiag)
    w=O.weightnorm_weights(V,g,mask)   # HWIO
    return torch.tensor(w.transpose(3,2,0,1).astype(np.float32))
def stack(mode):
    x=torch.tensor(z.astype(np.float32)); c=torch.tensor(ctx.astype(np.float32))
    def conv(x,nm,zd):
        w=weights(params[nm+'/V'],params[nm+'/g'],zd); b=torch.tensor(params[nm+'/b'].astype(np.float32)).view(1,-1,1,1)
        if mode=='f64': return (F.conv2d(x.double(),w.double(),padding=1)+b.double())
        if mode=='f32': return F.conv2d(x,w,padding=1)+b
        if mode=='bf16': return conv_split(x,w,1,[(0,0)])+b
        if mode=='x3': return conv_split(x,w,2,[(0,0),(0,1),(1,0)])+b
        if mode=='x6': return conv_split(x,w,3,[(0,0),(0,1),(1,0),(0,2),(1,1),(2,0)])+b
    h=x if mode!='f64' else x.double()
    for i in range(d):
        h=conv(h,'layer_%d'%i,False)
        if i==0: h=h+(c if mode!='f64' else c.double())
        h=F.elu(h)
        if mode!='f64': h=h.float()
    m=conv(h,'layer_out_0',True); s=conv(h,'layer_out_1',True)
    zz=(x.double()-0.1*m.double())/torch.exp(0.1*s.double())
    return m.double(), s.double(), zz
ref=stack('f64')
for mode in ['f32','bf16','x3','x6']:
    out=stack(mode)
    print(mode, ' '.join('%.2e'%float((a-b).abs().max()) for a,b in zip(out,ref)))
