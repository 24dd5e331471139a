#!/usr/bin/env python3
"""Epilogue ablation of the row-space grouped GEMM at the headline shape (balanced routing): the same launch with GELU / ReLU /
bias-only / plain epilogues (forward) and GELU / ReLU / plain activation-gradient epilogues.  usage: python tools/epilogue_ab.py"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
from competesmoe_amd import ops, _lib as L
T,K,E,D,F=32768,2,64,4096,11008
M=T*K; dev="cuda"; bf=torch.bfloat16
g=torch.Generator(device=dev).manual_seed(0)
counts=torch.full((E,),M//E,dtype=torch.int64); off=torch.zeros(E+1,dtype=torch.int32); off[1:]=counts.cumsum(0); off=off.to(dev)
xs=torch.randn(M,D,device=dev,generator=g).to(bf); h=torch.randn(M,F,device=dev,generator=g).to(bf)
W1=(torch.randn(E,F,D,device=dev,generator=g)*0.02).to(bf); W2=(torch.randn(E,D,F,device=dev,generator=g)*0.02).to(bf)
b1=torch.zeros(E,F,device=dev,dtype=bf); ar=torch.arange(E,device=dev,dtype=torch.int64)
p1=W1.data_ptr()+ar*(F*D*2); p2=W2.data_ptr()+ar*(D*F*2); pb1=b1.data_ptr()+ar*(F*2)
def t(fn,n=5):
    fn(); torch.cuda.synchronize(); s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/n
for name,fn in [
 ("nt1 bias+GELU 2 outputs", lambda: ops.grouped_gemm(xs,p1,L.B_NK,D,F,off,E,bias_ptrs=pb1,epilogue=L.EPI_BIAS_ACT,act=L.ACT_GELU,want_c2=True)),
 ("nt1 bias+ReLU 2 outputs", lambda: ops.grouped_gemm(xs,p1,L.B_NK,D,F,off,E,bias_ptrs=pb1,epilogue=L.EPI_BIAS_ACT,act=L.ACT_RELU,want_c2=True)),
 ("nt1 bias only 1 output ", lambda: ops.grouped_gemm(xs,p1,L.B_NK,D,F,off,E,bias_ptrs=pb1,epilogue=L.EPI_BIAS)),
 ("nt1 plain              ", lambda: ops.grouped_gemm(xs,p1,L.B_NK,D,F,off,E)),
 ("nn1 ACTGRAD GELU       ", lambda: ops.grouped_gemm(xs,p2,L.B_KN,F,F,off,E,epilogue=L.EPI_ACTGRAD,act=L.ACT_GELU,aux=h)),
 ("nn1 ACTGRAD ReLU       ", lambda: ops.grouped_gemm(xs,p2,L.B_KN,F,F,off,E,epilogue=L.EPI_ACTGRAD,act=L.ACT_RELU,aux=h)),
 ("nn1 plain              ", lambda: ops.grouped_gemm(xs,p2,L.B_KN,F,F,off,E)),
]:
    print(name, round(t(fn),3),"ms",flush=True)
