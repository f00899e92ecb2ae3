import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from supertonic_amd import binding
from supertonic_amd.arch import tiny_arch
def bf16bits(x):
    u=np.ascontiguousarray(x,np.float32).view(np.uint32).astype(np.uint64); u=(u+0x7FFF+((u>>16)&1))>>16
    return u.astype(np.uint16)
e=binding.Engine(0,"bf16"); e.load_synthetic(tiny_arch(),7)
rng=np.random.default_rng(0)
M,C,I=128,512,512
W1=(rng.standard_normal((I,C))).astype(np.float32); W2=(rng.standard_normal((C,I))).astype(np.float32)
xr=rng.standard_normal((M,C)).astype(np.float32)
got=e.op_ffn(xr,W1,np.zeros(I,np.float32),W2,None,None,np.zeros((M,C),np.float32),fused=True)
ring=got.view(np.uint16).reshape(-1)[:4*32768//2].reshape(4,32,64,8)   # stage, piece, lane, elem
w1b=bf16bits(W1).reshape(I,C); w2b=bf16bits(W2).reshape(C,I)
def w1_stage(t):
    out=np.zeros((32,64,8),np.uint16)
    for s in range(32):
        for lane in range(64):
            lr,lh=lane&31,lane>>5
            out[s,lane]=w1b[32*t+lr,16*s+8*lh:16*s+8*lh+8]
    return out
def w2_stage(t):
    out=np.zeros((32,64,8),np.uint16)
    for nt in range(16):
        for s in range(2):
            for lane in range(64):
                r,hf=lane&31,lane>>5
                base=32*t+16*s+4*hf
                out[nt*2+s,lane,:4]=w2b[32*nt+r,base:base+4]; out[nt*2+s,lane,4:]=w2b[32*nt+r,base+8:base+12]
    return out
exp=[w1_stage(0),w1_stage(1),w2_stage(0),w1_stage(2)]
for st in range(4):
    bad=[p for p in range(32) if not np.array_equal(ring[st,p],exp[st][p])]
    print("stage",st,"bad pieces",bad)
    for p in bad[:3]:
        print("   piece",p,"got",ring[st,p,0],"exp",exp[st][p,0], "all zero?", not ring[st,p].any())
