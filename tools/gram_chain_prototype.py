#!/usr/bin/env python3
"""numpy prototype of the two-pass form of the reference's fp32 Gram chains (csrc/anderson.hip gram_round_kernel + gram_chain_apply_kernel) on a
residual history of the loop itself (gpurun_out/real_history_m2.npz: `DUMP=96 python tools/gram_on_real_history.py` on the GPU box): blocks of
128 terms rounded for the binade predicted from the block sums (+- 1), accepted while the running sum provably stays inside the binade, walked
term by term otherwise - bit-equal to the sequential chain on every chain tried, ~16 of 256 blocks walked."""
import numpy as np
import os
d=np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gpurun_out', 'real_history_m2.npz'))
G=d['G96']
N=G.shape[1]; BLK=2048; NB=N//BLK; T=BLK//16
def binade(x): return int(np.floor(np.log2(x)))
def serial(a,b,S):
    for i in range(len(a)):
        S=np.float32(np.longdouble(a[i])*np.longdouble(b[i])+np.longdouble(S))
    return S
tot_fb=0; bad=0
for (i,j) in ((0,0),(0,1),(2,3),(4,4),(1,4)):
    a_all=G[i].reshape(NB,T,16); b_all=G[j].reshape(NB,T,16)
    # K4's block totals (float32-ish) -> predicted prefix per chain
    btot=(a_all.astype(np.float64)*b_all).sum(axis=(1,2))
    pref=np.concatenate([[0],np.cumsum(btot)])[:-1]/16
    for c in range(16):
        S=np.float32(0); fb=0; miss=0
        for blk in range(NB):
            a=a_all[blk,:,c]; b=b_all[blk,:,c]
            ok=False
            if S>0 and pref[blk]>0:
                e=binade(S); ep=binade(pref[blk])
                if abs(e-ep)<=1:
                    u=2.0**(e-23)
                    p=a.astype(np.float64)*b.astype(np.float64)
                    q=p/u; r=np.rint(q)
                    tie=np.any(np.abs(q-np.floor(q))==0.5)
                    big=np.any(np.abs(p)>=0.25*2.0**e)
                    A=np.abs(r).sum(); Nn=r.sum()
                    su=float(S)/u
                    if (not tie) and (not big) and su-A>=2**23+2 and su+A<=2**24-2:
                        S=np.float32(float(S)+Nn*u); ok=True
                else: miss+=1
            if not ok:
                S=serial(a,b,S); fb+=1
        tot_fb+=fb
        eq = (S==serial(G[i].reshape(-1,16)[:,c],G[j].reshape(-1,16)[:,c],np.float32(0)))
        bad+= (not eq)
        if c<2: print((i,j),c,'fallback blocks',fb,'mispredicted',miss,'equal',eq)
print('total chains bad',bad,'avg fallbacks',tot_fb/(5*16))
