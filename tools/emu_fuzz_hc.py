"""Fuzz of the precomputed-table LZ4HC kernels (lz4hip_hc_nat.hpp / lz4hip_hc_lcp.hpp) under the SIMT emulator against the oracle:
blocks that stress the exactness arguments -- tiny alphabets, runs of period 1-5 between junk, copies of earlier content, fuzzer-style
and record-like rows, long runs with single disturbed bytes; 192 blocks per round, one in eight close to 64 KiB.
usage: python tools/emu_fuzz_hc.py <seed> <rounds> [nat|lcp]     (TEST INFRASTRUCTURE: needs tests/simt and oracle/)"""
import os, sys, numpy as np, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import emu_helpers as emu
from oracle.oracle import Oracle
o=Oracle()
seed=int(sys.argv[1]); rounds=int(sys.argv[2]); kind=sys.argv[3] if len(sys.argv)>3 else 'lcp'
rng=np.random.default_rng(seed)
total=bad=0
for r in range(rounds):
    blocks=[]
    for i in range(192):
        mode=int(rng.integers(0,6))
        sz=int(rng.integers(13,6000)) if rng.integers(0,8) else int(rng.integers(30000,65537))
        if mode==0:   # tiny alphabet
            row=rng.integers(0,int(rng.integers(2,4)),sz).astype(np.uint8)
        elif mode==1: # runs of periods 1..5 with random junk
            row=rng.integers(0,256,sz).astype(np.uint8); pos=0
            while pos<sz:
                per=int(rng.integers(1,6)); ln=int(rng.integers(4,300))
                pat=rng.integers(0,3,per).astype(np.uint8)
                seg=np.tile(pat,ln//per+2)[:ln]; e=min(sz,pos+ln); row[pos:e]=seg[:e-pos]; pos=e+int(rng.integers(0,12))
        elif mode==2: # markov-ish repeats of earlier content
            row=rng.integers(0,8,sz).astype(np.uint8); pos=64
            while pos<sz-8:
                ln=int(rng.integers(4,80)); src=int(rng.integers(0,pos)); e=min(sz,pos+ln)
                for j in range(pos,e): row[j]=row[src+(j-pos)] if src+(j-pos)<j else row[j]
                pos=e+int(rng.integers(0,6))
        elif mode==3:
            row=o.gen(2,seed*131+r,i,1).reshape(-1)[:sz].copy()
        elif mode==4:
            row=o.gen(3,seed*131+r,i,1).reshape(-1)[:sz].copy()
            if rng.integers(0,2): row[:sz//3]=row[0]
        else:         # long runs with single-byte disturbances
            row=np.full(sz,int(rng.integers(0,256)),np.uint8)
            for _ in range(int(rng.integers(0,40))): row[int(rng.integers(0,sz))]=int(rng.integers(0,256))
        blocks.append(row)
    res,dst=emu.encode(blocks,hc=True,groups=1,**{kind:True})
    for i,a in enumerate(blocks):
        want=o.compress(a,hc=True); total+=1
        if res[i]!=len(want) or not np.array_equal(dst[i,:res[i]],want):
            bad+=1
            if bad<6:
                print("MISMATCH seed",seed,"round",r,"block",i,"size",a.size,res[i],len(want),flush=True)
                np.save(f"/tmp/hc_fuzz_bad_{kind}_{seed}_{r}_{i}.npy",a)
    print("seed",seed,"round",r,"total",total,"bad",bad,flush=True)
