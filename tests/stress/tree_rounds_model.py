"""CPU model of the encoder's round-wise tree build (csrc/kernels/tree.hpp, tree_fast_wave): all items
below rate(a) + rate(b) are merged pairwise in key order per round.  Compared with the oracle's trees
on random shapes; also prints a cost model of the hybrid (sorted round vs single merge).
usage: python tests/stress/tree_rounds_model.py"""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle.oracle import Oracle
o=Oracle()
def ref_tree(data):
    s=o.encode(data,0)
    tl=int(np.frombuffer(s[8:10].tobytes(),dtype='<i2')[0])
    return np.frombuffer(s[10:10+2*tl].tobytes(),dtype='<i2').tolist()
def rounds_tree(hist):
    items=[(int(hist[i])<<9)|(511-i) for i in range(256) if hist[i]]
    node=256; left={}; right={}; nrounds=0
    while True:
        items.sort()
        if len(items)==1:
            left[node]=511-(items[0]&511); root=node; node+=1; break
        nrounds+=1
        thr=(items[0]>>9)+(items[1]>>9)
        sel=sum(1 for x in items if (x>>9)<thr)
        pairs=sel//2
        new=[]
        for p in range(pairs):
            a,b=items[2*p],items[2*p+1]
            left[node+p]=511-(a&511); right[node+p]=511-(b&511)
            new.append((((a>>9)+(b>>9))<<9)|(511-(node+p)))
        items=new+items[2*pairs:]
        node+=pairs
    out=[]
    sys.setrecursionlimit(10000)
    def ser(n):
        out.append(n)
        if n<256: out.extend([-1,-1]); return
        ser(left[n]) if n in left else out.append(-1)
        ser(right[n]) if n in right else out.append(-1)
    ser(root)
    return out,nrounds
rng=np.random.default_rng(1)
bad=0; mr=0
from libhuffman_amd import datagen
for trial in range(1500):
    k=int(rng.integers(1,257))
    syms=rng.choice(256,size=k,replace=False)
    mode=trial%6
    if mode==0: w=np.ones(k,dtype=np.int64)
    elif mode==1: w=rng.integers(1,4,size=k)
    elif mode==2: w=rng.integers(1,200,size=k)
    elif mode==3: w=(2**rng.integers(0,6,size=k))
    elif mode==4: w=np.array([1]*(k//2)+[2]*(k-k//2))
    else:
        f=[1,1]
        while len(f)<k and f[-1]<3000: f.append(f[-1]+f[-2])
        w=np.array((f+[1]*k)[:k])
    data=np.repeat(syms.astype(np.uint8),w)
    rng.shuffle(data)
    hist=np.bincount(data,minlength=256)
    t1=ref_tree(data); t2,nr=rounds_tree(hist); mr=max(mr,nr)
    if t1!=t2:
        bad+=1
        if bad<3: print("MISMATCH",k,mode,t1[:12],t2[:12])
print("bad",bad,"max rounds",mr)
for kind in ("zipf255","uniform256","uniform255","logtext"):
    d=datagen.GENERATORS[kind](65536 if kind!="logtext" else 1<<20)
    h=np.bincount(d,minlength=256); t2,nr=rounds_tree(h); print(kind,"rounds",nr, "match", t2==ref_tree(d))

def cost(hist, par_min, c_par=(560,3300), c_seq=(58,520)):
    items=sorted((int(hist[i])<<9)|(511-i) for i in range(256) if hist[i])
    node=256; ins=cyc=0; npar=nseq=0
    while len(items)>1:
        thr=(items[0]>>9)+(items[1]>>9)
        sel=sum(1 for x in items if (x>>9)<thr)
        if sel>=par_min:
            pairs=sel//2
            new=[(((items[2*p]>>9)+(items[2*p+1]>>9))<<9)|(511-(node+p)) for p in range(pairs)]
            items=sorted(new+items[2*pairs:]); node+=pairs
            ins+=c_par[0]+38; cyc+=c_par[1]+340; npar+=1
        else:
            new=[(thr<<9)|(511-node)]
            items=sorted(new+items[2:]); node+=1
            ins+=c_seq[0]; cyc+=c_seq[1]; nseq+=1
    return ins,cyc,npar,nseq
for kind in ("zipf255","uniform256","logtext"):
    d=datagen.GENERATORS[kind](65536 if kind!="logtext" else 1<<20)
    h=np.bincount(d,minlength=256)
    print(kind,{pm:cost(h,pm) for pm in (2,8,16,24,32,1000)})
