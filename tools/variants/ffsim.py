import sys, numpy as np
sys.path.insert(0,'/root/repo')
from dint_amd import host
coll = host.synth_collection(30_000_000, universe=25_000_000, seed=12345)
d = host.build_dictionary(host.MULTI_PACKED, coll, max_sample_ints=20_000_000)
enc, units = host.encode_vroom(host.MULTI_PACKED, d, coll, unit_ints=256)
nxt = np.r_[units["in_off"][1:], enc.size].astype(np.int64)
ino = units["in_off"].astype(np.int64)
# spans: up to next unit start (headers between lists make it slightly longer)
span = nxt - ino
sel = enc[ino]
stride = np.where(sel >= 6, 4, 8)
L = (span - 1 + stride - 1) // stride
ok = (L >= 1) & (L <= 63) & (span <= 504)
print("units", len(units), "eligible", ok.mean(), "mean lanes", L[ok].mean(), "narrow share", (sel>=6).mean())
L = np.where(ok, L, 0)
def ff(ls, kopen=4, maxmem=8):
    used=[];mem=[];closed=0; bins=[]
    open_=[]
    for l in ls:
        if l==0: continue
        b=None
        for k,(u,m) in enumerate(open_):
            if u+l<=64 and m<maxmem: b=k;break
        if b is None:
            if len(open_)<kopen: open_.append([l,1])
            else:
                k=max(range(kopen), key=lambda k: open_[k][0])
                bins.append(open_[k][0]); open_[k]=[l,1]
        else:
            open_[b][0]+=l; open_[b][1]+=1
    bins+= [o[0] for o in open_]
    return bins
def ffd(ls, maxmem=8):
    bins=[]
    for l in sorted([x for x in ls if x], reverse=True):
        for b in bins:
            if b[0]+l<=64 and b[1]<maxmem: b[0]+=l;b[1]+=1;break
        else: bins.append([l,1])
    return [b[0] for b in bins]
def bfd(ls, maxmem=8):
    bins=[]
    for l in sorted([x for x in ls if x], reverse=True):
        best=None
        for b in bins:
            if b[0]+l<=64 and b[1]<maxmem and (best is None or b[0]>best[0]): best=b
        if best is None: bins.append([l,1])
        else: best[0]+=l;best[1]+=1
    return [b[0] for b in bins]
n=len(L)//64*64
for name,fn in (("ff4",lambda x: ff(x,4)),("ff8",lambda x: ff(x,8)),("ff16",lambda x: ff(x,16)),("ffd",ffd),("bfd",bfd)):
    tot=0;lan=0;lower=0
    for c in range(0,min(n,64*3000),64):
        b=fn(list(L[c:c+64])); tot+=len(b); lan+=sum(b); lower+=-(-sum(b)//64)
    print(name,"bundles/chunk",tot/3000,"fill",lan/tot,"lower bound bundles/chunk",lower/3000)
# width-homogeneous
selc = sel>=6
for name,fn in (("ff4-by-width",lambda x: ff(x,4)),("ffd-by-width",ffd)):
    tot=0;lan=0
    for c in range(0,min(n,64*3000),64):
        for w in (0,1):
            ls=[int(l) for l,s in zip(L[c:c+64],selc[c:c+64]) if s==w]
            b=fn(ls); tot+=len(b); lan+=sum(b)
    print(name,"bundles/chunk",tot/3000,"fill",lan/tot)
import collections
print(np.percentile(L[ok],[1,10,25,50,75,90,99]))
