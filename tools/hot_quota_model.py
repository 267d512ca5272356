#!/usr/bin/env python3
"""Development aid (CPU only): the LDS image of a multi-dictionary file under three ways of dividing it between the six
dictionaries — even quotas (until round 6), quotas handed on from the dictionaries that fit whole (choose_hot_set today), quotas
weighted by each dictionary's share of the stream's codewords (which the file does not record) — and the share of a bench-shaped
stream's codewords each keeps on chip. The numbers of profiles/r06_ab_hot_quota.txt.
usage: tools/hot_quota_model.py [postings=4e7]"""
import ctypes as C, os, subprocess, sys, tempfile
sys.path[:0] = [os.path.dirname(os.path.dirname(os.path.abspath(__file__)))]
import numpy as np
from dint_amd import host
SRC = r"""
#include <stdint.h>
#include <string.h>
static const uint8_t* vb(const uint8_t* in, uint32_t* v){ uint32_t x=0; for(unsigned s=0;;s+=7){uint8_t c=*in++; x+=(uint32_t)(c&127)<<s; if(c&128){*v=x;return in;}}}
// sizes: [6][65536]; hist: [6][65536]; blocks[12]
uint64_t walk(const uint8_t* enc, uint64_t bytes, const uint32_t* sizes, uint64_t* hist, uint64_t* blocks){
  const uint8_t* p=enc; const uint8_t* end=enc+bytes; uint64_t ints=0;
  while(p<end){ uint32_t n,u; p=vb(p,&n); p=vb(p,&u);
    uint32_t done=0;
    while(done<n){ uint32_t bn = n-done<256?n-done:256; uint8_t sc=*p++; blocks[sc]++; uint32_t i=0;
      if(sc<6){ const uint32_t* sz=sizes+65536u*sc; uint64_t* h=hist+65536u*sc;
        while(i<bn){ uint16_t s; memcpy(&s,p,2); p+=2; h[s]++; if(s==0){p+=2;i++;} else if(s==1){p+=4;i++;} else i+=sz[s]; } }
      else { const uint32_t* sz=sizes+65536u*(sc-6); uint64_t* h=hist+65536u*(sc-6);
        while(i<bn){ uint8_t s=*p++; h[s]++; if(s==0){p+=2;i++;} else if(s==1){p+=4;i++;} else i+=sz[s]; } }
      done+=bn; }
    ints+=n; }
  return ints; }
"""
N = int(float(sys.argv[1])) if len(sys.argv) > 1 else 60_000_000
tmp = tempfile.mkdtemp()
open(os.path.join(tmp, "walk.c"), "w").write(SRC)
subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", os.path.join(tmp, "walk.so"), os.path.join(tmp, "walk.c")], check=True)
lib = C.CDLL(os.path.join(tmp, "walk.so")); lib.walk.restype = C.c_uint64
p = host.synth_params(universe=25_000_000, seed=12345)
lens = host.synth_lengths(p, N)
coll = host.Collection(host.synth_gaps(p, lens), lens)
kind = host.KIND_BY_TYPE["multi_packed_dint"]
df = host.build_dictionary(kind, coll, max_sample_ints=20_000_000)
enc, _ = host.encode_vroom(kind, df, coll, unit_ints=256)
w = np.frombuffer(df, dtype="<u4")
n_start, n_off, n_tab = int(w[1]), int(w[2]), int(w[3])
start = np.concatenate([w[4:4+n_start].astype(np.int64),[int(w[2])]]); offs = w[4+n_start:4+n_start+n_off]; table = w[4+n_start+n_off:4+n_start+n_off+n_tab]
nd = n_start
print("dicts", nd, "entries", np.diff(start), "table words", n_tab)
sizes = np.ones((6, 65536), dtype=np.uint32)
for d in range(nd):
    k = int(min(start[d+1]-start[d],65536)); sizes[d,:k] = (offs[start[d]:start[d]+k] >> 24) + 1
hist = np.zeros((6, 65536), dtype=np.uint64); blocks = np.zeros(16, dtype=np.uint64)
assert lib.walk(C.c_void_p(enc.ctypes.data), C.c_uint64(enc.size), C.c_void_p(sizes.ctypes.data), C.c_void_p(hist.ctypes.data), C.c_void_p(blocks.ctypes.data)) == coll.num_postings
tot = int(hist.sum()); print("codewords", tot, "blocks by selector", blocks[:12])
print("codewords per dict", hist.sum(axis=1), (hist.sum(axis=1)/tot).round(3))
off = (offs & 0xFFFFFF).astype(np.int64)
kHot = 40960 - 352 - 16 * (132 + 260 + 4 * 176)
total_units = ((2*kHot - 256 - 8)//nd - 8) * nd
def evaluate(quota):
    covered = np.zeros(n_tab, dtype=bool); hot_k=[]; units=[]
    for d in range(nd):
        ne = int(min(start[d+1]-start[d], 65536)); u=0; k=0
        while k<ne:
            sl = start[d]+k; sz=int(sizes[d,k]); pw = sz if (k>=7 and sz<=16) else 0
            o=int(off[sl])
            if pw and table[o:o+pw].max()>0xFFFF: pw=0
            add = 2 + (int((~covered[o:o+pw]).sum()) if pw else 0)
            if u+add>quota[d]: break
            u+=add
            if pw: covered[o:o+pw]=True
            k+=1
        hot_k.append(k); units.append(u)
    return hot_k, units
def hits(hot_k): return sum(int(hist[d,:hot_k[d]].sum()) for d in range(nd))/tot
q0=[total_units//nd]*nd
hk,un=evaluate(q0); print("even split: hot_k",hk,"units",un,"bytes",2*sum(un)+512,"hit",round(hits(hk),4))
# water filling
q=list(q0)
for it in range(8):
    hk,un=evaluate(q)
    ne=[int(min(start[d+1]-start[d],65536)) for d in range(nd)]
    sat=[hk[d]==ne[d] for d in range(nd)]
    spare=sum(q[d]-un[d] for d in range(nd) if sat[d]); nun=sum(1 for d in range(nd) if not sat[d])
    if spare==0 or nun==0: break
    for d in range(nd):
        if sat[d]: q[d]=un[d]
        else: q[d]+=spare//nun
print("water filling: hot_k",hk,"units",un,"bytes",2*sum(un)+512,"hit",round(hits(hk),4), "iters", it)
# by use-share weighting among unsaturated (oracle knowledge of stream)
share=hist.sum(axis=1)[:nd]/tot
for alpha in (0.5,1.0):
    q=list(q0)
    for it in range(10):
        hk,un=evaluate(q)
        sat=[hk[d]==ne[d] for d in range(nd)]
        spare=sum(q[d]-un[d] for d in range(nd) if sat[d]); 
        unsat=[d for d in range(nd) if not sat[d]]
        if not unsat: break
        pool=spare+sum(q[d] for d in unsat)
        wts=np.array([share[d]**alpha for d in unsat]); wts/=wts.sum()
        newq=list(q)
        for d in range(nd):
            if sat[d]: newq[d]=un[d]
        for j,d in enumerate(unsat): newq[d]=int(pool*wts[j])
        if newq==q: break
        q=newq
    hk,un=evaluate(q)
    print("use-weighted alpha",alpha,": hot_k",hk,"bytes",2*sum(un)+512,"hit",round(hits(hk),4))
