#!/usr/bin/env python3
"""Development aid (CPU only): what share of a stream's codewords would find their integers on chip under other layouts of
the LDS image — the numbers behind DESIGN.md's "16-bit hot metadata" entry. Builds the bench's dictionary and a sample of
its stream, counts every codeword's uses (a small C walker, compiled on the spot) and lays the hot set out under:
  - today's image: 4-byte metadata word + u16 integers, the union of the hot entries' table intervals (each word once);
  - 2-byte metadata with the entry's size in a unary prefix: the integers' offset has 15 - log2(size) bits, in units of the
    entry's size, so every entry must sit at an address that is a multiple of its size — sharing only between windows
    that happen to be aligned (greedy, longest entries first);
  - the same costs without the alignment rule (what the 2-byte word would buy if the sharing survived).
usage: tools/hot_image_model.py [postings=6e7]"""
import ctypes as C, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np
from dint_amd import host

SRC = r"""
#include <stdint.h>
#include <string.h>
static const uint8_t* vb(const uint8_t* in, uint32_t* v){ uint32_t x=0; for(unsigned s=0;;s+=7){uint8_t c=*in++; x+=(uint32_t)(c&127)<<s; if(c&128){*v=x;return in;}}}
uint64_t walk(const uint8_t* enc, uint64_t bytes, const uint32_t* sizes, uint64_t* hist){
  const uint8_t* p=enc; const uint8_t* end=enc+bytes; uint64_t ints=0;
  while(p<end){ uint32_t n,u; p=vb(p,&n); p=vb(p,&u); uint32_t i=0;
    while(i<n){ uint16_t s; memcpy(&s,p,2); p+=2; hist[s]++; if(s==0){p+=2; i++;} else if(s==1){p+=4; i++;} else i+=sizes[s]; }
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
df = host.build_dictionary(host.SINGLE_PACKED, coll, max_sample_ints=20_000_000)
enc, _ = host.encode_vroom(host.SINGLE_PACKED, df, coll, unit_ints=16384)
w = np.frombuffer(df, dtype="<u4"); n_off, n_tab = int(w[1]), int(w[2]); offs = w[3:3 + n_off]; table = w[3 + n_off:3 + n_off + n_tab]
sizes = ((offs >> 24) + 1).astype(np.int64); off = (offs & 0xFFFFFF).astype(np.int64)
s32 = np.concatenate([sizes.astype(np.uint32), np.ones(65536 - n_off, dtype=np.uint32)])
hist = np.zeros(65536, dtype=np.uint64)
assert lib.walk(C.c_void_p(enc.ctypes.data), C.c_uint64(enc.size), C.c_void_p(s32.ctypes.data), C.c_void_p(hist.ctypes.data)) == coll.num_postings
tot = int(hist.sum())
print(f"{coll.num_postings} postings, {tot} codewords ({coll.num_postings / tot:.2f} integers each), {enc.size * 8 / coll.num_postings:.3f} bits per integer")
ents = [tuple(int(x) for x in table[off[i]:off[i] + sizes[i]]) if (i >= 7 and sizes[i] <= 16) else None for i in range(n_off)]
budget = (40960 - 352 - 16 * (132 + 260 + 4 * 176)) * 4   # the image's bytes (dint_kernels.hpp: kHotImageWords)


def union_payload(k):   # today's: every table word a hot entry covers, once
    cov = np.zeros(n_tab, dtype=bool)
    for i in range(7, k):
        if ents[i] is not None and max(ents[i]) < 65536: cov[off[i]:off[i] + sizes[i]] = True
    return 256 + int(cov.sum())


def aligned_payload(k):  # entries at multiples of their size; shared where an aligned window of placed data holds the integers
    idx = sorted((i for i in range(7, k) if ents[i] is not None and max(ents[i]) < 65536), key=lambda i: -sizes[i])
    win = {(0,) * s: 0 for s in (1, 2, 4, 8, 16)}
    pos = 256
    for i in idx:
        e = ents[i]; s = len(e)
        if e in win: continue
        a = (pos + s - 1) // s * s; pos = a + s
        t = s
        while t >= 1:
            for j in range(0, s, t): win.setdefault(e[j:j + t], a + j)
            t //= 2
    return pos


def best(meta_bytes, payload, reach=None):
    lo, hi = 7, n_off
    while lo < hi:
        mid = (lo + hi + 1) // 2
        pay = payload(mid)
        if meta_bytes * (mid + 1) + 2 * pay <= budget and (reach is None or pay <= reach): lo = mid
        else: hi = mid - 1
    use = float(hist[:lo].sum()) / tot
    return lo, payload(lo), use


for name, mb, fn, reach in (("today: 4-byte metadata, u16 integers, union of intervals", 4, union_payload, None),
                            ("2-byte metadata (size in a unary prefix): size-aligned entries, 64 KB reach", 2, aligned_payload, 32768),
                            ("2-byte metadata if the sharing survived (no alignment rule)", 2, union_payload, 32768)):
    k, pay, use = best(mb, fn, reach)
    print(f"{name}: hot_k {k}, payload {pay} u16, {100 * use:.1f} % of the codewords on chip, {245 * (1 - use):.1f} cold requests per tile of 245 codewords")
