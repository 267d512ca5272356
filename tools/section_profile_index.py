#!/usr/bin/env python3
"""Development aid (GPU box): per-section wave cycles of the IN-INDEX decode (docs parts only: one DINT launch per decode),
from a -DDINT_PROFILE build. usage: DINT_HIP_LIB=dint_amd/variants/prof.so tools/section_profile_index.py [postings] [type] [--freqs]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host, device

postings = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000
typ = sys.argv[2] if len(sys.argv) > 2 else "single_packed_dint"
with_freqs = "--freqs" in sys.argv
kind = host.KIND_BY_TYPE[typ]
dev = torch.device("cuda:0")
sub = host.synth_collection(postings, universe=25_000_000, seed=777)
docids = host.gaps_to_docids(sub); freqs = host.synth_freqs(sub.num_postings, 5)
dd = host.build_dictionary(kind, sub, max_sample_ints=20_000_000)
fd = host.build_dictionary(kind, host.Collection(freqs - 1, sub.lens), max_sample_ints=20_000_000)
idx, offs = host.build_index(kind, dd, fd, docids, freqs, sub.lens)
blocks, total = device.index_posting_lists(idx, offs)
D, F = device.Dictionary(kind, dd), device.Dictionary(kind, fd)
padded = np.concatenate([idx, np.zeros(16, dtype=np.uint8)])
index_dev = torch.from_numpy(padded).to(dev)
table = device.BlockTable(D, blocks, padded.size)
docs_dev = torch.empty(total, dtype=torch.int32, device=dev)
freqs_dev = torch.empty(total, dtype=torch.int32, device=dev) if with_freqs else None
lib = device._lib if hasattr(device, "_lib") else C.CDLL(os.environ["DINT_HIP_LIB"])
prof = (C.c_ulonglong * 16)()
WAVES = 4096
names = {0: "outside (queue, exit)", 1: "bundle: classify", 2: "bundle: sizes + scans + cells", 3: "bundle: metas + literals", 4: "flag/delta/rank tables",
         5: "rotate + far prefetch", 7: "tails land", 8: "wait point", 9: "expand + stores", 10: "slow stores", 11: "bundle: cells, tails asked",
         12: "bundle front end (raw landed)", 13: "epilogue (-> the next bundle's bytes)", 14: "queue ticket", 15: "bundle: next bundle mapped + asked"}
for it in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    lib.dint_debug_read_profile(prof)  # clear
    e0.record(); table.decode(D, F if with_freqs else None, index_dev, padded.size, docs_dev, freqs_dev); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    assert lib.dint_debug_read_profile(prof) == 0
    if it < 3:
        continue
    tot = sum(prof)
    print(f"decode {it}: {ms:.3f} ms for the call; {tot / WAVES / 1e3:.0f}k cycles per wave if {WAVES} waves ran")
    for i in range(16):
        if prof[i]:
            print(f"  {i:2d} {names.get(i, '?'):40s} {100.0 * prof[i] / tot:5.1f} %   {prof[i] / total * 900:8.0f} cycles per 900 postings")
print("bit-exact:", bool(np.array_equal(docs_dev.cpu().numpy().view(np.uint32), docids)))
