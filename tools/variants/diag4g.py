import sys, numpy as np, torch
sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/oracle']
import conftest
from dint_amd import host, device
from test_index_cpu import get_index
kind=host.MULTI_PACKED
corpus=conftest.Corpus(400_000, universe=200_000, seed=7)
ix=get_index(corpus, kind)
blocks,total=device.index_posting_lists(ix.bytes, ix.offsets)
dd,fd=device.Dictionary(kind, ix.docs_dict), device.Dictionary(kind, ix.freqs_dict)
dev=torch.device("cuda",0)
import oracle
od,of=oracle.OracleDict(kind, ix.docs_dict), oracle.OracleDict(kind, ix.freqs_dict)
for cut, opts in ((len(ix.bytes)//2, {}), (len(ix.bytes)//2, {'bundles': 0}), (len(ix.bytes)//2, {'index_pair': 0}), (len(ix.bytes)//2, {'index_concurrent': 0}), (len(ix.bytes)//2, {'index_inline_tails': 0})):
    device.reset_options()
    for k_, v_ in opts.items(): device.set_option(k_, v_)
    print(opts)
    shift=(1<<32)-cut
    big=torch.zeros(shift+len(ix.bytes)+16,dtype=torch.uint8,device=dev)
    big[shift:shift+len(ix.bytes)]=torch.from_numpy(ix.bytes).to(dev)
    moved=blocks.copy(); moved["in_off"]+=np.uint64(shift)
    table=device.BlockTable(dd, moved, big.numel())
    for it in range(2):
        docids_dev=torch.full((total,),-1,dtype=torch.int32,device=dev); freqs_dev=torch.full((total,),-1,dtype=torch.int32,device=dev)
        table.decode(dd,fd,big,big.numel(),docids_dev,freqs_dev); torch.cuda.synchronize()
        f=freqs_dev.cpu().numpy().view(np.uint32); bad=np.flatnonzero(f!=ix.freqs)
        base=big.data_ptr()
        if bad.size:
            b=int(np.searchsorted(blocks["out_off"], bad[0], side="right")-1)
            print("cut",cut,"base %#x"%base,"decode", it, "bad", bad.size, "block", b, "n", blocks["n"][b], "block in_off (index rel)", int(blocks["in_off"][b]), "cut-in_off", cut-int(blocks["in_off"][b]), "next block in_off-cut", int(blocks["in_off"][b+1])-cut,
                  "abs lo32 of block", hex((base+int(moved["in_off"][b]))&0xFFFFFFFF), "pos in block", (bad[:20]-blocks["out_off"][b]).tolist(), "got", f[bad[:4]].tolist(), "want", ix.freqs[bad[:4]].tolist(), flush=True)
        else: print("cut",cut,"base %#x"%base,"decode",it,"ok", flush=True)
    del table, big; torch.cuda.empty_cache()
