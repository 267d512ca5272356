"""Development aid: in-index decode under rocprofv3 (kernel breakdown of dint_decode_posting_blocks)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host, device
torch.cuda.init()
dev = torch.device("cuda:0")
sub = host.synth_collection(int(float(sys.argv[1])) if len(sys.argv) > 1 else 100_000_000, universe=25_000_000, seed=777)
docids = host.gaps_to_docids(sub); freqs = host.synth_freqs(sub.num_postings, 5)
kind = host.SINGLE_PACKED
dd = host.build_dictionary(kind, sub, max_sample_ints=20_000_000)
fd = host.build_dictionary(kind, host.Collection(freqs - 1, sub.lens), max_sample_ints=20_000_000)
idx, offs = host.build_index(kind, dd, fd, docids, freqs, sub.lens)
blocks, total = device.index_posting_lists(idx, offs)
D, F = device.Dictionary(kind, dd), device.Dictionary(kind, fd)
padded = np.concatenate([idx, np.zeros(16, dtype=np.uint8)])
index_dev = torch.from_numpy(padded).to(dev)
blocks_dev = torch.from_numpy(np.ascontiguousarray(blocks).view(np.uint8).copy()).to(dev)
docs_dev = torch.empty(total, dtype=torch.int32, device=dev); freqs_dev = torch.empty(total, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream
for _ in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st = device._lib.dint_decode_posting_blocks(D._h, F._h, index_dev.data_ptr(), padded.size, blocks_dev.data_ptr(), len(blocks),
                                                docs_dev.data_ptr(), freqs_dev.data_ptr(), total, stream)
    assert st == 0
    print("call ms", (time.perf_counter() - t0) * 1e3)
print("ok", np.array_equal(docs_dev.cpu().numpy().view(np.uint32), docids), np.array_equal(freqs_dev.cpu().numpy().view(np.uint32), freqs))
