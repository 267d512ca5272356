"""Throughput of the other BASELINE configs on one GPU (development aid; bench.py is the judged line).
usage: tools/bench_configs.py [postings]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
import numpy as np, torch
from dint_amd import host, device

postings = int(float(sys.argv[1])) if len(sys.argv) > 1 else 200_000_000
coll = host.synth_collection(postings, universe=25_000_000, seed=12345)
dev = torch.device("cuda:0")
for typ in ("single_rect_dint", "single_packed_dint", "multi_packed_dint"):
    kind = host.KIND_BY_TYPE[typ]
    t = time.time()
    d_file = host.build_dictionary(kind, coll, max_sample_ints=20_000_000)
    # multi: one block per unit, so that consecutive blocks are bundled several to a tile
    enc, units = host.encode_vroom(kind, d_file, coll, unit_ints=256 if kind == host.MULTI_PACKED else 8192)
    d = device.Dictionary(kind, d_file)
    enc_dev = torch.from_numpy(enc).to(dev); units_dev = device.units_to_device(units, dev)
    out_dev = torch.empty(coll.num_postings, dtype=torch.int32, device=dev)
    ms = []
    for i in range(6):
        d.decode_units(enc_dev, units_dev, len(units), out_dev); torch.cuda.synchronize(); ms.append(d.last_kernel_ms())
    ok = np.array_equal(out_dev.cpu().numpy().view(np.uint32), coll.gaps)
    m = float(np.median(ms[1:]))
    print(f"{typ:20s} bpi {enc.size * 8 / coll.num_postings:.3f}  {m:.3f} ms  {coll.num_postings / m / 1e6:.1f} G ints/s  "
          f"{(4 * coll.num_postings + enc.size) / m / 1e6:.0f} GB/s  hot {d.info().hot_entries}  bit-exact {ok}  (setup {time.time() - t:.0f}s)", flush=True)
# in-index: docs + freqs
sub = host.synth_collection(min(postings, 100_000_000), universe=25_000_000, seed=777)
docids = host.gaps_to_docids(sub); freqs = host.synth_freqs(sub.num_postings, 5)
kind = host.SINGLE_PACKED
dd = host.build_dictionary(kind, sub, max_sample_ints=20_000_000)
fd = host.build_dictionary(kind, host.Collection(freqs - 1, sub.lens), max_sample_ints=20_000_000)
idx, offs = host.build_index(kind, dd, fd, docids, freqs, sub.lens)
blocks, total = device.index_posting_lists(idx, offs)
D, F = device.Dictionary(kind, dd), device.Dictionary(kind, fd)
got_d, got_f = device.decode_posting_lists(D, F, idx, blocks, total)
# device-resident timing of the C call (it synchronises its stream and owns a temporary workspace)
import ctypes as C
padded = np.concatenate([idx, np.zeros(16, dtype=np.uint8)])
index_dev = torch.from_numpy(padded).to(dev)
blocks_dev = torch.from_numpy(np.ascontiguousarray(blocks).view(np.uint8).copy()).to(dev)
docs_dev = torch.empty(total, dtype=torch.int32, device=dev); freqs_dev = torch.empty(total, dtype=torch.int32, device=dev)
stream = torch.cuda.current_stream(dev).cuda_stream
times = {}
for label, fptr in (("docs+freqs", freqs_dev.data_ptr()), ("docs only", None)):
    best = 1e9
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st = device._lib.dint_decode_posting_blocks(D._h, F._h if fptr else None, index_dev.data_ptr(), padded.size, blocks_dev.data_ptr(),
                                                    len(blocks), docs_dev.data_ptr(), fptr, total, stream)
        assert st == 0
        best = min(best, time.perf_counter() - t0)
    times[label] = best
ok = np.array_equal(docs_dev.cpu().numpy().view(np.uint32), docids) and np.array_equal(freqs_dev.cpu().numpy().view(np.uint32), freqs)
print(f"in-index single_packed: {total} postings, {idx.size * 8 / total:.3f} bits/posting (docs+freqs), {len(blocks)} blocks; "
      f"device-resident call: docs+freqs {times['docs+freqs'] * 1e3:.2f} ms ({total / times['docs+freqs'] / 1e9:.1f} G postings/s), "
      f"docs only {times['docs only'] * 1e3:.2f} ms ({total / times['docs only'] / 1e9:.1f} G postings/s); bit-exact {ok and np.array_equal(got_d, docids) and np.array_equal(got_f, freqs)}")
