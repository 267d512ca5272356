"""In-index path on the GPU: posting lists (dict_posting_list layout) -> docIDs and freqs,
bit-exact against the oracle's document_enumerator walk and the index builder's input."""
import numpy as np
import pytest

import oracle
from dint_amd import host
from test_index_cpu import get_index

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    import torch

    assert torch.cuda.is_available()
    from dint_amd import device as dev

    return dev


@pytest.fixture(autouse=True)
def _options_back_to_default(device):
    yield
    device.reset_options()


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.RECTANGULAR, host.MULTI_PACKED])
@pytest.mark.parametrize("corpus_name", ["small_corpus", "dense_corpus", "sparse_corpus"])
def test_posting_lists_match_oracle(device, request, kind, corpus_name):
    corpus = request.getfixturevalue(corpus_name)
    ix = get_index(corpus, kind)
    blocks, total = device.index_posting_lists(ix.bytes, ix.offsets)
    assert total == corpus.coll.num_postings
    assert int(blocks["n"].sum()) == total and (blocks["n"] <= 256).all()
    dd = device.Dictionary(kind, ix.docs_dict)
    fd = device.Dictionary(kind, ix.freqs_dict)
    docids, freqs = device.decode_posting_lists(dd, fd, ix.bytes, blocks, total)
    assert np.array_equal(docids, ix.docids)
    assert np.array_equal(freqs, ix.freqs)
    # and the oracle's walk of a sample of lists agrees with both
    od, of = oracle.OracleDict(kind, ix.docs_dict), oracle.OracleDict(kind, ix.freqs_dict)
    for i in range(0, len(ix.lens), max(1, len(ix.lens) // 50)):
        if ix.lens[i] == 0:
            continue
        d, f = oracle.posting_list_decode(od, of, ix.bytes, int(ix.offsets[i]))
        lo, hi = int(ix.bounds[i]), int(ix.bounds[i + 1])
        assert np.array_equal(docids[lo:hi], d) and np.array_equal(freqs[lo:hi], f)


def test_docs_only(device, small_corpus):
    ix = get_index(small_corpus, host.SINGLE_PACKED)
    blocks, total = device.index_posting_lists(ix.bytes, ix.offsets)
    dd = device.Dictionary(host.SINGLE_PACKED, ix.docs_dict)
    docids, freqs = device.decode_posting_lists(dd, None, ix.bytes, blocks, total)
    assert freqs is None and np.array_equal(docids, ix.docids)


def test_block_table_matches_the_list_directories(device, small_corpus):
    ix = get_index(small_corpus, host.SINGLE_PACKED)
    blocks, _ = device.index_posting_lists(ix.bytes, ix.offsets)
    first = np.r_[True, blocks["list"][1:] != blocks["list"][:-1]]
    assert (blocks["base"][first] == 0).all()
    # every block's max is the last docID of that block
    ends = (blocks["out_off"] + blocks["n"] - 1).astype(np.int64)
    assert np.array_equal(blocks["max"], ix.docids[ends])
    # bases chain: base = previous max + 1 inside a list
    same = ~first
    assert np.array_equal(blocks["base"][same], blocks["max"][np.flatnonzero(same) - 1] + 1)


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
def test_prepared_block_table_decodes_asynchronously(device, small_corpus, kind):
    """dint_block_table_create + dint_decode_block_table: prepared once, decoded twice on a side stream without a
    host synchronisation in between; docIDs come straight out of the decode kernels (fused prefix sums)."""
    import torch

    ix = get_index(small_corpus, kind)
    blocks, total = device.index_posting_lists(ix.bytes, ix.offsets)
    dd, fd = device.Dictionary(kind, ix.docs_dict), device.Dictionary(kind, ix.freqs_dict)
    dev = torch.device("cuda", 0)
    padded = np.concatenate([ix.bytes, np.zeros(16, np.uint8)])
    index_dev = torch.from_numpy(padded).to(dev)
    table = device.BlockTable(dd, blocks, padded.size)
    side = torch.cuda.Stream(dev)
    outs = []
    with torch.cuda.stream(side):
        for _ in range(2):
            docids_dev = torch.full((total,), -1, dtype=torch.int32, device=dev)
            freqs_dev = torch.full((total,), -1, dtype=torch.int32, device=dev)
            table.decode(dd, fd, index_dev, padded.size, docids_dev, freqs_dev, stream=side.cuda_stream)
            outs.append((docids_dev, freqs_dev))
    side.synchronize()
    for docids_dev, freqs_dev in outs:
        assert np.array_equal(docids_dev.cpu().numpy().view(np.uint32), ix.docids)
        assert np.array_equal(freqs_dev.cpu().numpy().view(np.uint32), ix.freqs)
    # docs only
    docids_dev = torch.empty(total, dtype=torch.int32, device=dev)
    table.decode(dd, None, index_dev, padded.size, docids_dev, None)
    torch.cuda.synchronize()
    assert np.array_equal(docids_dev.cpu().numpy().view(np.uint32), ix.docids)


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
@pytest.mark.parametrize("with_freqs", [True, False])
def test_a_taught_table_decodes_in_one_launch_from_its_first_decode(device, small_corpus, kind, with_freqs):
    """dint_block_table_learn: the sizing pass over the index at set-up — where the docs parts end, the freqs parts' units, both
    bundle schedules — so that the caller's FIRST decode is already the one launch (dint_block_table_ready), and bit-exact;
    an untaught table gets there under its first two decodes, as before."""
    import torch

    ix = get_index(small_corpus, kind)
    blocks, total = device.index_posting_lists(ix.bytes, ix.offsets)
    dd, fd = device.Dictionary(kind, ix.docs_dict), device.Dictionary(kind, ix.freqs_dict)
    dev = torch.device("cuda", 0)
    padded = np.concatenate([ix.bytes, np.zeros(16, np.uint8)])
    index_dev = torch.from_numpy(padded).to(dev)
    taught, plain = device.BlockTable(dd, blocks, padded.size), device.BlockTable(dd, blocks, padded.size)
    assert not taught.ready(with_freqs) and not plain.ready(with_freqs)
    taught.learn(dd, fd if with_freqs else None, index_dev, padded.size)
    assert taught.ready(with_freqs) and (with_freqs or not taught.ready(True))
    for table, passes in ((taught, 1), (plain, 3)):
        for _ in range(passes):
            docids_dev = torch.full((total,), -1, dtype=torch.int32, device=dev)
            freqs_dev = torch.full((total,), -1, dtype=torch.int32, device=dev) if with_freqs else None
            table.decode(dd, fd if with_freqs else None, index_dev, padded.size, docids_dev, freqs_dev)
            torch.cuda.synchronize()
            assert np.array_equal(docids_dev.cpu().numpy().view(np.uint32), ix.docids)
            if with_freqs:
                assert np.array_equal(freqs_dev.cpu().numpy().view(np.uint32), ix.freqs)
    assert plain.ready(with_freqs)
    # a docs-only table taught without freqs learns the freqs side under later decodes
    if not with_freqs:
        for _ in range(3):
            docids_dev = torch.empty(total, dtype=torch.int32, device=dev)
            freqs_dev = torch.empty(total, dtype=torch.int32, device=dev)
            taught.decode(dd, fd, index_dev, padded.size, docids_dev, freqs_dev)
            torch.cuda.synchronize()
            assert np.array_equal(freqs_dev.cpu().numpy().view(np.uint32), ix.freqs)
            assert np.array_equal(docids_dev.cpu().numpy().view(np.uint32), ix.docids)


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
def test_block_table_side_streams_are_joined_to_the_callers_stream(device, small_corpus, kind):
    """From its second decode on a table runs the freqs launch and the short blocks' decoder on streams of its own. What
    the caller puts on ITS stream around a call must still be ordered with them: the poison written before a call is there
    before any of the three kernels writes, and a copy issued right behind the call sees all of their outputs — six
    calls back to back into the same two buffers, docs + freqs and docs only in turn, no host synchronisation between."""
    import torch

    ix = get_index(small_corpus, kind)
    blocks, total = device.index_posting_lists(ix.bytes, ix.offsets)
    dd, fd = device.Dictionary(kind, ix.docs_dict), device.Dictionary(kind, ix.freqs_dict)
    dev = torch.device("cuda", 0)
    padded = np.concatenate([ix.bytes, np.zeros(16, np.uint8)])
    index_dev = torch.from_numpy(padded).to(dev)
    table = device.BlockTable(dd, blocks, padded.size)
    stream = torch.cuda.Stream(dev)
    docids_dev = torch.empty(total, dtype=torch.int32, device=dev)
    freqs_dev = torch.empty(total, dtype=torch.int32, device=dev)
    got = []
    with torch.cuda.stream(stream):
        for i in range(6):
            with_freqs = i % 2 == 0
            docids_dev.fill_(-1)
            freqs_dev.fill_(-1)
            table.decode(dd, fd if with_freqs else None, index_dev, padded.size, docids_dev, freqs_dev if with_freqs else None,
                         stream=stream.cuda_stream)
            got.append((with_freqs, docids_dev.clone(), freqs_dev.clone()))
    stream.synchronize()
    for with_freqs, d, f in got:
        assert np.array_equal(d.cpu().numpy().view(np.uint32), ix.docids)
        if with_freqs:
            assert np.array_equal(f.cpu().numpy().view(np.uint32), ix.freqs)
        else:
            assert (f.cpu().numpy() == -1).all()


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
def test_block_table_keeps_nothing_from_a_decode_that_skipped_blocks(device, small_corpus, kind):
    """A prepared table keeps what its decodes learn (exact spans, the freqs parts' units, both bundle schedules). A
    decode whose out_capacity is too small for some blocks skips them — nothing may be learnt from it — and a schedule
    built under one capacity must not serve a smaller one: first call too small, then four docs + freqs decodes
    (the cached freqs schedule is in use from the third), then a too-small one again with canaries behind it."""
    import torch

    ix = get_index(small_corpus, kind)
    blocks, total = device.index_posting_lists(ix.bytes, ix.offsets)
    dd, fd = device.Dictionary(kind, ix.docs_dict), device.Dictionary(kind, ix.freqs_dict)
    dev = torch.device("cuda", 0)
    padded = np.concatenate([ix.bytes, np.zeros(16, np.uint8)])
    index_dev = torch.from_numpy(padded).to(dev)
    table = device.BlockTable(dd, blocks, padded.size)
    cut = total - 3 * 256 - 17   # the last blocks do not fit

    def run(capacity):
        docids_dev = torch.full((total + 1024,), -5, dtype=torch.int32, device=dev)
        freqs_dev = torch.full((total + 1024,), -5, dtype=torch.int32, device=dev)
        table.decode(dd, fd, index_dev, padded.size, docids_dev[:capacity], freqs_dev[:capacity])
        torch.cuda.synchronize()
        return docids_dev.cpu().numpy(), freqs_dev.cpu().numpy()

    def check_short(docids, freqs):
        fits = (blocks["out_off"] + blocks["n"]).astype(np.int64) <= cut
        for b in np.flatnonzero(~fits):
            lo, n = int(blocks["out_off"][b]), int(blocks["n"][b])
            assert (docids[lo:lo + n] == -5).all() and (freqs[lo:lo + n] == -5).all()
        assert (docids[cut:] == -5).all() and (freqs[cut:] == -5).all(), "written past the capacity"
        keep = np.zeros(total, bool)
        for b in np.flatnonzero(fits):
            keep[int(blocks["out_off"][b]):int(blocks["out_off"][b]) + int(blocks["n"][b])] = True
        assert np.array_equal(docids[:total].view(np.uint32)[keep], ix.docids[keep])
        assert np.array_equal(freqs[:total].view(np.uint32)[keep], ix.freqs[keep])

    check_short(*run(cut))
    for _ in range(4):
        docids, freqs = run(total)
        assert np.array_equal(docids[:total].view(np.uint32), ix.docids)
        assert np.array_equal(freqs[:total].view(np.uint32), ix.freqs)
        assert (docids[total:] == -5).all() and (freqs[total:] == -5).all()
    check_short(*run(cut))
    docids, freqs = run(total)
    assert np.array_equal(docids[:total].view(np.uint32), ix.docids) and np.array_equal(freqs[:total].view(np.uint32), ix.freqs)


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
@pytest.mark.parametrize("inline_tails", [0, 1])
def test_short_blocks_inside_the_docs_launch_or_in_their_own(device, small_corpus, kind, inline_tails):
    """A created block table carries its short blocks as tickets (longest first, one lane per block, as many as fit a
    wave's LDS scratch) that the docs launch's waves decode before their DINT work; dint_set_option(index_inline_tails, 0)
    brings the launch of their own back. Both ways, five decodes (the first learns the spans, the second builds the
    schedules, from the third on the kernels compiled without the unit queue run), with and without freqs, into poisoned
    buffers with canaries behind them: bit-exact, nothing written past the last posting."""
    import torch

    ix = get_index(small_corpus, kind)
    blocks, total = device.index_posting_lists(ix.bytes, ix.offsets)
    assert (blocks["n"] < 256).sum() > 64 and (blocks["n"] < 256).sum() < len(blocks)
    dd, fd = device.Dictionary(kind, ix.docs_dict), device.Dictionary(kind, ix.freqs_dict)
    dev = torch.device("cuda", 0)
    padded = np.concatenate([ix.bytes, np.zeros(16, np.uint8)])
    index_dev = torch.from_numpy(padded).to(dev)
    with device.options(index_inline_tails=inline_tails):
        table = device.BlockTable(dd, blocks, padded.size)
        for i in range(5):
            with_freqs = i != 3
            docids_dev = torch.full((total + 512,), -9, dtype=torch.int32, device=dev)
            freqs_dev = torch.full((total + 512,), -9, dtype=torch.int32, device=dev)
            table.decode(dd, fd if with_freqs else None, index_dev, padded.size, docids_dev[:total], freqs_dev[:total] if with_freqs else None)
            torch.cuda.synchronize()
            d, f = docids_dev.cpu().numpy(), freqs_dev.cpu().numpy()
            assert np.array_equal(d[:total].view(np.uint32), ix.docids), i
            assert (d[total:] == -9).all()
            if with_freqs:
                assert np.array_equal(f[:total].view(np.uint32), ix.freqs), i
                assert (f[total:] == -9).all()
            else:
                assert (f == -9).all()
        table.close()


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
def test_blocks_that_fit_no_tile(device, small_corpus, kind):
    """A full block whose 256 postings are nearly all exceptions is more than 504 bytes and fits no tile: the bundle schedule
    leaves it to the unit queue, and once a prepared table's schedules are known such stragglers go through a small second
    launch of the general kernel while everything else runs the kernels compiled without the queue (one launch for docs +
    short blocks + freqs). Lists of 700 postings among an ordinary
    collection: gaps of 70 000 and more (32-bit literals, six bytes a posting); literals alternating with the collection's
    own small gaps (hot and cold dictionary entries between exceptions); runs of consecutive docIDs (run codewords) between
    literals. Five decodes (the pair launch from the third), against the encoder's input and the oracle's walk."""
    import torch

    coll = small_corpus.coll
    r = np.random.default_rng(99)
    extra = [r.integers(70_000, 400_000, 700, dtype=np.uint64).astype(np.uint32) for _ in range(2)]
    alt = r.integers(70_000, 400_000, 700, dtype=np.uint64).astype(np.uint32)
    alt[1::2] = coll.gaps[1000:1000 + 350]                 # every other gap one of the corpus's own
    runs = r.integers(1_000, 60_000, 700, dtype=np.uint64).astype(np.uint32)  # 16-bit literals ...
    for at in range(0, 700, 50):
        runs[at:at + 17] = 0                                # ... and runs of consecutive docIDs
    extra += [alt, runs]
    where = [5, len(coll.lens) // 3, len(coll.lens) // 2, len(coll.lens) - 3]   # the stragglers' lists among the others
    b = coll.list_bounds()
    gaps_parts, lens = [], []
    for i in range(len(coll.lens)):
        for w, e in zip(where, extra):
            if i == w:
                gaps_parts.append(e)
                lens.append(len(e))
        gaps_parts.append(coll.gaps[int(b[i]):int(b[i + 1])])
        lens.append(int(coll.lens[i]))
    big = host.Collection(np.concatenate(gaps_parts), np.asarray(lens, dtype=coll.lens.dtype))
    docids = host.gaps_to_docids(big)
    freqs = host.synth_freqs(big.num_postings, 4)
    docs_dict = small_corpus.dict_file(kind)
    freqs_dict = host.build_dictionary(kind, host.Collection(freqs - 1, big.lens))
    idx, offs = host.build_index(kind, docs_dict, freqs_dict, docids, freqs, big.lens)
    blocks, total = device.index_posting_lists(idx, offs)
    spans = np.diff(np.r_[blocks["in_off"], idx.size].astype(np.int64))
    assert ((blocks["n"] == 256) & (spans > 1200)).sum() >= 4 and ((blocks["n"] == 256) & (spans > 520)).sum() >= 8, "no block too long for a tile in this index"
    dd, fd = device.Dictionary(kind, docs_dict), device.Dictionary(kind, freqs_dict)
    dev = torch.device("cuda", 0)
    padded = np.concatenate([idx, np.zeros(16, np.uint8)])
    index_dev = torch.from_numpy(padded).to(dev)
    table = device.BlockTable(dd, blocks, padded.size)
    for i in range(5):
        docids_dev = torch.full((total + 256,), -3, dtype=torch.int32, device=dev)
        freqs_dev = torch.full((total + 256,), -3, dtype=torch.int32, device=dev)
        table.decode(dd, fd, index_dev, padded.size, docids_dev[:total], freqs_dev[:total])
        torch.cuda.synchronize()
        d, f = docids_dev.cpu().numpy(), freqs_dev.cpu().numpy()
        assert np.array_equal(d[:total].view(np.uint32), docids), i
        assert np.array_equal(f[:total].view(np.uint32), freqs), i
        assert (d[total:] == -3).all() and (f[total:] == -3).all()
    table.close()
    od, of = oracle.OracleDict(kind, docs_dict), oracle.OracleDict(kind, freqs_dict)
    bounds = big.list_bounds()
    for w in where:  # (the inserted lists sit at positions where[k] + k)
        i = w + where.index(w)
        dl, fl = oracle.posting_list_decode(od, of, idx, int(offs[i]))
        lo, hi = int(bounds[i]), int(bounds[i + 1])
        assert hi - lo == 700 and np.array_equal(docids[lo:hi], dl) and np.array_equal(freqs[lo:hi], fl)


@pytest.mark.parametrize("inline_tails", [0, 1])
@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
def test_blocks_the_expansion_leaves_as_gaps_become_docids_on_the_spot(device, kind, inline_tails):
    """Blocks the expansion cannot turn into docIDs on the fly — a dictionary entry holding a value >= 65536 (a constant
    stride of 70000: the dictionary learns runs of 69999: slow codewords), or more than 256 slots in the block (every gap a
    32-bit exception: two tiles) — are summed by the wave that decoded them, behind its own stores (gaps_to_docids_here):
    in a bundle, in a segment, in the stragglers' launch. No flags are left, no fix-up launch follows. Five decodes of a
    prepared table (the one launch from the third), docs + freqs and docs only."""
    import torch

    r = np.random.default_rng(77)
    stride = (np.arange(700, dtype=np.uint64) * 70000).astype(np.uint32)
    wild = np.cumsum(r.integers(100000, 3000000, 700, dtype=np.uint64)).astype(np.uint32)
    both = np.union1d(stride, wild).astype(np.uint32)
    dense = np.arange(0, 2_000_000, 7, dtype=np.uint32)
    mixed = np.union1d(stride[:300], np.arange(5, 40_000, 3, dtype=np.uint32)).astype(np.uint32)  # slow codewords among ordinary ones
    lists = [stride, wild, both, dense, mixed, stride[:256], wild[:255]]
    docids = np.concatenate(lists)
    lens = np.array([len(x) for x in lists], dtype=np.uint32)
    gaps = np.concatenate([host.docids_to_gaps(x) for x in lists])
    coll = host.Collection(gaps, lens)
    freqs = (1 + (np.arange(docids.size, dtype=np.uint32) % 3)).astype(np.uint32)
    dd_file = host.build_dictionary(kind, coll)
    fd_file = host.build_dictionary(kind, host.Collection(freqs - 1, lens))
    idx, offs = host.build_index(kind, dd_file, fd_file, docids, freqs, lens)
    blocks, total = device.index_posting_lists(idx, offs)
    assert total == docids.size
    dd, fd = device.Dictionary(kind, dd_file), device.Dictionary(kind, fd_file)
    dev = torch.device("cuda", 0)
    padded = np.concatenate([idx, np.zeros(16, np.uint8)])
    index_dev = torch.from_numpy(padded).to(dev)
    with device.options(index_inline_tails=inline_tails):
        table = device.BlockTable(dd, blocks, padded.size)
        for i in range(5):
            with_freqs = i != 1
            docids_dev = torch.full((total + 256,), -3, dtype=torch.int32, device=dev)
            freqs_dev = torch.full((total + 256,), -3, dtype=torch.int32, device=dev)
            table.decode(dd, fd if with_freqs else None, index_dev, padded.size, docids_dev[:total], freqs_dev[:total] if with_freqs else None)
            torch.cuda.synchronize()
            d, f = docids_dev.cpu().numpy(), freqs_dev.cpu().numpy()
            assert np.array_equal(d[:total].view(np.uint32), docids), i
            assert (d[total:] == -3).all()
            if with_freqs:
                assert np.array_equal(f[:total].view(np.uint32), freqs), i
        table.close()
    # the one-shot call (no prepared table: general kernels, the short blocks by a launch of their own)
    d1, f1 = device.decode_posting_lists(dd, fd, idx, blocks, total)
    assert np.array_equal(d1, docids) and np.array_equal(f1, freqs)
    od, of = oracle.OracleDict(kind, dd_file), oracle.OracleDict(kind, fd_file)
    for i in range(len(lists)):
        dl, fl = oracle.posting_list_decode(od, of, idx, int(offs[i]))
        assert np.array_equal(dl, lists[i])


def test_two_block_tables_over_one_index_on_two_threads(device, small_corpus):
    """Two prepared block tables over ONE resident index and ONE pair of dictionaries, decoded from two host threads on a
    stream each for a few seconds (one taught, one learning under its first decodes): a table's decodes are ordered on its own
    stream, nothing but the immutable dictionaries and the index is shared (include/dint_hip.h, threading)."""
    import threading
    import time

    import torch

    kind = host.SINGLE_PACKED
    ix = get_index(small_corpus, kind)
    blocks, total = device.index_posting_lists(ix.bytes, ix.offsets)
    dd, fd = device.Dictionary(kind, ix.docs_dict), device.Dictionary(kind, ix.freqs_dict)
    dev = torch.device("cuda", 0)
    padded = np.concatenate([ix.bytes, np.zeros(16, np.uint8)])
    index_dev = torch.from_numpy(padded).to(dev)
    tables = [device.BlockTable(dd, blocks, padded.size), device.BlockTable(dd, blocks, padded.size)]
    tables[0].learn(dd, fd, index_dev, padded.size)
    torch.cuda.synchronize()
    want_d = torch.from_numpy(ix.docids.view(np.int32)).to(dev)
    want_f = torch.from_numpy(ix.freqs.view(np.int32)).to(dev)
    errors, rounds = [], [0, 0]
    stop_at = time.monotonic() + 3.0

    def worker(k):
        try:
            stream = torch.cuda.Stream(dev)
            with torch.cuda.stream(stream):
                while time.monotonic() < stop_at:
                    docids_dev = torch.full((total,), -1, dtype=torch.int32, device=dev)
                    with_freqs = rounds[k] % 3 != 2
                    freqs_dev = torch.full((total,), -1, dtype=torch.int32, device=dev) if with_freqs else None
                    tables[k].decode(dd, fd if with_freqs else None, index_dev, padded.size, docids_dev, freqs_dev, stream=stream.cuda_stream)
                    stream.synchronize()
                    assert torch.equal(docids_dev, want_d), ("docids", k, rounds[k])
                    assert not with_freqs or torch.equal(freqs_dev, want_f), ("freqs", k, rounds[k])
                    rounds[k] += 1
        except BaseException as e:  # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert min(rounds) >= 10, rounds


@pytest.mark.parametrize("kind", [host.SINGLE_PACKED, host.MULTI_PACKED])
def test_an_index_that_straddles_4_gib(device, small_corpus, kind):
    """Byte offsets are 64-bit everywhere (dint_block_ref::in_off, the end offsets, the freqs parts' starts): the same small
    index placed so that its bytes straddle offset 2^32 of a 4.3 GB device buffer — blocks below the line, across it (a docs
    part below, its freqs part above; a part cut by the line) and above it — decodes to the same docIDs and freqs. Found by
    tools/inindex_scale.py at 5e9 postings (round 6: an index of 4.32 GB, 48 wrong freqs in the one block on the line)."""
    import torch

    ix = get_index(small_corpus, kind)
    blocks, total = device.index_posting_lists(ix.bytes, ix.offsets)
    dd, fd = device.Dictionary(kind, ix.docs_dict), device.Dictionary(kind, ix.freqs_dict)
    dev = torch.device("cuda", 0)
    for cut in (len(ix.bytes) // 2, len(ix.bytes) // 3 + 1, 777):  # bytes of the index below the line
        shift = (1 << 32) - cut
        big = torch.zeros(shift + len(ix.bytes) + 16, dtype=torch.uint8, device=dev)
        big[shift:shift + len(ix.bytes)] = torch.from_numpy(ix.bytes).to(dev)
        moved = blocks.copy()
        moved["in_off"] += np.uint64(shift)
        table = device.BlockTable(dd, moved, big.numel())
        for taught in (False, True):
            if taught:
                table.learn(dd, fd, big, big.numel())
            for _ in range(3):
                docids_dev = torch.full((total,), -1, dtype=torch.int32, device=dev)
                freqs_dev = torch.full((total,), -1, dtype=torch.int32, device=dev)
                table.decode(dd, fd, big, big.numel(), docids_dev, freqs_dev)
                torch.cuda.synchronize()
                assert np.array_equal(docids_dev.cpu().numpy().view(np.uint32), ix.docids), (cut, taught)
                got_f = freqs_dev.cpu().numpy().view(np.uint32)
                bad = np.flatnonzero(got_f != ix.freqs)
                if bad.size:
                    b = int(np.searchsorted(blocks["out_off"], bad[0], side="right") - 1)
                    detail = dict(cut=cut, taught=taught, n_bad=int(bad.size), block=b, n=int(blocks["n"][b]), base=hex(big.data_ptr()),
                                  block_in_off=int(blocks["in_off"][b]), next_in_off=int(blocks["in_off"][b + 1]),
                                  pos_in_block=(bad[:24] - int(blocks["out_off"][b])).tolist(), got=got_f[bad[:6]].tolist(),
                                  want=ix.freqs[bad[:6]].tolist(), index_bytes=len(ix.bytes), info=table.info())
                    assert False, detail
        del table, big
        torch.cuda.empty_cache()
