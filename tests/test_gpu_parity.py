"""GPU parity: the HIP decode path (through the C ABI) against the CPU oracle, bit for bit."""
import numpy as np
import pytest

import oracle
from dint_amd import host

pytestmark = pytest.mark.gpu

SINGLE_KINDS = [host.SINGLE_PACKED, host.RECTANGULAR]


@pytest.fixture(scope="module")
def device():
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    from dint_amd import device as dev  # fails loudly if libdint_hip.so is missing

    return dev


def test_wave_scan(device):
    r = np.random.default_rng(0)
    for _ in range(8):
        v = r.integers(0, 300, 64).astype(np.uint32)
        assert np.array_equal(device.debug_wave_scan(v), np.cumsum(v, dtype=np.uint32))
    assert np.array_equal(device.debug_wave_scan(np.ones(64)), np.arange(1, 65, dtype=np.uint32))


@pytest.mark.parametrize("kind", SINGLE_KINDS)
@pytest.mark.parametrize("corpus_name", ["small_corpus", "dense_corpus", "sparse_corpus"])
def test_stream_matches_oracle(device, request, kind, corpus_name):
    corpus = request.getfixturevalue(corpus_name)
    enc, units = corpus.encoded(kind)
    d = device.Dictionary(kind, corpus.dict_file(kind))
    out, ends, _ = device.decode_stream(d, enc, units, corpus.coll.num_postings)
    want, _ = oracle.OracleDict(kind, corpus.dict_file(kind)).decode_stream(enc, corpus.coll.num_postings)
    assert np.array_equal(want, corpus.coll.gaps)          # the oracle agrees with the encoder's input
    assert np.array_equal(out, want)                       # the device agrees with the oracle
    # end offsets: each unit ends where the next unit of the same list starts
    same = units["list"][1:] == units["list"][:-1]
    assert np.array_equal(ends[:-1][same], units["in_off"][1:][same])
    assert ends[-1] == enc.size


@pytest.mark.parametrize("kind", SINGLE_KINDS)
def test_index_stream_equals_encoder_sidecar(device, small_corpus, kind):
    """The host pre-pass finds the same list boundaries the encoder recorded."""
    enc, units = small_corpus.encoded(kind)
    d = device.Dictionary(kind, small_corpus.dict_file(kind))
    for unit_ints in (0, 256, 4096):
        idx, total, lists = d.index_stream(enc, unit_ints)
        assert total == small_corpus.coll.num_postings
        assert lists == np.count_nonzero(small_corpus.coll.lens)
        assert int(idx["n"].sum()) == total
        out, _, _ = device.decode_stream(d, enc, idx, total)
        assert np.array_equal(out, small_corpus.coll.gaps)
    first = np.r_[True, units["list"][1:] != units["list"][:-1]]
    idx0, _, _ = d.index_stream(enc, 0)
    assert np.array_equal(idx0["in_off"], units["in_off"][first])


@pytest.mark.parametrize("kind", SINGLE_KINDS)
def test_decode_list_call_shape(device, small_corpus, kind):
    """Coder::decode(dict, in, out, universe, n) -> in_end, one list at a time."""
    enc, _ = small_corpus.encoded(kind)
    od = oracle.OracleDict(kind, small_corpus.dict_file(kind))
    d = device.Dictionary(kind, small_corpus.dict_file(kind))
    off = 0
    checked = 0
    while off < enc.size and checked < 40:
        n, _, payload = oracle.header_read(enc, off)
        want, used = od.decode_list(enc, payload, n)
        got, consumed = d.decode_list(enc, payload, n)
        assert np.array_equal(got, want)
        assert consumed == used
        off = payload + used
        checked += 1
