"""The fuzz plans on the CPU: the generator's substitution (tests/fuzz_streams.py), the C oracle and the naive Python decoder
agree on random dictionary files x random decoder-legal slot streams no encoder emits — integers AND end offsets — and the
generator has not drifted from the digests committed under tests/golden/ (the GPU leg is tests/test_gpu_fuzz.py)."""
import json
import os

import numpy as np
import pytest

import fuzz_streams as F
import oracle
import pydecode

GOLDEN = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fuzz_digests.json")))
PARSE = {F.RECT: pydecode.parse_rectangular, F.SINGLE: pydecode.parse_single_packed, F.MULTI: pydecode.parse_multi_packed}
VROOM = F.plan(*GOLDEN["vroom_plan"])
INDEX = F.index_plan(*GOLDEN["index_plan"])


@pytest.mark.parametrize("case", VROOM, ids=lambda c: f"seed{c[0]}")
def test_vroom_case(case):
    D, S = F.build_case(case)
    want = GOLDEN["vroom"][str(case[0])]
    assert F.digest(D, S) == want["digest"] and len(S.lists) == want["lists"]
    od = oracle.OracleDict(D.kind, D.file)
    entry, enc = PARSE[D.kind](D.file), bytes(S.enc)
    # every list through the oracle: integers and the returned end pointer
    list_end = {}
    for li, (off, n, first) in enumerate(S.lists):
        got, used = od.decode_list(S.enc, off, n)
        assert np.array_equal(got, S.expect[first:first + n]), f"list {li}"
        list_end[li] = off + used
        if li % 12 == 0:  # and a sample through the third decoder
            py, end = (pydecode.decode_multi if D.kind == F.MULTI else pydecode.decode_single)(entry, enc, off, n)
            assert py == S.expect[first:first + n].tolist() and end == off + used
    # the whole stream through the oracle's framing loop (vroom_env/decode.cpp:139-150)
    out, lists = od.decode_stream(S.enc, len(S.expect))
    assert lists == len(S.lists) and np.array_equal(out, S.expect)
    # the generator's unit table: the last unit of every list ends where the oracle's decode of the list ended
    last = np.r_[S.units["list"][1:] != S.units["list"][:-1], True]
    assert np.array_equal(S.ends[last], np.array([list_end[int(l)] for l in S.units["list"][last]], dtype=np.uint64))
    # every unit decodes on its own to its stretch of the output (single: the oracle started at the unit's first slot)
    if D.kind != F.MULTI:
        for u in S.units[:: max(1, len(S.units) // 60)]:
            got, _ = od.decode_list(S.enc, int(u["in_off"]), int(u["n"]))
            assert np.array_equal(got, S.expect[int(u["out_off"]): int(u["out_off"]) + int(u["n"])])


@pytest.mark.parametrize("case", INDEX, ids=lambda c: f"seed{c[0]}")
def test_index_case(case):
    Dd, Df, X = F.build_index_case(case)
    want = GOLDEN["index"][str(case[0])]
    assert F.index_digest(Dd, Df, X) == want["digest"] and len(X.offsets) - 1 == want["lists"]
    od, of = oracle.OracleDict(Dd.kind, Dd.file), oracle.OracleDict(Df.kind, Df.file)
    ed, ef, raw = PARSE[Dd.kind](Dd.file), PARSE[Df.kind](Df.file), bytes(X.index)
    for i in range(len(X.offsets) - 1):
        d, f = oracle.posting_list_decode(od, of, X.index, int(X.offsets[i]))
        lo, hi = int(X.bounds[i]), int(X.bounds[i + 1])
        assert np.array_equal(d, X.docids[lo:hi]) and np.array_equal(f, X.freqs[lo:hi]), f"list {i}"
        if i % 10 == 0:
            pd, pf = pydecode.decode_posting_list(ed, ef, raw, int(X.offsets[i]), Dd.kind == F.MULTI)
            assert pd == X.docids[lo:hi].tolist() and pf == X.freqs[lo:hi].tolist()


def test_interpolative_writer_against_the_oracle():
    """The generator's own bit_writer (fuzz_streams._BitWriter) and the oracle's restatement of
    interpolative_block::encode write the same bytes; the Python reader reads them back."""
    r = np.random.default_rng(99)
    for _ in range(300):
        n = int(r.integers(1, 256))
        v = np.where(r.random(n) < 0.5, 0, r.integers(0, 1 << int(r.integers(1, 24)), n)).astype(np.uint32)
        for known in (True, False):
            mine = F.interpolative_block(v, known)
            theirs = oracle.interpolative_encode(v, int(v.sum()) if known else 0xFFFFFFFF)
            assert mine == bytes(theirs)
            back, end = pydecode.decode_interpolative(mine, 0, int(v.sum()) if known else 0xFFFFFFFF, n)
            assert back == v.tolist() and end == len(mine)
