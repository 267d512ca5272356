"""The C++ operator surface (dint/coders.hpp) and the dint_decode CLI on the GPU."""
import json
import os
import subprocess

import numpy as np
import pytest

from dint_amd import host

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TYPES = {host.RECTANGULAR: "single_rect_dint", host.SINGLE_PACKED: "single_packed_dint",
         host.MULTI_PACKED: "multi_packed_dint"}


@pytest.fixture(scope="module")
def roundtrip_binary(tmp_path_factory):
    exe = tmp_path_factory.mktemp("cpp") / "coder_roundtrip"
    subprocess.run(["g++", "-O2", "-std=c++17", f"-I{ROOT}/include", f"-I{ROOT}/dint_amd/csrc/host",
                    os.path.join(ROOT, "tests", "cpp", "coder_roundtrip.cpp"), "-o", str(exe),
                    f"-L{ROOT}/dint_amd", "-ldint_hip", f"-Wl,-rpath,{ROOT}/dint_amd"], check=True)
    return str(exe)


@pytest.mark.parametrize("mode", [0, 1, 2, 3])
def test_coder_call_shape_round_trip(roundtrip_binary, small_corpus, tmp_path, mode):
    kind = {0: host.RECTANGULAR, 1: host.SINGLE_PACKED, 2: host.MULTI_PACKED, 3: host.SINGLE_PACKED}[mode]
    (tmp_path / "dict.bin").write_bytes(small_corpus.dict_file(kind))
    bounds = small_corpus.coll.list_bounds()
    longest = int(np.argmax(small_corpus.coll.lens))
    small_corpus.coll.gaps[int(bounds[longest]):int(bounds[longest + 1])].tofile(tmp_path / "gaps.bin")
    r = subprocess.run([roundtrip_binary, str(mode), str(tmp_path / "dict.bin"), str(tmp_path / "gaps.bin")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


@pytest.mark.parametrize("kind", list(TYPES))
def test_decode_cli_reports_the_reference_keys(small_corpus, tmp_path, kind):
    enc, _ = small_corpus.encoded(kind)
    (tmp_path / "test.bin").write_bytes(enc.tobytes())
    (tmp_path / "dict.bin").write_bytes(small_corpus.dict_file(kind))
    exe = os.path.join(ROOT, "dint_amd", "bin", "dint_decode")
    r = subprocess.run([exe, TYPES[kind], str(tmp_path / "test.bin"), "--dict", str(tmp_path / "dict.bin")],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    line = json.loads(r.stdout.strip().splitlines()[-1])
    for key in ("filename", "num_sequences", "num_integers", "type", "tot_elapsed_time", "ns_x_int", "ints_x_sec"):
        assert key in line                                  # vroom_env/statistics.hpp:26-34
    assert int(line["num_integers"]) == small_corpus.coll.num_postings
    assert int(line["num_sequences"]) == np.count_nonzero(small_corpus.coll.lens)
    assert line["type"] == TYPES[kind] and int(line["ints_x_sec"]) > 0
