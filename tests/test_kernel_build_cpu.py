"""What the decode kernels' BUILD has to look like (CPU: hipcc cross-compiles gfx950, ~25 s): properties a source change can
break without any test's integers changing — until a run under memory pressure does.

* The chunked bundle loops request the next bundle's stream bytes with an inline-asm buffer load the compiler does not track
  and wait for it with a counted s_waitcnt (kernels/bundles.inc, bundle_raw_async / wait_all_but_youngest): no
  compiler-generated instruction may touch the load's destination registers between the two (tools/check_inflight.py).
* decode_multi_bundles_kernel's chunk touch loads into v126 / v127, which amdgpu_num_vgpr(126) keeps out of the register
  allocator's hands: nothing outside inline asm may name them.
* No spilled vector register in the vroom kernels' loops (a reload is a wait for every store in flight: DESIGN.md 4e)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC") or shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")


def _tool():
    spec = importlib.util.spec_from_file_location("check_inflight", os.path.join(ROOT, "tools", "check_inflight.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="module")
def build(tmp_path_factory):
    tool = _tool()
    return tool, tool.compile_to_asm(str(tmp_path_factory.mktemp("isa")))


ASM_LOAD_KERNELS = ["decode_multi_bundles_kernel", "decode_single_index_bundles_kernel", "decode_multi_index_bundles_kernel",
                    "decode_single_index_pair_kernel", "decode_multi_index_pair_kernel"]


@pytest.mark.parametrize("kernel", ASM_LOAD_KERNELS)
def test_nothing_touches_a_register_an_asm_load_is_still_writing(build, kernel):
    tool, text = build
    n_loads, n_waits, bad = tool.scan(tool.kernel_body(text, kernel))
    assert n_loads >= 1 and n_waits >= 1, "the bundle loop's asm-issued load / counted wait are gone from this kernel"
    assert not bad, "\n".join(f"line {ln}: {code} touches in-flight v{t} (loaded at line {at})" for ln, code, t, at in bad)


def test_the_chunk_touch_registers_are_the_asm_s_alone(build):
    tool, text = build
    body = tool.kernel_body(text, "decode_multi_bundles_kernel")
    assert "v126" in body and "v127" in body, "the chunk touch is gone"
    assert tool.untracked_uses(body, (126, 127)) == []


def test_register_and_spill_ceilings(build):
    tool, text = build
    res = {k.split("dint_dev")[1]: v for k, v in tool.kernel_resources(text).items() if "dint_dev" in k}

    def of(name):
        hit = [v for k, v in res.items() if name + "E" in k]
        assert len(hit) == 1, name
        return hit[0]

    for name in ("decode_single_kernel", "decode_multi_bundles_kernel"):  # the two throughput kernels: nothing spilled, no scratch
        r = of(name)
        assert r["vgpr_spill_count"] == 0 and r["private_segment_fixed_size"] == 0 and r["vgpr_count"] <= 128, (name, r)
    for name in ASM_LOAD_KERNELS[1:]:  # in-index bundle / pair kernels: no spilled vector register either
        assert of(name)["vgpr_spill_count"] == 0, (name, of(name))
    for k, r in res.items():  # 16 waves a CU: a kernel of the 1024-thread family over 128 registers would not launch
        if "decode_" in k:
            assert r["vgpr_count"] <= 128, (k, r)
