"""The register budget of the persistent Winograd kernels, read from the ISA hipcc emits (no GPU needed: hipcc cross-compiles).

K10 / K17 run one wave per SIMD on all 512 registers; round 6 took the last spilled registers out of every instantiation by reading
the accumulators through asm ``v_accvgpr_read_b32`` and put the Winograd transforms on packed additions.
A change that makes hipcc spill again, or that silently drops the packed forms, shows here before it shows on a GPU."""
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "depthmodelhardening_amd", "csrc")


def _hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", shutil.which("hipcc")):
        if cand and os.path.exists(cand):
            return cand
    return None


def _isa(tmp_path, name):
    hipcc = _hipcc()
    if hipcc is None:
        pytest.skip("hipcc not found")
    out = os.path.join(str(tmp_path), name + ".s")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-I" + os.path.join(REPO, "include"),
                    "-I" + CSRC, "--offload-device-only", "-S", os.path.join(CSRC, name + ".hip"), "-o", out],
                   check=True, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    return open(out).read()


def _kernels(isa, stem):
    """{mangled name: (vgpr_spill_count, body text)} of the kernels whose name contains ``stem``."""
    spills = {m.group(1): int(m.group(2)) for m in
              re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", isa) if stem in m.group(1)}
    bodies = {m.group(1): m.group(2) for m in re.finditer(r"^(_Z\S+):.*?\n(.*?)\.Lfunc_end\d+:", isa, re.S | re.M) if stem in m.group(1)}
    return {k: (v, bodies.get(k, "")) for k, v in spills.items()}


@pytest.mark.parametrize("src,stem,count", [("wino_conv", "wino_conv_kernel", 12), ("wino32_conv", "wino32_conv_kernel", 2)])
def test_no_spilled_registers_and_packed_transforms(tmp_path, src, stem, count):
    ks = _kernels(_isa(tmp_path, src), stem)
    assert len(ks) == count, sorted(ks)
    for name, (spilled, body) in ks.items():
        assert spilled == 0, (name, spilled)
        assert "scratch_" not in body, name
        assert body.count("v_pk_add_f32") >= 100, (name, body.count("v_pk_add_f32"))    # both transforms + the epilogue
        reads = body.count("v_accvgpr_read_b32")
        assert 256 <= reads <= 2 * 256 + 64, (name, reads)      # every accumulator once per epilogue copy (whole / partial item)
