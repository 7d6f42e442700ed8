"""Build-time check on the generated gfx950 code (no GPU needed: hipcc cross-compiles).

One instruction form is banned: a packed fp32 op (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) whose DESTINATION register pair
is also a source that is read with an op_sel cross-selection.  hipcc produced it once, for the conv0 taps; that kernel gave
sporadically wrong values whenever kernels of another stream shared its CUs (tools/race_probe.py).  The
source no longer leads the compiler there; this test fails if a compiler or code change brings the form back anywhere."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "loco-asr_amd", "csrc")
PAT = re.compile(r"^\s*(v_pk_\w+)\s+(v\[\d+:\d+\]),\s*([^,]+),\s*([^,]+)(?:,\s*([^ ]+))?(.*)$")


def offenders(asm_path):
    bad = []
    for line in open(asm_path):
        m = PAT.match(line)
        if not m:
            continue
        _, dst, s0, s1, s2, mods = m.groups()
        sel = re.search(r"op_sel:\[([0-9,]+)\]", mods)
        if not sel:
            continue
        bits = [int(x) for x in sel.group(1).split(",")]
        for i, src in enumerate((s0.strip(), s1.strip(), (s2 or "").strip())):
            if i < len(bits) and bits[i] == 1 and src == dst:
                bad.append(line.strip())
    return bad


@pytest.mark.parametrize("src", sorted(os.path.basename(p) for p in glob.glob(os.path.join(CSRC, "*.hip"))))
def test_no_in_place_packed_op_with_cross_selected_source(src, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path / (src + ".s"))
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out],
                   check=True, capture_output=True)
    assert os.path.getsize(out) > 0
    bad = offenders(out)
    assert not bad, bad[:5]


def test_the_detector_sees_the_form():
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as fh:
        fh.write("\tv_pk_fma_f32 v[32:33], v[4:5], v[32:33], v[42:43] op_sel:[0,1,0]\n\tv_pk_fma_f32 v[36:37], v[36:37], v[34:35], s[2:3] op_sel_hi:[1,1,0]\n")
    assert len(offenders(fh.name)) == 1
    os.unlink(fh.name)
