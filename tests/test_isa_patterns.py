"""Build-time check on the generated gfx950 code (no GPU needed: hipcc cross-compiles).

One instruction encoding is banned from every kernel of the library: a packed fp32 op (v_pk_fma_f32 / v_pk_mul_f32 /
v_pk_add_f32) with an `op_sel` bit set, i.e. whose LOW result lane sources the HIGH dword of a register pair.

Why ("the concurrency miscompare"; reproducer and raw data under tools/conv0_race/): a conv0 kernel whose ten
taps hipcc had compiled to `v_pk_fma_f32 ... op_sel:[0,1,0]` produced wrong even-channel outputs -- and only those -- whenever
workgroups of the attention kernel were resident on the same CUs.  Every wrong element equals the correct sum minus exactly
one tap, w[c][k] * x[k]; the taps that go missing are, frame parity by frame parity, precisely the taps whose instruction
carries op_sel:[0,1,0] in that half of the loop body (k = 1, 5, 9 for the first frame of a pair, all odd k for the second);
taps encoded with op_sel_hi only, the high lane and the odd channels are never affected.  The kernel is bit-exact alone and
beside GEMM / LayerNorm kernels.  The low lane's multiplicand is read as zero, sporadically, under that co-residency: an
undocumented hazard of this encoding as far as the ISA guide goes, so the library does not emit it -- the build uses
-fno-slp-vectorize (the forms came from the compiler's SLP packing of scalar code) and this test fails if the encoding
reappears anywhere."""
import glob
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "loco-asr_amd", "csrc")
PAT = re.compile(r"^\s*(v_pk_(?:fma|mul|add)_f32)\s+.*\bop_sel:\[([0-9,]+)\]")


def makefile_flags():
    """The flags the library is really built with (loco-asr_amd/csrc/Makefile), minus what -S does not take."""
    text = open(os.path.join(CSRC, "Makefile")).read()
    flags = re.search(r"^CXXFLAGS \?= (.*)$", text, re.M).group(1).replace("$(ARCH)", "gfx950").split()
    flags += re.search(r"^override CXXFLAGS \+= (.*)$", text, re.M).group(1).split()  # what no environment can take away
    return [f for f in flags if f not in ("-fPIC",)]


def offenders(asm_path):
    bad = []
    for line in open(asm_path):
        m = PAT.match(line)
        if m and "1" in m.group(2):
            bad.append(line.strip())
    return bad


@pytest.mark.parametrize("src", sorted(os.path.basename(p) for p in glob.glob(os.path.join(CSRC, "*.hip"))))
def test_no_packed_fp32_op_cross_selects_its_low_lane(src, tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path / (src + ".s"))
    subprocess.run([hipcc] + makefile_flags() + ["-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out], check=True, capture_output=True)
    assert os.path.getsize(out) > 0
    bad = offenders(out)
    assert not bad, bad[:5]


def test_the_build_disables_slp_packing():
    assert "-fno-slp-vectorize" in makefile_flags()
    text = open(os.path.join(CSRC, "Makefile")).read()
    assert re.search(r"^override CXXFLAGS \+= .*-fno-slp-vectorize", text, re.M), "a CXXFLAGS from the environment must not drop the flag"
    assert "check_isa.py $@.tmp" in text, "`make` must inspect the linked library before installing it"


def test_the_library_that_is_loaded_carries_no_banned_encoding():
    """Not the flags, the product: disassemble the gfx950 code objects inside the libloco_asr.so the package loads."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(CSRC, "check_isa.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    lib = os.path.join(ROOT, "loco-asr_amd", "libloco_asr.so")
    if not os.path.exists(os.path.join(mod.LLVM, "llvm-objdump")):
        pytest.skip("llvm-objdump not available")
    n, mfma, bad = mod.check(lib)
    assert n >= 9 and mfma > 4000, (n, mfma)  # every kernel translation unit is in there, matrix instructions included
    assert not bad, bad[:5]
    assert mod.offenders_in_text("\tv_pk_fma_f32 v[32:33], v[4:5], v[32:33], v[42:43] op_sel:[0,1,0]   // 000000001A2C: D3B04020\n")


def test_the_split_precision_kernels_use_no_scratch():
    """Every instantiation of the default mode's kernels keeps its state in registers: a spill inside the GEMM's k-loop cost 8 % when
    it happened, and an argument struct that is indexed at run time lands in scratch silently (the q | k | v scatter did,
    with six plane pointers to choose from, until it addressed its planes by one stride).  Read from the metadata of the code objects
    in the library that is loaded; the exact-fp32 GEMM's 36 bytes (blocked accumulation) are the one known exception."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_isa", os.path.join(CSRC, "check_isa.py"))
    check_isa = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(check_isa)
    lib = os.path.join(ROOT, "loco-asr_amd", "libloco_asr.so")
    if not os.path.exists(lib) or not os.path.exists(os.path.join(check_isa.LLVM, "llvm-readelf")):
        pytest.skip("library or llvm-readelf not available")
    users = check_isa.scratch_users(lib)
    unexpected = {k: v for k, v in users.items() if "gemm_f32_kernel" not in k}
    assert not unexpected, unexpected
    assert all(v <= 64 for v in users.values()), users


def test_the_detector_sees_the_form():
    import tempfile
    with tempfile.NamedTemporaryFile("w", suffix=".s", delete=False) as fh:
        fh.write("\tv_pk_fma_f32 v[32:33], v[4:5], v[32:33], v[42:43] op_sel:[0,1,0]\n"           # in place, cross-selected
                 "\tv_pk_fma_f32 v[32:33], v[12:13], v[40:41], v[32:33] op_sel:[0,1,0]\n"          # not in place: just as wrong
                 "\tv_pk_add_f32 v[18:19], v[18:19], v[18:19] op_sel:[0,1] op_sel_hi:[1,0]\n"      # the SLP vectoriser's horizontal add
                 "\tv_pk_fma_f32 v[36:37], v[36:37], v[34:35], s[2:3] op_sel_hi:[1,1,0]\n"         # fine: high lane reading a low dword
                 "\tv_pk_fma_f32 v[36:37], v[22:23], v[32:33], 0 op_sel_hi:[1,0,0]\n")             # fine
    assert len(offenders(fh.name)) == 3
    os.unlink(fh.name)


@pytest.mark.parametrize("src", ["gemm_f16x3.hip", "attention_f16x3.hip"])
def test_m0_is_written_only_for_the_lds_dma(src, tmp_path):
    """gemm_f16x3.hip and attention_f16x3.hip issue their LDS-DMA as inline asm that writes M0 (the piece's LDS address) without declaring the clobber --
    hipcc treats M0 as reserved and rejects it in a clobber list.  That is sound only while the compiler itself never keeps a value
    in M0 in that translation unit: every instruction that names m0 must be the asm's own `s_mov_b32 m0, sN`, followed (after its
    one wait state) by the global_load_lds_dwordx4 that consumes it."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    out = str(tmp_path / (src + ".s"))
    subprocess.run([hipcc] + makefile_flags() + ["-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", out], check=True,
                   capture_output=True)
    lines = [ln.split(";")[0].strip() for ln in open(out)]
    lines = [ln for ln in lines if ln and not ln.startswith(".")]
    uses = [i for i, ln in enumerate(lines) if re.search(r"\bm0\b", ln)]
    assert len(uses) > 30  # the DMAs are there
    for i in uses:
        assert re.fullmatch(r"s_mov_b32 m0, (s\d+|vcc_lo|vcc_hi|ttmp\d+)", lines[i]), lines[i]  # the asm's "s" operand: any scalar register
        assert lines[i + 1] == "s_nop 0" and lines[i + 2].startswith("global_load_lds_dwordx4 v"), lines[i:i + 3]
