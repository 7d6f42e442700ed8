#!/usr/bin/env python3
"""Check the gfx950 code objects INSIDE a built libloco_asr.so for the banned instruction encoding.

    python3 check_isa.py ../libloco_asr.so        (run by `make` on every link, by tests/test_isa_patterns.py on the library the
                                                    package loads, and by _lib.load() when LOCO_ASR_LIB points at another build)

Banned: a packed fp32 op (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) with an `op_sel` bit set, i.e. whose LOW result lane sources
the HIGH dword of a register pair.  On gfx950 that lane's product was measured to be lost, sporadically, while waves of the attention
kernel share the SIMD (tools/conv0_race/).  The cause is bounded by evidence (which encoding, which co-runner, which
lane), not proven from documentation -- so the guard inspects what is actually shipped, not the flags it was meant to be built with.
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

PAT = re.compile(r"\b(v_pk_(?:fma|mul|add)_f32)\s+.*\bop_sel:\[([0-9,]+)\]")
LLVM = os.environ.get("LOCO_LLVM_BIN", "/opt/rocm/lib/llvm/bin")


def offenders_in_text(text):
    bad = []
    for line in text.splitlines():
        m = PAT.search(line)
        if m and "1" in m.group(2):
            bad.append(line.strip())
    return bad


def disassemble(lib_path):
    """-> (number of gfx950 code objects, their disassembly).  Raises when the tools or the code objects are missing."""
    objdump = os.path.join(LLVM, "llvm-objdump")
    if not os.path.exists(objdump):
        raise FileNotFoundError(objdump)
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, local)
        subprocess.run([objdump, "--offloading", local], check=True, capture_output=True, cwd=tmp)  # writes lib.so.<n>.<triple> files
        objs = sorted(f for f in os.listdir(tmp) if "amdgcn" in f and f.endswith("gfx950"))
        if not objs:
            raise RuntimeError(f"{lib_path}: no gfx950 code object found")
        text = []
        for f in objs:
            r = subprocess.run([objdump, "-d", "--mcpu=gfx950", os.path.join(tmp, f)], check=True, capture_output=True, text=True)
            text.append(r.stdout)
        return len(objs), "\n".join(text)


def scratch_users(lib_path):
    """-> {kernel symbol: private segment bytes} for every kernel of the library's gfx950 code objects that uses scratch (register
    spills or dynamically indexed locals), from the code objects' metadata notes."""
    readelf = os.path.join(LLVM, "llvm-readelf")
    objdump = os.path.join(LLVM, "llvm-objdump")
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib_path, local)
        subprocess.run([objdump, "--offloading", local], check=True, capture_output=True, cwd=tmp)
        for f in sorted(f for f in os.listdir(tmp) if "amdgcn" in f and f.endswith("gfx950")):
            notes = subprocess.run([readelf, "--notes", os.path.join(tmp, f)], check=True, capture_output=True, text=True).stdout
            name = None
            for line in notes.splitlines():
                m = re.match(r"\s*\.name:\s+(\S+)", line)
                if m:
                    name = m.group(1)
                m = re.match(r"\s*\.private_segment_fixed_size:\s+(\d+)", line)
                if m and name and int(m.group(1)):
                    out[name] = int(m.group(1))
    return out


def check(lib_path):
    n, text = disassemble(lib_path)
    return n, text.count("v_mfma_"), offenders_in_text(text)


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "libloco_asr.so")
    n, mfma, bad = check(path)
    if bad:
        print(f"check_isa: {path}: {len(bad)} packed fp32 instructions cross-select their low lane (banned, see the docstring):", file=sys.stderr)
        for line in bad[:10]:
            print("   ", line, file=sys.stderr)
        sys.exit(1)
    print(f"check_isa: {os.path.basename(path)}: {n} gfx950 code objects, {mfma} MFMA instructions, no banned packed-fp32 op_sel encoding")
