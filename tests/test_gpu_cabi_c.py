"""The boundary from a plain C host: examples/cabi_forward.c is compiled against include/loco_asr.h, linked with
libloco_asr.so and run as a separate process -- no Python, no torch in it.  Its output file must equal what the Python
wrapper returns for the same weights and audio bit for bit (both are the same loco_forward call)."""
import os
import shutil
import subprocess

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_c_host_reproduces_the_python_wrapper(tmp_path):
    from gpu_util import la, model
    gcc = shutil.which("gcc")
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    exe = str(tmp_path / "cabi_forward")
    subprocess.run([gcc, "-O2", os.path.join(ROOT, "examples", "cabi_forward.c"), "-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(rocm, "include"), "-D__HIP_PLATFORM_AMD__", "-L", os.path.join(ROOT, "loco-asr_amd"), "-lloco_asr",
                    "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "loco-asr_amd"),
                    "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", exe], check=True)
    m, sd = model()
    with open(tmp_path / "weights.bin", "wb") as fw, open(tmp_path / "manifest.txt", "w") as fm:
        for k, v in sd.items():
            v = np.ascontiguousarray(v, dtype=np.float32)
            fw.write(v.tobytes())
            fm.write(f"{k} {v.ndim} {' '.join(str(d) for d in v.shape)}\n")
    B, L = 2, 24000
    x, _ = la.synth.batch([L] * B)
    x.astype(np.float32).tofile(tmp_path / "wave.f32")
    res = subprocess.run([exe, str(tmp_path / "weights.bin"), str(tmp_path / "manifest.txt"), str(tmp_path / "wave.f32"), str(B), str(L),
                          str(tmp_path / "out.f32")], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr
    T = la.synth.conv_out_length(L)
    got = np.fromfile(tmp_path / "out.f32", dtype=np.float32).reshape(B, T, 768)
    # the C host uploads no sinusoid table, so positions come from the library's own generator: same values up to the
    # last bit of sinf/cosf, not necessarily the same bits as torch's table
    ref = m.speecht5.encoder(input_values=torch.from_numpy(x).cuda()).last_hidden_state.cpu().numpy()
    rel = np.linalg.norm(got.astype(np.float64) - ref) / np.linalg.norm(ref)
    assert rel < 2e-6, rel
    assert "workspace" in res.stdout
    assert "3 forwards in flight on 3 streams: 3 of 3 bit-identical" in res.stdout, res.stdout
    assert "packed forward of 5 clips (batches of 2, 1, 2)" in res.stdout and "the short batch has 37 frames" in res.stdout, res.stdout
