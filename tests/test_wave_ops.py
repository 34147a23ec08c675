"""wave_ops.h (the DPP wavefront reductions the latency-bound kernels use): a stand-alone HIP program, built with hipcc
on the GPU box and run there, compares them with sequential host sums."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.gpu
def test_dpp_reductions_match_host_sums(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = str(tmp_path / "wave_ops_test")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-o", exe,
                           os.path.join(HERE, "hip_unit", "wave_ops_test.hip")], stderr=subprocess.DEVNULL)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 bad" in r.stdout
