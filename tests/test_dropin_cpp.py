"""The C++ drop-in adapter (stereo_amd/adapter/stereo_dropin.hpp): compile check on CPU,
run against the oracle on the GPU.  cv::Mat / Eigen::Matrix3f come from TEST-ONLY stand-ins
under tests/cpp/mock (OpenCV and Eigen are not installed in this image)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def test_adapter_compiles_against_the_c_abi():
    subprocess.check_call(["make", "-C", CPP, "-B", "test_dropin"], stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(CPP, "test_dropin"))


@pytest.mark.gpu
def test_adapter_matches_oracle_on_gpu():
    exe = os.path.join(CPP, "test_dropin")
    assert os.path.exists(exe), "build it first: __graft_entry__.build()"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "DROPIN OK" in out.stdout, out.stdout + out.stderr
