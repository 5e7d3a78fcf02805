"""C++ host mirror of the Go API (steroscopic-hardware_amd/host): builds and links on CPU;
on the GPU box its run.go/run.stream.go counterpart must reproduce the oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle
import steroscopic_hardware_amd as sha

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "steroscopic-hardware_amd", "lib")
EXE = os.path.join(LIB, "run_stream")


def test_host_mirror_is_built_and_exports_the_go_api_names():
    assert os.path.exists(os.path.join(LIB, "libdespair_host.so")) and os.path.exists(EXE)
    syms = subprocess.check_output(["nm", "-DC", os.path.join(LIB, "libdespair_host.so")], text=True)
    for name in ("despair::SetupConcurrentSAD", "despair::AssembleDisparityMap", "despair::RunSad",
                 "despair::SumAbsoluteDifferences", "despair::SetDefaultParams", "despair::DefaultParams",
                 "despair::LoadPNG", "despair::SavePNG"):
        assert name in syms, name
    go = open(os.path.join(REPO, "steroscopic-hardware_amd", "host", "go", "sad_hip.go")).read()
    for name in ("func SetupConcurrentSAD(", "func RunSad(", "func AssembleDisparityMap(", "//go:build cgo && hip"):
        assert name in go, name


@pytest.mark.gpu
def test_run_stream_matches_oracle(tmp_path):
    L, R = sha.synth.synth_pair(42, 320, 200)
    L.tofile(tmp_path / "l.raw"); R.tofile(tmp_path / "r.raw")
    out = subprocess.run([EXE, str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), "320", "200", "9", "64",
                          str(tmp_path / "o.raw"), "2"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    got = np.fromfile(tmp_path / "o.raw", np.uint8).reshape(200, 320)
    assert np.array_equal(got, oracle.disparity_closed(L, R, 9, 64))
    assert out.stdout.count("Elapsed time:") == 2
    assert f"= {oracle.sum_absolute_differences(L, R, 100, 50, 95, 50, 9)}" in out.stdout
