"""C++ host mirror of the Go API (steroscopic-hardware_amd/host): builds and links on CPU, its host
logic runs under ThreadSanitizer over a CPU test double of the C ABI; on the GPU box its
run.go / run.stream.go counterparts must reproduce the oracle."""
import hashlib
import os
import re
import subprocess

import numpy as np
import pytest

import oracle
import steroscopic_hardware_amd as sha

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "steroscopic-hardware_amd")
LIB = os.path.join(PKG, "lib")
EXE = os.path.join(LIB, "run_stream")
GO = os.path.join(PKG, "host", "go")


def test_host_mirror_is_built_and_exports_the_go_api_names():
    assert os.path.exists(os.path.join(LIB, "libdespair_host.so")) and os.path.exists(EXE)
    syms = subprocess.check_output(["nm", "-DC", os.path.join(LIB, "libdespair_host.so")], text=True)
    for name in ("despair::SetupConcurrentSAD", "despair::AssembleDisparityMap", "despair::RunSad",
                 "despair::SumAbsoluteDifferences", "despair::SetDefaultParams", "despair::DefaultParams",
                 "despair::LoadPNG", "despair::SavePNG", "despair::LoadPicture", "despair::DevicePool::RunSadBatch"):
        assert name in syms, name


def test_go_shim_and_patches_are_reviewable():
    """f-2 without a Go toolchain: the shim keeps the reference's exported identifiers, carries no
    copy of the reference's worker loop (its fallback calls the reference's own functions), and the
    two patches apply to the files they name (checked with patch --dry-run where the reference tree
    exists; never at run time on the GPU box)."""
    hip = open(os.path.join(GO, "sad_hip.go")).read()
    cpu = open(os.path.join(GO, "sad_cpu.go")).read()
    assert hip.startswith("//go:build cgo && hip") and cpu.startswith("//go:build !(cgo && hip)")
    for name in ("func SetupConcurrentSAD(", "func RunSad(", "func RunSadFrame(", "func NewPinnedGray(", "func AssembleDisparityMap(", "FrameLevel"):
        assert name in hip and name in cpu, name
    assert "cpuRegion" not in hip and "SumAbsoluteDifferences(" not in hip        # no restated reference loop
    for ref in ("setupConcurrentSADGo(", "runSadGo("):
        assert ref in cpu
    assert "setupConcurrentSADGo(" in hip                                         # every fallback is the reference's own worker
    assert "assembleDisparityMapGo(" in cpu
    assert "sync.RWMutex" in hip and "runtime.LockOSThread()" in hip and 'os.Getenv("SADHIP_DEVICE")' in hip
    assert not os.path.exists(os.path.join(GO, "output_fastpath.go.txt"))
    patches = sorted(os.listdir(os.path.join(GO, "patches")))
    assert len(patches) == 2
    p1 = open(os.path.join(GO, "patches", patches[0])).read()
    p2 = open(os.path.join(GO, "patches", patches[1])).read()
    assert p1.startswith("--- a/pkg/despair/sad.go") and "+func setupConcurrentSADGo(" in p1 and "+func runSadGo(" in p1
    # the frame loop only READS the parameters: it calls RunSadFrame, which writes nothing back
    assert p2.startswith("--- a/pkg/camera/output.go") and "despair.RunSadFrame(leftImg, rightImg, *params)" in p2 and "despair.RunSad(" not in p2
    # each image travels with its OWN size (the reference tolerates a smaller right image, sad.go:225-233)
    assert "sadhip_disparity_lr_u8" in hip and "right.Rect.Dx()" in hip and "sadhip_disparity_region_u8" not in hip and "C.sadhip_disparity_u8" not in hip
    ref = "/root/reference"
    if os.path.isdir(os.path.join(ref, "pkg", "despair")):
        for pf in patches:
            r = subprocess.run(["patch", "-p1", "--dry-run", "--batch", "-d", ref, "-i", os.path.join(GO, "patches", pf)],
                               capture_output=True, text=True)
            assert r.returncode == 0, r.stdout + r.stderr


def test_host_logic_under_thread_sanitizer():
    """Context leases, channels, error propagation and the pool wrapper of host/despair.cpp over
    tests/host/fake_sadhip.cpp, compiled with -fsanitize=thread (sanitizers on the CPU build only)."""
    subprocess.check_call(["make", "-C", os.path.join(PKG, "host"), "tsan"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(LIB, "host_stress_tsan")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert re.search(r"OK contexts=\d+ calls=\d+ violations=0", r.stdout), r.stdout
    assert "ThreadSanitizer" not in r.stderr, r.stderr[-4000:]


def test_host_logic_under_address_sanitizer_with_mismatched_picture_sizes():
    """The same program under AddressSanitizer + UBSan.  Its phase 5 runs RunSad and the pipeline worker on
    left / right images of different sizes with the right image allocated at exactly its own size: a binding
    that passed the left image's dimensions for both (round 2's did) reads past it and fails here."""
    subprocess.check_call(["make", "-C", os.path.join(PKG, "host"), "asan-host"], stdout=subprocess.DEVNULL)
    r = subprocess.run([os.path.join(LIB, "host_stress_asan")], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0 halt_on_error=1 exitcode=67"))
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert re.search(r"OK contexts=\d+ calls=\d+ violations=0", r.stdout), r.stdout
    assert "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr[-4000:]


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


@pytest.mark.gpu
def test_run_stream_device_pool_and_failure_exit(tmp_path):
    """--devices 0,0: the native pool (two contexts on the one GPU) computes every frame; a device
    error makes run_stream exit 3 instead of hanging or writing a partly black map."""
    L, R = sha.synth.synth_pair(43, 400, 150)
    L.tofile(tmp_path / "l.raw"); R.tofile(tmp_path / "r.raw")
    args = [EXE, str(tmp_path / "l.raw"), str(tmp_path / "r.raw"), "400", "150", "16", "64", str(tmp_path / "o.raw"), "5"]
    out = subprocess.run(args + ["--devices", "0,0"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr + out.stdout
    assert "Pool of 2 contexts: 5 frames" in out.stdout and "Pool of 2 contexts, page-locked frames: 5 frames" in out.stdout
    assert np.array_equal(np.fromfile(tmp_path / "o.raw", np.uint8).reshape(150, 400), oracle.disparity_closed(L, R, 16, 64))
    bad = subprocess.run(args[:6] + ["0", str(tmp_path / "o2.raw"), "2"], capture_output=True, text=True, timeout=60)   # MaxDisparity 0
    assert bad.returncode == 3 and "despair error" in bad.stderr
    assert not os.path.exists(tmp_path / "o2.raw")
    nodev = subprocess.run(args + ["--devices", "0,77"], capture_output=True, text=True, timeout=60)
    assert nodev.returncode == 3


@pytest.mark.gpu
def test_run_keeps_pictures_in_memory_one_upload_per_image(tmp_path):
    """f-3: run (= examples/run.go) on the reference's own pair: decode -> ONE sadhip_disparity_rgba_u8
    call (uploads: one per image, gray on the device) -> SavePNG; digest of SURVEY 8c."""
    from PIL import Image
    golden = os.path.join(REPO, "tests", "golden")
    out_png = tmp_path / "out.png"
    r = subprocess.run([os.path.join(LIB, "run"), os.path.join(golden, "L_00001.png"), os.path.join(golden, "R_00001.png"),
                        str(out_png), "7", "32"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout
    got = np.array(Image.open(out_png))
    assert got.dtype == np.uint8 and got.shape == (480, 640)
    assert hashlib.sha256(got.tobytes()).hexdigest() == "03a658082e809b4211a1216839d23a7123f1187b56f18dff0b21b6863c8a9d6e"
    assert re.search(r"depth map generated elapsed=[\d.]+ms .*uploads 2\)", r.stdout), r.stdout
