import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    _ensure_built()


def pytest_sessionstart(session):
    """Which box is this?  One line on stderr (not captured: pytest.ini sets --capture=sys), so that a result -- or a native
    abort -- of a GPU run can be tied to the host, driver and device it happened on (round 4 saw box-to-box differences:
    DESIGN.md section 6 "box to box")."""
    try:
        import platform
        import subprocess
        import torch
        if not torch.cuda.is_available():
            return
        info = {"host": platform.node(), "kernel": platform.release()}
        for name, path in (("amdgpu", "/sys/module/amdgpu/version"), ("kfd_topology_nodes", "/sys/class/kfd/kfd/topology/generation_id")):
            try:
                info[name] = open(path).read().strip()
            except OSError:
                pass
        info["torch_hip"] = getattr(torch.version, "hip", None)
        info["device"] = torch.cuda.get_device_name(0)
        try:
            import steroscopic_hardware_amd as sha
            d = sha.capi.device_describe(0)
            info["pci"], info["numa_node"] = d["pci"], d["numa_node"]
        except Exception as e:                                    # never fail a session over the banner
            info["describe_error"] = repr(e)[:80]
        try:
            out = subprocess.run(["/opt/rocm/bin/rocm-smi", "--showfwinfo"], capture_output=True, text=True, timeout=20).stdout
            fw = {l.split(":")[0].split()[-2]: l.split(":")[-1].strip() for l in out.splitlines() if any(k in l for k in ("MEC firmware", "SDMA firmware", "SMC firmware", "VBIOS"))}
            info["firmware"] = fw
        except Exception:
            pass
        sys.stderr.write("[gpu box] " + " ".join(f"{k}={v}" for k, v in info.items()) + "\n")
        sys.stderr.flush()
    except Exception:
        pass


def _ensure_built():
    """Built artefacts are git-ignored: build whatever is missing (hipcc cross-compiles gfx950
    without a GPU), so a fresh checkout can run the suite directly."""
    import subprocess
    pkg = os.path.join(REPO, "steroscopic-hardware_amd")
    need = [
        (os.path.join(pkg, "lib", "libsadhip.so"), ["make", "-C", os.path.join(pkg, "csrc")]),
        (os.path.join(pkg, "lib", "run_stream"), ["make", "-C", os.path.join(pkg, "host")]),
        (os.path.join(pkg, "lib", "png_tool"), ["make", "-C", os.path.join(pkg, "host")]),
        (os.path.join(REPO, "oracle", "libdespair_oracle.so"), ["make", "-C", os.path.join(REPO, "oracle"), "libdespair_oracle.so"]),
    ]
    for path, cmd in need:
        if not os.path.exists(path):
            subprocess.check_call(cmd, stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def hwfx():
    """The reference's own hardware fixtures (tests/golden/make_golden.py)."""
    return dict(np.load(os.path.join(GOLDEN, "hardware_fixtures.npz")))


@pytest.fixture(scope="session")
def refout():
    """Outputs of the reference's own sad.c / test.py, generated in the build container."""
    return dict(np.load(os.path.join(GOLDEN, "reference_outputs.npz")))


@pytest.fixture(scope="session")
def testdata_pair():
    """testdata/L_00001.png + R_00001.png -> gray via the Go NRGBA formula (oracle)."""
    from PIL import Image
    import oracle
    out = []
    for n in ("L_00001.png", "R_00001.png"):
        im = Image.open(os.path.join(GOLDEN, n))
        assert im.mode == "RGBA"
        out.append(oracle.gray_from_nrgba(np.array(im)))
    return tuple(out)


@pytest.fixture(scope="session")
def testdata_rgba():
    from PIL import Image
    return tuple(np.array(Image.open(os.path.join(GOLDEN, n))) for n in ("L_00001.png", "R_00001.png"))


def load_real_pixels(name):
    """A picture of tests/golden/ as the u8 pixel array a Go decoder would hold (h, w, 4 or 3)."""
    from PIL import Image
    im = Image.open(os.path.join(GOLDEN, name))
    assert im.mode in ("RGBA", "RGB")
    return np.array(im)


def load_real_gray(name):
    """... and its gray plane by the Go formula the reference's callers apply (pkg/camera/output.go:142-162,
    pkg/despair/gray.go:43-58): 8-bit RGBA files decode to *image.NRGBA, opaque RGB files to *image.RGBA."""
    import oracle
    a = load_real_pixels(name)
    return oracle.gray_from_nrgba(a) if a.shape[2] == 4 else oracle.gray_from_rgb8(a)


def _real_pairs():
    import json
    with open(os.path.join(GOLDEN, "real_pairs.json")) as f:
        return json.load(f)


REAL_PAIRS = _real_pairs()          # every picture pair the reference holds in testdata/, with the pinned digests
REAL_CASES = [(k, bd) for k in sorted(REAL_PAIRS) for bd in sorted(REAL_PAIRS[k]["maps"], key=lambda s: tuple(map(int, s.split("/"))))]


# ---- host <-> device copies of the GPU tests --------------------------------------------------------------------------
# Through torch's page-locked allocator, never from / to pageable numpy memory: round 4 saw a process end with "Memory
# access fault by GPU ... on address <a host heap page>" while the HIP runtime was copying from pageable arrays it had
# locked in place.  The library no longer does that; the tests' own copies do not either.
def to_device(a):
    import torch
    t = torch.from_numpy(np.ascontiguousarray(a))
    p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    p.copy_(t)
    d = p.cuda()
    torch.cuda.synchronize()
    return d


def to_host(t):
    import torch
    p = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
    p.copy_(t)
    torch.cuda.synchronize()
    return p.numpy().copy()
