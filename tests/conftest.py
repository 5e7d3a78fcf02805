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
