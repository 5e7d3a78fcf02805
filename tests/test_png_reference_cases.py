"""The reference's own tests for this path are pkg/despair/png_test.go (the SAD
functions have none).  Same cases, same fixtures (generated in the test, 10x10), against the
Python mirror of png.go.  Go's encoder writes both its *image.RGBA and its opaque *image.NRGBA
fixture as 8-bit RGB, which LoadPNG then decodes to *image.RGBA -> convertRGBAToGray -> all zeros
(gray.go:20-40); TestEndToEnd compares zeros with zeros.  Here the default is the luminance and
refbug_rgba=True is that behaviour."""
import os

import numpy as np
import pytest
from PIL import Image

import oracle
import steroscopic_hardware_amd as sha

despair = sha.despair


def _gray_fixture(path):                                   # createGrayTestImage, png_test.go:257-270
    y, x = np.mgrid[0:10, 0:10]
    Image.fromarray(((x + y) % 256).astype(np.uint8), "L").save(path)


def _rgb_fixture(path):                                    # createRGBATestImage / createNRGBATestImage, :272-308
    y, x = np.mgrid[0:10, 0:10]
    rgb = np.stack([x * 25 % 256, y * 25 % 256, (x + y) * 15 % 256], axis=-1).astype(np.uint8)
    Image.fromarray(rgb, "RGB").save(path)                 # opaque: Go writes colour type 2 for both
    return rgb


def test_load_png_gray_and_errors(tmp_path):               # TestLoadPNG (the cases that need no device)
    _gray_fixture(tmp_path / "gray.png")
    got = despair.LoadPNG(str(tmp_path / "gray.png"))
    assert got is not None and got.shape == (10, 10)
    y, x = np.mgrid[0:10, 0:10]
    assert np.array_equal(got, (x + y) % 256)
    with pytest.raises(Exception):
        despair.LoadPNG(str(tmp_path / "nonexistent.png"))
    (tmp_path / "invalid.png").write_bytes(b"not a png")
    with pytest.raises(Exception):
        despair.LoadPNG(str(tmp_path / "invalid.png"))
    with pytest.raises(Exception):                          # TestMustLoadPNG "Should panic on error"
        despair.MustLoadPNG(str(tmp_path / "nonexistent.png"))


def test_save_png(tmp_path):                               # TestSavePNG, TestMustSavePNG
    y, x = np.mgrid[0:10, 0:10]
    img = (x * y % 256).astype(np.uint8)
    despair.SavePNG(str(tmp_path / "valid.png"), img)
    assert os.path.getsize(tmp_path / "valid.png") > 0      # png_test.go:136-160: file exists, decodes, same bounds
    back = Image.open(tmp_path / "valid.png")
    assert back.size == (10, 10) and back.mode == "L" and np.array_equal(np.array(back), img)
    with pytest.raises(Exception):
        despair.SavePNG(str(tmp_path / "nonexistent" / "test.png"), img)
    despair.MustSavePNG(str(tmp_path / "must.png"), img)
    with pytest.raises(Exception):
        despair.MustSavePNG(str(tmp_path / "nonexistent" / "test.png"), img)


@pytest.mark.gpu
def test_load_png_rgba_and_nrgba(tmp_path):                # TestLoadPNG "Load RGBA image" / "Load NRGBA image"
    rgb = _rgb_fixture(tmp_path / "rgba.png")
    for kw in ({}, {"refbug_rgba": True}):
        got = despair.LoadPNG(str(tmp_path / "rgba.png"), **kw)
        assert got is not None and got.shape == (10, 10)
    assert np.array_equal(despair.LoadPNG(str(tmp_path / "rgba.png")), oracle.gray_from_rgb8(rgb))
    assert np.array_equal(despair.LoadPNG(str(tmp_path / "rgba.png"), refbug_rgba=True), oracle.gray_from_rgb8(rgb, refbug=True))
    rgba = np.dstack([rgb, np.full((10, 10), 200, np.uint8)])          # a translucent NRGBA file takes the generic path
    Image.fromarray(rgba, "RGBA").save(tmp_path / "nrgba.png")
    assert np.array_equal(despair.LoadPNG(str(tmp_path / "nrgba.png")), oracle.gray_from_nrgba(rgba))


@pytest.mark.gpu
@pytest.mark.parametrize("refbug", [False, True])
def test_end_to_end(tmp_path, refbug):                     # TestEndToEnd, png_test.go:202-253
    _rgb_fixture(tmp_path / "original.png")
    img = despair.LoadPNG(str(tmp_path / "original.png"), refbug_rgba=refbug)
    despair.SavePNG(str(tmp_path / "saved.png"), img)
    reloaded = despair.LoadPNG(str(tmp_path / "saved.png"))
    assert reloaded.shape == img.shape
    for y in range(0, 10, 3):
        for x in range(0, 10, 3):
            assert reloaded[y, x] == img[y, x]
    assert img.any() != refbug                              # the reference's run of this test sees only zeros
