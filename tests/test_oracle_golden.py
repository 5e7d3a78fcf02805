"""Pins the CPU oracle (oracle/despair_oracle.c) before anything is compared against it:
the reference's own fixtures, outputs of the reference's own C and Python run in the build
container, the survey digests, and literal == closed form.  No GPU."""
import hashlib
import os

import numpy as np
import pytest

import oracle
from conftest import REAL_PAIRS, load_real_gray

# region where Go / sad.c / test.py semantics coincide (last paragraph)
WIN, D = 15, 64
HS = WIN // 2
XS = slice(D + HS, 128 - HS)          # 71 <= x < 121


def test_reference_sadc_run_matches_committed_fixture(hwfx, refout):
    # hardware/sad.c compiled unmodified (oracle/_ref/sad_ref) reproduces hardware/exp_disp.mem
    assert np.array_equal(refout["sadc_run_exp_disp"], hwfx["exp_disp_sadc"])
    assert np.array_equal(refout["sadc_run_pgm"], hwfx["disparity_pgm"])


def test_reference_testpy_run_matches_committed_fixture(hwfx, refout):
    assert np.array_equal(refout["testpy_patches"], hwfx["exp_disp_p"])
    assert np.array_equal(hwfx["left_image_p"], hwfx["patch_L"])
    assert np.array_equal(hwfx["right_image_p"], hwfx["patch_R"])


def test_oracle_raw_vs_sadc_fixture(hwfx):
    """hardware/exp_disp.mem (sad.c semantics, patch 1): equals the Go function's raw best d on
    7 <= y < 121, 71 <= x < 121."""
    L, R = hwfx["patch_L"][1], hwfx["patch_R"][1]
    got = oracle.disparity_literal(L, R, WIN, D, raw=True)
    assert np.array_equal(got[HS:128 - HS, XS], hwfx["exp_disp_sadc"][HS:128 - HS, XS])


@pytest.mark.parametrize("k", range(4))
def test_oracle_raw_vs_testpy_fixture(hwfx, k):
    """hardware/mems/exp_disp_p.mem (test.py zero-pad semantics), all rows, 71 <= x < 121."""
    L, R = hwfx["patch_L"][k], hwfx["patch_R"][k]
    got = oracle.disparity_literal(L, R, WIN, D, raw=True)
    assert np.array_equal(got[:, XS], hwfx["exp_disp_p"][k][:, XS])
    # scaled output = best*255/D (sad.go:91-93)
    sc = oracle.disparity_literal(L, R, WIN, D)
    assert np.array_equal(sc, (got.astype(np.int32) * 255 // D).astype(np.uint8))


def test_mirror_sadc_full_patch(hwfx):
    L, R = hwfx["patch_L"][1], hwfx["patch_R"][1]
    assert np.array_equal(oracle.mirror_sadc(L, R, WIN, D), hwfx["exp_disp_sadc"])


@pytest.mark.parametrize("k", range(4))
def test_mirror_testpy_full_patch(hwfx, k):
    assert np.array_equal(oracle.mirror_testpy(hwfx["patch_L"][k], hwfx["patch_R"][k], WIN, D), hwfx["exp_disp_p"][k])


def test_mirror_testpy_synthetic_cases(refout):
    n = int(refout["testpy_ncases"][0])
    assert n >= 5
    for i in range(n):
        L, R = refout[f"testpy_case{i}_L"], refout[f"testpy_case{i}_R"]
        win, maxd = (int(v) for v in refout[f"testpy_case{i}_params"])
        exp = refout[f"testpy_case{i}_out"]
        assert np.array_equal(oracle.mirror_testpy(L, R, win, maxd), exp)
        # and the Go function agrees with test.py wherever windows are full and all d valid
        hs = win // 2
        h, w = L.shape
        if w - hs > maxd + hs:
            got = oracle.disparity_literal(L, R, win, maxd, raw=True)
            assert np.array_equal(got[:, maxd + hs:w - hs], exp[:, maxd + hs:w - hs])


def test_verilog_strip_fixtures(hwfx):
    """mems/exp_disp_n.mem = 12 12 15 0F and old_mems/exp_disp.mem = 23 0E 29 31
    (Verilog convention: right window slides to larger columns, d < 50)."""
    for L, R, exp in ((hwfx["strips_n_L"], hwfx["strips_n_R"], hwfx["exp_disp_n"]),
                      (hwfx["strips_old_L"], hwfx["strips_old_R"], hwfx["exp_disp_old"])):
        for k in range(4):
            d, _ = oracle.mirror_verilog_strip(L[k], R[k], 15, 64)
            assert d == int(exp[k])
    assert [int(v) for v in hwfx["exp_disp_n"]] == [0x12, 0x12, 0x15, 0x0F]
    assert [int(v) for v in hwfx["exp_disp_old"]] == [0x23, 0x0E, 0x29, 0x31]


def test_verilog_image_mirror_is_the_strip_search_at_every_pixel(hwfx):
    """oracle.mirror_verilog (what the GPU's MIRROR_VERILOG mode is checked against) on a testbench
    strip gives the fixture value at (x, y) = (WIN/2, WIN/2) = col_index 0, and equals the strip
    function at every other col_index too."""
    for L, R, exp in ((hwfx["strips_n_L"], hwfx["strips_n_R"], hwfx["exp_disp_n"]),
                      (hwfx["strips_old_L"], hwfx["strips_old_R"], hwfx["exp_disp_old"])):
        for k in range(4):
            img = oracle.mirror_verilog(L[k], R[k], 15, 64)
            assert int(img[7, 7]) == int(exp[k])
            assert not img[:7].any() and not img[8:].any() and not img[:, :7].any() and not img[:, 57:].any()
            for ci in (1, 5, 20, 49):                      # col_index ci: the strip starting at column ci
                d, _ = oracle.mirror_verilog_strip(np.ascontiguousarray(L[k][:, ci:]), np.ascontiguousarray(R[k][:, ci:]), 15, 64)
                assert int(img[7, 7 + ci]) == d


def test_single_window_sad_kats(hwfx):
    """old_mems/exp_sad.mem = SAD of the first 15x15 window (d = 0) of each strip;
    winL/winR.mem -> 18116 (SAD_test.v:34-48).  Exercises SumAbsoluteDifferences (sad.go:205-244)."""
    assert [int(v) for v in hwfx["exp_sad_old"]] == [0x462A, 0x5368, 0x4ADF, 0x485A]
    for k in range(4):
        L, R = np.ascontiguousarray(hwfx["strips_old_L"][k]), np.ascontiguousarray(hwfx["strips_old_R"][k])
        assert oracle.sum_absolute_differences(L, R, 7, 7, 7, 7, 15) == int(hwfx["exp_sad_old"][k])
    wl = np.ascontiguousarray(hwfx["winL"].reshape(15, 15)); wr = np.ascontiguousarray(hwfx["winR"].reshape(15, 15))
    assert oracle.sum_absolute_differences(wl, wr, 7, 7, 7, 7, 15) == 18116
    assert int(np.abs(wl.astype(int) - wr.astype(int)).sum()) == 18116


def test_survey_digests_real_pair(testdata_pair):
    """SURVEY.md section 8c digests (from the survey's own transliteration of sad.go)."""
    L, R = testdata_pair
    assert hashlib.sha256(L.tobytes()).hexdigest() == "cf6b08f963bde1f74ca786b8ca787ac036bcf245cbe21e0e055cca225be8a527"
    assert hashlib.sha256(R.tobytes()).hexdigest() == "354836e34618a75d1c69ae178ac5729065056f972dcdab3eb41c1877c4b4ee89"
    d = oracle.disparity_literal(L, R, 7, 32)
    assert hashlib.sha256(d.tobytes()).hexdigest() == "03a658082e809b4211a1216839d23a7123f1187b56f18dff0b21b6863c8a9d6e"
    assert len(np.unique(d)) == 33 and int((d == 0).sum()) == 21287
    assert np.array_equal(d, oracle.disparity_closed(L, R, 7, 32))
    d2 = oracle.disparity_closed(L, R, 9, 64)
    assert hashlib.sha256(d2.tobytes()).hexdigest() == "0bde916590100d42d2fb6c0aca7081ba4f912c2b03a8f3bb5a8d9946454317fb"


@pytest.mark.parametrize("name", sorted(REAL_PAIRS))
def test_real_pairs_pinned_digests(name):
    """Every picture pair of the reference's testdata/ (four 640x480 RGBA pairs, two 1920x1080 RGB pairs): the gray
    planes and the COMPLETE disparity maps of the literal restatement at the parameters the GPU tests use equal the
    digests tests/golden/make_golden.py recorded (there: literal == closed form on each).  Real frames carry what the
    synthetic generator avoids -- exact ties, zero SADs, saturated windows (7 500 .. 130 000 zero pixels per map)."""
    e = REAL_PAIRS[name]
    L, R = load_real_gray(e["left"]), load_real_gray(e["right"])
    assert L.shape == R.shape == (e["h"], e["w"])
    assert hashlib.sha256(L.tobytes()).hexdigest() == e["gray_left_sha256"]
    assert hashlib.sha256(R.tobytes()).hexdigest() == e["gray_right_sha256"]
    for key, m in e["maps"].items():
        b, d = map(int, key.split("/"))
        got = oracle.disparity_rows_mt(L, R, b, d, 0, e["h"], os.cpu_count() or 1)
        assert hashlib.sha256(got.tobytes()).hexdigest() == m["sha256"], (name, key)
        assert int((got == 0).sum()) == m["zeros"] and len(np.unique(got)) == m["distinct"]
        hs = b // 2
        assert not got[:, :hs + 1].any()


def test_real_pairs_cover_the_reference_testdata():
    assert sorted(REAL_PAIRS) == ["L_00001.png", "L_00002.png", "L_00335.png", "L_01000.png", "im0-bs.png", "im0.png"]
    assert REAL_PAIRS["L_00001.png"]["maps"]["7/32"]["sha256"] == "03a658082e809b4211a1216839d23a7123f1187b56f18dff0b21b6863c8a9d6e"
    assert REAL_PAIRS["L_00001.png"]["gray_left_sha256"] == "cf6b08f963bde1f74ca786b8ca787ac036bcf245cbe21e0e055cca225be8a527"
    assert (REAL_PAIRS["im0.png"]["w"], REAL_PAIRS["im0.png"]["h"]) == (1920, 1080)


SHAPES = [  # (w, h, block, maxdisp)
    (64, 48, 7, 32), (40, 33, 9, 64), (33, 20, 16, 64), (50, 40, 31, 256), (1, 1, 7, 32),
    (5, 4, 15, 16), (97, 31, 3, 16), (70, 25, 5, 48), (128, 20, 15, 64), (30, 30, 1, 8), (45, 17, 2, 5),
]


@pytest.mark.parametrize("w,h,block,maxdisp", SHAPES)
def test_literal_equals_closed_form(w, h, block, maxdisp):
    L, R = oracle.synth_pair(100 + w + h, w, h)
    a = oracle.disparity_literal(L, R, block, maxdisp)
    b = oracle.disparity_closed(L, R, block, maxdisp)
    assert np.array_equal(a, b)
    assert np.array_equal(oracle.disparity_literal(L, R, block, maxdisp, raw=True),
                          oracle.disparity_closed(L, R, block, maxdisp, raw=True))
    hs = block // 2
    assert not a[:, :min(hs + 1, w)].any()        # columns x <= hs can only select d = 0


def test_flat_images_all_tie_to_zero():
    L = np.full((20, 50), 77, np.uint8)
    assert not oracle.disparity_literal(L, L.copy(), 9, 64).any()


def test_region_is_region_local_and_tiling_invariant():
    L, R = oracle.synth_pair(7, 80, 40)
    full = oracle.disparity_literal(L, R, 7, 32)
    reg = oracle.disparity_literal(L, R, 7, 32, region=(13, 5, 61, 29))
    assert reg.shape == (24, 48)
    assert np.array_equal(reg, full[5:29, 13:61])
    band = oracle.disparity_rows_mt(L, R, 7, 32, 8, 31, nthreads=3)
    assert np.array_equal(band, full[8:31])


def test_maxdisp_zero_is_an_error():
    L, R = oracle.synth_pair(1, 16, 8)
    with pytest.raises(ZeroDivisionError):
        oracle.disparity_literal(L, R, 7, 0)


def test_gray_formulas():
    rng = np.random.default_rng(0)
    rgba = rng.integers(0, 256, (9, 13, 4), dtype=np.uint8)
    g = oracle.gray_from_nrgba(rgba)
    a = rgba[..., 3].astype(np.uint64) * 0x101
    c16 = [(rgba[..., i].astype(np.uint64) * 0x101 * a) // 0xffff for i in range(3)]
    exp = ((19595 * c16[0] + 38470 * c16[1] + 7471 * c16[2] + (1 << 15)) >> 24).astype(np.uint8)
    assert np.array_equal(g, exp)
    rgb = rgba[..., :3].copy()
    opaque = np.concatenate([rgb, np.full((9, 13, 1), 255, np.uint8)], axis=2)
    assert np.array_equal(oracle.gray_from_rgb8(rgb), oracle.gray_from_nrgba(opaque))
    assert not oracle.gray_from_rgb8(rgb, refbug=True).any()      # gray.go:35-37 always yields 0


def test_synth_generator_c_equals_numpy():
    import steroscopic_hardware_amd as sha
    for seed, w, h in ((2, 64, 48), (3, 97, 13), (1000, 31, 7), (5, 200, 9)):
        Lc, Rc = oracle.synth_pair(seed, w, h)
        Ln, Rn = sha.synth.synth_pair(seed, w, h)
        assert np.array_equal(Lc, Ln) and np.array_equal(Rc, Rn)
    assert sha.synth.candidates_per_pair(1920, 1080, 9, 64) == 132261120
    assert sha.synth.candidates_per_pair(640, 480, 7, 32) == 9838080
    assert sha.synth.candidates_per_pair(3840, 2160, 15, 128) == 1050209280
