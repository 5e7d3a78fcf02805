#!/usr/bin/env python3
"""Regenerates tests/golden/*.  Run ONLY in the build container (needs /root/reference).

What it writes (all of it DATA: inputs and expected outputs, no reference source text):

  hardware_fixtures.npz
      The reference's own fixture files for this path, converted from
      one-hex-byte-per-line ``.mem`` text to uint8 arrays:
        patch_L[4,128,128], patch_R[4,128,128]   hardware/mems/img_{L,R}_patch_{0..3}.raw
        exp_disp_p[4,128,128]                    hardware/mems/exp_disp_p.mem   (test.py semantics)
        exp_disp_sadc[128,128]                   hardware/exp_disp.mem          (sad.c semantics, patch 1)
        left_image_p / right_image_p             hardware/mems/{left,right}_image_p.mem (== the .raw)
        strips_n_L/R[4,15,64], exp_disp_n[4]     hardware/mems/*_n.mem          (Verilog convention)
        strips_old_L/R[4,15,64], exp_disp_old[4], exp_sad_old[4]   hardware/old_mems/*
        winL[225], winR[225]                     hardware/old_mems/win{L,R}.mem (SAD_test.v:34-35)
        disparity_pgm                            hardware/disparity.pgm (raw bytes)
  reference_outputs.npz
      Outputs of the reference's OWN code run here:
        sadc_run_exp_disp[128,128]   oracle/_ref/sad_ref (= hardware/sad.c compiled unmodified)
                                     executed on mems/img_{L,R}_patch_1.raw
        sadc_run_pgm                 its disparity.pgm
        testpy_*                     hardware/test.py::compute_disparity_patch imported by path and
                                     run on the 4 patches and on small seeded synthetic inputs
  L_00001.png / R_00001.png
      testdata/L_00001.png, R_00001.png copied verbatim (BASELINE config 1 input pair).
  {L,R}_{00002,00335,01000}.png, im0.png, im1.png, im0-bs.png, im1-bs.png
      every other picture pair the reference holds in testdata/ (three more 640x480 RGBA pairs, two 1920x1080 RGB
      pairs), copied verbatim: real content -- exact ties, zero SADs, saturated regions -- for the GPU parity tests.
  real_pairs.json
      per pair and (BlockSize, MaxDisparity): sha256 of the gray planes (Go formula: NRGBA for the 640x480 pairs,
      color.GrayModel on opaque RGB for im0/im1 = what pkg/camera/output.go:145 yields) and of the complete
      disparity map of the LITERAL restatement (oracle.disparity_rows_mt), plus zero counts; the restatement and the
      closed form are asserted equal on each before anything is written.  tests/test_oracle_golden.py re-derives
      them on the CPU, tests/test_parity_gpu.py holds the HIP path to them.
"""
import hashlib
import importlib.util
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))


def read_mem(path, width=1):
    vals = [int(t, 16) for t in open(path).read().split()]
    return np.array(vals, dtype=np.uint8 if width == 1 else np.uint32)


REAL_640 = ("00001", "00002", "00335", "01000")                 # testdata/{L,R}_<k>.png, 640x480 RGBA
REAL_1080 = (("im0.png", "im1.png"), ("im0-bs.png", "im1-bs.png"))   # 1920x1080 RGB
PARAMS_640 = ((7, 32), (9, 64), (16, 64))
PARAMS_1080 = ((9, 64), (16, 64), (15, 128))


def load_gray(path):
    """What the reference's callers feed the path: Go decodes 8-bit RGBA files as *image.NRGBA and opaque RGB files
    as *image.RGBA; color.GrayModel.Convert (pkg/camera/output.go:145,160; pkg/despair/gray.go:43-58) on them."""
    from PIL import Image
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle
    im = Image.open(path)
    a = np.array(im)
    if im.mode == "RGBA":
        return oracle.gray_from_nrgba(a)
    assert im.mode == "RGB"
    return oracle.gray_from_rgb8(a)


def write_real_pairs():
    import json
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    import oracle
    thr = os.cpu_count() or 1
    rec = {}
    pairs = [(f"L_{k}.png", f"R_{k}.png", PARAMS_640) for k in REAL_640] + [(a, b, PARAMS_1080) for a, b in REAL_1080]
    for ln, rn, params in pairs:
        L, R = load_gray(os.path.join(HERE, ln)), load_gray(os.path.join(HERE, rn))
        e = {"left": ln, "right": rn, "w": int(L.shape[1]), "h": int(L.shape[0]),
             "gray_left_sha256": hashlib.sha256(L.tobytes()).hexdigest(),
             "gray_right_sha256": hashlib.sha256(R.tobytes()).hexdigest(), "maps": {}}
        for b, d in params:
            m = oracle.disparity_rows_mt(L, R, b, d, 0, L.shape[0], thr)
            assert np.array_equal(m, oracle.disparity_closed(L, R, b, d)), (ln, b, d)
            e["maps"][f"{b}/{d}"] = {"sha256": hashlib.sha256(m.tobytes()).hexdigest(), "zeros": int((m == 0).sum()),
                                     "distinct": int(len(np.unique(m)))}
            print(ln, b, d, e["maps"][f"{b}/{d}"])
        rec[ln] = e
    with open(os.path.join(HERE, "real_pairs.json"), "w") as f:
        json.dump(rec, f, indent=1, sort_keys=True)
        f.write("\n")


def main():
    hw = os.path.join(REF, "hardware")
    fx = {}
    fx["patch_L"] = np.stack([np.fromfile(f"{hw}/mems/img_L_patch_{k}.raw", np.uint8).reshape(128, 128) for k in range(4)])
    fx["patch_R"] = np.stack([np.fromfile(f"{hw}/mems/img_R_patch_{k}.raw", np.uint8).reshape(128, 128) for k in range(4)])
    fx["exp_disp_p"] = read_mem(f"{hw}/mems/exp_disp_p.mem").reshape(4, 128, 128)
    fx["exp_disp_sadc"] = read_mem(f"{hw}/exp_disp.mem").reshape(128, 128)
    fx["left_image_p"] = read_mem(f"{hw}/mems/left_image_p.mem").reshape(4, 128, 128)
    fx["right_image_p"] = read_mem(f"{hw}/mems/right_image_p.mem").reshape(4, 128, 128)
    fx["strips_n_L"] = read_mem(f"{hw}/mems/left_image_n.mem").reshape(4, 15, 64)
    fx["strips_n_R"] = read_mem(f"{hw}/mems/right_image_n.mem").reshape(4, 15, 64)
    fx["exp_disp_n"] = read_mem(f"{hw}/mems/exp_disp_n.mem")
    fx["strips_old_L"] = read_mem(f"{hw}/old_mems/left_image.mem").reshape(4, 15, 64)
    fx["strips_old_R"] = read_mem(f"{hw}/old_mems/right_image.mem").reshape(4, 15, 64)
    fx["exp_disp_old"] = read_mem(f"{hw}/old_mems/exp_disp.mem")
    fx["exp_sad_old"] = read_mem(f"{hw}/old_mems/exp_sad.mem", width=2)
    fx["winL"] = read_mem(f"{hw}/old_mems/winL.mem")
    fx["winR"] = read_mem(f"{hw}/old_mems/winR.mem")
    fx["disparity_pgm"] = np.fromfile(f"{hw}/disparity.pgm", np.uint8)
    np.savez_compressed(os.path.join(HERE, "hardware_fixtures.npz"), **fx)

    # --- the reference's own C, compiled unmodified (oracle/Makefile 'ref' target)
    subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle"), "ref"])
    ro = {}
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(f"{td}/mems")
        for s in "LR":
            shutil.copy(f"{hw}/mems/img_{s}_patch_1.raw", f"{td}/mems/")
        subprocess.check_call([os.path.join(REPO, "oracle/_ref/sad_ref")], cwd=td)
        ro["sadc_run_exp_disp"] = read_mem(f"{td}/exp_disp.mem").reshape(128, 128)
        ro["sadc_run_pgm"] = np.fromfile(f"{td}/disparity.pgm", np.uint8)
    assert np.array_equal(ro["sadc_run_exp_disp"], fx["exp_disp_sadc"]), "sad.c run != committed exp_disp.mem"
    assert np.array_equal(ro["sadc_run_pgm"], fx["disparity_pgm"]), "sad.c run != committed disparity.pgm"

    # --- the reference's own Python golden model, imported by path (main() is __name__-guarded)
    spec = importlib.util.spec_from_file_location("ref_hw_test", f"{hw}/test.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    tp = np.stack([mod.compute_disparity_patch(fx["patch_L"][k], fx["patch_R"][k], 64, 15) for k in range(4)])
    assert np.array_equal(tp, fx["exp_disp_p"]), "test.py run != committed exp_disp_p.mem"
    ro["testpy_patches"] = tp
    # small seeded synthetic cases through the same function: (h, w, win, maxd, seed)
    cases = [(40, 48, 7, 16, 11), (33, 64, 5, 32, 12), (24, 100, 9, 64, 13), (17, 23, 3, 16, 14), (50, 30, 15, 20, 15)]
    for i, (h, w, win, maxd, seed) in enumerate(cases):
        rng = np.random.default_rng(seed)
        L = rng.integers(0, 256, (h, w), dtype=np.uint8)
        shift = 2 + i
        R = np.roll(L, -shift, axis=1)
        R[:, w - shift:] = rng.integers(0, 256, (h, shift), dtype=np.uint8)
        noise = rng.integers(0, 16, (h, w)) == 0
        R = np.where(noise, R ^ 3, R).astype(np.uint8)
        ro[f"testpy_case{i}_L"] = L
        ro[f"testpy_case{i}_R"] = R
        ro[f"testpy_case{i}_params"] = np.array([win, maxd], dtype=np.int32)
        ro[f"testpy_case{i}_out"] = mod.compute_disparity_patch(L, R, maxd, win)
    ro["testpy_ncases"] = np.array([len(cases)], dtype=np.int32)
    np.savez_compressed(os.path.join(HERE, "reference_outputs.npz"), **ro)

    names = [f"{s}_{k}.png" for k in REAL_640 for s in "LR"] + [n for pair in REAL_1080 for n in pair]
    for n in names:
        shutil.copy(f"{REF}/testdata/{n}", os.path.join(HERE, n))
        os.chmod(os.path.join(HERE, n), 0o644)
        print(n, hashlib.sha256(open(os.path.join(HERE, n), "rb").read()).hexdigest())
    write_real_pairs()
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    sys.exit(main())
