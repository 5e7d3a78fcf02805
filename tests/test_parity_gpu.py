"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle, the committed
golden fixtures, and -- at BASELINE sizes -- size-independent properties.  Bit-exact (u8 / int).

Run on the MI355X box:  python -m pytest tests -m gpu -x -q
"""
import hashlib
import os

import numpy as np
import pytest

import oracle
import steroscopic_hardware_amd as sha
from steroscopic_hardware_amd import capi
from conftest import REAL_CASES, REAL_PAIRS, load_real_gray, load_real_pixels, to_device, to_host

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

BOTH = [0, capi.FLAG_FORCE_GENERIC]          # dispatcher's choice, and the generic kernel


@pytest.fixture(scope="module")
def ctx():
    assert os.path.exists(capi.LIB_PATH), "libsadhip.so must be built in-tree (no fallback)"
    c = capi.Context(0, 3840, 2160, 1)
    yield c
    c.close()


def _expect(L, R, block, maxdisp, raw=False):
    return oracle.disparity_closed(L, R, block, maxdisp, raw=raw)


# ---- config 2: 640x480 synthetic, block 7, D 32 (bit-exact vs despair) ----------------------
@pytest.mark.parametrize("flags", BOTH)
def test_config2_640x480_b7_d32(ctx, flags):
    L, R = sha.synth.synth_pair(2, 640, 480)
    exp = oracle.disparity_literal(L, R, 7, 32)          # the literal restatement of sad.go:55-94
    got = ctx.disparity(L, R, 7, 32, flags)
    assert np.array_equal(got, exp)


# ---- config 1: the reference's real pair through the Go gray formula --------------------------
@pytest.mark.parametrize("flags", BOTH)
def test_config1_real_pair(ctx, testdata_pair, flags):
    L, R = testdata_pair
    got = ctx.disparity(L, R, 7, 32, flags)
    assert hashlib.sha256(got.tobytes()).hexdigest() == "03a658082e809b4211a1216839d23a7123f1187b56f18dff0b21b6863c8a9d6e"
    assert np.array_equal(got, oracle.disparity_literal(L, R, 7, 32))
    got2 = ctx.disparity(L, R, 9, 64, flags)
    assert hashlib.sha256(got2.tobytes()).hexdigest() == "0bde916590100d42d2fb6c0aca7081ba4f912c2b03a8f3bb5a8d9946454317fb"


def test_real_pair_default_params_16_64(ctx, testdata_pair):
    """examples/run.go:30 calls RunSad(..., 16, 64): even BlockSize -> 17x17 window."""
    L, R = testdata_pair
    assert np.array_equal(sha.despair.RunSad(L, R, 16, 64), _expect(L, R, 16, 64))
    assert (sha.despair.DefaultParams().BlockSize, sha.despair.DefaultParams().MaxDisparity) == (16, 64)


# ---- every real picture pair the reference holds (testdata/), complete frames vs the literal restatement ----------
_REAL_GRAY = {}


def _real_gray(name):
    if name not in _REAL_GRAY:
        e = REAL_PAIRS[name]
        _REAL_GRAY[name] = (load_real_gray(e["left"]), load_real_gray(e["right"]))
    return _REAL_GRAY[name]


@pytest.mark.parametrize("name,key", REAL_CASES)
def test_real_inputs_complete_frames(ctx, name, key):
    """testdata/{L,R}_{00001,00002,00335,01000}.png (640x480 RGBA) at 7/32, 9/64 and the defaults 16/64; testdata/im0 + im1
    and im0-bs + im1-bs (1920x1080 RGB -- the headline config's size) at 9/64 (config 3's kernel), 16/64 and 15/128, through
    the gray formula pkg/camera/output.go:145 applies.  Real frames are where exact ties, zero SADs and saturated windows
    live (up to 130 000 zero-disparity pixels per map); the synthetic generator avoids them on purpose.  COMPLETE maps:
    (1) equal to the literal restatement of sad.go:55-94 run here, (2) equal to the digest pinned in the build
    container (tests/golden/real_pairs.json), (3) the generic kernel agrees, (4) 1080p: the same map as frames 0 and n-1 of
    a device-resident batch (the batched launch geometry: few tall strips), (5) the fused RGB(A) entry (device gray), (6) the
    same entry with the conversion in the copy threads (SADHIP_FLAG_HOST_GRAY)."""
    import torch
    e = REAL_PAIRS[name]
    b, d = map(int, key.split("/"))
    L, R = _real_gray(name)
    h, w = L.shape
    exp = oracle.disparity_rows_mt(L, R, b, d, 0, h, min(16, os.cpu_count() or 1))
    assert hashlib.sha256(exp.tobytes()).hexdigest() == e["maps"][key]["sha256"]
    got = ctx.disparity(L, R, b, d)
    assert capi.last_kernel_kind() == capi.KIND_COLWAVE
    assert np.array_equal(got, exp), (name, key, np.argwhere(got != exp)[:4])
    assert hashlib.sha256(got.tobytes()).hexdigest() == e["maps"][key]["sha256"]
    assert int((got == 0).sum()) == e["maps"][key]["zeros"]
    assert np.array_equal(ctx.disparity(L, R, b, d, capi.FLAG_FORCE_GENERIC), exp)
    fused = ctx.disparity_rgba(load_real_pixels(e["left"]), load_real_pixels(e["right"]), b, d)
    assert np.array_equal(fused, exp), (name, key, "fused gray + disparity")
    # (6) the same pictures with the gray conversion in the copy threads (SADHIP_FLAG_HOST_GRAY, ABI 5): real alpha 251..255
    hostg = ctx.disparity_rgba(load_real_pixels(e["left"]), load_real_pixels(e["right"]), b, d, capi.FLAG_HOST_GRAY)
    assert np.array_equal(hostg, exp), (name, key, "host gray + disparity")
    # device-resident batch: the real pair first and last, synthetic frames (or the other real pairs) between
    n = 12 if w == 1920 else 40
    mid = [sha.synth.synth_pair(4000 + i, w, h) for i in range(3)]
    Lb = np.stack([L] + [mid[i % 3][0] for i in range(n - 2)] + [L])
    Rb = np.stack([R] + [mid[i % 3][1] for i in range(n - 2)] + [R])
    Ls, Rs = to_device(Lb), to_device(Rb)
    out = torch.full_like(Ls, 0xCD)
    ctx.disparity_dev(Ls, Rs, out, b, d)
    torch.cuda.synchronize()
    assert capi.last_kernel_kind() == capi.KIND_COLWAVE
    o = to_host(out)
    assert np.array_equal(o[0], exp) and np.array_equal(o[n - 1], exp), (name, key, "batched layout")
    if w == 640:
        assert np.array_equal(o[1], _expect(mid[0][0], mid[0][1], b, d))


def test_real_1080p_pair_raw_ties_and_first_minimum(ctx):
    """The first-minimum rule (strict <, sad.go:79) on real 1080p content: wherever several candidates tie for the minimum
    the LOWEST d must come out.  im0/im1 at 9/64, raw d: every pixel against the closed form's raw output on three bands,
    and the count of pixels whose minimum is attained more than once (from the oracle's SAD table of a band) is > 0 --
    i.e. the test really exercises ties."""
    L, R = _real_gray("im0.png")
    raw = ctx.disparity(L, R, 9, 64, capi.FLAG_RAW)
    for y0 in (0, 500, 1068):
        assert np.array_equal(raw[y0:y0 + 12], oracle.disparity_closed(L, R, 9, 64, region=(0, y0, 1920, y0 + 12), raw=True))
    # ties on one band, counted with plain numpy from full windows (interior columns only)
    y, hs, D = 540, 4, 64
    Lf, Rf = L.astype(np.int32), R.astype(np.int32)
    xs = np.arange(D + hs, 1920 - hs)
    sads = np.empty((D + 1, len(xs)), np.int64)
    for dd in range(D + 1):
        ad = np.abs(Lf[y - hs:y + hs + 1, D:] - Rf[y - hs:y + hs + 1, D - dd:1920 - dd])       # columns D.. of L vs shifted R
        cs = np.concatenate([np.zeros(1, np.int64), ad.sum(0).cumsum()])
        sads[dd] = cs[xs - D + hs + 1] - cs[xs - D - hs]
    best = sads.min(0)
    ties = (sads == best[None]).sum(0) > 1
    assert ties.sum() > 0, "no tied minima on this band: pick another"
    assert np.array_equal(raw[y, xs], sads.argmin(0))                 # argmin = first minimum = lowest d
    assert (raw[y, xs][ties] == sads.argmin(0)[ties]).all()


# ---- hardware fixtures through the GPU (raw d) -------------------------------------------------
@pytest.mark.parametrize("flags", BOTH)
def test_hardware_fixture_regions(ctx, hwfx, flags):
    hs, D = 7, 64
    for k in range(4):
        got = ctx.disparity(hwfx["patch_L"][k], hwfx["patch_R"][k], 15, 64, flags | capi.FLAG_RAW)
        assert np.array_equal(got[:, D + hs:128 - hs], hwfx["exp_disp_p"][k][:, D + hs:128 - hs])
    got = ctx.disparity(hwfx["patch_L"][1], hwfx["patch_R"][1], 15, 64, flags | capi.FLAG_RAW)
    assert np.array_equal(got[hs:128 - hs, D + hs:128 - hs], hwfx["exp_disp_sadc"][hs:128 - hs, D + hs:128 - hs])


def test_hardware_mirror_modes_regenerate_the_reference_fixtures(ctx, hwfx, refout):
    """f-4: with the semantics of hardware/sad.c and hardware/test.py the GPU reproduces the
    reference's committed golden files in full (every pixel, not only the region shared with Go)."""
    for k in range(4):
        got = ctx.disparity(hwfx["patch_L"][k], hwfx["patch_R"][k], 15, 64, capi.FLAG_MIRROR_TESTPY)
        assert np.array_equal(got, hwfx["exp_disp_p"][k])                    # hardware/mems/exp_disp_p.mem
    got = ctx.disparity(hwfx["patch_L"][1], hwfx["patch_R"][1], 15, 64, capi.FLAG_MIRROR_SADC)
    assert np.array_equal(got, hwfx["exp_disp_sadc"])                        # hardware/exp_disp.mem
    for i in range(int(refout["testpy_ncases"][0])):                         # test.py run on seeded inputs
        win, maxd = (int(v) for v in refout[f"testpy_case{i}_params"])
        got = ctx.disparity(refout[f"testpy_case{i}_L"], refout[f"testpy_case{i}_R"], win, maxd, capi.FLAG_MIRROR_TESTPY)
        assert np.array_equal(got, refout[f"testpy_case{i}_out"])
    L, R = sha.synth.synth_pair(31, 200, 90)                                 # and at another size vs the CPU mirrors
    assert np.array_equal(ctx.disparity(L, R, 9, 48, capi.FLAG_MIRROR_SADC), oracle.mirror_sadc(L, R, 9, 48))
    assert np.array_equal(ctx.disparity(L, R, 9, 48, capi.FLAG_MIRROR_TESTPY), oracle.mirror_testpy(L, R, 9, 48))


def test_verilog_mirror_mode(ctx, hwfx):
    """a-10: hardware/compute_SAD.v's convention on the GPU (SADHIP_FLAG_MIRROR_VERILOG): the right
    window slides to LARGER columns, d < MAX_DISP exclusive.  The reference's own fixtures
    mems/exp_disp_n.mem = 12 12 15 0F and old_mems/exp_disp.mem = 23 0E 29 31 come out at the strip's
    centre pixel; other sizes against the CPU mirror."""
    for L, R, exp in ((hwfx["strips_n_L"], hwfx["strips_n_R"], hwfx["exp_disp_n"]),
                      (hwfx["strips_old_L"], hwfx["strips_old_R"], hwfx["exp_disp_old"])):
        got = [int(ctx.disparity(np.ascontiguousarray(L[k]), np.ascontiguousarray(R[k]), 15, 64, capi.FLAG_MIRROR_VERILOG)[7, 7]) for k in range(4)]
        assert got == [int(v) for v in exp]
        for k in range(4):
            assert np.array_equal(ctx.disparity(np.ascontiguousarray(L[k]), np.ascontiguousarray(R[k]), 15, 64, capi.FLAG_MIRROR_VERILOG),
                                  oracle.mirror_verilog(L[k], R[k], 15, 64))
    assert capi.last_kernel_kind() == capi.KIND_MIRROR_VERILOG
    assert [int(v) for v in hwfx["exp_disp_n"]] == [0x12, 0x12, 0x15, 0x0F] and [int(v) for v in hwfx["exp_disp_old"]] == [0x23, 0x0E, 0x29, 0x31]
    for seed, w, h, win, nd in ((1, 200, 40, 15, 64), (2, 90, 33, 9, 16), (3, 64, 15, 15, 64), (4, 31, 31, 31, 8), (5, 300, 20, 7, 200)):
        L, R = sha.synth.synth_pair(seed, w, h)
        assert np.array_equal(ctx.disparity(L, R, win, nd, capi.FLAG_MIRROR_VERILOG), oracle.mirror_verilog(L, R, win, nd)), (w, h, win, nd)
    L, R = sha.synth.synth_pair(6, 120, 50)                      # region + batch entry points take the flag too
    reg = (10, 5, 100, 40)
    assert np.array_equal(ctx.disparity_region(L, R, 9, 32, reg, capi.FLAG_MIRROR_VERILOG), oracle.mirror_verilog(L, R, 9, 32)[5:40, 10:100])


def test_flags_rejected_and_kernel_kind_is_what_ran(ctx):
    """Unknown or conflicting flags are SADHIP_ERR_ARG; sadhip_last_kernel_kind reports the kernel the
    last call really launched and always equals the plan (no silent fallback)."""
    L, R = sha.synth.synth_pair(8, 300, 40)
    for fl in (32, 1 << 20, capi.FLAG_MIRROR_SADC | capi.FLAG_MIRROR_TESTPY, capi.FLAG_MIRROR_VERILOG | capi.FLAG_MIRROR_SADC):
        with pytest.raises(capi.SadHipError) as ei:
            ctx.disparity(L, R, 9, 64, fl)
        assert ei.value.status == capi.ERR_ARG
        with pytest.raises(capi.SadHipError) as ei:
            ctx.disparity_region(L, R, 9, 64, (0, 0, 10, 10), fl)
        assert ei.value.status == capi.ERR_ARG
    for (w, h, b, d, fl) in ((300, 40, 9, 64, 0), (300, 40, 9, 64, capi.FLAG_FORCE_GENERIC), (300, 40, 33, 64, 0), (300, 40, 16, 64, 0),
                             (300, 40, 31, 256, 0), (300, 40, 9, 300, 0), (100, 30, 3, 96, 0), (300, 40, 15, 64, capi.FLAG_MIRROR_SADC),
                             (300, 40, 15, 64, capi.FLAG_MIRROR_TESTPY)):
        A, B = sha.synth.synth_pair(9, w, h)
        ctx.disparity(A, B, b, d, fl)
        assert capi.last_kernel_kind() == capi.kernel_kind(w, h, b, d, fl), (w, h, b, d, fl)
    for block in range(2, 32):                                   # every plan launch_pass can be asked for really launches
        for d in (16, 64, 65, 128, 256):
            w = d + 2 * (block // 2) + 60
            A, B = sha.synth.synth_pair(block * 1000 + d, w, 2 * (block // 2) + 3)
            if capi.kernel_kind(w, A.shape[0], block, d) == capi.KIND_COLWAVE:
                ctx.disparity(A, B, block, d)
                assert capi.last_kernel_kind() == capi.KIND_COLWAVE, (block, d)


def test_device_pool_round_robin(ctx):
    """SURVEY 8e / VERDICT r1: sadhip_pool_* -- one context + host thread per listed device, frame i
    on entry i mod G.  Two contexts on device 0 stand in for two GPUs on the one-GPU box."""
    pairs = [sha.synth.synth_pair(1000 + i, 320, 96) for i in range(7)]
    with capi.Pool([0, 0], 320, 96) as pool:
        assert pool.size() == 2
        outs = pool.batch([p[0] for p in pairs], [p[1] for p in pairs], 9, 64)
        for (L, R), o in zip(pairs, outs):
            assert np.array_equal(o, _expect(L, R, 9, 64))
        assert pool.batch([], [], 9, 64) == []
        one = pool.batch([pairs[0][0]], [pairs[0][1]], 16, 64)   # fewer frames than devices
        assert np.array_equal(one[0], _expect(*pairs[0], 16, 64))
        with pytest.raises(capi.SadHipError) as ei:
            pool.batch([pairs[0][0]], [pairs[0][1]], 9, 0)
        assert ei.value.status == capi.ERR_MAXDISP
        big = np.zeros((200, 400), np.uint8)
        with pytest.raises(capi.SadHipError) as ei:                # a worker's failure reaches the caller with its detail
            pool.batch([big, big, big], [big, big, big], 9, 64)
        assert ei.value.status == capi.ERR_DIMS and "device entry" in str(ei.value)
        k_ms, wall_ms = pool.last_timing()
        assert wall_ms > 0
    with pytest.raises(capi.SadHipError) as ei:
        capi.Pool([0, 99], 64, 64)
    assert ei.value.status == capi.ERR_NO_DEVICE


def test_context_growth_under_concurrent_callers(ctx):
    """The Python mirror's process-wide context grows when a larger frame arrives; a second thread
    doing so while the first is inside the library must not destroy the context under it
    (VERDICT r1 weak #6).  Two threads alternate small and large frames."""
    import threading
    d = sha.despair
    small = sha.synth.synth_pair(71, 300, 80)
    errs = []

    def worker(k):
        try:
            for i in range(4):
                w, h = (1930 + 40 * (i + k), 1090 + 8 * i) if (i + k) % 2 else (300, 80)      # grows past 1920x1080 every other call
                if (w, h) == (300, 80):
                    L, R = small
                else:
                    L, R = sha.synth.synth_pair(72 + i, w, h)
                got = d.RunSad(L, R, 9, 64)
                if not np.array_equal(got[:6], oracle.disparity_closed(L, R, 9, 64, region=(0, 0, w, 6))):
                    errs.append(("mismatch", k, i))
        except Exception as e:                                     # noqa: BLE001
            errs.append(("exception", k, repr(e)))

    ts = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    d.SetDefaultParams(d.Parameters(BlockSize=16, MaxDisparity=64))
    assert not errs, errs


def test_pipeline_error_reaches_the_consumer(ctx):
    """A device error inside the SetupConcurrentSAD worker: AssembleDisparityMap raises it (no short,
    partly black map) and the feeder is not left blocked."""
    import threading
    d = sha.despair
    L, R = sha.synth.synth_pair(21, 200, 300)
    d.SetDefaultParams(d.Parameters(BlockSize=7, MaxDisparity=0))         # the reference divides by zero here; the ABI refuses
    inp, outp = d.SetupConcurrentSAD(2)
    feeder = threading.Thread(target=lambda: [inp.send(d.InputChunk(L, R, (0, y, 200, y + 1))) for y in range(300)])
    feeder.start()
    with pytest.raises(capi.SadHipError) as ei:
        d.AssembleDisparityMap(outp, (0, 0, 200, 300), 300)
    assert ei.value.status == capi.ERR_MAXDISP
    feeder.join(timeout=60)
    assert not feeder.is_alive()
    inp.close()
    d.SetDefaultParams(d.Parameters(BlockSize=16, MaxDisparity=64))


# ---- adversarial / edge shapes ------------------------
EDGE = [  # (w, h, block, maxdisp)
    (1, 1, 7, 32), (5, 4, 15, 16), (40, 33, 9, 64), (33, 20, 16, 64), (50, 40, 31, 256),
    (97, 31, 3, 16), (70, 25, 5, 48), (128, 20, 15, 64), (30, 30, 1, 8), (45, 17, 2, 5),
    (257, 37, 9, 64), (300, 70, 7, 32), (513, 9, 11, 96), (64, 200, 13, 16), (1000, 12, 9, 128),
    (130, 66, 9, 1), (90, 50, 21, 40),
]


@pytest.mark.parametrize("flags", BOTH)
@pytest.mark.parametrize("w,h,block,maxdisp", EDGE)
def test_edge_shapes(ctx, w, h, block, maxdisp, flags):
    L, R = sha.synth.synth_pair(100 + w + h, w, h)
    assert np.array_equal(ctx.disparity(L, R, block, maxdisp, flags), _expect(L, R, block, maxdisp))
    assert np.array_equal(ctx.disparity(L, R, block, maxdisp, flags | capi.FLAG_RAW),
                          np.minimum(_expect(L, R, block, maxdisp, raw=True), 255))


def test_maximum_parameters():
    """The largest parameters the ABI accepts (SADHIP_MAX_BLOCK 255, SADHIP_MAX_MAXDISP 4096) and one past them:
    a 32-pass column-wave job (4097 candidates, keys travelling from pass to pass), the same range on a frame
    narrower than it, and a window larger than the image -- against the generic kernel and, on a few rows, the oracle."""
    rng = np.random.default_rng(4096)
    W, H, D = 4300, 14, capi.MAX_MAXDISP
    ctx = capi.Context(0, W, 64, 1)                                                              # wider than the shared fixture's frames
    L = rng.integers(0, 256, (H, W), dtype=np.uint8)
    R = np.roll(L, -1500, axis=1) ^ (rng.integers(0, 32, (H, W), dtype=np.uint8) == 0).astype(np.uint8)   # true disparity 1500
    got = ctx.disparity(L, R, 9, D)
    assert capi.last_kernel_kind() == capi.kernel_kind(W, H, 9, D) == capi.KIND_COLWAVE and len(capi.plan_describe(W, H, 9, D)) >= 32
    assert np.array_equal(got, ctx.disparity(L, R, 9, D, capi.FLAG_FORCE_GENERIC))
    assert np.array_equal(got[6:8], oracle.disparity_closed(L, R, 9, D, region=(0, 6, W, 8)))
    assert (got[:, 3000:4000] == 1500 * 255 // D).mean() > 0.9                                  # and it finds the shift
    narrow_L, narrow_R = np.ascontiguousarray(L[:, :700]), np.ascontiguousarray(R[:, :700])    # W < D: every pixel has x - h + 1 candidates
    for block in (3, 9):
        g = ctx.disparity(narrow_L, narrow_R, block, D)
        assert np.array_equal(g, ctx.disparity(narrow_L, narrow_R, block, D, capi.FLAG_FORCE_GENERIC)), block
        assert np.array_equal(g[3:5], oracle.disparity_literal(narrow_L, narrow_R, block, D, region=(0, 3, 700, 5))), block
    small_L, small_R = sha.synth.synth_pair(255, 96, 40)
    g = ctx.disparity(small_L, small_R, capi.MAX_BLOCK, 64)                                     # window 255 x 255 > the image
    assert capi.last_kernel_kind() == capi.KIND_GENERIC_SLIDING                                 # 255^2 * 65 per pixel: the running-sum form
    assert np.array_equal(g, oracle.disparity_literal(small_L, small_R, capi.MAX_BLOCK, 64))
    for block, maxdisp, status in ((capi.MAX_BLOCK + 1, 64, capi.ERR_BLOCK), (9, capi.MAX_MAXDISP + 1, capi.ERR_MAXDISP)):
        with pytest.raises(capi.SadHipError) as e:
            ctx.disparity(small_L, small_R, block, maxdisp)
        assert e.value.status == status
    ctx.close()


def test_generic_path_is_bounded_running_sum_kernel(ctx):
    """the one-thread-per-pixel generic kernel is O(BlockSize^2 * D) per pixel and the ABI accepts
    BlockSize up to 255 at any frame size.  Above SADHIP_GENERIC_MAX_WORK byte differences per pixel the generic path runs
    its running-sum form (kernels_sliding.hip, kind 5): same function, O(D) per pixel.
    (1) small frames, every kind of parameter that reaches it -- large odd / even blocks, windows larger than the frame,
        D > w, regions, raw / scaled, a right image smaller than the left one -- against the literal oracle;
    (2) against the naive generic kernel just below / above the bound (the two forms agree where both can run);
    (3) 1920x1080, BlockSize 63, MaxDisparity 64: three oracle bands (literal restatement) and a wall-time bound -- the
        naive kernel needs ~0.2 s for this frame, the call must finish in well under that."""
    import time
    rng = np.random.default_rng(63)
    for (w, h, block, D) in ((96, 40, 255, 64), (200, 70, 63, 64), (333, 45, 33, 64), (150, 90, 101, 16), (64, 64, 32, 100),
                             (40, 30, 47, 300), (700, 24, 35, 128), (130, 130, 129, 9), (517, 33, 31, 300), (5, 4, 255, 4096)):
        L = rng.integers(0, 256, (h, w), dtype=np.uint8)
        R = np.roll(L, -min(D, 11), axis=1) ^ (rng.integers(0, 8, (h, w), dtype=np.uint8) == 0).astype(np.uint8)
        assert capi.kernel_kind(w, h, block, D, capi.FLAG_FORCE_GENERIC) == capi.KIND_GENERIC_SLIDING, (w, h, block, D)
        raw = D <= 255
        got = ctx.disparity(L, R, block, D, capi.FLAG_FORCE_GENERIC | (capi.FLAG_RAW if raw else 0))
        assert capi.last_kernel_kind() == capi.KIND_GENERIC_SLIDING
        assert np.array_equal(got, oracle.disparity_literal(L, R, block, D, raw=raw)), (w, h, block, D)
        if w > 8 and h > 8:
            reg = (w // 5, 2, w - 3, h - 1)
            assert np.array_equal(ctx.disparity_region(L, R, block, D, reg, capi.FLAG_FORCE_GENERIC),
                                  oracle.disparity_literal(L, R, block, D, region=reg)), (w, h, block, D, "region")
            Rs = np.ascontiguousarray(R[: h - 7, : w - 9])                        # smaller right image: dropped terms
            assert np.array_equal(ctx.disparity_lr(L, Rs, block, D), oracle.disparity_literal_lr(L, Rs, block, D)), (w, h, block, D, "lr")
    flat = np.full((50, 120), 9, np.uint8)
    assert not ctx.disparity(flat, flat.copy(), 63, 64).any()                     # every candidate ties: d = 0
    # the two generic forms agree either side of the bound: 31^2 * 68 = 65 348 (naive), 31^2 * 69 = 66 309 (running sums)
    L, R = sha.synth.synth_pair(31, 400, 60)
    a = ctx.disparity(L, R, 31, 67, capi.FLAG_FORCE_GENERIC | capi.FLAG_RAW); ka = capi.last_kernel_kind()
    b = ctx.disparity(L, R, 31, 68, capi.FLAG_FORCE_GENERIC | capi.FLAG_RAW); kb = capi.last_kernel_kind()
    assert (ka, kb) == (capi.KIND_GENERIC, capi.KIND_GENERIC_SLIDING)
    assert np.array_equal(a, oracle.disparity_closed(L, R, 31, 67, raw=True)) and np.array_equal(b, oracle.disparity_closed(L, R, 31, 68, raw=True))
    # a device-resident batch through the same form
    import torch
    Ls, Rs = sha.synth.synth_batch_torch([70, 71, 72], 600, 80, "cuda:0")
    out = torch.full_like(Ls, 0xCD)
    ctx.disparity_dev(Ls, Rs, out, 41, 64)
    torch.cuda.synchronize()
    assert capi.last_kernel_kind() == capi.KIND_GENERIC_SLIDING
    for i in range(3):
        Li, Ri = sha.synth.synth_pair(70 + i, 600, 80)
        assert np.array_equal(to_host(out[i]), oracle.disparity_closed(Li, Ri, 41, 64))
    # (3) the full-size case with a wall-time bound
    L, R = sha.synth.synth_pair(63, 1920, 1080)
    ctx.disparity(L, R, 63, 64)                                                   # warm-up (module load, scratch)
    t0 = time.perf_counter()
    got = ctx.disparity(L, R, 63, 64)
    wall = time.perf_counter() - t0
    k_ms, _ = ctx.last_timing()
    assert capi.last_kernel_kind() == capi.KIND_GENERIC_SLIDING
    for y0 in (0, 537, 1076):
        assert np.array_equal(got[y0:y0 + 4], oracle.disparity_rows_mt(L, R, 63, 64, y0, y0 + 4, min(16, os.cpu_count() or 1))), y0
    assert wall < 0.1 and k_ms < 60, (wall, k_ms)                                 # naive: 63^2 * 65 * 2.07e6 = 5.3e11 byte differences, ~0.2 s


def test_random_parameter_sweep(ctx):
    """60 seeded random (w, h, BlockSize, MaxDisparity) draws over the ranges the WebUI allows and
    beyond (cmd/handlers/params.go:37-52: block 3..31 odd, maxDisp 16..256 step 16), every kernel
    variant the dispatcher can pick, against the oracle."""
    rng = np.random.default_rng(2024)
    kinds = set()
    for i in range(60):
        w = int(rng.integers(1, 700)); h = int(rng.integers(1, 60))
        block = int(rng.integers(1, 33)); maxdisp = int(rng.choice([1, 5, 16, 32, 48, 64, 80, 96, 128, 144, 192, 256]))
        L, R = sha.synth.synth_pair(5000 + i, w, h)
        if i % 7 == 0:
            R = L.copy()                                      # exact matches: SAD 0, ties everywhere
        kinds.add((capi.kernel_kind(w, h, block, maxdisp), block // 2 > 4))
        got = ctx.disparity(L, R, block, maxdisp, capi.FLAG_RAW)
        assert np.array_equal(got, np.minimum(_expect(L, R, block, maxdisp, raw=True), 255)), (w, h, block, maxdisp)
    assert (1, False) in kinds and (1, True) in kinds and (0, True) in kinds    # P=4, P=8 and generic all exercised


def test_left_edge_candidate_ranges(ctx):
    """sad.go:66-69: pixel x only has candidates d <= min(D, max(0, x-h)).  The fast kernel removes
    the others with penalised column sums (or per-key masks for narrow frames / h = 6, 7): inputs
    on which a missing candidate would win, or tie at the largest possible SAD, around every
    switch-over (w vs D+2h+1, h vs 5, packed vs unpacked)."""
    rng = np.random.default_rng(99)
    for block in (3, 7, 9, 11, 13, 15, 16):
        hh = block // 2
        for D in (16, 64):
            for w in (D + 2 * hh + 1, D + 2 * hh + 2, D + 2 * hh + 3, D + 2 * hh + 40, 300):
                hgt = 2 * hh + 5
                base = rng.integers(0, 256, size=(hgt, w + D), dtype=np.uint8)
                cases = {
                    "planted": (np.ascontiguousarray(base[:, :w]), np.ascontiguousarray(base[:, D:D + w])),   # R[x-D] = L[x]: true shift D
                    "planted-rev": (np.ascontiguousarray(base[:, D:D + w]), np.ascontiguousarray(base[:, :w])),
                    "extreme": (np.full((hgt, w), 255, np.uint8), np.zeros((hgt, w), np.uint8)),             # every SAD is the maximum
                    "zero-left": (np.zeros((hgt, w), np.uint8), np.ascontiguousarray(base[:, :w])),
                }
                Lx = np.full((hgt, w), 255, np.uint8); Rx = np.zeros((hgt, w), np.uint8)
                Rx[:, : w // 2] = 255                          # d = 0 is the worst candidate on the right half
                cases["half"] = (Lx, Rx)
                for name, (L, R) in cases.items():
                    exp = np.minimum(_expect(L, R, block, D, raw=True), 255)
                    got = ctx.disparity(L, R, block, D, capi.FLAG_RAW)
                    assert np.array_equal(got, exp), (name, block, D, w)
    L, R = sha.synth.synth_pair(41, 400, 40)
    for reg in ((0, 0, 20, 40), (3, 5, 90, 30), (60, 0, 200, 40)):             # regions starting inside the edge zone
        assert np.array_equal(ctx.disparity_region(L, R, 9, 64, reg), oracle.disparity_closed(L, R, 9, 64, region=reg))


def test_every_plan_shape(ctx):
    """Every way plan_for() can cover D+1 candidates: exact plans, the 16k+1-th candidate on the last
    wave, chunks that run past MaxDisparity (candidates that exist for no pixel), packed h <= 5
    (penalised sums), packed h = 6, 7 (lane masks; unpacked when no exact plan exists), unpacked
    h = 8 -- on random frames and on frames whose SADs are the largest possible (u16 headroom)."""
    rng = np.random.default_rng(7)
    w = 420
    for block in (3, 5, 7, 9, 11, 13, 15, 16, 17):
        hh = block // 2
        hgt = 2 * hh + 6
        Lr = rng.integers(0, 256, size=(hgt, w), dtype=np.uint8)
        Rr = rng.integers(0, 256, size=(hgt, w), dtype=np.uint8)
        Lx = np.full((hgt, w), 255, np.uint8); Rx = np.zeros((hgt, w), np.uint8)
        Rx[:, w // 3:] = 255; Rx[:, 2 * w // 3:] = 0
        for D in (16, 17, 18, 19, 20, 33, 47, 64, 65, 70, 100, 128, 130, 192, 255, 256):
            if D <= 128 or block <= 9:
                assert capi.kernel_kind(w, hgt, block, D) == 1, (block, D)     # the column-wave kernel, not the fallback
            for name, (L, R) in (("random", (Lr, Rr)), ("extreme", (Lx, Rx))):
                raw = D <= 255                             # raw d is a u8: the scaled output covers D = 256
                exp = _expect(L, R, block, D, raw=raw)
                got = ctx.disparity(L, R, block, D, capi.FLAG_RAW if raw else 0)
                assert np.array_equal(got, exp), (name, block, D)


def test_u16_ring_chunks_of_four_and_mixed_chunk_plans(ctx):
    """Round 4: the unpacked kernels (BlockSize >= 16) for 16k+1 candidates with k odd.  MaxDisparity 16 plans as 4 waves x 4
    candidates + 1 on the u16 ring (8-byte operand reads); 48 / 80 / 112 as MIXED workgroups -- 4 / 8 / 12 waves of 8
    candidates followed by 4 waves of 4, + 1 -- and 176 / 208 / 240 with such a pass after a full pass of 128.  Random,
    shifted, extreme (largest SADs) and flat frames, widths of one to three tiles and every w mod 8: every frame against
    the generic path on the device (an independent implementation, itself held to the oracle by its own tests) and rows
    of every frame against the oracle itself."""
    import torch
    rng = np.random.default_rng(44)
    for block, w in ((16, 437), (17, 700), (19, 1030), (23, 517), (27, 941), (31, 1203)):
        hh = block // 2
        hgt = 2 * hh + 9
        for D in (16, 48, 80, 112, 176, 208, 240):
            pl = capi.plan_describe(w, hgt, block, D)
            assert pl and pl[-1]["w16"] == 1 and pl[-1]["extra"] == 1, (block, D, pl)
            assert pl[-1]["nwh"] == (0 if D == 16 else 4) and (pl[-1]["nd"], pl[-1]["nw"]) == ((4, 4) if D == 16 else (8, {48: 8, 80: 12, 112: 16}[D % 128])), (block, D, pl)
            L = rng.integers(0, 256, size=(4, hgt, w), dtype=np.uint8)
            R = rng.integers(0, 256, size=(4, hgt, w), dtype=np.uint8)
            R[1] = np.roll(L[1], -min(D, 29), axis=1)
            L[2] = 255; R[2] = 0; R[2][:, w // 3:] = 255; R[2][:, 2 * w // 3:] = 0
            L[3] = 91; R[3] = 91
            Ls, Rs = to_device(L), to_device(R)
            a, b = torch.full_like(Ls, 0xCD), torch.full_like(Ls, 0xEE)
            ctx.disparity_dev(Ls, Rs, a, block, D, capi.FLAG_RAW)
            assert capi.last_kernel_kind() == capi.KIND_COLWAVE
            ctx.disparity_dev(Ls, Rs, b, block, D, capi.FLAG_RAW | capi.FLAG_FORCE_GENERIC)
            torch.cuda.synchronize()
            got = to_host(a)
            assert np.array_equal(got, to_host(b)), (block, w, D)
            for i in range(4):                                   # two rows of every frame against the oracle, through the host path
                one = ctx.disparity_region(L[i], R[i], block, D, (0, hh + 3, w, hh + 5), capi.FLAG_RAW)
                assert np.array_equal(one, got[i][hh + 3:hh + 5]), (block, w, D, i)
                assert np.array_equal(one, oracle.disparity_closed(L[i], R[i], block, D, region=(0, hh + 3, w, hh + 5), raw=True)), (block, w, D, i)
    # and as a device-resident batch at 1080p width (several tiles, the batched strip geometry), against the generic kernel
    for block, D in ((16, 48), (21, 80), (31, 112), (16, 16)):
        g = torch.Generator(device="cuda").manual_seed(block * 1000 + D)
        Ls = torch.randint(0, 256, (6, 70, 1920), dtype=torch.uint8, device="cuda", generator=g)
        Rs = torch.roll(Ls, -17, dims=2) ^ (torch.randint(0, 16, Ls.shape, device="cuda", generator=g) == 0).to(torch.uint8)
        a, b = torch.empty_like(Ls), torch.empty_like(Ls)
        ctx.disparity_dev(Ls, Rs, a, block, D)
        ctx.disparity_dev(Ls, Rs, b, block, D, capi.FLAG_FORCE_GENERIC)
        torch.cuda.synchronize()
        assert torch.equal(a, b), (block, D)


def test_unpacked_kernels_any_max_disparity(ctx):
    """Round 4: inside a pass the unpacked kernels' keys hold d - dbase (at most 8 bits), the key that travels between passes
    holds d: the u16 ring and the 2^27 penalty no longer limit MaxDisparity.  BlockSize 16 .. 31 with D = 256 (two passes of 16
    waves on the u16 ring where round 3 ran four of 64 candidates), 300, 511, 512, 1000 and 4096 (round 3: BlockSize 31 from D 512
    on was the generic kernel's) -- random, shifted and extreme frames against the oracle, and the generic path on the device."""
    import torch
    rng = np.random.default_rng(46)
    for block, w, D in ((16, 700, 256), (17, 600, 300), (21, 900, 511), (31, 800, 512), (27, 1300, 1000), (31, 300, 256), (19, 4400, 4096), (16, 517, 257)):
        hh = block // 2
        hgt = 2 * hh + 5
        pl = capi.plan_describe(w, hgt, block, D)
        assert capi.kernel_kind(w, hgt, block, D) == capi.KIND_COLWAVE and len(pl) >= 2 and not pl[0]["packed"], (block, D, pl)
        if block >= 18:
            assert all(p["w16"] for p in pl), (block, D, pl)
        L = rng.integers(0, 256, size=(3, hgt, w), dtype=np.uint8)
        R = rng.integers(0, 256, size=(3, hgt, w), dtype=np.uint8)
        R[1] = np.roll(L[1], -min(D - 3, w // 2), axis=1)               # a true shift deep in the range (the later passes win)
        L[2] = 255; R[2] = 0; R[2][:, w // 3:] = 255
        big = capi.Context(0, w, hgt, 1) if w > 3840 else ctx
        Ls, Rs = to_device(L), to_device(R)
        a, b = torch.full_like(Ls, 0xCD), torch.full_like(Ls, 0xEE)
        big.disparity_dev(Ls, Rs, a, block, D)
        assert capi.last_kernel_kind() == capi.KIND_COLWAVE
        big.disparity_dev(Ls, Rs, b, block, D, capi.FLAG_FORCE_GENERIC)
        torch.cuda.synchronize()
        got = to_host(a)
        assert np.array_equal(got, to_host(b)), (block, w, D)
        for i in range(3):
            reg = (0, hh + 1, w, hh + 3)
            assert np.array_equal(got[i][hh + 1:hh + 3], oracle.disparity_rows_mt(L[i], R[i], block, D, hh + 1, hh + 3, min(16, os.cpu_count() or 1))), (block, w, D, i)
        if big is not ctx:
            big.close()


def test_packed_mixed_chunk_plans(ctx):
    """Round 4, the packed kernels (BlockSize <= 15): 4 waves of ND candidates followed by 4 waves of ND - 4 (+ the odd one) --
    81 = 4*12 + 4*8 + 1 (BlockSize 11..15), 113 = 4*16 + 4*12 + 1, 145 = 4*20 + 4*16 + 1, 177 = 4*24 + 4*20 + 1 (BlockSize <= 9), and the
    last pass of 208 / 240.  BlockSize 9 and 15 hold four row loops there (two chunk sizes x triple-column / plain form): widths
    around their tile boundaries (one tile + 2, two tiles + 2, ...), so that both forms run in one launch.  Random, shifted,
    extreme and flat frames; every frame against the generic path on the device, two rows of each against the oracle."""
    import torch
    rng = np.random.default_rng(45)
    cases = {3: (112, 144, 176, 240), 5: (112, 176), 7: (112, 144, 240), 9: (112, 144, 176, 240), 11: (80, 112, 208, 240),
             13: (80, 112, 208), 15: (80, 112, 208, 240)}
    for block, Ds in cases.items():
        hh = block // 2
        hgt = 2 * hh + 9
        for D in Ds:
            pl = capi.plan_describe(1920, hgt, block, D)
            assert pl and pl[-1]["packed"] == 1 and pl[-1]["nwh"] == 4 and pl[-1]["nw"] == 8 and pl[-1]["extra"] == 1, (block, D, pl)
            for w in ((258, 507, 745) if block == 9 else (258, 498, 741) if block == 15 else (301, 770)):
                if w <= D + 2 * hh + 2:
                    continue                                    # narrow frames run the unpacked variant
                L = rng.integers(0, 256, size=(4, hgt, w), dtype=np.uint8)
                R = rng.integers(0, 256, size=(4, hgt, w), dtype=np.uint8)
                R[1] = np.roll(L[1], -min(D, 37), axis=1)
                L[2] = 255; R[2] = 0; R[2][:, w // 3:] = 255; R[2][:, 2 * w // 3:] = 0
                L[3] = 91; R[3] = 91
                Ls, Rs = to_device(L), to_device(R)
                a, b = torch.full_like(Ls, 0xCD), torch.full_like(Ls, 0xEE)
                ctx.disparity_dev(Ls, Rs, a, block, D, capi.FLAG_RAW)
                assert capi.last_kernel_kind() == capi.KIND_COLWAVE
                ctx.disparity_dev(Ls, Rs, b, block, D, capi.FLAG_RAW | capi.FLAG_FORCE_GENERIC)
                torch.cuda.synchronize()
                got = to_host(a)
                assert np.array_equal(got, to_host(b)), (block, w, D)
                for i in range(4):
                    reg = (0, hh + 3, w, hh + 5)
                    assert np.array_equal(got[i][hh + 3:hh + 5], oracle.disparity_closed(L[i], R[i], block, D, region=reg, raw=True)), (block, w, D, i)
    for block, D in ((9, 112), (15, 80), (7, 176)):               # 1080p width, the batched strip geometry
        g = torch.Generator(device="cuda").manual_seed(block * 1000 + D)
        Ls = torch.randint(0, 256, (6, 60, 1920), dtype=torch.uint8, device="cuda", generator=g)
        Rs = torch.roll(Ls, -23, dims=2) ^ (torch.randint(0, 16, Ls.shape, device="cuda", generator=g) == 0).to(torch.uint8)
        a, b = torch.empty_like(Ls), torch.empty_like(Ls)
        ctx.disparity_dev(Ls, Rs, a, block, D)
        ctx.disparity_dev(Ls, Rs, b, block, D, capi.FLAG_FORCE_GENERIC)
        torch.cuda.synchronize()
        assert torch.equal(a, b), (block, D)


def test_multi_pass_disparity_ranges(ctx):
    """D+1 candidates beyond the 16 waves of a workgroup run as several launches that hand a u32 key
    per pixel from pass to pass (colwave_plan.hip, plan_job): every UI value of MaxDisparity
    (cmd/handlers/params.go:45-52: up to 256) for every BlockSize the column-wave kernel covers,
    and larger ranges, incl. frames whose SADs are the largest possible."""
    rng = np.random.default_rng(11)
    for block, Ds in ((9, (400, 500)), (11, (208, 224, 240, 256, 300)), (13, (208, 256)), (15, (208, 224, 240, 256)),
                      (16, (208, 256, 300)), (17, (256,))):
        hh = block // 2
        hgt = 2 * hh + 5
        for D in Ds:
            w = D + 2 * hh + 150
            assert capi.kernel_kind(w, hgt, block, D) == 1, (block, D)
            Lr = rng.integers(0, 256, size=(hgt, w), dtype=np.uint8)
            Rr = np.roll(Lr, -(D - 3), axis=1)                 # true shift D-3: found only in the last pass
            Rr[:, -(D - 3):] = rng.integers(0, 256, size=(hgt, D - 3), dtype=np.uint8)
            Lx = np.full((hgt, w), 255, np.uint8); Rx = np.zeros((hgt, w), np.uint8); Rx[:, w // 2:] = 255
            for name, (L, R) in (("shifted", (Lr, Rr)), ("extreme", (Lx, Rx))):
                got = ctx.disparity(L, R, block, D)
                assert np.array_equal(got, _expect(L, R, block, D)), (name, block, D)
    L, R = sha.synth.synth_pair(91, 900, 60)                    # a batch (frames chunked against the key scratch) and a region
    got = ctx.disparity_batch([L, R, L], [R, L, L], 11, 256)
    for g, (a, b) in zip(got, ((L, R), (R, L), (L, L))):
        assert np.array_equal(g, _expect(a, b, 11, 256))
    reg = (300, 7, 777, 51)
    assert np.array_equal(ctx.disparity_region(L, R, 15, 240, reg), oracle.disparity_closed(L, R, 15, 240, region=reg))


def test_large_blocks_two_lane_halo(ctx):
    """BlockSize 18..31 (the UI goes to 31, cmd/handlers/params.go:37-44): windows wider than a
    lane's 8 columns + one neighbour lane; unpacked sums, penalties up to 31 columns deep, and
    the multi-pass split at 64 candidates."""
    rng = np.random.default_rng(23)
    for block in (18, 19, 21, 23, 25, 27, 29, 31):
        hh = block // 2
        hgt = 2 * hh + 4
        for D, w in ((16, 150), (64, 200), (65, 200), (100, 230), (256, 330)):
            assert capi.kernel_kind(w, hgt, block, D) == 1, (block, D)
            Lr = rng.integers(0, 256, size=(hgt, w), dtype=np.uint8)
            Rr = rng.integers(0, 256, size=(hgt, w), dtype=np.uint8)
            Lx = np.full((hgt, w), 255, np.uint8); Rx = np.zeros((hgt, w), np.uint8); Rx[:, w // 2:] = 255
            for name, (L, R) in (("random", (Lr, Rr)), ("extreme", (Lx, Rx))):
                raw = D <= 255
                got = ctx.disparity(L, R, block, D, capi.FLAG_RAW if raw else 0)
                assert np.array_equal(got, _expect(L, R, block, D, raw=raw)), (name, block, D)
    L, R = sha.synth.synth_pair(55, 1100, 90)                   # several tiles (480 output columns each) and strips
    for block in (19, 31):
        assert np.array_equal(ctx.disparity(L, R, block, 64), ctx.disparity(L, R, block, 64, capi.FLAG_FORCE_GENERIC))
    assert capi.kernel_kind(640, 480, 33, 64) == capi.KIND_GENERIC_SLIDING   # beyond 31: the generic path (running-sum form at this size)


def test_tall_frame_many_strips_and_tiles(ctx):
    """Strip seams (2h warm-up rows) and tile seams (halo lanes) on a frame taller/wider than any tile."""
    L, R = sha.synth.synth_pair(77, 1100, 700)
    for block, maxdisp in ((9, 64), (5, 16), (13, 48)):
        got = ctx.disparity(L, R, block, maxdisp)
        for reg in ((0, 0, 1100, 3), (0, 697, 1100, 700), (240, 300, 270, 330), (490, 100, 530, 110), (1080, 0, 1100, 700)):
            assert np.array_equal(got[reg[1]:reg[3], reg[0]:reg[2]], oracle.disparity_closed(L, R, block, maxdisp, region=reg)), (block, maxdisp, reg)
        assert np.array_equal(got, ctx.disparity(L, R, block, maxdisp, capi.FLAG_FORCE_GENERIC))


@pytest.mark.parametrize("flags", BOTH)
def test_flat_images_tie_to_zero(ctx, flags):
    L = np.full((90, 300), 200, np.uint8)
    assert not ctx.disparity(L, L.copy(), 9, 64, flags).any()
    # ties between distinct d: periodic texture -> lowest d must win (strict '<', sad.go:79)
    x = np.arange(300)
    P = np.tile(((x % 8) * 30).astype(np.uint8), (60, 1))
    got = ctx.disparity(P, P.copy(), 9, 64, flags | capi.FLAG_RAW)
    assert np.array_equal(got, _expect(P, P.copy(), 9, 64, raw=True))
    assert not got.any()


@pytest.mark.parametrize("flags", BOTH)
def test_strided_inputs_and_output(ctx, flags):
    big_l, big_r = sha.synth.synth_pair(77, 400, 120)
    L, R = big_l[10:100, 37:337], big_r[10:100, 37:337]          # non-contiguous views, odd base alignment
    out = np.full((90, 333), 0xAB, np.uint8)
    ctx.disparity(L, R, 9, 64, flags, out=out[:, 5:305])
    exp = _expect(np.ascontiguousarray(L), np.ascontiguousarray(R), 9, 64)
    assert np.array_equal(out[:, 5:305], exp)
    assert (out[:, :5] == 0xAB).all() and (out[:, 305:] == 0xAB).all()     # nothing written outside


@pytest.mark.parametrize("flags", BOTH)
def test_regions_like_the_callers(ctx, flags):
    """InputChunk.Region semantics: OutputCamera row bands (output.go:172-187) and RunSad tiles
    (sad.go:138-153); output is region-local row-major (sad.go:48-50,91)."""
    L, R = sha.synth.synth_pair(9, 320, 131)
    full = _expect(L, R, 9, 64)
    chunk = max(1, 131 // (32 * 4))
    for y in (0, 1, 64, 129, 130):
        reg = (0, y, 320, min(y + chunk, 131))
        assert np.array_equal(ctx.disparity_region(L, R, 9, 64, reg, flags), full[reg[1]:reg[3], reg[0]:reg[2]])
    for reg in sha.despair.run_sad_tiles(320, 131, 8)[::7] + [(311, 120, 320, 131), (0, 0, 1, 1), (100, 50, 101, 51)]:
        assert np.array_equal(ctx.disparity_region(L, R, 9, 64, reg, flags), full[reg[1]:reg[3], reg[0]:reg[2]])


def _guarded_copy(a):
    """A copy of the 2-D u8 array whose LAST byte is the last byte before a PROT_NONE page: a host-side read of
    even one byte past the image (e.g. a copy sized by the other image's dimensions) faults instead of passing."""
    import ctypes
    import mmap
    page = mmap.PAGESIZE
    n = a.size
    body = (n + page - 1) // page * page
    m = mmap.mmap(-1, body + page)
    base = ctypes.addressof(ctypes.c_char.from_buffer(m))
    libc = ctypes.CDLL(None, use_errno=True)
    assert libc.mprotect(ctypes.c_void_p(base + body), ctypes.c_size_t(page), 0) == 0, ctypes.get_errno()   # PROT_NONE
    g = np.frombuffer(m, np.uint8, n, body - n).reshape(a.shape)
    g[...] = a
    return g, m          # keep m alive


def test_left_and_right_pictures_of_different_sizes(ctx):
    """pkg/camera/output.go:134-162 decodes left.png and right.png independently and the reference copes with a
    right image of another size: windows clamp to the LEFT image, the right window is walked with the same extent
    and the loops break where it leaves right.Rect.Max (sad.go:212-218,225-227,231-233).  sadhip_disparity_lr_u8
    and the three bindings' RunSad / worker do the same -- compared here with the literal restatement run with the
    right image's own size -- and nothing is read outside the right image (it ends at a PROT_NONE page)."""
    rng = np.random.default_rng(77)
    W, H = 200, 96
    L = rng.integers(0, 256, (H, W), dtype=np.uint8)
    full = np.roll(L, -5, axis=1) ^ (rng.integers(0, 64, (H, W), dtype=np.uint8) == 0).astype(np.uint8)
    cases = [(150, 70), (200, 60), (120, 96), (1, 1), (37, 3), (199, 95), (260, 120), (200, 120), (230, 96)]
    for rw, rh in cases:
        Rsrc = np.zeros((rh, rw), np.uint8)
        Rsrc[:min(rh, H), :min(rw, W)] = full[:min(rh, H), :min(rw, W)]
        if rw > W or rh > H:
            Rsrc[H:, :] = 201; Rsrc[:, W:] = 202                      # never addressed (right indices <= left indices)
        R, keep = _guarded_copy(Rsrc)
        smaller = rw < W or rh < H
        for block, maxdisp in ((9, 64), (7, 16), (16, 32)):
            exp = oracle.disparity_literal_lr(L, R, block, maxdisp)
            got = ctx.disparity_lr(L, R, block, maxdisp)
            assert np.array_equal(got, exp), (rw, rh, block, maxdisp)
            # a smaller right image runs the generic kernel (dropped window terms); a larger one is the ordinary job
            assert capi.last_kernel_kind() == (capi.KIND_GENERIC if smaller else capi.kernel_kind(W, H, block, maxdisp))
            if not smaller:
                assert np.array_equal(got, _expect(L, np.ascontiguousarray(R[:H, :W]), block, maxdisp))
        # regions like the worker's chunks (sad.go:47-101), region-local output
        for reg in ((0, 0, W, 7), (13, 40, 181, 77), (W - 9, H - 5, W, H)):
            exp = oracle.disparity_literal_lr(L, R, 9, 64, region=reg)
            assert np.array_equal(ctx.disparity_lr(L, R, 9, 64, region=reg), exp), (rw, rh, reg)
        # the Python mirror of the Go API: RunSad and the channel worker
        assert np.array_equal(sha.despair.RunSad(L, R, 9, 64), oracle.disparity_literal_lr(L, R, 9, 64))
        inp, outp = sha.despair.SetupConcurrentSAD(2)
        bands = [(0, y, W, min(y + 17, H)) for y in range(0, H, 17)]
        for b in bands:
            inp.send(sha.despair.InputChunk(L, R, b))
        inp.close()
        assert np.array_equal(sha.despair.AssembleDisparityMap(outp, (0, 0, W, H), len(bands)), oracle.disparity_literal_lr(L, R, 9, 64))
        del R, keep
    # hardware-mirror modes have one image size: a smaller right image is an error there, not a guess
    with pytest.raises(capi.SadHipError) as e:
        ctx.disparity_lr(L, np.zeros((50, 50), np.uint8), 15, 64, flags=capi.FLAG_MIRROR_SADC)
    assert e.value.status == capi.ERR_DIMS
    with pytest.raises(capi.SadHipError) as e:
        ctx.disparity_lr(L, np.zeros((50, 50), np.uint8), 9, 64, region=(0, 0, W + 1, H))
    assert e.value.status == capi.ERR_REGION


def test_channel_api_round_trip(ctx):
    """examples/run.stream.go:30-59 shape: long-lived pipeline, row-band chunks, assemble."""
    d = sha.despair
    L, R = sha.synth.synth_pair(21, 200, 97)
    d.SetDefaultParams(d.Parameters(BlockSize=7, MaxDisparity=32))
    inp, outp = d.SetupConcurrentSAD(32)
    for _ in range(2):
        chunkSize = max(1, 97 // (32 * 4))
        n = 0
        for y in range(0, 97, chunkSize):
            inp.send(d.InputChunk(L, R, (0, y, 200, min(y + chunkSize, 97))))
            n += 1
        got = d.AssembleDisparityMap(outp, (0, 0, 200, 97), n)
        assert np.array_equal(got, _expect(L, R, 7, 32))       # hole-free, unlike sad.go:181-184
    inp.close()
    assert list(outp) == []
    d.SetDefaultParams(d.Parameters(BlockSize=16, MaxDisparity=64))


@pytest.mark.timeout(180)
def test_channel_api_reference_example_shape_whole_frame_fed_before_reading(ctx):
    """examples/run.stream.go:30-59 as it is: 640x480, 32 workers, chunks of Dy / (32 * 4) = 3 rows = 160 chunks, ALL sent
    before AssembleDisparityMap starts to read.  The reference has room for 5 x numWorkers = 160 chunks under way (two
    channels of 64 and 32 goroutines holding one each); a device worker with the same two channels and one chunk in hand had
    room for 129 and would have left this feeder blocked for ever.  Also: the chunks queued together run as one device call
    per group (same bytes as the frame-level call), chunks of two frames in one stream, tiles (sad.go:138-153) that do not
    continue each other, and an empty region."""
    d = sha.despair
    L, R = sha.synth.synth_pair(31, 640, 480)
    L2, R2 = sha.synth.synth_pair(32, 640, 480)
    d.SetDefaultParams(d.Parameters(BlockSize=16, MaxDisparity=32))      # run.stream.go:15-19
    want, want2 = _expect(L, R, 16, 32), _expect(L2, R2, 16, 32)
    inp, outp = d.SetupConcurrentSAD(32)
    chunkSize = max(1, 480 // (32 * 4))
    for l, r, w_ in ((L, R, want), (L2, R2, want2), (L, R, want)):
        n = 0
        for y in range(0, 480, chunkSize):
            inp.send(d.InputChunk(l, r, (0, y, 640, min(y + chunkSize, 480))))
            n += 1
        assert n == 160
        assert np.array_equal(d.AssembleDisparityMap(outp, (0, 0, 640, 480), n), w_)
    # two frames' bands interleaved, RunSad's tiles, an empty region: nothing continues anything
    regs = [(0, 0, 640, 100), (0, 100, 320, 200), (320, 100, 640, 200), (0, 200, 640, 200), (0, 200, 640, 480)]
    for reg in regs:
        inp.send(d.InputChunk(L, R, reg)); inp.send(d.InputChunk(L2, R2, reg))
    got = [next(iter(outp)) for _ in range(2 * len(regs))]
    for i, o in enumerate(got):
        x0, y0, x1, y1 = regs[i // 2]
        assert o.Region == regs[i // 2]
        assert np.array_equal(o.DisparityData.reshape(y1 - y0, x1 - x0), (want if i % 2 == 0 else want2)[y0:y1, x0:x1])
    inp.close()
    assert list(outp) == []
    d.SetDefaultParams(d.Parameters(BlockSize=16, MaxDisparity=64))


def test_batch_host_pipeline(ctx):
    pairs = [sha.synth.synth_pair(1000 + i, 256, 64) for i in range(5)]
    outs = ctx.disparity_batch([p[0] for p in pairs], [p[1] for p in pairs], 9, 64)
    for (L, R), o in zip(pairs, outs):
        assert np.array_equal(o, _expect(L, R, 9, 64))
    assert ctx.disparity_batch([], [], 9, 64) == []
    k_ms, t_ms = ctx.last_timing()
    assert k_ms > 0 and t_ms >= 0


def test_page_locked_host_memory_paths(ctx):
    """sadhip_host_alloc / sadhip_host_register: frames in page-locked memory run as groups of frames per launch with
    asynchronous linear copies (frames that follow each other in memory as one copy).  Same results as the pageable
    path on every layout: one contiguous block, scattered frames, a batch that is only partly page-locked (ordinary
    path), registered numpy memory, the single-frame entry points, the pool."""
    W, H, B, D = 320, 200, 9, 64
    n = 19                                                            # not a multiple of the group size (8)
    pairs = [sha.synth.synth_pair(500 + i, W, H) for i in range(n)]
    exp = [_expect(l, r, B, D) for l, r in pairs]
    with capi.HostBuffer(3 * n * W * H) as hb:
        blk = hb.view((3, n, H, W))
        for i, (l, r) in enumerate(pairs):
            blk[0, i], blk[1, i] = l, r
        blk[2] = 0xEE
        # 1. one contiguous block per plane: runs of adjacent frames travel as single copies
        outs = ctx.disparity_batch([blk[0, i] for i in range(n)], [blk[1, i] for i in range(n)], B, D, outs=[blk[2, i] for i in range(n)])
        assert all(np.array_equal(outs[i], exp[i]) for i in range(n))
        assert capi.last_kernel_kind() == capi.kernel_kind(W, H, B, D) == capi.KIND_COLWAVE
        k_ms, t_ms = ctx.last_timing()
        assert 0 < k_ms <= t_ms
        # 2. scattered order (no two consecutive frames adjacent), outputs too
        order = [(7 * i) % n for i in range(n)]
        blk[2] = 0xEE
        outs = ctx.disparity_batch([blk[0, i] for i in order], [blk[1, i] for i in order], B, D, outs=[blk[2, i] for i in order])
        assert all(np.array_equal(blk[2, i], exp[i]) for i in range(n))
        # 3. outputs in pageable memory: the whole batch takes the ordinary path, same answer
        outs = ctx.disparity_batch([blk[0, i] for i in range(n)], [blk[1, i] for i in range(n)], B, D)
        assert all(np.array_equal(outs[i], exp[i]) for i in range(n))
        # 4. single-frame entry points with page-locked inputs (linear async upload), whole frame and region
        got = ctx.disparity(blk[0, 3], blk[1, 3], B, D, out=blk[2, 0])
        assert np.array_equal(got, exp[3])
        assert np.array_equal(ctx.disparity_region(blk[0, 5], blk[1, 5], B, D, (11, 50, 300, 130)), exp[5][50:130, 11:300])
        assert np.array_equal(ctx.disparity_lr(blk[0, 6], blk[1, 6][:150, :], B, D),
                              oracle.disparity_literal_lr(np.ascontiguousarray(blk[0, 6]), np.ascontiguousarray(blk[1, 6][:150, :]), B, D))
        # 5. strided rows inside page-locked memory (row stride > width)
        wide = hb.view((3, n, H, W))[0, :2].reshape(2 * H, W)      # reuse: two frames' worth of rows
        Ls, Rs = pairs[1][0][:, :250], pairs[1][1][:, :250]
        wide[:H, :250], wide[H:, :250] = Ls, Rs
        assert np.array_equal(ctx.disparity(wide[:H, :250], wide[H:, :250], B, D), _expect(np.ascontiguousarray(Ls), np.ascontiguousarray(Rs), B, D))
        # 5b. strided rows in a batch: inputs AND outputs are column windows of wider page-locked images; nothing outside
        #     an output window may be touched (a linear download would carry the bytes between rows)
        wide3 = hb.view((3, 6, H, W))
        win = slice(20, 20 + 250)
        for i in range(6):
            wide3[0, i] = pairs[i][0]; wide3[1, i] = pairs[i][1]
        wide3[2] = 0xA5
        outs = ctx.disparity_batch([wide3[0, i][:, win] for i in range(6)], [wide3[1, i][:, win] for i in range(6)], B, D,
                                   outs=[wide3[2, i][:, win] for i in range(6)])
        for i in range(6):
            assert np.array_equal(wide3[2, i][:, win], _expect(np.ascontiguousarray(pairs[i][0][:, win]), np.ascontiguousarray(pairs[i][1][:, win]), B, D)), i
            assert (wide3[2, i][:, :20] == 0xA5).all() and (wide3[2, i][:, 270:] == 0xA5).all(), i       # untouched
        # 6. the pool over page-locked frames (two contexts on this GPU)
        for i, (l, r) in enumerate(pairs):
            blk[0, i], blk[1, i] = l, r
        with capi.Pool([0, 0], W, H) as pool:
            outs = pool.batch([blk[0, i] for i in range(n)], [blk[1, i] for i in range(n)], B, D, outs=[blk[2, i] for i in range(n)])
            assert all(np.array_equal(outs[i], exp[i]) for i in range(n))
    # 7. memory the caller owns, page-locked in place: whole pages of a mapping of its own (ABI 5: the only kind accepted;
    #    numpy's heap arrays are refused -- tests/test_abi_cpu.py::test_host_register_contract_is_checked_before_any_lock)
    with capi.MappedBuffer(3 * 4 * H * W) as mb:
        own = mb.view((3, 4, H, W))
        for i in range(4):
            own[0, i], own[1, i] = pairs[i]
        outs = ctx.disparity_batch([own[0, i] for i in range(4)], [own[1, i] for i in range(4)], B, D, outs=[own[2, i] for i in range(4)])
        assert all(np.array_equal(outs[i], exp[i]) for i in range(4))
        del own, outs
    heap = np.zeros(1 << 16, np.uint8)
    assert capi.lib().sadhip_host_register(heap.ctypes.data, heap.nbytes) == capi.ERR_ARG      # unaligned and / or in the program break
    # 8. error codes, no crash: unknown pointers, double free
    l = capi.lib()
    junk = np.zeros(16, np.uint8)
    assert l.sadhip_host_free(junk.ctypes.data) == capi.ERR_ARG
    assert l.sadhip_host_unregister(junk.ctypes.data) == capi.ERR_ARG
    assert l.sadhip_host_free(None) == capi.ERR_NULL and l.sadhip_host_alloc(0, None) == capi.ERR_NULL
    import ctypes
    p = ctypes.c_void_p()
    assert l.sadhip_host_alloc(0, ctypes.byref(p)) == capi.ERR_ARG
    assert l.sadhip_host_alloc(4096, ctypes.byref(p)) == capi.OK and p.value
    assert l.sadhip_host_free(p) == capi.OK and l.sadhip_host_free(p) == capi.ERR_ARG


def test_page_locked_batch_slot_reuse_and_failure_drains_the_streams(tmp_path):
    """the page-locked batch pipeline has 4 slots of 8 frames; a slot is reused from the fifth group on (wait for
    its download, read its kernel events, make the upload stream wait) -- only batches of more than 32 frames get there.
    (1) 43 distinct small frames (5 groups + 3 frames) in page-locked memory, every frame against the oracle, twice (the
    second call starts on used slots).  (2) In the diagnostics build a failure is injected after the kernel of group 4
    (the first reused slot) has been queued: the call must return the error with NOTHING in flight -- outputs overwritten
    right after the return stay as written -- and the same context must run the whole batch correctly afterwards."""
    n, w, h, B, D = 43, 64, 48, 9, 32
    frames = [sha.synth.synth_pair(5000 + i, w, h) for i in range(n)]
    exp = [oracle.disparity_closed(L, R, B, D) for L, R in frames]
    with capi.Context(0, w, h, 1) as c, capi.HostBuffer(3 * n * w * h) as hb:
        blk = hb.view((3, n, h, w))
        for i, (L, R) in enumerate(frames):
            blk[0, i] = L; blk[1, i] = R
        for rep in range(2):
            blk[2] = 0xEE
            outs = c.disparity_batch([blk[0, i] for i in range(n)], [blk[1, i] for i in range(n)], B, D, outs=[blk[2, i] for i in range(n)])
            for i in range(n):
                assert np.array_equal(outs[i], exp[i]), (rep, i)
    import subprocess, sys, textwrap
    script = textwrap.dedent(f"""
        import os, sys, time
        sys.path[:0] = [{REPO!r}, {os.path.join(REPO, "oracle")!r}]
        import numpy as np
        import steroscopic_hardware_amd as sha, oracle
        from steroscopic_hardware_amd import capi
        capi.LIB_PATH = os.path.join(os.path.dirname(capi.LIB_PATH), "libsadhip_tune.so")
        n, w, h, B, D = {n}, {w}, {h}, {B}, {D}
        frames = [sha.synth.synth_pair(5000 + i, w, h) for i in range(n)]
        with capi.Context(0, w, h, 1) as c, capi.HostBuffer(3 * n * w * h) as hb:
            blk = hb.view((3, n, h, w))
            for i, (L, R) in enumerate(frames):
                blk[0, i] = L; blk[1, i] = R
            blk[2] = 0xEE
            args = ([blk[0, i] for i in range(n)], [blk[1, i] for i in range(n)], B, D)
            os.environ["SADHIP_TUNE_FAIL_GROUP"] = "4"
            try:
                c.disparity_batch(*args, outs=[blk[2, i] for i in range(n)])
                print("NOERROR"); sys.exit(1)
            except capi.SadHipError as e:
                assert e.status == capi.ERR_HIP and "injected failure" in str(e), e
            blk[2] = 0x5A                                    # anything still in flight would overwrite this
            blk[0] ^= 0xFF                                   # ... or read this
            time.sleep(0.2)
            assert (blk[2] == 0x5A).all(), "a download landed after the failing call returned"
            blk[0] ^= 0xFF
            del os.environ["SADHIP_TUNE_FAIL_GROUP"]
            outs = c.disparity_batch(*args, outs=[blk[2, i] for i in range(n)])
            for i, (L, R) in enumerate(frames):
                assert np.array_equal(outs[i], oracle.disparity_closed(L, R, B, D)), i
        print("DRAINED-OK")
    """)
    r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "DRAINED-OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_host_register_refuses_overlapping_ranges():
    """the registry of page-locked ranges is sorted and DISJOINT (is_pinned looks at one predecessor, removal goes
    by start address): a registration that shares a page with a known range -- a sub-range of a sadhip_host_alloc block, a
    sub- or super-range of an earlier registration, a partial overlap -- is SADHIP_ERR_ARG and registers nothing.  ABI 5:
    registrations are whole pages of a mapping of the caller's own (here one anonymous mmap that outlives them all)."""
    import ctypes
    l = capi.lib()
    with capi.HostBuffer(1 << 20) as hb:
        base = hb.view((1 << 20,)).ctypes.data
        for off, size in ((0, 4096), (8192, 4096), ((1 << 20) - 4096, 4096)):
            assert l.sadhip_host_register(ctypes.c_void_p(base + off), size) == capi.ERR_ARG, (off, size)
    import mmap
    mm = mmap.mmap(-1, (1 << 20) + 8192)                                                   # a mapping of our own (ABI 5 accepts nothing else)
    a = np.frombuffer(mm, dtype=np.uint8)
    p = a.ctypes.data
    PG = 4096
    assert p % PG == 0
    assert l.sadhip_host_register(ctypes.c_void_p(p + 16 * PG), 64 * PG) == capi.OK
    for off, size in ((16 * PG, 64 * PG), (17 * PG, PG), (0, 32 * PG), (70 * PG, 32 * PG), (0, 200 * PG), (79 * PG, 2 * PG)):
        assert l.sadhip_host_register(ctypes.c_void_p(p + off), size) == capi.ERR_ARG, (off, size)
        assert b"shares a page" in l.sadhip_last_error(), l.sadhip_last_error()
    for off, size in ((80 * PG - 100, PG), (15 * PG, PG + 1), (16 * PG - 1, 1)):           # not whole pages: refused before the overlap check
        assert l.sadhip_host_register(ctypes.c_void_p(p + off), size) == capi.ERR_ARG, (off, size)
        assert b"whole 4 KiB pages" in l.sadhip_last_error(), l.sadhip_last_error()
    assert l.sadhip_host_register(ctypes.c_void_p(p + 15 * PG), PG) == capi.OK             # the neighbouring page: nothing shared
    assert l.sadhip_host_unregister(ctypes.c_void_p(p + 15 * PG)) == capi.OK
    assert l.sadhip_host_register(ctypes.c_void_p(p), 16 * PG) == capi.OK                  # adjacent pages, nothing shared
    assert l.sadhip_host_unregister(ctypes.c_void_p(p + PG)) == capi.ERR_ARG                # not the start of a range
    assert l.sadhip_host_unregister(ctypes.c_void_p(p)) == capi.OK
    assert l.sadhip_host_unregister(ctypes.c_void_p(p + 16 * PG)) == capi.OK
    assert l.sadhip_host_unregister(ctypes.c_void_p(p + 16 * PG)) == capi.ERR_ARG
    a[:] = 1                                                                                # the mapping is still ours
    del a
    mm.close()


def test_pool_workers_are_pinned_next_to_their_device():
    """sadhip_pool_describe (ABI 4): every worker thread of the pool is pinned at creation to the CPUs of its device's NUMA
    node (PCI bus id -> sysfs local_cpulist, cut down to this process's affinity mask); best effort -- asserted as far as
    the box lets it be read: bus id, node, and when pinned a CPU list inside the process's own mask."""
    import re
    with capi.Pool([0, 0], 640, 480) as pool:
        d = pool.describe()
        assert [e["entry"] for e in d] == [0, 1] and all(e["device"] == 0 for e in d)
        for e in d:
            assert re.fullmatch(r"[0-9a-f]{4}:[0-9a-f]{2}:[0-9a-f]{2}\.[0-9a-f]", e["pci"]) or e["pci"] == "?", e
            assert e["pinned"] in (0, 1) and isinstance(e["numa_node"], int)
            sysfs = f"/sys/bus/pci/devices/{e['pci']}/local_cpulist"
            if e["pinned"]:
                cpus = set()
                for part in e["cpus"].split(","):
                    a, _, b = part.partition("-")
                    cpus.update(range(int(a), int(b or a) + 1))
                assert cpus and cpus <= set(os.sched_getaffinity(0)), e
                assert os.path.exists(sysfs)
            else:
                assert e["cpus"] == "-"
        one = capi.device_describe(0)                            # the same lookup for a host that drives one context itself
        assert one["device"] == 0 and one["pci"] == d[0]["pci"] and one["numa_node"] == d[0]["numa_node"]
        assert one["cpus"] <= set(os.sched_getaffinity(0)) and (bool(one["cpus"]) == bool(d[0]["pinned"]))
        with pytest.raises(capi.SadHipError) as ei:
            capi.device_describe(99)
        assert ei.value.status == capi.ERR_NO_DEVICE
        # per-entry page-locked memory, allocated and first-touched by the entry's worker thread
        with pool.host_alloc(1, 3 * 320 * 100) as hb:
            v = hb.view((3, 100, 320))
            assert not v.any()                                    # first-touched (zeroed) by the worker
            v[0], v[1] = sha.synth.synth_pair(78, 320, 100)
            outs = pool.batch([v[0]], [v[1]], 9, 32, outs=[v[2]])
            assert np.array_equal(outs[0], oracle.disparity_closed(v[0], v[1], 9, 32))
        with pytest.raises(capi.SadHipError) as ei:
            pool.host_alloc(2, 4096)
        assert ei.value.status == capi.ERR_ARG
        # WHERE such memory lands.  The runtime places page-locked memory by the allocating thread's current device
        # (the worker now selects its own before it allocates) and the thread runs on the device's CPUs: ask the kernel which
        # node the pages are on (move_pages in its query form) and compare with the device's node -- as far as a one-GPU box
        # can show it: both entries are device 0 here; skipped where the box hides the answer.
        import ctypes
        for entry in (0, 1):
            if not (d[entry]["pinned"] and d[entry]["numa_node"] >= 0):
                continue
            with pool.host_alloc(entry, 64 << 20) as hb:
                base = hb.array.ctypes.data
                n = 64
                pages = (ctypes.c_void_p * n)(*[base + i * (1 << 20) for i in range(n)])
                status = (ctypes.c_int * n)(*([-1] * n))
                libc = ctypes.CDLL(None, use_errno=True)
                rc = libc.syscall(279, 0, ctypes.c_ulong(n), pages, None, status, 0)      # move_pages(self, n, pages, NULL, status, 0)
                if rc != 0 or any(s < 0 for s in status):
                    print(f"[pool placement] entry {entry}: move_pages unavailable here (rc {rc}, errno {ctypes.get_errno()}, status {sorted(set(status))})")
                    continue
                nodes = sorted(set(status))
                allowed = set()
                for ln in open("/proc/self/status"):
                    if ln.startswith("Mems_allowed_list:"):
                        for part in ln.split(":", 1)[1].strip().split(","):
                            a_, _, b_ = part.partition("-")
                            allowed.update(range(int(a_), int(b_ or a_) + 1))
                print(f"[pool placement] entry {entry}: device node {d[entry]['numa_node']}, pages on node(s) {nodes}, memory nodes this process may use {sorted(allowed)}")
                if d[entry]["numa_node"] in allowed:              # (a cgroup that keeps this process off the device's node decides otherwise)
                    assert nodes == [d[entry]["numa_node"]], (entry, nodes, d[entry])
        # and the pool still computes: two contexts on the one device, frames round-robin
        L, R = sha.synth.synth_pair(77, 320, 100)
        outs = pool.batch([L] * 5, [R] * 5, 9, 32)
        assert all(np.array_equal(o, oracle.disparity_closed(L, R, 9, 32)) for o in outs)


def test_pageable_memory_is_staged_in_bands_by_the_copy_threads(ctx):
    """Caller memory the library did not page-lock itself never reaches a copy engine: the host
    threads of the context copy it into / out of page-locked planes first -- rows split over the copy threads, frames
    above 8 MB in bands.  Frames large enough for more than one band (17 MB planes, 33 MB decoded pictures), strided
    views with odd base alignment on both sides, a guard page right behind the last input byte, region calls with a
    column offset, and a small frame (one thread, one band): same bytes as the device-resident entry point, nothing
    written outside the caller's rows."""
    import torch
    W, H, B, D = 8192, 2100, 5, 16
    big_l, big_r = sha.synth.synth_pair(4242, W + 96, H + 3)
    L, R = big_l[1:H + 1, 37:W + 37], big_r[2:H + 2, 11:W + 11]            # non-contiguous views, odd base alignment
    Lc, Rc = np.ascontiguousarray(L), np.ascontiguousarray(R)
    c = capi.Context(0, W, H, 1)
    try:
        dL, dR = to_device(Lc[None]), to_device(Rc[None])
        dO = torch.empty_like(dL)
        c.disparity_dev(dL, dR, dO, B, D)
        torch.cuda.synchronize()
        exp = to_host(dO[0])
        for y0 in (0, 1047, H - 4):                                           # band seams and both ends against the oracle
            assert np.array_equal(exp[y0:y0 + 4], oracle.disparity_rows_mt(Lc, Rc, B, D, y0, y0 + 4, 8))
        out = np.full((H, W + 45), 0xAB, np.uint8)
        c.disparity(L, R, B, D, out=out[:, 5:W + 5])
        assert np.array_equal(out[:, 5:W + 5], exp)
        assert (out[:, :5] == 0xAB).all() and (out[:, W + 5:] == 0xAB).all()
        # the last byte of each input is the last byte before an inaccessible page
        gl, ml = _guarded_copy(Lc)
        gr, mr = _guarded_copy(Rc)
        assert np.array_equal(c.disparity(gl, gr, B, D), exp)
        # three frames in flight through the batch entry point, outputs in caller arrays
        outs = [np.full((H, W), 0x11 * (i + 1), np.uint8) for i in range(4)]
        c.disparity_batch([gl, Lc, Lc, gl], [gr, Rc, gr, Rc], B, D, outs=outs)
        assert all(np.array_equal(o, exp) for o in outs)
        wide = np.full((5, H, W + 45), 0xCD, np.uint8)                      # strided frames in, strided rows out
        c.disparity_batch([L] * 5, [R] * 5, B, D, outs=[wide[i, :, 7:W + 7] for i in range(5)])
        assert all(np.array_equal(wide[i, :, 7:W + 7], exp) for i in range(5))
        assert (wide[:, :, :7] == 0xCD).all() and (wide[:, :, W + 7:] == 0xCD).all()
        del gl, gr; ml.close(); mr.close()
        # regions: a column offset inside the staged result rows, bands of the inputs only
        for reg in ((4097, 1000, 8192, 2100), (3, 0, 8000, 64), (8191, 2099, 8192, 2100)):
            assert np.array_equal(c.disparity_region(L, R, B, D, reg), exp[reg[1]:reg[3], reg[0]:reg[2]])
    finally:
        c.close()
    # decoded pictures: 33 MB each at 4K, strided (a sub-picture of a larger one)
    rng = np.random.default_rng(99)
    pic = rng.integers(0, 256, (2, 2160 + 2, 3840 + 8, 4), dtype=np.uint8)
    lp, rp = pic[0, 1:2161, 3:3843], pic[1, 2:2162, 5:3845]
    gl_, gr_ = ctx.gray(lp), ctx.gray(rp)
    assert np.array_equal(gl_, oracle.gray_from_nrgba(np.ascontiguousarray(lp)))
    assert np.array_equal(ctx.disparity_rgba(lp, rp, 7, 32), ctx.disparity(gl_, gr_, 7, 32))
    # a small frame: below the threshold of the copy threads
    l, r = sha.synth.synth_pair(5, 200, 90)
    assert np.array_equal(ctx.disparity(l, r, 9, 64), _expect(l, r, 9, 64))
    # the number of copy threads is the host's to choose (sadhip_ctx_set_copy_threads): same bytes with 1, 2, 32 and back
    assert ctx.copy_threads() == 6
    l1, r1 = sha.synth.synth_pair(7100, 1920, 1080)
    ref6 = ctx.disparity(l1, r1, 9, 64)
    for n in (1, 2, 32, 6):
        ctx.set_copy_threads(n)
        assert ctx.copy_threads() == n
        assert np.array_equal(ctx.disparity(l1, r1, 9, 64), ref6), n
    for n in (0, 33, -1):
        with pytest.raises(capi.SadHipError) as ei:
            ctx.set_copy_threads(n)
        assert ei.value.status == capi.ERR_ARG and ctx.copy_threads() == 6
    # two pool workers at once, each with copy threads of its own, frames in pageable memory
    pairs = [sha.synth.synth_pair(7000 + i, 1920, 1080) for i in range(3)]
    want = [ctx.disparity(l, r, 9, 64) for l, r in pairs]
    for (l, r), o in zip(pairs, want):
        for y0 in (0, 538, 1076):
            assert np.array_equal(o[y0:y0 + 4], oracle.disparity_rows_mt(l, r, 9, 64, y0, y0 + 4, 8))
    with capi.Pool([0, 0], 1920, 1080) as pool:
        outs = pool.batch([pairs[i % 3][0] for i in range(9)], [pairs[i % 3][1] for i in range(9)], 9, 64)
        assert all(np.array_equal(outs[i], want[i % 3]) for i in range(9))
        pool.set_copy_threads(2)
        outs = pool.batch([pairs[i % 3][0] for i in range(5)], [pairs[i % 3][1] for i in range(5)], 9, 64)
        assert all(np.array_equal(outs[i], want[i % 3]) for i in range(5))
        with pytest.raises(capi.SadHipError):
            pool.set_copy_threads(0)


def test_device_resident_batch(ctx):
    import torch
    Ls, Rs = sha.synth.synth_batch_torch([1000, 1001, 1002], 300, 80, "cuda:0")
    out = torch.full_like(Ls, 0xCD)
    ctx.disparity_dev(Ls, Rs, out, 9, 64)
    torch.cuda.synchronize()
    for i in range(3):
        L, R = sha.synth.synth_pair(1000 + i, 300, 80)
        assert np.array_equal(to_host(Ls[i]), L) and np.array_equal(to_host(Rs[i]), R)
        assert np.array_equal(to_host(out[i]), _expect(L, R, 9, 64))


_TR_SHAPES = {
    # BlockSize 9 (tile 256 columns, step 248): 258 = one tile + 2, 260 / 508: a triple-form tile whose last column is w-5, the
    # closest it gets to the row's end, 506 = two tiles + 2, ... and every w mod 4
    9: ((257, 20, 16), (258, 20, 64), (259, 22, 64), (260, 20, 64), (261, 20, 64), (505, 20, 64), (506, 20, 64), (507, 21, 33),
        (508, 20, 64), (600, 24, 128), (745, 20, 64), (1001, 18, 64), (530, 20, 200), (1920, 12, 64)),
    # BlockSize 15 (tile 256 columns, step 240, two-lane halos): 498 = step + tile + 2, 738 = two steps + tile + 2; the leftmost
    # tile always runs the plain form (it masks keys), with D 256 (second pass: candidates up to 256 > 248 - 7) so does the
    # second tile; D 16 .. 256 also walks the chunk sizes 4, 8, 12, 16
    15: ((257, 34, 16), (258, 34, 64), (261, 34, 64), (496, 34, 64), (497, 33, 64), (498, 34, 64), (499, 34, 128), (500, 34, 32),
         (501, 36, 64), (737, 34, 48), (738, 34, 128), (740, 34, 64), (742, 34, 256), (1001, 32, 128), (1300, 32, 256), (3840, 30, 128)),
}


@pytest.mark.parametrize("block", sorted(_TR_SHAPES))
def test_batches_triple_column_form(ctx, block):
    """BlockSize 9 and 15 run two forms of the row loop in one kernel (colwave_kernel.h, Cw::TR3): the triple-column
    form in the workgroups of the tiles left of the image's last two columns (BlockSize 15: and right of every pixel
    that lacks a candidate of the pass -- the triple form carries no key masks), the plain form in the rest.
    Widths around every tile boundary of that choice and every w mod 4, single- and multi-pass disparity ranges,
    random / shifted / extreme / flat frames; every frame against the generic kernel (itself pinned to the oracle
    above) and four of them against the oracle itself."""
    import torch
    n = 64 if block == 9 else 24
    rng = np.random.default_rng(909 + block)
    for w, h, D in _TR_SHAPES[block]:
        Lb = rng.integers(0, 256, size=(n, h, w), dtype=np.uint8)
        Rb = rng.integers(0, 256, size=(n, h, w), dtype=np.uint8)
        s = min(D, 37)
        Rb[1] = np.roll(Lb[1], -s, axis=1)                      # a true shift
        Lb[2] = 255; Rb[2] = 0; Rb[2][:, w // 2:] = 255         # the largest possible SADs (15 * 15 * 255 = 57 375 < 2^16)
        Lb[3] = 77; Rb[3] = 77                                   # flat: every candidate ties
        Rb[4] = Lb[4]                                            # identical: SAD 0 at d = 0
        Lb[5] = 255; Rb[5] = 0                                   # every window at the maximum: ties at the largest keys
        Ls, Rs = to_device(Lb), to_device(Rb)
        out = torch.full_like(Ls, 0xCD)
        ref = torch.full_like(Ls, 0xEE)
        raw = capi.FLAG_RAW if D <= 255 else 0
        ctx.disparity_dev(Ls, Rs, out, block, D, raw)
        assert capi.last_kernel_kind() == capi.KIND_COLWAVE
        ctx.disparity_dev(Ls, Rs, ref, block, D, raw | capi.FLAG_FORCE_GENERIC)
        torch.cuda.synchronize()
        got, exp = to_host(out), to_host(ref)
        assert np.array_equal(got, exp), (block, w, h, D, np.argwhere(got != exp)[:4])
        for i in (0, 1, 2, n - 1):                               # and against the oracle itself
            assert np.array_equal(got[i], _expect(Lb[i], Rb[i], block, D, raw=bool(raw))), (block, w, D, i)
        # a single frame and a region of one (other strip counts, other output origin)
        one = ctx.disparity(Lb[0], Rb[0], block, D, raw)
        assert np.array_equal(one, got[0]), (block, w, D)
        reg = (w // 3, 3, w - 1, h - 2)
        assert np.array_equal(ctx.disparity_region(Lb[1], Rb[1], block, D, reg, raw), got[1][reg[1]:reg[3], reg[0]:reg[2]]), (block, w, D)


def test_multi_pass_batch_beyond_one_grid(ctx):
    """a multi-pass job (D + 1 > 129 candidates: a u32 key per pixel travels between the
    passes) over more frames than one sub-launch covers (32768): the key rows of the later
    sub-launches must be their own.  33 000 tiny frames, D = 130, device-resident."""
    import torch
    n, w, h, block, D = 33000, 230, 6, 3, 200
    assert len(capi.plan_describe(w, h, block, D)) >= 2
    g = torch.Generator(device="cuda").manual_seed(5)
    Ls = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device="cuda", generator=g)
    Rs = torch.randint(0, 256, (n, h, w), dtype=torch.uint8, device="cuda", generator=g)
    out = torch.empty_like(Ls)
    ctx.disparity_dev(Ls, Rs, out, block, D)
    torch.cuda.synchronize()
    assert capi.last_kernel_kind() == capi.KIND_COLWAVE
    for i in (0, 1, 32767, 32768, 32769, n - 1):
        L, R = to_host(Ls[i]), to_host(Rs[i])
        assert np.array_equal(to_host(out[i]), _expect(L, R, block, D)), i
    # every frame against the generic kernel (independent second implementation), on the device
    gen = torch.empty_like(Ls)
    ctx.disparity_dev(Ls, Rs, gen, block, D, capi.FLAG_FORCE_GENERIC)
    torch.cuda.synchronize()
    assert torch.equal(out, gen)


# ---- SumAbsoluteDifferences (sad.go:205-244), incl. the independent right clamp ----------------
def test_sad_windows_kats_and_clamps(ctx, hwfx):
    wl = np.ascontiguousarray(hwfx["winL"].reshape(15, 15)); wr = np.ascontiguousarray(hwfx["winR"].reshape(15, 15))
    assert int(ctx.sad_windows(wl, wr, 15, [(7, 7, 7, 7)])[0]) == 18116
    for k in range(4):
        L, R = np.ascontiguousarray(hwfx["strips_old_L"][k]), np.ascontiguousarray(hwfx["strips_old_R"][k])
        assert int(ctx.sad_windows(L, R, 15, [(7, 7, 7, 7)])[0]) == int(hwfx["exp_sad_old"][k])
    L, R = sha.synth.synth_pair(5, 60, 40)
    Rs = np.ascontiguousarray(R[:33, :47])                    # smaller right image: early breaks
    rng = np.random.default_rng(1)
    q = np.stack([rng.integers(0, 60, 400), rng.integers(0, 40, 400),
                  rng.integers(-20, 70, 400), rng.integers(-10, 50, 400)], axis=1).astype(np.int32)
    for block in (1, 7, 16, 31):
        got = ctx.sad_windows(L, Rs, block, q)
        exp = [oracle.sum_absolute_differences(L, Rs, *map(int, r), block) for r in q]
        assert got.tolist() == exp
    assert sha.despair.SumAbsoluteDifferences(L, R, 30, 20, 25, 20, 9) == oracle.sum_absolute_differences(L, R, 30, 20, 25, 20, 9)


# ---- gray conversion (gray.go:43-58 / output.go:142-162) ----------------------------------------
def test_gray_and_fused_rgba(ctx, testdata_rgba, testdata_pair):
    la, ra = testdata_rgba
    assert np.array_equal(ctx.gray(la), testdata_pair[0])
    assert np.array_equal(ctx.gray(ra), testdata_pair[1])
    rng = np.random.default_rng(3)
    for shape in ((7, 13, 4), (33, 65, 3), (1, 1, 4), (20, 1027, 4)):
        pix = rng.integers(0, 256, shape, dtype=np.uint8)
        exp = oracle.gray_from_nrgba(pix) if shape[2] == 4 else oracle.gray_from_rgb8(pix)
        assert np.array_equal(ctx.gray(pix), exp)
    got = ctx.disparity_rgba(la, ra, 7, 32)
    assert hashlib.sha256(got.tobytes()).hexdigest() == "03a658082e809b4211a1216839d23a7123f1187b56f18dff0b21b6863c8a9d6e"
    got = sha.despair.RunSadPixels(la, ra, 9, 64)              # the Python mirror's one-call path (f-3)
    assert hashlib.sha256(got.tobytes()).hexdigest() == "0bde916590100d42d2fb6c0aca7081ba4f912c2b03a8f3bb5a8d9946454317fb"
    sha.despair.SetDefaultParams(sha.despair.Parameters(BlockSize=16, MaxDisparity=64))
    # SADHIP_FLAG_HOST_GRAY (ABI 5): the copy threads convert while they stage -- same map, bit for bit, as the device conversion
    got = ctx.disparity_rgba(la, ra, 7, 32, capi.FLAG_HOST_GRAY)
    assert hashlib.sha256(got.tobytes()).hexdigest() == "03a658082e809b4211a1216839d23a7123f1187b56f18dff0b21b6863c8a9d6e"
    assert capi.last_kernel_kind() == capi.KIND_COLWAVE
    sha.despair.SetHostGray(True)                              # the shims' switch (Go: despair.SetHostGray): RunSadImages / RunSadPixels pass the flag
    try:
        got = sha.despair.RunSadImages(la, ra, sha.despair.Parameters(BlockSize=9, MaxDisparity=64))
        assert hashlib.sha256(got.tobytes()).hexdigest() == "0bde916590100d42d2fb6c0aca7081ba4f912c2b03a8f3bb5a8d9946454317fb"
    finally:
        sha.despair.SetHostGray(False)
    with pytest.raises(capi.SadHipError) as ei:                 # the flag belongs to the picture entry point only
        ctx.disparity(testdata_pair[0], testdata_pair[1], 7, 32, capi.FLAG_HOST_GRAY)
    assert ei.value.status == capi.ERR_ARG


def test_host_gray_flag_every_alpha_class_and_layout(ctx):
    """SADHIP_FLAG_HOST_GRAY of sadhip_disparity_rgba_u8: luminance on the host (host/gray_rows.h, vector code with an opaque
    fast path per 8 pixels, a general-alpha path, RGB, scalar row ends) must equal oracle.gray_from_nrgba / gray_from_rgb8 bit
    for bit, through the whole call: maps equal the oracle's on its gray planes and the device-gray call's.  Opaque, alpha in
    251..255 (the reference's testdata), sparse and arbitrary alpha; widths around the vector width; RGB; row strides of a
    sub-picture; raw output; a 1080p pair large enough for the copy threads to split."""
    rng = np.random.default_rng(77)
    cases = [(40, 64, 4, "opaque"), (37, 67, 4, "near"), (33, 131, 4, "sparse"), (25, 95, 4, "any"), (41, 70, 3, "rgb"), (30, 9, 4, "any"),
             (30, 9, 3, "rgb"), (1080, 1920, 4, "near"), (1080, 1920, 3, "rgb")]
    for h, w, c, kind in cases:
        la = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
        s = 5
        ra = np.roll(la, -s, axis=1).copy()
        ra[..., :3] ^= (rng.integers(0, 16, (h, w, 1)) == 0).astype(np.uint8)
        if c == 4:
            for a in (la, ra):
                if kind == "opaque": a[..., 3] = 255
                elif kind == "near": a[..., 3] = rng.integers(251, 256, (h, w))
                elif kind == "sparse": a[..., 3] = np.where(rng.integers(0, 9, (h, w)) == 0, rng.integers(0, 256, (h, w)), 255)
        gl, gr = ((oracle.gray_from_nrgba(a) if c == 4 else oracle.gray_from_rgb8(a)) for a in (la, ra))
        B, D = (9, 64) if w >= 64 + 11 else (3, 4)
        flags = capi.FLAG_RAW if h < 1000 else 0
        got = ctx.disparity_rgba(la, ra, B, D, flags | capi.FLAG_HOST_GRAY)
        dev = ctx.disparity_rgba(la, ra, B, D, flags)
        assert np.array_equal(got, dev), (h, w, c, kind)
        if h < 1000:
            assert np.array_equal(got, _expect(gl, gr, B, D, raw=True)), (h, w, c, kind)
        else:
            assert np.array_equal(got, ctx.disparity(gl, gr, B, D)), (h, w, c, kind)
            rows = (0, 8), (536, 544), (1072, 1080)
            for (y0, y1), band in _literal_rows(gl, gr, B, D, rows).items():
                assert np.array_equal(got[y0:y1], band), (kind, y0)
    # a sub-picture: rows `stride` bytes apart inside a wider picture, nothing read outside the window's rows
    big = rng.integers(0, 256, (50, 200, 4), dtype=np.uint8)
    big[..., 3] = 255
    wl, wr = big[3:43, 10:150], big[5:45, 20:160]
    got = ctx.disparity_rgba(wl, wr, 7, 32, capi.FLAG_RAW | capi.FLAG_HOST_GRAY)
    assert np.array_equal(got, _expect(oracle.gray_from_nrgba(np.ascontiguousarray(wl)), oracle.gray_from_nrgba(np.ascontiguousarray(wr)), 7, 32, raw=True))


# ---- error behaviour at the boundary ------------------------------------------------------------
def test_error_codes(ctx):
    L = np.zeros((16, 16), np.uint8)
    for kw, status in (((L, L, 7, 0), capi.ERR_MAXDISP), ((L, L, 0, 32), capi.ERR_BLOCK),
                       ((L, L, 256, 32), capi.ERR_BLOCK), ((L, L, 7, 5000), capi.ERR_MAXDISP)):
        with pytest.raises(capi.SadHipError) as ei:
            ctx.disparity(*kw)
        assert ei.value.status == status
    with pytest.raises(capi.SadHipError) as ei:
        ctx.disparity_region(L, L, 7, 32, (0, 0, 17, 4))
    assert ei.value.status == capi.ERR_REGION
    with pytest.raises(capi.SadHipError) as ei:
        ctx.disparity_region(L, L, 7, 32, (4, 4, 4, 8))
    assert ei.value.status == capi.ERR_REGION
    small = capi.Context(0, 32, 32, 1)
    with pytest.raises(capi.SadHipError) as ei:
        small.disparity(np.zeros((40, 40), np.uint8), np.zeros((40, 40), np.uint8), 7, 32)
    assert ei.value.status == capi.ERR_DIMS
    small.close()
    with pytest.raises(capi.SadHipError) as ei:
        capi.Context(99, 32, 32, 1)
    assert ei.value.status == capi.ERR_NO_DEVICE


# ---- BASELINE full sizes: oracle on a sample + size-independent properties ----------------------
def _props(ctx, L, R, block, maxdisp, flags=0):
    h, w = L.shape
    hs = block // 2
    got = ctx.disparity(L, R, block, maxdisp, flags | capi.FLAG_RAW)
    sc = ctx.disparity(L, R, block, maxdisp, flags)
    assert np.array_equal(sc, (got.astype(np.int32) * 255 // maxdisp).astype(np.uint8))       # scaling law
    assert not got[:, :hs + 1].any()                                                          # x <= hs -> 0
    xs = np.arange(w)
    assert (got <= np.minimum(maxdisp, np.maximum(0, xs - hs))[None, :]).all()                # candidate range
    assert np.array_equal(got, ctx.disparity(L, R, block, maxdisp, flags | capi.FLAG_RAW))    # deterministic
    # tiling invariance: a band computed alone equals the same rows of the frame (halo-free chunks, sad.go:69-77)
    y0 = h // 3
    assert np.array_equal(ctx.disparity_region(L, R, block, maxdisp, (0, y0, w, y0 + 5), flags | capi.FLAG_RAW), got[y0:y0 + 5])
    # planted shift: the generator's true disparity wins wherever the window is full and un-perturbed enough
    s = 3 + (xs[None, :] * 7) // w + (np.arange(h)[:, None] * 5) // h
    inner = (slice(hs, h - hs), slice(maxdisp + hs, w - 16 - hs))
    assert (got[inner] == s[inner]).mean() > 0.8
    return got


def _literal_rows(L, R, block, maxdisp, rows):
    """Full-width rows of the LITERAL restatement of sad.go:55-94 (multithreaded), scaled output."""
    from concurrent.futures import ThreadPoolExecutor
    rows = list(rows)
    with ThreadPoolExecutor(min(16, os.cpu_count() or 1)) as ex:      # short ranges: one oracle thread each, ranges in parallel
        res = list(ex.map(lambda r: oracle.disparity_rows_mt(L, R, block, maxdisp, r[0], r[1], 1), rows))
    return dict(zip(rows, res))


def _seam_rows(w, h, block, maxdisp, n=1, span=4):
    """Row ranges across every REAL strip seam of a launch of n frames, from the library's own launch geometry
    (sadhip_plan_describe_batch: choose_strips() cuts a frame into strips of `rows` output rows -- 60 to 90 of them for
    a single 1080p frame, a handful for a long batch -- per pass), plus the frame's top and bottom.  Returns
    (merged row ranges, passes as (strips, rows))."""
    geo = capi.plan_describe_batch(w, h, block, maxdisp, n)
    assert geo and all(g["device"] == 1 for g in geo), geo          # occupancy asked from the runtime of THIS device
    marks = {(0, span), (h - span, h)}
    for g in geo:
        assert (h + g["rows"] - 1) // g["rows"] == g["strips"]
        for k in range(1, g["strips"]):
            y = k * g["rows"]
            marks.add((max(0, y - span // 2), min(h, y + span // 2)))
    merged = []
    for ya, yb in sorted(marks):
        if merged and ya <= merged[-1][1]:
            merged[-1] = (merged[-1][0], max(merged[-1][1], yb))
        else:
            merged.append((ya, yb))
    return merged, [(g["strips"], g["rows"]) for g in geo]


def test_config3_1080p_b9_d64(ctx):
    L, R = sha.synth.synth_pair(3, 1920, 1080)
    got = _props(ctx, L, R, 9, 64)
    sc = ctx.disparity(L, R, 9, 64)
    assert capi.last_kernel_kind() == capi.KIND_COLWAVE
    # the whole width on 2 rows either side of EVERY real strip seam of this launch, against the literal CPU restatement
    rows, passes = _seam_rows(1920, 1080, 9, 64)
    assert passes[0][0] >= 8 and sum(b - a for a, b in rows) >= 64, passes
    for (ya, yb), exp in _literal_rows(L, R, 9, 64, rows).items():
        assert np.array_equal(sc[ya:yb], exp), (ya, yb, passes)
    for (x0, y0, x1, y1) in ((0, 500, 120, 540), (1800, 300, 1920, 340), (900, 536, 1100, 546)):
        assert np.array_equal(got[y0:y1, x0:x1], oracle.disparity_closed(L, R, 9, 64, region=(x0, y0, x1, y1), raw=True))
    assert np.array_equal(got, ctx.disparity(L, R, 9, 64, capi.FLAG_FORCE_GENERIC | capi.FLAG_RAW))   # two independent kernels agree


def test_config5_4k_b15_d128(ctx):
    L, R = sha.synth.synth_pair(5, 3840, 2160)
    got = _props(ctx, L, R, 15, 128)
    for (x0, y0, x1, y1) in ((0, 0, 3840, 2), (0, 2158, 3840, 2160), (0, 1000, 200, 1012), (3700, 77, 3840, 89), (1900, 1075, 2000, 1085)):
        assert np.array_equal(got[y0:y1, x0:x1], oracle.disparity_closed(L, R, 15, 128, region=(x0, y0, x1, y1), raw=True))
    band = (0, 700, 3840, 724)
    assert np.array_equal(got[700:724], ctx.disparity_region(L, R, 15, 128, band, capi.FLAG_FORCE_GENERIC | capi.FLAG_RAW))
    sc = ctx.disparity(L, R, 15, 128)
    assert capi.last_kernel_kind() == capi.KIND_COLWAVE
    rows, passes = _seam_rows(3840, 2160, 15, 128)            # 2 rows either side of every real strip seam, literal oracle
    assert sum(b - a for a, b in rows) >= 64, passes
    for (ya, yb), exp in _literal_rows(L, R, 15, 128, rows).items():
        assert np.array_equal(sc[ya:yb], exp), (ya, yb, passes)


def test_batched_layout_complete_frames_vs_literal_oracle(ctx):
    """The batched layout (blockIdx.x = strip + strips * frame, few tall strips) against the LITERAL restatement on
    COMPLETE frames: 128 device-resident 1080p pairs (config 4's per-GPU shard on 8 GPUs) in one launch; the first,
    the last and one middle frame compared in full, every other frame on the rows around this launch's strip seams'
    first seam, and the plan of this batch length is what the seam rows come from (bench.py checks 17 frames of its
    1024-pair launches the same way, outside the test suite)."""
    import torch
    n, W, H, B, D = 128, 1920, 1080, 9, 64
    seeds = [1000 + 8 * i for i in range(n)]                       # rank 0's shard of config 4 on 8 GPUs
    parts = [sha.synth.synth_batch_torch(seeds[c:c + 32], W, H, "cuda:0") for c in range(0, n, 32)]
    Ls, Rs = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts]); del parts
    out = torch.empty_like(Ls)
    ctx.disparity_dev(Ls, Rs, out, B, D)
    torch.cuda.synchronize()
    assert capi.last_kernel_kind() == capi.KIND_COLWAVE
    rows, passes = _seam_rows(W, H, B, D, n=n)
    thr = min(16, os.cpu_count() or 1)
    for i in (0, 61, n - 1):                                       # complete frames
        L, R = to_host(Ls[i]), to_host(Rs[i])
        assert np.array_equal(to_host(out[i]), oracle.disparity_rows_mt(L, R, B, D, 0, H, thr)), (i, passes)
    for i in range(1, n - 1, 9):                                   # the others: around every seam of this launch
        L, R, o = to_host(Ls[i]), to_host(Rs[i]), to_host(out[i])
        for ya, yb in rows[:6]:
            assert np.array_equal(o[ya:yb], oracle.disparity_rows_mt(L, R, B, D, ya, yb, thr)), (i, ya, yb, passes)


def test_small_single_frames_run_short_strips_and_stay_exact(ctx):
    """Round 5: a launch that even 4-row strips cannot spread over all workgroup slots of its kernel gets strips down to 4 rows
    (kernels_fast.hip; one 640x480 pair -- the reference's own frame size -- at its defaults 16 / 64 was 120 workgroups on 120 of
    256 CUs with the 8-row floor).  Complete frames against the oracle for the sizes and parameters where the rule acts, the
    geometry the library reports (rows per strip < 8 there, >= 8 for a launch that fills the chip), and the same frames as part
    of a batch (which keeps the taller strips)."""
    import torch
    short = 0
    for (w, h) in ((640, 480), (320, 240), (200, 100)):
        for (b, d) in ((16, 64), (7, 32), (9, 64), (3, 16), (21, 96), (31, 64), (15, 128)):
            if w <= d + 2 * (b // 2) + 2 and b < 6:
                continue
            L, R = sha.synth.synth_pair(9000 + w + b + d, w, h)
            exp = _expect(L, R, b, d)
            got = ctx.disparity(L, R, b, d)
            assert np.array_equal(got, exp), (w, h, b, d)
            if capi.last_kernel_kind() != capi.KIND_COLWAVE:
                continue
            geo = capi.plan_describe_batch(w, h, b, d, 1)
            assert all(g["device"] == 1 and g["rows"] >= 4 for g in geo), geo
            short += any(g["rows"] < 8 for g in geo)
            Ls, Rs = to_device(np.stack([L] * 3)), to_device(np.stack([R] * 3))
            out = torch.empty_like(Ls)
            ctx.disparity_dev(Ls, Rs, out, b, d)
            torch.cuda.synchronize()
            assert np.array_equal(to_host(out[1]), exp), (w, h, b, d, "batch of 3")
    assert short >= 8, short                                    # the rule acted (640x480 at 16 / 64 and 7 / 32 among them)
    big = capi.plan_describe_batch(1920, 1080, 9, 64, 128)
    assert all(g["rows"] >= 8 for g in big), big


def test_config4_whole_stream_1024_pairs_one_launch(ctx):
    """BASELINE configs[3] at its stated length, the launch bench.py times: 1024 device-resident 1920x1080 pairs (frame i =
    seed 1000 + i, examples/run.stream.go:30-59's stream), ONE launch at 9 / 64.  Frames 0, 341, 682, 1023 complete against
    the literal restatement; EVERY frame against the generic kernel on the device (the independent second implementation);
    the launch's geometry is the 1024-pair plan's."""
    import torch
    n, W, H, B, D = 1024, 1920, 1080, 9, 64
    parts = [sha.synth.synth_batch_torch(list(range(1000 + c, 1000 + c + 32)), W, H, "cuda:0") for c in range(0, n, 32)]
    Ls, Rs = torch.cat([p[0] for p in parts]), torch.cat([p[1] for p in parts]); del parts
    out = torch.full_like(Ls, 0xCD)
    ctx.disparity_dev(Ls, Rs, out, B, D)
    torch.cuda.synchronize()
    assert capi.last_kernel_kind() == capi.KIND_COLWAVE
    geo = capi.plan_describe_batch(W, H, B, D, n)
    assert len(geo) == 1 and geo[0]["tiles"] == 8 and geo[0]["device"] == 1, geo
    thr = min(16, os.cpu_count() or 1)
    for i in (0, 341, 682, 1023):
        L, R = to_host(Ls[i]), to_host(Rs[i])
        if i in (0, 1023):                                         # the device generator is the host generator
            hl, hr = sha.synth.synth_pair(1000 + i, W, H)
            assert np.array_equal(L, hl) and np.array_equal(R, hr), i
        assert np.array_equal(to_host(out[i]), oracle.disparity_rows_mt(L, R, B, D, 0, H, thr)), (i, geo)
    gen = torch.full_like(Ls[:128], 0xEE)
    for c in range(0, n, 128):                                     # 128 frames of the generic kernel at a time (0.55 s each)
        ctx.disparity_dev(Ls[c:c + 128], Rs[c:c + 128], gen, B, D, capi.FLAG_FORCE_GENERIC)
        torch.cuda.synchronize()
        assert capi.last_kernel_kind() == capi.KIND_GENERIC
        same = (out[c:c + 128] == gen).flatten(1).all(1)
        assert bool(same.all()), (c, torch.nonzero(~same).flatten().tolist()[:8])


def test_config5_complete_4k_frame_vs_literal_oracle(ctx):
    """BASELINE configs[4]: ONE COMPLETE 3840x2160 frame at 15 / 128 against the literal restatement (2.4e11 byte
    differences: a few seconds on the box's 16 host threads), single call and as frame 1 of a three-frame
    device-resident batch (the batched launch geometry)."""
    import torch
    W, H, B, D = 3840, 2160, 15, 128
    L, R = sha.synth.synth_pair(5, W, H)
    exp = oracle.disparity_rows_mt(L, R, B, D, 0, H, min(16, os.cpu_count() or 1))
    got = ctx.disparity(L, R, B, D)
    assert capi.last_kernel_kind() == capi.KIND_COLWAVE
    assert np.array_equal(got, exp), np.argwhere(got != exp)[:4]
    L2, R2 = sha.synth.synth_pair(6, W, H)
    Ls, Rs = to_device(np.stack([L2, L, L2])), to_device(np.stack([R2, R, R2]))
    out = torch.full_like(Ls, 0xCD)
    ctx.disparity_dev(Ls, Rs, out, B, D)
    torch.cuda.synchronize()
    o = to_host(out)
    assert np.array_equal(o[1], exp) and np.array_equal(o[0], o[2])
    assert np.array_equal(o[0], ctx.disparity(L2, R2, B, D))


def test_config4_stream_shard(ctx):
    """A shard of the 1024-frame stream (frames i = rank, rank+G, ...): device-resident batch,
    every frame checked against a band of the oracle + a checksum of checksums across two runs."""
    import torch
    seeds = [1000 + i for i in (0, 8, 16, 1016)]
    Ls, Rs = sha.synth.synth_batch_torch(seeds, 1920, 1080, "cuda:0")
    out = torch.empty_like(Ls)
    ctx.disparity_dev(Ls, Rs, out, 9, 64)
    torch.cuda.synchronize()
    o = to_host(out)
    d1 = hashlib.sha256(b"".join(hashlib.sha256(o[i].tobytes()).digest() for i in range(len(seeds)))).hexdigest()
    for i, sd in enumerate(seeds):
        L, R = sha.synth.synth_pair(sd, 1920, 1080)
        assert np.array_equal(o[i, 300:304], oracle.disparity_closed(L, R, 9, 64, region=(0, 300, 1920, 304)))
        assert np.array_equal(o[i], ctx.disparity(L, R, 9, 64))          # batch == single-frame host path
    out.zero_()
    ctx.disparity_dev(Ls, Rs, out, 9, 64)
    torch.cuda.synchronize()
    o2 = to_host(out)
    d2 = hashlib.sha256(b"".join(hashlib.sha256(o2[i].tobytes()).digest() for i in range(len(seeds)))).hexdigest()
    assert d1 == d2
