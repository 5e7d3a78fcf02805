"""PNG decode/encode around the path: the C++ codec of the host mirror
against Pillow on every colour type, and -- on the GPU box -- the examples/run.go counterpart end
to end against the digests SURVEY.md section 8c pins for the reference's own stereo pair."""
import hashlib
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "steroscopic-hardware_amd", "lib")
TOOL = os.path.join(LIB, "png_tool")
RUN = os.path.join(LIB, "run")
GOLDEN = os.path.join(REPO, "tests", "golden")
KIND = {"Gray": 0, "Gray16": 1, "RGBA": 2, "NRGBA": 3, "RGBA64": 4, "NRGBA64": 5, "Paletted": 6}   # png_io.hpp GoImage


def _decode(path, tmp_path):
    raw = tmp_path / "dec.raw"
    r = subprocess.run([TOOL, "decode", str(path), str(raw)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    w, h, c, d16, kind = (int(v) for v in r.stdout.split())
    a = np.fromfile(raw, np.uint8)
    return a.reshape(h, w, c * (2 if d16 else 1)), kind


def test_codec_matches_pillow_on_every_colour_type(tmp_path):
    rng = np.random.default_rng(5)
    w, h = 37, 23
    rgba = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    cases = {
        "gray8": (Image.fromarray(rgba[..., 0], "L"), "Gray", lambda im: np.array(im)[..., None]),
        "gray1": (Image.fromarray((rgba[..., 0] > 127).astype(np.uint8) * 255, "L").convert("1"), "Gray",
                  lambda im: np.array(im.convert("L"))[..., None]),
        "rgb8": (Image.fromarray(rgba[..., :3], "RGB"), "RGBA", lambda im: np.array(im)),
        "rgba8": (Image.fromarray(rgba, "RGBA"), "NRGBA", lambda im: np.array(im)),
        "la8": (Image.fromarray(rgba[..., :2], "LA"), "NRGBA", lambda im: np.array(im.convert("RGBA"))),
        "pal8": (Image.fromarray(rgba[..., :3], "RGB").quantize(64), "Paletted", lambda im: np.array(im.convert("RGBA"))),
    }
    for name, (im, kind, expect) in cases.items():
        path = tmp_path / f"{name}.png"
        im.save(path)
        got, k = _decode(path, tmp_path)
        assert k == KIND[kind], name
        assert np.array_equal(got, expect(Image.open(path))), name
    g16 = rng.integers(0, 65536, size=(h, w), dtype=np.uint16)
    Image.fromarray(g16).save(tmp_path / "g16.png")                      # mode I;16 -> 16-bit gray PNG
    got, k = _decode(tmp_path / "g16.png", tmp_path)
    assert k == KIND["Gray16"] and np.array_equal(got[..., 0].astype(np.uint16) << 8 | got[..., 1], g16)


def test_golden_pair_decodes_like_pillow(tmp_path):
    for n in ("L_00001.png", "R_00001.png"):
        got, k = _decode(os.path.join(GOLDEN, n), tmp_path)
        assert k == KIND["NRGBA"]                                        # 8-bit RGBA: *image.NRGBA in Go
        assert np.array_equal(got, np.array(Image.open(os.path.join(GOLDEN, n))))


def test_encoder_round_trips_and_rejects_garbage(tmp_path):
    a = np.random.default_rng(6).integers(0, 256, size=(41, 67), dtype=np.uint8)
    a.tofile(tmp_path / "a.raw")
    r = subprocess.run([TOOL, "encode", str(tmp_path / "a.raw"), "67", "41", str(tmp_path / "a.png")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    im = Image.open(tmp_path / "a.png")
    assert im.mode == "L" and np.array_equal(np.array(im), a)
    got, k = _decode(tmp_path / "a.png", tmp_path)
    assert k == KIND["Gray"] and np.array_equal(got[..., 0], a)
    bad = bytearray(open(tmp_path / "a.png", "rb").read()); bad[40] ^= 0xff
    open(tmp_path / "bad.png", "wb").write(bad)
    r = subprocess.run([TOOL, "decode", str(tmp_path / "bad.png"), str(tmp_path / "x.raw")], capture_output=True, text=True)
    assert r.returncode != 0 and r.stderr.strip()


def test_decoder_survives_mutated_files(tmp_path):
    """Structure-aware mutations (scanline bytes, stream length, IHDR fields, PLTE/tRNS) with valid
    CRCs: the decoder must answer (decoded or a clean error), never crash or hang.  Runs the
    ASan/UBSan build of the codec when `make -C host asan` has produced it."""
    import random
    import struct
    import zlib
    tool = os.path.join(LIB, "png_tool_asan") if os.path.exists(os.path.join(LIB, "png_tool_asan")) else TOOL
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (13, 19, 4), dtype=np.uint8)
    seeds = []
    for i, im in enumerate((Image.fromarray(a, "RGBA"), Image.fromarray(a[..., 0], "L"), Image.fromarray(a[..., :3], "RGB").quantize(16),
                            Image.fromarray(a[..., :2], "LA"), Image.fromarray(rng.integers(0, 65536, (7, 9), dtype=np.uint16)))):
        im.save(tmp_path / f"s{i}.png")
        seeds.append((tmp_path / f"s{i}.png").read_bytes())

    def chunks(b):
        pos, out = 8, []
        while pos + 12 <= len(b):
            n = struct.unpack(">I", b[pos:pos + 4])[0]
            out.append([b[pos + 4:pos + 8], bytearray(b[pos + 8:pos + 8 + n])])
            pos += 12 + n
        return out

    random.seed(11)
    answered = decoded = 0
    for it in range(250):
        ch = chunks(seeds[it % len(seeds)])
        raw = bytearray(zlib.decompress(b"".join(bytes(d) for t, d in ch if t == b"IDAT")))
        ch = [c for c in ch if c[0] != b"IDAT"]
        m = random.random()
        if m < 0.35:
            for _ in range(random.randint(1, 6)):
                raw[random.randrange(len(raw))] = random.randrange(256)
        elif m < 0.5:
            raw = raw[:random.randrange(len(raw))] if random.random() < 0.5 else raw + bytes(random.randrange(256) for _ in range(20))
        elif m < 0.8:
            ch[0][1][random.randrange(13)] = random.choice([0, 1, 2, 3, 4, 6, 8, 16, 255, random.randrange(256)])
        else:
            for c in ch:
                if c[0] in (b"PLTE", b"tRNS") and len(c[1]):
                    c[1][:] = c[1][:random.randrange(len(c[1]))]
            ch.insert(1, [b"tRNS", bytearray(random.randrange(256) for _ in range(random.randint(0, 8)))])
        ch.insert(len(ch) - 1, [b"IDAT", bytearray(zlib.compress(bytes(raw), 1))])
        blob = bytearray(b"\x89PNG\r\n\x1a\n")
        for t, d in ch:
            blob += struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + bytes(d)))
        (tmp_path / "m.png").write_bytes(blob)
        r = subprocess.run([tool, "decode", str(tmp_path / "m.png"), str(tmp_path / "m.raw")], capture_output=True, text=True, timeout=30)
        assert r.returncode in (0, 1), (it, r.returncode, r.stderr[-500:])
        answered += 1
        decoded += r.returncode == 0
    assert answered == 250 and 20 < decoded < 230


@pytest.mark.gpu
def test_run_example_end_to_end(tmp_path, testdata_pair):
    """examples/run.go through the C++ mirror: LoadPNG (device gray) -> RunSad -> SavePNG."""
    import oracle
    L, R = testdata_pair
    out = tmp_path / "output.png"
    r = subprocess.run([RUN, os.path.join(GOLDEN, "L_00001.png"), os.path.join(GOLDEN, "R_00001.png"), str(out), "7", "32"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr + r.stdout
    got = np.array(Image.open(out))
    assert hashlib.sha256(got.tobytes()).hexdigest() == "03a658082e809b4211a1216839d23a7123f1187b56f18dff0b21b6863c8a9d6e"
    r = subprocess.run([RUN, os.path.join(GOLDEN, "L_00001.png"), os.path.join(GOLDEN, "R_00001.png"), str(out)],
                       capture_output=True, text=True, timeout=120)                     # run.go's own parameters: 16, 64
    assert r.returncode == 0, r.stderr + r.stdout
    assert np.array_equal(np.array(Image.open(out)), oracle.disparity_closed(L, R, 16, 64))


@pytest.mark.gpu
def test_python_mirror_load_save(tmp_path, testdata_pair):
    import steroscopic_hardware_amd as sha
    L, _ = testdata_pair
    g = sha.despair.LoadPNG(os.path.join(GOLDEN, "L_00001.png"))
    assert np.array_equal(g, L)
    sha.despair.SavePNG(str(tmp_path / "g.png"), g)
    assert np.array_equal(sha.despair.LoadPNG(str(tmp_path / "g.png")), L)          # gray PNG: copied
    rgb = np.array(Image.open(os.path.join(GOLDEN, "L_00001.png")).convert("RGB"))
    Image.fromarray(rgb, "RGB").save(tmp_path / "rgb.png")
    import oracle
    assert np.array_equal(sha.despair.LoadPNG(str(tmp_path / "rgb.png")), oracle.gray_from_rgb8(rgb))
    assert not sha.despair.LoadPNG(str(tmp_path / "rgb.png"), refbug_rgba=True).any()   # gray.go:20-40
