"""Python mirror of the Go API surface of pkg/despair, running on the HIP library.

Same names, argument meaning and ownership rules as the reference
(pkg/despair/sad.go, params.go); used by tests and examples so they read like reference
callers (examples/run.go, examples/run.stream.go, pkg/camera/output.go).  The C++ mirror and
the cgo shim (host/) are the artefacts a Go maintainer consumes; this module is their ctypes
twin.  Images are 2-D uint8 numpy arrays (image.Gray with Rect.Min == (0,0)); a rectangle is
(x0, y0, x1, y1) like image.Rect.

Differences from the reference, all deliberate:
  * AssembleDisparityMap writes every chunk it receives (the reference drops the last one,
    sad.go:181-184);
  * MaxDisparity < 1 raises instead of panicking with a divide by zero (sad.go:92);
  * a device error inside the worker is carried to the consumer: AssembleDisparityMap raises it
    instead of returning a partly black map (the reference has no error path at all);
  * there is no CPU path here: without the HIP library / a GPU every call raises.
The device is chosen once per process from the environment variable SADHIP_DEVICE (default 0;
the reference has no configuration precedent, SURVEY section 5 "config").
"""
import contextlib
import os
import queue
import threading
from dataclasses import dataclass

import numpy as np

from . import capi


@dataclass
class Parameters:
    """params.go:32-37"""
    BlockSize: int = 16
    MaxDisparity: int = 64


_default_params = Parameters(BlockSize=16, MaxDisparity=64)   # params.go:13-18
_default_params_mu = threading.Lock()


def SetDefaultParams(params):
    """params.go:21-25"""
    global _default_params
    with _default_params_mu:
        _default_params = Parameters(params.BlockSize, params.MaxDisparity)


def DefaultParams():
    """params.go:28-30"""
    return _default_params


@dataclass
class InputChunk:
    """sad.go:11-15"""
    Left: np.ndarray
    Right: np.ndarray
    Region: tuple


@dataclass
class OutputChunk:
    """sad.go:17-21"""
    DisparityData: np.ndarray   # flat uint8, row-major within Region
    Region: tuple


_CLOSED = object()


class Chan:
    """Minimal Go-channel stand-in: send / iterate / close.  ``error`` is the pipeline's error slot:
    the worker stores its exception there before closing the output channel."""

    def __init__(self, cap):
        self._q = queue.Queue(maxsize=cap)
        self.error = None

    def send(self, v):
        self._q.put(v)

    def close(self):
        self._q.put(_CLOSED)

    def try_recv(self):
        """What is queued right now, without waiting: (True, value), or (False, None) when nothing is (a close marker stays
        queued for the iterating consumer)."""
        try:
            v = self._q.get_nowait()
        except queue.Empty:
            return False, None
        if v is _CLOSED:
            self._q.put(_CLOSED)
            return False, None
        return True, v

    def __iter__(self):
        while True:
            v = self._q.get()
            if v is _CLOSED:
                self._q.put(_CLOSED)   # let other consumers see the close too
                return
            yield v


_ctx_lock = threading.Lock()
_ctx = None
_ctx_cap = (0, 0)
_DEVICE = int(os.environ.get("SADHIP_DEVICE", "0"))          # read once (SURVEY section 5 "config")


@contextlib.contextmanager
def _context(w, h):
    """Process-wide device context, grown on demand (the worker pool of sad.go:41-104).
    The lock is held for the WHOLE call that uses the context: a concurrent caller with a larger
    frame can then never destroy it while another thread is inside sadhip_* with the old handle
    (OutputCamera and RunSad callers are concurrent in the reference: output.go:37, examples/run.go:30).
    Calls on one context are serialised inside the library anyway."""
    global _ctx, _ctx_cap
    with _ctx_lock:
        if _ctx is None or w > _ctx_cap[0] or h > _ctx_cap[1]:
            if _ctx is not None:
                _ctx.close()
                _ctx = None
            _ctx_cap = (max(w, _ctx_cap[0], 1920), max(h, _ctx_cap[1], 1080))
            _ctx = capi.Context(_DEVICE, _ctx_cap[0], _ctx_cap[1], 1)
        yield _ctx


def SetupConcurrentSAD(numWorkers):
    """sad.go:29-113.  Returns (inputChan, outputChan).  One device worker replaces the
    goroutine pool; parameters are snapshotted per group of chunks that were queued together (sad.go:51-53: the
    reference's workers would each take theirs within the same microseconds); closing the input
    channel stops the worker, which then closes the output channel (sad.go:106-110).
    On a device error the worker records it in outputChan.error, closes the output channel and
    keeps draining the input channel, so a feeder blocked in send() is released and
    AssembleDisparityMap raises instead of returning a short map."""
    if numWorkers <= 0:
        numWorkers = 32
    # sad.go:36-37: 2 x numWorkers slots each, plus one chunk held by every goroutine that waits to send = 5 x numWorkers
    # chunks under way before anyone reads -- examples/run.stream.go:46-58 feeds a whole frame (160 chunks for 640x480 with
    # 32 workers) before it reads.  One device worker holds no crowd: the output channel gets the goroutines' share.
    inp, outp = Chan(numWorkers * 2), Chan(numWorkers * 3)

    MAX_GROUP = 512                                            # chunks answered by one device call

    def worker():
        failed = False
        pending = None
        it = iter(inp)
        try:
            while True:
                if pending is not None:
                    chunk, pending = pending, None
                else:
                    try:
                        chunk = next(it)
                    except StopIteration:
                        break
                if failed:
                    continue                                   # drain: never leave the feeder blocked
                # the chunks already queued behind this one that continue it downwards (same images, same columns):
                # one device call over their common rows, one answer per chunk (host/despair.hpp has the argument)
                group = [chunk]
                while len(group) < MAX_GROUP:
                    ok, nxt = inp.try_recv()
                    if not ok:
                        break
                    f, l = group[0], group[-1]
                    if (nxt.Left is f.Left and nxt.Right is f.Right and nxt.Region[0] == f.Region[0] and nxt.Region[2] == f.Region[2]
                            and nxt.Region[1] == l.Region[3] and nxt.Region[3] > nxt.Region[1] and l.Region[3] > l.Region[1]):
                        group.append(nxt)
                    else:
                        pending = nxt                          # starts the next group
                        break
                with _default_params_mu:
                    p = Parameters(_default_params.BlockSize, _default_params.MaxDisparity)   # sad.go:51-53, once per group
                h, w = chunk.Left.shape
                x0, y0, x1 = group[0].Region[0], group[0].Region[1], group[0].Region[2]
                allr = (x0, y0, x1, group[-1].Region[3])
                try:
                    if allr[2] <= allr[0] or allr[3] <= allr[1]:
                        data = np.zeros((max(allr[3] - allr[1], 0), max(allr[2] - allr[0], 0)), np.uint8)   # sad.go:48-50: an empty region
                    else:
                        with _context(w, h) as c:
                            if chunk.Right.shape == chunk.Left.shape:
                                data = c.disparity_region(chunk.Left, chunk.Right, p.BlockSize, p.MaxDisparity, allr)
                            else:   # independently decoded pictures (output.go:134-162): the reference's clamps, sad.go:212-233
                                data = c.disparity_lr(chunk.Left, chunk.Right, p.BlockSize, p.MaxDisparity, allr)
                except Exception as e:                          # noqa: BLE001 -- carried to the consumer
                    outp.error = e
                    failed = True
                    outp.close()
                    continue
                for g in group:
                    rows = data[g.Region[1] - y0:g.Region[3] - y0]
                    outp.send(OutputChunk(DisparityData=np.ascontiguousarray(rows).reshape(-1), Region=g.Region))
        finally:
            if not failed:
                outp.close()

    threading.Thread(target=worker, daemon=True).start()
    return inp, outp


def AssembleDisparityMap(outputChan, dimensions, chunks):
    """sad.go:172-202, without the dropped-chunk bug: all ``chunks`` chunks are written."""
    x0, y0, x1, y1 = dimensions
    disparityMap = np.zeros((y1 - y0, x1 - x0), np.uint8)
    i = 0
    for chunk in outputChan:
        cx0, cy0, cx1, cy1 = chunk.Region
        disparityMap[cy0 - y0:cy1 - y0, cx0 - x0:cx1 - x0] = chunk.DisparityData.reshape(cy1 - cy0, cx1 - cx0)
        i += 1
        if i >= chunks:
            break
    if i < chunks:                                             # the stream ended early: never a silent black band
        raise outputChan.error if getattr(outputChan, "error", None) is not None else \
            RuntimeError(f"AssembleDisparityMap: output channel closed after {i} of {chunks} chunks")
    return disparityMap


def RunSad(left, right, blockSize, maxDisparity):
    """sad.go:119-169: sets the global parameters (:123-126) and returns the full map.
    One frame-level device call replaces tiling + channel assembly (values do not depend on
    the tiling: every worker reads the full shared images, sad.go:69-77).  The map has the LEFT image's
    size (sad.go:131,168); a right image of another size is handled as the reference's loops do
    (sadhip_disparity_lr_u8: sad.go:212-218,225-227,231-233), never read past its end."""
    SetDefaultParams(Parameters(BlockSize=blockSize, MaxDisparity=maxDisparity))
    return RunSadFrame(left, right, Parameters(BlockSize=blockSize, MaxDisparity=maxDisparity))


def RunSadFrame(left, right, p):
    """The Go shim's RunSadFrame (host/go/sad_hip.go): RunSad WITHOUT the write to the process-wide parameters --
    what a frame loop calls that only reads them (pkg/camera/output.go:169)."""
    h, w = left.shape
    with _context(w, h) as c:
        if right.shape != left.shape:
            return c.disparity_lr(left, right, p.BlockSize, p.MaxDisparity)
        return c.disparity(left, right, p.BlockSize, p.MaxDisparity)


_host_gray = 0


def SetHostGray(on):
    """The Go shim's SetHostGray (libsadhip ABI 5, SADHIP_FLAG_HOST_GRAY): where RunSadImages / RunSadPixels convert two
    colour pictures of one size to gray -- False (default): uploaded as they are, converted on the device; True: libsadhip's
    copy threads write luminance rows while they stage the pictures (same integer formula, bit for bit)."""
    global _host_gray
    _host_gray = capi.FLAG_HOST_GRAY if on else 0


def RunSadImages(left_pix, right_pix, p):
    """The Go shim's RunSadImages: two DECODED pictures in (HxW uint8 = *image.Gray, HxWx4 = *image.NRGBA, HxWx3 =
    opaque RGB, which Go holds as *image.RGBA with alpha 255), disparity map out, without the per-pixel
    color.GrayModel.Convert loops of pkg/camera/output.go:142-147,157-162 on the host (patches/0003): two colour
    pictures of one size and channel count are ONE sadhip_disparity_rgba_u8 call (gray on the device); gray pictures
    go as they are; pictures of different sizes or channel counts are converted one by one (sadhip_gray_u8) and sent
    through the lr entry, each with its own size.  Only READS p."""
    if left_pix.ndim == 3 and right_pix.ndim == 3 and left_pix.shape == right_pix.shape:
        h, w = left_pix.shape[:2]
        with _context(w, h) as c:
            return c.disparity_rgba(left_pix, right_pix, p.BlockSize, p.MaxDisparity, _host_gray)

    def gray_of(pix):
        if pix.ndim == 2:
            return pix
        with _context(pix.shape[1], pix.shape[0]) as c:
            return c.gray(pix)
    return RunSadFrame(gray_of(left_pix), gray_of(right_pix), p)


def RunSadPixels(left_pix, right_pix, blockSize, maxDisparity):
    """The in-memory handoff of SURVEY 8 f-3 for decoded RGB(A) pictures (HxWx3 or HxWx4 uint8, as
    png.Decode leaves an *image.NRGBA / opaque RGB): one sadhip_disparity_rgba_u8 call -- one upload per
    image, the Go luminance (gray.go:43-58) on the device, no host gray plane -- instead of LoadPNG's gray
    conversion followed by RunSad's second upload.  Sets the global parameters like RunSad (sad.go:123-126)."""
    SetDefaultParams(Parameters(BlockSize=blockSize, MaxDisparity=maxDisparity))
    h, w = left_pix.shape[:2]
    with _context(w, h) as c:
        return c.disparity_rgba(left_pix, right_pix, blockSize, maxDisparity, _host_gray)


def SumAbsoluteDifferences(left, right, leftX, leftY, rightX, rightY, blockSize):
    """sad.go:205-244 (one query through the batched device entry point)."""
    h, w = left.shape
    with _context(max(w, right.shape[1]), max(h, right.shape[0])) as c:
        return int(c.sad_windows(left, right, blockSize, [(leftX, leftY, rightX, rightY)])[0])


def LoadPNG(filename, refbug_rgba=False):
    """png.go:10-41: decode, then the reference's type switch -- a gray PNG is copied
    (gray.go:15-17), everything else goes through the Go luminance formula on alpha-premultiplied
    channels (gray.go:43-58), computed on the device.  An opaque 8-bit RGB file decodes to
    *image.RGBA in Go and hits convertRGBAToGray, which returns all zeros (gray.go:20-40);
    that is reproduced only with refbug_rgba=True.  16-bit files: use the C++ mirror."""
    from PIL import Image
    im = Image.open(filename)
    if im.mode in ("1", "L") and "transparency" not in im.info:
        return np.array(im.convert("L"))
    if im.mode == "RGB" and "transparency" not in im.info:
        pix = np.array(im)
        if refbug_rgba:
            return np.zeros(pix.shape[:2], np.uint8)
    elif im.mode in ("RGBA", "LA", "P", "L", "RGB", "1"):
        pix = np.array(im.convert("RGBA"))        # non-premultiplied, like *image.NRGBA / the palette entries
    else:
        raise ValueError(f"LoadPNG {filename}: unsupported PNG mode {im.mode}")
    h, w = pix.shape[:2]
    with _context(w, h) as c:
        return c.gray(pix)


def MustLoadPNG(filename):
    return LoadPNG(filename)


def SavePNG(filename, img):
    """png.go:54-66: 8-bit gray, fastest compression (png.BestSpeed)."""
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(img, np.uint8), "L").save(filename, format="PNG", compress_level=1)


def MustSavePNG(filename, img):
    SavePNG(filename, img)


def run_sad_tiles(w, h, num_cpu):
    """The tile rectangles RunSad feeds (sad.go:128-153), for callers that mirror its chunking."""
    import math
    numChunks = num_cpu * 4 * 4
    chunkWidth = int(math.sqrt((w * h) // numChunks))
    horChunks = max(1, w // chunkWidth) if chunkWidth > 0 else 1
    verChunks = max(1, numChunks // horChunks)
    chunkWidth = w // horChunks
    chunkHeight = h // verChunks
    rects = []
    for sy in range(0, h, max(chunkHeight, 1)):
        for sx in range(0, w, max(chunkWidth, 1)):
            rects.append((sx, sy, min(sx + chunkWidth, w), min(sy + chunkHeight, h)))
    return rects
