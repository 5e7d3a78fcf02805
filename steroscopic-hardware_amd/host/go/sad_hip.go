//go:build cgo && hip

// MI355X backend of pkg/despair.  New file, added next to sad.go after patches/0001 (which
// gives the reference's Go entry points the unexported names setupConcurrentSADGo / runSadGo /
// assembleDisparityMapGo; sad_cpu.go re-exports them for the default build).  This file exports
// the same identifiers -- SetupConcurrentSAD, RunSad, AssembleDisparityMap -- and routes the
// arithmetic to libsadhip.so (include/sadhip.h) through cgo.  Types (InputChunk, OutputChunk),
// SumAbsoluteDifferences, Parameters and the PNG helpers stay where they are, shared by both builds.
//
//	CGO_ENABLED=1 go build -tags hip .
//
// Device: environment variable SADHIP_DEVICE (HIP ordinal, default 0), read once.
// Fallback: any non-zero status logs a warning and runs the reference's own Go loop (the
// unexported functions of sad.go) -- there is no second copy of that loop here.  A frame-level call
// (RunSadFrame, RunSadImages) falls back with the parameters it was given (runSadFrameGo, sad_frame.go).
//
// NOTE: the build container has no Go toolchain (`go: command not found`), so this file is not
// compiled by the repository's tests; the identical logic is exercised through the C++ mirror
// (host/despair.cpp, incl. a ThreadSanitizer run) and the ctypes mirror (despair.py).
package despair

// Header and library locations come from the environment, so the file works wherever the two
// artefacts are installed:
//	CGO_CFLAGS="-I<repo>/include" CGO_LDFLAGS="-L<repo>/steroscopic-hardware_amd/lib" go build -tags hip .

/*
#cgo LDFLAGS: -lsadhip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
#include <stdlib.h>
#include "sadhip.h"
*/
import "C"

import (
	"fmt"
	"image"
	"log/slog"
	"os"
	"runtime"
	"strconv"
	"sync"
	"sync/atomic"
	"unsafe"
)

// FrameLevel: RunSad is one device call (see sad_cpu.go).
const FrameLevel = true

// hipCtx is the process-wide device context, grown on demand.  Callers hold the READ lock for the
// whole C call (with): growing takes the write lock, so a context is never destroyed while another
// goroutine is inside libsadhip with its handle (OutputCamera and RunSad callers are concurrent:
// pkg/camera/output.go:37, examples/run.go:30).
type hipCtx struct {
	mu     sync.RWMutex
	h      *C.sadhip_ctx
	w, ht  int
	device int
	once   sync.Once
}

var ctx hipCtx

func (c *hipCtx) init() {
	c.once.Do(func() {
		// the header this file was compiled against and the library the process loaded must be the same ABI
		if got := int(C.sadhip_abi_version()); got != int(C.SADHIP_ABI_VERSION) {
			panic(fmt.Sprintf("despair: libsadhip ABI %d loaded, built against %d", got, int(C.SADHIP_ABI_VERSION)))
		}
		if v, err := strconv.Atoi(os.Getenv("SADHIP_DEVICE")); err == nil && v >= 0 {
			c.device = v
		}
	})
}

// statusErr must run on the OS thread that made the failing call: sadhip_last_error is thread-local.
func statusErr(what string, st C.int) error {
	return fmt.Errorf("%s: %s (%s)", what, C.GoString(C.sadhip_status_string(st)), C.GoString(C.sadhip_last_error()))
}

// with runs fn with a context able to hold w x h frames.  The goroutine is pinned to its OS thread
// from the C call to the reading of the thread-local error detail.
func (c *hipCtx) with(w, h int, fn func(*C.sadhip_ctx) error) error {
	c.init()
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	for {
		c.mu.RLock()
		if c.h != nil && w <= c.w && h <= c.ht {
			err := fn(c.h)
			c.mu.RUnlock()
			return err
		}
		c.mu.RUnlock()
		c.mu.Lock() // waits for every call in flight
		if c.h == nil || w > c.w || h > c.ht {
			if c.h != nil {
				C.sadhip_ctx_destroy(c.h)
				c.h = nil
			}
			nw, nh := max(w, c.w, 1920), max(h, c.ht, 1080)
			var out *C.sadhip_ctx
			if st := C.sadhip_ctx_create(C.int(c.device), C.int(nw), C.int(nh), 1, &out); st != C.SADHIP_OK {
				err := statusErr("sadhip_ctx_create", st)
				c.mu.Unlock()
				return err
			}
			c.h, c.w, c.ht = out, nw, nh
		}
		c.mu.Unlock()
	}
}

// regionHIP computes one chunk on the device.  Go pointers are only borrowed for the call
// (libsadhip's host thread copies them into / out of page-locked planes of its own -- no device-side
// copy ever touches Go memory -- and has finished with them when it returns, also on errors), as
// cgo requires.
func regionHIP(left, right *image.Gray, p Parameters, r image.Rectangle) ([]uint8, error) {
	data := make([]uint8, r.Dx()*r.Dy())
	if len(data) == 0 {
		return data, nil
	}
	if len(left.Pix) == 0 || len(right.Pix) == 0 {
		return nil, fmt.Errorf("empty image") // the Go path handles it (it yields zeros)
	}
	// Each image travels with its OWN size: left.png and right.png are decoded independently
	// (pkg/camera/output.go:134-162) and the reference clamps per image (sad.go:212-218) and breaks where
	// the right window leaves right.Rect.Max (sad.go:225-227,231-233).  Passing the left image's
	// dimensions for both would read past the end of a smaller right.Pix.
	err := ctx.with(left.Rect.Dx(), left.Rect.Dy(), func(h *C.sadhip_ctx) error {
		st := C.sadhip_disparity_lr_u8(h,
			(*C.uint8_t)(unsafe.Pointer(&left.Pix[0])), C.int(left.Rect.Dx()), C.int(left.Rect.Dy()), C.int(left.Stride),
			(*C.uint8_t)(unsafe.Pointer(&right.Pix[0])), C.int(right.Rect.Dx()), C.int(right.Rect.Dy()), C.int(right.Stride),
			C.int(p.BlockSize), C.int(p.MaxDisparity), 0,
			C.int(r.Min.X), C.int(r.Min.Y), C.int(r.Max.X), C.int(r.Max.Y),
			(*C.uint8_t)(unsafe.Pointer(&data[0])), C.int(r.Dx()))
		runtime.KeepAlive(left)
		runtime.KeepAlive(right)
		if st != C.SADHIP_OK {
			return statusErr("sadhip_disparity_lr_u8", st)
		}
		return nil
	})
	return data, err
}

// goFallback is a lazily started instance of the reference's own worker pool (sad.go:29-113),
// used for chunks of the channel API the device refused: like every chunk of that API it runs with the
// process-wide parameters, snapshotted by the worker itself (sad.go:51-53).
var goFallback struct {
	once sync.Once
	mu   sync.Mutex
	in   chan<- InputChunk
	out  <-chan OutputChunk
}

func regionGo(chunk InputChunk) OutputChunk {
	goFallback.once.Do(func() { goFallback.in, goFallback.out = setupConcurrentSADGo(1) })
	goFallback.mu.Lock() // one chunk in flight: the answer that comes back is this chunk's
	defer goFallback.mu.Unlock()
	goFallback.in <- chunk
	return <-goFallback.out
}

// maxGroup is the number of chunks one device call answers at most.
const maxGroup = 512

// continues reports whether next carries on where last stopped: same two images, same columns, the rows right below
// (the row bands every caller of the reference feeds: examples/run.stream.go:46-58, sad.go:138-153 per tile row).
func continues(first, last, next InputChunk) bool {
	return next.Left == first.Left && next.Right == first.Right &&
		next.Region.Min.X == first.Region.Min.X && next.Region.Max.X == first.Region.Max.X &&
		next.Region.Min.Y == last.Region.Max.Y && next.Region.Dy() > 0 && last.Region.Dy() > 0
}

// SetupConcurrentSAD sets up a concurrent SAD processing pipeline (sad.go:29-113): same channels,
// the same number of chunks under way; one device worker replaces the goroutine pool.  Chunks that are queued TOGETHER and
// continue each other downwards are computed by ONE device call over their common rows and answered
// chunk by chunk, with one parameter snapshot for the group -- the reference's workers would each have
// taken theirs within the same microseconds (sad.go:51-53).
func SetupConcurrentSAD(numWorkers int) (chan<- InputChunk, <-chan OutputChunk) {
	if numWorkers <= 0 {
		numWorkers = runtime.NumCPU() * 4
	}
	// sad.go:36-37 gives both channels 2*numWorkers slots and each of its numWorkers goroutines holds a chunk
	// while it waits to send: 5*numWorkers chunks can be under way before anyone reads -- examples/run.stream.go:46-58
	// feeds a whole frame (160 chunks for its 640x480 picture with 32 workers) before it reads.  One device worker
	// holds no such crowd: the output channel gets the goroutines' share.
	inputChan := make(chan InputChunk, numWorkers*2)
	outputChan := make(chan OutputChunk, numWorkers*3)
	go func() {
		defer close(outputChan)
		var pending *InputChunk
		for {
			var chunk InputChunk
			if pending != nil {
				chunk, pending = *pending, nil
			} else {
				c, ok := <-inputChan
				if !ok {
					return
				}
				chunk = c
			}
			group := []InputChunk{chunk}
		gather:
			for len(group) < maxGroup {
				select {
				case next, ok := <-inputChan:
					if !ok {
						break gather // closed: the outer receive sees it again
					}
					if continues(group[0], group[len(group)-1], next) {
						group = append(group, next)
					} else {
						n := next
						pending = &n // starts the next group
						break gather
					}
				default:
					break gather // nothing else queued right now
				}
			}
			params := *DefaultParams() // per group, like sad.go:51-53 per chunk
			all := image.Rect(group[0].Region.Min.X, group[0].Region.Min.Y, group[0].Region.Max.X, group[len(group)-1].Region.Max.Y)
			data, err := regionHIP(chunk.Left, chunk.Right, params, all)
			if err != nil {
				slog.Warn("HIP disparity failed, using the Go path", "err", err)
				for _, g := range group {
					outputChan <- regionGo(g)
				}
				continue
			}
			row := all.Dx()
			for _, g := range group {
				lo, hi := (g.Region.Min.Y-all.Min.Y)*row, (g.Region.Max.Y-all.Min.Y)*row
				outputChan <- OutputChunk{DisparityData: data[lo:hi:hi], Region: g.Region}
			}
		}
	}()
	return inputChan, outputChan
}

// RunSad computes the full disparity map with one device call (sad.go:119-169) and, like the
// reference, records the parameters as the process defaults (sad.go:123-126).
func RunSad(left, right *image.Gray, blockSize, maxDisparity int) *image.Gray {
	SetDefaultParams(Parameters{BlockSize: blockSize, MaxDisparity: maxDisparity})
	return RunSadFrame(left, right, Parameters{BlockSize: blockSize, MaxDisparity: maxDisparity})
}

// RunSadFrame is RunSad WITHOUT the write to the process-wide parameters: what a frame loop calls that
// only READS them (pkg/camera/output.go:169 -- patches/0002).  Calling RunSad there would write the
// values read at the start of the frame back over an update the UI handler made meanwhile
// (cmd/handlers/params.go:56-59) and lose it.
func RunSadFrame(left, right *image.Gray, p Parameters) *image.Gray {
	out := image.NewGray(left.Rect) // sad.go:135,168: the LEFT image's rectangle
	if len(out.Pix) == 0 {
		return out
	}
	data, err := regionHIP(left, right, p, left.Rect) // one call; left and right with their own sizes
	if err == nil {
		for y := 0; y < left.Rect.Dy(); y++ {
			copy(out.Pix[y*out.Stride:y*out.Stride+left.Rect.Dx()], data[y*left.Rect.Dx():(y+1)*left.Rect.Dx()])
		}
		return out
	}
	slog.Warn("HIP disparity failed, using the Go path", "err", err)
	// the reference's own workers with the parameters of THIS call (runSadFrameGo, sad_frame.go): the fallback
	// computes what the device would have, whatever the process-wide parameters are by now
	return runSadFrameGo(left, right, p)
}

// rgba8 reports whether img's pixels can go to sadhip_disparity_rgba_u8 as they are: 8-bit, four interleaved
// channels, NON-premultiplied alpha -- *image.NRGBA (what png.Decode returns for colour type 6), or *image.RGBA
// whose alpha is 255 everywhere (colour type 2; premultiplied == non-premultiplied there).  pix starts at
// Rect.Min.
func rgba8(img image.Image) (pix []uint8, stride int, ok bool) {
	switch m := img.(type) {
	case *image.NRGBA:
		if m.Rect.Empty() {
			return nil, 0, false
		}
		return m.Pix[m.PixOffset(m.Rect.Min.X, m.Rect.Min.Y):], m.Stride, true
	case *image.RGBA:
		if m.Rect.Empty() || !m.Opaque() {
			return nil, 0, false
		}
		return m.Pix[m.PixOffset(m.Rect.Min.X, m.Rect.Min.Y):], m.Stride, true
	}
	return nil, 0, false
}

// hostGray: SetHostGray.  0 = the pictures are uploaded as they are and converted on the device (the default).
var hostGray atomic.Uint32

// SetHostGray chooses where RunSadImages converts decoded pictures to gray (libsadhip ABI 5, SADHIP_FLAG_HOST_GRAY):
// false (default) uploads the pictures as they are and converts on the device; true lets libsadhip's copy threads
// write luminance rows while they stage the pictures -- the same integer formula, bit for bit -- so that a quarter
// of the bytes cross the bus (one 1080p NRGBA pair: 16.6 MB -> 4.1 MB; it costs the host the conversion on six
// threads for ~0.1 ms per call).  Either way no Go loop touches a pixel.
func SetHostGray(on bool) {
	if on {
		hostGray.Store(uint32(C.SADHIP_FLAG_HOST_GRAY))
	} else {
		hostGray.Store(0)
	}
}

// imagesHIP is one sadhip_disparity_rgba_u8 call: one upload per picture (staged through libsadhip's page-locked planes),
// the Go luminance -- color.GrayModel on alpha-premultiplied 16-bit channels, bit for bit -- and the disparity
// search on the device, one download.  No gray plane exists on the host.
func imagesHIP(lpix []uint8, lstride int, rpix []uint8, rstride int, w, h int, p Parameters) (*image.Gray, error) {
	out := image.NewGray(image.Rect(0, 0, w, h))
	err := ctx.with(w, h, func(c *C.sadhip_ctx) error {
		st := C.sadhip_disparity_rgba_u8(c,
			(*C.uint8_t)(unsafe.Pointer(&lpix[0])), (*C.uint8_t)(unsafe.Pointer(&rpix[0])), 4,
			C.int(w), C.int(h), C.int(lstride), C.int(rstride),
			C.int(p.BlockSize), C.int(p.MaxDisparity), C.uint(hostGray.Load()),
			(*C.uint8_t)(unsafe.Pointer(&out.Pix[0])), C.int(out.Stride))
		runtime.KeepAlive(lpix)
		runtime.KeepAlive(rpix)
		if st != C.SADHIP_OK {
			return statusErr("sadhip_disparity_rgba_u8", st)
		}
		return nil
	})
	return out, err
}

// RunSadImages computes the disparity map of two DECODED pictures -- what png.Decode hands
// OutputCamera.processDepthMap (pkg/camera/output.go:138,153; patches/0003) -- without the two per-pixel
// color.GrayModel.Convert loops of output.go:142-147,157-162 on the host:
//
//	*image.NRGBA, or *image.RGBA with alpha 255 everywhere, both of one size  one sadhip_disparity_rgba_u8 call
//	                                                                          (gray conversion on the device)
//	*image.Gray (what StaticCamera saves as left.png / right.png,            RunSadFrame on the pictures' own
//	             pkg/camera/static.go:141)                                    pixels: the conversion is the identity
//	anything else (16-bit, paletted, premultiplied alpha < 255), or          GrayOf on the host (the reference's
//	pictures of different sizes                                               loop), then RunSadFrame, which sends
//	                                                                          each picture with its own size
//
// Like RunSadFrame it only READS p: nothing is written to the process-wide parameters.  The result's rectangle
// is the left picture's, moved to the origin (sad.go:135,168 on a gray image made by image.NewGray(Bounds())).
func RunSadImages(left, right image.Image, p Parameters) *image.Gray {
	lb, rb := left.Bounds(), right.Bounds()
	if lpix, ls, ok := rgba8(left); ok && lb.Size() == rb.Size() {
		if rpix, rs, ok := rgba8(right); ok {
			out, err := imagesHIP(lpix, ls, rpix, rs, lb.Dx(), lb.Dy(), p)
			if err == nil {
				return out
			}
			slog.Warn("HIP disparity on decoded pictures failed, converting on the host", "err", err)
		}
	}
	return RunSadFrame(GrayOf(left), GrayOf(right), p)
}

// NewPinnedGray returns an *image.Gray whose Pix lives in page-locked memory from sadhip_host_alloc (C memory,
// not Go heap memory, so it may stay pinned across calls) and the function that frees it.  Frames a capture pipeline
// fills in such images (pkg/camera/serial.go reads one 1920x1080 gray frame per UART transfer) reach the device with
// asynchronous copies at bus speed; everything else about the image is ordinary.  Falls back to image.NewGray when
// the allocation fails.  Call free only when no RunSad / pipeline chunk using the image is in flight.
func NewPinnedGray(r image.Rectangle) (img *image.Gray, free func()) {
	n := r.Dx() * r.Dy()
	if n <= 0 {
		return image.NewGray(r), func() {}
	}
	var p unsafe.Pointer
	if st := C.sadhip_host_alloc(C.size_t(n), &p); st != C.SADHIP_OK {
		return image.NewGray(r), func() {}
	}
	pix := unsafe.Slice((*uint8)(p), n)
	return &image.Gray{Pix: pix, Stride: r.Dx(), Rect: r}, func() { C.sadhip_host_free(p) }
}

// AssembleDisparityMap assembles the disparity map from output chunks (sad.go:172-202).
// Unlike the reference it writes all `chunks` chunks (the reference breaks before writing the
// last one it receives, sad.go:181-184, leaving one rectangle black).
func AssembleDisparityMap(outputChan <-chan OutputChunk, dimensions image.Rectangle, chunks int) *image.Gray {
	disparityMap := image.NewGray(dimensions)
	i := 0
	for chunk := range outputChan {
		width := chunk.Region.Dx()
		for y := range chunk.Region.Dy() {
			row := disparityMap.PixOffset(chunk.Region.Min.X, chunk.Region.Min.Y+y)
			copy(disparityMap.Pix[row:row+width], chunk.DisparityData[y*width:(y+1)*width])
		}
		i++
		if i >= chunks {
			break
		}
	}
	return disparityMap
}
