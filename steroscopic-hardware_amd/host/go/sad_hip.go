//go:build cgo && hip

// Package despair -- MI355X backend.  Drop this file (and sad_nohip_tag.txt's build-tag edit)
// into pkg/despair of conneroisu/steroscopic-hardware: it keeps every exported identifier of
// sad.go (SetupConcurrentSAD, RunSad, AssembleDisparityMap, InputChunk, OutputChunk) and routes
// the arithmetic to libsadhip.so through cgo.  Build with:
//
//	CGO_ENABLED=1 go build -tags hip .
//
// and give the existing pure-Go sad.go the complementary constraint `//go:build !(cgo && hip)`
// so release builds (CGO_ENABLED=0, .goreleaser.yaml:15-17) are unchanged.
//
// NOTE: this container has no Go toolchain (`go: command not found`), so this file is not
// compiled by the repo's tests; the identical logic is exercised through the C++ mirror
// (host/despair.cpp) and the ctypes mirror (despair.py), which call the same C ABI.
package despair

/*
#cgo CFLAGS:  -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../lib -lsadhip -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib
#include <stdlib.h>
#include "sadhip.h"
*/
import "C"

import (
	"fmt"
	"image"
	"log/slog"
	"runtime"
	"sync"
	"unsafe"
)

// InputChunk represents a portion of the image to process (sad.go:11-15).
type InputChunk struct {
	Left, Right *image.Gray
	Region      image.Rectangle
}

// OutputChunk represents the processed disparity data for a region (sad.go:17-21).
type OutputChunk struct {
	DisparityData []uint8
	Region        image.Rectangle
}

type hipCtx struct {
	mu   sync.Mutex
	h    *C.sadhip_ctx
	w, ht int
}

var ctx hipCtx

// get returns a context able to hold w x h frames, growing it on demand.
func (c *hipCtx) get(w, h int) (*C.sadhip_ctx, error) {
	c.mu.Lock()
	defer c.mu.Unlock()
	if c.h == nil || w > c.w || h > c.ht {
		if c.h != nil {
			C.sadhip_ctx_destroy(c.h)
			c.h = nil
		}
		nw, nh := max(w, c.w, 1920), max(h, c.ht, 1080)
		var out *C.sadhip_ctx
		if st := C.sadhip_ctx_create(0, C.int(nw), C.int(nh), 1, &out); st != C.SADHIP_OK {
			return nil, fmt.Errorf("sadhip_ctx_create: %s (%s)",
				C.GoString(C.sadhip_status_string(st)), C.GoString(C.sadhip_last_error()))
		}
		c.h, c.w, c.ht = out, nw, nh
	}
	return c.h, nil
}

func statusErr(what string, st C.int) error {
	return fmt.Errorf("%s: %s (%s)", what, C.GoString(C.sadhip_status_string(st)), C.GoString(C.sadhip_last_error()))
}

// regionHIP computes one chunk on the device.  Go pointers are only borrowed for the call
// (libsadhip copies into pinned staging before returning), as cgo requires.
func regionHIP(left, right *image.Gray, p Parameters, r image.Rectangle) ([]uint8, error) {
	h, err := ctx.get(left.Rect.Dx(), left.Rect.Dy())
	if err != nil {
		return nil, err
	}
	data := make([]uint8, r.Dx()*r.Dy())
	if len(data) == 0 {
		return data, nil
	}
	st := C.sadhip_disparity_region_u8(h,
		(*C.uint8_t)(unsafe.Pointer(&left.Pix[0])), (*C.uint8_t)(unsafe.Pointer(&right.Pix[0])),
		C.int(left.Rect.Dx()), C.int(left.Rect.Dy()), C.int(left.Stride), C.int(right.Stride),
		C.int(p.BlockSize), C.int(p.MaxDisparity), 0,
		C.int(r.Min.X), C.int(r.Min.Y), C.int(r.Max.X), C.int(r.Max.Y),
		(*C.uint8_t)(unsafe.Pointer(&data[0])))
	runtime.KeepAlive(left)
	runtime.KeepAlive(right)
	if st != C.SADHIP_OK {
		return nil, statusErr("sadhip_disparity_region_u8", st)
	}
	return data, nil
}

// SetupConcurrentSAD sets up a concurrent SAD processing pipeline (sad.go:29-113).
// One device worker replaces the goroutine pool; on any device error the chunk is computed by
// the pure-Go worker body (cpuRegion, the unchanged loop of sad.go:55-94) so the WebUI keeps working.
func SetupConcurrentSAD(numWorkers int) (chan<- InputChunk, <-chan OutputChunk) {
	if numWorkers <= 0 {
		numWorkers = runtime.NumCPU() * 4
	}
	inputChan := make(chan InputChunk, numWorkers*2)
	outputChan := make(chan OutputChunk, numWorkers*2)
	go func() {
		defer close(outputChan)
		for chunk := range inputChan {
			defaultParamsMu.Lock()
			params := *defaultParams.Load() // per chunk, like sad.go:51-53
			defaultParamsMu.Unlock()
			data, err := regionHIP(chunk.Left, chunk.Right, params, chunk.Region)
			if err != nil {
				slog.Warn("HIP disparity failed, using the Go path", "err", err)
				data = cpuRegion(chunk.Left, chunk.Right, params, chunk.Region)
			}
			outputChan <- OutputChunk{DisparityData: data, Region: chunk.Region}
		}
	}()
	return inputChan, outputChan
}

// RunSad computes the full disparity map with one device call (sad.go:119-169).
func RunSad(left, right *image.Gray, blockSize, maxDisparity int) *image.Gray {
	SetDefaultParams(Parameters{BlockSize: blockSize, MaxDisparity: maxDisparity})
	out := image.NewGray(left.Rect)
	h, err := ctx.get(left.Rect.Dx(), left.Rect.Dy())
	if err == nil {
		st := C.sadhip_disparity_u8(h,
			(*C.uint8_t)(unsafe.Pointer(&left.Pix[0])), (*C.uint8_t)(unsafe.Pointer(&right.Pix[0])),
			C.int(left.Rect.Dx()), C.int(left.Rect.Dy()), C.int(left.Stride), C.int(right.Stride),
			C.int(blockSize), C.int(maxDisparity), 0,
			(*C.uint8_t)(unsafe.Pointer(&out.Pix[0])), C.int(out.Stride))
		runtime.KeepAlive(left)
		runtime.KeepAlive(right)
		if st == C.SADHIP_OK {
			return out
		}
		err = statusErr("sadhip_disparity_u8", st)
	}
	slog.Warn("HIP disparity failed, using the Go path", "err", err)
	copy(out.Pix, cpuRegion(left, right, Parameters{blockSize, maxDisparity}, left.Rect))
	return out
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

// cpuRegion is the reference worker body (sad.go:55-94) kept as the fallback path.
func cpuRegion(left, right *image.Gray, params Parameters, region image.Rectangle) []uint8 {
	data := make([]uint8, region.Dx()*region.Dy())
	if params.MaxDisparity <= 0 {
		return data // the reference panics here (divide by zero, sad.go:92)
	}
	for y := range region.Dy() {
		globalY := region.Min.Y + y
		for x := range region.Dx() {
			minSAD := int(^uint(0) >> 1)
			best := 0
			for d := 0; d <= params.MaxDisparity; d++ {
				if (region.Min.X+x)-d < left.Rect.Min.X {
					continue
				}
				sad := SumAbsoluteDifferences(left, right, region.Min.X+x, globalY, region.Min.X+x-d, globalY, params.BlockSize)
				if sad < minSAD {
					minSAD, best = sad, d
					if sad == 0 {
						break
					}
				}
			}
			data[y*region.Dx()+x] = uint8((best * 255) / params.MaxDisparity)
		}
	}
	return data
}
