// Shared by both builds of pkg/despair (no build tag).  New file, added next to sad.go after patches/0001:
// what RunSadFrame / RunSadImages need on the HOST in either build -- the reference's gray conversion as a
// function, and a frame computed by the reference's own workers with the parameters of the call (patch 0001
// gives the worker pool the optional `fixed` argument; its loops are untouched).
package despair

import (
	"image"
	"image/color"
	"runtime"
)

// setupConcurrentSADGo is the reference's pipeline as it was: every chunk with the process-wide parameters.
func setupConcurrentSADGo(numWorkers int) (chan<- InputChunk, <-chan OutputChunk) {
	return setupConcurrentSADWith(numWorkers, nil)
}

// GrayOf is the conversion OutputCamera.processDepthMap applies to a decoded picture
// (pkg/camera/output.go:142-147,157-162): color.GrayModel.Convert per pixel into image.NewGray(img.Bounds()).
// An *image.Gray is returned as it is (the conversion is the identity on it).
func GrayOf(img image.Image) *image.Gray {
	if g, ok := img.(*image.Gray); ok {
		return g
	}
	b := img.Bounds()
	g := image.NewGray(b)
	for y := b.Min.Y; y < b.Max.Y; y++ {
		for x := b.Min.X; x < b.Max.X; x++ {
			g.SetGray(x, y, color.GrayModel.Convert(img.At(x, y)).(color.Gray))
		}
	}
	return g
}

// runSadFrameGo computes one frame on the CPU with the parameters p and writes nothing to the process-wide
// ones: the reference's workers (sad.go:43-103, `fixed` = &p) over row bands like OutputCamera's
// (pkg/camera/output.go:172-187), assembled without the dropped chunk of sad.go:181-184.
func runSadFrameGo(left, right *image.Gray, p Parameters) *image.Gray {
	out := image.NewGray(left.Rect)
	if left.Rect.Empty() {
		return out
	}
	workers := runtime.NumCPU()
	in, res := setupConcurrentSADWith(workers, &p)
	band := max(1, left.Rect.Dy()/(workers*4))
	n := (left.Rect.Dy() + band - 1) / band
	go func() {
		for y := left.Rect.Min.Y; y < left.Rect.Max.Y; y += band {
			in <- InputChunk{Left: left, Right: right,
				Region: image.Rect(left.Rect.Min.X, y, left.Rect.Max.X, min(y+band, left.Rect.Max.Y))}
		}
		close(in)
	}()
	for i := 0; i < n; i++ {
		c := <-res
		w := c.Region.Dx()
		for y := range c.Region.Dy() {
			row := out.PixOffset(c.Region.Min.X, c.Region.Min.Y+y)
			copy(out.Pix[row:row+w], c.DisparityData[y*w:(y+1)*w])
		}
	}
	return out
}
