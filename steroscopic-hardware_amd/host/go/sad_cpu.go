//go:build !(cgo && hip)

// Default build of pkg/despair (CGO_ENABLED=0 release builds, .goreleaser.yaml:15-17): the
// exported entry points are the reference's own functions of sad.go, unchanged -- patch
// 0001 only gave them unexported names so that the hip build (sad_hip.go) can export the same
// identifiers and still fall back to them.  New file, added next to sad.go.
package despair

import "image"

// FrameLevel reports whether RunSad is one device call (true in the hip build): pkg/camera's
// OutputCamera then hands over whole frames instead of row-band chunks (patch 0002).
const FrameLevel = false

// SetupConcurrentSAD sets up a concurrent SAD processing pipeline (sad.go:29-113).
func SetupConcurrentSAD(numWorkers int) (chan<- InputChunk, <-chan OutputChunk) {
	return setupConcurrentSADGo(numWorkers)
}

// RunSad sets up the pipeline, feeds the images and assembles the disparity map (sad.go:119-169).
func RunSad(left, right *image.Gray, blockSize, maxDisparity int) *image.Gray {
	return runSadGo(left, right, blockSize, maxDisparity)
}

// RunSadFrame exists so that pkg/camera compiles in both builds (patch 0002); with FrameLevel == false it is
// never called.  In the hip build it is RunSad without the write to the process-wide parameters.
func RunSadFrame(left, right *image.Gray, p Parameters) *image.Gray {
	return runSadGo(left, right, p.BlockSize, p.MaxDisparity)
}

// NewPinnedGray is image.NewGray in this build (the hip build returns page-locked memory); free is a no-op.
func NewPinnedGray(r image.Rectangle) (img *image.Gray, free func()) {
	return image.NewGray(r), func() {}
}

// AssembleDisparityMap assembles the disparity map from output chunks (sad.go:172-202).
func AssembleDisparityMap(outputChan <-chan OutputChunk, dimensions image.Rectangle, chunks int) *image.Gray {
	return assembleDisparityMapGo(outputChan, dimensions, chunks)
}
