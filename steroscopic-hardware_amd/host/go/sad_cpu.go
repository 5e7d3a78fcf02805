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

// RunSadFrame is RunSad WITHOUT the write to the process-wide parameters (the hip build: one device call;
// here: the reference's workers with p, runSadFrameGo).  pkg/camera compiles against it in both builds
// (patch 0002); with FrameLevel == false OutputCamera keeps its chunk pipeline and never calls it.
func RunSadFrame(left, right *image.Gray, p Parameters) *image.Gray {
	return runSadFrameGo(left, right, p)
}

// RunSadImages is what pkg/camera/output.go:142-190 does with two decoded pictures -- the gray conversion of
// both (GrayOf), then the disparity map -- with the parameters p (patch 0003; the hip build converts on the
// device).  With FrameLevel == false OutputCamera never calls it.
func RunSadImages(left, right image.Image, p Parameters) *image.Gray {
	return runSadFrameGo(GrayOf(left), GrayOf(right), p)
}

// SetHostGray is a no-op in this build (the hip build chooses between gray conversion on the device and in libsadhip's
// copy threads; here GrayOf converts on the host either way).
func SetHostGray(on bool) {}

// NewPinnedGray is image.NewGray in this build (the hip build returns page-locked memory); free is a no-op.
func NewPinnedGray(r image.Rectangle) (img *image.Gray, free func()) {
	return image.NewGray(r), func() {}
}

// AssembleDisparityMap assembles the disparity map from output chunks (sad.go:172-202).
func AssembleDisparityMap(outputChan <-chan OutputChunk, dimensions image.Rectangle, chunks int) *image.Gray {
	return assembleDisparityMapGo(outputChan, dimensions, chunks)
}
