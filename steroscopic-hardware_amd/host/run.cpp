// run.cpp -- counterpart of the reference's examples/run.go: load the stereo pair, RunSad with the
// default parameters of that example (16, 64), save the disparity map.
//   usage: run [left.png right.png [out.png [block maxdisp]]]
#include "despair.hpp"
#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv)
{
  const char* left = argc > 2 ? argv[1] : "./testdata/L_00001.png";     // run.go:21
  const char* right = argc > 2 ? argv[2] : "./testdata/R_00001.png";    // run.go:26
  const char* out = argc > 3 ? argv[3] : "output.png";                  // run.go:31
  const int block = argc > 5 ? atoi(argv[4]) : 16, maxdisp = argc > 5 ? atoi(argv[5]) : 64;   // run.go:30
  try {
    const despair::Gray L = despair::LoadPNG(left), R = despair::LoadPNG(right);
    const despair::Gray got = despair::RunSad(L, R, block, maxdisp);
    despair::SavePNG(out, got);
    printf("%s: %dx%d disparity map, block %d, max disparity %d\n", out, got.Rect.Dx(), got.Rect.Dy(), block, maxdisp);
  } catch (const despair::Error& e) {
    fprintf(stderr, "despair error %d: %s\n", e.status, e.what());    // run.go panics
    return 1;
  }
  return 0;
}
