// png_tool.cpp -- command-line face of png_io for tests (no GPU needed):
//   png_tool decode in.png out.raw   -> prints "w h channels depth16 kind", writes the samples
//   png_tool encode in.raw w h out.png  (8-bit gray)
#include "png_io.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>

int main(int argc, char** argv)
{
  std::string err;
  if (argc == 4 && !strcmp(argv[1], "decode")) {
    despair::PngImage im;
    if (!despair::read_png(argv[2], im, &err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    std::ofstream(argv[3], std::ios::binary).write(reinterpret_cast<const char*>(im.samples.data()), (std::streamsize)im.samples.size());
    printf("%d %d %d %d %d\n", im.w, im.h, im.channels, im.depth16 ? 1 : 0, (int)im.kind);
    return 0;
  }
  if (argc == 6 && !strcmp(argv[1], "encode")) {
    const int w = atoi(argv[3]), h = atoi(argv[4]);
    std::ifstream f(argv[2], std::ios::binary);
    std::vector<char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    if ((long long)buf.size() != (long long)w * h) { fprintf(stderr, "size mismatch\n"); return 1; }
    if (!despair::write_png_gray8(argv[5], reinterpret_cast<const uint8_t*>(buf.data()), w, h, w, &err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; }
    return 0;
  }
  fprintf(stderr, "usage: png_tool decode in.png out.raw | encode in.raw w h out.png\n");
  return 2;
}
