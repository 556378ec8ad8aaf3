// oracle/ref/ref_wire_tool.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Command-line driver around the pieces of the REFERENCE's frame ingest that compile in this image, built from the
// sources where they lie under /root/reference (nothing is copied into this repository, see oracle/ref/Makefile):
//   * external/squish  -- the DXT codec the reference itself decodes recorded colour with
//                         (framework/NetKinectArray.cpp:620 squish::DecompressImage(..., squish::kDxt1))
//   * framework/io/FileBuffer.{h,cpp} -- the reader of recordings/<sensor>.stream (NetKinectArray.cpp:709-749)
//
//   compress   dxt1|dxt5 <in.rgba> <w> <h> <out.dxt>     squish::CompressImage
//   decompress dxt1|dxt5 <in.dxt>  <w> <h> <out.rgba>    squish::DecompressImage
//   stream-append <file.stream> <in.bin>                 sys::FileBuffer::write (open "a")
//   stream-read   <file.stream> <colorsize> <depthsize> <frame> <out.bin>
//                 the access pattern of readFromFiles(): read(colorsize) then read(depthsize), frame+1 times
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include <squish.h>        // -I/root/reference/external/squish
#include <FileBuffer.h>    // -I/root/reference/framework/io

static std::vector<unsigned char> slurp(const char* p) {
  std::vector<unsigned char> v;
  FILE* f = fopen(p, "rb");
  if (!f) { perror(p); exit(1); }
  fseek(f, 0, SEEK_END); v.resize((size_t)ftell(f)); fseek(f, 0, SEEK_SET);
  if (!v.empty() && fread(v.data(), 1, v.size(), f) != v.size()) { perror(p); exit(1); }
  fclose(f);
  return v;
}
static void spill(const char* p, const void* d, size_t n) {
  FILE* f = fopen(p, "wb");
  if (!f || fwrite(d, 1, n, f) != n) { perror(p); exit(1); }
  fclose(f);
}

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "";
  if ((mode == "compress" || mode == "decompress") && argc == 7) {
    const int flags = std::string(argv[2]) == "dxt5" ? squish::kDxt5 : squish::kDxt1;
    const int w = atoi(argv[4]), h = atoi(argv[5]);
    std::vector<unsigned char> in = slurp(argv[3]);
    const size_t blocks = (size_t)squish::GetStorageRequirements(w, h, flags), pixels = (size_t)w * h * 4;
    if (mode == "compress") {
      if (in.size() != pixels) { fprintf(stderr, "expected %zu bytes of RGBA\n", pixels); return 1; }
      std::vector<unsigned char> out(blocks);
      squish::CompressImage(in.data(), w, h, out.data(), flags);
      spill(argv[6], out.data(), out.size());
    } else {
      if (in.size() != blocks) { fprintf(stderr, "expected %zu bytes of blocks\n", blocks); return 1; }
      std::vector<unsigned char> out(pixels);
      squish::DecompressImage(out.data(), w, h, in.data(), flags);
      spill(argv[6], out.data(), out.size());
    }
    return 0;
  }
  if (mode == "stream-append" && argc == 4) {
    std::vector<unsigned char> in = slurp(argv[3]);
    sys::FileBuffer fb(argv[2]);
    if (!fb.open("a")) return 1;
    return fb.write(in.data(), (unsigned)in.size()) == in.size() ? 0 : 1;
  }
  if (mode == "stream-read" && argc == 7) {
    const unsigned cs = (unsigned)atoi(argv[3]), ds = (unsigned)atoi(argv[4]), frame = (unsigned)atoi(argv[5]);
    sys::FileBuffer fb(argv[2]);
    if (!fb.open("r")) return 1;
    fb.setLooping(false);
    std::vector<unsigned char> rec(cs + ds);
    for (unsigned k = 0; k <= frame; ++k)
      if (fb.read(rec.data(), cs) != cs || fb.read(rec.data() + cs, ds) != ds) { fprintf(stderr, "end of stream\n"); return 3; }
    spill(argv[6], rec.data(), rec.size());
    return 0;
  }
  fprintf(stderr, "usage: see the header of ref_wire_tool.cpp\n");
  return 2;
}
