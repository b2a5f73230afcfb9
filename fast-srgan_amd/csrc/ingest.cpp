// Host-side ingest: PNG files -> uint8 CHW .npy files, natively and in parallel.
//
// Replaces write_images_to_numpy_arrays of /root/reference/train.py:22-37 -- per image
//     np.save(path, np.transpose(np.array(Image.open(p).convert("RGB")).astype(np.uint8), (2, 0, 1)))
// on a 16-thread pool that spends its time in PIL and under the GIL.  Here every worker thread reads a file, inflates its IDAT
// stream with zlib, undoes the five PNG row filters, converts to RGB the way PIL's convert("RGB") does for the 8-bit colour types
// (grey replicated, palette looked up, alpha dropped -- never blended) and writes the .npy (format 1.0, '|u1', C order, shape
// (3, H, W)) that NumpyImagesDataset (dataloader.py:9-22) memory-maps.  No device code: the pool that follows (the uint8 images
// resident in HBM, crops cut by csrc/data.hip) is where the GPU comes in.
//
// Not decoded here (the call reports -4 for the file and the Python caller hands exactly those files to PIL): interlaced (Adam7)
// images, 16-bit samples, grey images with fewer than 8 bits.  DIV2K -- the reference's data set -- is 8-bit RGB, non-interlaced.
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include <atomic>
#include <string>
#include <thread>
#include <vector>

#include "fsr_hip.h"
#include "fsr_host.h"

namespace {

uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

bool read_file(const char* path, std::vector<unsigned char>& buf) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (n < 0) {
    fclose(f);
    return false;
  }
  buf.resize((size_t)n);
  const size_t got = n ? fread(buf.data(), 1, (size_t)n, f) : 0;
  fclose(f);
  return got == (size_t)n;
}

inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

// 0 = ok; -1 not a PNG / truncated; -2 corrupt stream; -4 a PNG this decoder leaves to the caller (see the header)
int decode_png_chw(const std::vector<unsigned char>& file, std::vector<unsigned char>& chw, int& H, int& W) {
  static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 8 + 25 || memcmp(file.data(), sig, 8) != 0) return -1;
  size_t pos = 8;
  int depth = 0, ctype = 0, interlace = 0;
  bool have_ihdr = false;
  std::vector<unsigned char> idat, plte;
  while (pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const unsigned char* type = &file[pos + 4];
    if (pos + 12 + (size_t)len > file.size()) return -1;
    const unsigned char* data = &file[pos + 8];
    if (!memcmp(type, "IHDR", 4)) {
      if (len < 13) return -1;
      W = (int)be32(data);
      H = (int)be32(data + 4);
      depth = data[8];
      ctype = data[9];
      interlace = data[12];
      have_ihdr = true;
    } else if (!memcmp(type, "PLTE", 4)) {
      plte.assign(data, data + len);
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!memcmp(type, "IEND", 4)) {
      break;
    }
    pos += 12 + (size_t)len;
  }
  if (!have_ihdr || W <= 0 || H <= 0 || idat.empty()) return -1;
  // the IHDR dimensions are untrusted: bound them (a data-set image is a few thousand pixels wide) and hold the raster they
  // claim against what a deflate stream of this size can possibly inflate to (at most ~1032 : 1) BEFORE allocating it
  if (W > (1 << 16) || H > (1 << 16)) return -2;
  if (interlace != 0) return -4;
  int ch;                                         // samples per pixel in the stream
  if (ctype == 0) ch = 1;
  else if (ctype == 2) ch = 3;
  else if (ctype == 3) ch = 1;
  else if (ctype == 4) ch = 2;
  else if (ctype == 6) ch = 4;
  else return -1;
  if (ctype == 3) {
    if (depth != 1 && depth != 2 && depth != 4 && depth != 8) return -1;
    if (plte.empty()) return -2;
  } else if (depth != 8) {
    return -4;                                    // 16-bit samples, low-depth grey: the caller's PIL path
  }
  const size_t rowbytes = ((size_t)W * ch * depth + 7) / 8;
  const size_t bpp = (size_t)(ch * depth + 7) / 8;            // filter distance in bytes (1 for sub-byte palettes)
  if ((rowbytes + 1) * (size_t)H > (size_t)1040 * idat.size() + 4096) return -2;
  std::vector<unsigned char> raw((rowbytes + 1) * (size_t)H);
  uLongf rawlen = (uLongf)raw.size();
  if (uncompress(raw.data(), &rawlen, idat.data(), (uLong)idat.size()) != Z_OK || rawlen != raw.size()) return -2;
  // undo the row filters in place (row r occupies raw[r * (rowbytes + 1) + 1 ..]; its first byte is the filter type)
  for (int y = 0; y < H; ++y) {
    unsigned char* row = &raw[(size_t)y * (rowbytes + 1) + 1];
    const unsigned char* up = y ? row - (rowbytes + 1) : nullptr;
    const int ft = row[-1];
    switch (ft) {
      case 0: break;
      case 1:
        for (size_t i = bpp; i < rowbytes; ++i) row[i] = (unsigned char)(row[i] + row[i - bpp]);
        break;
      case 2:
        if (up)
          for (size_t i = 0; i < rowbytes; ++i) row[i] = (unsigned char)(row[i] + up[i]);
        break;
      case 3:
        for (size_t i = 0; i < rowbytes; ++i) {
          const int a = i >= bpp ? row[i - bpp] : 0, b = up ? up[i] : 0;
          row[i] = (unsigned char)(row[i] + ((a + b) >> 1));
        }
        break;
      case 4:
        for (size_t i = 0; i < rowbytes; ++i) {
          const int a = i >= bpp ? row[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
          row[i] = (unsigned char)(row[i] + paeth(a, b, c));
        }
        break;
      default: return -2;
    }
  }
  // to planar RGB
  const size_t plane = (size_t)H * W;
  chw.resize(3 * plane);
  unsigned char* R = chw.data();
  unsigned char* G = R + plane;
  unsigned char* B = G + plane;
  for (int y = 0; y < H; ++y) {
    const unsigned char* row = &raw[(size_t)y * (rowbytes + 1) + 1];
    const size_t o = (size_t)y * W;
    if (ctype == 2 || ctype == 6) {
      for (int x = 0; x < W; ++x) {
        R[o + x] = row[(size_t)x * ch];
        G[o + x] = row[(size_t)x * ch + 1];
        B[o + x] = row[(size_t)x * ch + 2];
      }
    } else if (ctype == 0 || ctype == 4) {
      for (int x = 0; x < W; ++x) R[o + x] = G[o + x] = B[o + x] = row[(size_t)x * ch];
    } else {
      const int mask = (1 << depth) - 1, per = 8 / depth;
      for (int x = 0; x < W; ++x) {
        const int idx = depth == 8 ? row[x] : ((row[x / per] >> ((per - 1 - x % per) * depth)) & mask);
        if ((size_t)idx * 3 + 2 >= plte.size()) return -2;
        R[o + x] = plte[(size_t)idx * 3];
        G[o + x] = plte[(size_t)idx * 3 + 1];
        B[o + x] = plte[(size_t)idx * 3 + 2];
      }
    }
  }
  return 0;
}

// .npy format 1.0: magic, version, little-endian header length, a Python dict literal padded with spaces so that the data starts
// on a 64-byte boundary, newline-terminated
bool write_npy_u8_chw(const char* path, const unsigned char* data, int H, int W) {
  char dict[128];
  snprintf(dict, sizeof(dict), "{'descr': '|u1', 'fortran_order': False, 'shape': (3, %d, %d), }", H, W);
  std::string hdr(dict);
  const size_t base = 10 + hdr.size() + 1;
  hdr.append((64 - base % 64) % 64, ' ');
  hdr.push_back('\n');
  FILE* f = fopen(path, "wb");
  if (!f) return false;
  const unsigned char magic[8] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0};
  const unsigned char hl[2] = {(unsigned char)(hdr.size() & 0xff), (unsigned char)(hdr.size() >> 8)};
  bool ok = fwrite(magic, 1, 8, f) == 8 && fwrite(hl, 1, 2, f) == 2 && fwrite(hdr.data(), 1, hdr.size(), f) == hdr.size();
  const size_t n = (size_t)3 * H * W;
  ok = ok && fwrite(data, 1, n, f) == n;
  ok = (fclose(f) == 0) && ok;
  return ok;
}

}  // namespace

extern "C" int fsr_png_decode_chw(const char* png_path, unsigned char* out_chw, size_t capacity, int* height, int* width) {
  if (!png_path || !height || !width) return fsr_fail(-1, "fsr_png_decode_chw: null argument");
  std::vector<unsigned char> file, chw;
  if (!read_file(png_path, file)) return fsr_fail(-3, "fsr_png_decode_chw: cannot read %s", png_path);
  int H = 0, W = 0;
  int rc;
  try {
    rc = decode_png_chw(file, chw, H, W);
  } catch (...) {        // std::bad_alloc / std::length_error of a hostile header must not cross the C ABI
    rc = -2;
  }
  if (rc) return fsr_fail(rc, "fsr_png_decode_chw: %s: %s", png_path,
                          rc == -4 ? "interlaced / 16-bit / low-depth grey PNG: not decoded here" : (rc == -1 ? "not a PNG file" : "corrupt PNG stream"));
  *height = H;
  *width = W;
  if (out_chw) {
    if (capacity < chw.size()) return fsr_fail(-2, "fsr_png_decode_chw: %zu bytes needed, %zu given", chw.size(), capacity);
    memcpy(out_chw, chw.data(), chw.size());
  }
  return 0;
}

extern "C" int fsr_png_to_npy(const char* const* png_paths, const char* const* npy_paths, int count, int threads, int* status) {
  if (!png_paths || !npy_paths || count < 0) return fsr_fail(-1, "fsr_png_to_npy: null argument");
  if (threads < 1) threads = 1;
  if (threads > count) threads = count > 0 ? count : 1;
  std::atomic<int> next(0), failed(0);
  auto work = [&]() {
    std::vector<unsigned char> file, chw;
    for (;;) {
      const int i = next.fetch_add(1);
      if (i >= count) break;
      int H = 0, W = 0, rc = -3;
      try {
        if (read_file(png_paths[i], file)) {
          rc = decode_png_chw(file, chw, H, W);
          if (rc == 0 && !write_npy_u8_chw(npy_paths[i], chw.data(), H, W)) rc = -3;
        }
      } catch (...) {    // an exception escaping a std::thread terminates the process (rank 0 of a training job)
        rc = -2;
      }
      if (status) status[i] = rc;
      if (rc) failed.fetch_add(1);
    }
  };
  std::vector<std::thread> pool;
  for (int t = 1; t < threads; ++t) pool.emplace_back(work);
  work();
  for (auto& th : pool) th.join();
  return failed.load();      // number of files not converted (their status says why); 0 = all done
}
