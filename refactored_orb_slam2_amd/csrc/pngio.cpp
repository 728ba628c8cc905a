// pngio.cpp -- dependency-free PNG reader for the sequence driver: zlib only, no libpng / OpenCV.
// The reference's drivers read their datasets with cv::imread(..., IMREAD_UNCHANGED) (Source/Examples/Stereo/stereo_kitti.cc:
// 88-89, Source/Examples/RGB-D/rgbd_tum.cc: colour + 16-bit depth) and Tracking converts colour input with
// cvtColor(RGB2GRAY / BGR2GRAY) (L/src/Tracking.cc:164-178).  Supported, non-interlaced, all five filter types:
//   orbfe_png_read_gray    8-bit greyscale (KITTI, EuRoC, the DBoW2 demo images), grey + alpha, RGB / RGBA (converted with
//                          cvtColor's fixed-point weights R 4899, G 9617, B 1868, >> 14)
//   orbfe_png_read_gray16  16-bit greyscale (TUM RGB-D depth maps), samples returned in host byte order
// Sizes above 4095 x 4095 (the library's image limit) are refused before anything is allocated, and no exception leaves an
// entry point: a hostile IHDR is ORBFE_ERR_INVALID, an allocation failure ORBFE_ERR_ALLOC.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <new>
#include <vector>

#include "../../include/orbfe.h"

void orbfe_set_error(const char* fmt, ...);

#define PNG_MAX_SIDE 4095

static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static inline int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

static int read_file(const char* path, std::vector<uint8_t>& buf) {
  FILE* f = fopen(path, "rb");
  if (!f) { orbfe_set_error("cannot open %s", path); return ORBFE_ERR_INVALID; }
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  if (n < 8) { fclose(f); orbfe_set_error("%s: not a PNG file", path); return ORBFE_ERR_INVALID; }
  if (n > (long)(64u << 20)) { fclose(f); orbfe_set_error("%s: larger than any supported PNG", path); return ORBFE_ERR_INVALID; }
  buf.resize((size_t)n);
  const size_t got = fread(buf.data(), 1, (size_t)n, f);
  fclose(f);
  if (got != (size_t)n) { orbfe_set_error("%s: short read", path); return ORBFE_ERR_INVALID; }
  return ORBFE_OK;
}

struct PngHeader {
  int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0, channels = 0;
};

// Walks the chunks: header fields + the concatenated IDAT payload.
static int parse(const char* path, const std::vector<uint8_t>& b, PngHeader& hd, std::vector<uint8_t>* idat) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (b.size() < 33 || memcmp(b.data(), sig, 8) != 0 || memcmp(b.data() + 12, "IHDR", 4) != 0) {
    orbfe_set_error("%s: not a PNG file", path);
    return ORBFE_ERR_INVALID;
  }
  size_t pos = 8;
  bool end = false, have = false;
  while (!end && pos + 12 <= b.size()) {
    const uint32_t len = be32(&b[pos]);
    const uint8_t* type = &b[pos + 4];
    if (pos + 12 + (size_t)len > b.size()) { orbfe_set_error("%s: truncated chunk", path); return ORBFE_ERR_INVALID; }
    const uint8_t* data = &b[pos + 8];
    if (!memcmp(type, "IHDR", 4) && len >= 13) {
      const uint32_t w = be32(data), h = be32(data + 4);
      if (w < 1 || h < 1 || w > PNG_MAX_SIDE || h > PNG_MAX_SIDE) {
        orbfe_set_error("%s: %ux%u outside 1..%d", path, w, h, PNG_MAX_SIDE);
        return ORBFE_ERR_INVALID;
      }
      hd.w = (int)w; hd.h = (int)h; hd.depth = data[8]; hd.ctype = data[9]; hd.interlace = data[12];
      hd.channels = hd.ctype == 0 ? 1 : hd.ctype == 2 ? 3 : hd.ctype == 4 ? 2 : hd.ctype == 6 ? 4 : 0;
      have = true;
      if (!idat) return ORBFE_OK;
    } else if (!memcmp(type, "IDAT", 4)) {
      if (idat) idat->insert(idat->end(), data, data + len);
    } else if (!memcmp(type, "IEND", 4)) {
      end = true;
    }
    pos += 12 + (size_t)len;
  }
  if (!have) { orbfe_set_error("%s: no IHDR", path); return ORBFE_ERR_INVALID; }
  return ORBFE_OK;
}

// Reverses the row filter of one scanline in place (`cur` holds the filtered bytes, `prev` the previous reconstructed line).
static int unfilter(int ft, uint8_t* cur, const uint8_t* prev, size_t rowb, int bpp) {
  switch (ft) {
    case 0: return 0;
    case 1:
      for (size_t i = (size_t)bpp; i < rowb; i++) cur[i] = (uint8_t)(cur[i] + cur[i - bpp]);
      return 0;
    case 2:
      for (size_t i = 0; i < rowb; i++) cur[i] = (uint8_t)(cur[i] + prev[i]);
      return 0;
    case 3:
      for (size_t i = 0; i < (size_t)bpp && i < rowb; i++) cur[i] = (uint8_t)(cur[i] + (prev[i] >> 1));
      for (size_t i = (size_t)bpp; i < rowb; i++) cur[i] = (uint8_t)(cur[i] + ((cur[i - bpp] + prev[i]) >> 1));
      return 0;
    case 4:
      for (size_t i = 0; i < (size_t)bpp && i < rowb; i++) cur[i] = (uint8_t)(cur[i] + prev[i]);   // paeth(0, b, 0) = b
      for (size_t i = (size_t)bpp; i < rowb; i++) cur[i] = (uint8_t)(cur[i] + paeth(cur[i - bpp], prev[i], prev[i - bpp]));
      return 0;
    default: return 1;
  }
}

// Inflates and un-filters the whole image: `rows` receives h lines of rowb bytes, each preceded by its (consumed) filter byte.
static int decode_rows(const char* path, const PngHeader& hd, const std::vector<uint8_t>& idat, std::vector<uint8_t>& raw,
                       size_t* rowb_out) {
  const size_t rowb = (size_t)hd.w * hd.channels * (hd.depth / 8);
  raw.resize((rowb + 1) * (size_t)hd.h);
  uLongf outlen = (uLongf)raw.size();
  const int zr = uncompress(raw.data(), &outlen, idat.data(), (uLong)idat.size());
  if (zr != Z_OK || outlen != raw.size()) {
    orbfe_set_error("%s: zlib error %d (%lu of %zu bytes)", path, zr, (unsigned long)outlen, raw.size());
    return ORBFE_ERR_INVALID;
  }
  const int bpp = hd.channels * (hd.depth / 8);
  std::vector<uint8_t> zero(rowb, 0);
  const uint8_t* prev = zero.data();
  for (int y = 0; y < hd.h; y++) {
    uint8_t* line = &raw[(rowb + 1) * (size_t)y];
    if (unfilter(line[0], line + 1, prev, rowb, bpp)) {
      orbfe_set_error("%s: bad filter type %d in row %d", path, line[0], y);
      return ORBFE_ERR_INVALID;
    }
    prev = line + 1;
  }
  *rowb_out = rowb;
  return ORBFE_OK;
}

static int png_info(const char* path, int* w, int* h, int* depth, int* channels) {
  std::vector<uint8_t> b;
  int rc = read_file(path, b);
  if (rc) return rc;
  PngHeader hd;
  if ((rc = parse(path, b, hd, nullptr))) return rc;
  if (w) *w = hd.w;
  if (h) *h = hd.h;
  if (depth) *depth = hd.depth;
  if (channels) *channels = hd.channels;
  return ORBFE_OK;
}

static int png_read_gray(const char* path, uint8_t* dst, int stride, int cap_rows, int* w_out, int* h_out, int camera_rgb) {
  std::vector<uint8_t> b, idat, raw;
  int rc = read_file(path, b);
  if (rc) return rc;
  PngHeader hd;
  if ((rc = parse(path, b, hd, &idat))) return rc;
  if (hd.depth != 8 || hd.channels == 0 || hd.interlace != 0) {
    orbfe_set_error("%s: unsupported PNG (%dx%d, depth %d, colour type %d, interlace %d): 8-bit grey / RGB(A), non-interlaced only%s",
                    path, hd.w, hd.h, hd.depth, hd.ctype, hd.interlace,
                    hd.depth == 16 && hd.ctype == 0 ? " (16-bit grey: orbfe_png_read_gray16)" : "");
    return ORBFE_ERR_INVALID;
  }
  if (w_out) *w_out = hd.w;
  if (h_out) *h_out = hd.h;
  if (hd.h > cap_rows || stride < hd.w) {
    orbfe_set_error("%s: %dx%d does not fit the destination", path, hd.w, hd.h);
    return ORBFE_ERR_CAPACITY;
  }
  size_t rowb = 0;
  if ((rc = decode_rows(path, hd, idat, raw, &rowb))) return rc;
  const int w = hd.w, channels = hd.channels;
  for (int y = 0; y < hd.h; y++) {
    const uint8_t* cur = &raw[(rowb + 1) * (size_t)y + 1];
    uint8_t* out = dst + (size_t)y * stride;
    if (channels == 1) memcpy(out, cur, (size_t)w);
    else if (channels == 2) for (int x = 0; x < w; x++) out[x] = cur[2 * (size_t)x];
    else
      for (int x = 0; x < w; x++) {   // cvtColor, 8-bit: (c0 * 4899 + G * 9617 + c2 * 1868 + (1 << 13)) >> 14
        // camera_rgb = 0: c0 = R, c2 = B -- the luminance, what Tracking::GrabImage* computes with Camera.RGB: 0 (BGR2GRAY on
        // imread's BGR data).  camera_rgb = 1: c0 = B, c2 = R -- what it computes with Camera.RGB: 1, the value of every settings
        // file the reference ships: RGB2GRAY applied to imread's BGR data (L/src/Tracking.cc:164-178)
        const uint8_t* p = &cur[(size_t)x * channels];
        const int c0 = camera_rgb ? p[2] : p[0], c2 = camera_rgb ? p[0] : p[2];
        out[x] = (uint8_t)((c0 * 4899 + p[1] * 9617 + c2 * 1868 + 8192) >> 14);
      }
  }
  return ORBFE_OK;
}

static int png_read_gray16(const char* path, uint16_t* dst, int stride_elems, int cap_rows, int* w_out, int* h_out) {
  std::vector<uint8_t> b, idat, raw;
  int rc = read_file(path, b);
  if (rc) return rc;
  PngHeader hd;
  if ((rc = parse(path, b, hd, &idat))) return rc;
  if (hd.depth != 16 || hd.ctype != 0 || hd.interlace != 0) {
    orbfe_set_error("%s: not a 16-bit greyscale non-interlaced PNG (%dx%d, depth %d, colour type %d, interlace %d)", path, hd.w,
                    hd.h, hd.depth, hd.ctype, hd.interlace);
    return ORBFE_ERR_INVALID;
  }
  if (w_out) *w_out = hd.w;
  if (h_out) *h_out = hd.h;
  if (hd.h > cap_rows || stride_elems < hd.w) {
    orbfe_set_error("%s: %dx%d does not fit the destination", path, hd.w, hd.h);
    return ORBFE_ERR_CAPACITY;
  }
  size_t rowb = 0;
  if ((rc = decode_rows(path, hd, idat, raw, &rowb))) return rc;
  for (int y = 0; y < hd.h; y++) {
    const uint8_t* cur = &raw[(rowb + 1) * (size_t)y + 1];
    uint16_t* out = dst + (size_t)y * stride_elems;
    for (int x = 0; x < hd.w; x++) out[x] = (uint16_t)((cur[2 * (size_t)x] << 8) | cur[2 * (size_t)x + 1]);   // PNG samples are big-endian
  }
  return ORBFE_OK;
}

// No C++ exception crosses the C ABI (SURVEY §8(b) Errors).
template <typename F>
static int guarded(const char* path, F f) {
  try {
    return f();
  } catch (const std::bad_alloc&) {
    orbfe_set_error("%s: out of memory", path ? path : "(null)");
    return ORBFE_ERR_ALLOC;
  } catch (...) {
    orbfe_set_error("%s: PNG reader failed", path ? path : "(null)");
    return ORBFE_ERR_INVALID;
  }
}

// w, h of a PNG file without decoding it
extern "C" int orbfe_png_info(const char* path, int* w, int* h) {
  if (!path || !w || !h) return ORBFE_ERR_INVALID;
  return guarded(path, [&] { return png_info(path, w, h, nullptr, nullptr); });
}
// ... and its bit depth (8 / 16) and channel count (1 grey, 2 grey + alpha, 3 RGB, 4 RGBA)
extern "C" int orbfe_png_info2(const char* path, int* w, int* h, int* depth, int* channels) {
  if (!path) return ORBFE_ERR_INVALID;
  return guarded(path, [&] { return png_info(path, w, h, depth, channels); });
}
// Decodes `path` into dst (h rows of w bytes, `stride` bytes apart; cap_rows >= h, stride >= w).
extern "C" int orbfe_png_read_gray(const char* path, uint8_t* dst, int stride, int cap_rows, int* w_out, int* h_out) {
  if (!path || !dst) return ORBFE_ERR_INVALID;
  return guarded(path, [&] { return png_read_gray(path, dst, stride, cap_rows, w_out, h_out, 0); });
}
// ... with the settings file's Camera.RGB flag for colour input (see the conversion above); grey files are unaffected
extern "C" int orbfe_png_read_gray2(const char* path, uint8_t* dst, int stride, int cap_rows, int* w_out, int* h_out, int camera_rgb) {
  if (!path || !dst) return ORBFE_ERR_INVALID;
  return guarded(path, [&] { return png_read_gray(path, dst, stride, cap_rows, w_out, h_out, camera_rgb != 0); });
}
// 16-bit greyscale (depth maps): h rows of w uint16, `stride_elems` elements apart.
extern "C" int orbfe_png_read_gray16(const char* path, uint16_t* dst, int stride_elems, int cap_rows, int* w_out, int* h_out) {
  if (!path || !dst) return ORBFE_ERR_INVALID;
  return guarded(path, [&] { return png_read_gray16(path, dst, stride_elems, cap_rows, w_out, h_out); });
}
