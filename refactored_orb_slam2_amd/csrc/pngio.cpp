// pngio.cpp -- dependency-free PNG reader for the sequence driver: zlib only, no libpng / OpenCV.
// The reference's drivers read their datasets with cv::imread(..., IMREAD_UNCHANGED) (Source/Examples/Stereo/stereo_kitti.cc:
// 88-89) and Tracking converts colour input with cvtColor(RGB2GRAY / BGR2GRAY) (L/src/Tracking.cc:164-178).  Supported:
// 8-bit greyscale (KITTI, EuRoC, the DBoW2 demo images) and 8-bit RGB / RGBA (converted with cvtColor's fixed-point weights
// R 4899, G 9617, B 1868, >> 14), non-interlaced; all five filter types.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <vector>

#include "../../include/orbfe.h"

void orbfe_set_error(const char* fmt, ...);

static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static int paeth(int a, int b, int c) {
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
  buf.resize((size_t)n);
  const size_t got = fread(buf.data(), 1, (size_t)n, f);
  fclose(f);
  if (got != (size_t)n) { orbfe_set_error("%s: short read", path); return ORBFE_ERR_INVALID; }
  return ORBFE_OK;
}

// w, h of a PNG file without decoding it
extern "C" int orbfe_png_info(const char* path, int* w, int* h) {
  if (!path || !w || !h) return ORBFE_ERR_INVALID;
  std::vector<uint8_t> b;
  int rc = read_file(path, b);
  if (rc) return rc;
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (b.size() < 33 || memcmp(b.data(), sig, 8) != 0 || memcmp(b.data() + 12, "IHDR", 4) != 0) {
    orbfe_set_error("%s: not a PNG file", path);
    return ORBFE_ERR_INVALID;
  }
  *w = (int)be32(b.data() + 16);
  *h = (int)be32(b.data() + 20);
  return ORBFE_OK;
}

// Decodes `path` into dst (h rows of w bytes, `stride` bytes apart; cap_rows >= h, stride >= w).
extern "C" int orbfe_png_read_gray(const char* path, uint8_t* dst, int stride, int cap_rows, int* w_out, int* h_out) {
  if (!path || !dst) return ORBFE_ERR_INVALID;
  std::vector<uint8_t> b;
  int rc = read_file(path, b);
  if (rc) return rc;
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (memcmp(b.data(), sig, 8) != 0) { orbfe_set_error("%s: not a PNG file", path); return ORBFE_ERR_INVALID; }
  size_t pos = 8;
  int w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
  std::vector<uint8_t> idat;
  bool end = false;
  while (!end && pos + 12 <= b.size()) {
    const uint32_t len = be32(&b[pos]);
    const uint8_t* type = &b[pos + 4];
    if (pos + 12 + (size_t)len > b.size()) { orbfe_set_error("%s: truncated chunk", path); return ORBFE_ERR_INVALID; }
    const uint8_t* data = &b[pos + 8];
    if (!memcmp(type, "IHDR", 4) && len >= 13) {
      w = (int)be32(data); h = (int)be32(data + 4); depth = data[8]; ctype = data[9]; interlace = data[12];
    } else if (!memcmp(type, "IDAT", 4)) {
      idat.insert(idat.end(), data, data + len);
    } else if (!memcmp(type, "IEND", 4)) {
      end = true;
    }
    pos += 12 + (size_t)len;
  }
  int channels = 0;
  if (ctype == 0) channels = 1;
  else if (ctype == 2) channels = 3;
  else if (ctype == 4) channels = 2;
  else if (ctype == 6) channels = 4;
  if (w < 1 || h < 1 || depth != 8 || channels == 0 || interlace != 0) {
    orbfe_set_error("%s: unsupported PNG (%dx%d, depth %d, colour type %d, interlace %d): 8-bit grey / RGB(A), non-interlaced only", path, w,
                    h, depth, ctype, interlace);
    return ORBFE_ERR_INVALID;
  }
  if (w_out) *w_out = w;
  if (h_out) *h_out = h;
  if (h > cap_rows || stride < w) { orbfe_set_error("%s: %dx%d does not fit the destination", path, w, h); return ORBFE_ERR_CAPACITY; }
  const size_t rowb = (size_t)w * channels;
  std::vector<uint8_t> raw((rowb + 1) * (size_t)h);
  uLongf outlen = (uLongf)raw.size();
  const int zr = uncompress(raw.data(), &outlen, idat.data(), (uLong)idat.size());
  if (zr != Z_OK || outlen != raw.size()) { orbfe_set_error("%s: zlib error %d (%lu of %zu bytes)", path, zr, (unsigned long)outlen, raw.size()); return ORBFE_ERR_INVALID; }
  std::vector<uint8_t> prev(rowb, 0), cur(rowb);
  for (int y = 0; y < h; y++) {
    const uint8_t* in = &raw[(rowb + 1) * (size_t)y];
    const int ft = in[0];
    in++;
    const int bpp = channels;
    for (size_t i = 0; i < rowb; i++) {
      const int a = i >= (size_t)bpp ? cur[i - bpp] : 0, bb = prev[i], c = i >= (size_t)bpp ? prev[i - bpp] : 0;
      int v = in[i];
      switch (ft) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += bb; break;
        case 3: v += (a + bb) >> 1; break;
        case 4: v += paeth(a, bb, c); break;
        default: orbfe_set_error("%s: bad filter type %d in row %d", path, ft, y); return ORBFE_ERR_INVALID;
      }
      cur[i] = (uint8_t)v;
    }
    uint8_t* out = dst + (size_t)y * stride;
    if (channels == 1) memcpy(out, cur.data(), (size_t)w);
    else if (channels == 2) for (int x = 0; x < w; x++) out[x] = cur[2 * (size_t)x];
    else
      for (int x = 0; x < w; x++) {   // cvtColor RGB2GRAY, 8-bit: (R * 4899 + G * 9617 + B * 1868 + (1 << 13)) >> 14
        const uint8_t* p = &cur[(size_t)x * channels];
        out[x] = (uint8_t)((p[0] * 4899 + p[1] * 9617 + p[2] * 1868 + 8192) >> 14);
      }
    prev.swap(cur);
  }
  return ORBFE_OK;
}
