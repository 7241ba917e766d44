// rt_image.cpp — PNG in and out for the host side (textures in, frames out).
//
// The reference decodes its textures with image 0.23 (`image::io::Reader::open(..).decode()`,
// src/main.rs:1226-1232, 1273-1279) and only ever reads them through `get_pixel`, i.e. as 8-bit
// RGBA with row 0 on top; gamma, chromaticity and colour-profile chunks are ignored by that
// decoder and are ignored here. It never writes an image (it blits into a window,
// src/main.rs:1116-1122); rt_image_write_png exists so that a headless run has an output.
// Supported input: non-interlaced PNG, 8 bits per sample (grey, grey+alpha, RGB, RGBA) or a
// palette of 1/2/4/8 bits. Anything else is refused with RT_ERR_IO rather than guessed at.

#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "rt_types.h"

struct rt_image {
  uint32_t w = 0, h = 0;
  std::vector<uint8_t> rgba;
};

namespace {

uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3]; }

bool read_file(const char* path, std::vector<uint8_t>* out) {
  FILE* f = std::fopen(path, "rb");
  if (!f) return false;
  uint8_t buf[1 << 16];
  size_t got;
  while ((got = std::fread(buf, 1, sizeof buf, f)) > 0) out->insert(out->end(), buf, buf + got);
  std::fclose(f);
  return true;
}

int paeth(int a, int b, int c) {
  const int p = a + b - c, pa = p > a ? p - a : a - p, pb = p > b ? p - b : b - p, pc = p > c ? p - c : c - p;
  return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

int decode_png(const std::vector<uint8_t>& file, const char* path, rt_image* img) {
  static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
  if (file.size() < 8 + 25 || std::memcmp(file.data(), sig, 8) != 0) {
    rt_set_error("%s: not a PNG file", path);
    return RT_ERR_IO;
  }
  uint32_t w = 0, h = 0;
  int depth = 0, ctype = -1, interlace = 0;
  std::vector<uint8_t> idat, palette, trns;
  size_t pos = 8;
  bool end = false;
  while (!end && pos + 12 <= file.size()) {
    const uint32_t len = be32(&file[pos]);
    const uint8_t* tag = &file[pos + 4];
    if (pos + 12 + (size_t)len > file.size()) {
      rt_set_error("%s: truncated PNG chunk", path);
      return RT_ERR_IO;
    }
    const uint8_t* body = &file[pos + 8];
    if (!std::memcmp(tag, "IHDR", 4) && len == 13) {
      w = be32(body);
      h = be32(body + 4);
      depth = body[8];
      ctype = body[9];
      interlace = body[12];
    } else if (!std::memcmp(tag, "PLTE", 4)) {
      palette.assign(body, body + len);
    } else if (!std::memcmp(tag, "tRNS", 4)) {
      trns.assign(body, body + len);
    } else if (!std::memcmp(tag, "IDAT", 4)) {
      idat.insert(idat.end(), body, body + len);
    } else if (!std::memcmp(tag, "IEND", 4)) {
      end = true;
    }  // gAMA, cHRM, iCCP, sRGB, pHYs, text: ignored, as the reference's decoder ignores them
    pos += 12 + (size_t)len;
  }
  if (w == 0 || h == 0 || ctype < 0 || idat.empty()) {
    rt_set_error("%s: PNG without header or image data", path);
    return RT_ERR_IO;
  }
  if (interlace != 0) {
    rt_set_error("%s: interlaced PNG is not supported", path);
    return RT_ERR_IO;
  }
  int channels;
  switch (ctype) {
    case 0: channels = 1; break;
    case 2: channels = 3; break;
    case 3: channels = 1; break;
    case 4: channels = 2; break;
    case 6: channels = 4; break;
    default:
      rt_set_error("%s: unknown PNG colour type %d", path, ctype);
      return RT_ERR_IO;
  }
  const bool depth_ok = ctype == 3 ? (depth == 1 || depth == 2 || depth == 4 || depth == 8) : depth == 8;
  if (!depth_ok) {
    rt_set_error("%s: PNG colour type %d with %d bits per sample is not supported", path, ctype, depth);
    return RT_ERR_IO;
  }
  if (ctype == 3 && palette.size() < 3) {
    rt_set_error("%s: palette PNG without a palette", path);
    return RT_ERR_IO;
  }
  const size_t bpp = (size_t)(channels * depth + 7) / 8;              // bytes per complete pixel, >= 1
  const size_t stride = ((size_t)w * channels * depth + 7) / 8;       // bytes per scanline
  // deflate cannot expand more than about 1032 times: a header that promises more pixels than the
  // compressed data can hold is damaged, and must not decide how much memory is asked for
  if (w > (1u << 20) || h > (1u << 20) || (stride + 1) * (size_t)h > idat.size() * 1100u + 65536u) {
    rt_set_error("%s: PNG header (%ux%u) does not fit its %zu bytes of image data", path, w, h, idat.size());
    return RT_ERR_IO;
  }
  std::vector<uint8_t> raw((stride + 1) * (size_t)h);
  uLongf raw_len = (uLongf)raw.size();
  const int zrc = uncompress(raw.data(), &raw_len, idat.data(), (uLong)idat.size());
  if (zrc != Z_OK || raw_len != raw.size()) {
    rt_set_error("%s: PNG image data does not inflate to %zu bytes (zlib %d)", path, raw.size(), zrc);
    return RT_ERR_IO;
  }
  // undo the scanline filters in place
  std::vector<uint8_t> prev(stride, 0);
  for (uint32_t y = 0; y < h; ++y) {
    uint8_t* line = &raw[(stride + 1) * (size_t)y];
    const int filter = line[0];
    uint8_t* cur = line + 1;
    for (size_t i = 0; i < stride; ++i) {
      const int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0;
      int v = cur[i];
      switch (filter) {
        case 0: break;
        case 1: v += a; break;
        case 2: v += b; break;
        case 3: v += (a + b) / 2; break;
        case 4: v += paeth(a, b, c); break;
        default:
          rt_set_error("%s: bad PNG filter type %d", path, filter);
          return RT_ERR_IO;
      }
      cur[i] = (uint8_t)v;
    }
    std::memcpy(prev.data(), cur, stride);
  }
  img->w = w;
  img->h = h;
  img->rgba.resize((size_t)w * h * 4);
  for (uint32_t y = 0; y < h; ++y) {
    const uint8_t* cur = &raw[(stride + 1) * (size_t)y + 1];
    uint8_t* out = &img->rgba[(size_t)y * w * 4];
    for (uint32_t x = 0; x < w; ++x, out += 4) {
      switch (ctype) {
        case 0:
          out[0] = out[1] = out[2] = cur[x];
          out[3] = 255;
          break;
        case 2:
          out[0] = cur[x * 3];
          out[1] = cur[x * 3 + 1];
          out[2] = cur[x * 3 + 2];
          out[3] = 255;
          break;
        case 3: {
          const size_t bit = (size_t)x * depth;
          const unsigned idx = (cur[bit / 8] >> (8 - depth - (bit % 8))) & ((1u << depth) - 1u);
          if ((size_t)idx * 3 + 2 >= palette.size()) {
            rt_set_error("%s: palette index %u out of range", path, idx);
            return RT_ERR_IO;
          }
          out[0] = palette[idx * 3];
          out[1] = palette[idx * 3 + 1];
          out[2] = palette[idx * 3 + 2];
          out[3] = idx < trns.size() ? trns[idx] : 255;
          break;
        }
        case 4:
          out[0] = out[1] = out[2] = cur[x * 2];
          out[3] = cur[x * 2 + 1];
          break;
        default:
          std::memcpy(out, cur + (size_t)x * 4, 4);
          break;
      }
    }
  }
  return RT_OK;
}

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
  v.push_back((uint8_t)(x >> 24));
  v.push_back((uint8_t)(x >> 16));
  v.push_back((uint8_t)(x >> 8));
  v.push_back((uint8_t)x);
}
void put_chunk(std::vector<uint8_t>& out, const char tag[4], const std::vector<uint8_t>& body) {
  put_be32(out, (uint32_t)body.size());
  const size_t start = out.size();
  out.insert(out.end(), tag, tag + 4);
  out.insert(out.end(), body.begin(), body.end());
  put_be32(out, (uint32_t)crc32(0L, out.data() + start, (uInt)(out.size() - start)));
}

}  // namespace

extern "C" int rt_image_load_png(const char* path, rt_image** out) {
  if (!path || !out) {
    rt_set_error("rt_image_load_png: null argument");
    return RT_ERR_INVALID;
  }
  std::unique_ptr<rt_image> img(new (std::nothrow) rt_image());
  if (!img) return RT_ERR_OOM;
  try {
    std::vector<uint8_t> file;
    if (!read_file(path, &file)) {
      rt_set_error("rt_image_load_png: cannot open %s", path);
      return RT_ERR_IO;
    }
    int rc = decode_png(file, path, img.get());
    if (rc != RT_OK) return rc;
  } catch (const std::bad_alloc&) {
    rt_set_error("rt_image_load_png: out of memory decoding %s", path);
    return RT_ERR_OOM;
  }
  *out = img.release();
  return RT_OK;
}

extern "C" void rt_image_destroy(rt_image* img) { delete img; }

extern "C" int rt_image_info(const rt_image* img, uint32_t* width, uint32_t* height, const uint8_t** rgba) {
  if (!img) {
    rt_set_error("rt_image_info: null image");
    return RT_ERR_INVALID;
  }
  if (width) *width = img->w;
  if (height) *height = img->h;
  if (rgba) *rgba = img->rgba.data();
  return RT_OK;
}

extern "C" int rt_image_write_png(const char* path, const uint8_t* rgba, uint32_t width, uint32_t height) {
  if (!path || !rgba || width == 0 || height == 0) {
    rt_set_error("rt_image_write_png: null argument or empty image");
    return RT_ERR_INVALID;
  }
  try {
    const size_t stride = (size_t)width * 4;
    std::vector<uint8_t> raw((stride + 1) * (size_t)height);
    for (uint32_t y = 0; y < height; ++y) {
      raw[(stride + 1) * y] = 0;  // filter: none
      std::memcpy(&raw[(stride + 1) * y + 1], rgba + stride * y, stride);
    }
    uLongf zlen = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(zlen);
    if (compress2(z.data(), &zlen, raw.data(), (uLong)raw.size(), 6) != Z_OK) {
      rt_set_error("rt_image_write_png: deflate failed");
      return RT_ERR_IO;
    }
    z.resize(zlen);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, width);
    put_be32(ihdr, height);
    ihdr.insert(ihdr.end(), {8, 6, 0, 0, 0});
    put_chunk(out, "IHDR", ihdr);
    put_chunk(out, "IDAT", z);
    put_chunk(out, "IEND", {});
    FILE* f = std::fopen(path, "wb");
    if (!f) {
      rt_set_error("rt_image_write_png: cannot create %s", path);
      return RT_ERR_IO;
    }
    const bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    std::fclose(f);
    if (!ok) {
      rt_set_error("rt_image_write_png: short write to %s", path);
      return RT_ERR_IO;
    }
  } catch (const std::bad_alloc&) {
    rt_set_error("rt_image_write_png: out of memory");
    return RT_ERR_OOM;
  }
  return RT_OK;
}
