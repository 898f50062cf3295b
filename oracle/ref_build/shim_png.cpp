// shim_png.cpp — TEST INFRASTRUCTURE ONLY.  PNG decode / encode behind the lodepng entry points the
// reference's util/image.cpp calls (ReadPNG / WritePNG), on the system zlib.  See shims/lodepng/lodepng.h.
#include <lodepng/lodepng.h>

#include <zlib.h>

#include <cstdint>
#include <cstdlib>
#include <cstring>

namespace {
struct Png {
    unsigned w = 0, h = 0, depth = 0, ctype = 0, interlace = 0;
    std::vector<unsigned char> idat, plte;
};
uint32_t be32(const unsigned char *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
unsigned parse(const unsigned char *in, size_t n, Png *png, LodePNGState *st) {
    static const unsigned char sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (n < 33 || memcmp(in, sig, 8)) return 28;
    size_t pos = 8;
    bool haveHdr = false;
    while (pos + 12 <= n) {
        uint32_t len = be32(in + pos);
        const unsigned char *type = in + pos + 4, *data = in + pos + 8;
        if (pos + 12 + len > n) return 30;
        if (!memcmp(type, "IHDR", 4)) {
            png->w = be32(data); png->h = be32(data + 4); png->depth = data[8]; png->ctype = data[9]; png->interlace = data[12];
            haveHdr = true;
        } else if (!memcmp(type, "PLTE", 4)) png->plte.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) png->idat.insert(png->idat.end(), data, data + len);
        else if (!memcmp(type, "sRGB", 4)) { if (st) st->info_png.srgb_defined = 1; }
        else if (!memcmp(type, "gAMA", 4)) { if (st) { st->info_png.gama_defined = 1; st->info_png.gama_gamma = be32(data); } }
        else if (!memcmp(type, "iCCP", 4)) { if (st) st->info_png.iccp_defined = 1; }
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + len;
    }
    return haveHdr ? 0 : 29;
}
int channels(unsigned ctype) { return ctype == 0 ? 1 : ctype == 2 ? 3 : ctype == 3 ? 1 : ctype == 4 ? 2 : 4; }
int paeth(int a, int b, int c) {
    int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}
// raw samples: 16-bit big-endian pairs kept as they are; sub-byte depths unpacked to one byte per sample
unsigned unfilter(const Png &png, std::vector<unsigned char> *samples) {
    if (png.interlace) return 1001;
    const int nc = channels(png.ctype);
    const size_t bpp = std::max<size_t>(1, (size_t)nc * png.depth / 8), stride = ((size_t)png.w * nc * png.depth + 7) / 8;
    std::vector<unsigned char> raw((stride + 1) * png.h);
    uLongf rawLen = raw.size();
    if (uncompress(raw.data(), &rawLen, png.idat.data(), png.idat.size()) != Z_OK || rawLen != raw.size()) return 1002;
    std::vector<unsigned char> prev(stride, 0), cur(stride);
    samples->clear();
    for (unsigned y = 0; y < png.h; ++y) {
        const unsigned char *row = raw.data() + (stride + 1) * y;
        const int ft = row[0];
        for (size_t i = 0; i < stride; ++i) {
            int a = i >= bpp ? cur[i - bpp] : 0, b = prev[i], c = i >= bpp ? prev[i - bpp] : 0, x = row[1 + i];
            int v = ft == 0 ? x : ft == 1 ? x + a : ft == 2 ? x + b : ft == 3 ? x + ((a + b) >> 1) : x + paeth(a, b, c);
            cur[i] = (unsigned char)v;
        }
        if (png.depth >= 8) samples->insert(samples->end(), cur.begin(), cur.end());
        else
            for (size_t s = 0; s < (size_t)png.w * nc; ++s) {
                size_t bit = s * png.depth;
                samples->push_back((cur[bit >> 3] >> (8 - png.depth - (bit & 7))) & ((1u << png.depth) - 1));
            }
        prev = cur;
    }
    return 0;
}
}  // namespace

const char *lodepng_error_text(unsigned code) {
    switch (code) {
    case 0: return "no error";
    case 28: return "incorrect PNG signature";
    case 1001: return "interlaced PNG is not supported by the oracle build's decoder";
    case 1002: return "zlib could not inflate the image data";
    case 1003: return "unsupported PNG conversion in the oracle build's decoder";
    default: return "malformed PNG";
    }
}
unsigned lodepng_inspect(unsigned *w, unsigned *h, LodePNGState *state, const unsigned char *in, size_t insize) {
    Png png;
    if (unsigned e = parse(in, insize, &png, state)) return e;
    *w = png.w; *h = png.h;
    state->info_png.color.colortype = (LodePNGColorType)png.ctype;
    state->info_png.color.bitdepth = png.depth;
    return 0;
}
unsigned lodepng::decode(std::vector<unsigned char> &out, unsigned &w, unsigned &h, const unsigned char *in, size_t insize,
                         LodePNGColorType want, unsigned wantDepth) {
    Png png;
    if (unsigned e = parse(in, insize, &png, nullptr)) return e;
    std::vector<unsigned char> s;
    if (unsigned e = unfilter(png, &s)) return e;
    w = png.w; h = png.h;
    const int nc = channels(png.ctype), wnc = want == LCT_GREY ? 1 : want == LCT_RGB ? 3 : want == LCT_RGBA ? 4 : 0;
    if (!wnc) return 1003;
    const size_t npix = (size_t)w * h;
    // every sample as 16 bits (8-bit v -> v * 257, sub-byte v -> scaled to 8 bits first), then narrowed
    auto sample16 = [&](size_t i) -> unsigned {
        if (png.depth == 16) return (unsigned)s[2 * i] << 8 | s[2 * i + 1];
        unsigned v = s[i];
        if (png.depth < 8 && png.ctype != 3) v = v * 255u / ((1u << png.depth) - 1);
        return v * 257u;
    };
    out.clear();
    auto put = [&](unsigned v16) {
        if (wantDepth == 16) { out.push_back(v16 >> 8); out.push_back(v16 & 255); }
        else out.push_back(png.depth == 16 ? (v16 >> 8) : (v16 / 257u));
    };
    for (size_t p = 0; p < npix; ++p) {
        unsigned rgba[4] = {0, 0, 0, 65535};
        if (png.ctype == 3) {
            unsigned idx = s[p];
            if (3 * idx + 2 >= png.plte.size()) return 1003;
            for (int c = 0; c < 3; ++c) rgba[c] = png.plte[3 * idx + c] * 257u;
        } else if (nc <= 2) {
            rgba[0] = rgba[1] = rgba[2] = sample16(p * nc);
            if (nc == 2) rgba[3] = sample16(p * nc + 1);
        } else {
            for (int c = 0; c < nc; ++c) rgba[c] = sample16(p * nc + c);
        }
        if (want == LCT_GREY) {
            if (nc > 2 || png.ctype == 3) return 1003;  // ReadPNG only asks for grey from grey sources
            put(rgba[0]);
        } else
            for (int c = 0; c < wnc; ++c) put(rgba[c]);
    }
    return 0;
}
unsigned lodepng::decode(std::vector<unsigned char> &out, unsigned &w, unsigned &h, LodePNGState &state, const unsigned char *in, size_t insize) {
    return decode(out, w, h, in, insize, state.info_raw.colortype, state.info_raw.bitdepth);
}

unsigned lodepng_encode_memory(unsigned char **out, size_t *outsize, const unsigned char *image, unsigned w, unsigned h,
                               LodePNGColorType colortype, unsigned bitdepth) {
    if (bitdepth != 8 || (colortype != LCT_GREY && colortype != LCT_RGB && colortype != LCT_RGBA)) return 1003;
    const int nc = colortype == LCT_GREY ? 1 : colortype == LCT_RGB ? 3 : 4;
    std::vector<unsigned char> raw;
    for (unsigned y = 0; y < h; ++y) {
        raw.push_back(0);
        raw.insert(raw.end(), image + (size_t)y * w * nc, image + (size_t)(y + 1) * w * nc);
    }
    uLongf zl = compressBound(raw.size());
    std::vector<unsigned char> z(zl);
    if (compress2(z.data(), &zl, raw.data(), raw.size(), 6) != Z_OK) return 1002;
    std::vector<unsigned char> png = {137, 80, 78, 71, 13, 10, 26, 10};
    auto chunk = [&](const char *type, const unsigned char *d, size_t n) {
        unsigned char l[4] = {(unsigned char)(n >> 24), (unsigned char)(n >> 16), (unsigned char)(n >> 8), (unsigned char)n};
        png.insert(png.end(), l, l + 4);
        size_t start = png.size();
        png.insert(png.end(), type, type + 4);
        png.insert(png.end(), d, d + n);
        uint32_t crc = crc32(0, png.data() + start, (uInt)(png.size() - start));
        unsigned char c[4] = {(unsigned char)(crc >> 24), (unsigned char)(crc >> 16), (unsigned char)(crc >> 8), (unsigned char)crc};
        png.insert(png.end(), c, c + 4);
    };
    unsigned char hdr[13] = {(unsigned char)(w >> 24), (unsigned char)(w >> 16), (unsigned char)(w >> 8), (unsigned char)w,
                             (unsigned char)(h >> 24), (unsigned char)(h >> 16), (unsigned char)(h >> 8), (unsigned char)h,
                             8, (unsigned char)colortype, 0, 0, 0};
    chunk("IHDR", hdr, 13);
    chunk("IDAT", z.data(), zl);
    chunk("IEND", nullptr, 0);
    *out = (unsigned char *)malloc(png.size());
    memcpy(*out, png.data(), png.size());
    *outsize = png.size();
    return 0;
}
