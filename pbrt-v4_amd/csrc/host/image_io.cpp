// image_io.cpp — image output for the film (util/image.cpp:WritePFM/ReadPFM semantics: bottom-to-top
// scanlines, little-endian scale -1).  EXR is written uncompressed, 32-bit float, channels B,G,R.
#include "scene.h"

#include <zlib.h>

#include <atomic>
#include <cmath>
#include <cstdio>
#include <algorithm>
#include <cstring>
#include <fstream>
#include <sstream>

namespace wf {

[[noreturn]] static void Die(const std::string &loc, const std::string &msg) { throw SceneError("Error: " + loc + (loc.empty() ? "" : ": ") + msg); }
bool ReadPFM(const std::string &path, std::vector<float> *rgb, int *w, int *h);

// ---------------------------------------------------------------------------------------------------------------
// ColorEncoding (util/color.h:402-477, util/color.cpp:183-279)
static const float *SRGBToLinearLUT() {
    static std::vector<float> lut;
    if (lut.empty()) {
        std::ifstream f(SpectralData::Get().DataDir() + "/srgb_to_linear_lut.txt");
        std::string line;
        while (std::getline(f, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::istringstream ls(line);
            std::string tok;
            while (ls >> tok) lut.push_back((float)strtod(tok.c_str(), nullptr));
        }
        if (lut.size() != 256) Die("", "data/srgb_to_linear_lut.txt: expected 256 entries (tools/extract_srgb_lut.py)");
    }
    return lut.data();
}
static float Poly(float, float c) { return c; }
template <typename... A> static float Poly(float t, float c, A... rest) { return std::fma(t, Poly(t, rest...), c); }  // EvaluatePolynomial (util/math.h)
// LinearToSRGB / LinearToSRGB8 (util/color.h:494-518)
static float LinearToSRGB(float value) {
    if (value <= 0.0031308f) return 12.92f * value;
    float sqrtValue = std::sqrt(std::max(0.f, value));
    float p = Poly(sqrtValue, -0.0016829072605308378f, 0.03453868659826638f, 0.7642611304733891f, 2.0041169284241644f,
                   0.7551545191665577f, -0.016202083165206348f);
    float q = Poly(sqrtValue, 4.178892964897981e-7f, -0.00004375359692957097f, 0.03467195408529984f, 0.6085338522168684f,
                   1.8970238036421054f, 1.f);
    return p / q * value;
}
// SRGBToLinear (util/color.h:520-531)
static float SRGBToLinear(float value) {
    if (value <= 0.04045f) return value * (1 / 12.92f);
    float p = Poly(value, -0.0163933279112946f, -0.7386328024653209f, -11.199318357635072f, -47.46726633009393f, -36.04572663838034f);
    float q = Poly(value, -0.004261480793199332f, -19.140923959601675f, -59.096406619244426f, -18.225745396846637f, 1.f);
    return p / q * value;
}
ColorEnc ColorEnc::Parse(const std::string &name) {
    ColorEnc e;
    if (name == "linear") { e.kind = 0; return e; }
    if (name == "sRGB") { e.kind = 1; return e; }
    std::istringstream ss(name);
    std::string a, b, c;
    ss >> a >> b;
    if (a != "gamma" || b.empty() || (ss >> c)) Die("", name + ": expected \"gamma <value>\" for color encoding");
    e.kind = 2;
    e.gamma = (float)atof(b.c_str());
    if (e.gamma == 0) Die("", b + ": unable to parse gamma value");
    return e;
}
float ColorEnc::ToLinear(uint8_t v) const {
    if (kind == 0) return v / 255.f;
    if (kind == 1) return SRGBToLinearLUT()[v];
    return std::pow(float(v) / 255.f, gamma);   // GammaColorEncoding::applyLUT
}
uint8_t ColorEnc::FromLinear(float v) const {
    auto clampf = [](float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); };
    if (kind == 0) return uint8_t(clampf(v * 255.f + 0.5f, 0, 255));
    if (kind == 1) {
        if (v <= 0) return 0;
        if (v >= 1) return 255;
        return uint8_t(clampf(std::round(255.f * LinearToSRGB(v) + 0.f), 0, 255));
    }
    // GammaColorEncoding::FromLinear: a 1024-entry table of Clamp(255 pow(i / 1023, 1 / gamma) + .5, 0, 255) stored as floats
    int i = (int)clampf(v * 1023.f, 0, 1023);
    float t = clampf(255.f * std::pow(float(i) / 1023.f, 1.f / gamma) + .5f, 0, 255);
    return uint8_t(t);
}
float ColorEnc::ToFloatLinear(float v) const { return kind == 0 ? v : kind == 1 ? SRGBToLinear(v) : std::pow(v, gamma); }

float HostImage::Quantize(float v) const {
    if (format == U256) return enc.ToLinear(enc.FromLinear(v));
    if (format == Half) return RoundToHalf(v);
    return v;
}
uint32_t HostImage::QuantizeCode(float v) const {
    if (format == U256) return enc.FromLinear(v);
    if (format == Half) return FloatToHalfBits(v);
    return 0;
}
void HostImage::SelectChannels(int first, int count) {
    if (first == 0 && count == nc) return;
    const size_t n = (size_t)w * h;
    if (format == U256) {
        std::vector<uint8_t> o(n * count);
        for (size_t i = 0; i < n; ++i) for (int c = 0; c < count; ++c) o[i * count + c] = p8[i * nc + first + c];
        p8.swap(o);
    } else {
        std::vector<float> o(n * count);
        for (size_t i = 0; i < n; ++i) for (int c = 0; c < count; ++c) o[i * count + c] = p32[i * nc + first + c];
        p32.swap(o);
    }
    nc = count;
}

// ---------------------------------------------------------------------------------------------------------------
// PNG: chunk walk, zlib inflate, scanline un-filtering (PNG specification 1.2, sections 5, 6, 9).  What the decoded
// samples become follows ReadPNG (util/image.cpp:1260-1376): grey(+alpha) -> "Y", everything else -> R G B (A only for
// colour type 6); 8 bits stay bytes with the encoding, 16 bits become halves of ToFloatLinear(v / 65535).
namespace {
struct PngFile {
    uint32_t w = 0, h = 0;
    int depth = 0, ctype = 0, interlace = 0;
    std::vector<uint8_t> idat, plte;
};
uint32_t BE32(const uint8_t *p) { return (uint32_t)p[0] << 24 | (uint32_t)p[1] << 16 | (uint32_t)p[2] << 8 | p[3]; }
int PaethPredictor(int a, int b, int c) {
    int p = a + b - c, pa = std::abs(p - a), pb = std::abs(p - b), pc = std::abs(p - c);
    if (pa <= pb && pa <= pc) return a;
    return pb <= pc ? b : c;
}
}  // namespace
static void ReadPNG(const std::string &path, const ColorEnc &encIn, HostImage *img) {
    std::vector<uint8_t> file;
    {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) Die("", path + ": unable to open file");
        fseek(f, 0, SEEK_END);
        long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        file.resize(n > 0 ? n : 0);
        if (n > 0 && fread(file.data(), 1, n, f) != (size_t)n) { fclose(f); Die("", path + ": read error"); }
        fclose(f);
    }
    static const uint8_t sig[8] = {137, 80, 78, 71, 13, 10, 26, 10};
    if (file.size() < 33 || memcmp(file.data(), sig, 8)) Die("", path + ": incorrect PNG signature, it's no PNG or corrupted");
    PngFile png;
    for (size_t pos = 8; pos + 12 <= file.size();) {
        const uint32_t len = BE32(&file[pos]);
        const uint8_t *type = &file[pos + 4], *data = &file[pos + 8];
        if (pos + 12 + (size_t)len > file.size()) Die("", path + ": truncated PNG chunk");
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            png.w = BE32(data); png.h = BE32(data + 4); png.depth = data[8]; png.ctype = data[9]; png.interlace = data[12];
        } else if (!memcmp(type, "PLTE", 4)) png.plte.assign(data, data + len);
        else if (!memcmp(type, "IDAT", 4)) png.idat.insert(png.idat.end(), data, data + len);
        else if (!memcmp(type, "IEND", 4)) break;
        pos += 12 + (size_t)len;
    }
    const int ct = png.ctype;
    const int srcNc = ct == 0 ? 1 : ct == 2 ? 3 : ct == 3 ? 1 : ct == 4 ? 2 : ct == 6 ? 4 : 0;
    if (!png.w || !png.h || png.w > 65536u || png.h > 65536u || !srcNc || (png.depth != 1 && png.depth != 2 && png.depth != 4 && png.depth != 8 && png.depth != 16))
        Die("", path + ": malformed PNG header (dimensions / colour type / bit depth)");
    if (png.interlace > 1) Die("", path + ": malformed PNG header (interlace method)");
    const size_t bpp = std::max<size_t>(1, (size_t)srcNc * png.depth / 8), stride = ((size_t)png.w * srcNc * png.depth + 7) / 8;
    const size_t bitsPerPixel = (size_t)srcNc * png.depth;
    // un-filter `rows` scanlines of `rowBytes` bytes in place (the filter byte stays in front of each row)
    auto unfilter = [&](uint8_t *base, size_t rows, size_t rowBytes) {
        for (size_t y = 0; y < rows; ++y) {
            uint8_t *row = base + (rowBytes + 1) * y + 1;
            const uint8_t *up = y ? row - (rowBytes + 1) : nullptr;
            const int ft = row[-1];
            if (ft > 4) Die("", path + ": corrupt PNG filter type");
            for (size_t i = 0; i < rowBytes; ++i) {
                const int a = i >= bpp ? row[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
                int pred = 0;
                switch (ft) {
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: pred = PaethPredictor(a, b, c); break;
                }
                row[i] = (uint8_t)(row[i] + pred);
            }
        }
    };
    std::vector<uint8_t> raw((stride + 1) * png.h);
    if (!png.interlace) {
        uLongf rawLen = raw.size();
        if (uncompress(raw.data(), &rawLen, png.idat.data(), png.idat.size()) != Z_OK || rawLen != raw.size())
            Die("", path + ": corrupt PNG image data");
        unfilter(raw.data(), png.h, stride);
    } else {
        // Adam7 (PNG specification 1.2, section 8.2): seven reduced images, each filtered on its own, concatenated in one zlib stream;
        // their pixels are scattered into the non-interlaced layout the code below reads
        static const int xs[7] = {0, 4, 0, 2, 0, 1, 0}, ys[7] = {0, 0, 4, 0, 2, 0, 1}, dx[7] = {8, 8, 4, 4, 2, 2, 1}, dy[7] = {8, 8, 8, 4, 4, 2, 2};
        size_t total = 0, pw[7], ph[7], prow[7];
        for (int p = 0; p < 7; ++p) {
            pw[p] = png.w > (uint32_t)xs[p] ? (png.w - xs[p] + dx[p] - 1) / dx[p] : 0;
            ph[p] = png.h > (uint32_t)ys[p] ? (png.h - ys[p] + dy[p] - 1) / dy[p] : 0;
            prow[p] = (pw[p] * bitsPerPixel + 7) / 8;
            if (pw[p] && ph[p]) total += (prow[p] + 1) * ph[p];
        }
        std::vector<uint8_t> passes(total);
        uLongf rawLen = passes.size();
        if (uncompress(passes.data(), &rawLen, png.idat.data(), png.idat.size()) != Z_OK || rawLen != passes.size())
            Die("", path + ": corrupt PNG image data");
        size_t off = 0;
        for (int p = 0; p < 7; ++p) {
            if (!pw[p] || !ph[p]) continue;
            unfilter(&passes[off], ph[p], prow[p]);
            for (size_t py = 0; py < ph[p]; ++py) {
                const uint8_t *src = &passes[off + (prow[p] + 1) * py + 1];
                uint8_t *dst = &raw[(stride + 1) * (ys[p] + py * dy[p]) + 1];
                for (size_t px = 0; px < pw[p]; ++px) {
                    const size_t x = xs[p] + px * dx[p];
                    if (bitsPerPixel >= 8) memcpy(dst + x * (bitsPerPixel / 8), src + px * (bitsPerPixel / 8), bitsPerPixel / 8);
                    else {
                        const size_t sb = px * bitsPerPixel, db = x * bitsPerPixel;
                        const unsigned v = (src[sb >> 3] >> (8 - bitsPerPixel - (sb & 7))) & ((1u << bitsPerPixel) - 1);
                        dst[db >> 3] = (uint8_t)((dst[db >> 3] & ~(((1u << bitsPerPixel) - 1) << (8 - bitsPerPixel - (db & 7)))) | (v << (8 - bitsPerPixel - (db & 7))));
                    }
                }
            }
            off += (prow[p] + 1) * ph[p];
        }
    }
    const bool grey = ct == 0 || ct == 4, hasAlpha = ct == 6, wide = png.depth == 16;
    const int nc = grey ? 1 : (hasAlpha ? 4 : 3);
    img->w = (int)png.w; img->h = (int)png.h; img->nc = nc;
    img->enc = encIn;
    img->format = wide ? HostImage::Half : HostImage::U256;
    const size_t npix = (size_t)png.w * png.h;
    if (wide) img->p32.resize(npix * nc); else img->p8.resize(npix * nc);
    // sample s of row y: 16-bit big-endian, a byte, or depth bits (MSB first) scaled to a byte (palette indices are not scaled)
    auto sample = [&](uint32_t y, size_t s) -> unsigned {
        const uint8_t *row = &raw[(stride + 1) * y + 1];
        if (png.depth == 16) return (unsigned)row[2 * s] << 8 | row[2 * s + 1];
        if (png.depth == 8) return row[s];
        const size_t bit = s * png.depth;
        unsigned v = (row[bit >> 3] >> (8 - png.depth - (bit & 7))) & ((1u << png.depth) - 1);
        return ct == 3 ? v : v * 255u / ((1u << png.depth) - 1);
    };
    for (uint32_t y = 0; y < png.h; ++y)
        for (uint32_t x = 0; x < png.w; ++x) {
            unsigned v[4] = {0, 0, 0, 0};
            if (ct == 3) {
                const unsigned idx = sample(y, x);
                if (3 * (size_t)idx + 2 >= png.plte.size()) Die("", path + ": PNG palette index out of range");
                for (int c = 0; c < 3; ++c) v[c] = png.plte[3 * idx + c];
            } else if (grey) v[0] = sample(y, (size_t)x * srcNc);
            else for (int c = 0; c < nc; ++c) v[c] = sample(y, (size_t)x * srcNc + c);
            const size_t o = ((size_t)y * png.w + x) * nc;
            for (int c = 0; c < nc; ++c) {
                if (wide) img->p32[o + c] = RoundToHalf(encIn.ToFloatLinear(v[c] / 65535.f));
                else img->p8[o + c] = (uint8_t)v[c];
            }
        }
}

static void ReadPFMImage(const std::string &path, HostImage *img) {
    std::vector<float> rgb;
    int w = 0, h = 0;
    if (!ReadPFM(path, &rgb, &w, &h)) Die("", path + ": unable to read PFM file");
    bool grey = false;
    { FILE *f = fopen(path.c_str(), "rb"); char m[3] = {0, 0, 0}; if (f) { if (fread(m, 1, 2, f) == 2) grey = m[1] == 'f'; fclose(f); } }
    img->format = HostImage::Float;
    img->w = w; img->h = h; img->nc = grey ? 1 : 3;
    img->p32.resize((size_t)w * h * img->nc);
    for (size_t i = 0; i < (size_t)w * h; ++i)
        for (int c = 0; c < img->nc; ++c) img->p32[i * img->nc + c] = rgb[i * 3 + c];
}


// ---------------------------------------------------------------------------------------------------------------
// OpenEXR, single-part scan-line files (OpenEXR file layout: magic, version, attribute list, line-offset table, chunks;
// compression NONE / RLE / ZIPS / ZIP — the zlib-based ones; PIZ, PXR24, B44 and DWA files are refused with a message).
// What the pixels become follows ReadEXR (util/image.cpp:1055-1166): all channels of one type, half -> PixelFormat::Half,
// float -> PixelFormat::Float; here the channels are put in the order Y | R G B | R G B A the image-map and light code selects.
static float HalfBitsToFloat(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1f, man = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) bits = sign;
        else {  // subnormal half: man * 2^-24, exact in float
            float v = (float)man * 5.9604644775390625e-08f;
            memcpy(&bits, &v, 4);
            bits |= sign;
        }
    } else if (exp == 31) bits = sign | 0x7f800000u | (man << 13);
    else bits = sign | ((exp + 112) << 23) | (man << 13);
    float f;
    memcpy(&f, &bits, 4);
    return f;
}
static void ReadEXR(const std::string &path, HostImage *img) {
    std::vector<uint8_t> file;
    {
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) Die("", "Unable to read image file \"" + path + "\": cannot open");
        fseek(f, 0, SEEK_END);
        long n = ftell(f);
        fseek(f, 0, SEEK_SET);
        file.resize(n > 0 ? n : 0);
        if (n > 0 && fread(file.data(), 1, n, f) != (size_t)n) { fclose(f); Die("", path + ": read error"); }
        fclose(f);
    }
    auto fail = [&](const std::string &why) { Die("", "Unable to read image file \"" + path + "\": " + why); };
    size_t pos = 0;
    auto need = [&](size_t n) { if (pos > file.size() || n > file.size() - pos) fail("file is truncated"); };
    auto rd32 = [&]() { need(4); uint32_t v; memcpy(&v, &file[pos], 4); pos += 4; return v; };
    auto rdStr = [&]() { std::string r; while (true) { need(1); char c = (char)file[pos++]; if (!c) break; r.push_back(c); } return r; };
    if (rd32() != 20000630u) fail("not an OpenEXR file");
    const uint32_t version = rd32();
    const bool tiled = (version & 0x200u) != 0;   // single-part tiled file: level (0, 0) is read (ONE_LEVEL, or the finest level of a MIP / RIP map)
    if (version & 0x1800u) fail("deep / multi-part OpenEXR files are not supported by this build");
    struct Chan { std::string name; int type; };
    std::vector<Chan> chans;
    int compression = -1, dw[4] = {0, 0, -1, -1}, lineOrder = 0;
    uint32_t tileW = 0, tileH = 0;
    while (true) {
        std::string name = rdStr();
        if (name.empty()) break;
        std::string type = rdStr();
        const uint32_t size = rd32();
        need(size);
        const uint8_t *v = &file[pos];
        if (name == "channels") {
            size_t p = 0;
            while (p < size && v[p]) {
                Chan c;
                while (p < size && v[p]) c.name.push_back((char)v[p++]);
                ++p;
                if (p > size || size - p < 16) fail("malformed channel list");
                int32_t t; memcpy(&t, v + p, 4);
                c.type = t;
                int32_t xs, ys; memcpy(&xs, v + p + 8, 4); memcpy(&ys, v + p + 12, 4);
                if (xs != 1 || ys != 1) fail("sub-sampled channels are not supported by this build");
                p += 16;
                chans.push_back(c);
            }
        } else if (name == "compression") { if (size != 1) fail("malformed compression attribute"); compression = v[0]; }
        else if (name == "dataWindow") { if (size != 16) fail("malformed dataWindow attribute"); memcpy(dw, v, 16); }
        else if (name == "lineOrder") { if (size != 1) fail("malformed lineOrder attribute"); lineOrder = v[0]; }
        else if (name == "tiles") {   // tiledesc: xSize, ySize, mode (level mode | rounding mode << 4)
            if (size != 9) fail("malformed tiles attribute");
            memcpy(&tileW, v, 4); memcpy(&tileH, v + 4, 4);
            if ((v[8] & 0xf) > 2) fail("malformed tiles attribute (level mode)");
        }
        else if (name == "chromaticities" && size >= 32) {
            // RGBColorSpace::Lookup (util/colorspace.cpp): this build's image maps are sRGB / Rec.709
            float c[8]; memcpy(c, v, 32);
            const float srgb[8] = {.64f, .33f, .3f, .6f, .15f, .06f, .3127f, .3290f};
            for (int i = 0; i < 8; ++i) if (std::abs(c[i] - srgb[i]) > 1e-3f) fail("only sRGB / Rec.709 chromaticities are supported by this build");
        }
        pos += size;
    }
    (void)lineOrder;
    const long long wl = (long long)dw[2] - dw[0] + 1, hl = (long long)dw[3] - dw[1] + 1;
    if (wl <= 0 || hl <= 0 || wl > 65536 || hl > 65536 || chans.empty() || chans.size() > 64) fail("malformed header (data window / channel list)");
    const int w = (int)wl, h = (int)hl;
    for (const Chan &c : chans) {
        if (c.type != chans[0].type) fail("images with multiple channel types are not supported");
        if (c.type != 1 && c.type != 2) fail("only half and float channels are supported");
    }
    const bool isHalf = chans[0].type == 1;
    const int bytesPer = isHalf ? 2 : 4;
    int linesPerChunk;
    switch (compression) {
    case 0: case 1: case 2: linesPerChunk = 1; break;   // NONE, RLE, ZIPS
    case 3: linesPerChunk = 16; break;                   // ZIP
    default: fail("compression method " + std::to_string(compression) + " (PIZ / PXR24 / B44 / DWA) is not supported by this build (NONE, RLE, ZIPS, ZIP are)"); return;
    }
    // destination channel of every file channel (the file stores them in alphabetical order)
    auto find = [&](const char *n) { for (size_t i = 0; i < chans.size(); ++i) if (chans[i].name == n) return (int)i; return -1; };
    int src[4] = {-1, -1, -1, -1}, nc;
    if (find("R") >= 0 && find("G") >= 0 && find("B") >= 0) { src[0] = find("R"); src[1] = find("G"); src[2] = find("B"); src[3] = find("A"); nc = src[3] >= 0 ? 4 : 3; }
    else if (chans.size() == 1) { src[0] = 0; nc = 1; }
    else if (find("Y") >= 0) { src[0] = find("Y"); nc = 1; }
    else { fail("image doesn't have R, G, and B channels"); return; }
    img->format = isHalf ? HostImage::Half : HostImage::Float;
    img->w = w; img->h = h; img->nc = nc;
    img->p32.assign((size_t)w * h * nc, 0.f);
    std::vector<uint8_t> raw, tmp;
    // one compressed block -> `expect` raw bytes (rows of channels); NONE / RLE / ZIPS / ZIP (OpenEXR file layout, "scan line / tile blocks")
    auto decodeBlock = [&](const uint8_t *data, size_t dataSize, size_t expect) {
        raw.resize(expect);
        if (dataSize == expect) { memcpy(raw.data(), data, expect); return; }   // stored uncompressed (also when compression did not help)
        if (compression == 0) fail("chunk size does not match the data window");
        tmp.resize(expect);
        if (compression == 1) {
            // RLE: a signed count byte; negative = that many literal bytes, otherwise count + 1 copies of the next byte
            size_t o = 0, i = 0;
            while (i < dataSize) {
                int c = (int8_t)data[i++];
                if (c < 0) { size_t n = (size_t)-c; if (i + n > dataSize || o + n > expect) fail("corrupt RLE data"); memcpy(&tmp[o], &data[i], n); o += n; i += n; }
                else { size_t n = (size_t)c + 1; if (i >= dataSize || o + n > expect) fail("corrupt RLE data"); memset(&tmp[o], data[i++], n); o += n; }
            }
            if (o != expect) fail("corrupt RLE data");
        } else {
            uLongf outLen = expect;
            if (uncompress(tmp.data(), &outLen, data, dataSize) != Z_OK || outLen != expect) fail("corrupt zip data");
        }
        // undo the byte predictor, then re-interleave the two halves
        for (size_t i = 1; i < expect; ++i) tmp[i] = (uint8_t)(tmp[i - 1] + tmp[i] - 128);
        const size_t half = (expect + 1) / 2;
        for (size_t i = 0; i < expect; ++i) raw[i] = (i & 1) ? tmp[half + i / 2] : tmp[i / 2];
    };
    // rows of `bw` pixels per channel -> the image rectangle at (x0, y0)
    auto scatter = [&](int x0, int y0, int bw, int nLines) {
        const size_t rowBytes = (size_t)bw * bytesPer * chans.size();
        for (int l = 0; l < nLines; ++l) {
            const int y = y0 + l;
            const uint8_t *line = raw.data() + rowBytes * l;
            for (int c = 0; c < nc; ++c) {
                const uint8_t *cp = line + (size_t)src[c] * bw * bytesPer;
                for (int x = 0; x < bw; ++x) {
                    float v;
                    if (isHalf) { uint16_t hb; memcpy(&hb, cp + 2 * (size_t)x, 2); v = HalfBitsToFloat(hb); }
                    else memcpy(&v, cp + 4 * (size_t)x, 4);
                    img->p32[((size_t)y * w + x0 + x) * nc + c] = v;
                }
            }
        }
    };
    const size_t tablePos = pos;
    if (tiled) {
        if (tileW == 0 || tileH == 0 || tileW > 65536u || tileH > 65536u) fail("malformed tiles attribute");
        const int ntx = (w + (int)tileW - 1) / (int)tileW, nty = (h + (int)tileH - 1) / (int)tileH;
        // the offset table lists the tiles of level (0, 0) first (then the coarser levels of a MIP / RIP map, which are not read)
        need((size_t)8 * ntx * nty);
        std::vector<char> seen((size_t)ntx * nty, 0);
        for (int t = 0; t < ntx * nty; ++t) {
            uint64_t off; memcpy(&off, &file[tablePos + 8 * (size_t)t], 8);
            if (off > file.size() || file.size() - off < 20) fail("tile offset out of range");
            int32_t tx, ty, lx, ly, dataSize;
            memcpy(&tx, &file[off], 4); memcpy(&ty, &file[off + 4], 4); memcpy(&lx, &file[off + 8], 4); memcpy(&ly, &file[off + 12], 4); memcpy(&dataSize, &file[off + 16], 4);
            if (lx != 0 || ly != 0 || tx < 0 || ty < 0 || tx >= ntx || ty >= nty || seen[(size_t)ty * ntx + tx]) fail("malformed tile header");
            if (dataSize < 0 || (size_t)dataSize > file.size() - off - 20) fail("tile size out of range");
            seen[(size_t)ty * ntx + tx] = 1;
            const int x0 = tx * (int)tileW, y0 = ty * (int)tileH;
            const int bw = std::min((int)tileW, w - x0), bh = std::min((int)tileH, h - y0);
            decodeBlock(&file[off + 20], (size_t)dataSize, (size_t)bw * bytesPer * chans.size() * bh);
            scatter(x0, y0, bw, bh);
        }
        return;
    }
    const int nChunks = (h + linesPerChunk - 1) / linesPerChunk;
    need((size_t)8 * nChunks);
    const size_t lineBytes = (size_t)w * bytesPer * chans.size();
    for (int ck = 0; ck < nChunks; ++ck) {
        uint64_t off; memcpy(&off, &file[tablePos + 8 * (size_t)ck], 8);
        if (off > file.size() || file.size() - off < 8) fail("chunk offset out of range");
        int32_t y0, dataSize;
        memcpy(&y0, &file[off], 4); memcpy(&dataSize, &file[off + 4], 4);
        if (dataSize < 0 || (size_t)dataSize > file.size() - off - 8) fail("chunk size out of range");
        if (y0 < dw[1] || y0 > dw[3]) fail("chunk outside the data window");
        const int nLines = std::min(linesPerChunk, dw[3] - y0 + 1);
        decodeBlock(&file[off + 8], (size_t)dataSize, lineBytes * nLines);
        scatter(0, y0 - dw[1], w, nLines);
    }
}

void ReadImage(const std::string &path, const ColorEnc &enc, HostImage *img) {
    *img = HostImage();
    const size_t dot = path.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : path.substr(dot + 1);
    for (char &c : ext) c = (char)tolower(c);
    if (ext == "pfm") ReadPFMImage(path, img);
    else if (ext == "png") ReadPNG(path, enc, img);
    else if (ext == "exr") ReadEXR(path, img);
    else if (ext == "qoi") ReadQOI(path, img);
    else if (ext == "hdr") ReadHDR(path, img);
    else if (ext == "tga") ReadTGA(path, img);   // (the reference reaches TGA through stb_image's catch-all stbi_load, util/image.cpp:888-916)
    else Die("", path + ": no support for reading images with this extension (this build reads .pfm, .png, .exr, .qoi, .hdr and .tga)");
}

bool WritePFM(const std::string &path, const float *rgb, int w, int h) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    fprintf(f, "PF\n%d %d\n-1\n", w, h);
    for (int y = h - 1; y >= 0; --y) fwrite(rgb + (size_t)3 * w * y, sizeof(float), (size_t)3 * w, f);
    fclose(f);
    return true;
}

bool ReadPFM(const std::string &path, std::vector<float> *rgb, int *w, int *h) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) return false;
    char magic[8];
    float scale;
    if (fscanf(f, "%7s %d %d %f", magic, w, h, &scale) != 4 || fgetc(f) == EOF) { fclose(f); return false; }
    int nc = strcmp(magic, "PF") == 0 ? 3 : (strcmp(magic, "Pf") == 0 ? 1 : 0);
    if (!nc) { fclose(f); return false; }
    std::vector<float> row((size_t)nc * *w);
    rgb->assign((size_t)3 * *w * *h, 0.f);
    for (int y = *h - 1; y >= 0; --y) {
        if (fread(row.data(), 4, row.size(), f) != row.size()) { fclose(f); return false; }
        for (int x = 0; x < *w; ++x)
            for (int c = 0; c < 3; ++c) {
                float v = row[(size_t)nc * x + (nc == 3 ? c : 0)];
                if (scale > 0) { uint32_t u; memcpy(&u, &v, 4); u = __builtin_bswap32(u); memcpy(&v, &u, 4); }
                (*rgb)[((size_t)y * *w + x) * 3 + c] = v * (scale < 0 ? -scale : scale);
            }
    }
    fclose(f);
    return true;
}

static bool WriteEXR(const std::string &path, const float *rgb, int w, int h) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    auto put32 = [&](uint32_t v) { fwrite(&v, 4, 1, f); };
    auto putStr = [&](const char *s) { fwrite(s, 1, strlen(s) + 1, f); };
    put32(20000630); put32(2);
    putStr("channels"); putStr("chlist"); put32(3 * 18 + 1);
    for (const char *c : {"B", "G", "R"}) { putStr(c); put32(2); put32(0); put32(1); put32(1); }
    fputc(0, f);
    putStr("compression"); putStr("compression"); put32(1); fputc(0, f);
    putStr("dataWindow"); putStr("box2i"); put32(16); put32(0); put32(0); put32(w - 1); put32(h - 1);
    putStr("displayWindow"); putStr("box2i"); put32(16); put32(0); put32(0); put32(w - 1); put32(h - 1);
    putStr("lineOrder"); putStr("lineOrder"); put32(1); fputc(0, f);
    putStr("pixelAspectRatio"); putStr("float"); put32(4); { float one = 1; fwrite(&one, 4, 1, f); }
    putStr("screenWindowCenter"); putStr("v2f"); put32(8); { float z[2] = {0, 0}; fwrite(z, 4, 2, f); }
    putStr("screenWindowWidth"); putStr("float"); put32(4); { float one = 1; fwrite(&one, 4, 1, f); }
    fputc(0, f);
    uint64_t tableStart = (uint64_t)ftell(f);
    uint64_t lineBytes = 8 + (uint64_t)12 * w;
    for (int y = 0; y < h; ++y) { uint64_t off = tableStart + 8 * (uint64_t)h + lineBytes * y; fwrite(&off, 8, 1, f); }
    std::vector<float> chan(w);
    for (int y = 0; y < h; ++y) {
        put32((uint32_t)y); put32((uint32_t)(12 * w));
        for (int c : {2, 1, 0}) {
            for (int x = 0; x < w; ++x) chan[x] = rgb[((size_t)y * w + x) * 3 + c];
            fwrite(chan.data(), 4, w, f);
        }
    }
    fclose(f);
    return true;
}

// float -> half -> float with round-to-nearest-even (util/float.h Half(float) ctor), what storing into a
// PixelFormat::Half image and reading it back does (film.cpp:536, util/image.h)
float RoundToHalf(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = x & 0x80000000u, mag = x & 0x7fffffffu;
    if (mag >= 0x7f800000u) return f;  // inf / nan
    float a;
    memcpy(&a, &mag, 4);
    if (a >= 65520.f) { uint32_t inf = sign | 0x7f800000u; float r; memcpy(&r, &inf, 4); return r; }
    float r;
    if (a < 6.103515625e-05f) {  // half subnormal range: quantum 2^-24
        float q = a * 16777216.f;             // exact
        float rq = __builtin_nearbyintf(q);   // RN-even in the default rounding mode
        r = rq / 16777216.f;
    } else {
        uint32_t m = mag;
        uint32_t rem = m & 0x1fffu, base = m & ~0x1fffu;
        if (rem > 0x1000u || (rem == 0x1000u && (base & 0x2000u))) base += 0x2000u;
        memcpy(&r, &base, 4);
    }
    uint32_t rb;
    memcpy(&rb, &r, 4);
    rb |= sign;
    memcpy(&r, &rb, 4);
    return r;
}

// the reference logs every NaN channel (LOG_ERROR, util/image.h:428); a handful are enough here
static void ReportNaN(size_t pixel, int w, int c) {
    static std::atomic<int> reported{0};
    if (reported.fetch_add(1) < 8) fprintf(stderr, "Error: NaN at pixel %d,%d comp %d\n", (int)(pixel % (size_t)w), (int)(pixel / (size_t)w), c);
}

// RGBFilm::GetPixelRGB (film.h:258-275, no splats) + RGBFilm::GetImage (film.cpp:533-565)
void FilmToRGB(const wf_film &F, const double *film, int w, int h, float *rgb, bool saveFP16) {
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        const double *px = film + 4 * i;
        float c[3] = {(float)px[0], (float)px[1], (float)px[2]};
        float weightSum = (float)px[3];
        if (weightSum != 0) { c[0] /= weightSum; c[1] /= weightSum; c[2] /= weightSum; }
        float o[3];
        for (int r = 0; r < 3; ++r) {
            o[r] = 0;
            for (int k = 0; k < 3; ++k) o[r] += F.outputRGBFromSensorRGB[r][k] * c[k];
        }
        if (saveFP16) {
            for (int r = 0; r < 3; ++r) { if (o[r] > 65504.f) o[r] = 65504.f; }
        }
        // Image::SetChannel (util/image.h:425-432): a NaN is reported and stored as 0
        for (int r = 0; r < 3; ++r) {
            if (o[r] != o[r]) { ReportNaN(i, w, r); o[r] = 0; }
            if (saveFP16) o[r] = RoundToHalf(o[r]);
        }
        rgb[3 * i] = o[0]; rgb[3 * i + 1] = o[1]; rgb[3 * i + 2] = o[2];
    }
}

// float -> half bits, round to nearest even (the values written below have already been through RoundToHalf: exact)
uint16_t FloatToHalfBits(float f) {
    const float r = RoundToHalf(f);
    uint32_t x;
    memcpy(&x, &r, 4);
    const uint32_t sign = (x >> 16) & 0x8000u, mag = x & 0x7fffffffu;
    if (mag >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((mag & 0x7fffffu) ? 0x200u : 0));
    if (mag == 0) return (uint16_t)sign;
    const int e = (int)(mag >> 23) - 127;
    if (e < -14) {   // half subnormal: value = m * 2^-24
        float a;
        memcpy(&a, &mag, 4);
        return (uint16_t)(sign | (uint32_t)(a * 16777216.f));
    }
    return (uint16_t)(sign | (uint32_t)((e + 15) << 10) | ((mag >> 13) & 0x3ffu));
}
// Image::WriteEXR for an image with named channels (util/image.cpp:1173-1253): single part, scan lines, no compression; the file
// stores channels alphabetically.  data[(y * w + x) * nc + c]; half = PixelFormat::Half channels (values already representable).
bool WriteEXRChannels(const std::string &path, const std::vector<std::string> &names, const float *data, int w, int h, bool half) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) return false;
    const int nc = (int)names.size();
    std::vector<int> order(nc);
    for (int i = 0; i < nc; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return names[a] < names[b]; });
    auto put32 = [&](uint32_t v) { fwrite(&v, 4, 1, f); };
    auto putStr = [&](const char *s) { fwrite(s, 1, strlen(s) + 1, f); };
    put32(20000630); put32(2);
    uint32_t chSize = 1;
    for (const std::string &n : names) chSize += (uint32_t)n.size() + 1 + 16;
    putStr("channels"); putStr("chlist"); put32(chSize);
    for (int c : order) { putStr(names[c].c_str()); put32(half ? 1 : 2); put32(0); put32(1); put32(1); }
    fputc(0, f);
    putStr("compression"); putStr("compression"); put32(1); fputc(0, f);
    putStr("dataWindow"); putStr("box2i"); put32(16); put32(0); put32(0); put32(w - 1); put32(h - 1);
    putStr("displayWindow"); putStr("box2i"); put32(16); put32(0); put32(0); put32(w - 1); put32(h - 1);
    putStr("lineOrder"); putStr("lineOrder"); put32(1); fputc(0, f);
    putStr("pixelAspectRatio"); putStr("float"); put32(4); { float one = 1; fwrite(&one, 4, 1, f); }
    putStr("screenWindowCenter"); putStr("v2f"); put32(8); { float z[2] = {0, 0}; fwrite(z, 4, 2, f); }
    putStr("screenWindowWidth"); putStr("float"); put32(4); { float one = 1; fwrite(&one, 4, 1, f); }
    fputc(0, f);
    const uint64_t tableStart = (uint64_t)ftell(f), bytesPer = half ? 2 : 4, lineBytes = 8 + bytesPer * nc * (uint64_t)w;
    for (int y = 0; y < h; ++y) { uint64_t off = tableStart + 8 * (uint64_t)h + lineBytes * y; fwrite(&off, 8, 1, f); }
    std::vector<float> chan(w);
    std::vector<uint16_t> chanH(w);
    for (int y = 0; y < h; ++y) {
        put32((uint32_t)y); put32((uint32_t)(bytesPer * nc * w));
        for (int c : order) {
            for (int x = 0; x < w; ++x) chan[x] = data[((size_t)y * w + x) * nc + c];
            if (half) { for (int x = 0; x < w; ++x) chanH[x] = FloatToHalfBits(chan[x]); fwrite(chanH.data(), 2, w, f); }
            else fwrite(chan.data(), 4, w, f);
        }
    }
    fclose(f);
    return true;
}

// SpectralFilm::GetImage (film.cpp:961-1027): R G B (GetPixelRGB) followed by one channel "S0.<bucket centre>nm" per bucket (the '.' of
// the number written as ','), c = bucketSums / weightSums where the weight is positive.  film = the RGB accumulators [pixels][4],
// spectral = [pixels][2 * n_buckets]; out = [pixels][3 + n_buckets].
void SpectralFilmImage(const wf_film &F, const double *film, const double *spectral, int w, int h, bool saveFP16, std::vector<std::string> *names, std::vector<float> *out) {
    const int nb = F.n_buckets, nc = 3 + nb;
    names->assign({"R", "G", "B"});
    for (int i = 0; i < nb; ++i) {
        const float t = (i + 0.5f) / nb;
        char buf[64];
        snprintf(buf, sizeof(buf), "%.3fnm", (double)((1 - t) * F.lambda_min + t * F.lambda_max));
        std::string lam = buf;
        std::replace(lam.begin(), lam.end(), '.', ',');
        names->push_back("S0." + lam);
    }
    out->assign((size_t)w * h * nc, 0.f);
    std::vector<float> rgb((size_t)w * h * 3);
    FilmToRGB(F, film, w, h, rgb.data(), false);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        float *o = &(*out)[i * nc];
        for (int c = 0; c < 3; ++c) {
            float v = rgb[3 * i + c];
            if (saveFP16) { if (v > 65504.f) v = 65504.f; v = RoundToHalf(v); }
            o[c] = v;
        }
        const double *sp = spectral + (size_t)2 * nb * i;
        for (int b = 0; b < nb; ++b) {
            float c = 0;
            if (sp[nb + b] > 0) {
                c = (float)(sp[b] / sp[nb + b]);   // (+ splatScale * bucketSplats / filterIntegral: no splats on this path)
                if (c != c) { ReportNaN(i, w, 3 + b); c = 0; }   // Image::SetChannel
                if (saveFP16) { if (c > 65504.f) c = 65504.f; c = RoundToHalf(c); }
            }
            o[3 + b] = c;
        }
    }
}

// GBufferFilm::GetImage (film.cpp:688-803): 25 channels from the RGB accumulators [pixels][4] and the per-pixel records
void GBufferFilmImage(const wf_film &F, const double *film, const wf_gbuffer_pixel *gb, int w, int h, bool saveFP16, std::vector<std::string> *names, std::vector<float> *out) {
    names->assign({"R", "G", "B", "Albedo.R", "Albedo.G", "Albedo.B", "P.X", "P.Y", "P.Z", "dzdx", "dzdy", "N.X", "N.Y", "N.Z", "Ns.X", "Ns.Y", "Ns.Z", "u", "v",
                   "Variance.R", "Variance.G", "Variance.B", "RelativeVariance.R", "RelativeVariance.G", "RelativeVariance.B"});
    const int nc = 25;
    out->assign((size_t)w * h * nc, 0.f);
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        const double *px = film + 4 * i;
        const wf_gbuffer_pixel &g = gb[i];
        float rgb[3] = {(float)px[0], (float)px[1], (float)px[2]}, alb[3] = {(float)g.rgb_albedo_sum[0], (float)g.rgb_albedo_sum[1], (float)g.rgb_albedo_sum[2]};
        const float weightSum = (float)px[3], gws = (float)g.gbuffer_weight_sum;
        float pt[3] = {g.p_sum[0], g.p_sum[1], g.p_sum[2]}, uv[2] = {g.uv_sum[0], g.uv_sum[1]}, dzdx = g.dzdx_sum, dzdy = g.dzdy_sum;
        if (weightSum != 0) for (int c = 0; c < 3; ++c) { rgb[c] /= weightSum; alb[c] /= weightSum; }
        if (gws != 0) {
            for (int c = 0; c < 3; ++c) pt[c] /= gws;
            uv[0] /= gws; uv[1] /= gws;
            dzdx /= gws; dzdy /= gws;
        }
        float o[3];
        for (int r = 0; r < 3; ++r) {
            o[r] = 0;
            for (int k = 0; k < 3; ++k) o[r] += F.outputRGBFromSensorRGB[r][k] * rgb[k];
        }
        if (saveFP16) for (int c = 0; c < 3; ++c) if (o[c] > 65504.f) o[c] = 65504.f;
        auto normalized = [](const float v[3], float n[3]) {
            const float l2 = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
            if (!(l2 > 0)) { n[0] = n[1] = n[2] = 0; return; }
            const float l = std::sqrt(l2);
            n[0] = v[0] / l; n[1] = v[1] / l; n[2] = v[2] / l;
        };
        float n[3], ns[3];
        normalized(g.n_sum, n);
        normalized(g.ns_sum, ns);
        float var[3], rel[3];
        for (int c = 0; c < 3; ++c) {
            var[c] = g.var_n[c] > 1 ? g.var_s[c] / (g.var_n[c] - 1) : 0.f;
            rel[c] = (g.var_n[c] < 1 || g.var_mean[c] == 0) ? 0.f : var[c] / g.var_mean[c];
        }
        const float ch[nc] = {o[0], o[1], o[2], alb[0], alb[1], alb[2], pt[0], pt[1], pt[2], std::fabs(dzdx), std::fabs(dzdy), n[0], n[1], n[2], ns[0], ns[1], ns[2],
                              uv[0], uv[1], var[0], var[1], var[2], rel[0], rel[1], rel[2]};
        float *dst = &(*out)[i * nc];
        for (int c = 0; c < nc; ++c) {
            float v = ch[c];
            if (v != v) { ReportNaN(i, w, c); v = 0; }   // Image::SetChannel
            dst[c] = saveFP16 ? RoundToHalf(v) : v;
        }
    }
}

bool WriteFilmImage(const SceneTables &T, const std::string &path, std::vector<float> &rgb, int w, int h) {
    const size_t dot = path.find_last_of('.');
    const bool exr = dot != std::string::npos && path.substr(dot) == ".exr";
    if (!exr && T.sRGBFromFilmRGB.size() == 9) {
        fprintf(stderr, "Warning: %s: converting pixel colors to sRGB to match output image format.\n", path.c_str());
        const float *m = T.sRGBFromFilmRGB.data();
        for (size_t i = 0; i < (size_t)w * h; ++i) {   // Mul<RGB>(m, channels): sums accumulated from 0, left to right (util/math.h:1404-1413)
            const float v[3] = {rgb[3 * i], rgb[3 * i + 1], rgb[3 * i + 2]};
            for (int r = 0; r < 3; ++r) {
                float acc = 0;
                for (int c = 0; c < 3; ++c) acc += m[3 * r + c] * v[c];
                rgb[3 * i + r] = acc;
            }
        }
    }
    return WriteImage(path, rgb.data(), w, h);
}

bool WriteImage(const std::string &path, const float *rgb, int w, int h) {
    size_t dot = path.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : path.substr(dot);
    if (ext == ".exr") return WriteEXR(path, rgb, w, h);
    return WritePFM(path, rgb, w, h);
}

}  // namespace wf
