// image_formats.cpp — the image-input formats Image::Read reaches through third-party decoders that are not in the reference
// checkout (util/image.cpp:876-922, 1697-1757): .qoi (ext/qoi), .hdr (Radiance RGBE through stb_image's stbi_loadf) and .tga (stb_image's
// stbi_load fallback for any other extension).  Own decoders written from the published format descriptions; what the decoded
// samples BECOME follows the reference: QOI -> 8-bit R G B (A), encoding sRGB or linear from the file's colour-space byte;
// HDR -> float R G B with stb's mantissa * 2^(e - 136) conversion; stb's 8-bit loads -> sRGB-encoded Y | R G B (grey+alpha and RGBA
// lose their alpha: SelectChannels).  Every size and offset of the input is checked: malformed files raise SceneError.
#include "scene.h"

#include <cmath>
#include <cstring>

namespace wf {

static void Die(const std::string &where, const std::string &msg) { throw SceneError(where.empty() ? msg : where + ": " + msg); }

static std::vector<uint8_t> ReadWholeFile(const std::string &path) {
    std::vector<uint8_t> file;
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) Die("", path + ": unable to open file");
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    file.resize(n > 0 ? n : 0);
    if (n > 0 && fread(file.data(), 1, n, f) != (size_t)n) { fclose(f); Die("", path + ": read error"); }
    fclose(f);
    return file;
}
static constexpr size_t kMaxPixels = (size_t)1 << 30;

// ---------------------------------------------------------------------------------------------------------------
// QOI ("Quite OK Image", specification version 1.0): 14-byte header "qoif" width height channels colorspace (big endian),
// then chunks QOI_OP_RGB / RGBA / INDEX / DIFF / LUMA / RUN over a running pixel and a 64-entry colour index, 8-byte end marker.
void ReadQOI(const std::string &path, HostImage *img) {
    const std::vector<uint8_t> file = ReadWholeFile(path);
    if (file.size() < 14 + 8 || memcmp(file.data(), "qoif", 4)) Die("", path + ": not a QOI file");
    auto be32 = [&](size_t o) { return (uint32_t)file[o] << 24 | (uint32_t)file[o + 1] << 16 | (uint32_t)file[o + 2] << 8 | file[o + 3]; };
    const uint32_t w = be32(4), h = be32(8);
    const int channels = file[12], colorspace = file[13];
    if (w == 0 || h == 0 || (channels != 3 && channels != 4) || colorspace > 1 || (size_t)w * h > kMaxPixels) Die("", path + ": malformed QOI header");
    img->format = HostImage::U256;
    img->w = (int)w; img->h = (int)h; img->nc = channels;
    img->enc = colorspace == 0 ? ColorEnc() : ColorEnc::Linear();   // QOI_SRGB = 0, QOI_LINEAR = 1
    img->p8.resize((size_t)w * h * channels);
    uint8_t index[64][4];
    memset(index, 0, sizeof(index));
    uint8_t px[4] = {0, 0, 0, 255};
    size_t p = 14;
    const size_t end = file.size() - 8;
    int run = 0;
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        if (run > 0) --run;
        else if (p < end) {
            const int b1 = file[p++];
            auto need = [&](size_t k) { if (p + k > end) Die("", path + ": truncated QOI data"); };
            if (b1 == 0xfe) { need(3); px[0] = file[p]; px[1] = file[p + 1]; px[2] = file[p + 2]; p += 3; }
            else if (b1 == 0xff) { need(4); px[0] = file[p]; px[1] = file[p + 1]; px[2] = file[p + 2]; px[3] = file[p + 3]; p += 4; }
            else if ((b1 & 0xc0) == 0x00) memcpy(px, index[b1], 4);
            else if ((b1 & 0xc0) == 0x40) {
                px[0] = (uint8_t)(px[0] + ((b1 >> 4) & 3) - 2);
                px[1] = (uint8_t)(px[1] + ((b1 >> 2) & 3) - 2);
                px[2] = (uint8_t)(px[2] + (b1 & 3) - 2);
            } else if ((b1 & 0xc0) == 0x80) {
                need(1);
                const int b2 = file[p++];
                const int vg = (b1 & 0x3f) - 32;
                px[0] = (uint8_t)(px[0] + vg - 8 + ((b2 >> 4) & 0x0f));
                px[1] = (uint8_t)(px[1] + vg);
                px[2] = (uint8_t)(px[2] + vg - 8 + (b2 & 0x0f));
            } else run = b1 & 0x3f;
            memcpy(index[(px[0] * 3 + px[1] * 5 + px[2] * 7 + px[3] * 11) % 64], px, 4);
        }
        memcpy(&img->p8[i * channels], px, channels);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Radiance RGBE (.hdr), as stb_image reads it: header lines up to an empty one (FORMAT=32-bit_rle_rgbe required), "-Y h +X w",
// then flat RGBE pixels or new-style run-length encoded scanlines (2 2 hi lo, four component planes).  stbi__hdr_convert:
// exponent byte 0 -> black, otherwise component * (float)ldexp(1.0f, e - (128 + 8)).
void ReadHDR(const std::string &path, HostImage *img) {
    const std::vector<uint8_t> file = ReadWholeFile(path);
    size_t p = 0;
    auto line = [&]() {
        std::string s;
        while (p < file.size() && file[p] != '\n') { if (s.size() < 1023) s.push_back((char)file[p]); ++p; }
        if (p < file.size()) ++p;
        return s;
    };
    const std::string magic = line();
    if (magic != "#?RADIANCE" && magic != "#?RGBE") Die("", path + ": not a Radiance HDR file");
    bool valid = false;
    while (true) {
        if (p >= file.size()) Die("", path + ": truncated HDR header");
        const std::string l = line();
        if (l.empty()) break;
        if (l == "FORMAT=32-bit_rle_rgbe") valid = true;
    }
    if (!valid) Die("", path + ": unsupported HDR format (32-bit_rle_rgbe only)");
    const std::string res = line();
    int w = 0, h = 0;
    if (sscanf(res.c_str(), "-Y %d +X %d", &h, &w) != 2 || w <= 0 || h <= 0 || (size_t)w * h > kMaxPixels) Die("", path + ": unsupported HDR data layout (-Y h +X w only)");
    img->format = HostImage::Float;
    img->w = w; img->h = h; img->nc = 3;
    img->p32.resize((size_t)w * h * 3);
    auto convert = [&](const uint8_t rgbe[4], float *out) {
        if (rgbe[3] != 0) {
            const float f1 = (float)std::ldexp(1.0f, (int)rgbe[3] - (128 + 8));
            out[0] = rgbe[0] * f1; out[1] = rgbe[1] * f1; out[2] = rgbe[2] * f1;
        } else out[0] = out[1] = out[2] = 0;
    };
    auto flatFrom = [&](size_t firstPixel) {
        for (size_t i = firstPixel; i < (size_t)w * h; ++i) {
            if (p + 4 > file.size()) Die("", path + ": truncated HDR data");
            convert(&file[p], &img->p32[3 * i]);
            p += 4;
        }
    };
    if (w < 8 || w >= 32768) { flatFrom(0); return; }
    std::vector<uint8_t> scan((size_t)w * 4);
    for (int j = 0; j < h; ++j) {
        if (p + 4 > file.size()) Die("", path + ": truncated HDR data");
        const int c1 = file[p], c2 = file[p + 1], len = file[p + 2];
        if (c1 != 2 || c2 != 2 || (len & 0x80)) {
            // not run-length encoded: stb reads the whole image flat from here (it only gets here on the first scanline)
            if (j != 0) Die("", path + ": corrupt HDR scanline");
            flatFrom(0);
            return;
        }
        if ((len << 8 | file[p + 3]) != w) Die("", path + ": invalid decoded scanline length (corrupt HDR)");
        p += 4;
        for (int k = 0; k < 4; ++k) {
            int i = 0;
            while (i < w) {
                if (p >= file.size()) Die("", path + ": truncated HDR data");
                int count = file[p++];
                if (count > 128) {
                    count -= 128;
                    if (count == 0 || count > w - i || p >= file.size()) Die("", path + ": corrupt HDR run");
                    const uint8_t v = file[p++];
                    for (int z = 0; z < count; ++z) scan[(size_t)(i++) * 4 + k] = v;
                } else {
                    if (count == 0 || count > w - i || p + count > file.size()) Die("", path + ": corrupt HDR run");
                    for (int z = 0; z < count; ++z) scan[(size_t)(i++) * 4 + k] = file[p++];
                }
            }
        }
        for (int i = 0; i < w; ++i) convert(&scan[(size_t)i * 4], &img->p32[3 * ((size_t)j * w + i)]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Truevision TGA as stb_image loads it (image types 1 / 2 / 3 and their run-length encoded forms 9 / 10 / 11; 8, 15 / 16, 24, 32 bits per
// pixel; colour-mapped files), rows flipped to top-to-bottom unless descriptor bit 5 says they already are, BGR(A) -> RGB(A), 15 / 16-bit
// pixels expanded per channel as (c * 255) / 31.  The reference keeps 1 channel as Y, drops the alpha of 2- and 4-channel images,
// and treats the bytes as sRGB-encoded (util/image.cpp:888-916).
void ReadTGA(const std::string &path, HostImage *img) {
    const std::vector<uint8_t> file = ReadWholeFile(path);
    if (file.size() < 18) Die("", path + ": truncated TGA header");
    const int idLen = file[0], indexed = file[1];
    int imageType = file[2];
    const int palStart = file[3] | file[4] << 8, palLen = file[5] | file[6] << 8, palBits = file[7];
    const int w = file[12] | file[13] << 8, h = file[14] | file[15] << 8, bpp = file[16];
    const bool topDown = (file[17] >> 5) & 1;
    const bool rle = imageType >= 8;
    if (rle) imageType -= 8;
    auto comps = [&](int bits, bool grey) { return bits == 8 ? 1 : (bits == 15 ? 3 : bits == 16 ? (grey ? 2 : 3) : bits == 24 ? 3 : bits == 32 ? 4 : 0); };
    const bool rgb16 = !indexed ? (bpp == 15 || (bpp == 16 && imageType != 3)) : (palBits == 15 || palBits == 16);
    const int nc = indexed ? comps(palBits, false) : comps(bpp, imageType == 3);
    if (w <= 0 || h <= 0 || (size_t)w * h > kMaxPixels || nc == 0 || (imageType != 1 && imageType != 2 && imageType != 3) || (indexed != 0) != (imageType == 1) ||
        (indexed && bpp != 8 && bpp != 16))
        Die("", path + ": unsupported or malformed TGA header");
    size_t p = 18 + (size_t)idLen;
    std::vector<uint8_t> palette;
    auto readPixel = [&](const uint8_t *src, uint8_t *dst) {   // one stored pixel (file order) -> nc bytes, still BGR
        if (rgb16) {
            const unsigned px = src[0] | src[1] << 8;
            const unsigned r = (px >> 10) & 31, g = (px >> 5) & 31, b = px & 31;
            dst[0] = (uint8_t)((r * 255) / 31); dst[1] = (uint8_t)((g * 255) / 31); dst[2] = (uint8_t)((b * 255) / 31);   // already R G B
        } else memcpy(dst, src, nc);
    };
    const int palBytes = indexed ? (palBits + 7) / 8 : 0;
    if (indexed) {
        if (palLen <= 0) Die("", path + ": TGA colour map missing");
        p += (size_t)palStart;   // (stb_image skips palette_start BYTES here: reproduced)
        if (p + (size_t)palLen * palBytes > file.size()) Die("", path + ": truncated TGA colour map");
        palette.resize((size_t)palLen * nc);
        for (int i = 0; i < palLen; ++i) readPixel(&file[p + (size_t)i * palBytes], &palette[(size_t)i * nc]);
        p += (size_t)palLen * palBytes;
    }
    const int srcBytes = (bpp + 7) / 8;
    std::vector<uint8_t> data((size_t)w * h * nc);
    uint8_t raw[4] = {0, 0, 0, 0};
    int rleCount = 0;
    bool rleRepeating = false, readNext = true;
    for (size_t i = 0; i < (size_t)w * h; ++i) {
        if (rle) {
            if (rleCount == 0) {
                if (p >= file.size()) Die("", path + ": truncated TGA data");
                const int cmd = file[p++];
                rleCount = 1 + (cmd & 127);
                rleRepeating = cmd >> 7;
                readNext = true;
            } else if (!rleRepeating) readNext = true;
        } else readNext = true;
        if (readNext) {
            if (p + srcBytes > file.size()) Die("", path + ": truncated TGA data");
            if (indexed) {
                int idx = srcBytes == 1 ? file[p] : (file[p] | file[p + 1] << 8);
                if (idx >= palLen) idx = 0;
                memcpy(raw, &palette[(size_t)idx * nc], nc);
            } else readPixel(&file[p], raw);
            p += srcBytes;
            readNext = false;
        }
        memcpy(&data[i * nc], raw, nc);
        --rleCount;
    }
    // bottom-up files are flipped; BGR -> RGB for 24 / 32-bit pixels
    std::vector<uint8_t> out((size_t)w * h * nc);
    for (int y = 0; y < h; ++y) {
        const int sy = topDown ? y : h - 1 - y;
        memcpy(&out[(size_t)y * w * nc], &data[(size_t)sy * w * nc], (size_t)w * nc);
    }
    if (nc >= 3 && !rgb16)
        for (size_t i = 0; i < (size_t)w * h; ++i) std::swap(out[i * nc], out[i * nc + 2]);
    img->format = HostImage::U256;
    img->w = w; img->h = h; img->nc = nc;
    img->enc = ColorEnc();   // sRGB
    img->p8.swap(out);
    if (nc == 2) img->SelectChannels(0, 1);        // Y A -> Y
    else if (nc == 4) img->SelectChannels(0, 3);   // R G B A -> R G B
}

}  // namespace wf
