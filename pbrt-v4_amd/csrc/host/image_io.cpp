// image_io.cpp — image output for the film (util/image.cpp:WritePFM/ReadPFM semantics: bottom-to-top
// scanlines, little-endian scale -1).  EXR is written uncompressed, 32-bit float, channels B,G,R.
#include "scene.h"

#include <cstdio>
#include <cstring>

namespace wf {

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
static float RoundToHalf(float f) {
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
            for (int r = 0; r < 3; ++r) { if (o[r] > 65504.f) o[r] = 65504.f; o[r] = RoundToHalf(o[r]); }
        }
        rgb[3 * i] = o[0]; rgb[3 * i + 1] = o[1]; rgb[3 * i + 2] = o[2];
    }
}

bool WriteImage(const std::string &path, const float *rgb, int w, int h) {
    size_t dot = path.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : path.substr(dot);
    if (ext == ".exr") return WriteEXR(path, rgb, w, h);
    return WritePFM(path, rgb, w, h);
}

}  // namespace wf
