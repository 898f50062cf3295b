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

bool WriteImage(const std::string &path, const float *rgb, int w, int h) {
    size_t dot = path.find_last_of('.');
    std::string ext = dot == std::string::npos ? "" : path.substr(dot);
    if (ext == ".exr") return WriteEXR(path, rgb, w, h);
    return WritePFM(path, rgb, w, h);
}

}  // namespace wf
