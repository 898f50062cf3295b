// Build shim: stand-in for the OpenEXR 3 API surface referenced by the reference's util/image.cpp:1023-1253.
// Reading throws (the oracle build reads PFM / PNG).  WRITING works since round 3: OutputFile writes a minimal but valid OpenEXR
// file — version 2, single part, scan lines, NO compression, the header's channels (HALF / FLOAT, alphabetical as the format
// requires), data and display windows — so that the reference's SpectralFilm and GBufferFilm, which only write .exr
// (film.cpp:1045-1047, 760-762), can produce golden images in this build.  Attributes other than the required ones are dropped.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>
namespace Imath {
struct V2i { int x, y; V2i(int x = 0, int y = 0) : x(x), y(y) {} };
struct V2f { float x, y; V2f(float x = 0, float y = 0) : x(x), y(y) {} };
struct Box2i { V2i min, max; Box2i() {} Box2i(V2i a, V2i b) : min(a), max(b) {} };
}  // namespace Imath
namespace Imf {
enum PixelType { UINT = 0, HALF = 1, FLOAT = 2 };
struct Slice {
    PixelType type; char *base; size_t xStride, yStride;
    Slice(PixelType t = HALF, char *b = nullptr, size_t xs = 0, size_t ys = 0)
        : type(t), base(b), xStride(xs), yStride(ys) {}
};
class FrameBuffer {
  public:
    struct Iterator {
        std::map<std::string, Slice>::iterator it;
        const char *name() const { return it->first.c_str(); }
        Slice &slice() const { return it->second; }
        Iterator &operator++() { ++it; return *this; }
        bool operator!=(const Iterator &o) const { return it != o.it; }
    };
    void insert(const std::string &n, const Slice &s) { m_[n] = s; }
    Iterator begin() { return {m_.begin()}; }
    Iterator end() { return {m_.end()}; }
  private:
    std::map<std::string, Slice> m_;
};
struct Attribute { virtual ~Attribute() {} virtual const char *typeName() const { return "shim"; } };
template <typename T>
struct TypedAttribute : Attribute {
    T v;
    TypedAttribute() {}
    TypedAttribute(const T &v) : v(v) {}
    const T &value() const { return v; }
    T &value() { return v; }
};
struct M44f { float m[4][4]; M44f() {} M44f(const float a[4][4]) { for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) m[i][j] = a[i][j]; }
              const float *getValue() const { return &m[0][0]; }
              float *operator[](int i) { return m[i]; } const float *operator[](int i) const { return m[i]; } };
struct Chromaticities { Imath::V2f red, green, blue, white; Chromaticities() {}
    Chromaticities(Imath::V2f r, Imath::V2f g, Imath::V2f b, Imath::V2f w) : red(r), green(g), blue(b), white(w) {} };
using FloatAttribute = TypedAttribute<float>;
using IntAttribute = TypedAttribute<int>;
using StringAttribute = TypedAttribute<std::string>;
using StringVectorAttribute = TypedAttribute<std::vector<std::string>>;
using M44fAttribute = TypedAttribute<M44f>;
using ChromaticitiesAttribute = TypedAttribute<Chromaticities>;
struct Channel { PixelType type; Channel(PixelType t = HALF) : type(t) {} };
class ChannelList {
  public:
    struct ConstIterator {
        std::map<std::string, Channel>::const_iterator it;
        const char *name() const { return it->first.c_str(); }
        const Channel &channel() const { return it->second; }
        ConstIterator &operator++() { ++it; return *this; }
        bool operator!=(const ConstIterator &o) const { return it != o.it; }
    };
    void insert(const std::string &n, const Channel &c) { m_[n] = c; }
    ConstIterator begin() const { return {m_.begin()}; }
    ConstIterator end() const { return {m_.end()}; }
  private:
    std::map<std::string, Channel> m_;
};
class Header {
  public:
    struct ConstIterator {
        const Attribute *a;
        const char *name() const { return ""; }
        const Attribute &attribute() const { return *a; }
        ConstIterator &operator++() { return *this; }
        bool operator!=(const ConstIterator &) const { return false; }
    };
    Header() {}
    Header(const Imath::Box2i &disp, const Imath::Box2i &data) : disp_(disp), data_(data) {}
    const Imath::Box2i &dataWindow() const { return data_; }
    const Imath::Box2i &displayWindow() const { return disp_; }
    ChannelList &channels() { return ch_; }
    const ChannelList &channels() const { return ch_; }
    template <typename T> const T *findTypedAttribute(const char *) const { return nullptr; }
    template <typename T> void insert(const std::string &, const T &) {}
    ConstIterator begin() const { return {nullptr}; }
    ConstIterator end() const { return {nullptr}; }
  private:
    Imath::Box2i disp_, data_;
    ChannelList ch_;
};
class InputFile {
  public:
    InputFile(const char *) { throw std::runtime_error("OpenEXR unavailable in oracle build (use .pfm)"); }
    const Header &header() const { return h_; }
    void setFrameBuffer(const FrameBuffer &) {}
    void readPixels(int, int) {}
  private:
    Header h_;
};
class OutputFile {
  public:
    OutputFile(const char *name, const Header &h) : name_(name), h_(h) {}
    void setFrameBuffer(const FrameBuffer &fb) { fb_ = fb; }
    void writePixels(int nLines) {
        FILE *f = fopen(name_.c_str(), "wb");
        if (!f) throw std::runtime_error("cannot open " + name_ + " for writing");
        auto put32 = [&](int32_t v) { fwrite(&v, 4, 1, f); };
        auto putStr = [&](const std::string &s) { fwrite(s.c_str(), 1, s.size() + 1, f); };
        auto attr = [&](const char *n, const char *type, int32_t size) { putStr(n); putStr(type); put32(size); };
        const Imath::Box2i dw = h_.dataWindow(), disp = h_.displayWindow();
        const int w = dw.max.x - dw.min.x + 1, hgt = dw.max.y - dw.min.y + 1;
        if (nLines != hgt) { fclose(f); throw std::runtime_error("shim OutputFile: partial writes are not supported"); }
        put32(20000630); put32(2);
        // channels (the std::map iterates alphabetically): name, pixel type, pLinear + 3 reserved, x / y sampling
        int32_t chSize = 1;
        for (auto it = h_.channels().begin(); it != h_.channels().end(); ++it) chSize += (int32_t)strlen(it.name()) + 1 + 16;
        attr("channels", "chlist", chSize);
        for (auto it = h_.channels().begin(); it != h_.channels().end(); ++it) {
            putStr(it.name());
            put32((int32_t)it.channel().type); put32(0); put32(1); put32(1);
        }
        fputc(0, f);
        attr("compression", "compression", 1); fputc(0, f);
        attr("dataWindow", "box2i", 16); put32(dw.min.x); put32(dw.min.y); put32(dw.max.x); put32(dw.max.y);
        attr("displayWindow", "box2i", 16); put32(disp.min.x); put32(disp.min.y); put32(disp.max.x); put32(disp.max.y);
        attr("lineOrder", "lineOrder", 1); fputc(0, f);
        attr("pixelAspectRatio", "float", 4); { float one = 1; fwrite(&one, 4, 1, f); }
        attr("screenWindowCenter", "v2f", 8); { float z[2] = {0, 0}; fwrite(z, 4, 2, f); }
        attr("screenWindowWidth", "float", 4); { float one = 1; fwrite(&one, 4, 1, f); }
        fputc(0, f);
        size_t lineBytes = 0;
        for (auto it = h_.channels().begin(); it != h_.channels().end(); ++it) lineBytes += (size_t)w * (it.channel().type == HALF ? 2 : 4);
        const long tablePos = ftell(f);
        uint64_t off = (uint64_t)tablePos + 8ull * hgt;
        for (int y = 0; y < hgt; ++y) { fwrite(&off, 8, 1, f); off += 8 + lineBytes; }
        for (int y = 0; y < hgt; ++y) {
            put32(dw.min.y + y); put32((int32_t)lineBytes);
            for (auto it = fb_.begin(); it != fb_.end(); ++it) {   // (same alphabetical order as the header's channel list)
                const Slice &s = it.slice();
                const size_t bytes = s.type == HALF ? 2 : 4;
                for (int x = 0; x < w; ++x) fwrite(s.base + (size_t)(dw.min.x + x) * s.xStride + (size_t)(dw.min.y + y) * s.yStride, bytes, 1, f);
            }
        }
        fclose(f);
    }
  private:
    std::string name_;
    Header h_;
    FrameBuffer fb_;
};
}  // namespace Imf
