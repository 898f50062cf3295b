// Build shim: type-level stand-in for the OpenEXR 3 API surface referenced by the reference's
// util/image.cpp:1023-1253. Every file operation throws, so ReadEXR/WriteEXR report an error;
// the oracle build reads and writes PFM only.
#pragma once
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
    OutputFile(const char *, const Header &) { throw std::runtime_error("OpenEXR unavailable in oracle build (use .pfm)"); }
    void setFrameBuffer(const FrameBuffer &) {}
    void writePixels(int) {}
};
}  // namespace Imf
