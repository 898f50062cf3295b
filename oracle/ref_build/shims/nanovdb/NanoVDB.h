// Build shim: type-level stand-in for NanoVDB (openvdb feature/nanovdb @414bed84), covering the
// calls in the reference's media.h:599-679 and media.cpp:488-660. No grid can be loaded
// (io::readGrid throws), so NanoVDBMedium is unusable in the oracle build; GridMedium is used instead.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
namespace nanovdb {
template <typename T> struct Vec3 {
    T v[3];
    Vec3() : v{0, 0, 0} {}
    Vec3(T a, T b, T c) : v{a, b, c} {}
    T &operator[](int i) { return v[i]; }
    const T &operator[](int i) const { return v[i]; }
};
using Vec3R = Vec3<double>;
using Vec3f = Vec3<float>;
struct Coord { int v[3]; Coord() : v{0, 0, 0} {} Coord(int a, int b, int c) : v{a, b, c} {}
               int &operator[](int i) { return v[i]; } const int &operator[](int i) const { return v[i]; } };
template <typename V> struct BBox { V mn, mx; const V &min() const { return mn; } const V &max() const { return mx; } };
using CoordBBox = BBox<Coord>;
template <typename T> struct Tree {
    void extrema(T &a, T &b) const { a = b = T(0); }
};
template <typename T> struct ReadAccessor { T getValue(const Coord &) const { return T(0); } };
template <typename T> struct Grid {
    using TreeType = Tree<T>;
    template <typename V> V worldToIndexF(const V &p) const { return p; }
    BBox<Vec3R> worldBBox() const { return {}; }
    CoordBBox indexBBox() const { return {}; }
    const TreeType &tree() const { return tree_; }
    ReadAccessor<T> getAccessor() const { return {}; }
    TreeType tree_;
};
using FloatGrid = Grid<float>;
struct GridMetaData {
    bool isFogVolume() const { return false; }
    bool isUnknown() const { return true; }
    uint64_t activeVoxelCount() const { return 0; }
};
}  // namespace nanovdb
