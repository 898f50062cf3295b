// Build shim: see nanovdb/NanoVDB.h
#pragma once
#include <nanovdb/NanoVDB.h>
namespace nanovdb {
template <typename TreeT, int Order, bool UseCache> struct SampleFromVoxels {
    explicit SampleFromVoxels(const TreeT &) {}
    template <typename V> float operator()(const V &) const { return 0.f; }
};
}  // namespace nanovdb
