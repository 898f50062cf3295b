// Build shim: see nanovdb/NanoVDB.h
#pragma once
#include <nanovdb/NanoVDB.h>
#include <utility>
namespace nanovdb {
template <typename BufferT> class GridHandle {
  public:
    GridHandle() {}
    GridHandle(GridHandle &&) = default;
    GridHandle &operator=(GridHandle &&) = default;
    explicit operator bool() const { return false; }
    const GridMetaData *gridMetaData() const { return &md_; }
    template <typename T> const Grid<T> *grid() const { return nullptr; }
  private:
    GridMetaData md_;
};
}  // namespace nanovdb
