// Build shim: see nanovdb/NanoVDB.h
#pragma once
#include <nanovdb/util/GridHandle.h>
namespace nanovdb { namespace io {
template <typename BufferT>
GridHandle<BufferT> readGrid(const std::string &, const std::string &, int, const BufferT &) {
    throw std::runtime_error("NanoVDB unavailable in oracle build");
}
}}  // namespace nanovdb::io
