// Build shim: gzip input is unsupported in the oracle build (reference use: util/file.cpp:164-205).
#pragma once
#include <cstddef>
struct libdeflate_decompressor { int unused; };
enum libdeflate_result { LIBDEFLATE_SUCCESS = 0, LIBDEFLATE_BAD_DATA = 1,
                         LIBDEFLATE_SHORT_OUTPUT = 2, LIBDEFLATE_INSUFFICIENT_SPACE = 3 };
static inline libdeflate_decompressor *libdeflate_alloc_decompressor() {
    static libdeflate_decompressor d; return &d; }
static inline libdeflate_result libdeflate_gzip_decompress(libdeflate_decompressor *, const void *,
        size_t, void *, size_t, size_t *) { return LIBDEFLATE_BAD_DATA; }
