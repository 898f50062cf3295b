// Build shim: QOI declarations only; QOI I/O fails in the oracle build.
#pragma once
#define QOI_SRGB 0
#define QOI_LINEAR 1
typedef struct { unsigned int width, height; unsigned char channels, colorspace; } qoi_desc;
static inline void *qoi_encode(const void *, const qoi_desc *, int *) { return nullptr; }
static inline void *qoi_decode(const void *, int, qoi_desc *, int) { return nullptr; }
