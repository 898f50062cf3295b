// Build shim (test infrastructure): the part of the lodepng interface util/image.cpp calls, implemented in
// shim_png.cpp on top of the system zlib (the vendored lodepng submodule is absent from /root/reference).
// Decoding covers non-interlaced PNGs of every colour type at 8 and 16 bits (sub-byte grey / palette depths
// are expanded to 8 bits); encoding writes 8-bit grey / RGB with filter 0.
#pragma once
#include <cstddef>
#include <vector>
typedef enum LodePNGColorType { LCT_GREY = 0, LCT_RGB = 2, LCT_PALETTE = 3, LCT_GREY_ALPHA = 4,
                                LCT_RGBA = 6 } LodePNGColorType;
struct LodePNGColorMode { LodePNGColorType colortype; unsigned bitdepth; };
struct LodePNGInfo { LodePNGColorMode color; unsigned srgb_defined = 0; unsigned gama_defined = 0;
                     unsigned gama_gamma = 0; unsigned iccp_defined = 0; };
struct LodePNGState { LodePNGInfo info_png; LodePNGColorMode info_raw; };
static inline void lodepng_state_init(LodePNGState *) {}
static inline void lodepng_state_cleanup(LodePNGState *) {}
unsigned lodepng_inspect(unsigned *w, unsigned *h, LodePNGState *state, const unsigned char *in, size_t insize);
const char *lodepng_error_text(unsigned code);
unsigned lodepng_encode_memory(unsigned char **out, size_t *outsize, const unsigned char *image, unsigned w,
                               unsigned h, LodePNGColorType colortype, unsigned bitdepth);
namespace lodepng {
unsigned decode(std::vector<unsigned char> &out, unsigned &w, unsigned &h, const unsigned char *in,
                size_t insize, LodePNGColorType colortype = LCT_RGBA, unsigned bitdepth = 8);
unsigned decode(std::vector<unsigned char> &out, unsigned &w, unsigned &h, LodePNGState &state,
                const unsigned char *in, size_t insize);
}
