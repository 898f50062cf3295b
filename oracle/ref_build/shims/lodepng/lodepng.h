// Build shim: lodepng declarations only; PNG I/O reports an error in the oracle build.
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
static inline unsigned lodepng_inspect(unsigned *, unsigned *, LodePNGState *, const unsigned char *,
                                       size_t) { return 1; }
static inline const char *lodepng_error_text(unsigned) { return "PNG unavailable in oracle build"; }
static inline unsigned lodepng_encode_memory(unsigned char **, size_t *, const unsigned char *,
        unsigned, unsigned, LodePNGColorType, unsigned) { return 1; }
namespace lodepng {
static inline unsigned decode(std::vector<unsigned char> &, unsigned &, unsigned &,
        const unsigned char *, size_t, LodePNGColorType = LCT_RGBA, unsigned = 8) { return 1; }
static inline unsigned decode(std::vector<unsigned char> &, unsigned &, unsigned &, LodePNGState &,
        const unsigned char *, size_t) { return 1; }
}
