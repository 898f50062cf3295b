// Build shim: stb_image declarations only; every loader fails (PFM is the oracle build's image format).
#pragma once
typedef unsigned char stbi_uc;
static inline stbi_uc *stbi_load(const char *, int *, int *, int *, int) { return nullptr; }
static inline float *stbi_loadf(const char *, int *, int *, int *, int) { return nullptr; }
static inline void stbi_image_free(void *) {}
static inline const char *stbi_failure_reason() { return "stb_image unavailable in oracle build"; }
