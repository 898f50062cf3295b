// Build shim: type-level stand-in for Ptex (reference use: textures.cpp:586-684,756-775).
// PtexCache::get always fails, so "ptex" textures report an error in the oracle build.
#pragma once
#include <cstddef>
#include <string>
struct PtexErrorHandler { virtual ~PtexErrorHandler() {} virtual void reportError(const char *) = 0; };
namespace Ptex {
typedef std::string String;
struct Info { int numFaces = 0; };
class PtexTexture {
  public:
    int numChannels() { return 0; }
    Info getInfo() { return Info(); }
    void release() {}
};
class PtexCache {
  public:
    struct Stats { size_t memUsed = 0, peakMemUsed = 0, filesOpen = 0, peakFilesOpen = 0,
                   filesAccessed = 0, fileReopens = 0, blockReads = 0; };
    static PtexCache *create(int, size_t, bool, void *, PtexErrorHandler *) { static PtexCache c; return &c; }
    PtexTexture *get(const char *, String &err) { err = "Ptex unavailable in oracle build"; return nullptr; }
    void getStats(Stats &) {}
};
class PtexFilter {
  public:
    enum FilterType { f_point, f_bilinear, f_box, f_gaussian, f_bicubic, f_bspline };
    struct Options { FilterType filter; Options(FilterType f = f_box) : filter(f) {} };
    static PtexFilter *getFilter(PtexTexture *, const Options &) { return nullptr; }
    void eval(float *, int, int, int, float, float, float, float, float, float) {}
    void release() {}
};
}  // namespace Ptex
