// spectra.h — host-side spectral substrate: standard tables, named spectra, RGB colour spaces and the
// RGB->sigmoid-polynomial table; flattens Spectrum objects into wf_spectrum descriptors + a float pool.
// Restates (for the host only) util/spectrum.{h,cpp}, util/color.{h,cpp}, util/colorspace.{h,cpp} and the
// build-time generator cmd/rgb2spec_opt.cpp of the reference.
#pragma once

#include "hmath.h"

#include <map>
#include <memory>
#include <string>
#include <vector>

namespace wf {

// Host Spectrum: the seven reference Spectrum types (util/spectrum.h:48-67) as one tagged struct.
struct SpectrumH {
    int type = WF_SPEC_NONE;
    float c = 0;                         // CONSTANT value / BLACKBODY T
    float norm = 0;                      // BLACKBODY normalizationFactor
    float scale = 1;                     // RGB_UNBOUNDED / RGB_ILLUMINANT
    float c0 = 0, c1 = 0, c2 = 0;        // sigmoid polynomial
    int lambda_min = 360, lambda_max = 830;
    std::vector<float> lambdas, values;  // PIECEWISE (lambdas, values) / DENSE (values)
    const SpectrumH *illuminant = nullptr;  // RGB_ILLUMINANT: dense illuminant of the colour space
    float operator()(float lambda) const;
    float MaxValue() const;
    void Scale(float s) { for (float &v : values) v *= s; }
};
using SpectrumP = std::shared_ptr<SpectrumH>;

SpectrumP MakeConstant(float c);
SpectrumP MakePiecewise(const std::vector<float> &l, const std::vector<float> &v);
SpectrumP MakeFromInterleaved(const std::vector<float> &samples, bool normalize);
SpectrumP MakeDense(const SpectrumH &s, int lmin = 360, int lmax = 830);
SpectrumP MakeBlackbody(float T);
float InnerProduct(const SpectrumH &f, const SpectrumH &g);
float SpectrumToPhotometric(const SpectrumH &s);
void SpectrumToXYZ(const SpectrumH &s, float xyz[3]);
SpectrumP DaylightD(float temperature);  // Spectra::D

struct RGBToSpectrumTable {
    static constexpr int res = 64;
    std::vector<float> zNodes;  // [64]
    std::vector<float> coeffs;  // [3][64][64][64][3]
    void Lookup(const float rgb[3], float c[3]) const;  // util/color.cpp:31-68
};

struct ColorSpace {
    std::string name;
    float r[2], g[2], b[2], w[2];
    SpectrumP illuminant;  // dense
    Mat3 XYZFromRGB, RGBFromXYZ;
    const RGBToSpectrumTable *table = nullptr;
    void ToRGBCoeffs(const float rgb[3], float c[3]) const;  // clamps negatives to zero first
    SpectrumP Albedo(const float rgb[3]) const;
    SpectrumP Unbounded(const float rgb[3]) const;
    SpectrumP Illuminant(const float rgb[3]) const;
};

// Global spectral data (Spectra::Init, RGBToSpectrumTable::Init, RGBColorSpace::Init).
class SpectralData {
  public:
    // dataDir holds spectral_tables.txt; rgb2spec tables are generated on first use and cached in cacheDir
    static void Init(const std::string &dataDir, const std::string &cacheDir);
    static const SpectralData &Get();
    SpectrumP X, Y, Z;  // dense CIE matching functions
    std::map<std::string, SpectrumP> named;
    std::map<std::string, std::vector<float>> raw;  // raw tables by name
    SpectrumP Named(const std::string &name) const;
    const ColorSpace *GetColorSpace(const std::string &name) const;
    const ColorSpace *sRGB() const { return GetColorSpace("srgb"); }
    const std::string &DataDir() const { return dataDir; }

  private:
    mutable std::map<std::string, std::unique_ptr<ColorSpace>> colorSpaces;
    mutable std::map<std::string, std::unique_ptr<RGBToSpectrumTable>> tables;
    std::string cacheDir, dataDir;
    const RGBToSpectrumTable *GetTable(const std::string &gamut) const;
};
static constexpr float CIE_Y_integral = 106.856895f;

// cmd/rgb2spec_opt.cpp restated: Gauss-Newton fit of sigmoid-polynomial coefficients (Jakob & Hanika 2019)
void GenerateRGBToSpectrumTable(const std::string &gamut, RGBToSpectrumTable *out);

// Device-side pool builder: interns host spectra as wf_spectrum + floats.
class SpectrumPool {
  public:
    std::vector<wf_spectrum> spectra;
    std::vector<float> data;
    int Add(const SpectrumH &s);           // returns spectrum id
    int AddDense(const SpectrumH &s);      // LookupSpectrum (lights): returns offset of 471 floats
  private:
    std::map<std::vector<float>, int> denseCache;
    int PushDense(const SpectrumH &s);
};

}  // namespace wf
