// spectra.cpp — see spectra.h.  Reference: util/spectrum.cpp, util/color.cpp, util/colorspace.cpp,
// cmd/rgb2spec_opt.cpp (all /root/reference/src/pbrt/).
#include "spectra.h"

#include <sys/stat.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <fstream>
#include <mutex>
#include <sstream>
#include <thread>

namespace wf {

// ---- Spectrum evaluation ------------------------------------------------------------------------
float SpectrumH::operator()(float lambda) const {
    switch (type) {
    case WF_SPEC_CONSTANT: return c;
    case WF_SPEC_DENSE: {  // util/spectrum.h:430-437
        int offset = (int)std::lround(lambda) - lambda_min;
        if (offset < 0 || offset >= (int)values.size()) return 0;
        return values[offset];
    }
    case WF_SPEC_PIECEWISE: {  // util/spectrum.cpp:66-77
        if (lambdas.empty() || lambda < lambdas.front() || lambda > lambdas.back()) return 0;
        int o = FindInterval((int)lambdas.size(), [&](int i) { return lambdas[i] <= lambda; });
        float t = (lambda - lambdas[o]) / (lambdas[o + 1] - lambdas[o]);
        return Lerp(t, values[o], values[o + 1]);
    }
    case WF_SPEC_RGB_ALBEDO: return SigmoidPoly(lambda, c0, c1, c2);
    case WF_SPEC_RGB_UNBOUNDED: return scale * SigmoidPoly(lambda, c0, c1, c2);
    case WF_SPEC_RGB_ILLUMINANT:
        if (!illuminant) return 0;
        return scale * SigmoidPoly(lambda, c0, c1, c2) * (*illuminant)(lambda);
    case WF_SPEC_BLACKBODY: return Blackbody(lambda, c) * norm;
    default: return 0;
    }
}

static float SigmoidMax(float c0, float c1, float c2) {  // util/color.h:345-351
    float result = std::max(SigmoidPoly(360, c0, c1, c2), SigmoidPoly(830, c0, c1, c2));
    float lambda = -c1 / (2 * c0);
    if (lambda >= 360 && lambda <= 830) result = std::max(result, SigmoidPoly(lambda, c0, c1, c2));
    return result;
}

float SpectrumH::MaxValue() const {
    switch (type) {
    case WF_SPEC_CONSTANT: return c;
    case WF_SPEC_DENSE:
    case WF_SPEC_PIECEWISE: return values.empty() ? 0 : *std::max_element(values.begin(), values.end());
    case WF_SPEC_RGB_ALBEDO: return SigmoidMax(c0, c1, c2);
    case WF_SPEC_RGB_UNBOUNDED: return scale * SigmoidMax(c0, c1, c2);
    case WF_SPEC_RGB_ILLUMINANT: return illuminant ? scale * SigmoidMax(c0, c1, c2) * illuminant->MaxValue() : 0;
    case WF_SPEC_BLACKBODY: return 1.f;
    default: return 0;
    }
}

SpectrumP MakeConstant(float c) {
    auto s = std::make_shared<SpectrumH>();
    s->type = WF_SPEC_CONSTANT; s->c = c;
    return s;
}
SpectrumP MakePiecewise(const std::vector<float> &l, const std::vector<float> &v) {
    auto s = std::make_shared<SpectrumH>();
    s->type = WF_SPEC_PIECEWISE; s->lambdas = l; s->values = v;
    return s;
}
SpectrumP MakeDense(const SpectrumH &src, int lmin, int lmax) {
    auto s = std::make_shared<SpectrumH>();
    s->type = WF_SPEC_DENSE; s->lambda_min = lmin; s->lambda_max = lmax;
    s->values.resize(lmax - lmin + 1);
    for (int lambda = lmin; lambda <= lmax; ++lambda) s->values[lambda - lmin] = src((float)lambda);
    return s;
}
SpectrumP MakeBlackbody(float T) {  // util/spectrum.h:486-492
    auto s = std::make_shared<SpectrumH>();
    s->type = WF_SPEC_BLACKBODY; s->c = T;
    float lambdaMax = 2.8977721e-3f / T;
    s->norm = 1 / Blackbody(lambdaMax * 1e9f, T);
    return s;
}
float InnerProduct(const SpectrumH &f, const SpectrumH &g) {  // util/spectrum.h:765-770
    float integral = 0;
    for (float lambda = 360; lambda <= 830; ++lambda) integral += f(lambda) * g(lambda);
    return integral;
}
SpectrumP MakeFromInterleaved(const std::vector<float> &samples, bool normalize) {  // util/spectrum.cpp:130-160
    int n = (int)samples.size() / 2;
    std::vector<float> lambda, v;
    if (samples[0] > 360.f) { lambda.push_back(360.f - 1); v.push_back(samples[1]); }
    for (int i = 0; i < n; ++i) { lambda.push_back(samples[2 * i]); v.push_back(samples[2 * i + 1]); }
    if (lambda.back() < 830.f) { lambda.push_back(830.f + 1); v.push_back(v.back()); }
    SpectrumP spec = MakePiecewise(lambda, v);
    if (normalize) spec->Scale(CIE_Y_integral / InnerProduct(*spec, *SpectralData::Get().Y));
    return spec;
}
float SpectrumToPhotometric(const SpectrumH &s) {  // util/spectrum.cpp:37-47
    const SpectrumH *p = &s;
    if (s.type == WF_SPEC_RGB_ILLUMINANT) p = s.illuminant;
    return InnerProduct(*SpectralData::Get().Y, *p);
}
void SpectrumToXYZ(const SpectrumH &s, float xyz[3]) {  // util/spectrum.cpp:49-53
    const SpectralData &sd = SpectralData::Get();
    xyz[0] = InnerProduct(*sd.X, s) / CIE_Y_integral;
    xyz[1] = InnerProduct(*sd.Y, s) / CIE_Y_integral;
    xyz[2] = InnerProduct(*sd.Z, s) / CIE_Y_integral;
}
SpectrumP DaylightD(float temperature) {  // util/spectrum.cpp:2533-2566
    const SpectralData &sd = SpectralData::Get();
    float cct = temperature * 1.4388f / 1.4380f;
    if (cct < 4000) {
        SpectrumP bb = MakeBlackbody(cct);
        return MakeDense(*bb);
    }
    float x;
    auto Pow3 = [](float v) { float n2 = v; return n2 * n2 * v; };  // Pow<3>(v) = Pow<1>(v)^2 * v
    if (cct <= 7000)
        x = -4.607f * 1e9f / Pow3(cct) + 2.9678f * 1e6f / Sqr(cct) + 0.09911f * 1e3f / cct + 0.244063f;
    else
        x = -2.0064f * 1e9f / Pow3(cct) + 1.9018f * 1e6f / Sqr(cct) + 0.24748f * 1e3f / cct + 0.23704f;
    float y = -3 * x * x + 2.870f * x - 0.275f;
    float M = 0.0241f + 0.2562f * x - 0.7341f * y;
    float M1 = (-1.3515f - 1.7703f * x + 5.9114f * y) / M;
    float M2 = (0.0300f - 31.4424f * x + 30.0717f * y) / M;
    const auto &S0 = sd.raw.at("CIE_S0"), &S1 = sd.raw.at("CIE_S1"), &S2 = sd.raw.at("CIE_S2");
    std::vector<float> values(S0.size());
    for (size_t i = 0; i < S0.size(); ++i) values[i] = (S0[i] + S1[i] * M1 + S2[i] * M2) * 0.01;
    SpectrumP dpls = MakePiecewise(sd.raw.at("CIE_S_lambda"), values);
    return MakeDense(*dpls);
}

// ---- RGB -> spectrum table lookup (util/color.cpp:31-68) ------------------------------------------
void RGBToSpectrumTable::Lookup(const float rgb[3], float c[3]) const {
    if (rgb[0] == rgb[1] && rgb[1] == rgb[2]) {
        c[0] = 0; c[1] = 0;
        c[2] = (rgb[0] - .5f) / std::sqrt(rgb[0] * (1 - rgb[0]));
        return;
    }
    int maxc = (rgb[0] > rgb[1]) ? ((rgb[0] > rgb[2]) ? 0 : 2) : ((rgb[1] > rgb[2]) ? 1 : 2);
    float z = rgb[maxc];
    float x = rgb[(maxc + 1) % 3] * (res - 1) / z;
    float y = rgb[(maxc + 2) % 3] * (res - 1) / z;
    int xi = std::min((int)x, res - 2), yi = std::min((int)y, res - 2),
        zi = FindInterval(res, [&](int i) { return zNodes[i] < z; });
    float dx = x - xi, dy = y - yi, dz = (z - zNodes[zi]) / (zNodes[zi + 1] - zNodes[zi]);
    for (int i = 0; i < 3; ++i) {
        auto co = [&](int ddx, int ddy, int ddz) {
            return coeffs[((((size_t)maxc * res + (zi + ddz)) * res + (yi + ddy)) * res + (xi + ddx)) * 3 + i];
        };
        c[i] = Lerp(dz, Lerp(dy, Lerp(dx, co(0, 0, 0), co(1, 0, 0)), Lerp(dx, co(0, 1, 0), co(1, 1, 0))),
                    Lerp(dy, Lerp(dx, co(0, 0, 1), co(1, 0, 1)), Lerp(dx, co(0, 1, 1), co(1, 1, 1))));
    }
}

void ColorSpace::ToRGBCoeffs(const float rgb[3], float c[3]) const {
    float cl[3] = {std::max(0.f, rgb[0]), std::max(0.f, rgb[1]), std::max(0.f, rgb[2])};
    table->Lookup(cl, c);
}
SpectrumP ColorSpace::Albedo(const float rgb[3]) const {
    auto s = std::make_shared<SpectrumH>();
    s->type = WF_SPEC_RGB_ALBEDO;
    float c[3];
    ToRGBCoeffs(rgb, c);
    s->c0 = c[0]; s->c1 = c[1]; s->c2 = c[2];
    return s;
}
static void ScaledCoeffs(const ColorSpace &cs, const float rgb[3], SpectrumH *s) {  // util/spectrum.cpp:235-246
    float m = std::max({rgb[0], rgb[1], rgb[2]});
    s->scale = 2 * m;
    float in[3] = {0, 0, 0};
    if (s->scale) { in[0] = rgb[0] / s->scale; in[1] = rgb[1] / s->scale; in[2] = rgb[2] / s->scale; }
    float c[3];
    cs.ToRGBCoeffs(in, c);
    s->c0 = c[0]; s->c1 = c[1]; s->c2 = c[2];
}
SpectrumP ColorSpace::Unbounded(const float rgb[3]) const {
    auto s = std::make_shared<SpectrumH>();
    s->type = WF_SPEC_RGB_UNBOUNDED;
    ScaledCoeffs(*this, rgb, s.get());
    return s;
}
SpectrumP ColorSpace::Illuminant(const float rgb[3]) const {
    auto s = std::make_shared<SpectrumH>();
    s->type = WF_SPEC_RGB_ILLUMINANT;
    s->illuminant = illuminant.get();
    ScaledCoeffs(*this, rgb, s.get());
    return s;
}

// ---- global data ---------------------------------------------------------------------------------
static SpectralData *g_sd = nullptr;
const SpectralData &SpectralData::Get() {
    if (!g_sd) { fprintf(stderr, "SpectralData::Init was not called\n"); abort(); }
    return *g_sd;
}
SpectrumP SpectralData::Named(const std::string &name) const {
    auto it = named.find(name);
    return it == named.end() ? nullptr : it->second;
}

void SpectralData::Init(const std::string &dataDir, const std::string &cacheDir) {
    if (g_sd) return;
    SpectralData *sd = new SpectralData;
    g_sd = sd;
    sd->cacheDir = cacheDir;
    sd->dataDir = dataDir;
    for (const char *file : {"/spectral_tables.txt", "/sensor_tables.txt"}) {
        std::ifstream in(dataDir + file);
        if (!in) { fprintf(stderr, "cannot open %s%s\n", dataDir.c_str(), file); exit(1); }
        std::string line;
        while (std::getline(in, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::istringstream hs(line);
            std::string name; int n;
            hs >> name >> n;
            std::vector<float> v(n);
            for (int i = 0; i < n; ++i) { std::string tok; in >> tok; v[i] = strtof(tok.c_str(), nullptr); }
            std::getline(in, line);
            sd->raw[name] = v;
        }
    }
    // Spectra::Init (util/spectrum.cpp:2585-2596)
    const auto &lam = sd->raw.at("CIE_lambda");
    sd->X = MakeDense(*MakePiecewise(lam, sd->raw.at("CIE_X")));
    sd->Y = MakeDense(*MakePiecewise(lam, sd->raw.at("CIE_Y")));
    sd->Z = MakeDense(*MakePiecewise(lam, sd->raw.at("CIE_Z")));
    auto ill = [&](const char *table, const char *name) { sd->named[name] = MakeFromInterleaved(sd->raw.at(table), true); };
    auto plain = [&](const char *table, const char *name) { sd->named[name] = MakeFromInterleaved(sd->raw.at(table), false); };
    ill("CIE_Illum_A", "stdillum-A"); ill("CIE_Illum_D5000", "stdillum-D50"); ill("CIE_Illum_D6500", "stdillum-D65");
    ill("ACES_Illum_D60", "illum-acesD60");
    for (int i = 1; i <= 12; ++i) {
        std::string t = "CIE_Illum_F" + std::to_string(i), n = "stdillum-F" + std::to_string(i);
        ill(t.c_str(), n.c_str());
    }
    plain("GlassBK7_eta", "glass-BK7"); plain("GlassBAF10_eta", "glass-BAF10"); plain("GlassFK51A_eta", "glass-FK51A");
    plain("GlassLASF9_eta", "glass-LASF9"); plain("GlassSF5_eta", "glass-F5"); plain("GlassSF10_eta", "glass-F10");
    plain("GlassSF11_eta", "glass-F11");
    for (const char *m : {"Ag", "Al", "Au", "Cu", "CuZn", "MgO", "TiO2"}) {
        std::string e = std::string(m) + "_eta", k = std::string(m) + "_k";
        plain(e.c_str(), ("metal-" + std::string(m) + "-eta").c_str());
        plain(k.c_str(), ("metal-" + std::string(m) + "-k").c_str());
    }
    // the camera sensors' response curves (util/spectrum.cpp:2700-2830: "<sensor>_r", "_g", "_b"), data/sensor_tables.txt
    for (const auto &kv : sd->raw) {
        const std::string &n = kv.first;
        if (n.size() > 2 && n[n.size() - 2] == '_' && (n.back() == 'r' || n.back() == 'g' || n.back() == 'b') && islower((unsigned char)n[0]))
            sd->named[n] = MakeFromInterleaved(kv.second, false);
    }
}

static void FromxyY(const float xy[2], float out[3]) {
    float Y = 1;
    if (xy[1] == 0) { out[0] = out[1] = out[2] = 0; return; }
    out[0] = xy[0] * Y / xy[1]; out[1] = Y; out[2] = (1 - xy[0] - xy[1]) * Y / xy[1];
}

const ColorSpace *SpectralData::GetColorSpace(const std::string &n) const {
    std::string name;
    for (char ch : n) name.push_back((char)tolower(ch));
    auto it = colorSpaces.find(name);
    if (it != colorSpaces.end()) return it->second.get();
    struct Def { const char *name; float r[2], g[2], b[2]; const char *illum; const char *gamut; };
    static const Def defs[] = {  // util/colorspace.cpp:77-95
        {"srgb", {.64f, .33f}, {.3f, .6f}, {.15f, .06f}, "stdillum-D65", "sRGB"},
        {"dci-p3", {.68f, .32f}, {.265f, .690f}, {.15f, .06f}, "stdillum-D65", "DCI_P3"},
        {"rec2020", {.708f, .292f}, {.170f, .797f}, {.131f, .046f}, "stdillum-D65", "REC2020"},
        {"aces2065-1", {.7347f, .2653f}, {0.f, 1.f}, {.0001f, -.077f}, "illum-acesD60", "ACES2065_1"}};
    for (const Def &d : defs) {
        if (name != d.name) continue;
        auto cs = std::make_unique<ColorSpace>();
        cs->name = name;
        std::memcpy(cs->r, d.r, 8); std::memcpy(cs->g, d.g, 8); std::memcpy(cs->b, d.b, 8);
        SpectrumP illum = Named(d.illum);
        cs->illuminant = MakeDense(*illum);
        // util/colorspace.cpp:21-35
        float W[3];
        SpectrumToXYZ(*illum, W);
        cs->w[0] = W[0] / (W[0] + W[1] + W[2]);
        cs->w[1] = W[1] / (W[0] + W[1] + W[2]);
        float R[3], G[3], B[3];
        FromxyY(cs->r, R); FromxyY(cs->g, G); FromxyY(cs->b, B);
        Mat3 rgb = {{{R[0], G[0], B[0]}, {R[1], G[1], B[1]}, {R[2], G[2], B[2]}}};
        Mat3 inv;
        if (!Inverse(rgb, &inv)) { fprintf(stderr, "colour space matrix is singular\n"); exit(1); }
        float C[3];
        Mul3(inv, W, C);
        Mat3 diag = {{{C[0], 0, 0}, {0, C[1], 0}, {0, 0, C[2]}}};
        cs->XYZFromRGB = rgb * diag;
        if (!Inverse(cs->XYZFromRGB, &cs->RGBFromXYZ)) { fprintf(stderr, "colour space matrix is singular\n"); exit(1); }
        cs->table = GetTable(d.gamut);
        const ColorSpace *ret = cs.get();
        colorSpaces[name] = std::move(cs);
        return ret;
    }
    return nullptr;
}

const RGBToSpectrumTable *SpectralData::GetTable(const std::string &gamut) const {
    auto it = tables.find(gamut);
    if (it != tables.end()) return it->second.get();
    auto t = std::make_unique<RGBToSpectrumTable>();
    std::string path = cacheDir + "/rgb2spec_" + gamut + ".bin";
    const size_t nz = 64, nc = (size_t)3 * 64 * 64 * 64 * 3;
    bool ok = false;
    if (FILE *f = fopen(path.c_str(), "rb")) {
        t->zNodes.resize(nz); t->coeffs.resize(nc);
        ok = fread(t->zNodes.data(), 4, nz, f) == nz && fread(t->coeffs.data(), 4, nc, f) == nc;
        fclose(f);
    }
    if (!ok) {
        GenerateRGBToSpectrumTable(gamut, t.get());
        ::mkdir(cacheDir.c_str(), 0755);
        if (FILE *f = fopen(path.c_str(), "wb")) {
            fwrite(t->zNodes.data(), 4, nz, f);
            fwrite(t->coeffs.data(), 4, nc, f);
            fclose(f);
        }
    }
    const RGBToSpectrumTable *ret = t.get();
    tables[gamut] = std::move(t);
    return ret;
}

// ---- RGB -> spectrum table generation --------------------------------------------------------------
// Restates cmd/rgb2spec_opt.cpp:347-640,1190-1260: for each maximum channel l and grid point (x,y,z=scale[k])
// fit sigmoid(c0 λ'^2 + c1 λ' + c2) (λ' in [0,1]) to the target RGB in CIELAB with Gauss-Newton, warm-started
// along k, then re-express the polynomial in nanometres.  All arithmetic is double precision in the same
// order as the tool, so the float table is bit-identical to the one the reference builds at compile time.
namespace {
struct R2S {
    static constexpr int CIE_SAMPLES = 95, FINE = (CIE_SAMPLES - 1) * 3 + 1;
    double cie_x[CIE_SAMPLES], cie_y[CIE_SAMPLES], cie_z[CIE_SAMPLES], illum[CIE_SAMPLES];
    double lambda_tbl[FINE], rgb_tbl[3][FINE], rgb_to_xyz[3][3], xyz_to_rgb[3][3], xyz_whitepoint[3];

    static double interp(const double *data, double x) {
        x -= 360.0;
        x *= (CIE_SAMPLES - 1) / (830.0 - 360.0);
        int offset = (int)x;
        if (offset < 0) offset = 0;
        if (offset > CIE_SAMPLES - 2) offset = CIE_SAMPLES - 2;
        double weight = x - offset;
        return (1.0 - weight) * data[offset] + weight * data[offset + 1];
    }
    static double sigmoid(double x) { return 0.5 * x / std::sqrt(1.0 + x * x) + 0.5; }
    static double sqr(double x) { return x * x; }
    void cie_lab(double *p) const {
        double X = 0.0, Y = 0.0, Z = 0.0, Xw = xyz_whitepoint[0], Yw = xyz_whitepoint[1], Zw = xyz_whitepoint[2];
        for (int j = 0; j < 3; ++j) {
            X += p[j] * rgb_to_xyz[0][j];
            Y += p[j] * rgb_to_xyz[1][j];
            Z += p[j] * rgb_to_xyz[2][j];
        }
        auto f = [](double t) -> double {
            double delta = 6.0 / 29.0;
            if (t > delta * delta * delta) return cbrt(t);
            else return t / (delta * delta * 3.0) + (4.0 / 29.0);
        };
        p[0] = 116.0 * f(Y / Yw) - 16.0;
        p[1] = 500.0 * (f(X / Xw) - f(Y / Yw));
        p[2] = 200.0 * (f(Y / Yw) - f(Z / Zw));
    }
    void init_tables() {
        std::memset(rgb_tbl, 0, sizeof(rgb_tbl));
        std::memset(xyz_whitepoint, 0, sizeof(xyz_whitepoint));
        double h = (830.0 - 360.0) / (FINE - 1);
        for (int i = 0; i < FINE; ++i) {
            double lambda = 360.0 + i * h;
            double xyz[3] = {interp(cie_x, lambda), interp(cie_y, lambda), interp(cie_z, lambda)}, I = interp(illum, lambda);
            double weight = 3.0 / 8.0 * h;
            if (i == 0 || i == FINE - 1) ;
            else if ((i - 1) % 3 == 2) weight *= 2.f;
            else weight *= 3.f;
            lambda_tbl[i] = lambda;
            for (int k = 0; k < 3; ++k)
                for (int j = 0; j < 3; ++j) rgb_tbl[k][i] += xyz_to_rgb[k][j] * xyz[j] * I * weight;
            for (int k = 0; k < 3; ++k) xyz_whitepoint[k] += xyz[k] * I * weight;
        }
    }
    void eval_residual(const double *coeffs, const double *rgb, double *residual) const {
        double out[3] = {0.0, 0.0, 0.0};
        for (int i = 0; i < FINE; ++i) {
            double lambda = (lambda_tbl[i] - 360.0) / (830.0 - 360.0);
            double x = 0.0;
            for (int k = 0; k < 3; ++k) x = x * lambda + coeffs[k];
            double s = sigmoid(x);
            for (int j = 0; j < 3; ++j) out[j] += rgb_tbl[j][i] * s;
        }
        cie_lab(out);
        std::memcpy(residual, rgb, sizeof(double) * 3);
        cie_lab(residual);
        for (int j = 0; j < 3; ++j) residual[j] -= out[j];
    }
    void eval_jacobian(const double *coeffs, const double *rgb, double **jac) const {
        const double eps = 1e-4;
        double r0[3], r1[3], tmp[3];
        for (int i = 0; i < 3; ++i) {
            std::memcpy(tmp, coeffs, sizeof(double) * 3);
            tmp[i] -= eps;
            eval_residual(tmp, rgb, r0);
            std::memcpy(tmp, coeffs, sizeof(double) * 3);
            tmp[i] += eps;
            eval_residual(tmp, rgb, r1);
            for (int j = 0; j < 3; ++j) jac[j][i] = (r1[j] - r0[j]) * 1.0 / (2 * eps);
        }
    }
    static int LUPDecompose(double **A, int N, double Tol, int *P) {
        int i, j, k, imax;
        double maxA, *ptr, absA;
        for (i = 0; i <= N; i++) P[i] = i;
        for (i = 0; i < N; i++) {
            maxA = 0.0; imax = i;
            for (k = i; k < N; k++)
                if ((absA = fabs(A[k][i])) > maxA) { maxA = absA; imax = k; }
            if (maxA < Tol) return 0;
            if (imax != i) {
                j = P[i]; P[i] = P[imax]; P[imax] = j;
                ptr = A[i]; A[i] = A[imax]; A[imax] = ptr;
                P[N]++;
            }
            for (j = i + 1; j < N; j++) {
                A[j][i] /= A[i][i];
                for (k = i + 1; k < N; k++) A[j][k] -= A[j][i] * A[i][k];
            }
        }
        return 1;
    }
    static void LUPSolve(double **const A, const int *P, const double *b, int N, double *x) {
        for (int i = 0; i < N; i++) {
            x[i] = b[P[i]];
            for (int k = 0; k < i; k++) x[i] -= A[i][k] * x[k];
        }
        for (int i = N - 1; i >= 0; i--) {
            for (int k = i + 1; k < N; k++) x[i] -= A[i][k] * x[k];
            x[i] = x[i] / A[i][i];
        }
    }
    void gauss_newton(const double rgb[3], double coeffs[3], int it = 15) const {
        double r = 0;
        for (int i = 0; i < it; ++i) {
            double J0[3], J1[3], J2[3], *J[3] = {J0, J1, J2};
            double residual[3];
            eval_residual(coeffs, rgb, residual);
            eval_jacobian(coeffs, rgb, J);
            int P[4];
            if (LUPDecompose(J, 3, 1e-15, P) != 1) { fprintf(stderr, "rgb2spec: LU decomposition failed\n"); exit(1); }
            double x[3];
            LUPSolve(J, P, residual, 3, x);
            r = 0.0;
            for (int j = 0; j < 3; ++j) { coeffs[j] -= x[j]; r += residual[j] * residual[j]; }
            double max = std::max(std::max(coeffs[0], coeffs[1]), coeffs[2]);
            if (max > 200) for (int j = 0; j < 3; ++j) coeffs[j] *= 200 / max;
            if (r < 1e-6) break;
        }
    }
};
double smoothstep(double x) { return x * x * (3.0 - 2.0 * x); }
}  // namespace

void GenerateRGBToSpectrumTable(const std::string &gamut, RGBToSpectrumTable *out) {
    const SpectralData &sd = SpectralData::Get();
    auto r2s = std::make_unique<R2S>();
    // 5 nm samples of the colour matching functions = every fifth entry of the 1 nm tables
    const auto &X = sd.raw.at("CIE_X"), &Y = sd.raw.at("CIE_Y"), &Z = sd.raw.at("CIE_Z");
    // the tool's tables are decimal literals read as double; re-read ours from text for the same doubles
    std::map<std::string, std::vector<double>> dbl;
    {
        // (re-parse as double: float->double widening of the float-rounded value would differ)
        std::ifstream in(sd.DataDir() + "/spectral_tables.txt");
        std::string line;
        while (std::getline(in, line)) {
            if (line.empty() || line[0] == '#') continue;
            std::istringstream hs(line);
            std::string name; int n;
            hs >> name >> n;
            std::vector<double> v(n);
            for (int i = 0; i < n; ++i) { std::string tok; in >> tok; v[i] = strtod(tok.c_str(), nullptr); }
            std::getline(in, line);
            dbl[name] = v;
        }
    }
    (void)X; (void)Y; (void)Z;
    for (int i = 0; i < R2S::CIE_SAMPLES; ++i) {
        r2s->cie_x[i] = dbl["CIE_X"][5 * i];
        r2s->cie_y[i] = dbl["CIE_Y"][5 * i];
        r2s->cie_z[i] = dbl["CIE_Z"][5 * i];
    }
    // illuminant: interleaved (lambda, value) from 300 nm in 5 nm steps; normalised by the tool's constant
    // The standard-illuminant tables in the data file carry float-rounded decimals (46.638302); the
    // generator's own 5 nm table holds the CIE values at their published precision (46.6383).  Re-round to
    // that many decimals so the doubles are the ones the reference tool uses.
    auto fillIllum = [&](const char *table, double norm, int decimals) {
        const auto &t = dbl[table];
        for (int i = 0; i < R2S::CIE_SAMPLES; ++i) {
            double lambda = 360.0 + 5.0 * i;
            double val = 0;
            for (size_t k = 0; k + 1 < t.size(); k += 2)
                if (t[k] == lambda) { val = t[k + 1]; break; }
            char buf[64];
            snprintf(buf, sizeof(buf), "%.*f", decimals, val);
            val = strtod(buf, nullptr);
            r2s->illum[i] = val / norm;
        }
    };
    static const double xyz_to_srgb[3][3] = {{3.240479, -1.537150, -0.498535}, {-0.969256, 1.875991, 0.041556}, {0.055648, -0.204043, 1.057311}};
    static const double srgb_to_xyz[3][3] = {{0.412453, 0.357580, 0.180423}, {0.212671, 0.715160, 0.072169}, {0.019334, 0.119193, 0.950227}};
    static const double xyz_to_rec2020[3][3] = {{1.7166511880, -0.3556707838, -0.2533662814}, {-0.6666843518, 1.6164812366, 0.0157685458}, {0.0176398574, -0.0427706133, 0.9421031212}};
    static const double rec2020_to_xyz[3][3] = {{0.6369580483, 0.1446169036, 0.1688809752}, {0.2627002120, 0.6779980715, 0.0593017165}, {0.0000000000, 0.0280726930, 1.0609850577}};
    static const double xyz_to_dcip3[3][3] = {{2.4931748, -0.93126315, -0.40265882}, {-0.82950425, 1.7626965, 0.023625137}, {0.035853732, -0.07618918, 0.9570952}};
    static const double dcip3_to_xyz[3][3] = {{0.48663378, 0.26566276, 0.19817366}, {0.22900413, 0.69172573, 0.079269454}, {0., 0.04511256, 1.0437145}};
    static const double xyz_to_aces[3][3] = {{1.0498110175, 0.0000000000, -0.0000974845}, {-0.4959030231, 1.3733130458, 0.0982400361}, {0.0000000000, 0.0000000000, 0.9912520182}};
    static const double aces_to_xyz[3][3] = {{0.9525523959, 0.0000000000, 0.0000936786}, {0.3439664498, 0.7281660966, -0.0721325464}, {0.0000000000, 0.0000000000, 1.0088251844}};
    if (gamut == "sRGB") { fillIllum("CIE_Illum_D6500", 10566.864005283874576, 4); std::memcpy(r2s->xyz_to_rgb, xyz_to_srgb, 72); std::memcpy(r2s->rgb_to_xyz, srgb_to_xyz, 72); }
    else if (gamut == "REC2020") { fillIllum("CIE_Illum_D6500", 10566.864005283874576, 4); std::memcpy(r2s->xyz_to_rgb, xyz_to_rec2020, 72); std::memcpy(r2s->rgb_to_xyz, rec2020_to_xyz, 72); }
    else if (gamut == "DCI_P3") { fillIllum("CIE_Illum_D6500", 10566.864005283874576, 4); std::memcpy(r2s->xyz_to_rgb, xyz_to_dcip3, 72); std::memcpy(r2s->rgb_to_xyz, dcip3_to_xyz, 72); }
    else if (gamut == "ACES2065_1") {
        // the tool's own cie_d60 table (cmd/rgb2spec_opt.cpp:166-188, N(x) = x / 10536.3): not the ACES_Illum_D60 of util/spectrum.cpp re-sampled —
        // the two differ in the last digits, and the tool's initialiser list is one short (its 830 nm sample is 0)
        const auto &d60 = dbl["R2S_cie_d60"];
        if ((int)d60.size() != R2S::CIE_SAMPLES) { fprintf(stderr, "rgb2spec: data/spectral_tables.txt lacks R2S_cie_d60 (tools/extract_spectral_tables.py)\n"); exit(1); }
        for (int i = 0; i < R2S::CIE_SAMPLES; ++i) r2s->illum[i] = d60[i] / 10536.3;
        std::memcpy(r2s->xyz_to_rgb, xyz_to_aces, 72); std::memcpy(r2s->rgb_to_xyz, aces_to_xyz, 72);
    }
    else { fprintf(stderr, "rgb2spec: unsupported gamut %s\n", gamut.c_str()); exit(1); }
    r2s->init_tables();

    const int res = RGBToSpectrumTable::res;
    out->zNodes.resize(res);
    for (int k = 0; k < res; ++k) out->zNodes[k] = (float)smoothstep(smoothstep(k / double(res - 1)));
    out->coeffs.assign((size_t)3 * 3 * res * res * res, 0.f);
    const R2S &T = *r2s;
    for (int l = 0; l < 3; ++l) {
        auto work = [&](int j) {
            const double y = j / double(res - 1);
            for (int i = 0; i < res; ++i) {
                const double x = i / double(res - 1);
                double coeffs[3], rgb[3];
                int start = res / 5;
                auto emit = [&](int k) {
                    double b = (double)out->zNodes[k];
                    rgb[l] = b; rgb[(l + 1) % 3] = x * b; rgb[(l + 2) % 3] = y * b;
                    T.gauss_newton(rgb, coeffs);
                    double c0 = 360.0, c1 = 1.0 / (830.0 - 360.0);
                    double A = coeffs[0], B = coeffs[1], C = coeffs[2];
                    size_t idx = (((size_t)l * res + k) * res + j) * res + i;
                    out->coeffs[3 * idx + 0] = float(A * (R2S::sqr(c1)));
                    out->coeffs[3 * idx + 1] = float(B * c1 - 2 * A * c0 * (R2S::sqr(c1)));
                    out->coeffs[3 * idx + 2] = float(C - B * c0 * c1 + A * (R2S::sqr(c0 * c1)));
                };
                std::memset(coeffs, 0, sizeof(coeffs));
                for (int k = start; k < res; ++k) emit(k);
                std::memset(coeffs, 0, sizeof(coeffs));
                for (int k = start; k >= 0; --k) emit(k);
            }
        };
        unsigned nt = std::max(1u, std::thread::hardware_concurrency());
        std::vector<std::thread> threads;
        std::atomic<int> next{0};
        for (unsigned t = 0; t < nt; ++t)
            threads.emplace_back([&]() { for (int j; (j = next.fetch_add(1)) < res;) work(j); });
        for (auto &th : threads) th.join();
    }
}

// ---- device pool ----------------------------------------------------------------------------------
int SpectrumPool::PushDense(const SpectrumH &s) {
    std::vector<float> v(WF_NDENSE);
    for (int lambda = WF_LAMBDA_MIN; lambda <= WF_LAMBDA_MAX; ++lambda) v[lambda - WF_LAMBDA_MIN] = s((float)lambda);
    auto it = denseCache.find(v);
    if (it != denseCache.end()) return it->second;
    int off = (int)data.size();
    data.insert(data.end(), v.begin(), v.end());
    denseCache[v] = off;
    return off;
}
int SpectrumPool::AddDense(const SpectrumH &s) { return PushDense(s); }
int SpectrumPool::Add(const SpectrumH &s) {
    wf_spectrum d{};
    d.type = s.type; d.scale = s.scale; d.c0 = s.c0; d.c1 = s.c1; d.c2 = s.c2; d.offset = -1;
    switch (s.type) {
    case WF_SPEC_CONSTANT: d.c0 = s.c; break;
    case WF_SPEC_DENSE:
        if (s.lambda_min != WF_LAMBDA_MIN || s.lambda_max != WF_LAMBDA_MAX) { fprintf(stderr, "dense spectrum with non-default range\n"); exit(1); }
        d.offset = PushDense(s); d.n = WF_NDENSE; break;
    case WF_SPEC_PIECEWISE:
        d.offset = (int)data.size(); d.n = (int)s.lambdas.size();
        data.insert(data.end(), s.lambdas.begin(), s.lambdas.end());
        data.insert(data.end(), s.values.begin(), s.values.end());
        break;
    case WF_SPEC_RGB_ILLUMINANT: d.offset = PushDense(*s.illuminant); d.n = WF_NDENSE; break;
    case WF_SPEC_BLACKBODY: d.c0 = s.c; d.c1 = s.norm; break;
    default: break;
    }
    spectra.push_back(d);
    return (int)spectra.size() - 1;
}

}  // namespace wf
