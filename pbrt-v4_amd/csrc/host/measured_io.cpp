// measured_io.cpp — `Material "measured" "string filename" "x.bsdf"`: reads the tensor file of a measured BRDF (Dupuy & Jakob's format) and lays the
// five interpolants of MeasuredBxDFData out in the scene's table_data (wf_material::measured_table, include/wf_abi.h).
//   * the container: Tensor::Tensor (bxdfs.cpp:730-812) — "tensor_file\0", version 1.0, n_fields, then per field u16 name length, name, u16 rank,
//     u8 dtype, u64 byte offset, rank x u64 shape; every size and offset is checked against the file before it is used
//   * the field set and shapes: MeasuredBxDFData::Create (bxdfs.cpp:868-972)
//   * the tables: the PiecewiseLinear2D<N> constructor (util/sampling.h:1336-1443) — trapezoid conditional / marginal cdfs accumulated in double and
//     normalised per slice for vndf and luminance; ndf, sigma and spectra only rescaled by 1 / ((nx - 1)(ny - 1))
// The device side (csrc/common/wf_measured.h) reads the header words and evaluates Sample / Invert / Evaluate.
#include "scene.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>

namespace wf {

namespace {
[[noreturn]] void Fail(const std::string &fn, const std::string &why) { throw SceneError("Error: " + fn + ": Tensor: " + why); }

struct Field {
    int dtype = 0;
    std::vector<uint64_t> shape;
    const uint8_t *data = nullptr;
    size_t count = 0;
    const float *f32() const { return reinterpret_cast<const float *>(data); }
};
enum { kUInt8 = 1, kFloat32 = 10, kFloat64 = 11 };
size_t TypeSize(int t) {
    static const size_t sz[12] = {0, 1, 1, 2, 2, 4, 4, 8, 8, 2, 4, 8};
    return (t >= 1 && t <= 11) ? sz[t] : 0;
}

struct Tensor {
    std::vector<uint8_t> bytes;
    std::map<std::string, Field> fields;
    explicit Tensor(const std::string &fn) {
        FILE *f = fopen(fn.c_str(), "rb");
        if (!f) throw SceneError("Error: " + fn + ": unable to open file");
        fseek(f, 0, SEEK_END);
        long size = ftell(f);
        rewind(f);
        if (size < 12 + 2 + 4) { fclose(f); Fail(fn, "Invalid tensor file: too small, truncated?"); }
        bytes.resize((size_t)size);
        size_t got = fread(bytes.data(), 1, bytes.size(), f);
        fclose(f);
        if (got != bytes.size()) Fail(fn, "Unable to read the file.");
        size_t pos = 0;
        auto take = [&](void *dst, size_t n, const char *what) {
            if (n > bytes.size() - pos) Fail(fn, std::string("Unable to read ") + what + ".");
            memcpy(dst, bytes.data() + pos, n);
            pos += n;
        };
        uint8_t header[12], version[2];
        uint32_t nFields;
        take(header, 12, "header"); take(version, 2, "version"); take(&nFields, 4, "n_fields");
        if (memcmp(header, "tensor_file", 12) != 0) Fail(fn, "Invalid tensor file: invalid header.");
        if (version[0] != 1 || version[1] != 0) Fail(fn, "Invalid tensor file: unknown file version.");
        for (uint32_t i = 0; i < nFields; ++i) {
            uint16_t nameLength, ndim;
            uint8_t dtype;
            uint64_t offset;
            take(&nameLength, 2, "name_length");
            std::string name(nameLength, '\0');
            take(&name[0], nameLength, "name");
            take(&ndim, 2, "ndim"); take(&dtype, 1, "dtype"); take(&offset, 8, "offset");
            if (TypeSize(dtype) == 0) Fail(fn, "Invalid tensor file: unknown type.");
            Field fd;
            fd.dtype = dtype;
            fd.count = 1;
            for (int j = 0; j < ndim; ++j) {
                uint64_t s;
                take(&s, 8, "size_value");
                if (s != 0 && fd.count > bytes.size() / s) Fail(fn, "field \"" + name + "\" is larger than the file.");
                fd.count *= (size_t)s;
                fd.shape.push_back(s);
            }
            const size_t total = fd.count * TypeSize(dtype);
            if (offset > bytes.size() || total > bytes.size() - offset) Fail(fn, "Unable to read data.get().");
            fd.data = bytes.data() + offset;
            fields[name] = fd;
        }
    }
    const Field &field(const std::string &fn, const char *name) const {
        auto it = fields.find(name);
        if (it == fields.end()) throw SceneError("Error: " + fn + ": invalid BRDF file structure: no field \"" + name + "\"");
        return it->second;
    }
};

// appends one PiecewiseLinear2D to `table` and writes its 16 header words at table[hdr ...]
void BuildPL2D(std::vector<float> &table, size_t hdr, const float *data, int xs, int ys, int nParams, const int *paramRes, const float *const *paramValues,
               bool normalize, bool buildCdf) {
    auto setWord = [&](size_t k, int32_t v) { memcpy(&table[hdr + k], &v, 4); };
    int32_t paramSize[3] = {1, 1, 1}, paramStride[3] = {0, 0, 0}, paramOff[3] = {-1, -1, -1};
    uint32_t slices = 1;
    for (int i = nParams - 1; i >= 0; --i) {
        paramSize[i] = paramRes[i];
        paramStride[i] = paramRes[i] > 1 ? (int32_t)slices : 0;
        slices *= (uint32_t)paramRes[i];
    }
    for (int i = 0; i < nParams; ++i) {
        paramOff[i] = (int32_t)table.size();
        table.insert(table.end(), paramValues[i], paramValues[i] + paramRes[i]);
    }
    const size_t nValues = (size_t)xs * ys;
    const int32_t dataOff = (int32_t)table.size();
    table.resize(table.size() + slices * nValues);
    int32_t margOff = -1, condOff = -1;
    if (buildCdf) {
        margOff = (int32_t)table.size();
        table.resize(table.size() + (size_t)slices * ys);
        condOff = (int32_t)table.size();
        table.resize(table.size() + slices * nValues);
        for (uint32_t slice = 0; slice < slices; ++slice) {
            const float *in = data + slice * nValues;
            float *out = &table[dataOff + slice * nValues], *marg = &table[margOff + (size_t)slice * ys], *cond = &table[condOff + slice * nValues];
            for (int y = 0; y < ys; ++y) {
                double sum = 0;
                size_t i = (size_t)y * xs;
                cond[i] = 0.f;
                for (int x = 0; x < xs - 1; ++x, ++i) {
                    sum += .5 * ((double)in[i] + (double)in[i + 1]);
                    cond[i + 1] = (float)sum;
                }
            }
            marg[0] = 0.f;
            double sum = 0;
            for (int y = 0; y < ys - 1; ++y) {
                sum += .5 * ((double)cond[(size_t)(y + 1) * xs - 1] + (double)cond[(size_t)(y + 2) * xs - 1]);
                marg[y + 1] = (float)sum;
            }
            const float normalization = 1.f / marg[ys - 1];
            for (size_t i = 0; i < nValues; ++i) cond[i] *= normalization;
            for (int i = 0; i < ys; ++i) marg[i] *= normalization;
            for (size_t i = 0; i < nValues; ++i) out[i] = in[i] * normalization;
        }
    } else {
        for (uint32_t slice = 0; slice < slices; ++slice) {
            const float *in = data + slice * nValues;
            float *out = &table[dataOff + slice * nValues];
            float normalization = 1.f / ((float)(xs - 1) * (float)(ys - 1));
            if (normalize) {
                double sum = 0;
                for (int y = 0; y < ys - 1; ++y) {
                    size_t i = (size_t)y * xs;
                    for (int x = 0; x < xs - 1; ++x, ++i) {
                        float avg = .25f * (in[i] + in[i + 1] + in[i + xs] + in[i + 1 + xs]);
                        sum += (double)avg;
                    }
                }
                normalization = float(1.0 / sum);
            }
            for (size_t k = 0; k < nValues; ++k) out[k] = in[k] * normalization;
        }
    }
    setWord(0, xs); setWord(1, ys);
    for (int i = 0; i < 3; ++i) { setWord(2 + i, paramSize[i]); setWord(5 + i, paramStride[i]); setWord(8 + i, paramOff[i]); }
    setWord(11, dataOff); setWord(12, margOff); setWord(13, condOff);
    setWord(14, nParams); setWord(15, 0);
}
}  // namespace

// MeasuredBxDFData::Create (bxdfs.cpp:868-972); returns the offset of the header in `table`
int ReadMeasuredBRDF(const std::string &filename, std::vector<float> *tablePtr) {
    std::vector<float> &table = *tablePtr;
    Tensor tf(filename);
    const Field &theta_i = tf.field(filename, "theta_i"), &phi_i = tf.field(filename, "phi_i"), &ndf = tf.field(filename, "ndf"), &sigma = tf.field(filename, "sigma"),
                &vndf = tf.field(filename, "vndf"), &spectra = tf.field(filename, "spectra"), &luminance = tf.field(filename, "luminance"),
                &wavelengths = tf.field(filename, "wavelengths"), &description = tf.field(filename, "description"), &jacobian = tf.field(filename, "jacobian");
    const bool ok = description.shape.size() == 1 && description.dtype == kUInt8 && theta_i.shape.size() == 1 && theta_i.dtype == kFloat32 &&
                    phi_i.shape.size() == 1 && phi_i.dtype == kFloat32 && wavelengths.shape.size() == 1 && wavelengths.dtype == kFloat32 &&
                    ndf.shape.size() == 2 && ndf.dtype == kFloat32 && sigma.shape.size() == 2 && sigma.dtype == kFloat32 && vndf.shape.size() == 4 &&
                    vndf.dtype == kFloat32 && vndf.shape[0] == phi_i.shape[0] && vndf.shape[1] == theta_i.shape[0] && luminance.shape.size() == 4 &&
                    luminance.dtype == kFloat32 && luminance.shape[0] == phi_i.shape[0] && luminance.shape[1] == theta_i.shape[0] &&
                    luminance.shape[2] == luminance.shape[3] && spectra.dtype == kFloat32 && spectra.shape.size() == 5 && spectra.shape[0] == phi_i.shape[0] &&
                    spectra.shape[1] == theta_i.shape[0] && spectra.shape[2] == wavelengths.shape[0] && spectra.shape[3] == spectra.shape[4] &&
                    luminance.shape[2] == spectra.shape[3] && luminance.shape[3] == spectra.shape[4] && jacobian.shape.size() == 1 && jacobian.shape[0] == 1 &&
                    jacobian.dtype == kUInt8;
    if (!ok) throw SceneError("Error: " + filename + ": invalid BRDF file structure");
    // beyond the reference's checks: every interpolant needs a 2 x 2 grid at least (its patch size is 1 / (n - 1)) and a parameter axis one value
    for (const Field *f : {&ndf, &sigma})
        if (f->shape[0] < 2 || f->shape[1] < 2) throw SceneError("Error: " + filename + ": invalid BRDF file structure: ndf / sigma smaller than 2 x 2");
    if (vndf.shape[2] < 2 || vndf.shape[3] < 2 || luminance.shape[2] < 2 || phi_i.shape[0] < 1 || theta_i.shape[0] < 1 || wavelengths.shape[0] < 1)
        throw SceneError("Error: " + filename + ": invalid BRDF file structure: empty axis");
    const bool isotropic = phi_i.shape[0] <= 2;
    if (!isotropic) {
        const float *p = phi_i.f32();
        int reduction = (int)std::rint((2 * Pi) / (p[phi_i.shape[0] - 1] - p[0]));
        if (reduction != 1) throw SceneError("Error: " + filename + ": reduction " + std::to_string(reduction) + " (!= 1) not supported");
    }
    const size_t hdr = table.size();
    table.resize(hdr + WF_MEASURED_HEADER_WORDS, 0.f);
    const int32_t iso = isotropic ? 1 : 0;
    memcpy(&table[hdr], &iso, 4);
    const int res2[2] = {(int)phi_i.shape[0], (int)theta_i.shape[0]};
    const float *const val2[2] = {phi_i.f32(), theta_i.f32()};
    const int res3[3] = {(int)phi_i.shape[0], (int)theta_i.shape[0], (int)wavelengths.shape[0]};
    const float *const val3[3] = {phi_i.f32(), theta_i.f32(), wavelengths.f32()};
    BuildPL2D(table, hdr + 16, ndf.f32(), (int)ndf.shape[1], (int)ndf.shape[0], 0, nullptr, nullptr, false, false);
    BuildPL2D(table, hdr + 32, sigma.f32(), (int)sigma.shape[1], (int)sigma.shape[0], 0, nullptr, nullptr, false, false);
    BuildPL2D(table, hdr + 48, vndf.f32(), (int)vndf.shape[3], (int)vndf.shape[2], 2, res2, val2, true, true);
    BuildPL2D(table, hdr + 64, luminance.f32(), (int)luminance.shape[3], (int)luminance.shape[2], 2, res2, val2, true, true);
    BuildPL2D(table, hdr + 80, spectra.f32(), (int)spectra.shape[4], (int)spectra.shape[3], 3, res3, val3, false, false);
    if (table.size() > (size_t)INT32_MAX) throw SceneError("Error: " + filename + ": measured BRDF tables exceed the 2^31-float table_data");
    return (int)hdr;
}

}  // namespace wf
