// wf_scene.h — the read-only scene as the kernels see it: one struct of raw pointers into the flat,
// index-addressed tables of wf_scene_desc (include/wf_abi.h).  On the GPU the pointers are device
// addresses (csrc/hip/wf_backend.hip uploads them); the CPU checker under oracle/ points the same struct
// at the host tables.  No host pointers, no virtual calls, no tagged pointers cross to the device: the
// reference's TaggedPointer dispatch (util/taggedptr.h:736-870) becomes a switch on an integer tag.
#pragma once

#include <type_traits>
#include "wf_math.h"
#include "wf_noise.h"
#include "../../../include/wf_abi.h"

namespace wf {

// A pointer to one of the scene's tables (round 6).  The tables live in global memory, but a pointer that a kernel LOADS — the kernels read
// the view through its device-resident copy, the out-of-line callees take `const SceneView *` — is a generic pointer to the compiler, and
// every access through it a FLAT instruction: address on both the LDS and the memory path, both wait counters
// (k_mat_shade<diffuse>: 253 flat loads against 20 global ones).  Indexing a GPtr of scalars (vertex data, index triples, texels, tables of
// floats) reads through an address-space-1 pointer instead: a global load.  (An assumption `!is_shared && !is_private` on the pointer is not
// picked up by this toolchain's address-space inference, and a cast to address space 1 and back is folded away: measured on a stand-alone kernel.)
// Tables of records keep their generic references (the records are passed on by reference).  On the host: a plain pointer.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(WF_NO_GPTR)
#define WF_GLOBAL_AS __attribute__((address_space(1)))
#else
#define WF_GLOBAL_AS
#endif
template <typename T>
struct GPtr {
    T *p;
    GPtr() = default;
    WF_HD GPtr(T *q) : p(q) {}
    WF_HD operator T *() const { return p; }
    WF_HD T *operator->() const { return p; }
    template <typename I>
    WF_HD GPtr operator+(I off) const { return GPtr(p + off); }
    // scalars by value through the global address space; records by (generic) reference
    template <typename I, typename U = T>
    WF_HD typename std::enable_if<std::is_arithmetic<U>::value, typename std::remove_const<U>::type>::type operator[](I i) const { return ((WF_GLOBAL_AS T *)p)[i]; }
    template <typename I, typename U = T>
    WF_HD typename std::enable_if<!std::is_arithmetic<U>::value, U &>::type operator[](I i) const { return p[i]; }
};

// (round 6, device only) a triangle's vertex data de-indexed: what the material stage's interaction rebuild gathers — three indices, then three
// positions, normals and (u, v) pairs from three tables, 13 scattered sectors behind a dependent load — as ONE 96-byte record per triangle
// (zeros where the mesh has no normals / no (u, v)).  1 GB for the 10 M-triangle scene: what 288 GB are for.
struct alignas(16) ShadeTri { float p[9], n[9], uv[6]; };
struct SceneView {
    // geometry (util/mesh.h TriangleMesh buffers, flattened over all meshes)
    GPtr<const float> P, N, UV;
    GPtr<const ShadeTri> shadeTris;   // [nTriangles], or null (the host checker; WF_SHADE_TRIS=0)
    GPtr<const float> S;   // shading tangents of the meshes with WF_MESH_HAS_S (LoadS)
    GPtr<const int32_t> triIndices, triMesh;
    GPtr<const wf_mesh> meshes;
    GPtr<const wf_bvh_node> bvhNodes;
    GPtr<const int32_t> bvhPrims;
    int nTriangles, nBvhNodes;
    // shading
    GPtr<const wf_spectrum> spectra;
    GPtr<const float> spectrumData;
    GPtr<const wf_texture> textures;
    GPtr<const wf_material> materials;
    // lights
    GPtr<const wf_light> lights;
    GPtr<const int32_t> infiniteLights;
    GPtr<const wf_light_bvh_node> lightBvh;
    GPtr<const struct LightNodeX> lightBvhX;   // device only: every node's constants expanded once at upload (wf_lights.h ExpandLightNode)
    GPtr<const wf_transform> lightXforms;
    int nLights, nInfiniteLights, nLightBvhNodes, lightSampler;
    GPtr<const float> powerAlias;  // PowerLightSampler's AliasTable bins: nLights x {q, p, alias (int bits)}
    float allLightBounds[6];
    // image infinite lights
    GPtr<const wf_image_light> imageLights;
    GPtr<const wf_tex_image> texImages;
    GPtr<const float> tableData;
    GPtr<const float> rgb2specCoeffs;
    GPtr<const float> rgb2specZNodes;  // [64]
    GPtr<const int32_t> noisePerm;     // [512] Perlin permutation (procedural textures, cloud medium), or null
    int csIlluminantOffset;
    // participating media
    GPtr<const wf_medium> media;
    GPtr<const float> mediumData;
    // (round 4, device only; null on the host) the density grids of the GridMedium media once more, CORNER-PACKED: for every lattice
    // position the eight values a trilinear lookup there reads, 32 contiguous bytes — two dwordx4 loads instead of eight gathers
    // into four rows of the grid, at 8x the memory (4.3 GB for the 512^3 cloud of BASELINE configs[3]; what 288 GB are for).
    // gridCornerBase[medium id] = first float of the medium's table in gridCorners, or -1 (wf_media.h: GridLookupPacked)
    GPtr<const float> gridCorners;
    GPtr<const long long> gridCornerBase;
    // camera / film / filter / sampler
    wf_camera camera;
    wf_film film;
    wf_filter filter;
    GPtr<const float> filterData;
    wf_sampler sampler;
    GPtr<const uint32_t> sobol;  // WF_SOBOL_WORDS: SobolMatrices32 columns of dimensions 0 and 1 (2 x 52), then their byte tables (FillSobol2D)
    // integrator
    int maxDepth, regularize, haveMedia;
    int matTypeMask;  // bit t set: some material has wf_material_type t (which eval queues can be non-empty)
    int texNeedsFootprint;  // some texture's value depends on the TextureEvalContext (checkerboard, image) or some material
                            // is bump- or normal-mapped: selects the material-kernel variant that computes the differentials
    GPtr<const uint32_t> sobolMatrices;   // SobolSampler: SobolMatrices32 [1024][52]
    GPtr<const uint64_t> vdcSobol, vdcSobolInv;
    GPtr<const int32_t> haltonPrimes, haltonPermOffsets;
    GPtr<const uint16_t> haltonPerms;
    GPtr<const wf_quadric> quadrics;
    int nQuadrics;
    // object instances (include/wf_abi.h wf_instance): primitive id nTriangles + nQuadrics + instance index
    GPtr<const wf_instance> instances;
    GPtr<const wf_instance_def> instanceDefs;
    int nInstances;
    // AnimatedPrimitive (round 5): the AnimatedTransforms of animated shapes / instances (wf_instance.anim_plus1); haveAnimated: the scene has one
    GPtr<const wf_animated_transform> animated;
    int haveAnimated;
    int haveQuadricAlpha;   // some sphere / disk / cylinder / patch has an alpha texture (QuadricAlphaIntersectP)
    int haveCurves;         // some primitive is a Curve segment: its interaction is rebuilt from the ray (HitInteraction)
    int haveSubsurface;     // some material is a SubsurfaceMaterial: K12 runs, and every depth draws 3 more sample dimensions
    int haveMix;            // some material is a MixMaterial: hits on it store their resolved material id in ws.mixMat
    int haveAlpha;          // some mesh carries an alpha texture: selects the traversal-kernel variant with the alpha test
    wf_options options;
    // this struct in memory the kernels can read (device memory on the GPU, the object itself on the host): what the few
    // out-of-line device functions take instead of the by-value kernel argument (whose address must never be taken)
    GPtr<const SceneView> self;
    // LOG_FATAL of the reference inside a kernel body (so far: Curve::Sample / Curve::PDF "not implemented", shapes.cpp:736-760, reached
    // when a sample is drawn from an emissive curve): the body stores a WF_FATAL_* code here and carries on with a null result; wf_sync
    // (the CPU checker: the end of its render) turns the code into the reference's fatal error.  Null: nothing to report to.
    GPtr<int32_t> fatal;
};
enum { WF_FATAL_CURVE_SAMPLE = 1, WF_FATAL_CURVE_PDF = 2,   // (bit flags: several may be raised in one render)
       WF_FATAL_CHECK_HAIR = 4,      // HairBxDF ctor: CHECK(h >= -1 && h <= 1) / beta_m / beta_n (bxdfs.cpp:278-280)
       WF_FATAL_CHECK_NAN_PDF = 8,   // DielectricBxDF::Sample_f: CHECK(!IsNaN(pdf)) of the rough transmission (bxdfs.cpp:158)
       WF_FATAL_LEAN_VARIANT = 16 }; // internal: a lean kernel variant (WF_DEV_LEAN) met a texture graph it was built without
WF_HD const char *FatalMessage(int code) {
    return (code & WF_FATAL_LEAN_VARIANT) ? "internal error: a lean kernel variant was launched on a scene it cannot render" : (code & WF_FATAL_CURVE_SAMPLE) ? "Curve::Sample not implemented." : (code & WF_FATAL_CURVE_PDF) ? "Curve::PDF not implemented."
           : (code & WF_FATAL_CHECK_HAIR) ? "Check failed: h >= -1 && h <= 1 (HairBxDF)" : (code & WF_FATAL_CHECK_NAN_PDF) ? "Check failed: !IsNaN(pdf) (DielectricBxDF::Sample_f)"
           : "fatal error raised by a kernel";
}
// On the device the flag is raised with an atomic and WITHOUT a null test (the back end always allocates the word).  The obvious form,
// `if (sv.fatal) *sv.fatal = code;`, inlined into the light-sampling code of the material kernels, made k_eval_material<1, 0> corrupt memory
// (round 4: cornell64 rendered a different image on every run and one build faulted; the same source with this body, with the store
// through sv.self->fatal, or without the call in SphereSample / SpherePDF is bit-identical and repeatable — DESIGN.md 4.2).
#if defined(__HIP_DEVICE_COMPILE__)
WF_HD void RaiseFatal(const SceneView &sv, int code) { atomicOr(sv.fatal, code); }
#else
WF_HD void RaiseFatal(const SceneView &sv, int code) { if (sv.fatal) *sv.fatal |= code; }   // (the codes are OR-able bit flags)
#endif

// SobolMatrices32 dimensions 0 and 1 (util/sobolmatrices.cpp:40-58).  Dimension 0 is the van der Corput
// identity matrix, dimension 1 the Pascal-triangle matrix v[i] = v[i-1] ^ (v[i-1] >> 1); both are padded
// to SobolMatrixSize = 52 columns the way the reference table is (zeros, resp. the period-32 repeat).
// The ZSobol sampler (samplers.h:261-291) only ever asks for these two dimensions.
// Behind the 104 columns come the byte tables SobolSample uses for indices below 2^32: the sample value is
// the XOR of the columns selected by the index bits (a GF(2) matrix-vector product), so it is also the XOR of four
// 256-entry tables, one per index byte: lut[dim][k][b] = XOR of columns 8k+j over the set bits j of b.
constexpr int WF_SOBOL_COLUMNS = 104;
constexpr int WF_SOBOL_WORDS = WF_SOBOL_COLUMNS + 2 * 4 * 256;
inline void FillSobol2D(uint32_t out[WF_SOBOL_WORDS]) {
    uint32_t v = 0x80000000u;
    uint32_t d1[32];
    for (int i = 0; i < 32; ++i) { d1[i] = v; v ^= v >> 1; }
    for (int i = 0; i < 52; ++i) {
        out[i] = i < 32 ? (0x80000000u >> i) : 0u;
        out[52 + i] = d1[i & 31];
    }
    for (int dim = 0; dim < 2; ++dim)
        for (int k = 0; k < 4; ++k)
            for (int b = 0; b < 256; ++b) {
                uint32_t x = 0;
                for (int j = 0; j < 8; ++j)
                    if (b & (1 << j)) x ^= out[dim * 52 + 8 * k + j];
                out[WF_SOBOL_COLUMNS + (dim * 4 + k) * 256 + b] = x;
            }
}

// ---------------------------------------------------------------------------------------------
// Spectrum::Sample (util/spectrum.h) over the flattened descriptors
WF_HD S4 DenseSample(const SceneView &sv, int offset, const Wavelengths &lambda) {
    // DenselySampledSpectrum::Sample, util/spectrum.h:387-397 (lambda_min = 360, 471 values)
    S4 s;
    for (int i = 0; i < 4; ++i) {
        int o = (int)lround(lambda.lambda[i]) - WF_LAMBDA_MIN;
        s[i] = (o < 0 || o >= WF_NDENSE) ? 0.f : sv.spectrumData[offset + o];
    }
    return s;
}
WF_HD float PiecewiseEval(const float *lambdas, const float *values, int n, float lambda) {
    // util/spectrum.cpp:64-74
    if (n == 0 || lambda < lambdas[0] || lambda > lambdas[n - 1]) return 0;
    int o = FindInterval(n, [&](int i) { return lambdas[i] <= lambda; });
    float t = (lambda - lambdas[o]) / (lambdas[o + 1] - lambdas[o]);
    return Lerp(t, values[o], values[o + 1]);
}
WF_HD S4 SpectrumSample(const SceneView &sv, int id, const Wavelengths &lambda) {
    const wf_spectrum sp = sv.spectra[id];
    S4 s;
    switch (sp.type) {
    case WF_SPEC_CONSTANT: return S4c(sp.c0);
    case WF_SPEC_DENSE: return DenseSample(sv, sp.offset, lambda);
    case WF_SPEC_PIECEWISE:
        for (int i = 0; i < 4; ++i)
            s[i] = PiecewiseEval(sv.spectrumData + sp.offset, sv.spectrumData + sp.offset + sp.n, sp.n, lambda.lambda[i]);
        return s;
    case WF_SPEC_RGB_ALBEDO:
        for (int i = 0; i < 4; ++i) s[i] = SigmoidPoly(lambda.lambda[i], sp.c0, sp.c1, sp.c2);
        return s;
    case WF_SPEC_RGB_UNBOUNDED:
        for (int i = 0; i < 4; ++i) s[i] = sp.scale * SigmoidPoly(lambda.lambda[i], sp.c0, sp.c1, sp.c2);
        return s;
    case WF_SPEC_RGB_ILLUMINANT:
        for (int i = 0; i < 4; ++i) s[i] = sp.scale * SigmoidPoly(lambda.lambda[i], sp.c0, sp.c1, sp.c2);
        return s * DenseSample(sv, sp.offset, lambda);
    case WF_SPEC_BLACKBODY:
        for (int i = 0; i < 4; ++i) s[i] = Blackbody(lambda.lambda[i], sp.c0) * sp.c1;
        return s;
    default: return S4c(0.f);
    }
}
// Spectrum::operator()(lambda) for a single wavelength (materials.h:162 eta(lambda[0]))
WF_HD float SpectrumEval(const SceneView &sv, int id, float lambda) {
    const wf_spectrum sp = sv.spectra[id];
    switch (sp.type) {
    case WF_SPEC_CONSTANT: return sp.c0;
    case WF_SPEC_DENSE: {
        // DenselySampledSpectrum::operator(), util/spectrum.h:432-438
        int o = (int)lround(lambda) - WF_LAMBDA_MIN;
        return (o < 0 || o >= WF_NDENSE) ? 0.f : sv.spectrumData[sp.offset + o];
    }
    case WF_SPEC_PIECEWISE:
        return PiecewiseEval(sv.spectrumData + sp.offset, sv.spectrumData + sp.offset + sp.n, sp.n, lambda);
    case WF_SPEC_RGB_ALBEDO: return SigmoidPoly(lambda, sp.c0, sp.c1, sp.c2);
    case WF_SPEC_RGB_UNBOUNDED: return sp.scale * SigmoidPoly(lambda, sp.c0, sp.c1, sp.c2);
    case WF_SPEC_RGB_ILLUMINANT: {
        int o = (int)lround(lambda) - WF_LAMBDA_MIN;
        float ill = (o < 0 || o >= WF_NDENSE) ? 0.f : sv.spectrumData[sp.offset + o];
        return sp.scale * SigmoidPoly(lambda, sp.c0, sp.c1, sp.c2) * ill;
    }
    case WF_SPEC_BLACKBODY: return Blackbody(lambda, sp.c0) * sp.c1;
    default: return 0.f;
    }
}
WF_HD bool SpectrumIsConstant(const SceneView &sv, int id) { return sv.spectra[id].type == WF_SPEC_CONSTANT; }

// ---------------------------------------------------------------------------------------------
// LEAN DEVICE VARIANTS (round 5).  On gfx950 a kernel is allocated the registers of the hungriest function it can REACH — also of an
// out-of-line callee behind a branch the scene never takes.  Measured (tools/exp/light_vgprs.hip and its siblings): the interaction of a
// triangle hit incl. the instance transform needs 84 VGPRs, HitInteraction with the quadric / patch / curve callees reachable 235; the
// inline texture roots (constant / image / bilerp) 88, with the texture-graph evaluator reachable 225; the whole light sample 93, with
// the sampler of non-triangle emitters reachable 214.  The material kernels of rounds 1-4 therefore ran at 2 waves per SIMD on every
// scene.  A translation unit compiled with -DWF_LEAN (wf_mat.hip's lean variants: the back end launches them only on scenes without
// quadrics / patches / curves whose textures are all constants, image maps or bilerps — SceneLean()) compiles those callees OUT of the
// DEVICE code; a texture graph met by such a kernel raises WF_FATAL_LEAN_VARIANT instead of being mis-evaluated.  Host code is the same
// in every unit (no ODR difference on the host side; device code objects are per unit).
#if defined(__HIP_DEVICE_COMPILE__) && defined(WF_LEAN)
constexpr bool WF_DEV_LEAN = true;
#else
constexpr bool WF_DEV_LEAN = false;
#endif
// the image record of a lookup, read IN PLACE (round 5): a local copy (`const wf_tex_image im = *imp;`, rounds 1-4) lives in scratch — its
// level arrays are indexed dynamically — : 128 B written and 100-280 scratch loads per lookup in the MIP filter callees.
// -DWF_TEX_IMAGE_COPY=1 restores the copy (A/B builds).
#if defined(WF_TEX_IMAGE_COPY) && WF_TEX_IMAGE_COPY
#define WF_TEX_IMAGE_LOCAL(im, imp) const wf_tex_image im = *(imp)
#else
#define WF_TEX_IMAGE_LOCAL(im, imp) const wf_tex_image &im = *(imp)
#endif

// ---------------------------------------------------------------------------------------------
// Texture evaluation over the flattened texture nodes (textures.h:1092-1155): constant, scale, mix, 2D checkerboard.
// The nesting depth is bounded by WF_TEX_STACK frames of the explicit-stack evaluator (enforced by the host builder).  Image textures arrive with the
// image-texture row of SURVEY.md §8(f).
// TextureEvalContext (textures.h:33-61)
struct TexCtx {
    V3 p{0, 0, 0}, dpdx{0, 0, 0}, dpdy{0, 0, 0};
    N3 n{0, 0, 0};
    V2 uv{0, 0};
    float dudx = 0, dudy = 0, dvdx = 0, dvdy = 0;
};
// TextureMapping2D::Map: UVMapping, SphericalMapping, CylindricalMapping, PlanarMapping (textures.h:76-225)
struct TexCoord2 { V2 st; float dsdx, dsdy, dtdx, dtdy; };
// the non-uv mappings, out of line (pointer arguments only): inlined at every leaf of the texture-graph template they
// doubled the material kernels' compile time
WF_NI void TexMap2DP(const wf_transform *xf, const wf_texture *tp, const TexCtx *cp, TexCoord2 *out) {
    const wf_texture &t = *tp;
    const TexCtx &c = *cp;
    TexCoord2 r;
    const float(*m)[4] = xf->mInv;  // textureFromRender
    auto xfP = [&](V3 p) {
        float xp = m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3];
        float yp = m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3];
        float zp = m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3];
        float wp = m[3][0] * p.x + m[3][1] * p.y + m[3][2] * p.z + m[3][3];
        return wp == 1 ? V3{xp, yp, zp} : V3{xp, yp, zp} / wp;
    };
    auto xfV = [&](V3 v) {
        return V3{m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
                  m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z};
    };
    V3 pt = xfP(c.p);
    V3 dpdx = xfV(c.dpdx), dpdy = xfV(c.dpdy);
    if (t.mapping == WF_TEXMAP_PLANAR) {
        V3 vs{t.map[4], t.map[5], t.map[6]}, vt{t.map[7], t.map[8], t.map[9]};
        r.dsdx = Dot(vs, dpdx); r.dsdy = Dot(vs, dpdy);
        r.dtdx = Dot(vt, dpdx); r.dtdy = Dot(vt, dpdy);
        r.st = V2{t.map[2] + Dot(pt, vs), t.map[3] + Dot(pt, vt)};
        *out = r;
        return;
    }
    float x2y2 = Sqr(pt.x) + Sqr(pt.y);
    V3 dsdp = V3{-pt.y, pt.x, 0} / (2 * Pi * x2y2), dtdp;
    if (t.mapping == WF_TEXMAP_SPHERICAL) {
        float sqrtx2y2 = sqrt(x2y2);
        dtdp = 1 / (Pi * (x2y2 + Sqr(pt.z))) * V3{pt.x * pt.z / sqrtx2y2, pt.y * pt.z / sqrtx2y2, -sqrtx2y2};
        V3 vec = Normalize(pt);
        float phi = atan2(vec.y, vec.x);
        r.st = V2{SafeACos(vec.z) * InvPi, (phi < 0 ? phi + 2 * Pi : phi) * Inv2Pi};
    } else {
        dtdp = V3{0, 0, 1};
        r.st = V2{(Pi + atan2(pt.y, pt.x)) * Inv2Pi, pt.z};
    }
    r.dsdx = Dot(dsdp, dpdx); r.dsdy = Dot(dsdp, dpdy);
    r.dtdx = Dot(dtdp, dpdx); r.dtdy = Dot(dtdp, dpdy);
    *out = r;
}
WF_HD TexCoord2 TexMap2D(const SceneView &sv, const wf_texture &t, const TexCtx &c) {
    TexCoord2 r;
    if (t.mapping == WF_TEXMAP_UV) {
        const float su = t.map[0], sv_ = t.map[1], du = t.map[2], dv = t.map[3];
        r.dsdx = su * c.dudx; r.dsdy = su * c.dudy;
        r.dtdx = sv_ * c.dvdx; r.dtdy = sv_ * c.dvdy;
        r.st = V2{su * c.uv.x + du, sv_ * c.uv.y + dv};
        return r;
    }
    const wf_texture tt = t;
    const TexCtx cc = c;
    TexMap2DP(sv.lightXforms + t.xform, &tt, &cc, &r);
    return r;
}
// Checkerboard() for a 2D mapping (textures.cpp:183-207)
WF_HD float Checkerboard2D(const SceneView &sv, const wf_texture &t, const TexCtx &c) {
    TexCoord2 tcd = TexMap2D(sv, t, c);
    float dsdx = tcd.dsdx, dsdy = tcd.dsdy, dtdx = tcd.dtdx, dtdy = tcd.dtdy;
    float s = tcd.st.x, tt = tcd.st.y;
    auto d = [](float x) {
        float y = x / 2 - floor(x / 2) - 0.5f;
        return x / 2 + y * (1 - 2 * abs(y));
    };
    auto bf = [&](float x, float r) -> float {
        if (floor(x - r) == floor(x + r)) return (float)(1 - 2 * ((int)floor(x) & 1));
        return (d(x + r) - 2 * d(x) + d(x - r)) / Sqr(r);
    };
    float ds = fmax(abs(dsdx), abs(dsdy));
    float dt = fmax(abs(dtdx), abs(dtdy));
    ds *= 1.5f;
    dt *= 1.5f;
    return 0.5f - bf(s, ds) * bf(tt, dt) / 2;
}
// ---------------------------------------------------------------------------------------------
// Image textures: MIPMap::Filter for the point / bilinear / trilinear filters (util/mipmap.cpp:228-262), Image::
// BilerpChannel / GetChannel with wrap modes (util/image.h:96-147,265-292), RGB -> spectrum (util/spectrum.cpp:218-246)
// RGBToSpectrumTable::operator() (util/color.cpp:31-68) on the device copy of the table
WF_NI void RGBToSpectrumCoeffsP(const float *coeffs, const float *zNodes, float r, float g, float b, float *c0, float *c1, float *c2) {
    constexpr int res = 64;
    const float rgb[3] = {r, g, b};
    float c[3];
    if (rgb[0] == rgb[1] && rgb[1] == rgb[2]) {
        *c0 = 0; *c1 = 0;
        *c2 = (rgb[0] - .5f) / sqrt(rgb[0] * (1 - rgb[0]));
        return;
    }
    int maxc = (rgb[0] > rgb[1]) ? ((rgb[0] > rgb[2]) ? 0 : 2) : ((rgb[1] > rgb[2]) ? 1 : 2);
    float z = maxc == 0 ? rgb[0] : (maxc == 1 ? rgb[1] : rgb[2]);
    float cx = maxc == 0 ? rgb[1] : (maxc == 1 ? rgb[2] : rgb[0]);  // rgb[(maxc + 1) % 3]
    float cy = maxc == 0 ? rgb[2] : (maxc == 1 ? rgb[0] : rgb[1]);  // rgb[(maxc + 2) % 3]
    float x = cx * (res - 1) / z;
    float y = cy * (res - 1) / z;
    int xi = (int)x < res - 2 ? (int)x : res - 2, yi = (int)y < res - 2 ? (int)y : res - 2;
    int zi = FindInterval(res, [&](int i) { return zNodes[i] < z; });
    float dx = x - xi, dy = y - yi, dz = (z - zNodes[zi]) / (zNodes[zi + 1] - zNodes[zi]);
    for (int i = 0; i < 3; ++i) {
        auto co = [&](int ddx, int ddy, int ddz) {
            return coeffs[((((size_t)maxc * res + (zi + ddz)) * res + (yi + ddy)) * res + (xi + ddx)) * 3 + i];
        };
        c[i] = Lerp(dz, Lerp(dy, Lerp(dx, co(0, 0, 0), co(1, 0, 0)), Lerp(dx, co(0, 1, 0), co(1, 1, 0))),
                    Lerp(dy, Lerp(dx, co(0, 0, 1), co(1, 0, 1)), Lerp(dx, co(0, 1, 1), co(1, 1, 1))));
    }
    *c0 = c[0]; *c1 = c[1]; *c2 = c[2];
}
WF_HD void RGBToSpectrumCoeffs(const SceneView &sv, const float rgb[3], float c[3]) {
    RGBToSpectrumCoeffsP(sv.rgb2specCoeffs, sv.rgb2specZNodes, rgb[0], rgb[1], rgb[2], &c[0], &c[1], &c[2]);
}
WF_HD int ModI(int a, int b) { int r = a - (a / b) * b; return r < 0 ? r + b : r; }  // util/math.h:251-254
// IEEE half bit pattern -> float (exact; util/float.h Half::operator float)
WF_HD float HalfBitsToFloat(uint16_t h) {
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16, exp = (h >> 10) & 0x1fu, man = h & 0x3ffu;
    if (exp == 0) {
        const float v = (float)man * 5.9604644775390625e-08f;   // zero / subnormal half: man * 2^-24, exact in float
        return BitsToFloat(FloatToBits(v) | sign);
    }
    if (exp == 31) return BitsToFloat(sign | 0x7f800000u | (man << 13));
    return BitsToFloat(sign | ((exp + 112) << 23) | (man << 13));
}
WF_HD float ImageTexel(const float *table, const wf_tex_image &im, int level, int x, int y, int c) {
    const int rx = im.res[0] >> level > 0 ? im.res[0] >> level : 1, ry = im.res[1] >> level > 0 ? im.res[1] >> level : 1;
    // RemapPixelCoords (util/image.h:96-147)
    if (im.wrap == WF_WRAP_OCTAHEDRAL) {
        if (x < 0) { x = -x; y = ry - 1 - y; }
        else if (x >= rx) { x = 2 * rx - 1 - x; y = ry - 1 - y; }
        if (y < 0) { x = rx - 1 - x; y = -y; }
        else if (y >= ry) { x = rx - 1 - x; y = 2 * ry - 1 - y; }
        if (rx == 1) x = 0;
        if (ry == 1) y = 0;
        // Coordinates more than one image away (an octahedral map under a scaled (u, v) mapping) are still outside after the reflection:
        // the reference then reads past its pixel array (util/image.h:100-121 has no second step) — undefined there, clamped here so that
        // no kernel reads outside the table
        x = Clamp(x, 0, rx - 1);
        y = Clamp(y, 0, ry - 1);
    } else {
        if (!(x >= 0 && x < rx)) {
            if (im.wrap == WF_WRAP_REPEAT) x = ModI(x, rx);
            else if (im.wrap == WF_WRAP_CLAMP) x = Clamp(x, 0, rx - 1);
            else return 0.f;
        }
        if (!(y >= 0 && y < ry)) {
            if (im.wrap == WF_WRAP_REPEAT) y = ModI(y, ry);
            else if (im.wrap == WF_WRAP_CLAMP) y = Clamp(y, 0, ry - 1);
            else return 0.f;
        }
    }
    const size_t i = ((size_t)y * rx + x) * im.n_channels + c;
    const WF_GLOBAL_AS float *tg = (const WF_GLOBAL_AS float *)table;   // (the texel table is global memory: a global load, not a flat one — GPtr above)
    if (im.format == WF_TEXEL_FLOAT) return tg[im.level_offset[level] + i];
    // texels kept in the source image's format, as the reference's MIP levels are (Image::GetChannel, util/image.h:204-221): a quarter / half of
    // the bytes per gather, decoded through the encoding's 256-entry table / by widening the half
    if (im.format == WF_TEXEL_U8) return tg[im.lut_offset + reinterpret_cast<const WF_GLOBAL_AS uint8_t *>(tg + im.level_offset[level])[i]];
    return HalfBitsToFloat(reinterpret_cast<const WF_GLOBAL_AS uint16_t *>(tg + im.level_offset[level])[i]);
}
WF_HD float ImageBilerpChannel(const float *table, const wf_tex_image &im, int level, V2 p, int c) {
    const int rx = im.res[0] >> level > 0 ? im.res[0] >> level : 1, ry = im.res[1] >> level > 0 ? im.res[1] >> level : 1;
    float x = p.x * rx - 0.5f, y = p.y * ry - 0.5f;
    int xi = (int)floor(x), yi = (int)floor(y);
    float dx = x - xi, dy = y - yi;
    float v0 = ImageTexel(table, im, level, xi, yi, c), v1 = ImageTexel(table, im, level, xi + 1, yi, c);
    float v2 = ImageTexel(table, im, level, xi, yi + 1, c), v3 = ImageTexel(table, im, level, xi + 1, yi + 1, c);
    return ((1 - dx) * (1 - dy) * v0 + dx * (1 - dy) * v1 + (1 - dx) * dy * v2 + dx * dy * v3);
}
struct RGB3 { float r, g, b; };
// MIPMap::Texel<RGB> / Bilerp<RGB> (util/mipmap.cpp:214-226,286-298), Texel<Float> / Bilerp<Float> (:208-212,395-409)
WF_HD RGB3 MIPTexelRGB(const float *table, const wf_tex_image &im, int level, int x, int y) {
    if (im.n_channels >= 3) return RGB3{ImageTexel(table, im, level, x, y, 0), ImageTexel(table, im, level, x, y, 1), ImageTexel(table, im, level, x, y, 2)};
    float v = ImageTexel(table, im, level, x, y, 0);
    return RGB3{v, v, v};
}
WF_HD RGB3 MIPBilerpRGB(const float *table, const wf_tex_image &im, int level, V2 st) {
    if (im.n_channels >= 3) return RGB3{ImageBilerpChannel(table, im, level, st, 0), ImageBilerpChannel(table, im, level, st, 1), ImageBilerpChannel(table, im, level, st, 2)};
    float v = ImageBilerpChannel(table, im, level, st, 0);
    return RGB3{v, v, v};
}
WF_HD float MIPBilerpFloat(const float *table, const wf_tex_image &im, int level, V2 st) {
    // one inlined lookup serves Y (channel 0) and R G B A (the alpha channel, 3): this function is instantiated three times in the
    // out-of-line filter the traversal kernels' alpha test calls, and that callee's size and registers are paid at every call
    if (im.n_channels != 3) return ImageBilerpChannel(table, im, level, st, im.n_channels == 4 ? 3 : 0);
    float sum = 0;
    for (int c = 0; c < 3; ++c) sum += ImageBilerpChannel(table, im, level, st, c);
    return sum / 3;
}
// the texel triple NormalMap() reads (materials.h:86-95): Image::BilerpChannel of channels 0..2 at (u, 1-v), repeat wrap
WF_NI void NormalMapTexelP(const float *table, const wf_tex_image *imp, float u, float v, float *x, float *y, float *z) {
    WF_TEX_IMAGE_LOCAL(im, imp);
    const V2 uv{u, 1 - v};
    *x = 2 * ImageBilerpChannel(table, im, 0, uv, 0) - 1;
    *y = 2 * ImageBilerpChannel(table, im, 0, uv, 1) - 1;
    *z = 2 * ImageBilerpChannel(table, im, 0, uv, 2) - 1;
}
// the level choice of the non-EWA filters; returns false when the filter is wider than the image (top level texel)
WF_HD bool MIPLevel(const wf_tex_image &im, float dsdx, float dtdx, float dsdy, float dtdy, float *level, int *iLevel) {
    float width = 2 * fmax(fmax(abs(dsdx), abs(dtdx)), fmax(abs(dsdy), abs(dtdy)));
    const int nLevels = im.n_levels;
    *level = nLevels - 1 + log(fmax(width, 1e-8f)) * 1.442695040888963387004650940071f;
    if (*level >= nLevels - 1) return false;
    int il = (int)floor(*level);
    *iLevel = il > 0 ? il : 0;
    return true;
}
WF_HD float LengthSquared2(V2 v) { return Sqr(v.x) + Sqr(v.y); }
// MIPMap::EWA (util/mipmap.cpp:300-349): the texels inside the unit ellipse of the two footprint axes, Gaussian weights
// from the table.  FLOAT: Texel<Float> (channel 0 only, returned in .r)
template <bool FLOAT>
WF_HD RGB3 MIPEWA(const float *table, const wf_tex_image &im, int level, V2 st, V2 dst0, V2 dst1) {
    if (level >= im.n_levels) return FLOAT ? RGB3{ImageTexel(table, im, im.n_levels - 1, 0, 0, 0), 0, 0} : MIPTexelRGB(table, im, im.n_levels - 1, 0, 0);
    const int rx = im.res[0] >> level > 0 ? im.res[0] >> level : 1, ry = im.res[1] >> level > 0 ? im.res[1] >> level : 1;
    st.x = st.x * rx - 0.5f;
    st.y = st.y * ry - 0.5f;
    dst0.x *= rx; dst0.y *= ry;
    dst1.x *= rx; dst1.y *= ry;
    float A = Sqr(dst0.y) + Sqr(dst1.y) + 1;
    float B = -2 * (dst0.x * dst0.y + dst1.x * dst1.y);
    float C = Sqr(dst0.x) + Sqr(dst1.x) + 1;
    float invF = 1 / (A * C - Sqr(B) * 0.25f);
    A *= invF; B *= invF; C *= invF;
    float det = -Sqr(B) + 4 * A * C;
    float invDet = 1 / det;
    float uSqrt = SafeSqrt(det * C), vSqrt = SafeSqrt(A * det);
    int s0 = (int)ceil(st.x - 2 * invDet * uSqrt), s1 = (int)floor(st.x + 2 * invDet * uSqrt);
    int t0 = (int)ceil(st.y - 2 * invDet * vSqrt), t1 = (int)floor(st.y + 2 * invDet * vSqrt);
    RGB3 sum{0, 0, 0};
    float sumWts = 0;
    const float *lut = table + im.ewa_lut_offset;
    for (int it = t0; it <= t1; ++it) {
        float tt = it - st.y;
        for (int is = s0; is <= s1; ++is) {
            float ss = is - st.x;
            float r2 = A * Sqr(ss) + B * ss * tt + C * Sqr(tt);
            if (r2 < 1) {
                int index = (int)(r2 * 128);
                if (index > 127) index = 127;
                float weight = lut[index];
                if (FLOAT) sum.r += weight * ImageTexel(table, im, level, is, it, 0);
                else {
                    RGB3 tx = MIPTexelRGB(table, im, level, is, it);
                    sum.r += weight * tx.r; sum.g += weight * tx.g; sum.b += weight * tx.b;
                }
                sumWts += weight;
            }
        }
    }
    return RGB3{sum.r / sumWts, sum.g / sumWts, sum.b / sumWts};
}
// the EWA branch of MIPMap::Filter (util/mipmap.cpp:264-283)
template <bool FLOAT>
WF_HD RGB3 MIPFilterEWA(const float *table, const wf_tex_image &im, V2 st, V2 dst0, V2 dst1) {
    if (LengthSquared2(dst0) < LengthSquared2(dst1)) { V2 t = dst0; dst0 = dst1; dst1 = t; }
    float longerVecLength = sqrt(LengthSquared2(dst0)), shorterVecLength = sqrt(LengthSquared2(dst1));
    if (shorterVecLength * im.max_anisotropy < longerVecLength && shorterVecLength > 0) {
        float scale = longerVecLength / (shorterVecLength * im.max_anisotropy);
        dst1.x *= scale; dst1.y *= scale;
        shorterVecLength *= scale;
    }
    if (shorterVecLength == 0) return FLOAT ? RGB3{MIPBilerpFloat(table, im, 0, st), 0, 0} : MIPBilerpRGB(table, im, 0, st);
    float lod = fmax(0.f, im.n_levels - 1 + Log2f(shorterVecLength));
    int ilod = (int)floor(lod);
    RGB3 a = MIPEWA<FLOAT>(table, im, ilod, st, dst0, dst1), b = MIPEWA<FLOAT>(table, im, ilod + 1, st, dst0, dst1);
    float t = lod - ilod;
    return RGB3{Lerp(t, a.r, b.r), FLOAT ? 0.f : Lerp(t, a.g, b.g), FLOAT ? 0.f : Lerp(t, a.b, b.b)};
}
// EWA lookups out of line on their own: the point / bilinear / trilinear functions below are called from the traversal kernels' alpha
// test, and a callee's register use is paid at every call (callee-saved registers) whether the branch is taken or not
WF_NI void MIPFilterEWARGBP(const float *table, const wf_tex_image *imp, float s_, float t_, float dsdx, float dtdx, float dsdy, float dtdy, float *r, float *g, float *b) {
    WF_TEX_IMAGE_LOCAL(im, imp);
    RGB3 o = MIPFilterEWA<false>(table, im, V2{s_, t_}, V2{dsdx, dtdx}, V2{dsdy, dtdy});
    *r = o.r; *g = o.g; *b = o.b;
}
WF_NI float MIPFilterEWAFloatP(const float *table, const wf_tex_image *imp, float s_, float t_, float dsdx, float dtdx, float dsdy, float dtdy) {
    WF_TEX_IMAGE_LOCAL(im, imp);
    return MIPFilterEWA<true>(table, im, V2{s_, t_}, V2{dsdx, dtdx}, V2{dsdy, dtdy}).r;
}
WF_NI void MIPFilterRGBP(const float *table, const wf_tex_image *imp, float s_, float t_, float dsdx, float dtdx, float dsdy, float dtdy, float *r, float *g, float *b) {
    if (imp->filter == WF_MIP_EWA) { MIPFilterEWARGBP(table, imp, s_, t_, dsdx, dtdx, dsdy, dtdy, r, g, b); return; }
    WF_TEX_IMAGE_LOCAL(im, imp);
    const V2 st{s_, t_};
    RGB3 o = [&]() -> RGB3 {
    float level;
    int iLevel;
    if (!MIPLevel(im, dsdx, dtdx, dsdy, dtdy, &level, &iLevel)) return MIPTexelRGB(table, im, im.n_levels - 1, 0, 0);
    if (im.filter == WF_MIP_POINT) {
        const int rx = im.res[0] >> iLevel > 0 ? im.res[0] >> iLevel : 1, ry = im.res[1] >> iLevel > 0 ? im.res[1] >> iLevel : 1;
        return MIPTexelRGB(table, im, iLevel, (int)roundf(st.x * rx - 0.5f), (int)roundf(st.y * ry - 0.5f));
    }
    if (im.filter == WF_MIP_BILINEAR || iLevel == 0) return MIPBilerpRGB(table, im, iLevel, st);
    RGB3 a = MIPBilerpRGB(table, im, iLevel, st), bb = MIPBilerpRGB(table, im, iLevel + 1, st);
    float t = level - iLevel;
    return RGB3{Lerp(t, a.r, bb.r), Lerp(t, a.g, bb.g), Lerp(t, a.b, bb.b)};
    }();
    *r = o.r; *g = o.g; *b = o.b;
}
WF_HD RGB3 MIPFilterRGB(const SceneView &sv, int image, V2 st, float dsdx, float dtdx, float dsdy, float dtdy) {
    RGB3 o;
    MIPFilterRGBP(sv.tableData, sv.texImages + image, st.x, st.y, dsdx, dtdx, dsdy, dtdy, &o.r, &o.g, &o.b);
    return o;
}
WF_NI float MIPFilterFloatP(const float *table, const wf_tex_image *imp, float s_, float t_, float dsdx, float dtdx, float dsdy, float dtdy) {
    WF_TEX_IMAGE_LOCAL(im, imp);
    const V2 st{s_, t_};
    float level;
    int iLevel;
    if (!MIPLevel(im, dsdx, dtdx, dsdy, dtdy, &level, &iLevel)) return ImageTexel(table, im, im.n_levels - 1, 0, 0, 0);
    if (im.filter == WF_MIP_POINT) {
        const int rx = im.res[0] >> iLevel > 0 ? im.res[0] >> iLevel : 1, ry = im.res[1] >> iLevel > 0 ? im.res[1] >> iLevel : 1;
        return ImageTexel(table, im, iLevel, (int)roundf(st.x * rx - 0.5f), (int)roundf(st.y * ry - 0.5f), 0);
    }
    if (im.filter == WF_MIP_BILINEAR || iLevel == 0) return MIPBilerpFloat(table, im, iLevel, st);
    return Lerp(level - iLevel, MIPBilerpFloat(table, im, iLevel, st), MIPBilerpFloat(table, im, iLevel + 1, st));
}
// EWA = false: the traversal kernels' inline alpha test (GEN = 1), which is only chosen for scenes whose alpha maps are not EWA-
// filtered — a kernel's register allocation is the maximum over everything it can call, and the EWA loop must stay out of that
// The lookup of a texture evaluated without a footprint (all differentials zero: the alpha test of the traversal kernels, whose
// TextureEvalContext comes from an interaction nobody called ComputeDifferentials on).  MIPMap::Filter's level is then
// nLevels - 1 + log2(1e-8) < 0 for any pyramid, so point filtering reads a level-0 texel and bilinear / trilinear both return
// Bilerp(0, st): the same arithmetic as MIPFilterFloatP on these inputs, in a callee a tenth of its size.
WF_NI float MIPFilterFloatZeroP(const float *table, const wf_tex_image *imp, float s_, float t_) {
    WF_TEX_IMAGE_LOCAL(im, imp);
    if (im.filter == WF_MIP_POINT) return ImageTexel(table, im, 0, (int)roundf(s_ * im.res[0] - 0.5f), (int)roundf(t_ * im.res[1] - 0.5f), 0);
    return MIPBilerpFloat(table, im, 0, V2{s_, t_});
}
// EWA = false, ZERO = true: see above
template <bool EWA = true, bool ZERO = false>
WF_HD float MIPFilterFloat(const SceneView &sv, int image, V2 st, float dsdx, float dtdx, float dsdy, float dtdy) {
    if constexpr (ZERO) return MIPFilterFloatZeroP(sv.tableData, sv.texImages + image, st.x, st.y);
    if constexpr (EWA)
        if (sv.texImages[image].filter == WF_MIP_EWA) return MIPFilterEWAFloatP(sv.tableData, sv.texImages + image, st.x, st.y, dsdx, dtdx, dsdy, dtdy);
    return MIPFilterFloatP(sv.tableData, sv.texImages + image, st.x, st.y, dsdx, dtdx, dsdy, dtdy);
}
// FloatImageTexture::Evaluate (textures.h:579-591), SpectrumImageTexture::Evaluate (textures.cpp:300-328)
template <bool EWA = true, bool ZERO = false>
WF_HD float EvalFloatImageTexture(const SceneView &sv, const wf_texture &t, const TexCtx &c) {
    TexCoord2 tcd = TexMap2D(sv, t, c);
    float dsdx = tcd.dsdx, dsdy = tcd.dsdy, dtdx = tcd.dtdx, dtdy = tcd.dtdy;
    V2 st = tcd.st;
    st.y = 1 - st.y;
    float v = t.f0 * MIPFilterFloat<EWA, ZERO>(sv, t.i0, st, dsdx, dtdx, dsdy, dtdy);
    return t.f1 != 0 ? fmax(0.f, 1 - v) : v;
}
WF_HD S4 EvalSpectrumImageTexture(const SceneView &sv, const wf_texture &t, const Wavelengths &lambda, const TexCtx &c) {
    TexCoord2 tcd = TexMap2D(sv, t, c);
    float dsdx = tcd.dsdx, dsdy = tcd.dsdy, dtdx = tcd.dtdx, dtdy = tcd.dtdy;
    V2 st = tcd.st;
    st.y = 1 - st.y;
    RGB3 f = MIPFilterRGB(sv, t.i0, st, dsdx, dtdx, dsdy, dtdy);
    float rgb[3] = {t.f0 * f.r, t.f0 * f.g, t.f0 * f.b};
    if (t.f1 != 0) { rgb[0] = 1 - rgb[0]; rgb[1] = 1 - rgb[1]; rgb[2] = 1 - rgb[2]; }
    for (int k = 0; k < 3; ++k) rgb[k] = fmax(0.f, rgb[k]);
    float cf[3];
    S4 s;
    if (t.spectrum == 0) {
        // RGBAlbedoSpectrum(cs, Clamp(rgb, 0, 1))
        float in[3] = {Clamp(rgb[0], 0.f, 1.f), Clamp(rgb[1], 0.f, 1.f), Clamp(rgb[2], 0.f, 1.f)};
        RGBToSpectrumCoeffs(sv, in, cf);
        for (int i = 0; i < 4; ++i) s[i] = SigmoidPoly(lambda.lambda[i], cf[0], cf[1], cf[2]);
        return s;
    }
    // RGBUnboundedSpectrum / RGBIlluminantSpectrum: scale = 2 max(rgb), rsp = coeffs(rgb / scale)
    float m = fmax(fmax(rgb[0], rgb[1]), rgb[2]);
    float scale = 2 * m;
    float in[3] = {0, 0, 0};
    if (scale) { in[0] = rgb[0] / scale; in[1] = rgb[1] / scale; in[2] = rgb[2] / scale; }
    RGBToSpectrumCoeffs(sv, in, cf);
    for (int i = 0; i < 4; ++i) s[i] = scale * SigmoidPoly(lambda.lambda[i], cf[0], cf[1], cf[2]);
    if (t.spectrum == 2) return s * DenseSample(sv, sv.csIlluminantOffset, lambda);
    return s;
}

// ---------------------------------------------------------------------------------------------
// The procedural textures (the reference evaluates them through its UniversalTextureEvaluator): fbm, wrinkled, windy
// (textures.h:480-502,1079-1122), marble (textures.cpp:480-503), dots (textures.h:427-478 + InsidePolkaDot, textures.cpp:288-303),
// 3D checkerboard (textures.cpp:208-216).  Out of line: leaves of the texture graph that production scenes rarely have.
// TexCoord3D of PointTransformMapping::Map (textures.h:238-241): textureFromRender applied to p, dpdx, dpdy
struct TexCoord3 { V3 p, dpdx, dpdy; };
WF_HD TexCoord3 TexMap3D(const wf_transform *xf, const TexCtx &c) {
    const float(*m)[4] = xf->mInv;
    auto xfP = [&](V3 p) {
        float xp = m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3];
        float yp = m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3];
        float zp = m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3];
        float wp = m[3][0] * p.x + m[3][1] * p.y + m[3][2] * p.z + m[3][3];
        return wp == 1 ? V3{xp, yp, zp} : V3{xp, yp, zp} / wp;
    };
    auto xfV = [&](V3 v) {
        return V3{m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z, m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z};
    };
    return TexCoord3{xfP(c.p), xfV(c.dpdx), xfV(c.dpdy)};
}
WF_NI float NoiseFloatTextureP(const int32_t *perm, const wf_transform *xf, const wf_texture *tp, const TexCtx *cp) {
    const wf_texture &t = *tp;
    TexCoord3 c = TexMap3D(xf, *cp);
    if (t.type == WF_TEX_FLOAT_FBM) return FBm(perm, c.p, c.dpdx, c.dpdy, t.f0, t.i0);
    if (t.type == WF_TEX_FLOAT_WRINKLED) return Turbulence(perm, c.p, c.dpdx, c.dpdy, t.f0, t.i0);
    // WindyTexture::Evaluate
    float windStrength = FBm(perm, .1f * c.p, .1f * c.dpdx, .1f * c.dpdy, .5f, 3);
    float waveHeight = FBm(perm, c.p, c.dpdx, c.dpdy, .5f, 6);
    return abs(windStrength) * waveHeight;
}
WF_NI void MarbleTextureP(const int32_t *perm, const wf_transform *xf, const wf_texture *tp, const TexCtx *cp, const float *coeffs, const float *zNodes,
                          const float *lambda, float *out) {
    const wf_texture &t = *tp;
    TexCoord3 c = TexMap3D(xf, *cp);
    const float scale = t.f1, variation = t.map[10];
    c.p = c.p * scale;
    float marble = c.p.y + variation * FBm(perm, c.p, scale * c.dpdx, scale * c.dpdy, t.f0, t.i0);
    float tt = .5f + .5f * sin(marble);
    const float colors[9][3] = {{.58f, .58f, .6f}, {.58f, .58f, .6f}, {.58f, .58f, .6f}, {.5f, .5f, .5f}, {.6f, .59f, .58f},
                                {.58f, .58f, .6f}, {.58f, .58f, .6f}, {.2f, .2f, .33f}, {.58f, .58f, .6f}};
    const int nSeg = 9 - 3;
    int first = (int)floor(tt * nSeg);
    if (first > nSeg - 1) first = nSeg - 1;
    tt = tt * nSeg - first;
    float rgb[3];
    for (int k = 0; k < 3; ++k) {
        // EvaluateCubicBezier = BlossomCubicBezier(cp, u, u, u) (util/splines.h:17-28)
        float a0 = Lerp(tt, colors[first][k], colors[first + 1][k]), a1 = Lerp(tt, colors[first + 1][k], colors[first + 2][k]), a2 = Lerp(tt, colors[first + 2][k], colors[first + 3][k]);
        float b0 = Lerp(tt, a0, a1), b1 = Lerp(tt, a1, a2);
        rgb[k] = 1.5f * Lerp(tt, b0, b1);
    }
    float c0, c1, c2;
    RGBToSpectrumCoeffsP(coeffs, zNodes, rgb[0], rgb[1], rgb[2], &c0, &c1, &c2);   // RGBAlbedoSpectrum(*RGBColorSpace::sRGB, rgb)
    for (int i = 0; i < 4; ++i) out[i] = SigmoidPoly(lambda[i], c0, c1, c2);
}
WF_NI bool InsidePolkaDotP(const int32_t *perm, float s, float t) {
    int sCell = (int)floor(s + .5f), tCell = (int)floor(t + .5f);
    if (Noise(perm, sCell + .5f, tCell + .5f) > 0) {
        float radius = .35f;
        float maxShift = 0.5f - radius;
        float sCenter = sCell + maxShift * Noise(perm, sCell + 1.5f, tCell + 2.8f);
        float tCenter = tCell + maxShift * Noise(perm, sCell + 4.5f, tCell + 9.8f);
        float ds = s - sCenter, dt = t - tCenter;
        if (Sqr(ds) + Sqr(dt) < Sqr(radius)) return true;
    }
    return false;
}
// Checkerboard() for the 3D mapping (textures.cpp:208-216)
WF_NI float Checkerboard3DP(const wf_transform *xf, const TexCtx *cp) {
    TexCoord3 c = TexMap3D(xf, *cp);
    auto d = [](float x) {
        float y = x / 2 - floor(x / 2) - 0.5f;
        return x / 2 + y * (1 - 2 * abs(y));
    };
    auto bf = [&](float x, float r) -> float {
        if (floor(x - r) == floor(x + r)) return (float)(1 - 2 * ((int)floor(x) & 1));
        return (d(x + r) - 2 * d(x) + d(x - r)) / Sqr(r);
    };
    float dx = 1.5f * fmax(abs(c.dpdx.x), abs(c.dpdy.x));
    float dy = 1.5f * fmax(abs(c.dpdx.y), abs(c.dpdy.y));
    float dz = 1.5f * fmax(abs(c.dpdx.z), abs(c.dpdy.z));
    return 0.5f - 0.5f * bf(c.p.x, dx) * bf(c.p.y, dy) * bf(c.p.z, dz);
}
WF_HD float CheckerboardWeight(const SceneView &sv, const wf_texture &t, const TexCtx &c) {
    if (t.mapping == WF_TEXMAP_POINT3D) { TexCtx cc = c; return Checkerboard3DP(sv.lightXforms + t.xform, &cc); }
    return Checkerboard2D(sv, t, c);
}
WF_HD bool InsidePolkaDot(const SceneView &sv, const wf_texture &t, const TexCtx &c) {
    TexCoord2 st = TexMap2D(sv, t, c);
    return InsidePolkaDotP(sv.noisePerm, st.st.x, st.st.y);
}

// FloatTexture::Evaluate / SpectrumTexture::Evaluate over the flattened texture graph (textures.h:1140-1155: the universal
// evaluator dispatches to the texture, which evaluates its children the same way — any nesting).  The reference recurses through
// tagged pointers; the device code has no recursion.  Until round 3 the walk was a template over the remaining depth, fully
// inlined (three interior node types: the code grew ~7x per level and the bound was two interior levels; deeper graphs were
// refused).  Since round 4 it is an EXPLICIT-STACK evaluator inside the out-of-line graph functions below: a frame per interior
// node (node id | resume state, the weight, the first operand), WF_TEX_STACK frames, the same operations in the same order.
// (As part of the inlined material code an explicit stack had cost 470 spilled VGPRs; behind the call its arrays are the callee's
// scratch and the kernels' register budgets do not see them.)  The host builder refuses graphs nested deeper than WF_TEX_STACK.
// (16 frames.  Larger frame counts are not a free way around register pressure: with 48 frames — 2.6-2.9 KB of scratch per lane in
// the material kernels — the arealight_image golden rendered a different image on every run on the GPU while 16 frames are repeatable and
// bit-identical; the cause was not isolated, the kernels' scratch is kept small.)
#ifndef WF_TEX_STACK
#define WF_TEX_STACK 16
#endif
#ifndef WF_TEX_FRAMES_STRUCT
#define WF_TEX_FRAMES_STRUCT 1
#endif
// the three texture types a production scene's parameters usually are: what the material kernels and the traversal kernels' alpha test
// evaluate inline
template <bool EWA = true, bool ZERO = false>
WF_HD float EvalFloatTextureSimple(const SceneView &sv, const wf_texture &t, const TexCtx &tc) {
    if (t.type == WF_TEX_FLOAT_CONSTANT) return t.f0;
    if (t.type == WF_TEX_FLOAT_IMAGE) return EvalFloatImageTexture<EWA, ZERO>(sv, t, tc);
    // FloatBilerpTexture::Evaluate, textures.h:314-318
    TexCoord2 c = TexMap2D(sv, t, tc);
    const float v00 = t.f0, v01 = t.f1, v10 = t.map[10], v11 = t.map[11];
    return (1 - c.st.x) * (1 - c.st.y) * v00 + c.st.x * (1 - c.st.y) * v10 + (1 - c.st.x) * c.st.y * v01 + c.st.x * c.st.y * v11;
}
WF_HD bool IsSimpleFloatTexture(int type) { return type == WF_TEX_FLOAT_CONSTANT || type == WF_TEX_FLOAT_IMAGE || type == WF_TEX_FLOAT_BILERP; }
// leaves of a float graph: constant / image / bilerp and the noise textures
WF_HD bool IsLeafFloatTexture(int type) {
    return IsSimpleFloatTexture(type) || type == WF_TEX_FLOAT_FBM || type == WF_TEX_FLOAT_WRINKLED || type == WF_TEX_FLOAT_WINDY;
}
WF_HD float EvalFloatTextureLeaf(const SceneView &sv, const wf_texture *tp, const TexCtx &tc) {
    if (IsSimpleFloatTexture(tp->type)) return EvalFloatTextureSimple(sv, *tp, tc);
    TexCtx cc = tc;
    return NoiseFloatTextureP(sv.noisePerm, sv.lightXforms + tp->xform, tp, &cc);
}
// The explicit-stack walk of a float graph.  A frame = (node id << 3 | state, weight, first operand); the node records are read in
// place (no copies: the frames are all the scratch this costs).  States: 0 = entered; SCALE: 1 = the scale factor returned, 2 = the
// texture returned; MIX / CHECKERBOARD / DIRECTIONMIX: 1 = the amount returned (MIX only), 2 = tex0 ("tex1" of the reference's mix,
// weighted 1 - amt) returned, 3 = tex1 returned.  DOTS replaces its own frame by the chosen child (a tail call).
WF_HD float EvalFloatTextureStack(const SceneView &sv, int id, const TexCtx &tc) {
    // ONE array of mixed-type frames (WF_TEX_FRAMES_STRUCT = 1) rather than three scalar arrays: arrays of <= 128 bytes are turned into
    // VGPR vectors by AMDGPU's promote-alloca pass — 48 VGPRs at 16 frames, in kernels that sit at their register ceiling — while an
    // array of structs stays in scratch, where this cold path belongs.  Same box, spec scene, 16 spp (gpurun_out/r04j): diffuse /
    // conductor / coated diffuse 18.9 / 6.0 / 19.1 ms with the scalar arrays, 20.4 / 5.2 / 16.1 ms with the frames (44.6 -> 42.1 ms
    // with the dielectric kernel), equal ray counts.
#if WF_TEX_FRAMES_STRUCT
    struct Frame { int node; float w, a; };
    Frame fr[WF_TEX_STACK];
#define F_NODE(i) fr[i].node
#define F_W(i) fr[i].w
#define F_A(i) fr[i].a
#else
    int fnode[WF_TEX_STACK];
    float fw[WF_TEX_STACK], fa[WF_TEX_STACK];
#define F_NODE(i) fnode[i]
#define F_W(i) fw[i]
#define F_A(i) fa[i]
#endif
    int sp = 0;
    F_NODE(0) = id << 3;
    float ret = 0;
    while (sp >= 0) {
        const wf_texture *tp = sv.textures + (F_NODE(sp) >> 3);
        const int type = tp->type;
        int st = F_NODE(sp) & 7;
        if (st == 0) {
            if (IsLeafFloatTexture(type)) { ret = EvalFloatTextureLeaf(sv, tp, tc); --sp; continue; }
            if (type == WF_TEX_FLOAT_DOTS) { F_NODE(sp) = (InsidePolkaDot(sv, *tp, tc) ? tp->tex1 : tp->tex0) << 3; continue; }
            if (sp + 1 >= WF_TEX_STACK) { ret = 0; --sp; continue; }   // (refused at load: never reached)
            if (type == WF_TEX_FLOAT_SCALE) {
                // FloatScaledTexture::Evaluate, textures.h:1039-1044: the scale first
                F_NODE(sp) |= 1; F_NODE(++sp) = tp->tex1 << 3;
                continue;
            }
            if (type == WF_TEX_FLOAT_MIX) {
                // FloatMixTexture::Evaluate (textures.h:810-818): the amount first
                F_NODE(sp) |= 1; F_NODE(++sp) = tp->tex2 << 3;
                continue;
            }
            if (type == WF_TEX_FLOAT_CHECKERBOARD || type == WF_TEX_FLOAT_DIRECTIONMIX) {
                // FloatCheckerboardTexture::Evaluate (:370-378), FloatDirectionMixTexture::Evaluate (:839-847: amt * tex1 + (1 - amt) * tex2
                // = the mix form with tex0 = "tex2")
                ret = type == WF_TEX_FLOAT_DIRECTIONMIX ? AbsDot(tc.n, N3{tp->map[4], tp->map[5], tp->map[6]}) : CheckerboardWeight(sv, *tp, tc);
                st = 1;
            } else { ret = 0; --sp; continue; }
        }
        const int self = F_NODE(sp) & ~7;
        if (type == WF_TEX_FLOAT_SCALE) {
            if (st == 1) {
                if (ret == 0) { --sp; continue; }   // returns 0
                F_W(sp) = ret; F_NODE(sp) = self | 2; F_NODE(++sp) = tp->tex0 << 3;
                continue;
            }
            ret = ret * F_W(sp);
            --sp;
            continue;
        }
        // mix family
        if (st == 1) {
            F_W(sp) = ret;
            F_A(sp) = 0;
            if (ret != 1) { F_NODE(sp) = self | 2; F_NODE(++sp) = tp->tex0 << 3; continue; }
            st = 2; ret = 0;
        }
        if (st == 2) {
            F_A(sp) = ret;
            if (F_W(sp) != 0) { F_NODE(sp) = self | 3; F_NODE(++sp) = tp->tex1 << 3; continue; }
            ret = 0;
        }
        {
            const float w = F_W(sp);
            ret = (1 - w) * F_A(sp) + w * ret;
        }
        --sp;
    }
    return ret;
}
// Texture GRAPHS (scale / mix / checkerboard / directionmix over other textures) are evaluated out of line, through the
// memory-resident SceneView: inlined at every material parameter, the depth-bounded recursion was 80 % of the material
// kernels' code (683 KB -> 137 KB for the diffuse kernel, minutes -> seconds of compile time) for a path that scenes whose
// parameters are constants or image maps never take.  A root that is a constant, an image map or a bilerp stays inline.
WF_NI float EvalFloatTextureGraphP(const SceneView *svp, int id, const TexCtx *tc) { return EvalFloatTextureStack(*svp, id, *tc); }
WF_HD float EvalFloatTexture(const SceneView &sv, int id, const TexCtx &tc) {
    const int type = sv.textures[id].type;
    if (IsSimpleFloatTexture(type)) return EvalFloatTextureSimple(sv, sv.textures[id], tc);
    if constexpr (WF_DEV_LEAN) { RaiseFatal(sv, WF_FATAL_LEAN_VARIANT); return 0.f; }
    else {
    TexCtx tmp = tc;
    return EvalFloatTextureGraphP(sv.self, id, &tmp);
    }
}
WF_HD S4 EvalSpectrumTextureSimple(const SceneView &sv, const wf_texture &t, const Wavelengths &lambda, const TexCtx &tc) {
    if (t.type == WF_TEX_SPECTRUM_CONSTANT) return SpectrumSample(sv, t.spectrum, lambda);
    if (t.type == WF_TEX_SPECTRUM_IMAGE) return EvalSpectrumImageTexture(sv, t, lambda, tc);
    // SpectrumBilerpTexture::Evaluate (textures.h:340-344) through Bilerp(p, {v00, v10, v01, v11}) (util/math.h)
    TexCoord2 c = TexMap2D(sv, t, tc);
    S4 v00 = SpectrumSample(sv, t.spectrum, lambda), v10 = SpectrumSample(sv, t.tex0, lambda);
    S4 v01 = SpectrumSample(sv, t.tex1, lambda), v11 = SpectrumSample(sv, t.tex2, lambda);
    return ((1 - c.st.x) * (1 - c.st.y) * v00 + c.st.x * (1 - c.st.y) * v10 + (1 - c.st.x) * c.st.y * v01 + c.st.x * c.st.y * v11);
}
WF_HD bool IsSimpleSpectrumTexture(int type) { return type == WF_TEX_SPECTRUM_CONSTANT || type == WF_TEX_SPECTRUM_IMAGE || type == WF_TEX_SPECTRUM_BILERP; }
WF_HD bool IsLeafSpectrumTexture(int type) { return IsSimpleSpectrumTexture(type) || type == WF_TEX_SPECTRUM_MARBLE; }
WF_HD S4 EvalSpectrumTextureLeaf(const SceneView &sv, const wf_texture *tp, const Wavelengths &lambda, const TexCtx &tc) {
    if (IsSimpleSpectrumTexture(tp->type)) return EvalSpectrumTextureSimple(sv, *tp, lambda, tc);
    TexCtx cc = tc;
    S4 r;
    MarbleTextureP(sv.noisePerm, sv.lightXforms + tp->xform, tp, &cc, sv.rgb2specCoeffs, sv.rgb2specZNodes, lambda.lambda, r.v);
    return r;
}
// The explicit-stack walk of a spectrum graph; float operands (a scale factor, a mix amount) are float graphs of their own
// (EvalFloatTextureStack).  Frames and states as there.
WF_HD S4 EvalSpectrumTextureStack(const SceneView &sv, int id, const Wavelengths &lambda, const TexCtx &tc) {
#if WF_TEX_FRAMES_STRUCT
    struct Frame { int node; float w; S4 a; };   // (one array of mixed-type frames: see EvalFloatTextureStack)
    Frame fr[WF_TEX_STACK];
#else
    int fnode[WF_TEX_STACK];
    float fw[WF_TEX_STACK];
    S4 fa[WF_TEX_STACK];
#endif
    int sp = 0;
    F_NODE(0) = id << 3;
    S4 ret = S4c(0.f);
    while (sp >= 0) {
        const wf_texture *tp = sv.textures + (F_NODE(sp) >> 3);
        const int type = tp->type;
        const int st = F_NODE(sp) & 7;
        const int self = F_NODE(sp) & ~7;
        if (st == 0) {
            if (IsLeafSpectrumTexture(type)) { ret = EvalSpectrumTextureLeaf(sv, tp, lambda, tc); --sp; continue; }
            if (type == WF_TEX_SPECTRUM_DOTS) { F_NODE(sp) = (InsidePolkaDot(sv, *tp, tc) ? tp->tex1 : tp->tex0) << 3; continue; }
            if (sp + 1 >= WF_TEX_STACK) { ret = S4c(0.f); --sp; continue; }   // (refused at load: never reached)
            // the node's float operand — SpectrumScaledTexture's scale (textures.h:1059-1064: evaluated first, 0 ends the node), the amount of
            // SpectrumMixTexture (:840-850), the checkerboard's (:404-413) or the direction mix's (:880-890) weight — from ONE call site:
            // every inlined copy of the float walk would add its frames to the kernels' scratch
            float w;
            if (type == WF_TEX_SPECTRUM_SCALE || type == WF_TEX_SPECTRUM_MIX) w = EvalFloatTextureStack(sv, type == WF_TEX_SPECTRUM_SCALE ? tp->tex1 : tp->tex2, tc);
            else if (type == WF_TEX_SPECTRUM_DIRECTIONMIX) w = AbsDot(tc.n, N3{tp->map[4], tp->map[5], tp->map[6]});
            else if (type == WF_TEX_SPECTRUM_CHECKERBOARD) w = CheckerboardWeight(sv, *tp, tc);
            else { ret = S4c(0.f); --sp; continue; }
            F_W(sp) = w;
            if (type == WF_TEX_SPECTRUM_SCALE) {
                if (w == 0) { ret = S4c(0.f); --sp; continue; }
                F_NODE(sp) = self | 2; F_NODE(++sp) = tp->tex0 << 3;
                continue;
            }
            F_A(sp) = S4c(0.f);
            if (w != 1) { F_NODE(sp) = self | 2; F_NODE(++sp) = tp->tex0 << 3; continue; }
            ret = S4c(0.f);
            // falls to the "tex0 returned" step below with t0 = 0
        }
        if (type == WF_TEX_SPECTRUM_SCALE) {
            ret = ret * F_W(sp);
            --sp;
            continue;
        }
        if (st != 3) {
            F_A(sp) = ret;
            if (F_W(sp) != 0) { F_NODE(sp) = self | 3; F_NODE(++sp) = tp->tex1 << 3; continue; }
            ret = S4c(0.f);
        }
        {
            const float w = F_W(sp);
            ret = (1 - w) * F_A(sp) + w * ret;
        }
        --sp;
    }
    return ret;
}
#undef F_NODE
#undef F_W
#undef F_A
WF_NI void EvalSpectrumTextureGraphP(const SceneView *svp, int id, const Wavelengths *lambda, const TexCtx *tc, S4 *out) {
    *out = EvalSpectrumTextureStack(*svp, id, *lambda, *tc);
}
WF_HD S4 EvalSpectrumTexture(const SceneView &sv, int id, const Wavelengths &lambda, const TexCtx &tc) {
    const int type = sv.textures[id].type;
    if (IsSimpleSpectrumTexture(type)) return EvalSpectrumTextureSimple(sv, sv.textures[id], lambda, tc);
    if constexpr (WF_DEV_LEAN) { RaiseFatal(sv, WF_FATAL_LEAN_VARIANT); return S4c(0.f); }
    else {
    TexCtx tmp = tc;
    Wavelengths l = lambda;
    S4 r;
    EvalSpectrumTextureGraphP(sv.self, id, &l, &tmp, &r);
    return r;
    }
}

// ---------------------------------------------------------------------------------------------
// geometry accessors
WF_HD V3 LoadP(const SceneView &sv, int v) { return V3{sv.P[3 * v], sv.P[3 * v + 1], sv.P[3 * v + 2]}; }
WF_HD N3 LoadN(const SceneView &sv, int v) { return N3{sv.N[3 * v], sv.N[3 * v + 1], sv.N[3 * v + 2]}; }
WF_HD V3 LoadS(const SceneView &sv, const wf_mesh &mesh, int v) { const auto p = sv.S + 3 * (size_t)(mesh.first_s + (v - mesh.first_vertex)); return V3{p[0], p[1], p[2]}; }
WF_HD V2 LoadUV(const SceneView &sv, int v) { return V2{sv.UV[2 * v], sv.UV[2 * v + 1]}; }

}  // namespace wf
