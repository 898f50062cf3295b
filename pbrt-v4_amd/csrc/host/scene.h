// scene.h — host scene description: the .pbrt parser front end (parser.cpp) records entities the way
// BasicSceneBuilder/BasicScene do (src/pbrt/scene.{h,cpp}, parser.{h,cpp}, paramdict.{h,cpp}), and
// BuildSceneTables (scene_build.cpp) flattens them into the index-addressed tables of wf_scene_desc.
#pragma once

#include "hmath.h"
#include "spectra.h"

#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <stdexcept>

namespace wf {

// Scene errors (the reference's ErrorExit) and back-end failures (its CUDA_CHECK -> LOG_FATAL) are thrown; the extern "C" entry points
// of libwfhost.so catch them and return an error (wfh_last_error), the command-line programs print them and exit
struct SceneError : std::runtime_error { using std::runtime_error::runtime_error; };


// ---- parameters (paramdict.h) ---------------------------------------------------------------------
struct Param {
    std::string type, name;
    std::vector<float> floats;
    std::vector<int> ints;
    std::vector<std::string> strings;
    std::vector<uint8_t> bools;
    const ColorSpace *colorSpace = nullptr;
    mutable bool lookedUp = false;
    std::string loc;
};
enum class SpectrumType { Illuminant, Albedo, Unbounded };

class ParamSet {
  public:
    std::vector<Param> params;
    const ColorSpace *colorSpace = nullptr;
    float GetOneFloat(const std::string &name, float def) const;
    int GetOneInt(const std::string &name, int def) const;
    bool GetOneBool(const std::string &name, bool def) const;
    std::string GetOneString(const std::string &name, const std::string &def) const;
    std::vector<float> GetFloatArray(const std::string &name) const;
    std::vector<std::string> GetStringArray(const std::string &name) const;
    std::vector<int> GetIntArray(const std::string &name) const;
    std::vector<V3> GetPoint3fArray(const std::string &name) const;   // also vector3/normal
    std::vector<V2> GetPoint2fArray(const std::string &name) const;
    std::vector<V3> GetTuple3Array(const std::string &name, const char *type) const;
    bool HasParam(const std::string &name) const;
    V3 GetOnePoint3f(const std::string &name, V3 def) const;
    V3 GetOneVector3f(const std::string &name, V3 def) const;
    // spectrum-valued parameter ("rgb", "spectrum", "blackbody"); null if absent (paramdict.cpp:384-455)
    SpectrumP GetOneSpectrum(const std::string &name, SpectrumP def, SpectrumType st) const;
    std::string GetTexture(const std::string &name) const;  // "texture name" parameter or ""
    const Param *Find(const std::string &name) const;
    void ReportUnused(const std::string &what) const;
};

// ---- recorded entities (scene.h:SceneEntity family) -------------------------------------------------
struct Entity {
    std::string name;   // implementation name ("perspective", "diffuse", ...)
    ParamSet params;
    std::string loc;
};
struct TextureEntity : Entity {
    std::string texName, texType;  // "float" | "spectrum"
    Transform renderFromTexture;
};
struct LightEntity : Entity {
    Transform renderFromLight;
    std::string medium;
};
struct ShapeEntity : Entity {
    Transform renderFromObject;
    bool reverseOrientation = false;
    int materialIndex = -1;
    std::string materialName;
    int lightIndex = -1;
    std::string insideMedium, outsideMedium;
};
struct InstanceDefinition { std::string name; std::vector<ShapeEntity> shapes; };
struct InstanceUse {
    std::string name;
    Transform renderFromInstance;   // (animated: the start-time transformation)
    // AnimatedPrimitive (round 5): the use was created under an animated CTM (ActiveTransform StartTime / EndTime with different CTMs)
    bool animated = false;
    Transform renderFromInstanceEnd;
    float startTime = 0, endTime = 1;
    std::string loc;
};

struct RenderOptions {  // subset of PBRTOptions (options.h)
    int seed = 0;
    int nThreads = 0;
    bool quiet = false;
    bool disablePixelJitter = false, disableWavelengthJitter = false, disableTextureFiltering = false;
    int pixelSamples = -1;               // --spp override
    int pixelBounds[4] = {0, 0, 0, 0};   // --pixelbounds x0,x1,y0,y1 (all zero = unset)
    float cropWindow[4] = {0, 0, 0, 0};  // --cropwindow
    bool hasPixelBounds = false, hasCropWindow = false;
    std::string imageFile;
    bool quickRender = false;            // --quick: a quarter of the resolution, one sample per pixel (film.cpp:92-95, samplers.cpp)
    bool disableImageTextures = false;   // --disable-image-textures: every image map reduced to the coarsest level of its pyramid (util/mipmap.cpp:199-203)
    int renderingSpace = 1;              // --render-coord-sys / Option "rendercoordsys": 0 camera, 1 cameraworld (default), 2 world (cameras.cpp:27-47)
    float displacementEdgeScale = 1;     // --displacement-edge-scale (options.h: scales the target edge length of displaced meshes)
};

struct ParsedScene {
    Entity camera, film, sampler, filter, integrator, accelerator;
    Transform cameraFromWorld, worldFromCamera;  // CTM at Camera (start time)
    Transform worldFromCameraEnd;                // the end-time one: differs for a moving camera (ActiveTransform)
    Transform renderFromWorld;                   // CameraTransform::RenderFromWorld()
    float transformStartTime = 0, transformEndTime = 1;  // TransformTimes at the Camera directive
    std::string cameraMedium;
    const ColorSpace *filmColorSpace = nullptr;
    std::vector<TextureEntity> textures;
    std::vector<std::pair<std::string, Entity>> namedMaterials;
    std::vector<Entity> materials;
    std::vector<LightEntity> lights;
    std::vector<Entity> areaLights;
    std::vector<ShapeEntity> shapes;
    std::map<std::string, InstanceDefinition> instanceDefinitions;
    std::vector<InstanceUse> instances;
    // shapes created under an animated CTM (scene.cpp:277-290: AnimatedShapeSceneEntity): each is a hidden instance definition (its shapes in
    // object space) used once with the animated transformation; they precede the object instances among the top-level primitives (scene.cpp:1511-1577)
    std::vector<InstanceUse> animatedShapes;
    std::vector<std::pair<std::string, Entity>> media;
    std::map<std::string, Transform> mediaTransforms;
    std::string baseDir;
};

// Parses the given files (parser.cpp).  Exits with a message on syntax errors, like the reference.
void ParseFiles(const std::vector<std::string> &files, RenderOptions *opt, ParsedScene *scene);
void ParseString(const std::string &text, RenderOptions *opt, ParsedScene *scene);

// ---- flattened scene ------------------------------------------------------------------------------
// Owns every array wf_scene_desc points into.
struct SceneTables {
    wf_scene_desc desc{};
    std::vector<float> P, N, UV;
    std::vector<float> S;   // shading tangents of the meshes that have them (wf_mesh.first_s)
    std::vector<int32_t> triIndices, triMesh, bvhPrims, infiniteLights;
    std::vector<wf_mesh> meshes;
    std::vector<wf_quadric> quadrics;
    std::vector<wf_instance> instances;
    std::vector<wf_instance_def> instanceDefs;
    std::vector<wf_animated_transform> animated;   // wf_instance.anim_plus1
    int nTopBvhNodes = 0, nTopPrims = 0;
    std::vector<uint32_t> sobolMatrices;       // data/sobol_matrices.bin when the sampler is "sobol"
    std::vector<uint64_t> vdcSobol, vdcSobolInv;
    std::vector<int32_t> haltonPrimes, haltonPermOffsets;
    std::vector<uint16_t> haltonPerms;
    std::vector<wf_bvh_node> bvhNodes;
    SpectrumPool pool;
    std::vector<wf_texture> textures;
    std::vector<wf_material> materials;
    std::vector<wf_light> lights;
    std::vector<wf_light_bvh_node> lightBvh;
    std::vector<wf_transform> lightTransforms;
    std::vector<float> filterData, powerAlias;
    std::vector<wf_image_light> imageLights;
    std::vector<wf_tex_image> texImages;
    std::vector<float> tableData;
    std::vector<wf_medium> media;
    std::vector<float> mediumData;
    // Image::Write (util/image.cpp:986-1005): an image whose colour space is not sRGB is converted to sRGB when the file format cannot say
    // otherwise (everything but .exr).  9 floats (row major) = sRGB.RGBFromXYZ * film.XYZFromRGB when the film's colour space is not sRGB, else empty
    std::vector<float> sRGBFromFilmRGB;
    std::string imageFile;
    bool saveFP16 = true;  // Film "savefp16" (film.cpp:579)
    int spp = 1;
    // wavefront geometry (integrator.cpp:227-236)
    int scanlinesPerPass = 0, maxQueueSize = 0, nPasses = 0;
    bool materialTypePresent[WF_MAT_NTYPES] = {};
    std::vector<int32_t> noisePerm;  // the Perlin permutation (data/noise_perm.txt) when a texture or medium uses noise
    void Finalize();  // fills desc pointers/counters from the vectors
    // on-disk cache of the built tables: one flat file, every array verbatim.  Load() returns false when the
    // file is missing, truncated or from another ABI / build of the table builder.
    bool Save(const std::string &path) const;
    bool Load(const std::string &path);
};

void BuildSceneTables(const ParsedScene &scene, const RenderOptions &opt, SceneTables *out);

// geometry BVH (bvh_build.cpp): SAH build restating BVHAggregate (cpu/aggregates.cpp:140-387,505-521)
// prims: the primitives in the reference's creation order as (primitive id, render-space bounds).  Nodes and ordered
// primitive ids are APPENDED to *nodes / *orderedPrims (child and primitive offsets absolute); returns the root index.
// splitMethod: 0 = "sah" (cpu/aggregates.cpp:198-387), 1 = "hlbvh" (:389-503, 626-722)
int BuildBVH(const std::vector<std::pair<int, B3>> &prims, int maxPrimsInNode, std::vector<wf_bvh_node> *nodes, std::vector<int32_t> *orderedPrims, int splitMethod = 0, bool forceHost = false);
// Morton codes (10 bits per axis of the centroids' offsets in `bounds`) + stable sort, on the device: order[i] = input position of the
// i-th primitive in Morton order, codes[i] its code.  Non-zero return: not available (the host path is used).
typedef int (*MortonSortFn)(int n, const float *centroids, const float bounds[6], uint32_t *codes, uint32_t *order);
void SetMortonSort(MortonSortFn fn);
// the SAH build on the device (include/wf_abi.h: wf_build_bvh_sah), used for trees of at least WF_DEVICE_BVH_MIN primitives (default 200000) when a
// device is visible and WF_HOST_BVH_BUILD is not set: the same tree node for node; a non-zero return falls back to the host build
typedef int (*SahBuildFn)(int n, const float *bounds, int maxPrimsInNode, wf_bvh_node *nodesOut, int32_t *orderOut, int32_t *nNodesOut);
void SetSahBuild(SahBuildFn fn);
// Triangle::Bounds (shapes.cpp:283-290) of global triangle i
B3 TriangleBounds(const std::vector<float> &P, const std::vector<int32_t> &triIndices, int i);
// light BVH (lightbvh_build.cpp): BVHLightSampler ctor (lightsamplers.cpp:105-232)
struct LightBoundsH {
    B3 bounds;
    float phi = 0;
    V3 w{0, 0, 0};
    float cosTheta_o = 0, cosTheta_e = 0;
    bool twoSided = false;
    V3 Centroid() const { return (bounds.pMin + bounds.pMax) / 2; }
};
void BuildLightBVH(const std::vector<std::pair<int, LightBoundsH>> &bvhLightsIn, const B3 &allLightBounds,
                   std::vector<wf_light_bvh_node> *nodes, std::vector<wf_light> *lights);

// ColorEncoding (util/color.h:402-477): how 8-bit texels map to linear floats and back
struct ColorEnc {
    int kind = 1;     // 0 linear, 1 sRGB, 2 gamma
    float gamma = 1;
    static ColorEnc Linear() { ColorEnc e; e.kind = 0; return e; }
    static ColorEnc Parse(const std::string &name);   // "linear" | "sRGB" | "gamma <value>" (ColorEncoding::Get)
    std::string Key() const { return std::to_string(kind) + ":" + std::to_string(gamma); }
    float ToLinear(uint8_t v) const;       // ColorEncoding::ToLinear
    uint8_t FromLinear(float v) const;     // ColorEncoding::FromLinear
    float ToFloatLinear(float v) const;    // ColorEncoding::ToFloatLinear (16-bit PNG samples)
};
// An image as Image::Read leaves it (util/image.h): 8-bit texels keep their bytes + encoding, 16-bit PNG samples are
// stored rounded to half precision, .pfm is float.  Channels are Y | R G B | R G B A, rows top to bottom.
struct HostImage {
    enum Format { U256 = 0, Half = 1, Float = 2 };
    int format = Float;
    int w = 0, h = 0, nc = 0;
    ColorEnc enc;
    std::vector<uint8_t> p8;   // U256
    std::vector<float> p32;    // Half (already rounded to half) and Float
    float Get(size_t i) const { return format == U256 ? enc.ToLinear(p8[i]) : p32[i]; }   // Image::GetChannel's decode
    float Quantize(float v) const;   // what SetChannel / CopyRectIn followed by GetChannel gives back for this format
    uint32_t QuantizeCode(float v) const;   // the stored code itself: the byte (U256) or the half bit pattern (Half)
    void SelectChannels(int first, int count);   // Image::SelectChannels for a channel range
};
// Image::Read (util/image.cpp:1000-1040) for .pfm and .png; throws SceneError with the reference's wording
void ReadImage(const std::string &path, const ColorEnc &enc, HostImage *img);
// image_formats.cpp: .qoi, Radiance .hdr and .tga as the reference's third-party decoders (ext/qoi, stb_image) read them
void ReadQOI(const std::string &path, HostImage *img);
void ReadHDR(const std::string &path, HostImage *img);
void ReadTGA(const std::string &path, HostImage *img);
// a NanoVDB float grid expanded over its index bounding box (nanovdb_io.cpp; values[(z * dim[1] + y) * dim[0] + x], origin min)
struct VdbGrid {
    bool found = false;
    int min[3] = {0, 0, 0}, dim[3] = {0, 0, 0};
    std::vector<float> values;
    float invMat[9] = {}, vec[3] = {};   // Map: index = invMat * (p - vec) (worldToIndexF)
    double worldBBox[6] = {};
    float background = 0;
    int gridClass = 0;
};
void ReadNanoVDBGrid(const std::string &filename, const std::string &gridName, VdbGrid *out);
// a measured BRDF (.bsdf tensor file) appended to a table_data vector; returns the header offset (measured_io.cpp)
int ReadMeasuredBRDF(const std::string &filename, std::vector<float> *table);
float RoundToHalf(float f);
uint16_t FloatToHalfBits(float f);

// Shape "loopsubdiv" (loopsubdiv.cpp): the limit-surface triangle mesh of a control mesh, in object space
void LoopSubdivide(int nLevels, const std::vector<int> &vertexIndices, const std::vector<V3> &P, std::vector<int> *outIndices,
                   std::vector<V3> *outP, std::vector<V3> *outN);

// image output (image_io.cpp)
bool WritePFM(const std::string &path, const float *rgb, int w, int h);
bool ReadPFM(const std::string &path, std::vector<float> *rgb, int *w, int *h);
bool WriteImage(const std::string &path, const float *rgb, int w, int h);  // by extension: .pfm, .exr
struct SceneTables;
// the film's RGB image as Film::WriteImage -> Image::Write leaves it in `path`: converted to sRGB for formats other than .exr when the film's
// colour space is another one (SceneTables::sRGBFromFilmRGB)
bool WriteFilmImage(const SceneTables &T, const std::string &path, std::vector<float> &rgb, int w, int h);
// film accumulators ([h][w][4] doubles: rgbSum, weightSum) -> output RGB, as RGBFilm::GetImage does
void FilmToRGB(const wf_film &F, const double *film, int w, int h, float *rgb, bool saveFP16);
bool WriteEXRChannels(const std::string &path, const std::vector<std::string> &names, const float *data, int w, int h, bool half);
void GBufferFilmImage(const wf_film &F, const double *film, const wf_gbuffer_pixel *gb, int w, int h, bool saveFP16, std::vector<std::string> *names, std::vector<float> *out);
void SpectralFilmImage(const wf_film &F, const double *film, const double *spectral, int w, int h, bool saveFP16, std::vector<std::string> *names, std::vector<float> *out);

}  // namespace wf
