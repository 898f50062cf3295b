/*
 * wf_abi.h — C ABI of the MI355X wavefront path-tracing back end (libwfhip.so).
 *
 * This is the drop-in boundary described in SURVEY.md §8(b).  In the reference the boundary is
 *   (1) class WavefrontAggregate                     src/pbrt/wavefront/integrator.h:32-54
 *   (2) the lambda dispatchers ParallelFor/Do        src/pbrt/wavefront/integrator.h:91-113
 *       and ForAllQueued                             src/pbrt/wavefront/workqueue.h:118-137
 *   (3) the memory resource handed to the integrator src/pbrt/wavefront/integrator.h:88-89
 * i.e. "the lambda is the kernel".  Here every lambda body is a named, hand-written HIP kernel and
 * every launch site is one extern "C" function; the host loop (pbrt-v4_amd/csrc/host/integrator.cpp)
 * is a restatement of WavefrontPathIntegrator::Render (src/pbrt/wavefront/integrator.cpp:290-493)
 * that calls them in the same order.
 *
 * Conventions
 *  - plain C: PODs, raw pointers, sizes.  No C++/torch types cross this boundary.
 *  - every entry point returns 0 on success or a non-zero hipError_t-like code; wf_last_error()
 *    returns a message.  (The reference aborts via CUDA_CHECK -> LOG_FATAL, gpu/util.h:35-49; the
 *    host wrapper does the same on a non-zero return.)
 *  - one in-order HIP stream per context; no entry point synchronises the host except
 *    wf_sync / wf_*_download.
 *  - scene tables are flat, index-addressed arrays (no host pointers live on the device).  The same
 *    wf_scene_desc (host memory) is what oracle/ consumes, so HIP and oracle see identical inputs.
 */
#ifndef WF_ABI_H
#define WF_ABI_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define WF_ABI_VERSION 12          /* 12 (round 5): wf_instance.anim_plus1, wf_scene_desc.n_animated / animated (AnimatedPrimitive) */
#define WF_NSPECTRUM 4           /* NSpectrumSamples, util/spectrum.h:36 */
#define WF_LAMBDA_MIN 360
#define WF_LAMBDA_MAX 830
#define WF_NDENSE 471            /* WF_LAMBDA_MAX - WF_LAMBDA_MIN + 1 */

/* ------------------------------------------------------------------------------------------- */
/* Scene description: flat host tables                                                          */
/* ------------------------------------------------------------------------------------------- */

typedef struct wf_transform {    /* util/transform.h:Transform — m and its inverse, row major */
    float m[4][4];
    float mInv[4][4];
} wf_transform;
/* AnimatedTransform (util/transform.h:444-560, util/transform.cpp:375-395): the transformations at the two ends of the
 * TransformTimes interval and their decomposition (translation, rotation quaternion x y z w, scale matrix) as the reference's
 * constructor computes it at load; AnimatedTransform::Interpolate (util/transform.cpp:1062-1081) is evaluated from these. */
typedef struct wf_animated_transform {
    wf_transform start, end;
    float start_time, end_time;
    int32_t actually_animated;    /* startTransform != endTransform */
    int32_t has_rotation;
    float T[2][3];
    float R[2][4];
    float S[2][4][4];
} wf_animated_transform;

/* Medium (media.h:226-352): HomogeneousMedium and GridMedium ("uniformgrid"), both with the Henyey-Greenstein
 * phase function.  Spectra are DenselySampledSpectrum tables (471 floats each in spectrum_data), already
 * multiplied by the medium's "scale" / Le scale as the reference's constructors do (media.h:233-241). */
enum wf_medium_type { WF_MEDIUM_HOMOGENEOUS = 0, WF_MEDIUM_GRID = 1, WF_MEDIUM_RGB_GRID = 2,
                      WF_MEDIUM_CLOUD = 3 /* media.h:430-525: procedural density from Perlin noise inside `bounds`, one homogeneous majorant */,
                      WF_MEDIUM_NANOVDB = 4 /* media.h:599-679: density (and temperature) FloatGrids, expanded to dense blocks over their index
                                               bounding boxes; 64^3 majorant grid over the world bounding box.  Parity unpinned (third-party format) */ };
typedef struct wf_medium {
    int32_t type;
    int32_t sigma_a_offset, sigma_s_offset, le_offset;   /* offsets into spectrum_data */
    float g;
    int32_t is_emissive;
    /* GridMedium only */
    float bounds[6];                 /* medium-space box p0, p1 */
    wf_transform render_from_medium;
    int32_t nx, ny, nz, density_offset;                 /* SampledGrid<Float> density, offsets into medium_data */
    int32_t le_nx, le_ny, le_nz, le_scale_offset;       /* SampledGrid<Float> LeScale (already times 1/photometric(Le)) */
    int32_t maj_res[3], maj_offset;                     /* MajorantGrid 16^3 (media.h:105-133) */
    /* RGBGridMedium (media.h:355-430): SampledGrid<RGBUnboundedSpectrum> sigma_a / sigma_s and SampledGrid<RGBIlluminantSpectrum>
     * Le as {c0, c1, c2, scale} per cell in medium_data (-1 = grid absent); le_offset = the colour space's dense illuminant */
    int32_t rgb_a_offset, rgb_s_offset, rgb_le_offset;
    float sigma_scale, le_scale;
    /* GridMedium "temperature" (media.h:305-318): SampledGrid<Float> of nx*ny*nz kelvins in medium_data, or -1 */
    int32_t temperature_offset;
    float temperature_shift, temperature_scale;
    /* CloudMedium */
    float cloud_density, cloud_wispiness, cloud_frequency;
    /* NanoVDBMedium: the dense density block = medium_data[density_offset ..], vdb_dim[0] * vdb_dim[1] * vdb_dim[2] floats, x fastest,
       voxel (i, j, k) at (i - vdb_min[0], ...); index coordinates of a medium-space point p: vdb_inv_mat * (p - vdb_vec)
       (nanovdb::Map::applyInverseMapF); vdb_background outside the block.  The temperature grid (temperature_offset; -1: none) likewise;
       le_scale = "Lescale", temperature_shift / temperature_scale as for the grid medium */
    int32_t vdb_min[3], vdb_dim[3];
    float vdb_inv_mat[9], vdb_vec[3], vdb_background;
    int32_t vdbt_min[3], vdbt_dim[3];
    float vdbt_inv_mat[9], vdbt_vec[3], vdbt_background;
} wf_medium;

/* Spectrum (util/spectrum.h:48-67 TaggedPointer family) flattened to a 32-byte descriptor.
 * Sample values live in wf_scene_desc::spectrum_data (float pool). */
enum wf_spectrum_type {
    WF_SPEC_NONE = 0,
    WF_SPEC_CONSTANT = 1,        /* c0 = value */
    WF_SPEC_DENSE = 2,           /* offset -> 471 floats for 360..830 nm (DenselySampledSpectrum) */
    WF_SPEC_PIECEWISE = 3,       /* offset -> n lambdas followed by n values */
    WF_SPEC_RGB_ALBEDO = 4,      /* sigmoid polynomial c0,c1,c2 (util/color.h:332-364) */
    WF_SPEC_RGB_UNBOUNDED = 5,   /* scale * sigmoid */
    WF_SPEC_RGB_ILLUMINANT = 6,  /* scale * sigmoid * dense illuminant at offset */
    WF_SPEC_BLACKBODY = 7        /* c0 = T, c1 = normalization factor */
};
typedef struct wf_spectrum {
    int32_t type;
    int32_t offset;
    int32_t n;
    float scale;
    float c0, c1, c2;
    int32_t pad;
} wf_spectrum;

/* Textures (textures.h).  Round 1 implements the "basic" evaluator set that the configs need. */
enum wf_texture_type {
    WF_TEX_FLOAT_CONSTANT = 0,     /* f0 */
    WF_TEX_SPECTRUM_CONSTANT = 1,  /* spectrum */
    WF_TEX_FLOAT_SCALE = 2,        /* tex0 * tex1 (float) */
    WF_TEX_SPECTRUM_SCALE = 3,     /* spectrum tex0 * float tex1 */
    WF_TEX_FLOAT_MIX = 4,          /* lerp(amount tex2, tex0, tex1) */
    WF_TEX_SPECTRUM_MIX = 5,
    WF_TEX_FLOAT_IMAGE = 6,        /* image id in i0, mapping in map */
    WF_TEX_SPECTRUM_IMAGE = 7,
    WF_TEX_FLOAT_CHECKERBOARD = 8,     /* 2D checkerboard over the UVMapping in map[0..3]; tex0 = "tex1", tex1 = "tex2" (textures.h:352-420) */
    WF_TEX_SPECTRUM_CHECKERBOARD = 9,
    WF_TEX_FLOAT_BILERP = 10,          /* textures.h:300-330: v00 = f0, v01 = f1, v10 = map[10], v11 = map[11] over the 2D mapping */
    WF_TEX_SPECTRUM_BILERP = 11,       /* spectra ids: v00 = spectrum, v10 = tex0, v01 = tex1, v11 = tex2 */
    WF_TEX_FLOAT_DIRECTIONMIX = 12,    /* textures.h:830-860: dir (render space, normalized) in map[4..6]; tex0 = "tex2", tex1 = "tex1" */
    WF_TEX_SPECTRUM_DIRECTIONMIX = 13,
    /* the procedural textures of the reference's "universal" evaluator (textures.h:427-502,775-800,1079-1122; util/noise.cpp).
       3D mapping = PointTransformMapping: textureFromRender = inverse of light_transforms[xform] */
    WF_TEX_FLOAT_FBM = 14,             /* i0 = octaves, f0 = omega ("roughness") */
    WF_TEX_FLOAT_WRINKLED = 15,        /* Turbulence: i0 = octaves, f0 = omega */
    WF_TEX_FLOAT_WINDY = 16,
    WF_TEX_SPECTRUM_MARBLE = 17,       /* i0 = octaves, f0 = omega, f1 = scale, map[10] = variation; RGBAlbedoSpectrum in sRGB */
    WF_TEX_FLOAT_DOTS = 18,            /* 2D mapping in map; tex0 = evaluated outside the dots, tex1 = inside (see scene_build.cpp for which
                                          parameter that is in the reference) */
    WF_TEX_SPECTRUM_DOTS = 19
};
typedef struct wf_texture {
    int32_t type;
    int32_t spectrum;            /* index into spectra, or -1 */
    int32_t tex0, tex1, tex2;    /* child texture ids, or -1 */
    int32_t i0;                  /* IMAGE: index into tex_images (SPECTRUM_IMAGE: `spectrum` holds the SpectrumType 0 albedo / 1 unbounded / 2 illuminant) */
    float f0, f1;                /* constant value; IMAGE: scale, invert flag */
    float map[12];               /* UVMapping: su, sv, du, dv (textures.h:76-104); PlanarMapping: ds, dt in [2..3], vs in [4..6], vt in [7..9] */
    int32_t mapping;             /* wf_tex_mapping */
    int32_t xform;               /* spherical / cylindrical / planar: renderFromTexture in light_transforms (textureFromRender = its inverse) */
} wf_texture;
enum wf_tex_mapping { WF_TEXMAP_UV = 0, WF_TEXMAP_SPHERICAL = 1, WF_TEXMAP_CYLINDRICAL = 2, WF_TEXMAP_PLANAR = 3,
                      WF_TEXMAP_POINT3D = 4 /* checkerboard "dimension" 3: PointTransformMapping through xform */ };

/* Materials (materials.h).  tex[] holds texture ids; meaning per type is listed in DESIGN.md and
 * mirrored by the WF_MT_* index constants below. */
enum wf_material_type {
    WF_MAT_INTERFACE = 0,            /* "interface": null material, ray passes through */
    WF_MAT_DIFFUSE = 1,              /* materials.h:449-482 */
    WF_MAT_CONDUCTOR = 2,            /* materials.h:485-548 */
    WF_MAT_DIELECTRIC = 3,           /* materials.h:144-215 */
    WF_MAT_THIN_DIELECTRIC = 4,      /* materials.h:218-264 */
    WF_MAT_DIFFUSE_TRANSMISSION = 5, /* materials.h:672-720 */
    WF_MAT_COATED_DIFFUSE = 6,       /* materials.h:551-608 */
    WF_MAT_COATED_CONDUCTOR = 7,     /* materials.h:611-669 */
    WF_MAT_SUBSURFACE = 8,           /* materials.h:696-790: dielectric boundary + TabulatedBSSRDF (K12, wf_sample_subsurface) */
    WF_MAT_HAIR = 9,                 /* materials.h:353-427: HairBxDF (bxdfs.h:921-1019); h = -1 + 2 v */
    WF_MAT_MEASURED = 10,            /* materials.h:849-892: MeasuredBxDF (bxdfs.h:1022-1069, bxdfs.cpp:998-1113) over the tables at measured_table */
    WF_MAT_NTYPES = 11,              /* the types above have an evaluation queue + kernel each */
    WF_MAT_MIX = 11                  /* materials.h:272-332: resolved to one of mix[0..1] when the hit is routed (intersect.h:92-97) */
};
/* tex[] slots */
#define WF_MT_REFLECTANCE 0   /* diffuse / conductor(reflectance) / difftrans / coated diffuse */
#define WF_MT_TRANSMITTANCE 1 /* difftrans */
#define WF_MT_ETA 1           /* conductor eta (spectrum tex) ; dielectric: spectra[eta_spectrum] */
#define WF_MT_K 2             /* conductor k */
#define WF_MT_UROUGH 3
#define WF_MT_VROUGH 4
#define WF_MT_THICKNESS 5     /* coated */
#define WF_MT_G 6             /* coated */
#define WF_MT_ALBEDO 7        /* coated */
#define WF_MT_AMOUNT 0        /* mix */
#define WF_MT_MFP 1           /* subsurface: mean free path (with WF_MT_REFLECTANCE), when sigma_a / sigma_s are not given */
#define WF_MT_SIGMA_A 5       /* subsurface (WF_MATFLAG_SSS_COEFFICIENTS) */
#define WF_MT_SIGMA_S 6
/* hair: sigma_a = WF_MT_SIGMA_A (spectrum) or colour = WF_MT_REFLECTANCE (spectrum) or the two melanin concentrations */
#define WF_MT_HAIR_ETA 1      /* float textures */
#define WF_MT_HAIR_ALPHA 2
#define WF_MT_HAIR_BETA_M 3
#define WF_MT_HAIR_BETA_N 4
#define WF_MT_HAIR_EUMELANIN 8
#define WF_MT_HAIR_PHEOMELANIN 9
#define WF_MT_NTEX 12
/* coated conductor: interface roughness in UROUGH/VROUGH, conductor in slots 8..11 */
#define WF_MT_COND_UROUGH 8
#define WF_MT_COND_VROUGH 9
#define WF_MT_COND_ETA 10
#define WF_MT_COND_K 11
typedef struct wf_material {
    int32_t type;
    int32_t flags;               /* bit0 remaproughness, bit1 conductor given by reflectance */
    int32_t tex[WF_MT_NTEX];
    int32_t eta_spectrum;        /* dielectric / coated: spectrum id of eta (Spectrum eta) */
    int32_t maxdepth, nsamples;  /* coated */
    float scale;                 /* difftrans scale */
    int32_t displacement;        /* float texture id or -1 */
    int32_t normalmap;           /* image id or -1 */
    int32_t mix[2];              /* WF_MAT_MIX: the two material ids; the "amount" float texture is tex[WF_MT_AMOUNT] */
    float sss_eta;               /* WF_MAT_SUBSURFACE: the scalar eta of SubsurfaceMaterial (materials.h:786) */
    int32_t sss_table;           /* offset into table_data of the material's BSSRDFTable (100 albedo x 64 radius samples, bssrdf.h:73-96):
                                    rhoSamples[100] radiusSamples[64] profile[6400] rhoEff[100] profileCDF[6400]; "scale" is `scale` above */
    int32_t measured_table;      /* WF_MAT_MEASURED: offset into table_data of the MeasuredBxDFData (bxdfs.cpp:846-865), WF_MEASURED_HEADER_WORDS
                                    int32 words (stored as float bit patterns) followed anywhere in table_data by the arrays they point to:
                                    [0] isotropic, [16 + 16 k ...] the PiecewiseLinear2D (util/sampling.h:1298-1745) k = 0 ndf, 1 sigma, 2 vndf,
                                    3 luminance, 4 spectra as: size_x, size_y, param_size[3], param_stride[3], param_values offset[3], data offset,
                                    marginal_cdf offset, conditional_cdf offset (-1: built without cdf); all offsets absolute in table_data */
} wf_material;
#define WF_MEASURED_HEADER_WORDS 96
#define WF_MATFLAG_REMAP_ROUGHNESS 1
#define WF_MATFLAG_CONDUCTOR_REFLECTANCE 2
#define WF_MATFLAG_SSS_COEFFICIENTS 4   /* subsurface: sigma_a / sigma_s textures given (else reflectance + mfp) */

/* Lights (lights.h). */
enum wf_light_type {
    WF_LIGHT_POINT = 0,
    WF_LIGHT_DISTANT = 1,
    WF_LIGHT_SPOT = 2,
    WF_LIGHT_DIFFUSE_AREA = 3,
    WF_LIGHT_UNIFORM_INFINITE = 4,
    WF_LIGHT_IMAGE_INFINITE = 5,
    WF_LIGHT_PROJECTION = 7,          /* lights.h:280-350: xform = renderFromLight * Scale(1,-1,1), xform2 = screenFromLight, image = RGB wf_tex_image */
    WF_LIGHT_PORTAL_INFINITE = 8,     /* lights.h:631-731 PortalImageInfiniteLight: image = index into image_lights (portal fields) */
    WF_LIGHT_GONIOMETRIC = 6          /* lights.h:353-404: xform = renderFromLight * swapYZ, image = one-channel wf_tex_image, area = mean texel */
};
typedef struct wf_light {
    int32_t type;
    int32_t flags;               /* bit0 twoSided (area), bit1 delta-position via zero alpha */
    int32_t spectrum_offset;     /* DenselySampledSpectrum (471 floats) in spectrum_data */
    float scale;
    int32_t tri;                 /* DIFFUSE_AREA: global primitive id (triangle, or n_triangles + sphere index) */
    float area;                  /* DIFFUSE_AREA: shape.Area() */
    int32_t bit_trail;           /* BVH light sampler: lightToBitTrail value, -1 if not in light BVH */
    int32_t infinite_index;      /* index in infinite_lights, or -1 */
    float pos[3];                /* POINT/SPOT: renderFromLight(0,0,0);  DISTANT: direction w (normalized) */
    float cosFalloffStart, cosFalloffEnd; /* SPOT */
    float sceneCenter[3];        /* DISTANT / infinite: set by Preprocess (lights.h:243,546) */
    float sceneRadius;
    int32_t xform;               /* index into light_transforms (SPOT / IMAGE_INFINITE), or -1 */
    int32_t image;               /* IMAGE_INFINITE: index into image_lights; GONIOMETRIC / PROJECTION: index into tex_images */
    int32_t xform2;              /* PROJECTION: screenFromLight in light_transforms */
    int32_t alpha_tex_plus1;     /* DIFFUSE_AREA: 1 + the float texture id of the emitter's alpha mask, 0 = none */
    float screen_bounds[4];      /* PROJECTION: screenBounds pMin.xy, pMax.xy */
} wf_light;
#define WF_LIGHTFLAG_TWOSIDED 1
#define WF_LIGHTFLAG_DELTA_POSITION 2 /* an emitter whose alpha is the constant 0 (lights.cpp:690-706): never hit, sampled without MIS */

/* PiecewiseConstant2D over [0,1]^2 (util/sampling.h:698-790): per row func[nx] + cdf[nx+1], row integrals,
 * marginal func[ny] + cdf[ny+1]; offsets into wf_scene_desc::table_data */
typedef struct wf_pc2d {
    int32_t nx, ny;
    int32_t cond_func_offset, cond_cdf_offset, cond_int_offset;
    int32_t marg_func_offset, marg_cdf_offset;
    float marg_int;
} wf_pc2d;
/* ImageInfiniteLight (lights.h:566-662): square equal-area (octahedral) RGB float image + its sampling
 * distribution and the MIS-compensated one (lights.cpp:1009-1040) */
typedef struct wf_image_light {
    int32_t res;                 /* image is res x res */
    int32_t pixel_offset;        /* res*res*3 floats (RGB interleaved, row 0 first) in table_data */
    wf_pc2d distribution, compensated;
    /* PortalImageInfiniteLight (lights.h:631-731): pixel_offset holds the image rectified to the portal's (alpha, beta) parametrisation */
    int32_t is_portal;
    int32_t func_offset;         /* res*res floats: the sampling function (WindowedPiecewiseConstant2D::func, util/sampling.h:895-990) */
    int32_t sat_offset;          /* res*res doubles (two float slots each, even offset): its summed-area table (util/sampling.h:830-892) */
    int32_t pad;
    float portal[4][3];          /* the portal quadrilateral in render space */
    float portal_frame[3][3];    /* Frame::FromXY(p03, p01): x, y, z */
} wf_image_light;

/* MIPMap of an image texture (util/mipmap.h:49-92): float pyramid levels (Image::GeneratePyramid,
 * util/image.cpp) stored one after the other in table_data, level 0 first, rows top to bottom, channels interleaved */
enum wf_wrap_mode { WF_WRAP_BLACK = 0, WF_WRAP_CLAMP = 1, WF_WRAP_REPEAT = 2, WF_WRAP_OCTAHEDRAL = 3 };
enum wf_mip_filter { WF_MIP_POINT = 0, WF_MIP_BILINEAR = 1, WF_MIP_TRILINEAR = 2, WF_MIP_EWA = 3 };
enum wf_texel_format { WF_TEXEL_FLOAT = 0, WF_TEXEL_U8 = 1, WF_TEXEL_HALF = 2 };
typedef struct wf_tex_image {
    int32_t res[2];
    int32_t n_levels, n_channels;   /* 1 (Y), 3 (R G B) or 4 (R G B A: float lookups return A, util/mipmap.cpp:403-405) */
    int32_t wrap, filter;
    int32_t level_offset[20];       /* float offsets of the levels in table_data */
    int32_t format;                 /* WF_TEXEL_FLOAT: n floats per level; WF_TEXEL_U8: the level's texels as bytes packed from its offset
                                       (PixelFormat::U256 images, util/image.h:40-60), value = table_data[lut_offset + code] — the 256-entry
                                       table of the image's ColorEncoding::ToLinear; WF_TEXEL_HALF: IEEE half bit patterns, 2 bytes per texel */
    int32_t lut_offset;
    int32_t ewa_lut_offset;         /* WF_MIP_EWA: the 128-entry Gaussian weight table (util/mipmap.cpp:59-191) in table_data */
    float max_anisotropy;           /* WF_MIP_EWA: MIPMapFilterOptions::maxAnisotropy */
} wf_tex_image;

/* Light BVH node, 32 bytes, same content as LightBVHNode/CompactLightBounds (lightsamplers.h:101-257) */
typedef struct wf_light_bvh_node {
    uint16_t w_oct[2];           /* OctahedralVector */
    float phi;
    uint32_t cos_bits;           /* qCosTheta_o:15 | qCosTheta_e:15 << 15 | twoSided << 30 */
    uint16_t qb[2][3];
    uint32_t child_or_light;     /* childOrLightIndex:31 | isLeaf << 31 */
    uint32_t pad;
} wf_light_bvh_node;

/* Geometry BVH node: the reference's LinearBVHNode (cpu/aggregates.cpp:129-137), 32 bytes. */
typedef struct wf_bvh_node {
    float bmin[3];
    float bmax[3];
    int32_t offset;              /* primitivesOffset (leaf) | secondChildOffset (interior) */
    uint16_t nprims;             /* 0 -> interior */
    uint8_t axis;
    uint8_t pad;
} wf_bvh_node;

/* Per-mesh record (util/mesh.h:TriangleMesh + the primitive wrapper, cpu/primitive.h) */
typedef struct wf_mesh {
    int32_t first_tri;           /* first global triangle id */
    int32_t ntris;
    int32_t first_vertex;        /* offset into P/N/UV */
    int32_t nverts;
    int32_t flags;               /* WF_MESH_* */
    int32_t material;            /* material id; -1 = interface (no material) */
    int32_t first_light;         /* light id of this mesh's first triangle's area light, or -1 */
    int32_t alpha_tex;           /* float texture id or -1 */
    int32_t medium_inside, medium_outside; /* medium ids or -1; both -1 = no MediumInterface */
    int32_t first_s;             /* WF_MESH_HAS_S: index of the mesh's first shading tangent in wf_scene_desc.S (vertex i of the mesh: first_s + i) */
    int32_t pad;
} wf_mesh;
#define WF_MESH_HAS_N 1
#define WF_MESH_HAS_UV 2
#define WF_MESH_FLIP_NORMAL 4    /* reverseOrientation ^ transformSwapsHandedness */
#define WF_MESH_HAS_MEDIUM_INTERFACE 8
#define WF_MESH_HAS_S 32            /* "S" shading tangents (util/mesh.h:43, shapes.h:951-959) */
#define WF_MESH_REVERSE_ORIENTATION 16  /* the shape's own reverseOrientation (Sphere::Sample flips by it alone, shapes.h:274) */

/* Sphere, Disk, Cylinder (shapes.h:107-383, 385-540, 543-748), kept in object space like the reference's: primitive
 * id n_triangles + index.  Material / area light / media / orientation live in a wf_mesh entry with ntris = 0 and
 * first_tri = that id. */
enum wf_quadric_type { WF_QUADRIC_SPHERE = 0, WF_QUADRIC_DISK = 1, WF_QUADRIC_CYLINDER = 2,
                       /* BilinearPatch (shapes.h:1279-1510): a primitive of the same id range, in RENDER space.  The record's
                          render_from_object storage holds the patch instead of a transformation: m = p00 p10 p01 p11 (12 floats, then
                          uv00.st uv10.st), mInv = n00 n10 n01 n11 (12 floats, then uv01.st uv11.st); pad[0] bit 0: has normals, bit 1: has uv */
                       WF_QUADRIC_BILINEAR = 3,
                       /* Curve (shapes.h:1200-1270): one u-range of a cubic Bezier curve.  radius = width[0], theta_z_min = width[1],
                          z_min = uMin, z_max = uMax, theta_z_max = normalAngle, phi_max = invSinNormalAngle, inner_radius = curve type
                          (0 flat, 1 cylinder, 2 ribbon), ext[0..11] = the object-space control points, ext[12..17] = the ribbon normals.
                          Its hit record holds (u, v, tHit). */
                       WF_QUADRIC_CURVE = 4 };
typedef struct wf_quadric {
    float radius, z_min, z_max, theta_z_min, theta_z_max, phi_max;  /* disk: z_min = z_max = height */
    int32_t mesh;
    int32_t type;                      /* wf_quadric_type */
    float inner_radius;                /* disk */
    float pad[3];
    wf_transform render_from_object;   /* m = renderFromObject, mInv = objectFromRender */
    float ext[20];                     /* WF_QUADRIC_CURVE: control points and ribbon normals */
} wf_quadric;

/* Object instances (ObjectBegin / ObjectInstance): the reference wraps an instance definition's own BVHAggregate
 * (built with maxPrimsInNode = 1, scene.cpp:1539-1543) in a TransformedPrimitive (cpu/primitive.h:83-118,
 * cpu/primitive.cpp:112-130) that enters the top-level BVH with the transformed bounds.  Here: the definition's
 * triangles live in the global vertex / triangle tables in the definition's own render space, its BVH nodes follow
 * the top-level ones in bvh_nodes (child and primitive offsets absolute), and an instance is the primitive
 * n_triangles + n_quadrics + its index.  Area lights inside definitions are not supported by the reference either. */
typedef struct wf_instance {
    wf_transform render_from_instance;  /* TransformedPrimitive::renderFromPrimitive: m and mInv */
    int32_t def;                        /* index into instance_defs */
    int32_t anim_plus1;                 /* AnimatedPrimitive (cpu/primitive.h:103): 1 + index into wf_scene_desc.animated, 0 = a static instance
                                           (render_from_instance then holds the start transformation) */
    int32_t pad[2];
} wf_instance;
typedef struct wf_instance_def {
    int32_t bvh_root;                   /* root of the definition's BVH in bvh_nodes */
    int32_t n_nodes;
    int32_t first_prim, n_prims;        /* its range of bvh_prims */
    float bounds[6];                    /* BVHAggregate::Bounds() = root bounds */
    int32_t pad[2];
} wf_instance_def;

enum wf_camera_type { WF_CAMERA_PERSPECTIVE = 0, WF_CAMERA_ORTHOGRAPHIC = 1, WF_CAMERA_SPHERICAL = 2, WF_CAMERA_REALISTIC = 3 };
typedef struct wf_camera {
    int32_t type;
    wf_transform cameraFromRaster;      /* cameras.h:ProjectiveCamera */
    wf_transform renderFromCamera;      /* CameraTransform::renderFromCamera.startTransform */
    float lensRadius, focalDistance;
    float shutterOpen, shutterClose;
    float dxCamera[3], dyCamera[3];
    float minPosDifferentialX[3], minPosDifferentialY[3];
    float minDirDifferentialX[3], minDirDifferentialY[3];
    int32_t medium;
    int32_t spherical_mapping;          /* SphericalCamera (cameras.h:370-420): 0 equal-area, 1 equirectangular */
    /* RealisticCamera (cameras.h:466-580): the lens prescription and the exit-pupil bounds computed at load, in table_data */
    int32_t n_lens_elements, lens_offset;        /* n x {curvatureRadius, thickness, eta, apertureRadius} (metres), front element first */
    int32_t n_exit_pupil_bounds, exit_pupil_offset; /* n x {pMin.x, pMin.y, pMax.x, pMax.y} by distance from the film centre */
    float physical_extent[4];                    /* film rectangle pMin.xy, pMax.xy (metres) */
    float film_diagonal;                         /* Film::Diagonal() (metres) */
    int32_t aperture_image;                      /* index into tex_images (one channel, level 0) or -1: circular stop */
    /* camera motion blur: CameraTransform::renderFromCamera as the AnimatedTransform it is (cameras.h:27-110).  When
       anim.actually_animated is 0 the kernels use renderFromCamera above (= anim.start) and nothing else of it. */
    wf_animated_transform anim;
} wf_camera;

enum wf_filter_type { WF_FILTER_BOX = 0, WF_FILTER_GAUSSIAN = 1, WF_FILTER_MITCHELL = 2,
                      WF_FILTER_SINC = 3, WF_FILTER_TRIANGLE = 4 };
typedef struct wf_filter {
    int32_t type;
    float radius[2];
    /* FilterSampler tables (filters.h:26-45, filters.cpp:133-147): f[ny][nx], and the
       PiecewiseConstant2D over it: per row func[nx] + cdf[nx+1], marginal func[ny] + cdf[ny+1]. */
    int32_t nx, ny;
    int32_t f_offset;            /* all offsets into filter_data (floats) */
    int32_t cond_func_offset, cond_cdf_offset, cond_int_offset;
    int32_t marg_func_offset, marg_cdf_offset;
    float marg_int;
    float domain_min[2], domain_max[2];
} wf_filter;

typedef struct wf_film {
    int32_t full_res[2];
    int32_t pixel_min[2], pixel_max[2];   /* pixelBounds */
    float imaging_ratio;                  /* PixelSensor::imagingRatio */
    float max_component_value;
    int32_t rbar_offset, gbar_offset, bbar_offset; /* dense spectra in spectrum_data */
    float XYZFromSensorRGB[3][3];
    float outputRGBFromSensorRGB[3][3];
    /* film type (film.h): RGBFilm, or SpectralFilm (film.h:401-530) — the RGB accumulators plus n_buckets spectral buckets over
       [lambda_min, lambda_max]; wavelengths are then sampled uniformly over that range (SampledWavelengths::SampleUniform) */
    int32_t type;                         /* enum wf_film_type */
    int32_t n_buckets;
    float lambda_min, lambda_max;
    /* GBufferFilm (film.h:319-400): the geometry channels are stored in `outputFromRender` space — the camera's (RenderFromCamera
       applied inversely: apply_inverse = 1, "coordinatesystem" "camera", the default) or the world's (WorldFromRender) */
    wf_transform gbuffer_from_render;
    int32_t apply_inverse;
    float RGBFromXYZ[3][3];               /* the film colour space's matrix (albedo -> RGB, SampledSpectrum::ToRGB) */
    int32_t illuminant_offset;            /* the film colour space's illuminant, dense, in spectrum_data (albedo * illuminant, film.cpp:631-633) */
} wf_film;
enum wf_film_type { WF_FILM_RGB = 0, WF_FILM_SPECTRAL = 1, WF_FILM_GBUFFER = 2 };
/* GBufferFilm::Pixel (film.h:375-387) beside the RGB accumulators: one record per pixel */
typedef struct wf_gbuffer_pixel {
    double gbuffer_weight_sum, rgb_albedo_sum[3];
    int64_t var_n[3];                     /* VarianceEstimator<Float> rgbVariance[3] (util/sampling.h:484-520): n, mean, S */
    float var_mean[3], var_s[3];
    float p_sum[3], dzdx_sum, dzdy_sum, n_sum[3], ns_sum[3], uv_sum[2];
    float pad;
} wf_gbuffer_pixel;

enum wf_sampler_type { WF_SAMPLER_ZSOBOL = 0, WF_SAMPLER_INDEPENDENT = 1, WF_SAMPLER_STRATIFIED = 2, WF_SAMPLER_PADDED_SOBOL = 3,
                       WF_SAMPLER_HALTON = 4, WF_SAMPLER_SOBOL = 5 /* samplers.h:479-565: needs wf_scene_desc.sobol_matrices */ };
enum wf_randomize { WF_RAND_NONE = 0, WF_RAND_PERMUTE_DIGITS = 1, WF_RAND_FAST_OWEN = 2, WF_RAND_OWEN = 3 };
typedef struct wf_sampler {
    int32_t type;
    int32_t spp;
    int32_t seed;
    int32_t randomize;
    int32_t log2spp, nBase4Digits;        /* ZSobol (samplers.h:228-240) */
    int32_t x_samples, y_samples, jitter; /* Stratified (samplers.h:503-575) */
    int32_t halton_base_scales[2], halton_base_exponents[2], halton_mult_inverse[2]; /* Halton (samplers.cpp:32-52) */
    int32_t sobol_scale;                  /* SobolSampler: RoundUpPow2(max(full resolution)) (samplers.h:489) */
} wf_sampler;

enum wf_light_sampler_type { WF_LS_UNIFORM = 0, WF_LS_POWER = 1, WF_LS_BVH = 2 };

typedef struct wf_options {               /* BasicPBRTOptions subset read by kernels (options.h:22-34) */
    int32_t seed;
    int32_t disable_pixel_jitter, disable_wavelength_jitter, disable_texture_filtering;
} wf_options;

typedef struct wf_scene_desc {
    int32_t abi_version;
    /* geometry */
    int32_t n_vertices, n_triangles, n_meshes, n_bvh_nodes;
    const float *P;              /* [n_vertices][3] render space */
    const float *N;              /* [n_vertices][3] (zero for meshes without normals) */
    const float *UV;             /* [n_vertices][2] */
    const int32_t *tri_indices;  /* [n_triangles][3] global vertex ids */
    const int32_t *tri_mesh;     /* [n_triangles + n_quadrics] mesh id of every primitive */
    const wf_mesh *meshes;
    const wf_bvh_node *bvh_nodes;
    const int32_t *bvh_prims;    /* [n_triangles + n_quadrics] primitive ids in BVH leaf order */
    float scene_bounds[6];
    /* shading */
    int32_t n_spectra, n_spectrum_floats, n_textures, n_materials;
    const wf_spectrum *spectra;
    const float *spectrum_data;
    const wf_texture *textures;
    const wf_material *materials;
    /* lights */
    int32_t n_lights, n_infinite_lights, n_light_bvh_nodes, n_light_transforms;
    const wf_light *lights;
    const int32_t *infinite_lights;
    const wf_light_bvh_node *light_bvh_nodes;
    const wf_transform *light_transforms;
    float all_light_bounds[6];
    int32_t light_sampler;       /* wf_light_sampler_type */
    const float *power_alias;    /* POWER: n_lights * {q, p, alias(as int bits)} */
    /* camera / film / filter / sampler */
    wf_camera camera;
    wf_film film;
    wf_filter filter;
    int32_t n_filter_floats;
    const float *filter_data;
    wf_sampler sampler;
    /* integrator */
    int32_t max_depth;
    int32_t regularize;
    int32_t have_media;
    wf_options options;
    /* image infinite lights + the RGB -> spectrum table of their colour space (util/color.h:368-395; sRGB) */
    int32_t n_image_lights, n_table_floats;
    const wf_image_light *image_lights;
    int32_t n_tex_images;
    const wf_tex_image *tex_images;  /* wf_texture.i0 of the IMAGE texture types */
    const float *table_data;
    const int32_t *noise_perm;           /* [512] Perlin permutation (util/noise.cpp:19-56), or null when no texture / medium uses noise */
    const float *rgb2spec_coeffs;        /* [3][64][64][64][3] or null when no image light needs it */
    float rgb2spec_znodes[64];
    int32_t cs_illuminant_offset;        /* dense illuminant of that colour space in spectrum_data */
    /* participating media (media.h): ids referenced by wf_mesh.medium_inside/outside, wf_camera.medium */
    int32_t n_media, n_medium_floats;
    const struct wf_medium *media;
    const float *medium_data;    /* density / Lescale / majorant grids */
    /* HaltonSampler tables (util/primes.h, util/lowdiscrepancy.h:26-62): null unless the sampler is WF_SAMPLER_HALTON */
    /* SobolSampler tables (util/sobolmatrices.cpp; data/sobol_matrices.bin): null unless the sampler is WF_SAMPLER_SOBOL */
    const uint32_t *sobol_matrices;      /* SobolMatrices32 [1024][52] */
    const uint64_t *vdc_sobol;           /* VdCSobolMatrices [25][52] */
    const uint64_t *vdc_sobol_inv;       /* VdCSobolMatricesInv [26][52] */
    const int32_t *halton_primes;        /* [1000] */
    const int32_t *halton_perm_offsets;  /* [1000] start of dimension d's nDigits x base digit permutations */
    const uint16_t *halton_perms;
    int64_t n_halton_perms;
    /* quadrics */
    int32_t n_quadrics, pad_quadrics;
    const wf_quadric *quadrics;
    /* object instances: the top-level BVH is bvh_nodes[0 .. n_top_bvh_nodes), over bvh_prims[0 .. n_top_prims) */
    int32_t n_instances, n_instance_defs, n_top_bvh_nodes, n_top_prims;
    const wf_instance *instances;
    const wf_instance_def *instance_defs;
    /* trianglemesh "S": per-vertex shading tangents in render space, only of the meshes that have them (wf_mesh.first_s) */
    int64_t n_tangents;
    const float *S;              /* [n_tangents][3] */
    /* AnimatedPrimitive (round 5): the AnimatedTransforms of animated shapes and object instances (wf_instance.anim_plus1); a shape created
       under an animated CTM is an instance definition of its own (its shapes in object space) used once with the animated transformation */
    int32_t n_animated, pad_animated;
    const wf_animated_transform *animated;
} wf_scene_desc;

/* ------------------------------------------------------------------------------------------- */
/* Work queues: SoA views (wavefront/workitems.soa, workqueue.h)                                 */
/* Every member is its own capacity-element device array; spectra/wavelengths are float4 arrays. */
/* ------------------------------------------------------------------------------------------- */

typedef struct wf_f4 { float x, y, z, w; } wf_f4;

/* PixelSampleState, wavefront/workitems.h:107-116 (visibleSurface omitted: RGBFilm never reads it) */
typedef struct wf_pixel_state {
    int32_t capacity;
    float *filterWeight;
    int32_t *pPixel_x, *pPixel_y;
    wf_f4 *lambda, *lambda_pdf;
    wf_f4 *L;
    wf_f4 *cameraRayWeight;
    /* RaySamples, workitems.h:41-104: direct{uc,u}, indirect{uc,u,rr} */
    wf_f4 *samples0;             /* direct.uc, direct.u.x, direct.u.y, indirect.uc */
    wf_f4 *samples1;             /* indirect.u.x, indirect.u.y, indirect.rr, unused */
} wf_pixel_state;

/* RayWorkItem, workitems.h:119-130 */
typedef struct wf_ray_queue {
    int32_t capacity;
    int32_t *size;
    float *ox, *oy, *oz, *dx, *dy, *dz;
    float *time;
    int32_t *medium;
    int32_t *depth;
    int32_t *pixelIndex;
    wf_f4 *lambda, *lambda_pdf;
    wf_f4 *beta, *r_u, *r_l;
    /* prevIntrCtx: LightSampleContext {Point3fi pi; Normal3f n, ns} */
    float *ctx_pi_lo_x, *ctx_pi_lo_y, *ctx_pi_lo_z, *ctx_pi_hi_x, *ctx_pi_hi_y, *ctx_pi_hi_z;
    float *ctx_n_x, *ctx_n_y, *ctx_n_z, *ctx_ns_x, *ctx_ns_y, *ctx_ns_z;
    float *etaScale;
    int32_t *flags;              /* bit0 specularBounce, bit1 anyNonSpecularBounces */
} wf_ray_queue;

/* Result of IntersectClosest per ray, used by parity tests and by the counting variant. */
typedef struct wf_hit_record {
    int32_t prim;                /* global triangle id (or n_triangles + quadric index), -1 = no hit */
    float t, b0, b1, b2;
    int32_t nodes_visited, tris_tested;
    int32_t instance;            /* index of the object instance the hit primitive was reached through, -1 = top level */
} wf_hit_record;

typedef struct wf_ctx wf_ctx;    /* opaque: device, stream, uploaded scene, queues, film */

typedef struct wf_render_stats { /* WavefrontPathIntegrator::Stats, integrator.h:174-181 */
    uint64_t camera_rays;
    uint64_t indirect_rays[64];
    uint64_t shadow_rays[64];
} wf_render_stats;

typedef struct wf_kernel_profile_entry {
    char name[64];
    int32_t launches;
    float total_ms, min_ms, max_ms;
} wf_kernel_profile_entry;

/* ------------------------------------------------------------------------------------------- */
/* Entry points                                                                                  */
/* ------------------------------------------------------------------------------------------- */

const char *wf_last_error(void);
int wf_abi_version(void);

/* gpu/util.cpp:28-110 GPUInit + gpu/memory.cpp: context, stream, plain hipMalloc arena */
int wf_ctx_create(int device, wf_ctx **out);
int wf_ctx_destroy(wf_ctx *ctx);
int wf_sync(wf_ctx *ctx);                                   /* GPUWait(), gpu/util.h:122 */
void *wf_stream(wf_ctx *ctx);                               /* hipStream_t as void* */

/* OptiXAggregate ctor + scene managed-memory residency (gpu/optix/aggregate.cpp:1183-1668):
   uploads the flat tables; BVH is built on the host (host/bvh_build.cpp) and arrives in desc. */
int wf_scene_upload(wf_ctx *ctx, const wf_scene_desc *desc);
int wf_aggregate_bounds(wf_ctx *ctx, float out_bounds[6]);  /* WavefrontAggregate::Bounds */

/* integrator.cpp:227-274 queue sizing + allocation; film pixels (film.h:302-307).
   pixels_per_pass = the reference's maxQueueSize (width x scanlinesPerPass).  samples_per_pass = how many
   sample indices of each pixel one pass carries (queue capacity = the product): 1 reproduces the
   reference's wavefront size; larger values fill the 256 CUs with one launch per stage (the film result
   is bit-identical for any value). */
int wf_queues_alloc(wf_ctx *ctx, int pixels_per_pass, int samples_per_pass);
/* the following passes carry sample indices s, s + sample_step, ..., n_samples of them (s = the sample_index
   argument of wf_gen_camera_rays / wf_gen_ray_samples / wf_render_pass) */
int wf_set_pass_samples(wf_ctx *ctx, int sample_step, int n_samples);
int wf_film_clear(wf_ctx *ctx);

/* K1: "Reset ray queue"/"Reset queues before tracing rays"/"Reset shadowRayQueue" (integrator.cpp:357-397,581-585) */
int wf_reset_ray_queue(wf_ctx *ctx, int which);
int wf_reset_stage_queues(wf_ctx *ctx, int depth);          /* also accumulates ray stats */
/* K2: GenerateCameraRays (wavefront/camera.cpp:31-80) */
int wf_gen_camera_rays(wf_ctx *ctx, int y0, int sample_index);
/* K3: GenerateRaySamples (wavefront/samples.cpp:29-66) */
int wf_gen_ray_samples(wf_ctx *ctx, int depth, int sample_index);
/* K4: WavefrontAggregate::IntersectClosest (integrator.h:37-43) */
int wf_intersect_closest(wf_ctx *ctx, int depth);
/* K5 + K6: SampleMediumInteraction (wavefront/media.cpp:22-257) then SampleMediumScattering<HGPhaseFunction>
   (:259-352) — delta tracking through the medium of every ray with ray.medium set; no-op without media */
int wf_medium_sample(wf_ctx *ctx, int depth);
/* K7/K8: HandleEscapedRays / HandleEmissiveIntersection (integrator.cpp:495-573) */
int wf_handle_escaped(wf_ctx *ctx, int depth);
int wf_handle_emissive(wf_ctx *ctx, int depth);
/* K9: EvaluateMaterialAndBSDF<M> (wavefront/surfscatter.cpp:57-328), one launch per material type */
int wf_eval_material(wf_ctx *ctx, int material_type, int depth);
/* K10: WavefrontAggregate::IntersectShadow (integrator.h:45-46) + RecordShadowRayResult */
int wf_intersect_shadow(wf_ctx *ctx, int depth);
/* K11: WavefrontAggregate::IntersectShadowTr (integrator.h:48-49; TraceTransmittance, intersect.h:165-274): the
   shadow-ray stage of scenes with media (ratio tracking through interface surfaces) */
int wf_intersect_shadow_tr(wf_ctx *ctx, int depth);
/* K12: SampleSubsurface (wavefront/subsurface.cpp:18-203), in its three launches; no-ops without a subsurface material.
   wf_subsurface_probe      "Get BSSRDF and enqueue probe ray": SubsurfaceMaterial::GetBSSRDF + TabulatedBSSRDF::SampleSp
   wf_intersect_one_random  WavefrontAggregate::IntersectOneRandom (integrator.h:51-52; wavefront/aggregate.cpp:90-115): every
                            intersection of the probe segment with a surface of the SAME material, one of them kept by weighted
                            reservoir sampling seeded with Hash(p0, p1)
   wf_subsurface_scatter    "Handle out-scattering after SSS": ProbeIntersectionToSample, then the indirect ray and the light
                            sample at the exit point; the caller traces the shadow rays (wf_intersect_shadow / _tr) afterwards */
int wf_subsurface_probe(wf_ctx *ctx, int depth);
int wf_intersect_one_random(wf_ctx *ctx);
int wf_subsurface_scatter(wf_ctx *ctx, int depth);
/* K13: UpdateFilm (wavefront/film.cpp:14-38) */
int wf_update_film(wf_ctx *ctx);

/* whole wavefront pass for one sample index over scanlines [y0, y0+scanlinesPerPass): the sequence
   integrator.cpp:357-434, enqueued without host synchronisation. */
int wf_render_pass(wf_ctx *ctx, int y0, int sample_index);

/* Image partition for multi-GPU rendering (the north_star's "image tiled across the GPUs"): this context renders the
   scanline strips rank, rank + count, rank + 2 count, ... of `height` lines each (interleaved: sky and foliage are dealt evenly).
   wf_render_pass's y0 then counts LOCAL lines (pixel_min.y + first local line of the band); *local_rows = lines owned.  Every
   context keeps a full-size film whose foreign lines stay zero, so summing the films (one reduce to rank 0) is a gather and the
   result is bit-identical to a single-context render.  count = 1 restores the whole image. */
int wf_set_strips(wf_ctx *ctx, int rank, int count, int height, int *local_rows);

/* results */
int wf_film_download(wf_ctx *ctx, double *rgb_sum_weight /* [H][W][4] */);
/* SpectralFilm only: the spectral accumulators, per pixel n_buckets bucketSums followed by n_buckets weightSums (doubles) */
int wf_film_spectral_download(wf_ctx *ctx, double *dst /* [H][W][2 * n_buckets] */);
/* GBufferFilm only: the per-pixel geometry / albedo / variance accumulators */
int wf_film_gbuffer_download(wf_ctx *ctx, wf_gbuffer_pixel *dst /* [H][W] */);
int wf_film_device_ptr(wf_ctx *ctx, void **dptr, uint64_t *nbytes); /* for the RCCL film reduce */
int wf_film_upload(wf_ctx *ctx, const double *rgb_sum_weight);
int wf_film_copy_to_device(wf_ctx *ctx, void *dst_device);         /* D2D, wf_film_device_ptr's size */
/* The gather of a strip-partitioned render inside one process (pbrt_amd --gpus N: one context and one host thread per device,
   SURVEY 8(b) / 8(e)): copies the scanline strips `src` owns (wf_set_strips) from its film into `dst`'s — hipMemcpyPeerAsync over xGMI
   between devices, device-to-device on one device; 1/N of the film per rank.  Both contexts hold the same scene and are idle.
   wf_stats_add: src's ray counters added to dst's. */
int wf_film_gather_strips(wf_ctx *dst, wf_ctx *src);
int wf_stats_add(wf_ctx *dst, wf_ctx *src);
int wf_film_copy_from_device(wf_ctx *ctx, const void *src_device);
int wf_stats_download(wf_ctx *ctx, wf_render_stats *out);
int wf_profile_report(wf_ctx *ctx, wf_kernel_profile_entry *entries, int max_entries, int *n_out);
/* per-launch hipEvent pairs on the context's stream (gpu/util.cpp:136-209): 0 off, 1 every launch,
   2 only the launches of the traversal stages ("Intersect closest", "Route hits", "Intersect closest: near-tie re-trace",
   "Intersect shadow") */
int wf_profile_enable(wf_ctx *ctx, int enabled);
/* sum of the recorded durations of the named kernel since the last wf_profile_report */
int wf_kernel_time_ms(wf_ctx *ctx, const char *name, double *total_ms, int *launches);

/* Stand-alone traversal entry points used by parity tests and the roofline counters:
   rays given as host arrays (o[3], d[3], tMax), results as wf_hit_record / occluded flags. */
int wf_trace_closest_host(wf_ctx *ctx, int n, const float *o, const float *d, const float *tmax,
                          wf_hit_record *out, int count_visits);
int wf_trace_any_host(wf_ctx *ctx, int n, const float *o, const float *d, const float *tmax,
                      int32_t *occluded, int32_t *nodes_visited, int32_t *tris_tested);
/* ... with the rays' TIMES (round 6): the reference's WavefrontAggregate reads ray.time (integrator.h:32-54) and an AnimatedPrimitive
   (cpu/primitive.cpp:132-158) is intersected with its transformation interpolated at that time.  On a scene with animated primitives
   the untimed calls above and below (and wf_trace_shadow_tr_host / wf_trace_one_random_host) return an error instead of answering
   for the start-time geometry; these two walk in the reference's order (cpu/aggregates.cpp:529-579) at time[i].  Any scene. */
int wf_trace_closest_host_t(wf_ctx *ctx, int n, const float *o, const float *d, const float *tmax, const float *time, wf_hit_record *out);
int wf_trace_any_host_t(wf_ctx *ctx, int n, const float *o, const float *d, const float *tmax, const float *time, int32_t *occluded);
/* The same two calls on caller-owned DEVICE buffers (what a GPU-resident host integrator binds: its RayQueue / ShadowRayQueue stay on
   the device): rays7 = n x {o[3], d[3], tMax} floats, out = n records / n flags, all device pointers of the context's
   device; the launches go to the context's stream (wf_stream) and are NOT synchronised — order them with the stream.  The production
   traversal (near ties resolved as in the render).  wf_device_alloc / free / upload / download are the plain allocation and copy calls
   of that device for callers without a HIP runtime of their own (uploads and downloads are synchronous). */
int wf_trace_closest_device(wf_ctx *ctx, int n, const float *rays7, wf_hit_record *out);
int wf_trace_any_device(wf_ctx *ctx, int n, const float *rays7, int32_t *occluded);
int wf_device_alloc(wf_ctx *ctx, uint64_t nbytes, void **dptr);
int wf_device_free(wf_ctx *ctx, void *dptr);
int wf_device_upload(wf_ctx *ctx, void *dst_device, const void *src_host, uint64_t nbytes);
int wf_device_download(wf_ctx *ctx, void *dst_host, const void *src_device, uint64_t nbytes);
/* WavefrontAggregate::IntersectShadowTr (integrator.h:48-50; TraceTransmittance, wavefront/intersect.h:165-274) on caller-supplied
   shadow rays of a scene with media: per ray o[3], d[3], tmax, the medium id of ray.medium (-1: none), the four wavelengths, and the
   item's Ld / r_u / r_l (4 floats each).  out_L[4 i ..] = Ld * T_ray / (r_u * r_u' + r_l * r_l').Average(), the value
   RecordShadowRayResult's caller adds to the pixel (zero when the ray is blocked or the transmittance roulette ends it). */
int wf_trace_shadow_tr_host(wf_ctx *ctx, int n, const float *o, const float *d, const float *tmax, const int32_t *medium, const float *lambda,
                            const float *Ld, const float *r_u, const float *r_l, float *out_L);
/* WavefrontAggregate::IntersectOneRandom (integrator.h:51-52) on caller-supplied probe segments p0 -> p1 (3 floats each): out[i] =
   the hit of a surface whose material id is material[i], chosen by the reference's weighted reservoir sampling (seed Hash(p0, p1))
   among all such hits along the segment, reservoir_pdf[i] its sample probability (0 and prim = -1: none) */
int wf_trace_one_random_host(wf_ctx *ctx, int n, const float *p0, const float *p1, const int32_t *material, wf_hit_record *out, float *reservoir_pdf);
/* Device part of the HLBVH build (cpu/aggregates.cpp:394-411): Morton codes (10 bits per axis of each centroid's
   offset inside `bounds` = min.xyz max.xyz) and their stable radix sort.  codes[i] / order[i] = code and input position of the i-th
   primitive in Morton order; the host builder emits the treelets and the SAH upper tree from them.  Needs no context; non-zero
   when no device is visible. */
int wf_morton_sort(int n, const float *centroids, const float bounds[6], uint32_t *codes, uint32_t *order);
/* The SAH build on the device (BVHAggregate::buildRecursive + flattenBVH, cpu/aggregates.cpp:198-387, 505-521): from the
   bounds of n primitives (min.xyz max.xyz each, input order) the reference's tree NODE FOR NODE — nodes_out[0 .. *n_nodes_out) in the
   reference's depth-first LinearBVHNode layout (leaf offsets = positions in order_out, interior offsets = second-child indices, both
   relative to this tree), order_out[n] = the input position of the primitive at each leaf slot (std::partition's order).  nodes_out needs
   room for 2 n - 1 records.  Needs no context; -1 when no device is visible, another negative value when the build gave up (degenerate
   input: the caller falls back to the host builder, csrc/host/bvh_build.cpp). */
int wf_build_bvh_sah(int n, const float *bounds, int max_prims_in_node, wf_bvh_node *nodes_out, int32_t *order_out, int32_t *n_nodes_out);
/* Sampler probe for parity tests: fills out[n][dims] with the sampler's values for pixel/sample (Get1D() from start_dim).
   ndims = -2: GetPixel2D() (2 floats per record); -3: ten times (Get2D, Get1D), the sequence of the reference's Sampler.ConsistentValues
   test (30 floats); -4: the ZSobol sample index at start_dim (ZSobolSampler.ValidIndices): its low / high 32 bits as two float bit patterns */
int wf_sampler_probe(wf_ctx *ctx, int n, const int32_t *px, const int32_t *py, const int32_t *sample_index,
                     int start_dim, int ndims, float *out);
/* Elementary-function probe for parity tests: out[i] = f(in[i]) evaluated on the device with the kernels' own
   routines (csrc/common/wf_libm.h, the restatement of the glibc 2.35 float libm the reference is linked against).
   fn: 0 sin, 1 cos, 2 exp, 3 log, 4 atan, 5 asin, 6 acos, 7 cosh, 8 atanh, 9 atan2 (in = n (y, x) pairs).
   Needs a context only (no scene). */
int wf_libm_probe(wf_ctx *ctx, int fn, int n, const float *in, float *out);
/* Known-answer probe for parity tests (csrc/common/wf_kat.h): n records of 16 uint64 in, n records of 8 uint64 out — the restated PCG32
   (util/rng.h: SetSequence / Advance / Uniform / operator-), MurmurHash64A / Hash / HashFloat / MixBits (util/hash.h) and IntersectTriangle
   (shapes.cpp:168-269) evaluated on the device; the reference's answers are tests/golden/kat_out.bin (oracle/ref_build/ref_kat.cpp).
   Needs a context only (no scene). */
int wf_kat_probe(wf_ctx *ctx, int n, const uint64_t *in, uint64_t *out);
/* Items evaluated by the material stage since the last wf_film_clear, per material type (out[wf_material_type], < WF_MAT_NTYPES), and
   the items of the medium-sample stage (out[WF_MAT_NTYPES]); out[12..15] = 0.  What bench.py's roofline_material / roofline_medium
   divide the SURVEY 8(d) bytes per item by. */
int wf_material_items_download(wf_ctx *ctx, uint64_t out[16]);
/* debug/parity access to queues: downloads the named SoA member */
int wf_queue_size(wf_ctx *ctx, const char *queue, int *size);
int wf_queue_download(wf_ctx *ctx, const char *queue, const char *member, void *dst, uint64_t nbytes);
/* algorithmic-byte counters accumulated by the traversal kernels when enabled */
typedef struct wf_traversal_counters {
    uint64_t closest_rays, closest_nodes, closest_tris, closest_hits;
    uint64_t shadow_rays, shadow_nodes, shadow_tris, shadow_unoccluded;
} wf_traversal_counters;
int wf_counters_enable(wf_ctx *ctx, int enabled);
int wf_counters_download(wf_ctx *ctx, wf_traversal_counters *out);
/* which rare paths of the production traversal ran since the last reset (the parity tests assert that a big-tree scene takes them):
   out[0] node-stack entries that left the LDS ring for the lane's HBM column, out[1] != 0: a push ran past the HBM column (the
   entry was dropped; wf_sync returns an error as well), out[2] near-tie rays re-walked in reference order inside the walk kernel,
   out[3] != 0: a launch dealt its rays through the shared cursor.  reset != 0 zeroes the words after reading. */
int wf_debug_counters(wf_ctx *ctx, uint64_t out[4], int reset);
/* Which kernel variants the uploaded scene runs (tests assert that a scene takes the path they mean to cover): key = "fast_ok" (the
   production traversal layout is in use), "gen_mode" (0 - 3: strength of the walk kernels), "gen_tri", "defer_general" (the two-class
   traversal), "anim_fast" (AnimatedPrimitives on the production walk), "lean_shade", "lean_type_<material type>", "rare_lights", "medium_lean" (the lean delta-tracking / transmittance kernels),
   "instances".  Introspection only; nothing in the reference corresponds. */
int wf_ctx_query(wf_ctx *ctx, const char *key, int64_t *value);
/* Host-only self-check of the production traversal layout (no device needed; CPU suite, tests/test_fastbvh_host.py): builds the
   QNode / LeafTri / instance-entry arrays wf_scene_upload would upload for `d` (round 6: with the top-level tree rebuilt over
   partially re-braided instances, WF_BRAID) and walks them on the host with n_rays random rays in double arithmetic, without
   pruning by distance.  out[0] QNodes, out[1] LeafTri records, out[2] instance entries, out[3] nodes visited by the rays,
   out[4] (ray, triangle) pairs that really intersect (brute force over top-level triangles and every (instance, triangle) pair),
   out[5] of those NOT among the triangles the walk tested (must be 0: the tree owes a superset), out[6] triangles tested,
   out[7] instance entries taken.  Replaces nothing in the reference (its accelerator is OptiX's, gpu/optix/aggregate.cpp). */
int wf_debug_fastbvh_check(const wf_scene_desc *d, int n_rays, uint64_t seed, int64_t out[8]);

#ifdef __cplusplus
}
#endif
#endif /* WF_ABI_H */
