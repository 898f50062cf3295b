// oracle/ref_build/ref_probe.cpp — TEST INFRASTRUCTURE ONLY (ours; links the shimmed build of the
// reference, libpbrt_ref.a).  Calls the reference's own leaf functions on seeded inputs and writes golden
// vectors:  ref_probe <outdir>  ->  <outdir>/<name>_in.bin, <outdir>/<name>_out.bin  (raw little-endian
// float32 / int32 records, layouts below).  tests/golden/ holds the committed result of running this in
// the build container (script: tools/make_golden.sh); oracle/wf_cpu/wf_probe.cpp evaluates the restated
// functions on the same *_in.bin.
#include <pbrt/pbrt.h>

#include <pbrt/bxdfs.h>
#include <pbrt/options.h>
#include <pbrt/samplers.h>
#include <pbrt/shapes.h>
#include <pbrt/util/math.h>
#include <pbrt/util/rng.h>
#include <pbrt/util/sampling.h>
#include <pbrt/util/scattering.h>
#include <pbrt/util/spectrum.h>
#include <pbrt/util/transform.h>
#include <pbrt/interaction.h>

#include <cstdio>
#include <string>
#include <vector>

using namespace pbrt;

static void writeBin(const std::string &path, const void *p, size_t bytes) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(1); }
    fwrite(p, 1, bytes, f);
    fclose(f);
}

struct Lcg {  // input generator shared with nothing: inputs are stored, not regenerated
    uint64_t s;
    explicit Lcg(uint64_t seed) : s(seed * 2862933555777941757ull + 3037000493ull) {}
    uint32_t u32() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 32); }
    float f01() { return (u32() >> 8) * (1.f / 16777216.f); }
    float range(float a, float b) { return a + (b - a) * f01(); }
};

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: ref_probe <outdir>\n"); return 1; }
    std::string dir = argv[1];
    Options = new PBRTOptions;  // GetOptions().seed == 0 (LayeredBxDF seeds its RNG from it, bxdfs.h:510)

    // ---- zsobol: in {px, py, sampleIndex} int32; out 12 floats: Get1D x 12 from dimension 0
    //      sampler = ZSobolSampler(16 spp, 400x400, FastOwen, seed 0)  (scenes/cornell-box.pbrt)
    {
        const int n = 1024, nd = 12;
        Lcg g(1);
        std::vector<int32_t> in(3 * n);
        std::vector<float> out((size_t)n * nd);
        for (int i = 0; i < n; ++i) {
            in[3 * i] = g.u32() % 400; in[3 * i + 1] = g.u32() % 400; in[3 * i + 2] = g.u32() % 16;
            ZSobolSampler s(16, Point2i(400, 400), RandomizeStrategy::FastOwen, 0);
            s.StartPixelSample(Point2i(in[3 * i], in[3 * i + 1]), in[3 * i + 2], 0);
            for (int d = 0; d < nd; ++d) out[(size_t)i * nd + d] = s.Get1D();
        }
        writeBin(dir + "/zsobol_in.bin", in.data(), in.size() * 4);
        writeBin(dir + "/zsobol_out.bin", out.data(), out.size() * 4);
    }
    // ---- zsobol2: 1920x1080, 64 spp, Get2D pairs at dimension 6+7*depth
    {
        const int n = 1024, nd = 8;
        Lcg g(2);
        std::vector<int32_t> in(3 * n);
        std::vector<float> out((size_t)n * nd);
        for (int i = 0; i < n; ++i) {
            in[3 * i] = g.u32() % 1920; in[3 * i + 1] = g.u32() % 1080; in[3 * i + 2] = g.u32() % 64;
            ZSobolSampler s(64, Point2i(1920, 1080), RandomizeStrategy::FastOwen, 0);
            s.StartPixelSample(Point2i(in[3 * i], in[3 * i + 1]), in[3 * i + 2], 13);
            for (int d = 0; d < nd; ++d) out[(size_t)i * nd + d] = s.Get1D();
        }
        writeBin(dir + "/zsobol2_in.bin", in.data(), in.size() * 4);
        writeBin(dir + "/zsobol2_out.bin", out.data(), out.size() * 4);
    }
    // ---- triangle: in {o[3], d[3], tMax, p0[3], p1[3], p2[3]} 16 floats; out {hit, b0, b1, b2, t} 5 floats
    {
        const int n = 6000;
        Lcg g(3);
        std::vector<float> in((size_t)n * 16), out((size_t)n * 5);
        for (int i = 0; i < n; ++i) {
            float *r = &in[(size_t)i * 16];
            Point3f p[3];
            for (int k = 0; k < 3; ++k) p[k] = Point3f(g.range(-5, 5), g.range(-5, 5), g.range(-5, 5));
            Point3f o(g.range(-10, 10), g.range(-10, 10), g.range(-10, 10));
            // aim at a point in (or near, or on an edge of) the triangle
            float b0 = g.range(-0.2f, 1.2f), b1 = g.range(-0.2f, 1.2f);
            if (i % 7 == 0) b0 = 0;           // edge
            if (i % 11 == 0) { b0 = 1; b1 = 0; }  // vertex
            Point3f target = b0 * p[0] + b1 * p[1] + (1 - b0 - b1) * p[2];
            Vector3f d = target - o;
            if (i % 3 == 0) d = Normalize(d);
            float tMax = (i % 5 == 0) ? g.range(0.2f, 1.5f) : Infinity;
            r[0] = o.x; r[1] = o.y; r[2] = o.z; r[3] = d.x; r[4] = d.y; r[5] = d.z; r[6] = tMax;
            for (int k = 0; k < 3; ++k) { r[7 + 3 * k] = p[k].x; r[8 + 3 * k] = p[k].y; r[9 + 3 * k] = p[k].z; }
            Ray ray(o, d);
            pstd::optional<TriangleIntersection> ti = IntersectTriangle(ray, tMax, p[0], p[1], p[2]);
            float *w = &out[(size_t)i * 5];
            w[0] = ti ? 1.f : 0.f;
            w[1] = ti ? ti->b0 : 0; w[2] = ti ? ti->b1 : 0; w[3] = ti ? ti->b2 : 0; w[4] = ti ? ti->t : 0;
        }
        writeBin(dir + "/triangle_in.bin", in.data(), in.size() * 4);
        writeBin(dir + "/triangle_out.bin", out.data(), out.size() * 4);
    }
    // ---- spherical triangle sampling: in {v0[3], v1[3], v2[3], p[3], u[2]} 14 floats;
    //      out {b0, b1, b2, pdf, invU0, invU1} 6 floats (InvertSphericalTriangleSample of the sampled direction)
    {
        const int n = 3000;
        Lcg g(4);
        std::vector<float> in((size_t)n * 14), out((size_t)n * 6);
        for (int i = 0; i < n; ++i) {
            float *r = &in[(size_t)i * 14];
            Point3f v[3];
            for (int k = 0; k < 3; ++k) v[k] = Point3f(g.range(-2, 2), g.range(-2, 2), g.range(-2, 2));
            Point3f p(g.range(-4, 4), g.range(-4, 4), g.range(-4, 4));
            Point2f u(g.f01(), g.f01());
            for (int k = 0; k < 3; ++k) { r[3 * k] = v[k].x; r[3 * k + 1] = v[k].y; r[3 * k + 2] = v[k].z; }
            r[9] = p.x; r[10] = p.y; r[11] = p.z; r[12] = u.x; r[13] = u.y;
            Float pdf;
            pstd::array<Float, 3> b = SampleSphericalTriangle({v[0], v[1], v[2]}, p, u, &pdf);
            float *w = &out[(size_t)i * 6];
            w[0] = b[0]; w[1] = b[1]; w[2] = b[2]; w[3] = pdf;
            Point3f ps = b[0] * v[0] + b[1] * v[1] + b[2] * v[2];
            Vector3f wdir = ps - p;
            Point2f iu(0, 0);
            if (pdf > 0 && LengthSquared(wdir) > 0) iu = InvertSphericalTriangleSample({v[0], v[1], v[2]}, p, Normalize(wdir));
            w[4] = iu.x; w[5] = iu.y;
        }
        writeBin(dir + "/sphtri_in.bin", in.data(), in.size() * 4);
        writeBin(dir + "/sphtri_out.bin", out.data(), out.size() * 4);
    }
    // ---- bxdfs: in {type, wo[3], wi[3], uc, u[2], eta, ax, ay, k} 13 floats (type 0 diffuse R=0.5, 1 dielectric,
    //      2 conductor eta=(eta,..) k=(k,..), 3 thin dielectric, 4 diffuse transmission R=.25 T=.5, 5 coated diffuse,
//      6 coated conductor; GetOptions().seed == 0)
    //      out {f[4](wo,wi), pdf(wo,wi), sample valid, sample f[4], sample wi[3], sample pdf, flags, eta} 16 floats
    {
        const int n = 7000;
        Lcg g(5);
        std::vector<float> in((size_t)n * 13), out((size_t)n * 16);
        for (int i = 0; i < n; ++i) {
            float *r = &in[(size_t)i * 13];
            int type = i % 7;
            Vector3f wo = Normalize(Vector3f(g.range(-1, 1), g.range(-1, 1), g.range(-1, 1)));
            Vector3f wi = Normalize(Vector3f(g.range(-1, 1), g.range(-1, 1), g.range(-1, 1)));
            float uc = g.f01();
            Point2f u(g.f01(), g.f01());
            float eta = g.range(1.0f, 2.5f);
            if (i % 13 == 0) eta = 1;
            float ax = (i % 4 == 0) ? 0.f : g.range(0.001f, 0.9f), ay = (i % 4 == 0) ? 0.f : ((i % 3 == 0) ? ax : g.range(0.001f, 0.9f));
            float kk = g.range(0.5f, 4.f);
            r[0] = (float)type; r[1] = wo.x; r[2] = wo.y; r[3] = wo.z; r[4] = wi.x; r[5] = wi.y; r[6] = wi.z;
            r[7] = uc; r[8] = u.x; r[9] = u.y; r[10] = eta; r[11] = ax; r[12] = ay;
            // conductor k rides in place of nothing: derive deterministically from eta so the record stays 13 floats
            kk = 2.f * eta;
            float *w = &out[(size_t)i * 16];
            SampledSpectrum f(0.f);
            Float pdf = 0;
            pstd::optional<BSDFSample> bs;
            TrowbridgeReitzDistribution distrib(ax, ay);
            if (type == 0) {
                DiffuseBxDF b(SampledSpectrum(0.5f));
                f = b.f(wo, wi, TransportMode::Radiance); pdf = b.PDF(wo, wi, TransportMode::Radiance); bs = b.Sample_f(wo, uc, u, TransportMode::Radiance);
            } else if (type == 1) {
                DielectricBxDF b(eta, distrib);
                f = b.f(wo, wi, TransportMode::Radiance); pdf = b.PDF(wo, wi, TransportMode::Radiance); bs = b.Sample_f(wo, uc, u, TransportMode::Radiance);
            } else if (type == 2) {
                ConductorBxDF b(distrib, SampledSpectrum(eta), SampledSpectrum(kk));
                f = b.f(wo, wi, TransportMode::Radiance); pdf = b.PDF(wo, wi, TransportMode::Radiance, BxDFReflTransFlags::All);
                bs = b.Sample_f(wo, uc, u, TransportMode::Radiance);
            } else if (type == 3) {
                ThinDielectricBxDF b(eta);
                f = b.f(wo, wi, TransportMode::Radiance); pdf = b.PDF(wo, wi, TransportMode::Radiance, BxDFReflTransFlags::All);
                bs = b.Sample_f(wo, uc, u, TransportMode::Radiance, BxDFReflTransFlags::All);
            } else if (type == 4) {
                DiffuseTransmissionBxDF b(SampledSpectrum(0.25f), SampledSpectrum(0.5f));
                f = b.f(wo, wi, TransportMode::Radiance); pdf = b.PDF(wo, wi, TransportMode::Radiance); bs = b.Sample_f(wo, uc, u, TransportMode::Radiance);
            } else if (type == 5) {
                // coated diffuse: thickness 0.01, albedo 0 or 0.3 (every other record), g = 0.2, maxDepth 10, nSamples 1 or 2
                CoatedDiffuseBxDF b(DielectricBxDF(eta, distrib), DiffuseBxDF(SampledSpectrum(0.5f)), 0.01f,
                                    SampledSpectrum((i / 7) % 2 ? 0.3f : 0.f), 0.2f, 10, 1 + (i / 14) % 2);
                f = b.f(wo, wi, TransportMode::Radiance); pdf = b.PDF(wo, wi, TransportMode::Radiance); bs = b.Sample_f(wo, uc, u, TransportMode::Radiance);
            } else {
                CoatedConductorBxDF b(DielectricBxDF(eta, distrib), ConductorBxDF(TrowbridgeReitzDistribution(ay, ax), SampledSpectrum(eta), SampledSpectrum(kk)),
                                      0.01f, SampledSpectrum((i / 7) % 2 ? 0.3f : 0.f), 0.2f, 10, 1 + (i / 14) % 2);
                f = b.f(wo, wi, TransportMode::Radiance); pdf = b.PDF(wo, wi, TransportMode::Radiance); bs = b.Sample_f(wo, uc, u, TransportMode::Radiance);
            }
            for (int c = 0; c < 4; ++c) w[c] = f[c];
            w[4] = pdf;
            w[5] = bs ? 1.f : 0.f;
            for (int c = 0; c < 4; ++c) w[6 + c] = bs ? bs->f[c] : 0.f;
            w[10] = bs ? bs->wi.x : 0; w[11] = bs ? bs->wi.y : 0; w[12] = bs ? bs->wi.z : 0;
            w[13] = bs ? bs->pdf : 0;
            w[14] = bs ? (float)(int)bs->flags : 0;
            w[15] = bs ? (bs->eta + (bs->pdfIsProportional ? 100.f : 0.f)) : 0;
        }
        writeBin(dir + "/bxdf_in.bin", in.data(), in.size() * 4);
        writeBin(dir + "/bxdf_out.bin", out.data(), out.size() * 4);
    }
    // ---- scalar math: in {x} ; out {FastExp(-|x|*20), SampleVisibleWavelengths(frac), VisibleWavelengthsPDF(360+470*frac),
    //      Blackbody(360+470*frac, 2000+4000*frac)}
    {
        const int n = 2048;
        Lcg g(6);
        std::vector<float> in(n), out((size_t)n * 4);
        for (int i = 0; i < n; ++i) {
            float x = g.f01();
            in[i] = x;
            out[4 * i] = FastExp(-x * 20);
            out[4 * i + 1] = SampleVisibleWavelengths(x);
            out[4 * i + 2] = VisibleWavelengthsPDF(360 + 470 * x);
            out[4 * i + 3] = Blackbody(360 + 470 * x, 2000 + 4000 * x);
        }
        writeBin(dir + "/scalar_in.bin", in.data(), in.size() * 4);
        writeBin(dir + "/scalar_out.bin", out.data(), out.size() * 4);
    }
    // ---- instance: what TransformedPrimitive::Intersect (cpu/primitive.cpp:112-125) does to a ray and to the interaction.
    //      in (32 floats): translate[3], rotate angle + axis[3], scale[3], ray o[3] d[3] tMax, interaction p[3] pError[3]
    //      n[3] dpdu[3] wo[3];   out (60): m[16], mInv[16], ApplyInverse(ray): o[3] d[3] tMax, transformed interaction:
    //      pi lo[3] hi[3], n[3], dpdu[3], shading.n[3] (FaceForward applied), wo[3]
    {
        const int n = 2048, wi = 32, wo_ = 60;
        Lcg g(7);
        std::vector<float> in((size_t)n * wi), out((size_t)n * wo_);
        for (int i = 0; i < n; ++i) {
            float *r = &in[(size_t)i * wi], *w = &out[(size_t)i * wo_];
            for (int k = 0; k < 3; ++k) r[k] = g.range(-5, 5);
            r[3] = g.range(-180, 180);
            for (int k = 0; k < 3; ++k) r[4 + k] = g.range(-1, 1);
            for (int k = 0; k < 3; ++k) r[7 + k] = (i % 5 == 0 ? -1.f : 1.f) * g.range(0.3f, 2.5f);
            if (i % 7 == 0) { r[3] = 0; r[7] = r[8] = r[9] = 1; }  // pure translation
            for (int k = 0; k < 3; ++k) r[10 + k] = g.range(-8, 8);
            Vector3f d = Normalize(Vector3f(g.range(-1, 1), g.range(-1, 1), g.range(-1, 1)));
            r[13] = d.x; r[14] = d.y; r[15] = d.z;
            r[16] = (i % 3 == 0) ? Infinity : g.range(0.1f, 20);
            for (int k = 0; k < 3; ++k) r[17 + k] = g.range(-3, 3);
            for (int k = 0; k < 3; ++k) r[20 + k] = (i % 4 == 0) ? 0.f : g.range(0, 1e-5f);
            Vector3f nn = Normalize(Vector3f(g.range(-1, 1), g.range(-1, 1), g.range(-1, 1)));
            r[23] = nn.x; r[24] = nn.y; r[25] = nn.z;
            for (int k = 0; k < 3; ++k) r[26 + k] = g.range(-2, 2);
            Vector3f wv = Normalize(Vector3f(g.range(-1, 1), g.range(-1, 1), g.range(-1, 1)));
            r[29] = wv.x; r[30] = wv.y; r[31] = wv.z;
            Transform t = Translate(Vector3f(r[0], r[1], r[2])) * Rotate(r[3], Vector3f(r[4], r[5], r[6])) * Scale(r[7], r[8], r[9]);
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b) { w[4 * a + b] = t.GetMatrix()[a][b]; w[16 + 4 * a + b] = t.GetInverseMatrix()[a][b]; }
            Float tMax = r[16];
            Ray ray = t.ApplyInverse(Ray(Point3f(r[10], r[11], r[12]), Vector3f(r[13], r[14], r[15])), &tMax);
            w[32] = ray.o.x; w[33] = ray.o.y; w[34] = ray.o.z; w[35] = ray.d.x; w[36] = ray.d.y; w[37] = ray.d.z; w[38] = tMax;
            Point3fi pi(Point3f(r[17], r[18], r[19]), Vector3f(r[20], r[21], r[22]));
            Normal3f ng(r[23], r[24], r[25]);
            Vector3f dpdu(r[26], r[27], r[28]), dpdv = Cross(Vector3f(ng), dpdu);
            SurfaceInteraction si(pi, Point2f(0.25f, 0.5f), Vector3f(r[29], r[30], r[31]), dpdu, dpdv, Normal3f(0, 0, 0), Normal3f(0, 0, 0), 0.f, false);
            si.n = ng;
            si.shading.n = -ng;  // so that the FaceForward at the end of the transform acts
            SurfaceInteraction ts = t(si);
            w[39] = ts.pi.x.LowerBound(); w[40] = ts.pi.y.LowerBound(); w[41] = ts.pi.z.LowerBound();
            w[42] = ts.pi.x.UpperBound(); w[43] = ts.pi.y.UpperBound(); w[44] = ts.pi.z.UpperBound();
            w[45] = ts.n.x; w[46] = ts.n.y; w[47] = ts.n.z;
            w[48] = ts.dpdu.x; w[49] = ts.dpdu.y; w[50] = ts.dpdu.z;
            w[51] = ts.shading.n.x; w[52] = ts.shading.n.y; w[53] = ts.shading.n.z;
            w[54] = ts.wo.x; w[55] = ts.wo.y; w[56] = ts.wo.z;
            w[57] = w[58] = w[59] = 0;
        }
        writeBin(dir + "/instance_in.bin", in.data(), in.size() * 4);
        writeBin(dir + "/instance_out.bin", out.data(), out.size() * 4);
    }
    return 0;
}
