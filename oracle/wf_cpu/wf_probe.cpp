// oracle/wf_cpu/wf_probe.cpp — TEST INFRASTRUCTURE ONLY.  Evaluates the restated leaf functions of
// pbrt-v4_amd/csrc/common/wf_*.h (host compilation) on the inputs recorded by oracle/ref_build/ref_probe
// (tests/golden/*_in.bin) and writes outputs in the same record layouts, so that tests/ can pin the
// restatement against the reference's own results (tests/golden/*_out.bin).
//   wf_probe <golden_dir> <out_dir>
#include "../../pbrt-v4_amd/csrc/common/wf_kernels.h"

#include <cstdio>
#include <string>
#include <vector>

using namespace wf;

template <typename T>
static std::vector<T> readBin(const std::string &path) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) { perror(path.c_str()); exit(1); }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    std::vector<T> v(sz / sizeof(T));
    if (fread(v.data(), sizeof(T), v.size(), f) != v.size()) exit(1);
    fclose(f);
    return v;
}
static void writeBin(const std::string &path, const void *p, size_t bytes) {
    FILE *f = fopen(path.c_str(), "wb");
    if (!f) { perror(path.c_str()); exit(1); }
    fwrite(p, 1, bytes, f);
    fclose(f);
}

template <typename B>
static void evalBxDF(const B &b, V3 wo, V3 wi, float uc, V2 u, float *w) {
    S4 f = b.f(wo, wi, MODE_RADIANCE);
    float pdf = b.PDF(wo, wi, MODE_RADIANCE);
    BSDFSample bs = b.Sample_f(wo, uc, u, MODE_RADIANCE);
    for (int c = 0; c < 4; ++c) w[c] = f[c];
    w[4] = pdf;
    w[5] = bs.valid ? 1.f : 0.f;
    for (int c = 0; c < 4; ++c) w[6 + c] = bs.valid ? bs.f[c] : 0.f;
    w[10] = bs.valid ? bs.wi.x : 0; w[11] = bs.valid ? bs.wi.y : 0; w[12] = bs.valid ? bs.wi.z : 0;
    w[13] = bs.valid ? bs.pdf : 0;
    w[14] = bs.valid ? (float)bs.flags : 0;
    w[15] = bs.valid ? (bs.eta + (bs.pdfIsProportional ? 100.f : 0.f)) : 0;
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: wf_probe <golden_dir> <out_dir>\n"); return 1; }
    std::string gd = argv[1], od = argv[2];
    static uint32_t sobol[WF_SOBOL_WORDS];
    FillSobol2D(sobol);
    {
        // the sampler's exact integer shortcuts against their defining expressions (util/lowdiscrepancy.h:165-180,
        // samplers.h:331): every value of the reduced modulo's domain, random operands, and operands above 2^32
        int bad = 0;
        uint64_t st = 0x9E3779B97F4A7C15ull;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
        for (long i = 0; i < 4000000; ++i) {
            uint64_t hd = rnd() >> (i % 3 == 0 ? 20 : 32 + (i % 29));
            uint32_t dm = 0x55555555u * (uint32_t)(i % 97);
            int want = (int)((MixBits(hd ^ dm) >> 24) % 24);
            if (ZSobol::PermutationIndex(hd, dm) != want) ++bad;
        }
        for (long i = 0; i < 2000000; ++i) {
            uint64_t a = rnd() >> (i % 2 ? 32 : 12 + (i % 20));
            for (int dim = 0; dim < 2; ++dim) {
                uint32_t v = 0;
                uint64_t t = a;
                for (int c = dim * 52; t != 0; t >>= 1, c++)
                    if (t & 1) v ^= sobol[c];
                float want = fmin(v * 0x1p-32f, OneMinusEpsilon);
                if (SobolSample(sobol, (int64_t)a, dim, WF_RAND_NONE, 0) != want) ++bad;
            }
        }
        if (bad) { fprintf(stderr, "wf_probe: sampler identity check failed (%d mismatches)\n", bad); return 2; }
    }
    auto sampler = [&](const char *name, int spp, int rx, int ry, int startDim, int nd) {
        std::vector<int32_t> in = readBin<int32_t>(gd + "/" + name + "_in.bin");
        int n = (int)in.size() / 3;
        SceneView sv{};
        sv.sobol = sobol;
        sv.sampler.type = WF_SAMPLER_ZSOBOL; sv.sampler.spp = spp; sv.sampler.seed = 0; sv.sampler.randomize = WF_RAND_FAST_OWEN;
        int log2spp = 31 - __builtin_clz((unsigned)spp);
        int res = 1;
        while (res < std::max(rx, ry)) res *= 2;
        sv.sampler.log2spp = log2spp;
        sv.sampler.nBase4Digits = (31 - __builtin_clz((unsigned)res)) + (log2spp + 1) / 2;
        std::vector<float> out((size_t)n * nd);
        for (int i = 0; i < n; ++i) {
            ZSobol s(sv);
            s.StartPixelSample(in[3 * i], in[3 * i + 1], in[3 * i + 2], startDim);
            for (int d = 0; d < nd; ++d) out[(size_t)i * nd + d] = s.Get1D();
        }
        writeBin(od + "/" + name + "_out.bin", out.data(), out.size() * 4);
    };
    sampler("zsobol", 16, 400, 400, 0, 12);
    sampler("zsobol2", 64, 1920, 1080, 13, 8);
    {
        std::vector<float> in = readBin<float>(gd + "/triangle_in.bin");
        int n = (int)in.size() / 16;
        std::vector<float> out((size_t)n * 5);
        for (int i = 0; i < n; ++i) {
            const float *r = &in[(size_t)i * 16];
            TriHit h{};
            bool hit = IntersectTriangle(V3{r[0], r[1], r[2]}, V3{r[3], r[4], r[5]}, r[6], V3{r[7], r[8], r[9]}, V3{r[10], r[11], r[12]}, V3{r[13], r[14], r[15]}, &h);
            float *w = &out[(size_t)i * 5];
            w[0] = hit; w[1] = hit ? h.b0 : 0; w[2] = hit ? h.b1 : 0; w[3] = hit ? h.b2 : 0; w[4] = hit ? h.t : 0;
        }
        writeBin(od + "/triangle_out.bin", out.data(), out.size() * 4);
    }
    {
        std::vector<float> in = readBin<float>(gd + "/sphtri_in.bin");
        int n = (int)in.size() / 14;
        std::vector<float> out((size_t)n * 6);
        for (int i = 0; i < n; ++i) {
            const float *r = &in[(size_t)i * 14];
            V3 v0{r[0], r[1], r[2]}, v1{r[3], r[4], r[5]}, v2{r[6], r[7], r[8]}, p{r[9], r[10], r[11]};
            float b[3], pdf;
            SampleSphericalTriangle(v0, v1, v2, p, V2{r[12], r[13]}, b, &pdf);
            float *w = &out[(size_t)i * 6];
            w[0] = b[0]; w[1] = b[1]; w[2] = b[2]; w[3] = pdf;
            V3 ps = b[0] * v0 + b[1] * v1 + b[2] * v2;
            V3 wd = ps - p;
            V2 iu{0, 0};
            if (pdf > 0 && LengthSquared(wd) > 0) iu = InvertSphericalTriangleSample(v0, v1, v2, p, Normalize(wd));
            w[4] = iu.x; w[5] = iu.y;
        }
        writeBin(od + "/sphtri_out.bin", out.data(), out.size() * 4);
    }
    {
        std::vector<float> in = readBin<float>(gd + "/bxdf_in.bin");
        int n = (int)in.size() / 13;
        std::vector<float> out((size_t)n * 16);
        for (int i = 0; i < n; ++i) {
            const float *r = &in[(size_t)i * 13];
            int type = (int)r[0];
            V3 wo{r[1], r[2], r[3]}, wi{r[4], r[5], r[6]};
            float uc = r[7];
            V2 u{r[8], r[9]};
            float eta = r[10], ax = r[11], ay = r[12], kk = 2.f * eta;
            float *w = &out[(size_t)i * 16];
            TrowbridgeReitz distrib(ax, ay);
            if (type == 0) evalBxDF(DiffuseBxDF{S4c(0.5f)}, wo, wi, uc, u, w);
            else if (type == 1) evalBxDF(DielectricBxDF{eta, distrib}, wo, wi, uc, u, w);
            else if (type == 2) evalBxDF(ConductorBxDF{distrib, S4c(eta), S4c(kk)}, wo, wi, uc, u, w);
            else if (type == 3) evalBxDF(ThinDielectricBxDF{eta}, wo, wi, uc, u, w);
            else if (type == 4) evalBxDF(DiffuseTransmissionBxDF{S4c(0.25f), S4c(0.5f)}, wo, wi, uc, u, w);
            else if (type == 5)
                evalBxDF(CoatedDiffuseBxDF{DielectricBxDF{eta, distrib}, DiffuseBxDF{S4c(0.5f)}, 0.01f, 0.2f, S4c((i / 7) % 2 ? 0.3f : 0.f), 10,
                                           1 + (i / 14) % 2, 0},
                         wo, wi, uc, u, w);
            else
                evalBxDF(CoatedConductorBxDF{DielectricBxDF{eta, distrib}, ConductorBxDF{TrowbridgeReitz(ay, ax), S4c(eta), S4c(kk)}, 0.01f, 0.2f,
                                             S4c((i / 7) % 2 ? 0.3f : 0.f), 10, 1 + (i / 14) % 2, 0},
                         wo, wi, uc, u, w);
        }
        writeBin(od + "/bxdf_out.bin", out.data(), out.size() * 4);
    }
    {
        std::vector<float> in = readBin<float>(gd + "/scalar_in.bin");
        int n = (int)in.size();
        std::vector<float> out((size_t)n * 4);
        for (int i = 0; i < n; ++i) {
            float x = in[i];
            out[4 * i] = FastExp(-x * 20);
            out[4 * i + 1] = SampleVisibleWavelengths(x);
            out[4 * i + 2] = VisibleWavelengthsPDF(360 + 470 * x);
            out[4 * i + 3] = Blackbody(360 + 470 * x, 2000 + 4000 * x);
        }
        writeBin(od + "/scalar_out.bin", out.data(), out.size() * 4);
    }
    {   // instance: the ray and interaction transforms of TransformedPrimitive::Intersect.  The matrices are taken from the
        // reference's output (they are the host parser's job, checked by the instances golden scene); everything else is
        // recomputed with InstanceRay / XfP3i / InstanceInteractionP / InstanceWoP
        std::vector<float> in = readBin<float>(gd + "/instance_in.bin"), ref = readBin<float>(gd + "/instance_out.bin");
        const int wi = 32, wo_ = 60, n = (int)in.size() / wi;
        std::vector<float> out((size_t)n * wo_);
        for (int i = 0; i < n; ++i) {
            const float *r = &in[(size_t)i * wi], *rf = &ref[(size_t)i * wo_];
            float *w = &out[(size_t)i * wo_];
            wf_instance inst{};
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b) { inst.render_from_instance.m[a][b] = rf[4 * a + b]; inst.render_from_instance.mInv[a][b] = rf[16 + 4 * a + b]; }
            for (int k = 0; k < 32; ++k) w[k] = rf[k];
            float tMax = r[16];
            V3 o, d;
            InstanceRay(inst, V3{r[10], r[11], r[12]}, V3{r[13], r[14], r[15]}, &tMax, &o, &d);
            w[32] = o.x; w[33] = o.y; w[34] = o.z; w[35] = d.x; w[36] = d.y; w[37] = d.z; w[38] = tMax;
            SurfIntr si{};
            si.pi = MakeP3i(V3{r[17], r[18], r[19]}, V3{r[20], r[21], r[22]});
            si.n = N3{r[23], r[24], r[25]};
            si.ns = -si.n;
            si.dpdu = V3{r[26], r[27], r[28]};
            InstanceInteractionP(&inst, &si);
            w[39] = si.pi.lo.x; w[40] = si.pi.lo.y; w[41] = si.pi.lo.z; w[42] = si.pi.hi.x; w[43] = si.pi.hi.y; w[44] = si.pi.hi.z;
            w[45] = si.n.x; w[46] = si.n.y; w[47] = si.n.z;
            w[48] = si.dpdu.x; w[49] = si.dpdu.y; w[50] = si.dpdu.z;
            w[51] = si.ns.x; w[52] = si.ns.y; w[53] = si.ns.z;
            // ts.wo = Normalize(t(si.wo)), si.wo = Normalize(wo) from the Interaction ctor
            V3 wv = Normalize(XfVector3(inst.render_from_instance.m, Normalize(V3{r[29], r[30], r[31]})));
            w[54] = wv.x; w[55] = wv.y; w[56] = wv.z;
            w[57] = w[58] = w[59] = 0;
        }
        writeBin(od + "/instance_out.bin", out.data(), out.size() * 4);
    }
    return 0;
}
