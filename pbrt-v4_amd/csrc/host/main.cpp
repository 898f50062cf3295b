// main.cpp — pbrt_amd: command-line front end with the reference CLI's flags for this path
// (src/pbrt/cmd/pbrt.cpp:105-213: --spp, --seed, --outfile, --quiet, --stats, --pixelbounds, --cropwindow,
// --gpu-device; --gpu/--wavefront are accepted and implied).  It always renders with the HIP wavefront
// back end: there is no CPU renderer in this program.
#include "integrator.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

using namespace wf;

static void usage() {
    fprintf(stderr,
            "usage: pbrt_amd [options] <scene.pbrt>\n"
            "  --spp <n>              override samples per pixel\n"
            "  --seed <n>             random seed\n"
            "  --outfile <file>       output image (.pfm or .exr)\n"
            "  --gpu-device <n>       HIP device index (default 0)\n"
            "  --gpus <n>             render on HIP devices 0 .. n-1: the image in interleaved 16-line strips, one host thread per device,\n"
            "                         the strips gathered peer to peer into device 0's film (bit-identical to one device)\n"
            "  --gpu-devices a,b,...  the same on the listed devices (one may be named several times: several contexts on one GPU)\n"
            "  --pixelbounds x0,x1,y0,y1 / --cropwindow x0,x1,y0,y1\n"
            "  --disable-pixel-jitter / --disable-wavelength-jitter / --disable-texture-filtering\n"
            "  --displacement-edge-scale <s>   scale the target edge length of displaced meshes\n"
            "  --render-coord-sys <name>       camera, cameraworld (default) or world\n"
            "  --pixel <x,y>  --debugstart <first[,count]>  --quick  --disable-image-textures  --nthreads <n>\n"
            "  --datadir <dir>        directory holding spectral_tables.txt\n"
            "  --stats                print ray counts and the per-kernel profile\n"
            "  --quiet, --gpu, --wavefront (accepted)\n");
}

static int Main(int argc, char **argv);
int main(int argc, char **argv) {
    try {
        return Main(argc, argv);
    } catch (const std::exception &e) {
        fprintf(stderr, "%s\n", e.what());
        return 1;
    }
}
static int Main(int argc, char **argv) {
    RenderOptions opt;
    std::string scenePath, dataDir;
    int device = 0;
    std::vector<int> devices;   // --gpus / --gpu-devices: more than one entry = the multi-device renderer
    bool stats = false;
    int debugFirst = -1, debugCount = 1;   // --debugstart
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        auto value = [&]() -> std::string {
            size_t eq = a.find('=');
            if (eq != std::string::npos) return a.substr(eq + 1);
            if (i + 1 >= argc) { fprintf(stderr, "missing value after %s\n", a.c_str()); exit(1); }
            return argv[++i];
        };
        auto is = [&](const char *name) { return a == name || a.rfind(std::string(name) + "=", 0) == 0; };
        if (is("--spp")) opt.pixelSamples = atoi(value().c_str());
        else if (is("--seed")) opt.seed = atoi(value().c_str());
        else if (is("--outfile")) opt.imageFile = value();
        else if (is("--gpu-devices")) {
            devices.clear();
            const std::string v = value();
            for (size_t p = 0; p < v.size();) {
                size_t q = v.find(',', p);
                if (q == std::string::npos) q = v.size();
                devices.push_back(atoi(v.substr(p, q - p).c_str()));
                p = q + 1;
            }
            if (devices.empty()) { usage(); return 1; }
        } else if (is("--gpus")) {
            const int n = atoi(value().c_str());
            if (n < 1) { usage(); return 1; }
            devices.clear();
            for (int k = 0; k < n; ++k) devices.push_back(k);
        } else if (is("--gpu-device")) device = atoi(value().c_str());
        else if (is("--datadir")) dataDir = value();
        else if (is("--pixelbounds")) {
            if (sscanf(value().c_str(), "%d,%d,%d,%d", &opt.pixelBounds[0], &opt.pixelBounds[1], &opt.pixelBounds[2], &opt.pixelBounds[3]) != 4) { usage(); return 1; }
            opt.hasPixelBounds = true;
        } else if (is("--cropwindow")) {
            if (sscanf(value().c_str(), "%f,%f,%f,%f", &opt.cropWindow[0], &opt.cropWindow[1], &opt.cropWindow[2], &opt.cropWindow[3]) != 4) { usage(); return 1; }
            opt.hasCropWindow = true;
        } else if (a == "--disable-pixel-jitter") opt.disablePixelJitter = true;
        else if (a == "--disable-wavelength-jitter") opt.disableWavelengthJitter = true;
        else if (a == "--disable-texture-filtering") opt.disableTextureFiltering = true;
        else if (a == "--displacement-edge-scale" && i + 1 < argc) opt.displacementEdgeScale = (float)atof(argv[++i]);
        else if (a == "--render-coord-sys" && i + 1 < argc) {
            const std::string v = argv[++i];
            if (v == "camera") opt.renderingSpace = 0;
            else if (v == "cameraworld") opt.renderingSpace = 1;
            else if (v == "world") opt.renderingSpace = 2;
            else { fprintf(stderr, "%s: unknown rendering coordinate system.\n", v.c_str()); return 1; }
        }
        else if (a == "--quick") opt.quickRender = true;
        else if (a == "--disable-image-textures") opt.disableImageTextures = true;
        else if (is("--pixel")) {   // cmd/pbrt.cpp:136-143
            int px, py;
            if (sscanf(value().c_str(), "%d,%d", &px, &py) != 2) { usage(); return 1; }
            opt.pixelBounds[0] = px; opt.pixelBounds[1] = px + 1; opt.pixelBounds[2] = py; opt.pixelBounds[3] = py + 1;
            opt.hasPixelBounds = true;
        } else if (is("--debugstart")) {   // wavefront/integrator.cpp:320-332: first sample index [, number of sample indices]
            int nv = sscanf(value().c_str(), "%d,%d", &debugFirst, &debugCount);
            if (nv < 1) { fprintf(stderr, "Expected either one or two integer values for --debugstart.\n"); return 1; }
            if (nv == 1) debugCount = 1;
        } else if (is("--nthreads")) { const std::string n = value(); setenv("WF_BUILD_THREADS", n.c_str(), 1); }
        else if (a == "--quiet") opt.quiet = true;
        else if (a == "--stats") stats = true;
        else if (a == "--gpu" || a == "--wavefront") {}
        else if (a == "--help" || a == "-h") { usage(); return 0; }
        else if (a[0] == '-') { fprintf(stderr, "unknown option %s\n", a.c_str()); usage(); return 1; }
        else scenePath = a;
    }
    if (scenePath.empty()) { usage(); return 1; }
    if (dataDir.empty()) {
        std::string self = argv[0];
        size_t p = self.rfind('/');
        dataDir = (p == std::string::npos ? std::string(".") : self.substr(0, p)) + "/../data";
    }
    SpectralData::Init(dataDir, dataDir + "/cache");
    SetMortonSort(&wf_morton_sort);   // as wfh_init does: the device parts of the BVH builds when a GPU is visible
    SetSahBuild(&wf_build_bvh_sah);
    ParsedScene parsed;
    ParseFiles({scenePath}, &opt, &parsed);
    SceneTables T;
    BuildSceneTables(parsed, opt, &T);
    const wf_film &F = T.desc.film;
    const int W = F.pixel_max[0] - F.pixel_min[0], H = F.pixel_max[1] - F.pixel_min[1];

    const int firstSample = debugFirst >= 0 ? debugFirst : 0, lastSample = debugFirst >= 0 ? debugFirst + debugCount : T.spp;
    std::unique_ptr<MultiDeviceRenderer> multi;
    std::unique_ptr<WavefrontRenderer> single;
    double seconds;
    if (devices.size() > 1) {
        if (F.type != WF_FILM_RGB) { fprintf(stderr, "Error: --gpus: only the \"rgb\" film is gathered across devices\n"); return 1; }
        multi.reset(new MultiDeviceRenderer(T, devices));
        if (stats) wf_profile_enable(multi->Primary().Context(), 1);
        std::vector<double> per;
        seconds = multi->Render(firstSample, lastSample, &per);
        if (!opt.quiet) {
            fprintf(stderr, "Rendering on %d devices:", (int)devices.size());
            for (size_t k = 0; k + 1 < per.size(); ++k) fprintf(stderr, " [%d] %.3f s", devices[k], per[k]);
            fprintf(stderr, ", gather %.4f s\n", per.back());
        }
    } else {
        single.reset(new WavefrontRenderer(T, devices.empty() ? device : devices[0]));
        if (stats) wf_profile_enable(single->Context(), 1);
        seconds = single->Render(firstSample, lastSample, 1);
    }
    WavefrontRenderer &renderer = multi ? multi->Primary() : *single;
    if (!opt.quiet) fprintf(stderr, "Rendering finished: %.3f s, %.2f Msamples/s\n", seconds, (double)W * H * (lastSample - firstSample) / seconds / 1e6);
    if (stats) {
        wf_render_stats st;
        renderer.Stats(&st);
        unsigned long long total = st.camera_rays;
        printf("  Wavefront integrator\n    Camera rays %20llu\n", (unsigned long long)st.camera_rays);
        for (int d = 1; d < 64; ++d)
            if (st.indirect_rays[d]) { printf("    Indirect rays, depth %-3d %12llu\n", d, (unsigned long long)st.indirect_rays[d]); total += st.indirect_rays[d]; }
        for (int d = 0; d < 64; ++d)
            if (st.shadow_rays[d]) { printf("    Shadow rays, depth %-3d   %12llu\n", d, (unsigned long long)st.shadow_rays[d]); total += st.shadow_rays[d]; }
        printf("    Total rays %21llu  (%.2f Mray/s)\n", total, total / seconds / 1e6);
        uint64_t items[16];
        if (wf_material_items_download(renderer.Context(), items) == 0) {   // (the denominators of bench.py's roofline_material / roofline_medium traffic)
            unsigned long long mat = 0;
            for (int t = 0; t < WF_MAT_NTYPES; ++t) mat += items[t];
            printf("    Material items %17llu\n    Medium-sample items %12llu\n", mat, (unsigned long long)items[WF_MAT_NTYPES]);
        }
        std::vector<wf_kernel_profile_entry> ent(64);
        int n = 0;
        wf_profile_report(renderer.Context(), ent.data(), (int)ent.size(), &n);
        float sum = 0;
        for (int i = 0; i < n; ++i) sum += ent[i].total_ms;
        printf("  Wavefront Kernel Profile\n");
        for (int i = 0; i < n; ++i)
            printf("    %-52s %6d launches %10.2f ms / %5.1f%% (avg %8.3f, min %8.3f, max %8.3f)\n", ent[i].name, ent[i].launches, ent[i].total_ms,
                   100.f * ent[i].total_ms / sum, ent[i].total_ms / ent[i].launches, ent[i].min_ms, ent[i].max_ms);
        printf("    Total GPU time: %.2f ms\n", sum);
    }
    std::vector<double> film((size_t)W * H * 4);
    renderer.DownloadFilm(film.data());
    std::vector<float> rgb((size_t)W * H * 3);
    if (F.type != WF_FILM_RGB) {   // SpectralFilm / GBufferFilm::WriteImage: the multi-channel .exr
        std::vector<std::string> names;
        std::vector<float> chans;
        if (F.type == WF_FILM_SPECTRAL) {
            std::vector<double> spectral((size_t)W * H * 2 * F.n_buckets);
            if (wf_film_spectral_download(renderer.Context(), spectral.data()) != 0) { fprintf(stderr, "Error: %s\n", wf_last_error()); return 1; }
            SpectralFilmImage(F, film.data(), spectral.data(), W, H, T.saveFP16, &names, &chans);
        } else {
            std::vector<wf_gbuffer_pixel> gb((size_t)W * H);
            if (wf_film_gbuffer_download(renderer.Context(), gb.data()) != 0) { fprintf(stderr, "Error: %s\n", wf_last_error()); return 1; }
            GBufferFilmImage(F, film.data(), gb.data(), W, H, T.saveFP16, &names, &chans);
        }
        if (!WriteEXRChannels(T.imageFile, names, chans.data(), W, H, T.saveFP16)) { fprintf(stderr, "Error: couldn't write %s\n", T.imageFile.c_str()); return 1; }
        return 0;
    }
    FilmToRGB(F, film.data(), W, H, rgb.data(), T.saveFP16);
    if (!WriteFilmImage(T, T.imageFile, rgb, W, H)) { fprintf(stderr, "Error: couldn't write %s\n", T.imageFile.c_str()); return 1; }
    return 0;
}
