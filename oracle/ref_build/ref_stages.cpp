// oracle/ref_build/ref_stages.cpp — TEST INFRASTRUCTURE ONLY (ours; links the shimmed reference build).
// Runs the reference's CPU WavefrontPathIntegrator stage by stage for sample index 0 of the first pass and
// dumps the queues after "Generate camera rays" and after IntersectClosest at depth 0, so that the restated
// stages can be compared item by item (tools/compare_stages.py).  Private members are reached with the
// test-only `#define private public` / `protected public` below; nothing in the reference is modified.
//   ref_stages scene.pbrt outdir
#include <algorithm>
#include <cstdio>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>
#include <atomic>
#include <mutex>
#include <thread>
#include <functional>
#include <unordered_map>
#include <set>
#include <list>
#include <array>
#include <optional>
#include <variant>
#include <iostream>
#include <fstream>
#include <cstring>
#include <cmath>
#include <chrono>
#include <condition_variable>
#include <future>
#include <shared_mutex>
#include <typeindex>
#include <typeinfo>

#define private public
#define protected public
#include <pbrt/pbrt.h>
#include <pbrt/cameras.h>
#include <pbrt/wavefront/integrator.h>
#undef private
#undef protected
#include <pbrt/options.h>
#include <pbrt/parser.h>
#include <pbrt/scene.h>
#include <pbrt/materials.h>

using namespace pbrt;

template <typename M>
static void dumpMat(FILE *f, MaterialEvalQueue *mq, int tag) {
    auto q = mq->Get<MaterialEvalWorkItem<M>>();
    for (int i = 0; i < q->Size(); ++i) {
        MaterialEvalWorkItem<M> w = (*q)[i];
        float rec[28] = {(float)tag, (float)w.pixelIndex,
                         w.pi.x.LowerBound(), w.pi.y.LowerBound(), w.pi.z.LowerBound(), w.pi.x.UpperBound(), w.pi.y.UpperBound(), w.pi.z.UpperBound(),
                         w.n.x, w.n.y, w.n.z, w.ns.x, w.ns.y, w.ns.z, w.dpdus.x, w.dpdus.y, w.dpdus.z, w.wo.x, w.wo.y, w.wo.z, w.uv[0], w.uv[1],
                         w.dpdu.x, w.dpdu.y, w.dpdu.z, w.dpdv.x, w.dpdv.y, w.dpdv.z};
        fwrite(rec, 4, 28, f);
    }
}

template <typename M>
static void dumpDiffs(FILE *f, MaterialEvalQueue *mq, WavefrontPathIntegrator *in) {
    auto q = mq->Get<MaterialEvalWorkItem<M>>();
    for (int i = 0; i < q->Size(); ++i) {
        MaterialEvalWorkItem<M> w = (*q)[i];
        Vector3f dpdx, dpdy;
        in->camera.Approximate_dp_dxy(Point3f(w.pi), w.n, w.time, in->samplesPerPixel, &dpdx, &dpdy);
        const CameraBase *cb = (const CameraBase *)in->camera.ptr();
        Point3f pc = cb->CameraFromRender(Point3f(w.pi), w.time);
        Normal3f nc = cb->CameraFromRender(w.n, w.time);
        Transform dz = RotateFromTo(Normalize(Vector3f(pc)), Vector3f(0, 0, 1));
        Point3f pd = dz(pc);
        Normal3f nd = dz(nc);
        float rec[19] = {(float)w.pixelIndex, dpdx.x, dpdx.y, dpdx.z, dpdy.x, dpdy.y, dpdy.z, pc.x, pc.y, pc.z, nc.x, nc.y, nc.z, pd.x, pd.y, pd.z, nd.x, nd.y, nd.z};
        fwrite(rec, 4, 19, f);
    }
}

int main(int argc, char **argv) {
    if (argc < 3) { fprintf(stderr, "usage: ref_stages scene.pbrt outdir\n"); return 1; }
    PBRTOptions opt;
    opt.wavefront = true;
    opt.quiet = true;
    opt.seed = 0;
    InitPBRT(opt);
    {
        BasicScene scene;
        BasicSceneBuilder builder(&scene);
        ParseFiles(&builder, {std::string(argv[1])});
        {   // object instances as the parser recorded them: m and mInv of renderFromInstance, 32 floats each
            FILE *f = fopen((std::string(argv[2]) + "/instances.bin").c_str(), "wb");
            for (const auto &inst : scene.instances)
                if (inst.renderFromInstance) {
                    float rec[32];
                    for (int a = 0; a < 4; ++a)
                        for (int b = 0; b < 4; ++b) { rec[4 * a + b] = inst.renderFromInstance->GetMatrix()[a][b]; rec[16 + 4 * a + b] = inst.renderFromInstance->GetInverseMatrix()[a][b]; }
                    fwrite(rec, 4, 32, f);
                }
            fclose(f);
        }
        WavefrontPathIntegrator *in = new WavefrontPathIntegrator(pstd::pmr::get_default_resource(), scene);
        std::string dir = argv[2];
        Bounds2i pb = in->film.PixelBounds();
        int y0 = pb.pMin.y;
        in->rayQueues[0]->Reset();
        in->GenerateCameraRays(y0, Transform(), 0);
        {
            FILE *f = fopen((dir + "/camera_rays.bin").c_str(), "wb");
            RayQueue *rq = in->rayQueues[0];
            for (int i = 0; i < rq->Size(); ++i) {
                RayWorkItem r = (*rq)[i];
                float rec[8] = {(float)r.pixelIndex, r.ray.o.x, r.ray.o.y, r.ray.o.z, r.ray.d.x, r.ray.d.y, r.ray.d.z, r.ray.time};
                fwrite(rec, 4, 8, f);
            }
            fclose(f);
        }
        // depth 0 (integrator.cpp:377-406)
        in->rayQueues[1]->Reset();
        if (in->escapedRayQueue) in->escapedRayQueue->Reset();
        in->hitAreaLightQueue->Reset();
        in->basicEvalMaterialQueue->Reset();
        in->universalEvalMaterialQueue->Reset();
        in->GenerateRaySamples(0, 0);
        in->aggregate->IntersectClosest(in->maxQueueSize, in->rayQueues[0], in->escapedRayQueue, in->hitAreaLightQueue, in->basicEvalMaterialQueue,
                                        in->universalEvalMaterialQueue, in->mediumSampleQueue, in->rayQueues[1]);
        {
            FILE *f = fopen((dir + "/mat_items.bin").c_str(), "wb");
            dumpMat<DiffuseMaterial>(f, in->basicEvalMaterialQueue, 1);
            dumpMat<ConductorMaterial>(f, in->basicEvalMaterialQueue, 2);
            dumpMat<DielectricMaterial>(f, in->basicEvalMaterialQueue, 3);
            dumpMat<CoatedDiffuseMaterial>(f, in->basicEvalMaterialQueue, 6);
            dumpMat<CoatedConductorMaterial>(f, in->basicEvalMaterialQueue, 7);
            fclose(f);
        }
        {
            {
                const CameraBase *cb = (const CameraBase *)in->camera.ptr();
                FILE *g = fopen((dir + "/camera_diffs.bin").c_str(), "wb");
                float rec[12] = {cb->minPosDifferentialX.x, cb->minPosDifferentialX.y, cb->minPosDifferentialX.z,
                                 cb->minPosDifferentialY.x, cb->minPosDifferentialY.y, cb->minPosDifferentialY.z,
                                 cb->minDirDifferentialX.x, cb->minDirDifferentialX.y, cb->minDirDifferentialX.z,
                                 cb->minDirDifferentialY.x, cb->minDirDifferentialY.y, cb->minDirDifferentialY.z};
                fwrite(rec, 4, 12, g);
                const Transform &rfc = cb->cameraTransform.renderFromCamera.startTransform;
                for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { float v = rfc.GetMatrix()[i][j]; fwrite(&v, 4, 1, g); }
                for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) { float v = rfc.GetInverseMatrix()[i][j]; fwrite(&v, 4, 1, g); }
                fclose(g);
            }
            FILE *f = fopen((dir + "/mat_diffs.bin").c_str(), "wb");
            for (MaterialEvalQueue *mq : {in->basicEvalMaterialQueue, in->universalEvalMaterialQueue}) {
                dumpDiffs<DiffuseMaterial>(f, mq, in);
                dumpDiffs<ConductorMaterial>(f, mq, in);
            }
            fclose(f);
        }
        {
            FILE *f = fopen((dir + "/samples.bin").c_str(), "wb");
            RayQueue *rq = in->rayQueues[0];
            for (int i = 0; i < rq->Size(); ++i) {
                RayWorkItem r = (*rq)[i];
                RaySamples rs = in->pixelSampleState.samples[r.pixelIndex];
                float rec[8] = {(float)r.pixelIndex, rs.direct.uc, rs.direct.u.x, rs.direct.u.y, rs.indirect.uc, rs.indirect.u.x, rs.indirect.u.y, rs.indirect.rr};
                fwrite(rec, 4, 8, f);
            }
            fclose(f);
        }
    }
    return 0;
}
