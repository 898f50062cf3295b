// tools/exp/light_vgprs.hip — which part of next-event estimation sets the register allocation of the NEE kernels (round 5):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -c tools/exp/light_vgprs.hip -o /tmp/lt.o && python tools/kernel_resources.py /tmp/lt.o
// k_ls (light-BVH descent) 62 VGPRs, k_tri (triangle emitter) 79, k_sph (the out-of-line quadric / patch emitter sampler) 214,
// k_li / k_both (everything inlined behind one switch) 214 -> without the quadric sampler: see DESIGN.md 4.2.
#include <hip/hip_runtime.h>
#include "../../pbrt-v4_amd/csrc/common/wf_kernels.h"
using namespace wf;
struct Req { LightCtx ctx; float u0; V2 u; Wavelengths lambda; };
__global__ void __launch_bounds__(256) k_ls(const SceneView *svp, const Req *in, int *outId, float *outPmf) {
    const SceneView &sv = *svp; Req r = in[threadIdx.x + blockIdx.x * 256];
    float pmf; int id = LightSamplerSample(sv, r.ctx, r.u0, &pmf); outId[threadIdx.x + blockIdx.x * 256] = id; outPmf[threadIdx.x + blockIdx.x * 256] = pmf;
}
__global__ void __launch_bounds__(256) k_li(const SceneView *svp, const Req *in, const int *ids, LightLiSample *out) {
    const SceneView &sv = *svp; Req r = in[threadIdx.x + blockIdx.x * 256];
    out[threadIdx.x + blockIdx.x * 256] = LightSampleLi<false>(sv, sv.lights[ids[threadIdx.x + blockIdx.x * 256]], r.ctx, r.u, r.lambda, true);
}
__global__ void __launch_bounds__(256) k_tri(const SceneView *svp, const Req *in, const int *ids, ShapeSampleR *out) {
    const SceneView &sv = *svp; Req r = in[threadIdx.x + blockIdx.x * 256];
    out[threadIdx.x + blockIdx.x * 256] = TriangleSample(sv, ids[threadIdx.x + blockIdx.x * 256], r.ctx.pi, r.ctx.ns, r.u);
}
__global__ void __launch_bounds__(256) k_sph(const SceneView *svp, const Req *in, const int *ids, ShapeSampleR *out) {
    const SceneView &sv = *svp; Req r = in[threadIdx.x + blockIdx.x * 256];
    out[threadIdx.x + blockIdx.x * 256] = SphereSample(sv, ids[threadIdx.x + blockIdx.x * 256], r.ctx.pi, r.ctx.n, r.ctx.ns, r.u);
}
__global__ void __launch_bounds__(256) k_both(const SceneView *svp, const Req *in, LightPick *out) {
    const SceneView &sv = *svp; Req r = in[threadIdx.x + blockIdx.x * 256];
    out[threadIdx.x + blockIdx.x * 256] = SampleLightDirect<false>(sv, r.ctx, r.u0, r.u, r.lambda);
}
