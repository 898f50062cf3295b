// wf_mat.hip — one translation unit per material type and texture-context variant (compiled with
// -DWF_MAT_INSTANCE=<wf_material_type> -DWF_MAT_TEXCTX=<0|1|2>):
// the K9 kernel "<Material> + BxDF eval" (EvaluateMaterialAndBSDF<M, BasicTextureEvaluator>,
// wavefront/surfscatter.cpp:57-328) and its launcher.  Split from wf_backend.hip so that the seven material
// kernels compile in parallel (the layered ones take minutes).
#include <hip/hip_runtime.h>

#include "../common/wf_kernels.h"

using namespace wf;

#if !defined(WF_MAT_INSTANCE) || !defined(WF_MAT_TEXCTX)
#error "compile with -DWF_MAT_INSTANCE=<wf_material_type> -DWF_MAT_TEXCTX=<0|1|2>"
#endif

constexpr int MBLOCK = 256;

// minimum waves per SIMD (the register budget: 2 -> 256 VGPRs, 3 -> 168).  Measured per material on the spec scene, 2 vs 3 waves: diffuse
// 24.1 -> 22.7 ms per 16 spp, conductor 6.27 -> 6.40, coated diffuse 17.4 -> 20.2 (its stochastic walks spill).  Rounds 3-4 ran the diffuse
// kernel at 3 waves for those 6 %; round 4 took it back: at 168 VGPRs the unit spills ~170 VGPRs beside ~300 SGPRs that the compiler
// spills THROUGH VGPR lanes, and that build was the common factor of three wrong-code incidents that no source change explains — a
// never-executed conditional store that made cornell64 differ from run to run, a 48-frame texture stack that made arealight_image
// unrepeatable, and (after an unrelated header change) a memory access fault on every render of cornell64; the same source at 2 waves,
// at -O2, with zero-initialised locals or with -mllvm -amdgpu-spill-sgpr-to-vgpr=0 is correct each time.
#ifndef WF_MAT_WAVES
#define WF_MAT_WAVES 2
#endif
// WF_MAT_SV_PTR (round 4): the scene view read through the pointer to its device-resident copy instead of from the kernel
// arguments — every field that is live across the kernel is an SGPR pair either way, but a by-value argument invites the compiler to keep
// all of them (300-400 spilled SGPRs per material kernel)
#ifndef WF_MAT_SV_PTR
#define WF_MAT_SV_PTR 1   // spec scene, 16 spp, same box: diffuse 21.4 -> 20.6, conductor 5.25 -> 4.98, coated diffuse 16.0 -> 15.7 ms; conductor 314 -> 204 spilled SGPRs (profiles/r04_material_sv_pointer_ab_sm16.txt)
#endif
#if WF_MAT_SV_PTR
template <int MAT, int TEXCTX>
__global__ void __launch_bounds__(MBLOCK, WF_MAT_WAVES) k_eval_material(const SceneView *__restrict__ svp, WorkState ws, int cur) {
    const SceneView &sv = *svp;
#else
template <int MAT, int TEXCTX>
__global__ void __launch_bounds__(MBLOCK, WF_MAT_WAVES) k_eval_material(const SceneView sv, WorkState ws, int cur) {
#endif
    const int n = ws.counters[(CNT_MAT0 + MAT) * CNT_STRIDE];
    // block-uniform trip count: BlockAlloc inside the body synchronises the workgroup
    for (int base = blockIdx.x * MBLOCK; base < n; base += gridDim.x * MBLOCK) {
        const int i = base + threadIdx.x;
        KEvalMaterial<MAT, TEXCTX>(sv, ws, cur, i, i < n);
    }
}

#define WF_CAT3_(a, b, c, d) a##b##c##d
#define WF_CAT3(a, b, c, d) WF_CAT3_(a, b, c, d)
// WF_MAT_TEXCTX = 1: some texture depends on the footprint, or some material has a displacement texture / normal map;
// 2: the same plus the rarely used light types (KEvalMaterial)
extern "C" void WF_CAT3(wf_launch_eval_material_, WF_MAT_INSTANCE, _, WF_MAT_TEXCTX)(hipStream_t stream, int grid, const SceneView *sv, const WorkState *ws, int cur) {
#if WF_MAT_SV_PTR
    hipLaunchKernelGGL((k_eval_material<WF_MAT_INSTANCE, WF_MAT_TEXCTX>), dim3(grid), dim3(MBLOCK), 0, stream, sv->self, *ws, cur);
#else
    hipLaunchKernelGGL((k_eval_material<WF_MAT_INSTANCE, WF_MAT_TEXCTX>), dim3(grid), dim3(MBLOCK), 0, stream, *sv, *ws, cur);
#endif
}
