// wf_mat.hip — one translation unit per material type (compiled with -DWF_MAT_INSTANCE=<wf_material_type>):
// the K9 kernel "<Material> + BxDF eval" (EvaluateMaterialAndBSDF<M, BasicTextureEvaluator>,
// wavefront/surfscatter.cpp:57-328) and its launcher.  Split from wf_backend.hip so that the seven material
// kernels compile in parallel (the layered ones take minutes).
#include <hip/hip_runtime.h>

#include "../common/wf_kernels.h"

using namespace wf;

#ifndef WF_MAT_INSTANCE
#error "compile with -DWF_MAT_INSTANCE=<wf_material_type>"
#endif

constexpr int MBLOCK = 256;

#ifndef WF_MAT_WAVES
#define WF_MAT_WAVES 2
#endif
template <int MAT, bool TEXCTX>
__global__ void __launch_bounds__(MBLOCK, WF_MAT_WAVES) k_eval_material(const SceneView sv, WorkState ws, int cur) {
    const int n = ws.counters[(CNT_MAT0 + MAT) * CNT_STRIDE];
    // block-uniform trip count: BlockAlloc inside the body synchronises the workgroup
    for (int base = blockIdx.x * MBLOCK; base < n; base += gridDim.x * MBLOCK) {
        const int i = base + threadIdx.x;
        KEvalMaterial<MAT, TEXCTX>(sv, ws, cur, i, i < n);
    }
}

#define WF_CAT2(a, b) a##b
#define WF_CAT(a, b) WF_CAT2(a, b)
extern "C" void WF_CAT(wf_launch_eval_material_, WF_MAT_INSTANCE)(hipStream_t stream, int grid, const SceneView *sv, const WorkState *ws, int cur) {
    // sv->texNeedsFootprint: some texture depends on the footprint, or some material has a displacement texture
    if (sv->texNeedsFootprint) hipLaunchKernelGGL((k_eval_material<WF_MAT_INSTANCE, true>), dim3(grid), dim3(MBLOCK), 0, stream, *sv, *ws, cur);
    else hipLaunchKernelGGL((k_eval_material<WF_MAT_INSTANCE, false>), dim3(grid), dim3(MBLOCK), 0, stream, *sv, *ws, cur);
}
