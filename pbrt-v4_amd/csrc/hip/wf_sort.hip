// wf_sort.hip — the key/value radix sort behind the ray-coherence pass (wf_backend.hip: SortRayQueue).  rocPRIM's device-wide
// radix sort is the library primitive; the keys, the permutation of the queues and everything on the tracing path are ours.
// Its own translation unit: the rocPRIM headers cost ~20 s of compile time.
#include <hip/hip_runtime.h>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

#include <cstdint>

// temp == nullptr: *tempBytes receives the scratch size needed for n pairs.  Sorts by the low `endBit` bits of the keys.
extern "C" int wf_sort_pairs_u32(hipStream_t stream, void *temp, size_t *tempBytes, const uint32_t *keysIn, uint32_t *keysOut, const uint32_t *valsIn, uint32_t *valsOut,
                                 unsigned n, unsigned endBit) {
    return (int)rocprim::radix_sort_pairs(temp, *tempBytes, keysIn, keysOut, valsIn, valsOut, n, 0u, endBit, stream);
}

// ---- HLBVH build, device part (cpu/aggregates.cpp:394-411): Morton codes of the primitive centroids + stable radix sort ------
// codes[i] / order[i]: code and input position of the i-th primitive in Morton order — what the host builder's treelet emission
// consumes (csrc/host/bvh_build.cpp).  The arithmetic is Bounds3f::Offset (util/vecmath.h:1374-1384) and EncodeMorton3
// (util/math.h:99-119) operation for operation, so the order is the one the reference's RadixSort produces.
namespace {
__device__ inline uint32_t LeftShift3(uint32_t x) {
    if (x == (1u << 10)) --x;
    x = (x | (x << 16)) & 0b00000011000000000000000011111111u;
    x = (x | (x << 8)) & 0b00000011000000001111000000001111u;
    x = (x | (x << 4)) & 0b00000011000011000011000011000011u;
    x = (x | (x << 2)) & 0b00001001001001001001001001001001u;
    return x;
}
__global__ void k_morton_codes(int n, const float *c, float minx, float miny, float minz, float maxx, float maxy, float maxz, uint32_t *codes, uint32_t *index) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float ox = c[3 * (size_t)i] - minx, oy = c[3 * (size_t)i + 1] - miny, oz = c[3 * (size_t)i + 2] - minz;
        if (maxx > minx) ox /= maxx - minx;
        if (maxy > miny) oy /= maxy - miny;
        if (maxz > minz) oz /= maxz - minz;
        ox *= 1024.f; oy *= 1024.f; oz *= 1024.f;
        codes[i] = (LeftShift3((uint32_t)oz) << 2) | (LeftShift3((uint32_t)oy) << 1) | LeftShift3((uint32_t)ox);
        index[i] = (uint32_t)i;
    }
}
}  // namespace

extern "C" int wf_morton_sort(int n, const float *centroids, const float bounds[6], uint32_t *codes, uint32_t *order) {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0) { (void)hipGetLastError(); return -1; }
    float *dc = nullptr;
    uint32_t *dk[2] = {nullptr, nullptr}, *dv[2] = {nullptr, nullptr};
    void *temp = nullptr;
    size_t tempBytes = 0;
    int rc = -1;
    do {
        if (hipMalloc((void **)&dc, (size_t)3 * n * sizeof(float)) != hipSuccess) break;
        bool ok = true;
        for (int k = 0; k < 2; ++k) ok = ok && hipMalloc((void **)&dk[k], (size_t)n * 4) == hipSuccess && hipMalloc((void **)&dv[k], (size_t)n * 4) == hipSuccess;
        if (!ok) break;
        if (hipMemcpy(dc, centroids, (size_t)3 * n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) break;
        hipLaunchKernelGGL(k_morton_codes, dim3((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256), dim3(256), 0, 0, n, dc, bounds[0], bounds[1], bounds[2], bounds[3], bounds[4],
                           bounds[5], dk[0], dv[0]);
        if (rocprim::radix_sort_pairs(nullptr, tempBytes, dk[0], dk[1], dv[0], dv[1], (unsigned)n, 0u, 30u, (hipStream_t)0) != hipSuccess) break;
        if (hipMalloc(&temp, tempBytes ? tempBytes : 1) != hipSuccess) break;
        if (rocprim::radix_sort_pairs(temp, tempBytes, dk[0], dk[1], dv[0], dv[1], (unsigned)n, 0u, 30u, (hipStream_t)0) != hipSuccess) break;
        if (hipMemcpy(codes, dk[1], (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) break;
        if (hipMemcpy(order, dv[1], (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) break;
        rc = 0;
    } while (false);
    (void)hipFree(dc); (void)hipFree(temp);
    for (int k = 0; k < 2; ++k) { (void)hipFree(dk[k]); (void)hipFree(dv[k]); }
    if (rc != 0) (void)hipGetLastError();
    return rc;
}
