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
