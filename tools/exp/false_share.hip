// build: hipcc --offload-arch=gfx950 -O2 tools/exp/false_share.hip -o tools/exp/false_share ; result on MI355X (round 3): 0 of 4 M records lost in 20 trials
// Do plain stores from workgroups on DIFFERENT XCDs into the same 128-byte line both survive?  (diagnostic for the near-tie queue:
// service workgroups write hit records whose neighbours in the line are written by worker workgroups)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
struct Rec { float a, b, c, d; };
__global__ void k(Rec *out, int *out4, int n, int rounds) {
    // block b writes records i with (i / 1) % gridDim.x == b  (adjacent 16-byte records from different blocks: blocks b and b+1 sit on different XCDs)
    for (int r = 0; r < rounds; ++r)
        for (int i = blockIdx.x + threadIdx.x * gridDim.x; i < n; i += gridDim.x * blockDim.x) {
            // spread the blocks' progress so that lines are touched at different times
            if ((i + r + blockIdx.x) % 7 == 0) __builtin_amdgcn_s_sleep(8);
            out[i] = Rec{(float)i, (float)r, 1.f, 2.f};
            out4[i] = i ^ r;
        }
}
int main() {
    const int n = 1 << 22, rounds = 3;
    Rec *d; int *d4;
    hipMalloc(&d, n * sizeof(Rec)); hipMalloc(&d4, n * sizeof(int));
    int bad = 0, bad4 = 0;
    for (int trial = 0; trial < 20; ++trial) {
        hipMemset(d, 0xff, n * sizeof(Rec)); hipMemset(d4, 0xff, n * sizeof(int));
        hipLaunchKernelGGL(k, dim3(512), dim3(256), 0, 0, d, d4, n, rounds);
        hipDeviceSynchronize();
        std::vector<Rec> h(n); std::vector<int> h4(n);
        hipMemcpy(h.data(), d, n * sizeof(Rec), hipMemcpyDeviceToHost); hipMemcpy(h4.data(), d4, n * sizeof(int), hipMemcpyDeviceToHost);
        for (int i = 0; i < n; ++i) { if (h[i].a != (float)i || h[i].b != (float)(rounds - 1)) ++bad; if (h4[i] != (i ^ (rounds - 1))) ++bad4; }
    }
    printf("16-byte records lost: %d, 4-byte words lost: %d (of %d x 20)\n", bad, bad4, n);
    return 0;
}
