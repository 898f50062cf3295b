// tools/exp/wwm_bracket/harness.cpp — runs k_syn of two code objects (default flags; -amdgpu-spill-sgpr-to-vgpr=0) and the host version of the
// same arithmetic on the same input and compares the three bit for bit.   harness case_default.co case_safe.co case_ref.so [n] [blocks]
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cstdint>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
int main(int argc, char **argv) {
    if (argc < 4) return 2;
    const int n = argc > 4 ? atoi(argv[4]) : 1 << 16, blocks = argc > 5 ? atoi(argv[5]) : 96;
    void *so = dlopen(argv[3], RTLD_NOW);
    if (!so) { fprintf(stderr, "%s\n", dlerror()); return 2; }
    const int NP = ((int (*)())dlsym(so, "ref_np"))(), NC = ((int (*)())dlsym(so, "ref_nc"))();
    auto ref = (void (*)(const void *))dlsym(so, "ref_syn");
    const size_t len = n < 1024 ? 1024 : n;
    // Params: p[NP], c[NC], n, out — the layout both compilers give the generated struct
    size_t off_c = (size_t)NP * 8, off_n = off_c + (size_t)NC * 4, off_out = (off_n + 4 + 7) / 8 * 8, size = off_out + 8;
    std::vector<unsigned char> hp(size, 0), dp(size, 0);
    std::vector<std::vector<float>> arrays(NP, std::vector<float>(len));
    uint32_t s = 12345;
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 16384.f - 2.f; };
    for (int j = 0; j < NP; ++j) {
        for (auto &x : arrays[j]) x = rnd();
        float *d = nullptr;
        CHK(hipMalloc(&d, len * 4));
        CHK(hipMemcpy(d, arrays[j].data(), len * 4, hipMemcpyHostToDevice));
        const float *h = arrays[j].data();
        memcpy(&hp[(size_t)j * 8], &h, 8);
        memcpy(&dp[(size_t)j * 8], &d, 8);
    }
    for (int j = 0; j < NC; ++j) { float c = rnd() * 0.5f; memcpy(&hp[off_c + 4 * j], &c, 4); memcpy(&dp[off_c + 4 * j], &c, 4); }
    memcpy(&hp[off_n], &n, 4); memcpy(&dp[off_n], &n, 4);
    std::vector<float> want(n), got(n);
    float *hout = want.data();
    memcpy(&hp[off_out], &hout, 8);
    ref(hp.data());
    int bad[2] = {0, 0};
    for (int v = 0; v < 2; ++v) {
        hipModule_t m; hipFunction_t f;
        CHK(hipModuleLoad(&m, argv[1 + v]));
        CHK(hipModuleGetFunction(&f, m, "k_syn"));
        float *dout = nullptr;
        CHK(hipMalloc(&dout, (size_t)n * 4));
        CHK(hipMemset(dout, 0xff, (size_t)n * 4));
        memcpy(&dp[off_out], &dout, 8);
        size_t sz = size;
        void *cfg[] = {HIP_LAUNCH_PARAM_BUFFER_POINTER, dp.data(), HIP_LAUNCH_PARAM_BUFFER_SIZE, &sz, HIP_LAUNCH_PARAM_END};
        CHK(hipModuleLaunchKernel(f, blocks, 1, 1, 256, 1, 1, 0, nullptr, nullptr, cfg));
        hipError_t e = hipDeviceSynchronize();
        if (e != hipSuccess) { printf("%s: %s\n", argv[1 + v], hipGetErrorString(e)); bad[v] = -1; break; }
        CHK(hipMemcpy(got.data(), dout, (size_t)n * 4, hipMemcpyDeviceToHost));
        int first = -1;
        for (int i = 0; i < n; ++i) {
            uint32_t a, b; memcpy(&a, &got[i], 4); memcpy(&b, &want[i], 4);
            const bool nanA = got[i] != got[i], nanB = want[i] != want[i];
            if (!(a == b || (nanA && nanB))) { if (first < 0) first = i; ++bad[v]; }
        }
        printf("%s: %d of %d outputs differ from the host's%s\n", argv[1 + v], bad[v], n, first >= 0 ? "" : " (bit-identical)");
        if (first >= 0) printf("    first: i = %d (lane %d of its wave)  device %a  host %a\n", first, first & 63, got[first], want[first]);
    }
    return bad[0] != 0 || bad[1] != 0;
}
