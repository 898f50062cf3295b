// libm_check.cpp — TEST INFRASTRUCTURE.  Compares the restated glibc float routines of
// pbrt-v4_amd/csrc/common/wf_libm.h with the live libm of this machine.
//   libm_check exhaustive            all 2^32 arguments of every one-argument function + 2^32 seeded atan2f pairs
//   libm_check quick                 2^24 strided arguments per function (CPU test suite)
//   libm_check eval <fn> < in > out  raw float32 stream through the *live libm* (used by the GPU tests and by
//                                    tools/make_libm_golden.py to produce tests/golden/libm_*.bin)
//   libm_check evalmine <fn> ...     the same through the restatement compiled for the host
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#include <string>
#include "../../pbrt-v4_amd/csrc/common/wf_libm.h"

typedef float (*F1)(float);
struct Fn { const char *name; F1 mine; F1 ref; };
static float r_sin(float x) { return sinf(x); }
static float r_cos(float x) { return cosf(x); }
static float r_exp(float x) { return expf(x); }
static float r_log(float x) { return logf(x); }
static float r_atan(float x) { return atanf(x); }
static float r_asin(float x) { return asinf(x); }
static float r_acos(float x) { return acosf(x); }
static float r_cosh(float x) { return coshf(x); }
static float r_sinh(float x) { return sinhf(x); }
static float r_expm1(float x) { return expm1f(x); }
static float r_atanh(float x) { return atanhf(x); }
static float r_tan(float x) { return tanf(x); }
static const Fn fns[] = {
    {"sin", glibc235::sinf, r_sin},     {"cos", glibc235::cosf, r_cos},     {"exp", glibc235::expf, r_exp},
    {"log", glibc235::logf, r_log},     {"atan", glibc235::atanf, r_atan},  {"asin", glibc235::asinf, r_asin},
    {"acos", glibc235::acosf, r_acos},  {"cosh", glibc235::coshf, r_cosh}, {"sinh", glibc235::sinhf, r_sinh}, {"expm1", glibc235::expm1f, r_expm1},  {"atanh", glibc235::atanhf, r_atanh}, {"tan", glibc235::tanf, r_tan},
};
static inline bool same(float a, float b) {
    if (a != a && b != b) return true;
    return glibc235::asuint(a) == glibc235::asuint(b);
}
static inline uint64_t splitmix(uint64_t &s) {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}
int main(int argc, char **argv) {
    std::string mode = argc > 1 ? argv[1] : "quick";
    if (mode == "eval" || mode == "evalmine") {
        const bool mine = mode == "evalmine";
        std::string fn = argv[2];
        std::vector<float> in;
        float buf[4096];
        size_t n;
        while ((n = fread(buf, 4, 4096, stdin)) > 0) in.insert(in.end(), buf, buf + n);
        std::vector<float> out;
        if (fn == "atan2") {
            for (size_t i = 0; i + 1 < in.size(); i += 2) out.push_back(mine ? glibc235::atan2f(in[i], in[i + 1]) : atan2f(in[i], in[i + 1]));
        } else {
            F1 f = nullptr;
            for (const Fn &e : fns) if (fn == e.name) f = mine ? e.mine : e.ref;
            if (!f) { fprintf(stderr, "unknown function %s\n", fn.c_str()); return 2; }
            for (float v : in) out.push_back(f(v));
        }
        fwrite(out.data(), 4, out.size(), stdout);
        return 0;
    }
    const bool exhaustive = mode == "exhaustive";
    const uint64_t stride = exhaustive ? 1 : 256;
    const int nt = (int)std::thread::hardware_concurrency();
    int bad_total = 0;
    for (const Fn &e : fns) {
        std::atomic<uint64_t> bad{0};
        std::atomic<uint32_t> first{0};
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                uint64_t lo = (1ull << 32) * t / nt, hi = (1ull << 32) * (t + 1) / nt;
                for (uint64_t u = lo + (stride > 1 ? (t * 37) % stride : 0); u < hi; u += stride) {
                    float x = glibc235::asfloat((uint32_t)u);
                    if (!same(e.mine(x), e.ref(x))) {
                        if (bad.fetch_add(1) == 0) first = (uint32_t)u;
                    }
                }
            });
        for (auto &x : th) x.join();
        printf("%-6s mismatches %llu", e.name, (unsigned long long)bad.load());
        if (bad) {
            float x = glibc235::asfloat(first);
            printf("  first at 0x%08x (%a): mine %a ref %a", first.load(), x, e.mine(x), e.ref(x));
        }
        printf("\n");
        bad_total += bad != 0;
    }
    {   // atan2f: seeded pairs — uniform bit patterns, plus pairs of comparable magnitude (the path's use)
        std::atomic<uint64_t> bad{0};
        const uint64_t N = exhaustive ? (1ull << 32) : (1ull << 24);
        std::vector<std::thread> th;
        for (int t = 0; t < nt; ++t)
            th.emplace_back([&, t] {
                uint64_t s = 0x1234 + t;
                for (uint64_t i = 0; i < N / nt; ++i) {
                    uint64_t r = splitmix(s);
                    float y = glibc235::asfloat((uint32_t)r), x = glibc235::asfloat((uint32_t)(r >> 32));
                    if (i & 1) {  // comparable magnitudes: exponents within +-8 of each other around 1
                        uint32_t ey = 119 + ((r >> 23) & 15), ex = 119 + ((r >> 55) & 15);
                        y = glibc235::asfloat(((uint32_t)r & 0x807fffffu) | (ey << 23));
                        x = glibc235::asfloat(((uint32_t)(r >> 32) & 0x807fffffu) | (ex << 23));
                    }
                    if (!same(glibc235::atan2f(y, x), atan2f(y, x))) {
                        if (bad.fetch_add(1) == 0) fprintf(stderr, "atan2 first mismatch y=%a x=%a mine %a ref %a\n", y, x, glibc235::atan2f(y, x), atan2f(y, x));
                    }
                }
            });
        for (auto &x : th) x.join();
        printf("atan2  mismatches %llu of %llu pairs\n", (unsigned long long)bad.load(), (unsigned long long)N);
        bad_total += bad != 0;
    }
    return bad_total ? 1 : 0;
}
