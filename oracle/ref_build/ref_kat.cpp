// oracle/ref_build/ref_kat.cpp — TEST INFRASTRUCTURE ONLY (ours; links the shimmed build of the reference, libpbrt_ref.a).
// Known answers for the leaf routines the reference's own unit tests pin exactly: util/rng_test.cpp (RNG.Reseed / Advance /
// OperatorMinus), util/hash_test.cpp (Hash.VarArgs / Unaligned) and shapes_test.cpp's Triangle.BadCases — computed by the reference's
// RNG, HashBuffer / Hash / HashFloat / MixBits and IntersectTriangle themselves:
//     ref_kat <outdir>  ->  <outdir>/kat_in.bin (records of 16 uint64), <outdir>/kat_out.bin (records of 8 uint64)
// The restated routines (csrc/common/wf_math.h, wf_shapes.h) must reproduce every record bit for bit on the host (oracle/_build/wf_kat)
// and on the device (wf_kat_probe, include/wf_abi.h); tests/test_reference_known_answers.py then runs the reference tests' property
// checks on those outputs.  Record layouts: csrc/common/wf_kat.h.
#include <pbrt/pbrt.h>

#include <pbrt/shapes.h>
#include <pbrt/util/hash.h>
#include <pbrt/util/rng.h>
#include <pbrt/util/vecmath.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

using namespace pbrt;

struct Lcg {  // input generator shared with nothing: inputs are stored, not regenerated
    uint64_t s;
    explicit Lcg(uint64_t seed) : s(seed * 2862933555777941757ull + 3037000493ull) {}
    uint32_t u32() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (uint32_t)(s >> 32); }
    uint64_t u64() { uint64_t a = u32(); return (a << 32) | u32(); }
    float f01() { return (u32() >> 8) * (1.f / 16777216.f); }
    float range(float a, float b) { return a + (b - a) * f01(); }
};

static uint64_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float bitsf(uint64_t u) { uint32_t v = (uint32_t)u; float f; memcpy(&f, &v, 4); return f; }

int main(int argc, char **argv) {
    if (argc < 2) { fprintf(stderr, "usage: ref_kat <outdir>\n"); return 1; }
    std::vector<uint64_t> in, out;
    auto rec = [&](uint64_t test) { in.resize(in.size() + 16, 0); out.resize(out.size() + 8, 0); in[in.size() - 16] = test; };
    auto I = [&]() { return &in[in.size() - 16]; };
    auto O = [&]() { return &out[out.size() - 8]; };
    Lcg g(20260924);
    // ---- test 0 / 1: RNG — in {test, sequence index, seed, seed given, Advance() argument}; out: the next 8 Uniform<uint32_t>() (test 0) or
    //      the bits of the next 8 Uniform<float>() (test 1).  The sequences of util/rng_test.cpp first (1234; 1234 / 6502; 32; 1337).
    auto rng = [&](int test, uint64_t seq, uint64_t seed, bool haveSeed, int64_t adv) {
        rec(test);
        I()[1] = seq; I()[2] = seed; I()[3] = haveSeed; I()[4] = (uint64_t)adv;
        RNG r;
        if (haveSeed) r.SetSequence(seq, seed); else r.SetSequence(seq);
        if (adv) r.Advance(adv);
        for (int k = 0; k < 8; ++k) O()[k] = test == 0 ? (uint64_t)r.Uniform<uint32_t>() : fbits(r.Uniform<float>());
    };
    for (int adv = 0; adv < 100; adv += 8) rng(0, 1234, 0, false, adv);
    for (int adv : {0, 5, 16, 37, 552, 992}) rng(1, 1234, 6502, true, adv);
    rng(1, 32, 0, false, 0);
    rng(0, 1337, 0, false, 0);
    for (int i = 0; i < 200; ++i) rng(i & 1, g.u64() >> (g.u32() % 64), g.u64(), g.u32() & 1, (int64_t)(g.u32() % 100000) - (i % 5 == 0 ? 50000 : 0));
    {   // the default-constructed generator (RNG rng; in the tests)
        rec(0);
        I()[3] = 2;
        RNG r;
        for (int k = 0; k < 8; ++k) O()[k] = r.Uniform<uint32_t>();
    }
    // ---- test 2: RNG::operator- — in {test, sequence index, draws of a, draws of b}; out {a - b, b - a}
    for (int i = 0; i < 64; ++i) {
        rec(2);
        const uint64_t seq = i < 8 ? 1337 : g.u64();
        const int na = 1 + g.u32() % 1000, nb = i % 3 == 0 ? 0 : g.u32() % 1000;
        I()[1] = seq; I()[2] = na; I()[3] = nb;
        RNG a(seq), b(seq);
        for (int k = 0; k < na; ++k) (void)a.Uniform<uint32_t>();
        for (int k = 0; k < nb; ++k) (void)b.Uniform<uint32_t>();
        O()[0] = (uint64_t)(a - b); O()[1] = (uint64_t)(b - a);
    }
    // ---- test 3: HashBuffer — in {test, length in bytes (<= 96), byte offset of the data inside the record's data area (0..7), data ...};
    //      out {HashBuffer(data, length), MixBits(first data word)}.  hash_test.cpp's buffers first.
    auto hashbuf = [&](const void *data, size_t len, int delta) {
        rec(3);
        I()[1] = len; I()[2] = delta;
        memcpy((char *)(I() + 3) + delta, data, len);
        uint64_t first = 0;
        memcpy(&first, data, len < 8 ? len : 8);
        O()[0] = HashBuffer((const char *)(I() + 3) + delta, len);
        O()[1] = MixBits(first);
    };
    {
        int64_t buf[] = {1, -12511, 31415821, 37};
        for (int i = 0; i < 4; ++i) {
            hashbuf(buf + i, 8, 0);
            if (O()[0] != Hash(buf[i])) { fprintf(stderr, "ref_kat: Hash.VarArgs does not hold in the reference build\n"); return 2; }
        }
        uint64_t ubuf[] = {0xfacebeef, 0x65028088, 0x13372048};
        for (int delta = 0; delta < 8; ++delta) hashbuf(ubuf, sizeof(ubuf), delta);
    }
    for (int i = 0; i < 300; ++i) {
        unsigned char d[96];
        for (unsigned char &c : d) c = (unsigned char)g.u32();
        hashbuf(d, i < 97 ? i : g.u32() % 97, g.u32() % 8);
    }
    // ---- test 4: Hash(args...) of the argument shapes the path uses, and HashFloat — in {test, kind, float or int bits ...};
    //      kind 0: Hash(int, int) (samplers.h:261); 1: Hash(Point3f) (lights.h:498); 2: Hash(Point3f, Vector3f) (cpu/primitive.cpp:60);
    //      3: Hash(Point3f, Float) + Hash(Vector3f) (media.cpp:44).  out {hash, bits of HashFloat(same args), second hash of kind 3}
    for (int i = 0; i < 400; ++i) {
        rec(4);
        const int kind = i % 4;
        I()[1] = kind;
        float f[7];
        for (float &x : f) x = i % 16 < 4 ? (float)(int)g.range(-3, 3) : g.range(-1000, 1000) * (g.u32() & 1 ? 1e-3f : 1.f);
        int a = (int)g.u32() % 4096, b = (int)g.u32() % 4096;
        if (kind == 0) {
            I()[2] = (uint64_t)(uint32_t)a; I()[3] = (uint64_t)(uint32_t)b;
            O()[0] = Hash(Point2i(a, b)); O()[1] = fbits(HashFloat(Point2i(a, b)));
            O()[2] = Hash(a, b);
        } else {
            for (int k = 0; k < 7; ++k) I()[2 + k] = fbits(f[k]);
            Point3f p(f[0], f[1], f[2]);
            Vector3f v(f[3], f[4], f[5]);
            if (kind == 1) { O()[0] = Hash(p); O()[1] = fbits(HashFloat(p)); }
            else if (kind == 2) { O()[0] = Hash(p, v); O()[1] = fbits(HashFloat(p, v)); }
            else { O()[0] = Hash(p, f[6]); O()[1] = fbits(HashFloat(p, f[6])); O()[2] = Hash(v); }
        }
    }
    // ---- test 5: IntersectTriangle — in {test, o.xy, o.z d.x, d.yz, tMax p0.x, p0.yz, p1.xy, p1.z p2.x, p2.yz} (two floats per word);
    //      out {hit, b0, b1, b2, t} as float bits.  Triangle.BadCases (shapes_test.cpp:435-449) first: must miss.
    auto tri = [&](const float r[16]) {
        rec(5);
        for (int k = 0; k < 8; ++k) I()[1 + k] = fbits(r[2 * k]) | (fbits(r[2 * k + 1]) << 32);
        Ray ray(Point3f(r[0], r[1], r[2]), Vector3f(r[3], r[4], r[5]));
        auto ti = IntersectTriangle(ray, r[6], Point3f(r[7], r[8], r[9]), Point3f(r[10], r[11], r[12]), Point3f(r[13], r[14], r[15]));
        O()[0] = fbits(ti ? 1.f : 0.f);
        if (ti) { O()[1] = fbits(ti->b0); O()[2] = fbits(ti->b1); O()[3] = fbits(ti->b2); O()[4] = fbits(ti->t); }
    };
    {
        const float bad[16] = {-1081.47925f, 99.9999542f, 87.7701111f, -32.1072998f, -183.355865f, -144.607635f, Infinity,
                               -1113.45459f, -79.049614f, -56.2431908f, -1113.45459f, -87.0922699f, -56.2431908f, -1113.45459f, -79.2090149f, -56.2431908f};
        tri(bad);
        if (bitsf(O()[0]) != 0.f) { fprintf(stderr, "ref_kat: Triangle.BadCases does not hold in the reference build\n"); return 2; }
        // rays through the vertices and along the edges of random thin triangles (what Triangle.Watertight shoots at)
        for (int i = 0; i < 600; ++i) {
            float r[16];
            for (int k = 7; k < 16; ++k) r[k] = g.range(-10, 10) * (i % 3 == 0 ? 1e-3f : 1.f);
            for (int k = 0; k < 3; ++k) r[k] = g.range(-0.5f, 0.5f);
            const int v = i % 3, w = (i + 1) % 3;
            float t = i % 2 ? g.f01() : 0.f;   // a vertex, or a point on the edge v -> w
            for (int k = 0; k < 3; ++k) r[3 + k] = ((1 - t) * r[7 + 3 * v + k] + t * r[7 + 3 * w + k]) - r[k];
            r[6] = Infinity;
            tri(r);
        }
    }
    const std::string dir = argv[1];
    FILE *f = fopen((dir + "/kat_in.bin").c_str(), "wb");
    fwrite(in.data(), 8, in.size(), f);
    fclose(f);
    f = fopen((dir + "/kat_out.bin").c_str(), "wb");
    fwrite(out.data(), 8, out.size(), f);
    fclose(f);
    printf("%zu records\n", out.size() / 8);
    return 0;
}
