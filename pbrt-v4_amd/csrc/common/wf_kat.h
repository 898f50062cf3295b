// wf_kat.h — known-answer probe of the leaf routines the reference's unit tests pin exactly (util/rng_test.cpp,
// util/hash_test.cpp, shapes_test.cpp Triangle.BadCases): one record in (16 uint64), one record out (8 uint64), evaluated by the restated
// routines of wf_math.h / wf_shapes.h on the host (oracle/_build/wf_kat) and on the device (wf_kat_probe, include/wf_abi.h).  The records
// and their reference answers are tests/golden/kat_{in,out}.bin, written by oracle/ref_build/ref_kat.cpp with the reference's own RNG,
// HashBuffer / Hash / HashFloat / MixBits and IntersectTriangle.  Layouts (in[0] = test):
//   0 / 1  RNG: in {-, sequence index, seed, 0 one-argument SetSequence / 1 two-argument / 2 default-constructed, Advance() argument};
//          out: the next 8 Uniform<uint32_t>() (0) or the bits of the next 8 Uniform<float>() (1)
//   2      RNG::operator-: in {-, sequence index, draws of a, draws of b}; out {a - b, b - a}
//   3      HashBuffer: in {-, length (<= 96), byte offset of the data inside the data area (0..7), data area ...}; out {hash, MixBits(first word)}
//   4      Hash(args...): in {-, kind, argument bits ...}: 0 Hash(Point2i), 1 Hash(Point3f), 2 Hash(Point3f, Vector3f), 3 Hash(Point3f, Float) and
//          Hash(Vector3f); out {hash, bits of HashFloat(same), second hash}
//   5      IntersectTriangle: in {-, 16 floats two per word: o, d, tMax, p0, p1, p2}; out {hit, b0, b1, b2, t} as float bits
#pragma once
#include "wf_shapes.h"

namespace wf {

WF_HD void KatRun(const uint64_t *in, uint64_t *out) {
    for (int k = 0; k < 8; ++k) out[k] = 0;
    const int test = (int)in[0];
    auto f32 = [](uint64_t w) { return BitsToFloat((uint32_t)w); };
    if (test == 0 || test == 1) {
        RNG r;
        if (in[3] == 1) r.SetSequence(in[1], in[2]);
        else if (in[3] == 0) r.SetSequence(in[1]);
        if ((int64_t)in[4] != 0) r.Advance((int64_t)in[4]);
        for (int k = 0; k < 8; ++k) out[k] = test == 0 ? (uint64_t)r.Uniform32() : (uint64_t)FloatToBits(r.UniformFloat());
    } else if (test == 2) {
        RNG a, b;
        a.SetSequence(in[1]); b.SetSequence(in[1]);
        for (uint64_t k = 0; k < in[2]; ++k) (void)a.Uniform32();
        for (uint64_t k = 0; k < in[3]; ++k) (void)b.Uniform32();
        out[0] = (uint64_t)(a - b); out[1] = (uint64_t)(b - a);
    } else if (test == 3) {
        const size_t len = (size_t)in[1];
        unsigned char data[104];
        for (int w = 0; w < 13; ++w)
            for (int b = 0; b < 8; ++b) data[8 * w + b] = (unsigned char)(in[3 + w] >> (8 * b));
        const unsigned char *p = data + (int)in[2];
        out[0] = MurmurHash64A(p, len, 0);
        uint64_t first = 0;
        for (size_t b = 0; b < (len < 8 ? len : 8); ++b) first |= (uint64_t)p[b] << (8 * b);
        out[1] = MixBits(first);
    } else if (test == 4) {
        const int kind = (int)in[1];
        if (kind == 0) {
            out[0] = Hash2i((int)(uint32_t)in[2], (int)(uint32_t)in[3]);
            out[1] = FloatToBits(HashToFloat(out[0]));
            out[2] = out[0];
        } else {
            const V3 p{f32(in[2]), f32(in[3]), f32(in[4])}, v{f32(in[5]), f32(in[6]), f32(in[7])};
            if (kind == 1) out[0] = Hash3f(p);
            else if (kind == 2) out[0] = Hash6f(p, v);
            else {
                uint32_t w[4] = {FloatToBits(p.x), FloatToBits(p.y), FloatToBits(p.z), (uint32_t)in[8]};
                out[0] = HashWords(w, 4);
                out[2] = Hash3f(v);
            }
            out[1] = FloatToBits(HashToFloat(out[0]));
        }
    } else if (test == 5) {
        float r[16];
        for (int k = 0; k < 8; ++k) { r[2 * k] = f32(in[1 + k]); r[2 * k + 1] = f32(in[1 + k] >> 32); }
        TriHit h{};
        const bool hit = IntersectTriangle(V3{r[0], r[1], r[2]}, V3{r[3], r[4], r[5]}, r[6], V3{r[7], r[8], r[9]}, V3{r[10], r[11], r[12]}, V3{r[13], r[14], r[15]}, &h);
        out[0] = FloatToBits(hit ? 1.f : 0.f);
        if (hit) { out[1] = FloatToBits(h.b0); out[2] = FloatToBits(h.b1); out[3] = FloatToBits(h.b2); out[4] = FloatToBits(h.t); }
    }
}

}  // namespace wf
