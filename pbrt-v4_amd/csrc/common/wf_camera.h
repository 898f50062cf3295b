// wf_camera.h — sampler, pixel filter, camera: everything GenerateCameraRays / GenerateRaySamples
// evaluate per pixel sample (wavefront/camera.cpp:31-80, wavefront/samples.cpp:29-66).
#pragma once

#include "wf_scene.h"
#include "wf_bxdf.h"
#include "wf_shapes.h"

namespace wf {

// ---------------------------------------------------------------------------------------------
// Sobol' sample generation (util/lowdiscrepancy.h:165-180) with the scramblers (:223-268)
WF_HD uint32_t FastOwenScramble(uint32_t v, uint32_t seed) {
    v = ReverseBits32(v);
    v ^= v * 0x3d20adea;
    v += seed;
    v *= (seed >> 16) | 1;
    v ^= v * 0x05526c56;
    v ^= v * 0x53a22864;
    return ReverseBits32(v);
}
WF_HD uint32_t OwenScramble(uint32_t v, uint32_t seed) {
    if (seed & 1) v ^= 1u << 31;
    for (int b = 1; b < 32; ++b) {
        uint32_t mask = (~0u) << (32 - b);
        if ((uint32_t)MixBits((v & mask) ^ seed) & (1u << b)) v ^= 1u << (31 - b);
    }
    return v;
}
WF_HD float SobolSample(const uint32_t *sobol, int64_t a, int dimension, int randomize, uint32_t hash) {
    uint32_t v = 0;
    if (((uint64_t)a >> 32) == 0) {
        // the same XOR of columns, four index bytes at a time (tables: FillSobol2D)
        const uint32_t *lut = sobol + WF_SOBOL_COLUMNS + dimension * 1024;
        const uint32_t x = (uint32_t)a;
        v = lut[x & 255u] ^ lut[256 + ((x >> 8) & 255u)] ^ lut[512 + ((x >> 16) & 255u)] ^ lut[768 + (x >> 24)];
    } else {
        for (int i = dimension * 52; a != 0; a >>= 1, i++)
            if (a & 1) v ^= sobol[i];
    }
    if (randomize == WF_RAND_PERMUTE_DIGITS) v = hash ^ v;
    else if (randomize == WF_RAND_FAST_OWEN) v = FastOwenScramble(v, hash);
    else if (randomize == WF_RAND_OWEN) v = OwenScramble(v, hash);
    return fmin(v * 0x1p-32f, OneMinusEpsilon);
}

// ZSobolSampler (samplers.h:225-370)
struct ZSobol {
    const uint32_t *sobol;
    int log2spp, nBase4Digits, randomize, seed;
    int dimension;
    uint64_t mortonIndex;

    WF_HD ZSobol(const SceneView &sv)
        : sobol(sv.sobol), log2spp(sv.sampler.log2spp), nBase4Digits(sv.sampler.nBase4Digits),
          randomize(sv.sampler.randomize), seed(sv.sampler.seed), dimension(0), mortonIndex(0) {}
    WF_HD void StartPixelSample(int px, int py, int index, int dim) {
        dimension = dim;
        mortonIndex = (EncodeMorton2((uint32_t)px, (uint32_t)py) << log2spp) | (uint64_t)index;
    }
    WF_HD static int Perm(int p, int digit) {
        // the 24 permutations of {0,1,2,3} in the reference's order (samplers.h:296-322), packed 2 bits
        // per element, element 0 in the low bits
        const uint8_t perms[24] = {
            0xE4, 0xB4, 0xD8, 0x78, 0x6C, 0x9C, 0xE1, 0xB1, 0xC9, 0x39, 0x2D, 0x8D,
            0xC6, 0x36, 0xD2, 0x72, 0x4E, 0x1E, 0x27, 0x87, 0x1B, 0x4B, 0x63, 0x93};
        return (perms[p] >> (2 * digit)) & 3;
    }
    // (MixBits(higherDigits ^ dimMix) >> 24) % 24 (samplers.h:331).  The arithmetic is integer, so any exact
    // rearrangement gives the reference's value; for operands below 2^32 (every image up to 65536^2 pixels at any
    // practical spp) the first multiplication is 32 x 64 bit and the 40-bit modulo reduces through 2^32 = 2^16 =
    // 2^8 = 1 (mod 3) to a 10-bit one — no 64-bit division on the device.
    WF_HD static int PermutationIndex(uint64_t higherDigits, uint32_t dimMix) {
        uint64_t v = higherDigits ^ dimMix;
        if ((v >> 32) != 0) return (int)((MixBits(v) >> 24) % 24);
        uint32_t x = (uint32_t)v;
        x ^= x >> 31;
        uint64_t m = (uint64_t)x * 0x7fb5d329728ea185ull;
        m ^= m >> 27;
        m *= 0x81dadef4bc2dd44dull;
        m ^= m >> 33;
        const uint64_t y = m >> 24;            // 40 bits
        const uint32_t low3 = (uint32_t)y & 7u;
        const uint64_t z = y >> 3;             // y % 24 = 8 * (z % 3) + low3
        const uint32_t zl = (uint32_t)z, zh = (uint32_t)(z >> 32);
        uint32_t s = (zl & 0xffffu) + (zl >> 16) + zh;  // = z (mod 3), < 2^17 + 32
        s = (s & 0xffu) + (s >> 8);                     // < 2^8 + 2^9 + 1
        const uint32_t q = (s * 43691u) >> 17;          // s / 3 for s < 2^16
        return (int)((s - 3u * q) * 8u + low3);
    }
    // Digits iHi .. iLo (inclusive) of the permuted index (the loop of samplers.h:324-337)
    WF_HD uint64_t IndexDigits(int iHi, int iLo) const {
        uint64_t sampleIndex = 0;
        const bool pow2Samples = log2spp & 1;
        for (int i = iHi; i >= iLo; --i) {
            int digitShift = 2 * i - (pow2Samples ? 1 : 0);
            int digit = (int)((mortonIndex >> digitShift) & 3);
            uint64_t higherDigits = mortonIndex >> (digitShift + 2);
            int p = PermutationIndex(higherDigits, 0x55555555u * (uint32_t)dimension);
            digit = Perm(p, digit);
            sampleIndex |= uint64_t(digit) << digitShift;
        }
        return sampleIndex;
    }
    // Digits whose value AND whose permutation depend on the pixel only (digitShift >= log2spp): the same for
    // every sample index of the pixel, so the device computes them once per pixel, dimension and pass
    // (KSampleTops) instead of once per ray.
    WF_HD int SplitDigit() const { return (log2spp + 1) / 2; }
    WF_HD uint64_t TopDigits() const { return IndexDigits(nBase4Digits - 1, SplitDigit()); }
    WF_HD uint64_t LowDigits() const {
        const bool pow2Samples = log2spp & 1;
        uint64_t sampleIndex = IndexDigits(SplitDigit() - 1, pow2Samples ? 1 : 0);
        if (pow2Samples) {
            int digit = (int)(mortonIndex & 1);
            sampleIndex |= (uint64_t)(digit ^ (int)(MixBits((mortonIndex >> 1) ^ (0x55555555u * (uint32_t)dimension)) & 1));
        }
        return sampleIndex;
    }
    bool haveTop = false;
    uint64_t top = 0;
    // the next Get*() call uses `t` as its TopDigits() (t must be TopDigits() of this pixel at the current dimension)
    WF_HD void SetTop(uint64_t t) { haveTop = true; top = t; }
    WF_HD uint64_t GetSampleIndex() {
        uint64_t t = haveTop ? top : TopDigits();
        haveTop = false;
        return t | LowDigits();
    }
    WF_HD float Get1D() {
        uint64_t sampleIndex = GetSampleIndex();
        ++dimension;
        uint32_t sampleHash = (uint32_t)Hash2i(dimension, seed);
        return SobolSample(sobol, (int64_t)sampleIndex, 0, randomize, sampleHash);
    }
    WF_HD V2 Get2D() {
        uint64_t sampleIndex = GetSampleIndex();
        dimension += 2;
        uint64_t bits = Hash2i(dimension, seed);
        uint32_t h0 = (uint32_t)bits, h1 = (uint32_t)(bits >> 32);
        return V2{SobolSample(sobol, (int64_t)sampleIndex, 0, randomize, h0),
                  SobolSample(sobol, (int64_t)sampleIndex, 1, randomize, h1)};
    }
    WF_HD V2 GetPixel2D() { return Get2D(); }
};

// The sampler of the scene behind the Sampler interface the stages use (base/sampler.h:40-64): ZSobol (above),
// IndependentSampler (samplers.h:442-482), StratifiedSampler (:503-575), PaddedSobolSampler (:130-222).
// RadicalInverse / ScrambledRadicalInverse / OwenScrambledRadicalInverse, util/lowdiscrepancy.h:86-159
WF_HD float RadicalInverseBase(unsigned base, uint64_t a) {
    uint64_t limit = ~0ull / base - base;
    float invBase = (float)1 / (float)base, invBaseM = 1;
    uint64_t reversedDigits = 0;
    while (a && reversedDigits < limit) {
        uint64_t next = a / base;
        uint64_t digit = a - next * base;
        reversedDigits = reversedDigits * base + digit;
        invBaseM *= invBase;
        a = next;
    }
    return fmin(reversedDigits * invBaseM, OneMinusEpsilon);
}
WF_HD float ScrambledRadicalInverseBase(unsigned base, uint64_t a, const uint16_t *perm) {
    uint64_t limit = ~0ull / base - base;
    float invBase = (float)1 / (float)base, invBaseM = 1;
    uint64_t reversedDigits = 0;
    int digitIndex = 0;
    while (1 - (base - 1) * invBaseM < 1 && reversedDigits < limit) {
        uint64_t next = a / base;
        int digitValue = (int)(a - next * base);
        reversedDigits = reversedDigits * base + perm[digitIndex * base + digitValue];
        invBaseM *= invBase;
        ++digitIndex;
        a = next;
    }
    return fmin(invBaseM * reversedDigits, OneMinusEpsilon);
}
WF_HD float OwenScrambledRadicalInverseBase(unsigned base, uint64_t a, uint32_t hash) {
    uint64_t limit = ~0ull / base - base;
    float invBase = (float)1 / (float)base, invBaseM = 1;
    uint64_t reversedDigits = 0;
    while (1 - invBaseM < 1 && reversedDigits < limit) {
        uint64_t next = a / base;
        int digitValue = (int)(a - next * base);
        uint32_t digitHash = (uint32_t)MixBits(hash ^ reversedDigits);
        digitValue = PermutationElement((uint32_t)digitValue, base, digitHash);
        reversedDigits = reversedDigits * base + digitValue;
        invBaseM *= invBase;
        a = next;
    }
    return fmin(invBaseM * reversedDigits, OneMinusEpsilon);
}
WF_HD uint64_t InverseRadicalInverse(uint64_t inverse, int base, int nDigits) {
    uint64_t index = 0;
    for (int i = 0; i < nDigits; ++i) {
        uint64_t digit = inverse % base;
        inverse /= base;
        index = index * base + digit;
    }
    return index;
}

struct PixelSampler {
    ZSobol z;
    int type, spp, seed, randomize, xs, ys, jitter;
    const uint32_t *sobol;
    RNG rng;
    int px = 0, py = 0, sampleIndex = 0, dimension = 0;
    // HaltonSampler (samplers.h:33-141)
    int hbs0, hbs1, hbe0, hbe1, hmi0, hmi1;  // baseScales, baseExponents, multInverse
    const int32_t *primes, *permOffsets;
    const uint16_t *perms;
    int64_t haltonIndex = 0;
    // SobolSampler (samplers.h:479-565)
    const uint32_t *sobolM;
    const uint64_t *vdc, *vdcInv;
    int sobolScale;
    int64_t sobolIndex = 0;

    WF_HD PixelSampler(const SceneView &sv)
        : z(sv), type(sv.sampler.type), spp(sv.sampler.spp), seed(sv.sampler.seed), randomize(sv.sampler.randomize),
          xs(sv.sampler.x_samples), ys(sv.sampler.y_samples), jitter(sv.sampler.jitter), sobol(sv.sobol),
          hbs0(sv.sampler.halton_base_scales[0]), hbs1(sv.sampler.halton_base_scales[1]), hbe0(sv.sampler.halton_base_exponents[0]),
          hbe1(sv.sampler.halton_base_exponents[1]), hmi0(sv.sampler.halton_mult_inverse[0]), hmi1(sv.sampler.halton_mult_inverse[1]),
          primes(sv.haltonPrimes), permOffsets(sv.haltonPermOffsets), perms(sv.haltonPerms), sobolM(sv.sobolMatrices), vdc(sv.vdcSobol),
          vdcInv(sv.vdcSobolInv), sobolScale(sv.sampler.sobol_scale) {}
    // SobolSample over the full table (util/lowdiscrepancy.h:167-181) + SobolSampler::SampleDimension (samplers.h:544-557)
    WF_HD float SobolDimension(int dim) const {
        uint32_t v = 0;
        int64_t a = sobolIndex;
        for (int i = dim * 52; a != 0; a >>= 1, i++)
            if (a & 1) v ^= sobolM[i];
        if (randomize != WF_RAND_NONE) {
            uint32_t hash = (uint32_t)Hash2i(dim, seed);
            if (randomize == WF_RAND_PERMUTE_DIGITS) v = hash ^ v;
            else if (randomize == WF_RAND_FAST_OWEN) v = FastOwenScramble(v, hash);
            else v = OwenScramble(v, hash);
        }
        return fmin(v * 0x1p-32f, OneMinusEpsilon);
    }
    WF_HD float HaltonDimension(int dim) const {
        unsigned base = (unsigned)primes[dim];
        if (randomize == WF_RAND_NONE) return RadicalInverseBase(base, (uint64_t)haltonIndex);
        if (randomize == WF_RAND_PERMUTE_DIGITS) return ScrambledRadicalInverseBase(base, (uint64_t)haltonIndex, perms + permOffsets[dim]);
        return OwenScrambledRadicalInverseBase(base, (uint64_t)haltonIndex, (uint32_t)MixBits(1 + ((uint64_t)dim << 4)));
    }
    WF_HD static uint64_t HashPixelSeed(int x, int y, int sd) { uint32_t w[3] = {(uint32_t)x, (uint32_t)y, (uint32_t)sd}; return HashWords(w, 3); }
    WF_HD static uint64_t HashPixelDimSeed(int x, int y, int dim, int sd) {
        uint32_t w[4] = {(uint32_t)x, (uint32_t)y, (uint32_t)dim, (uint32_t)sd};
        return HashWords(w, 4);
    }
    WF_HD void StartPixelSample(int x, int y, int index, int dim) {
        if (type == WF_SAMPLER_ZSOBOL) { z.StartPixelSample(x, y, index, dim); return; }
        if (type == WF_SAMPLER_HALTON) {
            // samplers.h:53-71
            haltonIndex = 0;
            int sampleStride = hbs0 * hbs1;
            if (sampleStride > 1) {
                auto Mod = [](int a, int b) { int r = a - (a / b) * b; return r < 0 ? r + b : r; };
                int pm[2] = {Mod(x, 128), Mod(y, 128)};
                for (int i = 0; i < 2; ++i) {
                    uint64_t dimOffset = InverseRadicalInverse((uint64_t)pm[i], i == 0 ? 2 : 3, i == 0 ? hbe0 : hbe1);
                    haltonIndex += (int64_t)(dimOffset * (uint64_t)(sampleStride / (i == 0 ? hbs0 : hbs1)) * (uint64_t)(i == 0 ? hmi0 : hmi1));
                }
                haltonIndex %= sampleStride;
            }
            haltonIndex += index * sampleStride;
            dimension = dim > 2 ? dim : 2;
            return;
        }
        px = x; py = y; sampleIndex = index; dimension = dim;
        if (type == WF_SAMPLER_SOBOL) {
            // samplers.h:502-506 + SobolIntervalToIndex (util/lowdiscrepancy.h:266-286)
            dimension = dim > 2 ? dim : 2;
            uint32_t m = 0;
            while ((1 << (m + 1)) <= sobolScale) ++m;  // Log2Int(scale)
            uint64_t frame = (uint64_t)index;
            if (m == 0) { sobolIndex = (int64_t)frame; return; }
            const uint32_t m2 = m << 1;
            uint64_t idx = frame << m2;
            uint64_t delta = 0;
            for (int c = 0; frame; frame >>= 1, ++c)
                if (frame & 1) delta ^= vdc[(m - 1) * 52 + c];
            uint64_t b = ((((uint64_t)((uint32_t)x)) << m) | ((uint32_t)y)) ^ delta;
            for (int c = 0; b; b >>= 1, ++c)
                if (b & 1) idx ^= vdcInv[(m - 1) * 52 + c];
            sobolIndex = (int64_t)idx;
            return;
        }
        if (type != WF_SAMPLER_PADDED_SOBOL) {
            rng.SetSequence(HashPixelSeed(x, y, seed));
            rng.Advance(index * 65536ull + dim);
        }
    }
    WF_HD void SetTop(uint64_t t) { z.SetTop(t); }
    WF_HD float PaddedDim(int dim, uint32_t a, uint32_t hash) const { return SobolSample(sobol, (int64_t)a, dim, randomize, hash); }
    WF_HD float Get1D() {
        if (type == WF_SAMPLER_ZSOBOL) return z.Get1D();
        if (type == WF_SAMPLER_HALTON) {
            if (dimension >= 1000) dimension = 2;
            return HaltonDimension(dimension++);
        }
        if (type == WF_SAMPLER_SOBOL) {
            if (dimension >= 1024) dimension = 2;
            return SobolDimension(dimension++);
        }
        if (type == WF_SAMPLER_INDEPENDENT) return rng.UniformFloat();
        uint64_t hash = HashPixelDimSeed(px, py, dimension, seed);
        if (type == WF_SAMPLER_STRATIFIED) {
            int n = xs * ys;
            int stratum = PermutationElement((uint32_t)sampleIndex, (uint32_t)n, (uint32_t)hash);
            ++dimension;
            float delta = jitter ? rng.UniformFloat() : 0.5f;
            return (stratum + delta) / n;
        }
        int index = PermutationElement((uint32_t)sampleIndex, (uint32_t)spp, (uint32_t)hash);
        dimension++;
        return PaddedDim(0, (uint32_t)index, (uint32_t)(hash >> 32));
    }
    WF_HD V2 Get2D() {
        if (type == WF_SAMPLER_ZSOBOL) return z.Get2D();
        if (type == WF_SAMPLER_HALTON) {
            if (dimension + 1 >= 1000) dimension = 2;
            int dim = dimension;
            dimension += 2;
            float a = HaltonDimension(dim), b = HaltonDimension(dim + 1);
            return V2{a, b};
        }
        if (type == WF_SAMPLER_SOBOL) {
            if (dimension + 1 >= 1024) dimension = 2;
            float a = SobolDimension(dimension), b = SobolDimension(dimension + 1);
            dimension += 2;
            return V2{a, b};
        }
        if (type == WF_SAMPLER_INDEPENDENT) { float a = rng.UniformFloat(); float b = rng.UniformFloat(); return V2{a, b}; }
        uint64_t hash = HashPixelDimSeed(px, py, dimension, seed);
        if (type == WF_SAMPLER_STRATIFIED) {
            int stratum = PermutationElement((uint32_t)sampleIndex, (uint32_t)(xs * ys), (uint32_t)hash);
            dimension += 2;
            int x = stratum % xs, y = stratum / xs;
            float dx = jitter ? rng.UniformFloat() : 0.5f;
            float dy = jitter ? rng.UniformFloat() : 0.5f;
            return V2{(x + dx) / xs, (y + dy) / ys};
        }
        int index = PermutationElement((uint32_t)sampleIndex, (uint32_t)spp, (uint32_t)hash);
        dimension += 2;
        return V2{PaddedDim(0, (uint32_t)index, (uint32_t)hash), PaddedDim(1, (uint32_t)index, (uint32_t)(hash >> 32))};
    }
    WF_HD V2 GetPixel2D() {
        if (type == WF_SAMPLER_HALTON)
            return V2{RadicalInverseBase(2, (uint64_t)(haltonIndex >> hbe0)), RadicalInverseBase(3, (uint64_t)(haltonIndex / hbs1))};
        if (type == WF_SAMPLER_SOBOL) {
            // samplers.h:525-536: dimensions 0 and 1 unscrambled, remapped to the pixel
            const int keep = randomize;
            randomize = WF_RAND_NONE;
            float u0 = SobolDimension(0), u1 = SobolDimension(1);
            randomize = keep;
            return V2{Clamp(u0 * sobolScale - px, 0.f, OneMinusEpsilon), Clamp(u1 * sobolScale - py, 0.f, OneMinusEpsilon)};
        }
        return Get2D();
    }
};

// ---------------------------------------------------------------------------------------------
// Filter::Sample (filters.h): box/triangle analytic, the others through FilterSampler (filters.h:26-45)
// = PiecewiseConstant2D::Sample (util/sampling.h:760-770) over PiecewiseConstant1D::Sample (:657-675).
// (PC1DSample: wf_shapes.h)
WF_HD float SampleTent(float u, float r) {
    // util/sampling.h:247-258
    // SampleDiscrete({0.5, 0.5}, u, nullptr, &u)
    float up = u * 1.f;  // sum of weights 0.5 + 0.5
    if (up == 1.f) up = NextFloatDown(up);
    int offset = 0;
    float sum = 0;
    while (sum + 0.5f <= up) { sum += 0.5f; ++offset; }
    float ur = fmin((up - sum) / 0.5f, OneMinusEpsilon);
    if (offset == 0) return -r + r * SampleLinear(ur, 0, 1);
    else return r * SampleLinear(ur, 1, 0);
}
struct FilterSampleR { V2 p; float weight; };
WF_HD FilterSampleR FilterSample(const SceneView &sv, V2 u) {
    const wf_filter &F = sv.filter;
    if (F.type == WF_FILTER_BOX) {
        // filters.h:67-70
        return {V2{Lerp(u.x, -F.radius[0], F.radius[0]), Lerp(u.y, -F.radius[1], F.radius[1])}, 1.f};
    }
    if (F.type == WF_FILTER_TRIANGLE) {
        // filters.h TriangleFilter::Sample
        return {V2{SampleTent(u.x, F.radius[0]), SampleTent(u.y, F.radius[1])}, 1.f};
    }
    const float *D = sv.filterData;
    float pdf1, pdf0;
    int iv, iu;
    float d1 = PC1DSample(D + F.marg_func_offset, D + F.marg_cdf_offset, F.ny, F.marg_int, F.domain_min[1],
                          F.domain_max[1], u.y, &pdf1, &iv);
    float d0 = PC1DSample(D + F.cond_func_offset + iv * F.nx, D + F.cond_cdf_offset + iv * (F.nx + 1), F.nx,
                          D[F.cond_int_offset + iv], F.domain_min[0], F.domain_max[0], u.x, &pdf0, &iu);
    float pdf = pdf0 * pdf1;
    return {V2{d0, d1}, D[F.f_offset + iv * F.nx + iu] / pdf};
}

// ---------------------------------------------------------------------------------------------
// transforms applied on the device (util/transform.h:310-348, 133-176)
WF_HD V3 XfPoint(const float m[4][4], V3 p) {
    float xp = m[0][0] * p.x + m[0][1] * p.y + m[0][2] * p.z + m[0][3];
    float yp = m[1][0] * p.x + m[1][1] * p.y + m[1][2] * p.z + m[1][3];
    float zp = m[2][0] * p.x + m[2][1] * p.y + m[2][2] * p.z + m[2][3];
    float wp = m[3][0] * p.x + m[3][1] * p.y + m[3][2] * p.z + m[3][3];
    if (wp == 1) return V3{xp, yp, zp};
    return V3{xp, yp, zp} / wp;
}
// Transform::ApplyInverse(Point3f), util/transform.h:386-398 (mInv passed in)
WF_HD V3 XfInvPointM(const float mi[4][4], V3 p) {
    float x = p.x, y = p.y, z = p.z;
    float xp = (mi[0][0] * x + mi[0][1] * y) + (mi[0][2] * z + mi[0][3]);
    float yp = (mi[1][0] * x + mi[1][1] * y) + (mi[1][2] * z + mi[1][3]);
    float zp = (mi[2][0] * x + mi[2][1] * y) + (mi[2][2] * z + mi[2][3]);
    float wp = (mi[3][0] * x + mi[3][1] * y) + (mi[3][2] * z + mi[3][3]);
    if (wp == 1) return V3{xp, yp, zp};
    return V3{xp, yp, zp} / wp;
}
WF_HD V3 XfVector(const float m[4][4], V3 v) {
    return V3{m[0][0] * v.x + m[0][1] * v.y + m[0][2] * v.z, m[1][0] * v.x + m[1][1] * v.y + m[1][2] * v.z,
              m[2][0] * v.x + m[2][1] * v.y + m[2][2] * v.z};
}
WF_HD N3 XfNormal(const float mInv[4][4], N3 n) {
    return N3{mInv[0][0] * n.x + mInv[1][0] * n.y + mInv[2][0] * n.z, mInv[0][1] * n.x + mInv[1][1] * n.y + mInv[2][1] * n.z,
              mInv[0][2] * n.x + mInv[1][2] * n.y + mInv[2][2] * n.z};
}
// Interval addition of an exact float to [lo,hi] with outward rounding, then the midpoint
// (util/math.h:873-875 with the host AddRoundDown/Up = NextFloatDown/Up(a+b), util/float.h:204-216)
WF_HD float IntervalAddMid(float lo, float hi, float v) {
    float l = NextFloatDown(lo + v), h = NextFloatUp(hi + v);
    return (l + h) / 2;
}
// Transform::operator()(const Ray &) for an exact origin: util/transform.h:336-348
WF_HD void XfRay(const float m[4][4], V3 *o, V3 *d) {
    float x = o->x, y = o->y, z = o->z;
    float xp = (m[0][0] * x + m[0][1] * y) + (m[0][2] * z + m[0][3]);
    float yp = (m[1][0] * x + m[1][1] * y) + (m[1][2] * z + m[1][3]);
    float zp = (m[2][0] * x + m[2][1] * y) + (m[2][2] * z + m[2][3]);
    float wp = (m[3][0] * x + m[3][1] * y) + (m[3][2] * z + m[3][3]);
    V3 pe;
    pe.x = gamma(3) * (abs(m[0][0] * x) + abs(m[0][1] * y) + abs(m[0][2] * z) + abs(m[0][3]));
    pe.y = gamma(3) * (abs(m[1][0] * x) + abs(m[1][1] * y) + abs(m[1][2] * z) + abs(m[1][3]));
    pe.z = gamma(3) * (abs(m[2][0] * x) + abs(m[2][1] * y) + abs(m[2][2] * z) + abs(m[2][3]));
    P3i oi = MakeP3i(V3{xp, yp, zp}, pe);
    (void)wp;  // affine transforms only (wp == 1): camera and identity motion
    V3 dd = XfVector(m, *d);
    float lengthSquared = LengthSquared(dd);
    if (lengthSquared > 0) {
        float dt = Dot(Abs(dd), oi.err()) / lengthSquared;
        V3 off = dd * dt;
        o->x = IntervalAddMid(oi.lo.x, oi.hi.x, off.x);
        o->y = IntervalAddMid(oi.lo.y, oi.hi.y, off.y);
        o->z = IntervalAddMid(oi.lo.z, oi.hi.z, off.z);
    } else *o = oi.mid();
    *d = dd;
}

// (AnimatedTransform::Interpolate and its helpers: wf_animated.h, shared with the AnimatedPrimitive code of wf_shapes.h)
// CameraTransform::RenderFromCamera(time) as a Transform: the interpolated transformation where the camera moves, the static one otherwise
// (AnimatedTransform::operator()(Ray) / (Point3f, time) / ApplyInverse(..., time): util/transform.cpp:964-1014, util/transform.h:459-477
// all reduce to Interpolate(time), which returns the end transforms outside (startTime, endTime))
WF_HD void CameraRenderFromCameraAt(const wf_camera &C, float time, wf_transform *out) {
    if (!C.anim.actually_animated) { *out = C.renderFromCamera; return; }
    AnimatedInterpolateP(&C.anim, time, out);
}

// ---------------------------------------------------------------------------------------------
// CameraBase::Approximate_dp_dxy (cameras.h:155-183) with RotateFromTo (util/transform.h:249-270): the texture
// footprint of a surface point, from the camera's minimum ray differentials (FindMinimumDifferentials,
// cameras.cpp:153-203, evaluated on the host)
// ANIM: compiled with the moving-camera path (CameraFromRender / RenderFromCamera at the ray's time); without it the static
// renderFromCamera is used — the back end selects a kernel variant compiled with it when the scene's camera moves
template <bool ANIM = true>
WF_HD void ApproximateDpDxy(const SceneView &sv, V3 p, N3 n, V3 *dpdx, V3 *dpdy, float time = 0) {
    const wf_camera &C = sv.camera;
    const float (*rfcM)[4] = C.renderFromCamera.m;
    const float (*rfcInv)[4] = C.renderFromCamera.mInv;
    wf_transform moving;
    if constexpr (ANIM)
        if (C.anim.actually_animated) {
            AnimatedInterpolateP(&C.anim, time, &moving);
            rfcM = moving.m; rfcInv = moving.mInv;
        }
    V3 pCamera = XfInvPointM(rfcInv, p);  // CameraFromRender(p, time)
    // RotateFromTo(Normalize(pCamera), (0, 0, 1))
    V3 from = Normalize(pCamera), to{0, 0, 1};
    V3 refl;
    if (abs(from.x) < 0.72f && abs(to.x) < 0.72f) refl = V3{1, 0, 0};
    else if (abs(from.y) < 0.72f && abs(to.y) < 0.72f) refl = V3{0, 1, 0};
    else refl = V3{0, 0, 1};
    V3 u = refl - from, v = refl - to;
    const float uu[3] = {u.x, u.y, u.z}, vv[3] = {v.x, v.y, v.z};
    float r[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            r[i][j] = ((i == j) ? 1 : 0) - 2 / Dot(u, u) * uu[i] * uu[j] - 2 / Dot(v, v) * vv[i] * vv[j] +
                      4 * Dot(u, v) / (Dot(u, u) * Dot(v, v)) * vv[i] * uu[j];
    auto fwdPoint = [&](V3 q) {  // Transform::operator()(Point3f): the fourth row and column are (0, 0, 0, 1)
        return V3{r[0][0] * q.x + r[0][1] * q.y + r[0][2] * q.z + 0.f, r[1][0] * q.x + r[1][1] * q.y + r[1][2] * q.z + 0.f,
                  r[2][0] * q.x + r[2][1] * q.y + r[2][2] * q.z + 0.f};
    };
    V3 pDownZ = fwdPoint(pCamera);
    // CameraFromRender(n, time) = renderFromCamera.ApplyInverse(Normal3f): m transposed (util/transform.h:409-415)
    const float (*m)[4] = rfcM;
    N3 nCam{m[0][0] * n.x + m[1][0] * n.y + m[2][0] * n.z, m[0][1] * n.x + m[1][1] * n.y + m[2][1] * n.z,
            m[0][2] * n.x + m[1][2] * n.y + m[2][2] * n.z};
    // DownZFromCamera(Normal3f): mInv transposed = r (mInv = Transpose(r))
    N3 nDownZ{r[0][0] * nCam.x + r[0][1] * nCam.y + r[0][2] * nCam.z, r[1][0] * nCam.x + r[1][1] * nCam.y + r[1][2] * nCam.z,
              r[2][0] * nCam.x + r[2][1] * nCam.y + r[2][2] * nCam.z};
    float d = nDownZ.z * pDownZ.z;
    V3 xo = V3{0, 0, 0} + V3{C.minPosDifferentialX[0], C.minPosDifferentialX[1], C.minPosDifferentialX[2]};
    V3 xd = V3{0, 0, 1} + V3{C.minDirDifferentialX[0], C.minDirDifferentialX[1], C.minDirDifferentialX[2]};
    float tx = -(Dot(nDownZ, xo) - d) / Dot(nDownZ, xd);
    V3 yo = V3{0, 0, 0} + V3{C.minPosDifferentialY[0], C.minPosDifferentialY[1], C.minPosDifferentialY[2]};
    V3 yd = V3{0, 0, 1} + V3{C.minDirDifferentialY[0], C.minDirDifferentialY[1], C.minDirDifferentialY[2]};
    float ty = -(Dot(nDownZ, yo) - d) / Dot(nDownZ, yd);
    V3 px = xo + xd * tx, py = yo + yd * ty;
    float sppScale = sv.options.disable_pixel_jitter ? 1.f : fmax(.125f, 1 / sqrt((float)sv.sampler.spp));
    auto invVec = [&](V3 q) {  // DownZFromCamera.ApplyInverse(Vector3f): mInv * q = r^T q
        return V3{r[0][0] * q.x + r[1][0] * q.y + r[2][0] * q.z, r[0][1] * q.x + r[1][1] * q.y + r[2][1] * q.z,
                  r[0][2] * q.x + r[1][2] * q.y + r[2][2] * q.z};
    };
    *dpdx = sppScale * XfVector(rfcM, invVec(px - pDownZ));
    *dpdy = sppScale * XfVector(rfcM, invVec(py - pDownZ));
}


// ---------------------------------------------------------------------------------------------
// RealisticCamera (cameras.h:466-580, cameras.cpp:695-951): rays traced from the film through the lens prescription
struct LensElement { float curvatureRadius, thickness, eta, apertureRadius; };
WF_HD LensElement LoadLensElement(const float *table, const wf_camera &C, int i) {
    const float *e = table + C.lens_offset + 4 * i;
    return LensElement{e[0], e[1], e[2], e[3]};
}
// RealisticCamera::IntersectSphericalElement (cameras.h:538-560)
WF_HD bool IntersectSphericalElement(float radius, float zCenter, V3 ro, V3 rd, float *t, N3 *n) {
    V3 o = ro - V3{0, 0, zCenter};
    float A = rd.x * rd.x + rd.y * rd.y + rd.z * rd.z;
    float B = 2 * (rd.x * o.x + rd.y * o.y + rd.z * o.z);
    float C = o.x * o.x + o.y * o.y + o.z * o.z - radius * radius;
    float t0, t1;
    if (!QuadraticF(A, B, C, &t0, &t1)) return false;
    bool useCloserT = (rd.z > 0) ^ (radius < 0);
    *t = useCloserT ? fmin(t0, t1) : fmax(t0, t1);
    if (*t < 0) return false;
    V3 nv = o + *t * rd;
    *n = FaceForward(Normalize(toN(nv)), -rd);
    return true;
}
// RealisticCamera::TraceLensesFromFilm (cameras.cpp:749-813); returns the weight (0 = blocked)
WF_HD float TraceLensesFromFilm(const SceneView &sv, const wf_camera &C, V3 co, V3 cd, V3 *oOut, V3 *dOut) {
    const float *table = sv.tableData;
    float elementZ = 0, weight = 1;
    V3 ro{co.x, co.y, -co.z}, rd{cd.x, cd.y, -cd.z};
    for (int i = C.n_lens_elements - 1; i >= 0; --i) {
        const LensElement element = LoadLensElement(table, C, i);
        elementZ -= element.thickness;
        float t;
        N3 n{0, 0, 0};
        bool isStop = (element.curvatureRadius == 0);
        if (isStop) {
            t = (elementZ - ro.z) / rd.z;
            if (t < 0) return 0;
        } else {
            float radius = element.curvatureRadius;
            float zCenter = elementZ + element.curvatureRadius;
            if (!IntersectSphericalElement(radius, zCenter, ro, rd, &t, &n)) return 0;
        }
        V3 pHit = ro + rd * t;
        if (isStop && C.aperture_image >= 0) {
            V2 uv{(pHit.x / element.apertureRadius + 1) / 2, (pHit.y / element.apertureRadius + 1) / 2};
            weight = ImageBilerpChannel(table, sv.texImages[C.aperture_image], 0, uv, 0);   // WrapMode::Black
            if (weight == 0) return 0;
        } else {
            if (Sqr(pHit.x) + Sqr(pHit.y) > Sqr(element.apertureRadius)) return 0;
        }
        ro = pHit;
        if (!isStop) {
            V3 w;
            float eta_i = element.eta;
            float eta_t = 1;
            if (i > 0) { float e = table[C.lens_offset + 4 * (i - 1) + 2]; if (e != 0) eta_t = e; }
            if (!Refract(Normalize(-rd), n, eta_t / eta_i, nullptr, &w)) return 0;
            rd = w;
        }
    }
    if (oOut) { *oOut = V3{ro.x, ro.y, -ro.z}; *dOut = V3{rd.x, rd.y, -rd.z}; }
    return weight;
}
// RealisticCamera::TraceLensesFromScene (cameras.cpp:959-1007), used at load to focus the lens
WF_HD float TraceLensesFromScene(const SceneView &sv, const wf_camera &C, V3 co, V3 cd, V3 *oOut, V3 *dOut) {
    const float *table = sv.tableData;
    float front = 0;
    for (int i = 0; i < C.n_lens_elements; ++i) front += table[C.lens_offset + 4 * i + 1];
    float elementZ = -front;
    // LensFromCamera = Scale(1, 1, -1) applied to the ray: Transform::operator()(Ray) moves the origin along d by its rounding error bound
    V3 ro, rd;
    {
        const float m[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, -1, 0}, {0, 0, 0, 1}};
        XfRayOffset(m, co, cd, &ro, &rd);
    }
    for (int i = 0; i < C.n_lens_elements; ++i) {
        const LensElement element = LoadLensElement(table, C, i);
        float t;
        N3 n{0, 0, 0};
        bool isStop = (element.curvatureRadius == 0);
        if (isStop) {
            t = (elementZ - ro.z) / rd.z;
            if (t < 0) return 0;
        } else {
            float radius = element.curvatureRadius;
            float zCenter = elementZ + element.curvatureRadius;
            if (!IntersectSphericalElement(radius, zCenter, ro, rd, &t, &n)) return 0;
        }
        V3 pHit = ro + rd * t;
        float r2 = pHit.x * pHit.x + pHit.y * pHit.y;
        if (r2 > element.apertureRadius * element.apertureRadius) return 0;
        ro = pHit;
        if (!isStop) {
            V3 wt;
            float eta_i = 1;
            if (i > 0) { float e = table[C.lens_offset + 4 * (i - 1) + 2]; if (e != 0) eta_i = e; }
            float eta_t = (element.eta != 0) ? element.eta : 1;
            if (!Refract(Normalize(-rd), n, eta_t / eta_i, nullptr, &wt)) return 0;
            rd = wt;
        }
        elementZ += element.thickness;
    }
    if (oOut) { *oOut = V3{ro.x, ro.y, -ro.z}; *dOut = V3{rd.x, rd.y, -rd.z}; }
    return 1;
}
// ---------------------------------------------------------------------------------------------
// GetCameraSample (samplers.h:796-814) + Perspective/OrthographicCamera::GenerateRay
// (cameras.cpp:404-428, 283-307) + the identity "movingFromCamera" the wavefront loop applies
// (wavefront/camera.cpp:64, integrator.cpp:364-368)
// CameraTransform::RenderFromCamera(Ray) = AnimatedTransform::operator()(const Ray &) (util/transform.cpp:964-973)
// (ANIM = false: a kernel instance for scenes whose camera does not move — the interpolation's registers and scratch stay out of it)
template <bool ANIM = true>
WF_HD void CameraXfRay(const wf_camera &C, float time, V3 *o, V3 *d) {
    if (ANIM && C.anim.actually_animated && time > C.anim.start_time) {
        if (time >= C.anim.end_time) XfRay(C.anim.end.m, o, d);
        else {
            wf_transform t;
            AnimatedInterpolateP(&C.anim, time, &t);
            XfRay(t.m, o, d);
        }
    } else XfRay(C.renderFromCamera.m, o, d);
}
struct CameraRayR { V3 o, d; float time; bool valid; float weight = 1; };
// applyMoving: the identity "movingFromCamera" transform of the wavefront loop (it still walks the origin off its rounding-error
// bound); Camera::GenerateRay itself, as FindMinimumDifferentials calls it at load, does not have it
template <bool ANIM = true>
WF_HD CameraRayR GenerateCameraRay(const SceneView &sv, V2 pFilm, float timeSample, V2 pLens, bool applyMoving = true) {
    const wf_camera &C = sv.camera;
    if (C.type == WF_CAMERA_REALISTIC) {
        // RealisticCamera::GenerateRay + SampleExitPupil (cameras.cpp:897-951)
        V2 s{pFilm.x / sv.film.full_res[0], pFilm.y / sv.film.full_res[1]};
        V2 pFilm2{(1 - s.x) * C.physical_extent[0] + s.x * C.physical_extent[2], (1 - s.y) * C.physical_extent[1] + s.y * C.physical_extent[3]};
        V3 pF{-pFilm2.x, pFilm2.y, 0};
        const float lensRearZ = sv.tableData[C.lens_offset + 4 * (C.n_lens_elements - 1) + 1];
        float rFilm = sqrt(Sqr(pF.x) + Sqr(pF.y));
        int rIndex = (int)(rFilm / (C.film_diagonal / 2) * C.n_exit_pupil_bounds);
        rIndex = rIndex < C.n_exit_pupil_bounds - 1 ? rIndex : C.n_exit_pupil_bounds - 1;
        const auto pb = sv.tableData + C.exit_pupil_offset + 4 * rIndex;
        CameraRayR none{V3{0, 0, 0}, V3{0, 0, 0}, 0, false, 0};
        if (pb[0] >= pb[2] || pb[1] >= pb[3]) return none;   // Bounds2::IsDegenerate
        V2 pLensS{(1 - pLens.x) * pb[0] + pLens.x * pb[2], (1 - pLens.y) * pb[1] + pLens.y * pb[3]};
        float pdf = 1 / ((pb[2] - pb[0]) * (pb[3] - pb[1]));
        float sinTheta = (rFilm != 0) ? pF.y / rFilm : 0;
        float cosTheta = (rFilm != 0) ? pF.x / rFilm : 1;
        V3 pPupil{cosTheta * pLensS.x - sinTheta * pLensS.y, sinTheta * pLensS.x + cosTheta * pLensS.y, lensRearZ};
        V3 fd = pPupil - pF;
        V3 o, d;
        float weight = TraceLensesFromFilm(sv, C, pF, fd, &o, &d);
        if (weight == 0) return none;
        float time = Lerp(timeSample, C.shutterOpen, C.shutterClose);
        CameraXfRay<ANIM>(C, time, &o, &d);
        d = Normalize(d);
        float cosT = Normalize(fd).z;
        weight *= Sqr(Sqr(cosT)) / (pdf * Sqr(lensRearZ));
        if (applyMoving) {
            const float I[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
            XfRay(I, &o, &d);
        }
        return {o, d, time, true, weight};
    }
    if (C.type == WF_CAMERA_SPHERICAL) {
        // SphericalCamera::GenerateRay, cameras.cpp:610-630
        V2 uv{pFilm.x / sv.film.full_res[0], pFilm.y / sv.film.full_res[1]};
        V3 dir;
        if (C.spherical_mapping == 1) {
            float theta = Pi * uv.y, phi = 2 * Pi * uv.x;
            dir = SphericalDirection(sin(theta), cos(theta), phi);
        } else {
            // WrapEqualAreaSquare, util/math.cpp:363-379
            if (uv.x < 0) { uv.x = -uv.x; uv.y = 1 - uv.y; }
            else if (uv.x > 1) { uv.x = 2 - uv.x; uv.y = 1 - uv.y; }
            if (uv.y < 0) { uv.x = 1 - uv.x; uv.y = -uv.y; }
            else if (uv.y > 1) { uv.x = 1 - uv.x; uv.y = 2 - uv.y; }
            dir = EqualAreaSquareToSphere(uv);
        }
        { float t = dir.y; dir.y = dir.z; dir.z = t; }
        V3 so{0, 0, 0};
        float stime = Lerp(timeSample, C.shutterOpen, C.shutterClose);
        CameraXfRay<ANIM>(C, stime, &so, &dir);
        const float I[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
        XfRay(I, &so, &dir);
        return {so, dir, stime, true};
    }
    V3 pCamera = XfPoint(C.cameraFromRaster.m, V3{pFilm.x, pFilm.y, 0});
    V3 o, d;
    if (C.type == WF_CAMERA_PERSPECTIVE) {
        o = V3{0, 0, 0};
        d = Normalize(pCamera);
    } else {
        o = pCamera;
        d = V3{0, 0, 1};
    }
    float time = Lerp(timeSample, C.shutterOpen, C.shutterClose);
    if (C.lensRadius > 0) {
        V2 dl = SampleUniformDiskConcentric(pLens);
        V2 pl{C.lensRadius * dl.x, C.lensRadius * dl.y};
        float ft = C.focalDistance / d.z;
        V3 pFocus = o + d * ft;
        o = V3{pl.x, pl.y, 0};  // both projective cameras (cameras.cpp:301,422)
        d = Normalize(pFocus - o);
    }
    CameraXfRay<ANIM>(C, time, &o, &d);
    // movingFromCamera == identity Transform
    const float I[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
    XfRay(I, &o, &d);
    return {o, d, time, true};
}

// One record of the sampler probes (wf_sampler_probe, wf_cpu --sampler-probe) from a sampler positioned by StartPixelSample:
// mode > 0: that many Get1D(); -2: GetPixel2D(); -3: ten times (Get2D, Get1D) — the sequence Sampler.ConsistentValues draws
// (samplers_test.cpp:46-52), 30 floats; -4: the ZSobol sample index (ZSobolSampler.ValidIndices, :168-196), low / high word as the bit
// patterns of two floats.
WF_HD void SamplerProbeRecord(PixelSampler &s, int mode, float *out) {
    if (mode == -2) { V2 p = s.GetPixel2D(); out[0] = p.x; out[1] = p.y; }
    else if (mode == -3) {
        for (int k = 0; k < 10; ++k) { V2 p = s.Get2D(); out[3 * k] = p.x; out[3 * k + 1] = p.y; out[3 * k + 2] = s.Get1D(); }
    } else if (mode == -4) {
        const uint64_t idx = s.z.GetSampleIndex();
        out[0] = BitsToFloat((uint32_t)idx); out[1] = BitsToFloat((uint32_t)(idx >> 32));
    } else
        for (int d = 0; d < mode; ++d) out[d] = s.Get1D();
}

}  // namespace wf
