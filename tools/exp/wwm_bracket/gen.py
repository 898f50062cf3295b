#!/usr/bin/env python3
"""tools/exp/wwm_bracket/gen.py seed NP NC NV WAVES out_prefix — a random kernel under SGPR and VGPR pressure, and the same arithmetic for the host.

NP uniform pointers + NC uniform floats are live across the whole body (more scalars than there are SGPRs: they are spilled through the lanes
of carrier VGPRs), NV per-lane values are live across it (more than the launch bound leaves VGPRs: carriers get split and spilled too),
inside randomly nested divergent branches and variable-trip loops.  Only + - * / sqrt fma fabs: correctly rounded on both sides, so
out_prefix.hip (kernel k_syn) and out_prefix_ref.cpp (ref_syn) must agree bit for bit.  DESIGN.md 4.6."""
import random
import sys

seed, NP, NC, NV, WAVES = [int(x) for x in sys.argv[1:6]]
prefix = sys.argv[6]
r = random.Random(seed)
body = []


def stmt(ind):
    a, b, c = r.randrange(NV), r.randrange(NV), r.randrange(NV)
    k = r.randrange(6)
    if k == 0:
        return ind + 'v%d = v%d * P.c[%d] + P.p[%d][(i + %d) & 1023];' % (a, b, r.randrange(NC), r.randrange(NP), r.randrange(99))
    if k == 1:
        return ind + 'v%d += helper(P.p[%d], v%d, (int)(fabsf(v%d)) & 3) / (fabsf(v%d) + P.c[%d]);' % (a, r.randrange(NP), b, c, c, r.randrange(NC))
    if k == 2:
        return ind + 'v%d = sqrtf(fabsf(v%d * P.c[%d])) - P.p[%d][i & 255];' % (a, b, r.randrange(NC), r.randrange(NP))
    if k == 3:
        return ind + 'v%d = v%d / (1.f + fabsf(P.p[%d][(i * 3) & 511])) + P.c[%d];' % (a, b, r.randrange(NP), r.randrange(NC))
    if k == 4:
        return ind + 'v%d = (v%d - v%d) * P.c[%d] + v%d;' % (a, b, c, r.randrange(NC), c)
    return ind + 'v%d = fmaf(v%d, v%d, P.c[%d]);' % (a, b, c, r.randrange(NC))


def block(depth, ind):
    for _ in range(r.randrange(3, 9)):
        k = r.random()
        if depth < 4 and k < 0.25:
            body.append(ind + 'if (v%d > v%d * P.c[%d]) {' % (r.randrange(NV), r.randrange(NV), r.randrange(NC)))
            block(depth + 1, ind + '  ')
            if r.random() < 0.6:
                body.append(ind + '} else {')
                block(depth + 1, ind + '  ')
            body.append(ind + '}')
        elif depth < 3 and k < 0.35:
            body.append(ind + 'for (int t%d = 0; t%d < ((int)fabsf(v%d * 5.f) & 7); ++t%d) {' % (depth, depth, r.randrange(NV), depth))
            block(depth + 1, ind + '  ')
            body.append(ind + '}')
        else:
            for _ in range(r.randrange(2, 10)):
                body.append(stmt(ind))


for v in range(NV):
    body.append('    float v%d = P.p[%d][i] * P.c[%d];' % (v, r.randrange(NP), r.randrange(NC)))
block(0, '    ')
body.append('    float acc = 0;')
for v in range(NV):
    body.append('    acc += v%d * P.c[%d];' % (v, r.randrange(NC)))
body.append('    P.out[i] = acc;')
params = 'struct Params { const float *p[%d]; float c[%d]; int n; float *out; };' % (NP, NC)
helper = 'float helper(const float *q, float x, int k) { float s = x; for (int i = 0; i < k; ++i) s = s * q[i & 7] + sqrtf(fabsf(s) + 1.f); return s; }'
open(prefix + '.hip', 'w').write('\n'.join(
    ['// generated: tools/exp/wwm_bracket/gen.py %s' % ' '.join(sys.argv[1:6]), '#include <hip/hip_runtime.h>', params, '__device__ __noinline__ ' + helper,
     'extern "C" __global__ void __launch_bounds__(256, %d) k_syn(Params P) {' % WAVES,
     '  for (int base = blockIdx.x * 256; base < P.n; base += gridDim.x * 256) {', '    const int i = base + threadIdx.x;', '    if (i >= P.n) continue;'] + body + ['  }', '}', '']))
open(prefix + '_ref.cpp', 'w').write('\n'.join(
    ['// generated: the host side of %s.hip' % prefix.split('/')[-1], '#include <cmath>', params, 'static __attribute__((noinline)) ' + helper,
     'extern "C" int ref_np() { return %d; }' % NP, 'extern "C" int ref_nc() { return %d; }' % NC,
     'extern "C" void ref_syn(const Params *Pp) {', '  const Params &P = *Pp;', '  for (int i = 0; i < P.n; ++i) {'] + body + ['  }', '}', '']))
