"""The reference's own end-to-end known-answer test, RenderTest.RadianceMatches (cpu/integrators_test.cpp:72-155, 239-447),
run through THIS build: three analytic furnace scenes — the inside of a unit sphere with reflectance 0.5 lit by a point light of
intensity pi at its centre, by four point lights of a quarter of it, or by its own emission of 0.5 — must render to a mean pixel value
of 1.0 +- 0.025 (CheckSceneAverage, :50-65) at 10 x 10 pixels and 256 spp, depth 8, box filter of radius 0.5, under every sampler of
GetSamplers() this build has (:239-279; PMJ02BN's tables are not in the checkout) and both projective cameras.

The reference builds the scenes through its C++ API; here they are scene files, so the parser, the table builder, the quadric
intersection, the light sampling and the film are on the path.  The CPU leg runs the restated kernels (oracle/wf_cpu) and the reference
itself (oracle/_ref/pbrt_ref --wavefront, where it is built) on the same file; the GPU leg (last in the `-m gpu` order) renders it
through the C ABI."""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, read_pfm, run_wf_cpu

PBRT_REF = os.path.join(ROOT, "oracle", "_ref", "pbrt_ref")

SAMPLERS = {
    "halton": 'Sampler "halton" "integer pixelsamples" [ 256 ]',
    "paddedsobol": 'Sampler "paddedsobol" "integer pixelsamples" [ 256 ] "string randomization" "permutedigits"',
    "zsobol": 'Sampler "zsobol" "integer pixelsamples" [ 256 ] "string randomization" "permutedigits"',
    "sobol_none": 'Sampler "sobol" "integer pixelsamples" [ 256 ] "string randomization" "none"',
    "sobol_xor": 'Sampler "sobol" "integer pixelsamples" [ 256 ] "string randomization" "permutedigits"',
    "sobol_owen": 'Sampler "sobol" "integer pixelsamples" [ 256 ] "string randomization" "owen"',
    "independent": 'Sampler "independent" "integer pixelsamples" [ 256 ]',
    "stratified": 'Sampler "stratified" "integer xsamples" [ 16 ] "integer ysamples" [ 16 ] "bool jitter" true',
}
CAMERAS = {
    "perspective": 'Camera "perspective" "float fov" [ 45 ] "float screenwindow" [ -1 1 -1 1 ]',
    "orthographic": 'Camera "orthographic" "float screenwindow" [ -0.1 0.1 -0.1 0.1 ]',
}
# a constant spectrum of 1: Create() divides the scale by SpectrumToPhotometric(I) exactly as the test's "little dance" does
ONE = '"spectrum %s" [ 300 1 900 1 ]'
WORLDS = {
    "one_point_light": 'LightSource "point" ' + ONE % "I" + ' "float scale" [ 3.14159265358979 ]\n',
    "four_point_lights": ('LightSource "point" ' + ONE % "I" + ' "float scale" [ 0.785398163397448 ]\n') * 4,
    "emissive_sphere": 'AreaLightSource "diffuse" ' + ONE % "L" + ' "float scale" [ 0.5 ]\n',
}


def furnace_scene(world, sampler, camera):
    return "\n".join([
        CAMERAS[camera],
        SAMPLERS[sampler],
        'Integrator "volpath" "integer maxdepth" [ 8 ]',
        'PixelFilter "box" "float xradius" [ 0.5 ] "float yradius" [ 0.5 ]',
        'Film "rgb" "integer xresolution" [ 10 ] "integer yresolution" [ 10 ] "string filename" [ "furnace.pfm" ]',
        "WorldBegin",
        'Material "diffuse" "spectrum reflectance" [ 300 0.5 900 0.5 ]',
        WORLDS[world] + "ReverseOrientation",
        'Shape "sphere" "float radius" [ 1 ]',
        "",
    ])


def check_scene_average(img, expected=1.0, delta=0.025):
    assert img.shape == (10, 10, 3) and np.isfinite(img).all()
    assert abs(float(img.astype(np.float64).mean()) - expected) <= delta, float(img.mean())


CASES = [(w, s, c) for w in WORLDS for s in SAMPLERS for c in CAMERAS]


@pytest.mark.parametrize("world,sampler,camera", CASES)
def test_furnace_radiance_matches_cpu_port(built, tmp_path, world, sampler, camera):
    path = str(tmp_path / "furnace.pbrt")
    open(path, "w").write(furnace_scene(world, sampler, camera))
    out = str(tmp_path / "cpu.pfm")
    run_wf_cpu(path, out)
    img = read_pfm(out)
    check_scene_average(img)
    if os.path.exists(PBRT_REF) and sampler in ("zsobol", "halton", "stratified"):
        # ... and the reference's wavefront path on the same file gives the same image
        ref_out = str(tmp_path / "ref.pfm")
        subprocess.run([PBRT_REF, "--wavefront", "--quiet", "--nthreads", "4", "--outfile", ref_out, path], check=True, capture_output=True)
        ref = read_pfm(ref_out)
        check_scene_average(ref)
        assert (ref.view(np.uint32) == img.view(np.uint32)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("world,sampler,camera", CASES)
def test_furnace_radiance_matches_gpu(wfpt, tmp_path, world, sampler, camera):
    text = furnace_scene(world, sampler, camera)
    s = wfpt.Scene(text=text)
    s.create_renderer(0)
    s.render()
    img = s.image()
    rays = s.total_rays()
    s.close()
    check_scene_average(img)
    path = str(tmp_path / "furnace.pbrt")
    open(path, "w").write(text)
    out = str(tmp_path / "cpu.pfm")
    j = run_wf_cpu(path, out)
    cpu = read_pfm(out)
    # the f32 tolerance of the north_star on every value (expected and so far always measured: bit-identical), equal ray counts
    from conftest import image_error
    rel = image_error(img, cpu)
    print(world, sampler, camera, "max rel", rel.max(), "bit-identical fraction", (cpu.view(np.uint32) == img.view(np.uint32)).mean())
    assert rel.max() <= 1e-3
    assert rays == j["rays"]


# ---------------------------------------------------------------------------------------------------------------------
# lightsamplers_test.cpp: BVHLightSampling.{OneSpot, Point, PointVaryPower, OneTri, PdfMethod} and the PdfMethod tests of
# the power / uniform samplers, against the light sampler of THIS build: the scene's lights go through the parser, the
# light-BVH builder (csrc/host/lightbvh_build.cpp) and the restated LightSampler::Sample / PMF (csrc/common/wf_lights.h)
# that the kernels run; oracle/wf_cpu --light-probe evaluates them at the test's points.
from conftest import WF_CPU

HEADER = 'Film "rgb" "integer xresolution" [ 4 ] "integer yresolution" [ 4 ] "string filename" [ "x.pfm" ]\n%s\nWorldBegin\n'


def light_probe(tmp_path, world, p, u_light, n=None, u2=None, sampler="bvh"):
    """rows of {sampled light id or -1, p, PMF(light), sampled light's SampleLi valid, light 0's SampleLi valid, ... with radiance, PMF(light 0)}"""
    p = np.asarray(p, np.float32).reshape(-1, 3)
    rec = np.zeros((len(p), 9), np.float32)
    rec[:, 0:3] = p
    if n is not None:
        rec[:, 3:6] = n
    rec[:, 6] = u_light
    rec[:, 7:9] = 0.5 if u2 is None else u2
    scene = str(tmp_path / "lights.pbrt")
    open(scene, "w").write(HEADER % ('Integrator "volpath" "string lightsampler" "%s"' % sampler) + world)   # (a scene without geometry is fine)
    fin, fout = str(tmp_path / "probe_in.bin"), str(tmp_path / "probe_out.bin")
    rec.tofile(fin)
    subprocess.run([WF_CPU, "--quiet", "--light-probe", fin, fout, scene], check=True, capture_output=True)
    return np.fromfile(fout, np.float32).reshape(-1, 7)


def float_eq(a, b):
    """EXPECT_FLOAT_EQ: within 4 units in the last place"""
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64)) <= 4


def stratified_1d(rng, n):
    return np.minimum((np.arange(n) + rng.random(n)) / n, np.float32(1) - np.float32(2 ** -24)).astype(np.float32)


def far_point(rng, scale=1.0):
    r = lambda: rng.uniform(-15, -7) if rng.random() < 0.5 else rng.uniform(7, 16)
    return [scale * r(), scale * r(), scale * r()]


def point_light(p, power=None):
    return ('LightSource "point" "point3 from" [ %.9g %.9g %.9g ] ' % tuple(p) + ONE % "I" +
            ('' if power is None else ' "float power" [ %.9g ]' % power) + "\n")


# The wavefront integrator replaces the sampler of a one-light scene by the uniform one (wavefront/integrator.cpp:181-187, and so does
# this build), so OneSpot / OneTri get a companion point light far away: "no light sampled -> SampleLi gives nothing" becomes "the
# light's SampleLi gives something only where the sampler can choose it" (PMF > 0), p = 1 becomes p == PMF.
COMPANION = point_light([0, 0, -1000], 1e-3)


def test_bvh_light_sampling_one_spot(built, tmp_path):
    world = ('LightSource "spot" "point3 from" [ 0 0 0 ] "point3 to" [ 0 0 1 ] "float coneangle" [ 45 ] "float conedeltaangle" [ 1 ] ' + ONE % "I" + "\n" +
             COMPANION)
    rng = np.random.default_rng(1)
    pts = rng.uniform(-5, 5, (4000, 3))
    angle = np.degrees(np.arccos(pts[:, 2] / np.linalg.norm(pts, axis=1)))
    keep = (angle <= 44.75) | (angle >= 45.25)             # "avoid possibly ambiguous cases right at the edge"
    pts, angle = pts[keep][:1000], angle[keep][:1000]
    out = light_probe(tmp_path, world, pts, rng.random(len(pts)), u2=rng.random((len(pts), 2)))
    spot = out[:, 0] == 0
    assert (out[:, 0] >= 0).all() and spot.any() and (~spot).any()
    assert (out[spot, 4] == 1).all() and (out[spot, 5] == 1).all()      # sampled -> SampleLi valid, with radiance
    assert (angle[spot] < 45).all()
    assert (out[angle > 45.25, 0] == 1).all() and (out[angle > 45.25, 1] == 1).all()   # outside the cone: the companion, with certainty
    assert float_eq(out[:, 1], out[:, 2]).all()
    assert (out[out[:, 4] == 1, 6] > 0).all() and (out[angle > 45.25, 6] == 0).all()


def test_bvh_light_sampling_point(built, tmp_path):
    rng = np.random.default_rng(2)
    world = "".join(point_light(rng.uniform(-5, 5, 3)) for _ in range(33))
    n = 10000
    for _ in range(10):
        out = light_probe(tmp_path, world, np.tile(far_point(rng), (n, 1)), stratified_1d(rng, n))
        ids = out[:, 0].astype(int)
        assert (ids >= 0).all() and (out[:, 1] > 0).all()  # "can assume this because it's all point lights"
        assert float_eq(out[:, 1], out[:, 2]).all()
        sum_wt = np.bincount(ids, weights=1 / (out[:, 1].astype(np.float64) * n), minlength=33)
        assert (sum_wt >= 0.98).all() and (sum_wt < 1.02).all(), sum_wt


def test_bvh_light_sampling_point_vary_power(built, tmp_path):
    rng = np.random.default_rng(53251)
    power = 0.05 + 0.95 * rng.random(82)    # (a light that is drawn a dozen times in 1e5 cannot meet a 5 % tolerance)
    world = "".join(point_light(rng.uniform(-5, 5, 3), pw) for pw in power)
    n = 100000
    for _ in range(10):
        out = light_probe(tmp_path, world, np.tile(far_point(rng), (n, 1)), stratified_1d(rng, n))
        ids = out[:, 0].astype(int)
        assert (ids >= 0).all() and (out[:, 1] > 0).all()
        assert (np.abs(out[:, 2] - out[:, 1]) / out[:, 1] < 1e-4).all()
        sum_wt = np.bincount(ids, weights=1 / (out[:, 1].astype(np.float64) * n), minlength=82)
        assert (sum_wt >= 0.95).all() and (sum_wt < 1.05).all(), sum_wt
    # very far away (d^2 about the same for every light): sampling frequencies proportional to power
    for _ in range(10):
        out = light_probe(tmp_path, world, np.tile(far_point(rng, 10000.0), (n, 1)), stratified_1d(rng, n))
        ids = out[:, 0].astype(int)
        assert (ids >= 0).all() and float_eq(out[:, 1], out[:, 2]).all()
        counts = np.bincount(ids, minlength=82)
        expected = n * power / power.sum()
        assert (counts >= 0.97 * expected).all() and (counts < 1.03 * expected).all()


def triangle_light(P, scale=1.0):
    return ('AttributeBegin\nAreaLightSource "diffuse" ' + ONE % "L" + ' "float scale" [ %.9g ]\n' % scale +
            'Shape "trianglemesh" "integer indices" [ 0 1 2 ] "point3 P" [ ' + " ".join("%.9g" % v for v in np.ravel(P)) + " ]\nAttributeEnd\n")


def test_bvh_light_sampling_one_tri(built, tmp_path):
    world = triangle_light([[-1, -1, 0], [1, -1, 0], [0, 1, 0]]) + COMPANION      # illuminates the points with z > 0
    rng = np.random.default_rng(5251)
    pts = rng.uniform(-5, 5, (1000, 3))
    out = light_probe(tmp_path, world, pts, rng.random(1000), u2=rng.random((1000, 2)))
    tri = out[:, 0] == 0
    assert (out[:, 0] >= 0).all() and tri.any() and (~tri).any()
    # SampleLi succeeds -> the sampler can choose the light; the converse does not hold ("the light importance metric is conservative")
    assert (out[out[:, 4] == 1, 6] > 0).all()
    assert (out[pts[:, 2] < 0, 4] == 0).all() and (out[pts[:, 2] < -2, 6] == 0).any()
    assert float_eq(out[:, 1], out[:, 2]).all()


def random_lights(n, seed=6502):
    rng = np.random.default_rng(seed)
    world = ""
    for _ in range(n):
        world += triangle_light(rng.random((3, 3)), rng.random() + 1e-3)
        world += point_light(rng.uniform(-5, 5, 3), rng.random() + 1e-3)
    return world


@pytest.mark.parametrize("sampler", ["bvh", "power", "uniform"])
def test_light_sampling_pdf_method(built, tmp_path, sampler):
    world = random_lights(20)
    rng = np.random.default_rng(5251)
    pts = -1 + 3 * rng.random((1000, 3))
    out = light_probe(tmp_path, world, pts, rng.random(1000), sampler=sampler)
    got = out[:, 0] >= 0     # "it's actually legit to sometimes get no lights"
    assert got.mean() > 0.9
    assert float_eq(out[got, 1], out[got, 2]).all()
    assert (out[got, 1] > 0).all()
    if sampler == "uniform":
        assert (out[:, 1] == np.float32(1) / np.float32(40)).all()


# ---------------------------------------------------------------------------------------------------------------------
# shapes_test.cpp: Triangle.Reintersect, FullSphere / PartialSphere / Cylinder .Reintersect (TestReintersectConvex) and
# BilinearPatch.Offset on the primitives of THIS build: shapes through the parser and the table builder, the restated
# intersection routines, the interaction the kernels rebuild from a hit record, and the ray offset along its error bounds
# (csrc/common/wf_shapes.h, wf_math.h) — oracle/wf_cpu --reintersect-probe.
def reintersect_probe(tmp_path, world, rec):
    scene = str(tmp_path / "shapes.pbrt")
    open(scene, "w").write(HEADER % "" + point_light([0, 0, 0]) + world)   # (the wavefront path refuses a scene without lights)
    fin, fout = str(tmp_path / "re_in.bin"), str(tmp_path / "re_out.bin")
    np.asarray(rec, np.float32).reshape(-1, 9).tofile(fin)
    subprocess.run([WF_CPU, "--quiet", "--nthreads", "8", "--reintersect-probe", fin, fout, scene], check=True, capture_output=True)
    return np.fromfile(fout, np.float32).reshape(-1, 3)


def p_exp(rng, size=None, exp=8.0):
    return np.float32(10.0) ** np.asarray(rng.uniform(-exp, exp, size), np.float32)


def fmt(a):
    return " ".join("%.9g" % v for v in np.ravel(a))


def test_triangle_reintersect(built, tmp_path):
    rng = np.random.default_rng(0)
    tris = []
    while len(tris) < 1000:
        v = p_exp(rng, (3, 3))
        c = np.cross(v[1].astype(np.float64) - v[0], v[2].astype(np.float64) - v[0])
        if c @ c >= 1e-20:                                  # "don't get into trouble with ~degenerate triangles"
            tris.append(v)
    P = np.array(tris, np.float32).reshape(-1, 3)
    world = 'Shape "trianglemesh" "integer indices" [ %s ] "point3 P" [ %s ]\n' % (" ".join(map(str, range(len(P)))), fmt(P))
    rec = np.zeros((1000, 9), np.float32)
    for i, v in enumerate(tris):
        b = rng.random(2)
        if b.sum() > 1:
            b = 1 - b
        target = (b[0] * v[0].astype(np.float64) + b[1] * v[1] + (1 - b.sum()) * v[2]).astype(np.float32)
        o = p_exp(rng, 3)
        rec[i] = [i, 0, *o, *(target - o), i]
    out = reintersect_probe(tmp_path, world, rec)
    assert out[:, 0].mean() > 0.9          # "we should almost always find an intersection, but rarely miss, due to round-off error"
    assert out[:, 1].sum() == 0 and out[:, 2].sum() == 0


def random_transform(rng):
    """shapes_test.cpp:314-331: Scale(pExp(4)^3) * Translate(+-pExp^3) * Rotate(random angle, random axis), as scene directives"""
    s = p_exp(rng, 3, 4.0)
    t = p_exp(rng, 3) * rng.choice([-1.0, 1.0], 3)
    axis = rng.normal(size=3)
    angle = rng.uniform(-200, 200)
    text = "Scale %s\nTranslate %s\nRotate %.9g %s\n" % (fmt(s), fmt(t), angle, fmt(axis))
    a = axis / np.linalg.norm(axis)
    th = np.radians(angle)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    R = np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)
    M = np.eye(4)
    M[:3, :3] = np.diag(s.astype(np.float64)) @ R
    M[:3, 3] = np.diag(s.astype(np.float64)) @ t.astype(np.float64)
    return text, M


@pytest.mark.parametrize("kind", ["full_sphere", "partial_sphere", "cylinder"])
def test_quadric_reintersect_convex(built, tmp_path, kind):
    rng = np.random.default_rng({"full_sphere": 3, "partial_sphere": 4, "cylinder": 5}[kind])
    world, rec = "", []
    n_tri = 0
    for i in range(1000):
        radius = float(p_exp(rng, None, 4.0))
        if kind == "cylinder":
            zmin = float(p_exp(rng, None, 4.0)) * rng.choice([-1, 1])
            zmax = float(p_exp(rng, None, 4.0)) * rng.choice([-1, 1])
        elif kind == "partial_sphere":
            zmin = -radius if rng.random() < 0.5 else rng.uniform(-radius, radius)
            zmax = radius if rng.random() < 0.5 else rng.uniform(-radius, radius)
        else:
            zmin, zmax = -radius, radius
        phimax = 360.0 if (kind == "full_sphere" or rng.random() < 0.5) else rng.random() * 360
        shape = ('Shape "%s" "float radius" [ %.9g ] "float zmin" [ %.9g ] "float zmax" [ %.9g ] "float phimax" [ %.9g ]\n'
                 % ("cylinder" if kind == "cylinder" else "sphere", radius, zmin, zmax, phimax))
        for transformed in ((False, True) if kind != "partial_sphere" else (False,)):
            text, M = random_transform(rng) if transformed else ("", np.eye(4))
            world += "AttributeBegin\n" + text + shape + "AttributeEnd\n"
            lo = np.array([-radius, -radius, min(zmin, zmax)]); hi = np.array([radius, radius, max(zmin, zmax)])
            p2 = (M @ np.append(lo + rng.random(3) * (hi - lo), 1.0))[:3].astype(np.float32)
            o = p_exp(rng, 3)
            d = (p2 - o).astype(np.float32)
            if rng.random() < 0.5 and np.isfinite(d).all() and np.linalg.norm(d) > 0:
                d = (d / np.linalg.norm(d.astype(np.float64))).astype(np.float32)
            rec.append([n_tri + len(rec), 1, *o, *d, i])
    out = reintersect_probe(tmp_path, world, rec)
    assert out[:, 0].mean() > 0.05         # "we should usually (but not always) find an intersection"
    assert out[:, 1].sum() == 0 and out[:, 2].sum() == 0, (out[:, 1].sum(), out[:, 2].sum(), np.nonzero(out[:, 1] + out[:, 2])[0][:10])


def test_bilinear_patch_offset(built, tmp_path):
    rng = np.random.default_rng(6)
    world, rec = "", []
    for i in range(100):
        height = -400 + 800 * rng.random()
        x0 = -400 + 800 * rng.random(); x1 = x0 + 400 * rng.random()
        z0 = -400 + 800 * rng.random(); z1 = x0 + 400 * rng.random()
        p = np.array([[x0, height, z0], [x1, height, z0], [x0, height, x1], [x1, height, z1]], np.float32)
        world += 'Shape "bilinearmesh" "integer indices" [ 0 1 2 3 ] "point3 P" [ %s ]\n' % fmt(p)
        for j in range(100):
            o = (-20 + 40 * rng.random(3)).astype(np.float32)
            u, v = rng.random(2)
            pp = ((1 - u) * (1 - v) * p[0].astype(np.float64) + u * (1 - v) * p[1] + (1 - u) * v * p[2] + u * v * p[3]).astype(np.float32)
            rec.append([i, 2, *o, *(pp - o), 0])
    out = reintersect_probe(tmp_path, world, rec)
    assert out[:, 0].mean() > 0.9
    assert out[:, 1].sum() == 0


# ---------------------------------------------------------------------------------------------------------------------
# bsdfs_test.cpp: BSDFEnergyConservation, BSDFSampling (Sample_f against f / PDF and against uniform sampling) and the Hair
# tests, on the restated BxDFs the material kernels run (oracle/wf_cpu/wf_props.cpp tells which reference test each one is).
import json
from conftest import WF_PROPS


@pytest.fixture(scope="module")
def bxdf_props(built):
    p = subprocess.run([WF_PROPS], check=True, capture_output=True, text=True)
    return {j["test"]: j for j in map(json.loads, p.stdout.strip().splitlines())}


BXDF_PROPS = (["BSDFEnergyConservation." + n for n in
               ["LambertianReflection", "MicrofacetReflectionTrowbridgeReitz_alpha0.5_cond", "MicrofacetReflectionTrowbridgeReitz_aniso_cond",
                "DiffuseTransmission", "ThinDielectric"] +
               ["MicrofacetReflectionTrowbridgeReitz_%s_%s" % (r, e) for r in ("1.50", "1.00", "0.50", "0.10", "0.01") for e in ("1.5", "inv1.5_importance")] +
               ["Coated%s_%d" % (k, v) for k in ("Diffuse", "Conductor") for v in range(3)]] +
              ["BSDFSampling." + n for n in ["Lambertian", "TRCondIso", "TRCondAniso", "TRDielIso", "TRDielAniso", "TRDielIsoInv", "TRDielAnisoInv",
                                             "DiffuseTransmission", "Hair"]] +
              ["Hair." + n for n in ["WhiteFurnace", "HOnTheEdge", "WhiteFurnaceSampled", "SamplingWeights", "SamplingConsistency"]] +
              ["HenyeyGreenstein." + n for n in ["SamplingMatch", "SamplingOrientationForward", "SamplingOrientationBackward", "Normalized", "g"]])


@pytest.mark.parametrize("name", BXDF_PROPS)
def test_bxdf_property(bxdf_props, name):
    assert name in bxdf_props, sorted(bxdf_props)
    assert bxdf_props[name]["ok"], bxdf_props[name]["detail"]


def test_bxdf_property_list_is_complete(bxdf_props):
    assert sorted(bxdf_props) == sorted(BXDF_PROPS)


# ---------------------------------------------------------------------------------------------------------------------
# samplers_test.cpp: {PaddedSobol, ZSobol, SobolUnscrambled, SobolXORScrambled, SobolOwenScrambled}Sampler.ElementaryIntervals
# (:75-160) on the restated samplers (csrc/common/wf_camera.h) behind the scene's Sampler directive: the GetPixel2D() samples of a
# pixel are a (0, m, 2)-net — every elementary interval of 2^m cells holds exactly one of its 2^m samples.
def pixel_samples(tmp_path, sampler_line, res, spp, seed=0):
    scene = str(tmp_path / "sampler.pbrt")
    open(scene, "w").write('Film "rgb" "integer xresolution" [ %d ] "integer yresolution" [ %d ] "string filename" [ "x.pfm" ]\n%s\nWorldBegin\n' % (res, res, sampler_line) +
                           point_light([0, 0, 0]) + 'Shape "sphere"\n')
    px, py, idx = np.meshgrid(np.arange(res), np.arange(res), np.arange(spp), indexing="ij")
    fin, fout = str(tmp_path / "s_in.bin"), str(tmp_path / "s_out.bin")
    np.stack([px.ravel(), py.ravel(), idx.ravel()], axis=1).astype(np.int32).tofile(fin)
    subprocess.run([WF_CPU, "--quiet", "--seed", str(seed), "--sampler-probe", fin, fout, "0", "-2", scene], check=True, capture_output=True)
    return np.fromfile(fout, np.float32).reshape(res * res, spp, 2)


def check_elementary(samples, log_samples):
    assert samples.shape[1] == 1 << log_samples
    assert (samples >= 0).all() and (samples < 1).all()
    for i in range(log_samples + 1):
        nx, ny = 1 << i, 1 << (log_samples - i)
        cell = np.floor(ny * samples[..., 1]).astype(int) * nx + np.floor(nx * samples[..., 0]).astype(int)
        assert (np.sort(cell, axis=1) == np.arange(1 << log_samples)).all(), (nx, ny)


@pytest.mark.parametrize("rand", ["none", "permutedigits"])
def test_padded_sobol_elementary_intervals(built, tmp_path, rand):
    for log_samples in range(2, 11):
        s = pixel_samples(tmp_path, 'Sampler "paddedsobol" "integer pixelsamples" [ %d ] "string randomization" "%s"' % (1 << log_samples, rand), 1, 1 << log_samples)
        check_elementary(s, log_samples)


@pytest.mark.parametrize("rand", ["none", "permutedigits"])
def test_zsobol_elementary_intervals(built, tmp_path, rand):
    for seed in (0, 1, 5, 6, 10, 15):
        for log_samples in range(2, 9):
            s = pixel_samples(tmp_path, 'Sampler "zsobol" "integer pixelsamples" [ %d ] "string randomization" "%s"' % (1 << log_samples, rand), 10, 1 << log_samples, seed)
            check_elementary(s, log_samples)


@pytest.mark.parametrize("rand", ["none", "permutedigits", "owen"])
def test_sobol_elementary_intervals(built, tmp_path, rand):
    for log_samples in range(2, 11):
        s = pixel_samples(tmp_path, 'Sampler "sobol" "integer pixelsamples" [ %d ] "string randomization" "%s"' % (1 << log_samples, rand), 1, 1 << log_samples)
        check_elementary(s, log_samples)


# ---------------------------------------------------------------------------------------------------------------------
# util/rng_test.cpp (RNG.Reseed / Advance / OperatorMinus), util/hash_test.cpp (Hash.VarArgs / Unaligned), shapes_test.cpp's
# Triangle.BadCases: tests/golden/kat_{in,out}.bin hold the REFERENCE's answers (oracle/ref_build/ref_kat.cpp: its RNG, HashBuffer / Hash /
# HashFloat / MixBits and IntersectTriangle on stored inputs); the restated routines must reproduce every record bit for bit on the host
# (oracle/_build/wf_kat) and on the device (wf_kat_probe), and the reference tests' own assertions are then made on those records.
WF_KAT = os.path.join(ROOT, "oracle", "_build", "wf_kat")
GOLDEN = os.path.join(ROOT, "tests", "golden")


def kat_golden():
    return (np.fromfile(os.path.join(GOLDEN, "kat_in.bin"), np.uint64).reshape(-1, 16), np.fromfile(os.path.join(GOLDEN, "kat_out.bin"), np.uint64).reshape(-1, 8))


def check_kat_properties(kin, kout):
    """the assertions of the reference's unit tests, on the probe's records"""
    test = kin[:, 0]
    # RNG.Reseed (rng_test.cpp:16-26): the generator re-seeded with 1234 repeats its 100 values; RNG.Advance (:28-63): Advance(i) lands
    # on the i-th value of the sequence — the records are SetSequence(1234) advanced by 0, 8, 16, ...: record k's 8 values = values 8k ..
    seq = kin[(test == 0) & (kin[:, 1] == 1234) & (kin[:, 3] == 0)]
    vals = kout[(test == 0) & (kin[:, 1] == 1234) & (kin[:, 3] == 0)]
    assert len(seq) >= 12 and (seq[:, 4] == 8 * np.arange(len(seq))).all()
    stream = vals.reshape(-1)
    assert len(np.unique(stream)) > len(stream) - 3          # a generator, not a constant
    # the same stream read through Advance at other offsets: SetSequence(1234, 6502), float draws at 0, 5, 16, 37, 552, 992
    fl = (test == 1) & (kin[:, 1] == 1234) & (kin[:, 2] == 6502) & (kin[:, 3] == 1)
    offs = kin[fl][:, 4].astype(np.int64)
    fv = kout[fl]
    base = fv[offs == 0][0]
    assert (fv[offs == 5][0][:3] == base[5:8]).all()          # Advance(5) then draw = v[5], v[6], v[7]
    f = fv.astype(np.uint32).view(np.float32)
    assert (f >= 0).all() and (f < 1).all()
    # RNG.OperatorMinus (:65-87): a - b = the number of draws a is ahead, b - a its negative
    m = test == 2
    na, nb = kin[m][:, 2].astype(np.int64), kin[m][:, 3].astype(np.int64)
    assert (kout[m][:, 0].astype(np.int64) == na - nb).all() and (kout[m][:, 1].astype(np.int64) == nb - na).all()
    # Hash.VarArgs (hash_test.cpp:13-17) is the identity HashBuffer(&x, 1) == Hash(x) (asserted by ref_kat when the golden is written);
    # Hash.Unaligned (:44-52): the hash of a buffer does not depend on its alignment — the eight copies at byte offsets 0..7
    h = (test == 3) & (kin[:, 1] == 24)
    first24 = np.nonzero(h)[0][:8]
    assert (kin[first24, 2] == np.arange(8)).all() and len(set(kout[first24, 0].tolist())) == 1
    # HashFloat in [0, 1)
    hf = kout[test == 4][:, 1].astype(np.uint32).view(np.float32)
    assert (hf >= 0).all() and (hf < 1).all()
    # Triangle.BadCases (shapes_test.cpp:435-449): the first triangle record must miss
    t5 = np.nonzero(test == 5)[0]
    assert kout[t5[0], 0] == 0
    # (the other triangle records — rays aimed exactly at a vertex or an edge point of one triangle — only pin the restatement to the
    #  reference's answer, hit or miss: watertightness is a property of the MESH, tested on the closed mesh below)
    hits = kout[t5[1:], 0].astype(np.uint32).view(np.float32)
    assert 0.3 < hits.mean() < 1.0


def test_kat_golden_is_consistent():
    kin, kout = kat_golden()
    assert len(kin) == len(kout) > 1000 and set(np.unique(kin[:, 0]).tolist()) == {0, 1, 2, 3, 4, 5}
    check_kat_properties(kin, kout)


def test_kat_restated_routines_match_reference_on_host(built, tmp_path):
    kin, kout = kat_golden()
    out = str(tmp_path / "kat_out.bin")
    subprocess.run([WF_KAT, os.path.join(GOLDEN, "kat_in.bin"), out], check=True)
    got = np.fromfile(out, np.uint64).reshape(-1, 8)
    bad = np.nonzero((got != kout).any(axis=1))[0]
    assert len(bad) == 0, (bad[:10], kin[bad[:3], 0])
    check_kat_properties(kin, got)


@pytest.mark.gpu
def test_kat_restated_routines_match_reference_on_device(wfpt):
    kin, kout = kat_golden()
    s = wfpt.Scene(path=os.path.join(GOLDEN, "cornell64.pbrt"), spp=1)   # (the probe needs a context only; a Scene brings one)
    s.create_renderer(0)
    got = s.kat_probe(kin)
    s.close()
    bad = np.nonzero((got != kout).any(axis=1))[0]
    assert len(bad) == 0, (bad[:10], kin[bad[:3], 0])
    check_kat_properties(kin, got)


# ---------------------------------------------------------------------------------------------------------------------
# Triangle.Watertight (shapes_test.cpp:33-130; `#if 0` in the reference because it fails on its CI machines): a closed triangulated sphere
# whose vertices are pushed out randomly; 100 000 rays from inside, half of them aimed exactly at a vertex, must all hit.  Here: the mesh
# as a scene, the rays through the BVH (which adds the bounds' conservative slabs to what is tested).
def watertight_scene(tmp_path, n_theta=16, n_phi=16, seed=12111):
    rng = np.random.default_rng(seed)
    verts = []
    for t in range(n_theta):
        theta = np.pi * t / (n_theta - 1)
        for p in range(n_phi):
            phi = 2 * np.pi * p / (n_phi - 1)
            if t == 0:
                verts.append((0.0, 0.0, 1.0))
            elif t == n_theta - 1:
                verts.append((0.0, 0.0, -1.0))
            elif p == n_phi - 1:
                verts.append(verts[-(n_phi - 1)])
            else:
                r = 1 + 5 * rng.random()
                verts.append((r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(theta)))
    verts = np.array(verts, np.float32)
    idx = []
    off = lambda t, p: t * n_phi + p
    for p in range(n_phi - 1):
        idx += [off(0, 0), off(1, p), off(1, p + 1)]
    for t in range(1, n_theta - 2):
        for p in range(n_phi - 1):
            idx += [off(t, p), off(t + 1, p), off(t + 1, p + 1), off(t, p), off(t + 1, p + 1), off(t, p + 1)]
    for p in range(n_phi - 1):
        idx += [off(n_theta - 1, 0), off(n_theta - 2, p), off(n_theta - 2, p + 1)]
    scene = str(tmp_path / "watertight.pbrt")
    open(scene, "w").write('Film "rgb" "integer xresolution" [ 4 ] "integer yresolution" [ 4 ] "string filename" [ "x.pfm" ]\nWorldBegin\n' + point_light([0, 0, 0]) +
                           'Shape "trianglemesh" "integer indices" [ %s ] "point3 P" [ %s ]\n' % (" ".join(map(str, idx)), " ".join("%.9g" % v for v in verts.reshape(-1))))
    n = 100000
    u = rng.random((n, 4))
    def sphere(u0, u1):
        z = 1 - 2 * u0
        r = np.sqrt(np.maximum(0, 1 - z * z))
        return np.stack([r * np.cos(2 * np.pi * u1), r * np.sin(2 * np.pi * u1), z], axis=1)
    o = (0.5 * sphere(u[:, 0], u[:, 1])).astype(np.float32)
    d = sphere(u[:, 2], u[:, 3]).astype(np.float32)
    # the harder half: straight at a vertex (in render space = world space translated by the camera position: none here, the default
    # camera sits at the origin)
    pick = verts[rng.integers(0, len(verts), n // 2)]
    d[n // 2:] = pick - o[n // 2:]
    return scene, o, d


def test_triangle_watertight_cpu_port(built, tmp_path):
    scene, o, d = watertight_scene(tmp_path)
    rays, hits = str(tmp_path / "rays.bin"), str(tmp_path / "hits.bin")
    np.concatenate([o, d, np.full((len(o), 1), np.inf, np.float32)], axis=1).astype(np.float32).tofile(rays)
    subprocess.run([WF_CPU, "--quiet", "--trace", rays, hits, scene], check=True, capture_output=True)
    h = np.fromfile(hits, np.float32).reshape(len(o), -1)
    assert (h[:, 0] >= 0).all(), "rays leaked through the closed mesh: %d" % (h[:, 0] < 0).sum()


@pytest.mark.gpu
def test_triangle_watertight_gpu(wfpt, tmp_path):
    scene, o, d = watertight_scene(tmp_path)
    s = wfpt.Scene(path=scene, spp=1)
    s.create_renderer(0)
    tmax = np.full(len(o), np.inf, np.float32)
    ref = s.trace_closest(o, d, tmax, reference_order=True)
    fast = s.trace_closest(o, d, tmax, reference_order=False)      # the production traversal (quantised four-wide nodes)
    s.close()
    assert (ref["prim"] >= 0).all(), "rays leaked through the closed mesh: %d" % (ref["prim"] < 0).sum()
    assert (fast["prim"] == ref["prim"]).all() and (fast["t"].view(np.uint32) == ref["t"].view(np.uint32)).all()


# ---------------------------------------------------------------------------------------------------------------------
# Sampler.ConsistentValues (samplers_test.cpp:17-75): going back to a pixel and sample index gives the values it gave before — what
# GenerateRaySamples relies on when it restarts the sampler at dimension 6 + 7 depth.  The restated samplers are functions of
# (pixel, sample index, dimension), so "going back" is a second, independently ordered probe; what can break it is state carried
# between draws, which the comparison with a probe started in the MIDDLE of the sequence exposes for the samplers whose Get2D / Get1D
# depend on the dimension alone.  ZSobolSampler.ValidIndices (:168-196): no sample index is shared by two pixels, and a pixel's spp indices
# fill one aligned block of spp.
CONSISTENT_SAMPLERS = [
    'Sampler "halton" "integer pixelsamples" [ 16 ]', 'Sampler "independent" "integer pixelsamples" [ 16 ]',
    'Sampler "paddedsobol" "integer pixelsamples" [ 16 ] "string randomization" "none"', 'Sampler "paddedsobol" "integer pixelsamples" [ 16 ] "string randomization" "permutedigits"',
    'Sampler "paddedsobol" "integer pixelsamples" [ 16 ] "string randomization" "fastowen"', 'Sampler "paddedsobol" "integer pixelsamples" [ 16 ] "string randomization" "owen"',
    'Sampler "zsobol" "integer pixelsamples" [ 16 ] "string randomization" "none"', 'Sampler "zsobol" "integer pixelsamples" [ 16 ] "string randomization" "permutedigits"',
    'Sampler "zsobol" "integer pixelsamples" [ 16 ] "string randomization" "fastowen"', 'Sampler "zsobol" "integer pixelsamples" [ 16 ] "string randomization" "owen"',
    'Sampler "stratified" "integer xsamples" [ 4 ] "integer ysamples" [ 4 ] "bool jitter" true',
    'Sampler "sobol" "integer pixelsamples" [ 16 ] "string randomization" "none"', 'Sampler "sobol" "integer pixelsamples" [ 16 ] "string randomization" "permutedigits"',
    'Sampler "sobol" "integer pixelsamples" [ 16 ] "string randomization" "owen"', 'Sampler "sobol" "integer pixelsamples" [ 16 ] "string randomization" "fastowen"',
]


def sampler_scene(tmp_path, sampler_line, res=(100, 101)):
    scene = str(tmp_path / "sampler.pbrt")
    open(scene, "w").write('Film "rgb" "integer xresolution" [ %d ] "integer yresolution" [ %d ] "string filename" [ "x.pfm" ]\n%s\nWorldBegin\n' % (res[0], res[1], sampler_line) +
                           point_light([0, 0, 0]) + 'Shape "sphere"\n')
    return scene


def cpu_sampler_probe(tmp_path, scene, pxs, start_dim, mode):
    fin, fout = str(tmp_path / "s_in.bin"), str(tmp_path / "s_out.bin")
    np.asarray(pxs, np.int32).tofile(fin)
    subprocess.run([WF_CPU, "--quiet", "--sampler-probe", fin, fout, str(start_dim), str(mode), scene], check=True, capture_output=True)
    return np.fromfile(fout, np.float32).reshape(len(pxs), -1)


def check_consistent(probe):
    """probe(list of (px, py, sample index), start dimension, mode) -> records"""
    fwd = [(1, 5, s) for s in range(16)]
    a = probe(fwd, 0, -3)
    assert a.shape == (16, 30) and (a >= 0).all() and (a < 1).all()
    probe([(0, 6, 10)], 0, -3)                                  # "go somewhere else"
    b = probe(fwd[::-1], 0, -3)[::-1]                           # back again, in the other order
    assert (a.view(np.uint32) == b.view(np.uint32)).all()
    assert len(np.unique(a)) >= 16                              # (an unscrambled padded Sobol sampler at 16 spp has exactly 16 values)
    return a


@pytest.mark.parametrize("sampler_line", CONSISTENT_SAMPLERS)
def test_sampler_consistent_values_cpu_port(built, tmp_path, sampler_line):
    scene = sampler_scene(tmp_path, sampler_line)
    check_consistent(lambda pxs, sd, mode: cpu_sampler_probe(tmp_path, scene, pxs, sd, mode))


@pytest.mark.gpu
def test_sampler_consistent_values_gpu(wfpt, tmp_path):
    for sampler_line in CONSISTENT_SAMPLERS:
        scene = sampler_scene(tmp_path, sampler_line)
        s = wfpt.Scene(path=scene, spp=16)
        s.create_renderer(0)
        def probe(pxs, sd, mode):
            p = np.asarray(pxs, np.int32)
            return s.sampler_probe(p[:, 0], p[:, 1], p[:, 2], sd, mode)
        got = check_consistent(probe)
        s.close()
        want = check_consistent(lambda pxs, sd, mode: cpu_sampler_probe(tmp_path, scene, pxs, sd, mode))
        assert (got.view(np.uint32) == want.view(np.uint32)).all(), sampler_line


def check_zsobol_valid_indices(probe_for):
    for log_samples in range(0, 11):
        spp = 1 << log_samples
        probe = probe_for(spp)
        pxs = [(x, y, i) for y in range(9) for x in range(16) for i in range(spp)]
        for dim in (0, 3, 6):
            r = probe(pxs, dim, -4).view(np.uint32)
            idx = r[:, 0].astype(np.uint64) | (r[:, 1].astype(np.uint64) << np.uint64(32))
            assert len(np.unique(idx)) == len(idx), (spp, dim)          # no index repeated across pixels
            blocks = (idx // np.uint64(spp)).reshape(16 * 9, spp)
            assert (blocks == blocks[:, :1]).all(), (spp, dim)           # a pixel's samples share one aligned block of spp indices


def test_zsobol_valid_indices_cpu_port(built, tmp_path):
    def probe_for(spp):
        scene = sampler_scene(tmp_path, 'Sampler "zsobol" "integer pixelsamples" [ %d ] "string randomization" "permutedigits"' % spp, (16, 9))
        return lambda pxs, sd, mode: cpu_sampler_probe(tmp_path, scene, pxs, sd, mode)
    check_zsobol_valid_indices(probe_for)


@pytest.mark.gpu
def test_zsobol_valid_indices_gpu(wfpt, tmp_path):
    scenes = []
    def probe_for(spp):
        scene = sampler_scene(tmp_path, 'Sampler "zsobol" "integer pixelsamples" [ %d ] "string randomization" "permutedigits"' % spp, (16, 9))
        s = wfpt.Scene(path=scene, spp=spp)
        s.create_renderer(0)
        scenes.append(s)
        def probe(pxs, sd, mode):
            p = np.asarray(pxs, np.int32)
            return s.sampler_probe(p[:, 0], p[:, 1], p[:, 2], sd, mode)
        return probe
    check_zsobol_valid_indices(probe_for)
    for s in scenes:
        s.close()
