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
