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
    s.close()
    check_scene_average(img)
    path = str(tmp_path / "furnace.pbrt")
    open(path, "w").write(text)
    out = str(tmp_path / "cpu.pfm")
    run_wf_cpu(path, out)
    assert (read_pfm(out).view(np.uint32) == img.view(np.uint32)).all()
