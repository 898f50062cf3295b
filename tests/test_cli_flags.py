"""The command-line options of the reference's render path (cmd/pbrt.cpp:129-204) and its `Option` directive, on the CPU: the checker
oracle/_build/wf_cpu (the product's host code + the shared kernels' bodies) against the reference build oracle/_ref/pbrt_ref run here with
the same flags — images bit-identical.  (Skipped where the reference build is absent: it needs /root/reference at build time.)"""
import os
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, WF_CPU, read_pfm

PBRT_REF = os.path.join(ROOT, "oracle", "_ref", "pbrt_ref")

CASES = [
    ("image_textures", ["--quick"]),                                   # a quarter of the resolution, one sample
    ("cornell64_independent", ["--quick"]),                            # ... except the independent sampler, which keeps its count
    ("cornell64_sobol", ["--quick"]),                                  # (the Sobol' scale follows the quartered film)
    ("cornell64_stratified", ["--quick"]),
    ("realistic_camera", ["--quick"]),
    ("image_textures", ["--spp", "4", "--disable-image-textures"]),   # every MIP pyramid reduced to its coarsest level
    ("png_textures", ["--spp", "4", "--disable-image-textures"]),
    ("cornell64", ["--spp", "4", "--pixel", "20,30"]),
    ("cornell64", ["--spp", "8", "--debugstart", "3,2"]),
    ("materials_lights", ["--spp", "8", "--debugstart", "5"]),
    ("displacement", ["--spp", "4", "--displacement-edge-scale", "2.5"]),
    ("camera_motion", ["--spp", "4", "--render-coord-sys", "camera"]),
    ("instances", ["--spp", "4", "--render-coord-sys", "world"]),
    ("cornell64", ["--spp", "4", "--cropwindow", "0.25,0.75,0.1,0.6"]),
    ("cornell64", ["--spp", "4", "--pixelbounds", "10,40,20,50"]),
    ("textures_bump", ["--spp", "4", "--disable-texture-filtering"]),
    ("cornell64", ["--spp", "4", "--disable-pixel-jitter"]),
    ("cornell64", ["--spp", "4", "--disable-wavelength-jitter", "--seed", "7"]),
]


@pytest.mark.parametrize("scene,flags", CASES, ids=[c[0] + " " + " ".join(c[1]) for c in CASES])
def test_cli_flag_matches_the_reference(tmp_path, scene, flags):
    if not os.path.exists(PBRT_REF):
        pytest.skip("oracle/_ref/pbrt_ref not built")
    path = os.path.join(GOLDEN, scene + ".pbrt")
    ref, ours = str(tmp_path / "ref.pfm"), str(tmp_path / "ours.pfm")
    subprocess.run([PBRT_REF, "--wavefront", "--quiet", "--seed", "0"] + flags + ["--outfile", ref, path], check=True, capture_output=True, cwd=str(tmp_path))
    subprocess.run([WF_CPU, "--quiet"] + flags + ["--outfile", ours, path], check=True, capture_output=True, cwd=str(tmp_path))
    a, b = read_pfm(ours), read_pfm(ref)
    assert a.shape == b.shape
    assert (a.view(np.uint32) == b.view(np.uint32)).all(), "fraction identical %f" % (a == b).mean()


def test_unknown_option_directive_is_an_error(tmp_path):
    """BasicSceneBuilder::Option rejects names it does not know (scene.cpp:489-575): so does the product's parser (typed names like
    "string rendercoordsys" included — the reference's syntax is `Option "name" value`)."""
    text = open(os.path.join(GOLDEN, "cornell64.pbrt")).read()
    for line in ('Option "string rendercoordsys" "world"', 'Option "nosuchoption" 1', 'Option "rendercoordsys" "sideways"'):
        p = str(tmp_path / "bad.pbrt")
        open(p, "w").write(line + "\n" + text)
        r = subprocess.run([WF_CPU, "--quiet", "--spp", "1", "--outfile", str(tmp_path / "o.pfm"), p], capture_output=True, text=True)
        assert r.returncode != 0, line


@pytest.mark.parametrize("film_cs,world_cs", [("rec2020", None), ("dci-p3", None), ("aces2065-1", None), ("aces2065-1", "srgb"), (None, "aces2065-1"), (None, "rec2020")])
def test_colour_spaces_match_the_reference(tmp_path, film_cs, world_cs):
    """ColorSpace before Film (the film's colour space: the sensor matrix, and Image::Write's conversion of a non-sRGB image to sRGB for .pfm
    output) and inside the world block (RGB parameters through that gamut's RGB -> spectrum table; ACES2065-1's is optimised against the
    table generator's own D60 table), against pbrt_ref run here."""
    if not os.path.exists(PBRT_REF):
        pytest.skip("oracle/_ref/pbrt_ref not built")
    text = open(os.path.join(GOLDEN, "film_whitebalance.pbrt")).read()
    if film_cs:
        text = 'ColorSpace "%s"\n' % film_cs + text
    if world_cs:
        text = text.replace("WorldBegin", 'WorldBegin\nColorSpace "%s"' % world_cs, 1)
    path = str(tmp_path / "cs.pbrt")
    open(path, "w").write(text)
    ref, ours = str(tmp_path / "ref.pfm"), str(tmp_path / "ours.pfm")
    subprocess.run([PBRT_REF, "--wavefront", "--quiet", "--seed", "0", "--spp", "4", "--outfile", ref, path], check=True, capture_output=True, cwd=str(tmp_path))
    subprocess.run([WF_CPU, "--quiet", "--spp", "4", "--outfile", ours, path], check=True, capture_output=True, cwd=str(tmp_path))
    a, b = read_pfm(ours), read_pfm(ref)
    assert (a.view(np.uint32) == b.view(np.uint32)).all(), "fraction identical %f" % (a == b).mean()
