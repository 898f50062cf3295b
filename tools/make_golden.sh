#!/bin/bash
# tools/make_golden.sh — regenerates tests/golden/ from the REAL reference (needs /root/reference and the
# shimmed build oracle/_ref/{ref_probe,pbrt_ref}: `make -C oracle ref`).  Run in the build container;
# the outputs are committed so that the GPU box (which has no /root/reference) can check against them.
set -e
cd "$(dirname "$0")/.."
G=tests/golden
mkdir -p $G
oracle/_ref/ref_probe $G
sed 's/"integer xresolution" \[ 400 \] "integer yresolution" \[ 400 \]/"integer xresolution" [ 64 ] "integer yresolution" [ 64 ] "bool savefp16" [ false ]/' scenes/cornell-box.pbrt > $G/cornell64.pbrt
python3 - <<'PY'
import sys
sys.path.insert(0, "tools")
import make_scenes
make_scenes.killeroo_like("tests/golden/blobs_small.pbrt", (96, 54), 4, rings=14, segs=20)
make_scenes.materials_lights("tests/golden/materials_lights.pbrt", (96, 54), 4)
make_scenes.media_box("tests/golden/media_box.pbrt", (64, 64), 4)
make_scenes.envmap_scene("tests/golden/envmap.pbrt", (64, 64), 4)
make_scenes.textures_bump("tests/golden/textures_bump.pbrt", (64, 64), 4)
make_scenes.image_textures("tests/golden/image_textures.pbrt", (64, 64), 4)
make_scenes.alpha_normalmap("tests/golden/alpha_normalmap.pbrt", (64, 64), 4)
make_scenes.mix_materials("tests/golden/mix_materials.pbrt", (64, 64), 64)
make_scenes.spheres("tests/golden/spheres.pbrt", (64, 64), 4)
make_scenes.quadrics("tests/golden/quadrics.pbrt", (64, 64), 4)
PY
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/cornell64_ref.pfm $G/cornell64.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/blobs_small_ref.pfm $G/blobs_small.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/materials_lights_ref.pfm $G/materials_lights.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/media_box_ref.pfm $G/media_box.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/envmap_ref.pfm $G/envmap.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/textures_bump_ref.pfm $G/textures_bump.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/image_textures_ref.pfm $G/image_textures.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/alpha_normalmap_ref.pfm $G/alpha_normalmap.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/spheres_ref.pfm $G/spheres.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/quadrics_ref.pfm $G/quadrics.pbrt
# MixMaterial: the reference's choice hashes heap pointers -> statistical comparison only (64 spp, block means)
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --outfile $G/mix_materials_ref.pfm $G/mix_materials.pbrt
# the textures scene through the spherical camera (equirectangular mapping)
sed 's/^Camera "perspective".*/Camera "spherical" "string mapping" "equirectangular"/; s/textures_bump.pfm/spherical_camera.pfm/' $G/textures_bump.pbrt > $G/spherical_camera.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/spherical_camera_ref.pfm $G/spherical_camera.pbrt
# RGBGridMedium: hand-written scene tests/golden/rgbgrid_medium.pbrt (6x5x4 cells, absorbing + scattering + emitting)
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/rgbgrid_medium_ref.pfm $G/rgbgrid_medium.pbrt
# GridMedium with a temperature grid (blackbody emission): hand-written scene tests/golden/tempgrid_medium.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/tempgrid_medium_ref.pfm $G/tempgrid_medium.pbrt
# spherical / cylindrical / planar texture mappings (checkerboard, imagemap, alpha): hand-written tests/golden/texture_mappings.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/texture_mappings_ref.pfm $G/texture_mappings.pbrt
# bilerp and directionmix textures (float and spectrum): hand-written tests/golden/textures_extra.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/textures_extra_ref.pfm $G/textures_extra.pbrt
# texture graphs nested ten levels deep (float and spectrum chains; as reflectance, roughness, displacement and alpha): hand-written
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/textures_deep_ref.pfm $G/textures_deep.pbrt
# image-textured diffuse area lights (triangles and a sphere): hand-written tests/golden/arealight_image.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/arealight_image_ref.pfm $G/arealight_image.pbrt
# alpha-masked emitters (checkerboard cut-out, fractional alpha, the invisible alpha-0 DeltaPosition case): hand-written
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/arealight_alpha_ref.pfm $G/arealight_alpha.pbrt
# PNG image maps (every colour type / depth, sRGB / linear / gamma encodings, non-power-of-two resize, RGBA alpha, PNG normal
# map and emitter image): fixtures from tools/make_png_fixtures.py, decoded in the reference build by oracle/ref_build/shim_png.cpp
python3 tools/make_png_fixtures.py
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/png_textures_ref.pfm $G/png_textures.pbrt
# EWA-filtered image maps (anisotropic footprints, the maxanisotropy clamp, float and RGB lookups): hand-written
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/textures_ewa_ref.pfm $G/textures_ewa.pbrt
# Curve shapes (flat / cylinder / ribbon, Bezier and b-spline of degree 2 and 3, hair strands, curves in object instances): hand-written
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/curves_ref.pfm $G/curves.pbrt
# RealisticCamera (tests/golden/dgauss50.dat: lens focusing, exit-pupil bounds, vignetting weights; the second with the built-in star aperture image)
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/realistic_camera_ref.pfm $G/realistic_camera.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/realistic_camera_star_ref.pfm $G/realistic_camera_star.pbrt
# PortalImageInfiniteLight (an environment map through a window; power light sampler): hand-written
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/portal_light_ref.pfm $G/portal_light.pbrt
# the same room lit through the portal by a uniform "L" with "illuminance", plus a gzipped binary PLY (ball.ply.gz)
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/portal_uniform_ref.pfm $G/portal_uniform.pbrt
# Shape "loopsubdiv" (closed, open and valence-3 control meshes; limit positions and normals): hand-written
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/loopsubdiv_ref.pfm $G/loopsubdiv.pbrt
# film "whitebalance" + "iso" (the cornell64 scene with a 4200 K sensor illuminant): film_whitebalance.pbrt is a sed of cornell64.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/film_whitebalance_ref.pfm $G/film_whitebalance.pbrt
# measured camera sensors (PixelSensor's least-squares XYZ matrix over the ColorChecker swatches; default and explicit white balance): seds of film_whitebalance.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/film_sensor_ref.pfm $G/film_sensor.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/film_sensor_wb_ref.pfm $G/film_sensor_wb.pbrt
# plymesh "displacement" (TriQuadMesh::Displace: triangle + quad faces, with / without normals, in an instance, as an emitter): hand-written, displace_*.ply
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/displacement_ref.pfm $G/displacement.pbrt
# a PLY file with triangle AND quad faces as an emitter and alpha-tested (the patches get their own mesh entry): hand-written
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/plymesh_mixed_ref.pfm $G/plymesh_mixed.pbrt
# camera motion blur (ActiveTransform StartTime / EndTime around Camera, TransformTimes, shutter inside / outside the interval):
# camera_motion.pbrt = image_textures.pbrt under a moving, rotating, scaling perspective camera with a lens; camera_motion_spherical.pbrt =
# spherical_camera.pbrt under a translating camera
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/camera_motion_ref.pfm $G/camera_motion.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/camera_motion_spherical_ref.pfm $G/camera_motion_spherical.pbrt
# Option "rendercoordsys" camera / world (+ "displacementedgescale"): camera_motion.pbrt and displacement.pbrt with the Option lines in front
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/rendercoordsys_camera_ref.pfm $G/rendercoordsys_camera.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/rendercoordsys_world_ref.pfm $G/rendercoordsys_world.pbrt
# parser / parameter torture scene (hand-written: see its header)
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/parser_torture_ref.pfm $G/parser_torture.pbrt
# a scene without geometry (the product keeps one unhittable placeholder triangle; the reference an empty aggregate)
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/empty_scene_ref.pfm $G/empty_scene.pbrt
# alpha textures on spheres / disks / cylinders / bilinear patches (re-intersection behind a rejected hit), an alpha-masked emissive sphere
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/quadrics_alpha_ref.pfm $G/quadrics_alpha.pbrt
# a goniometric light from an 8-bit R G B PNG (channel average re-quantised into an 8-bit "Y" image)
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/goniometric_png_ref.pfm $G/goniometric_png.pbrt
# object instancing (two definitions, five instances incl. a mirroring one): hand-written tests/golden/instances.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/instances_ref.pfm $G/instances.pbrt
# the reference's other BVH builder: blobs_small with `splitmethod "hlbvh"` (cpu/aggregates.cpp:389-503, 626-722)
sed 's/^WorldBegin/Accelerator "bvh" "string splitmethod" "hlbvh"\nWorldBegin/; s/killeroo-like.pfm/blobs_hlbvh.pfm/' $G/blobs_small.pbrt > $G/blobs_hlbvh.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/blobs_hlbvh_ref.pfm $G/blobs_hlbvh.pbrt
# procedural textures (fbm, wrinkled, windy, marble, dots, 3D checkerboard): hand-written tests/golden/textures_noise.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/textures_noise_ref.pfm $G/textures_noise.pbrt
# CloudMedium: hand-written tests/golden/cloud_medium.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/cloud_medium_ref.pfm $G/cloud_medium.pbrt
# HairMaterial / HairBxDF: hand-written tests/golden/hair.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/hair_ref.pfm $G/hair.pbrt
# BilinearPatch shapes (bilinearmesh + a plymesh with quad faces): hand-written tests/golden/bilinear.pbrt + bilinear_quads.ply
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/bilinear_ref.pfm $G/bilinear.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/bilinear_lights_ref.pfm $G/bilinear_lights.pbrt
# quadrics and bilinear patches inside object-instance definitions: hand-written tests/golden/instances_quadrics.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/instances_quadrics_ref.pfm $G/instances_quadrics.pbrt
# named scattering coefficients: the fog of media_box by preset, a blob of the subsurface scene by name
sed 's/^MakeNamedMedium "fog".*/MakeNamedMedium "fog" "string type" [ "homogeneous" ] "string preset" "Skimmilk" "float scale" [ 0.25 ] "float g" [ 0.3 ]/; s/media_box.pfm/media_preset.pfm/' $G/media_box.pbrt > $G/media_preset.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/media_preset_ref.pfm $G/media_preset.pbrt
# K12 subsurface scattering: the two blobs of blobs_small as SubsurfaceMaterials (reflectance + mfp; default coefficients with scale and g)
sed 's/^MakeNamedMaterial "blobA".*/MakeNamedMaterial "blobA" "string type" [ "subsurface" ] "rgb reflectance" [ 0.8 0.5 0.35 ] "rgb mfp" [ 0.25 0.12 0.06 ] "float eta" [ 1.4 ] "float roughness" [ 0.15 ]/; s/^MakeNamedMaterial "blobB".*/MakeNamedMaterial "blobB" "string type" [ "subsurface" ] "float scale" [ 4 ] "float g" [ 0.3 ]/; s/killeroo-like.pfm/subsurface.pfm/' $G/blobs_small.pbrt > $G/subsurface.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/subsurface_ref.pfm $G/subsurface.pbrt
sed 's/^MakeNamedMaterial "blobB".*/MakeNamedMaterial "blobB" "string type" [ "subsurface" ] "string name" "Skin1" "float scale" [ 6 ] "float g" [ 0.5 ]/; s/subsurface.pfm/subsurface_named.pfm/' $G/subsurface.pbrt > $G/subsurface_named.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/subsurface_named_ref.pfm $G/subsurface_named.pbrt
# goniometric + projection lights: hand-written scene tests/golden/lights_extra.pbrt (uses sky.pfm and wood.pfm)
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/lights_extra_ref.pfm $G/lights_extra.pbrt
# the same lights through the PowerLightSampler (alias table)
sed 's/Integrator "volpath"/Integrator "volpath" "string lightsampler" [ "power" ]/' $G/materials_lights.pbrt > $G/materials_lights_power.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/materials_lights_power_ref.pfm $G/materials_lights_power.pbrt
# the other pixel samplers on the Cornell box (scene defaults: independent 4 spp, stratified 4x4, paddedsobol 16, halton 16, sobol 16)
for smp in independent stratified paddedsobol halton sobol; do
  sed "s/^Sampler \"zsobol\".*/Sampler \"$smp\"/" $G/cornell64.pbrt > $G/cornell64_$smp.pbrt
  oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --outfile $G/cornell64_${smp}_ref.pfm $G/cornell64_$smp.pbrt
done
sed 's/^Sampler "zsobol".*/Sampler "sobol" "string randomization" "owen" "integer pixelsamples" 8/' $G/cornell64.pbrt > $G/cornell64_sobol_owen.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --outfile $G/cornell64_sobol_owen_ref.pfm $G/cornell64_sobol_owen.pbrt
# the named physical oracle (not sample-aligned): VolPathIntegrator at high spp, for mean comparisons
oracle/_ref/pbrt_ref --quiet --seed 0 --spp 256 --outfile $G/cornell64_volpath256.pfm $G/cornell64.pbrt
ls -la $G
python3 tools/make_libm_golden.py
# downscaled versions of the BENCHMARKED stand-ins (BASELINE configs[2], [4], [3]): the scene trees are regenerated by the tests
# (tools/make_scenes.py bench_small, deterministic) — only the reference's render and the tree hashes are committed
python3 - <<'PY'
import json, subprocess, sys, tempfile, os, shutil
sys.path.insert(0, "tools")
import make_scenes
hashes = {}
for name, cfg in make_scenes.BENCH_SMALL.items():
    td = tempfile.mkdtemp()
    path = make_scenes.bench_small(name, td)
    hashes[name] = make_scenes.tree_hash(td)
    subprocess.run(["oracle/_ref/pbrt_ref", "--wavefront", "--quiet", "--seed", "0", "--spp", str(cfg["spp"]), "--outfile",
                    os.path.abspath("tests/golden/%s_ref.pfm" % name), path], check=True)
    shutil.rmtree(td)
json.dump(hashes, open("tests/golden/bench_small_hashes.json", "w"), indent=1)
PY
# participating media + object instances: hand-edited tests/golden/media_instances.pbrt (media_box with its conductor box instanced)
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/media_instances_ref.pfm $G/media_instances.pbrt
# SpectralFilm (film.h:401-530): cornell64 with `Film "spectral"` (8 buckets over 380-780 nm, maxcomponentvalue 6); the oracle build's OpenEXR
# stand-in writes uncompressed scan-line files (oracle/ref_build/shims/ImfShim.h)
sed 's/^Film "rgb".*/Film "spectral" "integer nbuckets" [ 8 ] "float lambdamin" [ 380 ] "float lambdamax" [ 780 ] "float maxcomponentvalue" [ 6 ] "string filename" [ "spectral_film.exr" ] "integer xresolution" [ 64 ] "integer yresolution" [ 64 ] "bool savefp16" [ false ]/' $G/cornell64.pbrt > $G/spectral_film.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/spectral_film_ref.exr $G/spectral_film.pbrt
# GBufferFilm (film.h:319-400): the materials_lights scene with `Film "gbuffer"` — 25 channels (RGB, albedo from BxDF::rho with the
# reference's 16 fixed samples, camera-space position / normals / dz, uv, Welford variance)
sed 's/^Film "rgb".*/Film "gbuffer" "string filename" [ "gbuffer_film.exr" ] "integer xresolution" [ 64 ] "integer yresolution" [ 64 ] "bool savefp16" [ false ]/' $G/materials_lights.pbrt > $G/gbuffer_film.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/gbuffer_film_ref.exr $G/gbuffer_film.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/curves_alpha_ref.pfm $G/curves_alpha.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/animated_ref.pfm $G/animated.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/animated_sss_ref.pfm $G/animated_sss.pbrt
oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --spp 4 --outfile $G/face_indices_ref.pfm $G/face_indices.pbrt
