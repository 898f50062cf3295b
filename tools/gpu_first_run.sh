set -x
mkdir -p gpurun_out
rocminfo | grep -E "gfx|Compute Unit" | head -4
nproc
cd pbrt-v4_amd/_build
sed 's/"integer yresolution" \[ 400 \]/"integer yresolution" [ 400 ] "bool savefp16" [ false ]/' ../../scenes/cornell-box.pbrt > /tmp/c32.pbrt
timeout 300 ./pbrt_amd --stats --outfile /tmp/gpu.pfm /tmp/c32.pbrt > ../../gpurun_out/cornell_gpu_stats.txt 2>&1
echo rc=$?
tail -40 ../../gpurun_out/cornell_gpu_stats.txt
cd ../..
timeout 300 oracle/_build/wf_cpu --outfile /tmp/cpu.pfm /tmp/c32.pbrt
timeout 300 oracle/_ref/pbrt_ref --wavefront --quiet --seed 0 --outfile /tmp/ref.pfm /tmp/c32.pbrt 2>&1 | grep -v Warning | tail -2
python3 tools/compare_pfm.py /tmp/gpu.pfm /tmp/cpu.pfm
python3 tools/compare_pfm.py /tmp/gpu.pfm /tmp/ref.pfm
python3 tools/compare_pfm.py /tmp/cpu.pfm /tmp/ref.pfm
cp /tmp/gpu.pfm gpurun_out/cornell_gpu.pfm
