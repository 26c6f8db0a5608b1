set -e
cp mygpuraytracer_amd/libmi355x_pathtracer.so /tmp/keep.so
AB_REPS=2 bash tools/ab_bench.sh base T64 T128
cp .ab/libT64.so mygpuraytracer_amd/libmi355x_pathtracer.so
python tools/gpu_quick.py > gpurun_out/quick_T64.log 2>&1 || true
tail -15 gpurun_out/quick_T64.log
cp /tmp/keep.so mygpuraytracer_amd/libmi355x_pathtracer.so
