#!/bin/bash
# One profiling pass on the GPU box (run through gpurun): kernel trace + stats of the default bench command, then the two
# PMC passes (FETCH_SIZE, WRITE_SIZE) in runs of their own, as MI355X_MICROARCH.md prescribes.  Output under gpurun_out/.
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh'   then   python tools/collect_profiles.py TAG gpurun_out/prof_stats gpurun_out/prof_fetch gpurun_out/prof_write
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_stats $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write
PMCARGS="--lanes 1 --steps 24 --warmup 12 --no-cpu-baseline --no-extra-legs"
# what the counter passes ran, for the collectors' _how (tools/profile_meta.py: how_of)
for k in fetch write; do
  echo "rocprofv3 --pmc $(echo $k | tr a-z A-Z)_SIZE (a pass of its own, no tracing) -- python3 bench.py $PMCARGS: cornellObj.txt 1920x1080 depth 8 (C4), one launch set at a time; a launch covers 12 iterations (the default batch at 1080p)" > $R/gpurun_out/prof_${k}_how.txt
done
# (--no-extra-legs: the profiled process runs the C4 steps only, not the C5 / per-call legs the plain bench adds)
# kernels back to back (--lanes 1): per-kernel durations comparable with the hipEvent figures of bench.py's roofline leg
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --lanes 1 --no-extra-legs > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_stats.err
echo "stats pass done"
# the default command (three launch sets in flight: kernels overlap, their durations stretch)
rm -rf $R/gpurun_out/prof_stats2
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats2 -- python3 $R/bench.py --no-extra-legs > $R/gpurun_out/prof_bench2.json 2> $R/gpurun_out/prof_stats2.err
echo "stats pass (default command) done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py $PMCARGS > $R/gpurun_out/prof_fetch_bench.json 2> $R/gpurun_out/prof_fetch.err
echo "fetch pass done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/bench.py $PMCARGS > /dev/null 2> $R/gpurun_out/prof_write.err
echo "write pass done"
cd $R && python3 bench.py > gpurun_out/bench_plain.json 2> gpurun_out/bench_plain.err
tail -1 gpurun_out/bench_plain.json | cut -c1-300
