#!/usr/bin/env bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/refresh_profiles.sh'): collects everything profiles/ holds for the current
# kernels into gpurun_out/ (the only directory that travels back); copy the files into profiles/ afterwards.
set -x
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
rm -f $R/gpurun_out/counters.json
for wl in kitti_1242x375_d128_p8 cone_450x375_d64_p8 drivingstereo_1762x800_d192_p8 middlebury_2880x1988_d256_p8; do
  python3 $R/tools/profile_counters.py --workload $wl > $R/gpurun_out/counters_$wl.log 2>&1; tail -2 $R/gpurun_out/counters_$wl.log
done
cp $R/gpurun_out/counters.json $R/profiles/counters.json          # so that the bench runs below read fresh counters
# the headline loop under the kernel trace: per-kernel calls / average duration of the timed configuration.  60 timed steps after 2
# warm-up steps (248 launches per kernel, 8 of them warm-up) so that the CSV's average over ALL calls and the JSON line's average
# over the timed launches describe nearly the same set; tools/trace_summary.py gives the timed-only mean from the trace.
rm -rf $R/gpurun_out/prof_stats && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o stats --output-format csv -- python3 $R/bench.py --gpus 1 --steps 60 --warmup 2 --legs headline > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/prof_stats.log; echo rocprof=$?
python3 $R/tools/trace_summary.py $R/gpurun_out/prof_stats/stats_kernel_trace.csv $R/gpurun_out/bench_under_rocprof.json > $R/gpurun_out/kernel_trace_summary.json
cd $R && sleep 10 && python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo bench=$?
