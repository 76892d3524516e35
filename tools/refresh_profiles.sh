#!/usr/bin/env bash
# Run ON THE GPU BOX (gpurun -- 'bash tools/refresh_profiles.sh'): collects everything profiles/ holds for the current
# kernels into gpurun_out/ (the only directory that travels back); copy the files into profiles/ afterwards.
set -x
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "${BASH_SOURCE[0]}")/.." && pwd)}
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/profile_counters.py > $R/gpurun_out/counters.log 2>&1; tail -2 $R/gpurun_out/counters.log
cp $R/gpurun_out/counters.json $R/profiles/counters.json          # so that the bench runs below read fresh counters
rm -rf $R/gpurun_out/prof_stats && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_stats -o stats --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-host-boundary > $R/gpurun_out/bench_under_rocprof.json 2> $R/gpurun_out/prof_stats.log; echo rocprof=$?
cd $R && sleep 20 && python bench.py --alone > gpurun_out/bench.json 2> gpurun_out/bench.err; echo bench=$?
python tools/agg_clock.py > gpurun_out/agg_clock.txt 2>/dev/null
