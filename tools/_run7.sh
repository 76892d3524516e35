set -x
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/profile_counters.py > $GRAFT_REPO_ROOT/gpurun_out/r2_counters.log 2>&1; tail -3 $GRAFT_REPO_ROOT/gpurun_out/r2_counters.log
cd $GRAFT_REPO_ROOT && cp gpurun_out/counters.json profiles/counters.json
rm -rf gpurun_out/prof_r02 && cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02 -o r02 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-boundary > $GRAFT_REPO_ROOT/gpurun_out/r02_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r02.log; echo rocprof=$?
cd $GRAFT_REPO_ROOT && find gpurun_out/prof_r02 -name "*stats*" | head; python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; echo bench=$?; python - <<'PY'
import json
d=json.load(open('gpurun_out/r02_bench.json'))
print(json.dumps({k:d[k] for k in ('value','fps','ms_per_step','frames_verified','roofline','roofline_sum_wta','host_boundary','cpu_baseline','frame_traffic')}, indent=1))
PY
