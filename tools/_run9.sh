set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest_full.log 2>&1; echo exit=$? >> gpurun_out/r2_pytest_full.log; tail -4 gpurun_out/r2_pytest_full.log
grep -q "exit=0" gpurun_out/r2_pytest_full.log || exit 1
python -c "import __graft_entry__ as g; g.smoke()"
cd /tmp && export TMPDIR=/tmp
python3 $GRAFT_REPO_ROOT/tools/profile_counters.py > $GRAFT_REPO_ROOT/gpurun_out/r2_counters.log 2>&1; tail -3 $GRAFT_REPO_ROOT/gpurun_out/r2_counters.log
cd $GRAFT_REPO_ROOT && cp gpurun_out/counters.json profiles/counters.json
rm -rf gpurun_out/prof_r02 && cd /tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r02 -o r02 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-boundary > $GRAFT_REPO_ROOT/gpurun_out/r02_bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof_r02.log; echo rocprof=$?
cd $GRAFT_REPO_ROOT && python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err; echo bench=$?
for wl in middlebury_2880x1988_d256_p8 drivingstereo_1762x800_d192_p8 uhd_3840x2160_d128_p8 uhd_3840x2160_d256_p8; do for fl in 2 3; do
timeout -k 10 300 python bench.py --workload $wl --batch 2 --in-flight $fl --steps 16 --warmup 4 --no-cpu-baseline --no-host-boundary > gpurun_out/r02_${wl}_f$fl.json 2>> gpurun_out/r02_big.err; python - $wl $fl <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r02_{sys.argv[1]}_f{sys.argv[2]}.json')); print(sys.argv[1], 'in-flight', sys.argv[2], d['fps'],'fps', d['value'], 'Mdisp/s verified', d['frames_verified'], d['frames_mismatched'], d['frames_without_reference_digest'], 'latency', d['single_frame_latency_ms'], flush=True)
PY
done; done
timeout -k 10 300 python bench.py --mode tiles --steps 32 --warmup 6 > gpurun_out/r02_tiles1.json 2> gpurun_out/r02_tiles1.err; cat gpurun_out/r02_tiles1.json | cut -c1-400
