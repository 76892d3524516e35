set -x
python -m pytest tests -m gpu -x -q > gpurun_out/r2_pytest3.log 2>&1; echo exit=$? >> gpurun_out/r2_pytest3.log; tail -4 gpurun_out/r2_pytest3.log
grep -q "exit=0" gpurun_out/r2_pytest3.log || exit 1
python bench.py --no-cpu-baseline --no-host-boundary > gpurun_out/r2_bench3.json 2> gpurun_out/r2_bench3.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r2_bench3.json')); print(d['fps'], d['stage_ms_per_batch_launch'], d['frames_verified'], d['single_frame_latency_ms'])
PY
python bench.py --no-cpu-baseline --no-host-boundary --in-flight 1 > gpurun_out/r2_bench3_f1.json 2>> gpurun_out/r2_bench3.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/r2_bench3_f1.json')); print(d['fps'], d['stage_ms_per_batch_launch'])
PY
cd /tmp && export TMPDIR=/tmp && python3 $GRAFT_REPO_ROOT/tools/profile_counters.py 2>&1 | tail -40
