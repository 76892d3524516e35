set -x
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -k "tall or random_shapes or full_size" > gpurun_out/r2_pytest_median.log 2>&1; echo exit=$? >> gpurun_out/r2_pytest_median.log; tail -15 gpurun_out/r2_pytest_median.log
grep -q "exit=0" gpurun_out/r2_pytest_median.log || exit 1
for c in 1 0; do SGM_MEDIAN_CHAIN=$c timeout -k 10 300 python bench.py --workload uhd_3840x2160_d128_p8 --batch 1 --steps 24 --warmup 4 --no-cpu-baseline --no-host-boundary > gpurun_out/r2_uhd_chain$c.json 2>> gpurun_out/r2_uhd.err; python - $c <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r2_uhd_chain{sys.argv[1]}.json')); print('chain',sys.argv[1],d['fps'],'fps', d['stage_ms_per_batch_launch'], 'single', d['stage_ms_single_frame']['median'], d['single_frame_latency_ms'], d['frames_verified'])
PY
done
