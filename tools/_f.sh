python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c_driver_extension or driver" > gpurun_out/t.log 2>&1; tail -2 gpurun_out/t.log
SGM_DPL24=1 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "batch or padded or fast_path or tall or tiles" > gpurun_out/t24.log 2>&1; tail -2 gpurun_out/t24.log
for e in 0 1; do for f in 2 3; do
if [ $e = 1 ]; then export SGM_DPL24=1; else unset SGM_DPL24; fi
python bench.py --workload drivingstereo_1762x800_d192_p8 --batch 2 --in-flight $f --steps 20 --warmup 4 --no-cpu-baseline --no-host-boundary 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d=json.loads(l); print('dpl24=$e in-flight $f', d['fps'], d['frames_verified'], d['frames_mismatched'], d['stage_ms_per_batch_launch']['aggregate'], d['stage_ms_per_batch_launch']['sum'])"
done; done
