for m in 0xFF 0x03 0xFC 0x0C 0xF0 0x01; do
  SGM_DIR_MASK=$m python bench.py --no-cpu-baseline --no-host-boundary --in-flight 1 --steps 20 > gpurun_out/r2_mask_$m.json 2>> gpurun_out/r2_mask.err
  python - $m <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r2_mask_{sys.argv[1]}.json')); print(sys.argv[1], 'aggregate ms per 8 frames', d['stage_ms_per_batch_launch']['aggregate'], flush=True)
PY
done
