for hl in 0 32 64; do for fl in 1 2; do
  SGM_HL=$hl python bench.py --no-cpu-baseline --no-host-boundary --in-flight $fl --steps 40 > gpurun_out/r2_hl_${hl}_$fl.json 2>> gpurun_out/r2_hl.err
  python - $hl $fl <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r2_hl_{sys.argv[1]}_{sys.argv[2]}.json')); print('HL',sys.argv[1],'in-flight',sys.argv[2], d['fps'],'fps aggregate', d['stage_ms_per_batch_launch']['aggregate'], 'verified', d['frames_verified'], flush=True)
PY
done; done
