for b in 4 6 8 12 16 24; do for fl in 1 2 3; do
  python bench.py --no-cpu-baseline --no-host-boundary --in-flight $fl --batch $b --steps $((320 / b)) > gpurun_out/r2_b_${b}_$fl.json 2>> gpurun_out/r2_b.err
  python - $b $fl <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r2_b_{sys.argv[1]}_{sys.argv[2]}.json')); s=d['stage_ms_per_batch_launch']; b=int(sys.argv[1])
print('B',b,'in-flight',sys.argv[2], d['fps'],'fps; per frame: aggregate %.4f sum %.4f median %.4f speckle %.4f' % (s['aggregate']/b, s['sum']/b, s['median']/b, s['speckle']/b), 'verified', d['frames_verified'], d['frames_without_reference_digest'], flush=True)
PY
done; done
