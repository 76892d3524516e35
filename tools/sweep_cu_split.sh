#!/usr/bin/env bash
# sweep_cu_split.sh OUT.jsonl -- bench.py (KITTI, batches of 8) over stage-stream / CU-partition settings (sgm_set_stage_cus);
# one JSON line per setting with the setting prepended.  On the GPU box.
set -uo pipefail
out=${1:-gpurun_out/cu_split.jsonl}
mkdir -p "$(dirname "$out")"
: > "$out"
run() {   # label, in-flight, env..., then spec
    local label=$1 inflight=$2 spec=$3; shift 3
    local line
    line=$(env "$@" python bench.py --steps 100 --warmup 20 --host-instances "$inflight" --legs sustained ${spec:+--cu-split "$spec"} 2>>"$out.err" | tail -1)
    python - "$label" "$inflight" "$spec" "$*" "$line" >> "$out" <<'PY'
import json, sys
label, inflight, spec, env, line = sys.argv[1:6]
try:
    d = json.loads(line)
    st = d.get("stage_ms_per_batch_launch", {})
    print(json.dumps({"label": label, "in_flight": int(inflight), "cu_split": spec, "env": env, "fps": d["sustained"]["fps"], "fps_steps": d["fps"], "ok": d["sustained"]["frames_verified"], "bad": d["sustained"]["frames_mismatched"] + d["frames_mismatched"],
                      "agg": st.get("aggregate"), "sum": st.get("sum"), "median": st.get("median"), "speckle": st.get("speckle"), "lr": st.get("lrcheck"), "lat": d.get("single_frame_latency_ms")}))
except Exception as e:
    print(json.dumps({"label": label, "error": repr(e), "raw": line[-300:]}))
PY
    tail -1 "$out"
}
run base 4 "" A=1
run base_3 3 "" A=1
run base_2 2 "" A=1
run base_q8 4 "" GPU_MAX_HW_QUEUES=8
for p in 1 2 4; do
  run "post${p}_shared" 4 "post=0:$p" A=1
  run "post${p}_excl" 4 "post=0:$p,main=$p:$((32-p))" A=1
done
run sum_all 4 "post=0:0,sum=0:0" A=1
for q in 8 12 16; do
  run "post2_sum${q}_excl" 4 "post=0:2,sum=2:$q,main=$((2+q)):$((30-q))" A=1
  run "post2_sum${q}_shared" 4 "post=0:2,sum=2:$q" A=1
done
run post2_excl_3 3 "post=0:2,main=2:30" A=1
run post2_excl_2 2 "post=0:2,main=2:30" A=1
run post2_sum12_excl_3 3 "post=0:2,sum=2:12,main=14:18" A=1
