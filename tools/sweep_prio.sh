#!/usr/bin/env bash
# sweep_prio.sh OUT.jsonl -- bench.py (KITTI, batches of 8) over stage-stream dispatch priorities (sgm_set_stage_priority)
set -uo pipefail
out=${1:-gpurun_out/prio.jsonl}
mkdir -p "$(dirname "$out")"
: > "$out"
run() {
    local label=$1 inflight=$2 spec=$3
    local line
    line=$(python bench.py --steps 100 --warmup 20 --host-instances "$inflight" --legs sustained ${spec:+--cu-split "$spec"} 2>>"$out.err" | tail -1)
    python - "$label" "$inflight" "$spec" "$line" >> "$out" <<'PY'
import json, sys
label, inflight, spec, line = sys.argv[1:5]
try:
    d = json.loads(line)
    st = d.get("stage_ms_per_batch_launch", {})
    print(json.dumps({"label": label, "in_flight": int(inflight), "cu_split": spec, "fps": d["sustained"]["fps"], "fps_steps": d["fps"], "ok": d["sustained"]["frames_verified"], "bad": d["sustained"]["frames_mismatched"] + d["frames_mismatched"],
                      "agg": st.get("aggregate"), "sum": st.get("sum"), "median": st.get("median"), "speckle": st.get("speckle")}))
except Exception as e:
    print(json.dumps({"label": label, "error": repr(e), "raw": line[-300:]}))
PY
    tail -1 "$out"
}
run base 2 ""
run sumhi_posthi 2 "sum=p-1,post=p-1"
run posthi 2 "post=p-1"
run sumhi 2 "sum=p-1"
run mainlo_sum0_posthi 2 "main=p1,sum=p0,post=p-1"
run mainhi 2 "main=p-1,sum=p0,post=p0"
run mainhi_sumlo 2 "main=p-1,sum=p1,post=p0"
run sumhi_posthi_3 3 "sum=p-1,post=p-1"
run mainlo_sum0_posthi_3 3 "main=p1,sum=p0,post=p-1"
run mainlo_sum0_posthi_4 4 "main=p1,sum=p0,post=p-1"
