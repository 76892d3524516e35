#!/usr/bin/env bash
# sweep_instances.sh -- the host-pointer pipeline (sgm_stream: C, one pthread per instance, batches of 8 KITTI frames) with 3..6 instances,
# three alternating rounds.  On the GPU box.
cd "$(dirname "${BASH_SOURCE[0]}")/.."
for round in 1 2 3; do
  for n in 4 5 6 3; do
    printf "round %d instances %d: " "$round" "$n"
    timeout -k 10 60 soc_project_stereo_matching_amd/sgm_stream --instances "$n" --batch 8 --seconds 3 2>/dev/null | python3 -c 'import json,sys; d=json.loads(sys.stdin.read()); print(round(d["fps"],1), "fps")'
  done
done
