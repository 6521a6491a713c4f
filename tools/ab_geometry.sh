#!/bin/bash
# Engines x runs geometry sweep of the CQL bench inside one GPU call, round-robin (box-to-box and thermal drift cancel):
#   tools/ab_geometry.sh "2x96 2x128 1x128" [rounds=2]
# prints value, ms/step and the dominant launch's roofline fraction per configuration and round.
mkdir -p gpurun_out/geo
for round in $(seq 1 ${2:-2}); do
  for g in $1; do
    e=${g%x*}; r=${g#*x}
    python3 bench.py --steps 20 --warmup 5 --no-sides --no-cpu-baseline --engines-per-gpu $e --runs-per-gpu $r > gpurun_out/geo/$g.$round.json 2> gpurun_out/geo/$g.$round.err || { tail -3 gpurun_out/geo/$g.$round.err; exit 1; }
    python3 - gpurun_out/geo/$g.$round.json $g <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1])
r = d["roofline"]
print("%-7s %8.0f steps/s  %7.3f ms/step  %s %.1f us frac %.3f (occupied CUs: %s)" % (sys.argv[2], d["value"], d["ms_per_step"], r["kernel"], r["avg_launch_ms"] * 1e3, r["frac"], r.get("frac_of_occupied_cus")))
PY
  done
done
