#!/bin/bash
# tools/ab_geometry_p.sh PREC "E R" "E R" ... : engines-per-GPU x runs-per-engine geometries of one precision, round-robin twice inside one gpurun call
P=$1; shift
for rep in 1 2; do for er in "$@"; do
  e=${er% *}; r=${er#* }
  python bench.py --steps 40 --warmup 20 --min-reps 3 --min-seconds 1 --no-sides --no-cpu-baseline --engines-per-gpu $e --runs-per-gpu $r --precision $P --profile-steps 0 > gpurun_out/geo_${P}_${e}x${r}.json 2> gpurun_out/geo_${P}_${e}x${r}.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/geo_${P}_${e}x${r}.json").read().strip().split("\n")[-1])
print("precision $P  %d x %3d : %7.0f steps/s  %.3f ms/step" % ($e, $r, d["value"], d["ms_per_step"]), flush=True)
PY
done; done
