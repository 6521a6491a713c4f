#!/bin/bash
# engines x runs scan of bench.py (one GPU call; later entries run on a warmer chip): tools/er_scan.sh "E R" "E R" ...
for er in "$@"; do set -- $er; E=$1; R=$2
  python bench.py --steps 40 --warmup 20 --min-reps 5 --min-seconds 2 --no-sides --no-cpu-baseline --engines-per-gpu $E --runs-per-gpu $R --profile-steps 0 > gpurun_out/er.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/er.json").read().strip().split("\n")[-1])
print("E=$E R=$R  %7.0f steps/s  (min %7.0f max %7.0f)  %.3f ms/round" % (d["value"], d["reps"]["value_min"], d["reps"]["value_max"], d["ms_per_step"]), flush=True)
PY
done
