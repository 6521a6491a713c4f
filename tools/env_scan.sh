#!/bin/bash
# tools/env_scan.sh "VAR=val E R" ... : bench.py under different environment settings inside one gpurun call
for spec in "$@"; do set -- $spec; kv=$1; E=$2; R=$3
  env $kv python bench.py --steps 40 --warmup 20 --min-reps 5 --min-seconds 2 --no-sides --no-cpu-baseline --engines-per-gpu $E --runs-per-gpu $R --profile-steps 10 > gpurun_out/es.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/es.json").read().strip().split("\n")[-1])
t={x["name"]:x["ms_per_step"]*1e3 for x in d["roofline"]["table"]}
print("%-22s E=$E R=$R %7.0f steps/s (min %6.0f max %6.0f) | fwd1 %6.1f wgrad1 %6.1f dgrad1 %6.1f adam %6.1f | roof %.3f" % ("$kv", d["value"], d["reps"]["value_min"], d["reps"]["value_max"], t.get("critic.fwd1",0), t.get("critic.bwd.wgrad1",0), t.get("critic.bwd.dgrad1",0), t.get("adam",0), d["roofline"]["frac"]), flush=True)
PY
done
