#!/bin/bash
# tools/abn.sh v1 v2 v3 ... : round-robin twice over several engine builds inside one gpurun call ("base" = product library)
for rep in 1 2; do for v in "$@"; do
  if [ "$v" = base ]; then unset ORL_ENGINE_LIB; else export ORL_ENGINE_LIB=$PWD/offlinerl-kit_amd/liborlengine_$v.so; fi
  python bench.py --steps 40 --warmup 20 --min-reps 3 --min-seconds 1 --no-sides --no-cpu-baseline --engines-per-gpu ${E:-1} --runs-per-gpu ${R:-128} --precision ${PREC:-1} --profile-steps 10 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$v.json").read().strip().split("\n")[-1])
t={x["name"].replace("@p3", ""):x["ms_per_step"]*1e3 for x in d["roofline"]["table"]}
print("%-10s %7.0f steps/s | fwd1 %6.1f wgrad1 %6.1f dgrad1 %6.1f prepare %6.1f adam %6.1f" % ("$v", d["value"], t.get("critic.fwd1",0), t.get("critic.bwd.wgrad1",0), t.get("critic.bwd.dgrad1",0), t.get("prepare",0), t.get("adam",0)), flush=True)
PY
done; done
