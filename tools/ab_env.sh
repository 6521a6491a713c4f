#!/bin/bash
# tools/ab_env.sh "VAR=a VAR=b ..." [bench args]: round-robin twice over environment settings of the product library inside one GPU call
# ("-" = no setting); prints value and the per-tag times named in TAGS (default: the three big critic launches)
mkdir -p gpurun_out/abenv
TAGS=${TAGS:-critic.fwd1 critic.bwd.dgrad1 critic.bwd.wgrad1}
for rep in 1 2; do for kv in $1; do
  if [ "$kv" = "-" ]; then envs=""; else envs="$kv"; fi
  env $envs python bench.py --steps 40 --warmup 20 --min-reps 3 --min-seconds 1 --no-sides --no-cpu-baseline --engines-per-gpu ${E:-1} --runs-per-gpu ${R:-128} --profile-steps 10 --profile-dump gpurun_out/abenv/tags.txt ${@:2} > gpurun_out/abenv/out.json 2> gpurun_out/abenv/err.txt || { tail -3 gpurun_out/abenv/err.txt; exit 1; }
  python - "$kv" $TAGS <<'PY'
import json, sys
d = json.loads(open("gpurun_out/abenv/out.json").read().strip().split("\n")[-1])
t = {}
for line in open("gpurun_out/abenv/tags.txt"):
    f = line.split()
    if len(f) >= 4 and not line.startswith("#") and f[0] != "tag":
        try: t[f[0]] = float(f[3])
        except ValueError: pass
print("%-34s %7.0f steps/s | %s" % (sys.argv[1], d["value"], "  ".join("%s %6.1f" % (k, t.get(k, 0.0)) for k in sys.argv[2:])), flush=True)
PY
done; done
