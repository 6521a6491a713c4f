#!/bin/bash
# A/B of the z-major XCD mapping of batched few-tile GEMMs (ORL_GEMM_ZMAJOR_MAX=0 disables it), round-robin inside one GPU call
for rep in 1 2; do for zm in 16 0; do
  export ORL_GEMM_ZMAJOR_MAX=$zm
  for a in iql td3bc edac; do echo -n "zmajor_max=$zm $a: "; python3 tools/algo_run.py $a 128 1 40 | tail -1; done
  python bench.py --steps 40 --warmup 20 --min-reps 3 --min-seconds 1 --no-sides --no-cpu-baseline --engines-per-gpu 1 --runs-per-gpu 128 --profile-steps 0 > gpurun_out/ab_zm_$zm.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_zm_$zm.json").read().strip().split("\n")[-1]); print("zmajor_max=$zm cql 1x128: %.0f steps/s" % d["value"], flush=True)
PY
done; done
