#!/bin/bash
# tools/ab_algos_env.sh "VAR=a VAR=b -" : steady-state step time of the side workloads (tools/algo_run.py, 128 runs) under environment
# settings of the product library, round-robin inside one GPU call
for a in ${ALGOS:-iql td3bc edac}; do for rep in 1 2; do for kv in $1; do
  if [ "$kv" = "-" ]; then envs=""; else envs="$kv"; fi
  printf "%-6s %-30s " $a "$kv"; env $envs python3 tools/algo_run.py $a ${RUNS:-128} ${P:-1} 30 2>/dev/null | sed 's/.*precision [01]: //'
done; done; done
