#!/bin/bash
# tools/ab_algos.sh v1 v2 ... : steady-state step time of the side workloads (tools/algo_run.py, 128 runs) for several engine builds,
# round-robin inside one GPU call ("base" = product library, others = liborlengine_<v>.so from build.py --variant)
for a in ${ALGOS:-edac iql td3bc cql_h3}; do for rep in 1 2; do for v in "$@"; do
  if [ "$v" = base ]; then unset ORL_ENGINE_LIB; else export ORL_ENGINE_LIB=$PWD/offlinerl-kit_amd/liborlengine_$v.so; fi
  printf "%-6s %-6s " $a $v; python3 tools/algo_run.py $a 128 ${P:-1} 30 2>/dev/null | sed 's/.*precision [01]: //'
done; done; done
