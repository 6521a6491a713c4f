#!/bin/bash
# A/B: k-ranges of the 256-row wgrads of many nets (ORL_WGRAD_SMALL_KS=1 vs 2), round-robin inside one GPU call
for rep in 1 2; do for k in 2 1; do
  export ORL_WGRAD_SMALL_KS=$k
  for a in iql td3bc edac; do echo -n "small_ks=$k $a: "; python3 tools/algo_run.py $a 128 1 40 2>/dev/null | tail -1 | sed 's/.*precision 1: //'; done
done; done
