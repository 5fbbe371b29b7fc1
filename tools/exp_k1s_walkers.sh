#!/bin/bash
# K1S tuning sweep: walker wavefronts (CMDP_K1S_NW) x lanes per instance (CMDP_K1S_TEAM; 0 = the library's choice) on the three
# stochastic-dynamics families at a benchmark-sized and a large batch.   bash tools/exp_k1s_walkers.sh "4 0" "2 0" ...
cd ${GRAFT_REPO_ROOT:-/root/repo}
for cfg in "${@:-4 0}"; do
  set -- $cfg
  export CMDP_K1S_NW=$1; if [ "$2" != "0" ]; then export CMDP_K1S_TEAM=$2; else unset CMDP_K1S_TEAM; fi
  for fam in frozenlake20 minigrid_empty8 deepsea20_prand; do
    for B in 4096 131072; do
      python tools/prof_stoch.py --family $fam --instances $B --kernel k1s --steps 2000 --launches 3 | python -c "import sys,json; l=json.loads(sys.stdin.read()); print('NW=$1 TEAM=$2', l['family'], l['instances'], l['lds_plan']['instances_per_workgroup'], '%.3g' % l['transitions_per_s'])"
    done
  done
done
