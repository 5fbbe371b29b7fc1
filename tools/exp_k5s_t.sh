#!/bin/bash
# K5S: targets per lane (1 | 2) x wavefronts per group, C5 diameter with 3 targets checked against the CPU oracle each time.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for cfg in "1 16" "2 16" "2 8"; do
  set -- $cfg
  echo "== CMDP_K5S_T=$1 CMDP_K5S_NW=$2"
  CMDP_K5S_T=$1 CMDP_K5S_NW=$2 timeout -k 10 300 python3 $R/tools/run_c5.py --check 3 2>&1 | tail -1 | cut -c1-600
done
