#!/bin/bash
# K5C (clusters of workgroups per target group) against K5S on the C5 diameter, 3 targets checked against the CPU oracle.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp
for cl in "$@"; do
  echo "== CMDP_K5C=$cl"
  CMDP_K5C=$cl timeout -k 10 120 python3 $R/tools/run_c5.py --check 3 2>&1 | tail -1 | cut -c1-600 || exit 1
done
