#!/bin/bash
# Round 5: what part of the steady-state Gram's time is NOT proportional to its work?  Same unit size (6554 samples per unit),
# 0.5x / 1x / 2x the units; then the same work in half-size units.   usage (GPU box): bash tools/diag/gram_units_fit.sh
R=${GRAFT_REPO_ROOT:-$PWD}
for cfg in "32768 5" "65536 10" "131072 20" "32768 10" "65536 20" "65536 9" "65536 11" "65536 12"; do
  set -- $cfg
  VMCPDE_GRAM_NSPLIT=$2 VMCPDE_GRAM_ORDER=2 python3 $R/tools/probe_gram.py 4213 $1 7 nocentre
done
