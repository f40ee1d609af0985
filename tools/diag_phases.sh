#!/bin/bash
# whole-Cholesky time per cumulative chol_diag phase (VMCPDE_DIAG_PHASE = 0..5); the differences / 33 panels = per-phase cost
for ph in 0 1 2 3 4 5; do
  echo "phase $ph" >> gpurun_out/diag_phases.log
  VMCPDE_DIAG_PHASE=$ph timeout -k 10 90 python3 tools/probe_cholesky.py >> gpurun_out/diag_phases.log 2>&1 || exit 1
done
