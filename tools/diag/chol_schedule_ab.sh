# A/B of the Cholesky schedules: round-1 one-level schedule (32-wide diagonal block, side stream for large updates) against the
# two-level one at several super-panel widths (W = 1: one level, two tiles per workgroup).
R=${GRAFT_REPO_ROOT:-$PWD}
WS=${WS:-1 2 4 8 16}
for P in ${@:-4213 8192 16181}; do
  echo "== P=$P"
  VMCPDE_CHOL_DIAG16=0 python3 $R/tools/probe_cholesky.py $P 2>/dev/null | tail -1 | sed 's/^/round-1    /'
  python3 $R/tools/probe_cholesky.py $P 2>/dev/null | tail -1 | sed 's/^/default    /'
  for W in $WS; do
    VMCPDE_CHOL_SUPER=$W python3 $R/tools/probe_cholesky.py $P 2>/dev/null | tail -1 | sed "s/^/two-level W=$W /"
  done
done
