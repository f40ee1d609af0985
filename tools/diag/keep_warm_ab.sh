#!/bin/bash
# Round 5 prototype A/B: a flag-stopped compute-only kernel (tools/microbench/keep_warm.hip) on a side stream beside the solve's dependent
# chains, so that the chip's power state does not ramp down before the next S-build (tools/diag/gram_between_phases_probe.py: 8 ms of idling
# cost the next Gram 1.8 ms).  VMCPDE_KW = mode(0 bf16 MFMA, 1 fp32 VALU),burst,sleep16,workgroups,threads
#
# NOT REPRODUCIBLE FROM HEAD: the hook in TDVP.solve that read VMCPDE_KW and launched libkeepwarm.so's kernel was a prototype and was
# removed with the negative result (spectral loop -2.6 %, nothing elsewhere); no code in the tree reads VMCPDE_KW any more, so every run
# below now measures the same thing.  The script and profiles/r05_keep_warm_ab.txt are kept as the record of that removed experiment.
R=${GRAFT_REPO_ROOT:-$PWD}
run() { echo "== $ARGS $*"; env "$@" python3 $R/bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-beside ${ARGS} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), [round(v,3) for v in d['rank_stage_ms'][0].values()], round(d['step_span_ms']['median'],3))"; }
for ARGS in "--solver eigh" "--workload C4" ""; do
  run VMCPDE_KW=
  for kw in 0,16,0,256,256 0,16,4,256,256 0,16,16,256,256 0,16,0,1024,256 1,16,0,256,256 1,16,0,1024,256 0,16,2,512,64; do run VMCPDE_KW=$kw; done
done
