set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_19
mkdir -p $O
python - > $O/solve_small_check.txt 2>&1 <<'PY' || { tail -20 gpurun_out/r6_19/solve_small_check.txt; exit 1; }
import os, sys, torch
sys.path.insert(0, '.')
from vmc_pde_amd import _native as nat
for P in (1, 5, 128, 129, 206, 256, 300, 384, 500, 512):
    Ppad = (P + 127) // 128 * 128
    g = torch.Generator().manual_seed(P)
    M = torch.randn(P, 2 * P + 3, generator=g, dtype=torch.float64)
    A = torch.zeros((Ppad, Ppad), dtype=torch.float64)
    A[:P, :P] = M @ M.T / (2 * P + 3) + 0.1 * torch.eye(P, dtype=torch.float64)
    for i in range(P, Ppad): A[i, i] = 1.0
    F = torch.randn(P, generator=g, dtype=torch.float64)
    Ad, Fd = A.cuda(), F.cuda()
    need = nat.lib().vmcpde_chol_workspace_bytes(P)
    ws = torch.empty(need // 8, dtype=torch.float64, device='cuda')
    info = torch.zeros(1, dtype=torch.int32, device='cuda')
    nat.call('vmcpde_cholesky', nat.ptr(Ad), Ppad, P, nat.ptr(info), nat.ptr(ws), need, nat.stream())
    outs = []
    for mode in ('1', '0'):
        os.environ['VMCPDE_SOLVE_SMALL'] = mode
        u = torch.full((P,), float('nan'), dtype=torch.float64, device='cuda')
        nat.call('vmcpde_chol_solve', nat.ptr(Ad), Ppad, P, nat.ptr(Fd), nat.ptr(u), nat.ptr(ws), need, nat.stream())
        v = Fd.clone()
        nat.call('vmcpde_chol_solve', nat.ptr(Ad), Ppad, P, nat.ptr(v), nat.ptr(v), nat.ptr(ws), need, nat.stream())     # in place
        assert torch.equal(u, v)
        outs.append(u.cpu())
    ref = torch.linalg.solve(A[:P, :P], F)
    err = float((outs[0] - ref).abs().max() / ref.abs().max())
    print(f"P={P}: one launch == block-row chain bit for bit: {torch.equal(outs[0], outs[1])}; rel. error against LAPACK {err:.1e}", flush=True)
    assert torch.equal(outs[0], outs[1]) and err < 1e-10
PY
cat $O/solve_small_check.txt
python -m pytest tests/test_gpu_solve.py tests/test_gpu_lowrank_eigh.py tests/test_gpu_headline_parity.py -x -q -m gpu > $O/t1.log 2>&1 || { tail -60 $O/t1.log; exit 1; }
tail -3 $O/t1.log
for v in 1 0; do
  VMCPDE_SOLVE_SMALL=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-beside > $O/bench_ss$v.json 2> $O/bench_ss$v.err || { tail -20 $O/bench_ss$v.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6_19/bench*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(d['value'],3), round(d['ms_per_step'],3), d['rank_stage_ms'])
PY
