set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_15
mkdir -p $O
python - > $O/fused_reduce_check.txt 2>&1 <<'PY' || { tail -20 gpurun_out/r6_15/fused_reduce_check.txt; exit 1; }
import os, sys, torch
sys.path.insert(0, '.')
from vmc_pde_amd import _native as nat
dev = torch.device('cuda:0')
for (P, N) in ((4213, 65536), (4155, 32768), (300, 5000), (129, 2048)):
    Ppad = (P + 127) // 128 * 128
    ldj = (N + 15) // 16 * 16
    g = torch.Generator(device=dev).manual_seed(P)
    JT = torch.zeros((Ppad, ldj), dtype=torch.float64, device=dev)
    JT[:P, :N] = torch.randn((P, N), generator=g, dtype=torch.float64, device=dev)
    npk = nat.lib().vmcpde_gram_packed_doubles(P)
    ws = torch.empty(nat.lib().vmcpde_gram_workspace_bytes(P, N) // 8, dtype=torch.float64, device=dev)
    outs, times = {}, {}
    for mode in ('0', '1', '1'):
        os.environ['VMCPDE_GRAM_FUSED_REDUCE'] = mode
        packed = torch.full((npk,), float('nan'), dtype=torch.float64, device=dev)
        ws.fill_(float('nan'))
        def run():
            nat.call('vmcpde_gram', nat.ptr(JT), ldj, 0, P, N, None, None, nat.ptr(packed), nat.ptr(ws), ws.numel() * 8, nat.stream())
        run(); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record(); e1.synchronize(); ts.append(e0.elapsed_time(e1))
        outs.setdefault(mode, []).append(packed.clone()); times.setdefault(mode, []).append(sorted(ts)[2])
    same = all(torch.equal(outs['0'][0], o) for o in outs['1'])
    print(f"P={P} N={N}: separate reduce {times['0'][0]:.3f} ms, fused {times['1'][0]:.3f} / {times['1'][1]:.3f} ms, bit-identical {same}", flush=True)
    assert same
PY
cat $O/fused_reduce_check.txt
for v in 0 1; do
  VMCPDE_GRAM_FUSED_REDUCE=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-beside > $O/bench_fr$v.json 2> $O/bench_fr$v.err || { tail -20 $O/bench_fr$v.err; exit 1; }
  VMCPDE_GRAM_FUSED_REDUCE=$v python bench.py --workload C4 --steps 30 --warmup 5 --no-cpu-baseline --no-beside > $O/bench_C4_fr$v.json 2> $O/bench_C4_fr$v.err || { tail -20 $O/bench_C4_fr$v.err; exit 1; }
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r6_15/bench*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], round(d['value'],3), round(d['ms_per_step'],3), d['rank_stage_ms'])
PY
