"""GPU parity: Gram / moments kernels (csrc/gram.hip) through the C ABI against numpy on identical inputs."""
import numpy as np
import pytest
import torch

from vmc_pde_amd import _native as nat

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _pad(v, m):
    return (v + m - 1) // m * m


def _jt(P, N, seed, scale=1.0, mean=0.0):
    rng = np.random.default_rng(seed)
    J = rng.standard_normal((N, P)) * scale + mean * rng.standard_normal(P)
    ldj = _pad(N, 16)
    JT = torch.zeros((_pad(P, 128), ldj), dtype=torch.float64, device=DEV)
    JT[:P, :N] = torch.as_tensor(J.T)
    return J, JT, ldj


def _gram(JT, ldj, P, N, c=None, w=None, f32=0):
    npk = nat.lib().vmcpde_gram_packed_doubles(P)
    packed = torch.full((npk,), float("nan"), dtype=torch.float64, device=DEV)
    need = nat.lib().vmcpde_gram_workspace_bytes(P, N)
    ws = torch.empty(max(need // 8, 1), dtype=torch.float64, device=DEV)
    nat.call("vmcpde_gram", nat.ptr(JT), ldj, f32, P, N, nat.ptr(c), nat.ptr(w), nat.ptr(packed), nat.ptr(ws), ws.numel() * 8, nat.stream())
    return packed


def _unpack(packed, P):
    nb = _pad(P, 128) // 128
    S = np.zeros((nb * 128, nb * 128))
    pk = packed.cpu().numpy().reshape(-1, 128, 128)
    t = 0
    for i in range(nb):
        for j in range(i + 1):
            S[i * 128:(i + 1) * 128, j * 128:(j + 1) * 128] = pk[t] if i != j else np.tril(pk[t]) + np.tril(pk[t], -1).T
            if i != j:
                S[j * 128:(j + 1) * 128, i * 128:(i + 1) * 128] = pk[t].T
            t += 1
    return S[:P, :P]


def test_mfma_f64_layout_with_asymmetric_operand():
    # exact small integers: any row/col swap in the f64 C/D lane map (cdna_hip_programming.md:159-162) shows up bit-for-bit
    P, N = 256, 64
    J = (np.arange(N)[:, None] * 3 + np.arange(P)[None, :] * 7 + (np.arange(P)[None, :] ** 2) % 5) % 11 - 5.0
    JT = torch.zeros((256, 64), dtype=torch.float64, device=DEV)
    JT[:P, :N] = torch.as_tensor(J.T)
    S = _unpack(_gram(JT, 64, P, N), P)
    assert np.array_equal(S, J.T @ J)


@pytest.mark.parametrize("P,N", [(37, 1024), (128, 16), (129, 17), (411, 1000), (640, 4099), (4213, 2048)])
def test_gram_parity_with_centring(P, N):
    J, JT, ldj = _jt(P, N, seed=P + N, mean=3.0)
    c = torch.as_tensor(J.mean(0)).to(DEV)
    S = _unpack(_gram(JT, ldj, P, N, c=c), P)
    O = J - J.mean(0)
    ref = O.T @ O
    assert np.linalg.norm(S - ref) / np.linalg.norm(ref) < 1e-13       # rel-Frobenius, SURVEY 8(d): 1e-12
    assert np.array_equal(S, S.T)
    # uncentred variant (c = NULL)
    S0 = _unpack(_gram(JT, ldj, P, N), P)
    assert np.linalg.norm(S0 - J.T @ J) / np.linalg.norm(J.T @ J) < 1e-13


@pytest.mark.parametrize("P,N", [(129, 17), (411, 1000), (640, 4099)])
def test_gram_fp32_operand_equals_fp64_path_on_fp32_values(P, N, monkeypatch):
    # config 5: the fp32 buffer is widened while staging, so on fp32-representable data all three variants
    # (plain, centred, weighted) must be bit-identical to the fp64-operand kernel; row moments likewise.  (The register-staged loop:
    # it orders its sums like the fp64 kernel.  The DMA-staged loop over sample pairs that the plain variant takes since round 5 is
    # test_gram_fp32_dma_variant's.)
    monkeypatch.setenv("VMCPDE_GRAM_F32_DMA", "0")
    J, _, ldj = _jt(P, N, seed=P * 3 + N, mean=2.0)
    J32 = J.astype(np.float32)
    Ppad = _pad(P, 128)
    JT32 = torch.zeros((Ppad, ldj), dtype=torch.float32, device=DEV)
    JT32[:P, :N] = torch.as_tensor(J32.T)
    JT64 = JT32.double()
    c = torch.as_tensor(J32.astype(np.float64).mean(0)).to(DEV)
    w = torch.as_tensor(np.random.default_rng(1).normal(size=N)).to(DEV)
    for kw in (dict(), dict(c=c), dict(c=c, w=w)):
        a = _gram(JT32, ldj, P, N, f32=1, **kw)
        b = _gram(JT64, ldj, P, N, **kw)
        assert torch.equal(a, b), kw
    e = torch.as_tensor(np.random.default_rng(2).normal(size=N)).to(DEV)
    outs = []
    for buf, f32 in ((JT32, 1), (JT64, 0)):
        cE = torch.full((1,), 0.25, dtype=torch.float64, device=DEV)          # the Eloc shift, read from cE[0]
        rs, rd = torch.empty(P, dtype=torch.float64, device=DEV), torch.empty(P, dtype=torch.float64, device=DEV)
        nat.call("vmcpde_row_moments", nat.ptr(buf), ldj, f32, P, N, nat.ptr(e), nat.ptr(c), nat.ptr(cE), nat.ptr(rs), nat.ptr(rd), nat.stream())
        outs.append((rs.clone(), rd.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("P,N", [(129, 17), (256, 64), (411, 1000), (640, 4099), (1000, 16385), (4213, 2048)])
def test_gram_fp32_dma_variant(P, N, monkeypatch):
    """jt_f32 = 1, uncentred and unweighted (BASELINE config 5 in the steady state): the DMA-staged loop over sample pairs, both floats of a
    slot widened at the fragment read and fed to fp64 MFMAs.  (a) On small integers every product and partial sum is exact in fp64, so the
    result must equal the integer Gram bit for bit -- any slip in the pair packing, the swizzled image, the fp64 C/D map or the ragged tail
    shows up; (b) on real data it is the Gram of the stored fp32 values with another ORDER of the fp64 sums: equal to the register-staged
    loop (= the fp64 kernel on the same values) to a few ulp of the entry's own scale, symmetric, deterministic."""
    rng = np.random.default_rng(P + 7 * N)
    Ppad, ldj = _pad(P, 128), _pad(N, 16)
    Ji = rng.integers(-7, 8, size=(N, P)).astype(np.float32)
    JT32 = torch.zeros((Ppad, ldj), dtype=torch.float32, device=DEV)
    JT32[:P, :N] = torch.as_tensor(Ji.T)
    S = _unpack(_gram(JT32, ldj, P, N, f32=1), P)
    assert np.array_equal(S, Ji.astype(np.float64).T @ Ji.astype(np.float64))
    Jr = (rng.standard_normal((N, P)) * rng.uniform(1e-3, 1e3, size=P)).astype(np.float32) + 0.3
    JT32[:P, :N] = torch.as_tensor(Jr.T)
    a = _gram(JT32, ldj, P, N, f32=1)
    assert torch.equal(a, _gram(JT32, ldj, P, N, f32=1))                                # deterministic
    monkeypatch.setenv("VMCPDE_GRAM_F32_DMA", "0")
    b = _gram(JT32, ldj, P, N, f32=1)
    if N % 2:
        # an odd sample count keeps the register-staged loop (round 6): the pair loop would read the float BEHIND the last sample, and vmcpde.h
        # promises a zero pad only for jt_f32 = 2 -- a caller's garbage there must not enter the sums
        assert torch.equal(a, b)
        monkeypatch.delenv("VMCPDE_GRAM_F32_DMA")
        JT32[:P, N] = 7.0
        assert torch.equal(_gram(JT32, ldj, P, N, f32=1), a)
        return
    assert not torch.equal(a, b)                                                         # (it really is another loop)
    S1, S0 = _unpack(a, P), _unpack(b, P)
    assert np.array_equal(S1, S1.T)
    # entry (p, q) is a sum of N products of size ~ |J_p| |J_q| / N each: two differently ordered fp64 sums of it differ by a few
    # sqrt(N) ulp of sqrt(S_pp S_qq) (measured 5.4e-15 at N = 4099)
    scale = np.sqrt(np.outer(np.diag(S0), np.diag(S0)))
    assert (np.abs(S1 - S0) / scale).max() < 4e-16 * np.sqrt(N) + 2e-15


@pytest.mark.parametrize("P,N", [(129, 17), (256, 64), (411, 1000), (640, 4099), (1000, 16384)])
def test_gram_f32_blocks_variant(P, N):
    """jt_f32 = 2: fp32 MFMA sums over blocks of <= 128 samples, fp64 across blocks.  (a) On small integers every partial sum is
    exact in fp32, so the result must equal the integer Gram bit for bit -- any slip in the fp32 fragment / C-D maps, the
    sample-pair packing or the tail masking shows up; (b) on real data it agrees with the fp64 kernel on the same fp32
    values to ~1e-7 (asserted 1e-6 = the config-5 tolerance for S)."""
    rng = np.random.default_rng(P + 5 * N)
    Ppad, ldj = _pad(P, 128), _pad(N, 16)
    Ji = rng.integers(-3, 4, size=(N, P)).astype(np.float32)
    JT32 = torch.zeros((Ppad, ldj), dtype=torch.float32, device=DEV)
    JT32[:P, :N] = torch.as_tensor(Ji.T)
    S = _unpack(_gram(JT32, ldj, P, N, f32=2), P)
    assert np.array_equal(S, Ji.astype(np.float64).T @ Ji.astype(np.float64))
    Jr = rng.standard_normal((N, P)).astype(np.float32) + 0.3
    JT32[:P, :N] = torch.as_tensor(Jr.T)
    S2 = _unpack(_gram(JT32, ldj, P, N, f32=2), P)
    S1 = _unpack(_gram(JT32, ldj, P, N, f32=1), P)
    rel = np.linalg.norm(S2 - S1) / np.linalg.norm(S1)
    assert rel < 1e-6 and np.array_equal(S2, S2.T)
    assert rel > 0            # (it really is the fp32-block path)
    # centred / weighted requests fall back to the widening kernel
    c = torch.as_tensor(Jr.astype(np.float64).mean(0)).to(DEV)
    assert torch.equal(_gram(JT32, ldj, P, N, c=c, f32=2), _gram(JT32, ldj, P, N, c=c, f32=1))


def test_gram_weighted_is_SExp_form():
    # tdvp.py:33,47: (logp*O)^T (logp*O)
    P, N = 200, 777
    J, JT, ldj = _jt(P, N, seed=4)
    w = np.random.default_rng(9).standard_normal(N)
    c = torch.as_tensor(J.mean(0)).to(DEV)
    S = _unpack(_gram(JT, ldj, P, N, c=c, w=torch.as_tensor(w).to(DEV)), P)
    Ow = w[:, None] * (J - J.mean(0))
    assert np.linalg.norm(S - Ow.T @ Ow) / np.linalg.norm(Ow.T @ Ow) < 1e-13


def test_gram_is_deterministic_and_linear_in_samples():
    # size-independent properties at a larger shape: bitwise reproducible; Gram(all) = Gram(first half) + Gram(second half)
    P, N = 1000, 16384
    J, JT, ldj = _jt(P, N, seed=2)
    a = _gram(JT, ldj, P, N)
    b = _gram(JT, ldj, P, N)
    assert torch.equal(a, b)
    h = N // 2
    JT2 = JT[:, h:].contiguous()
    s = _gram(JT, ldj, P, h) + _gram(JT2, h, P, N - h)
    assert float((a - s).abs().max()) < 1e-10 * float(a.abs().max())


def test_row_moments_eloc_stats_and_finalize():
    P, N = 411, 1000
    J, JT, ldj = _jt(P, N, seed=8, mean=2.0)
    rng = np.random.default_rng(1)
    E = rng.standard_normal(N) * 2 + 0.7
    e = torch.as_tensor(E).to(DEV)
    sums = torch.zeros(4, dtype=torch.float64, device=DEV)
    nat.call("vmcpde_eloc_stats", nat.ptr(e), N, nat.ptr(sums), nat.stream())
    assert np.allclose(sums.cpu().numpy(), [E.sum(), (E ** 2).sum(), np.abs(E).sum(), E.max()], rtol=1e-13)
    # shifted one-pass moments with an arbitrary (wrong) shift must still finalise to the exact two-pass result
    c = torch.as_tensor(J.mean(0) + 0.3 * rng.standard_normal(P)).to(DEV)
    cE = torch.tensor([0.2], dtype=torch.float64, device=DEV)
    rowsum = torch.empty(P, dtype=torch.float64, device=DEV)
    rowdot = torch.empty(P, dtype=torch.float64, device=DEV)
    nat.call("vmcpde_row_moments", nat.ptr(JT), ldj, 0, P, N, nat.ptr(e), nat.ptr(c), nat.ptr(cE), nat.ptr(rowsum), nat.ptr(rowdot), nat.stream())
    Jc = J - c.cpu().numpy()
    assert np.allclose(rowsum.cpu().numpy(), Jc.sum(0), rtol=1e-12, atol=1e-10)
    assert np.allclose(rowdot.cpu().numpy(), (Jc * (E - 0.2)[:, None]).sum(0), rtol=1e-12, atol=1e-10)
    packed = _gram(JT, ldj, P, N, c=c)
    Pp = _pad(P, 128)
    S = torch.zeros((Pp, Pp), dtype=torch.float64, device=DEV)
    A = torch.zeros((Pp, Pp), dtype=torch.float64, device=DEV)
    F = torch.empty(P, dtype=torch.float64, device=DEV)
    mean = torch.empty(P, dtype=torch.float64, device=DEV)
    nat.call("vmcpde_gram_finalize", nat.ptr(packed), nat.ptr(rowsum), nat.ptr(rowdot), nat.ptr(sums), nat.ptr(c), nat.ptr(c), nat.ptr(cE), P, float(N),
             1e-3, None, nat.ptr(S), Pp, nat.ptr(A), Pp, nat.ptr(F), nat.ptr(mean), nat.stream())
    Oc = J - J.mean(0)
    S_ref = Oc.T @ Oc / N
    F_ref = (Oc * (E - E.mean())[:, None]).mean(0)
    Sg = S.cpu().numpy()[:P, :P]
    assert np.linalg.norm(Sg - S_ref) / np.linalg.norm(S_ref) < 1e-12
    assert np.array_equal(Sg, Sg.T)
    assert np.allclose(F.cpu().numpy(), F_ref, rtol=1e-11, atol=1e-13)
    assert np.allclose(mean.cpu().numpy(), J.mean(0), rtol=1e-13)
    Ag = A.cpu().numpy()[:P, :P]
    assert np.allclose(Ag, Sg + 1e-3 * np.diag(np.diag(Sg)), rtol=1e-15, atol=0)   # tdvp.py:50-51
    assert float(S[P:].abs().max()) == 0.0 and float(S[:, P:].abs().max()) == 0.0   # padding stays zero (H7)
    # uncentred Gram + mean outer product removed in finalize (c_gram = NULL): same S up to the cancellation eps*(1+m^2/var)
    packed0 = _gram(JT, ldj, P, N)
    S2 = torch.zeros((Pp, Pp), dtype=torch.float64, device=DEV)
    nat.call("vmcpde_gram_finalize", nat.ptr(packed0), nat.ptr(rowsum), nat.ptr(rowdot), nat.ptr(sums), nat.ptr(c), None, nat.ptr(cE), P, float(N),
             0.0, None, nat.ptr(S2), Pp, None, 0, None, None, nat.stream())
    S2g = S2.cpu().numpy()[:P, :P]
    assert np.linalg.norm(S2g - S_ref) / np.linalg.norm(S_ref) < 1e-11


# ---- the INT8 experiment (csrc/gram_i8.hip): error-free sliced Gram, same packed sums as the uncentred fp64 kernel ------------------------
def _gram_i8(JT, ldj, P, N, slices=6, bits=8, c=None, w=None, stats=None, f32=0):
    npk = nat.lib().vmcpde_gram_packed_doubles(P)
    packed = torch.full((npk,), float("nan"), dtype=torch.float64, device=DEV)
    need = nat.lib().vmcpde_gram_i8_workspace_bytes(P, N, slices, bits)
    ws = torch.empty(need // 8 + 1, dtype=torch.float64, device=DEV)
    nat.call("vmcpde_gram_i8", nat.ptr(JT), ldj, f32, P, N, slices, bits, nat.ptr(c), nat.ptr(w), nat.ptr(packed), nat.ptr(stats), nat.ptr(ws), ws.numel() * 8, nat.stream())
    return packed


@pytest.mark.parametrize("slices,bits", [(6, 8), (6, 7), (7, 7), (4, 8)])
def test_gram_i8_lane_maps_with_asymmetric_integer_operand(slices, bits):
    # exact small integers with an asymmetric pattern: a wrong A/B lane map, C/D map, slice order or digit weight shows up bit for bit
    P, N = 300, 200          # ragged in both directions: padded rows, a partial K-block, three tile rows
    J = (np.arange(N)[:, None] * 3 + np.arange(P)[None, :] * 7 + (np.arange(P)[None, :] ** 2) % 5) % 11 - 5.0
    JT = torch.zeros((384, 208), dtype=torch.float64, device=DEV)
    JT[:P, :N] = torch.as_tensor(J.T)
    assert np.array_equal(_unpack(_gram_i8(JT, 208, P, N, slices, bits), P), J.T @ J)
    # wider integers need several digits: still exact (every kept digit-pair sum is an exact integer; nothing is dropped below 2^40)
    # (four digits keep the pairs with s + t <= 3: two-digit integers -- 11 bits -- are the widest that stay exact)
    rng = np.random.default_rng(3)
    wide = 19 if slices >= 6 else 11
    J = rng.integers(-(1 << wide), 1 << wide, (N, P)).astype(np.float64)
    JT[:P, :N] = torch.as_tensor(J.T)
    assert np.array_equal(_unpack(_gram_i8(JT, 208, P, N, slices, bits), P), J.T @ J)


@pytest.mark.parametrize("P,N", [(37, 1024), (129, 17), (411, 1000), (640, 4099), (4213, 2048), (1030, 40000)])
def test_gram_i8_matches_fp64_kernel(P, N):
    """Random Jacobians with row scales over six decades and non-zero means: the sliced sums against the fp64 MFMA kernel (itself
    1e-13 from numpy, test_gram_parity_with_centring).  Six balanced 8-bit digits: rel-Frobenius <= 1e-12 (SURVEY 8d asks 1e-12 of the
    Gram kernels; measured 1e-13 at 4096 x 65536, profiles/r03_gram_i8_microbench.json); seven 7-bit digits 1e-13."""
    rng = np.random.default_rng(P + N)
    J, JT, ldj = _jt(P, N, seed=P * 3 + N, mean=0.7)
    scale = 10.0 ** rng.uniform(-3, 3, P)
    JT[:P] *= torch.as_tensor(scale, device=DEV)[:, None]
    ref = _unpack(_gram(JT, ldj, P, N), P)
    d = np.sqrt(np.diag(ref))
    for slices, bits, tol in ((6, 8, 1e-12), (7, 7, 1e-13)):
        got = _unpack(_gram_i8(JT, ldj, P, N, slices, bits), P)
        # scale-free (correlation form): the rows span six decades, so the plain Frobenius norm would only see the largest
        err = np.linalg.norm((got - ref) / np.outer(d, d)) / np.linalg.norm(ref / np.outer(d, d))
        assert err < tol, (slices, bits, err)
        assert np.array_equal(got, got.T)


def test_gram_i8_is_deterministic_and_handles_zero_and_constant_rows():
    P, N = 200, 3000
    J, JT, ldj = _jt(P, N, seed=5)
    JT[7] = 0.0                      # a dead parameter: exponent 0, all digits 0
    JT[11, :N] = 1.0                 # the appended "ones" row of the fused-moments Gram: one non-zero digit
    JT[13, :N] = -2.5e-150           # tiny values: 2^shift alone would overflow the exponent range
    a = _gram_i8(JT, ldj, P, N)
    b = _gram_i8(JT, ldj, P, N)
    assert torch.equal(a, b)         # integer atomics and fixed-order reductions: bitwise reproducible
    S = _unpack(a, P)
    ref = _unpack(_gram(JT, ldj, P, N), P)
    assert np.all(S[7] == 0.0) and S[11, 11] == float(N) and np.isfinite(S).all()
    assert abs(S[11, 3] - ref[11, 3]) <= 1e-12 * np.sqrt(ref[11, 11] * ref[3, 3])
    assert abs(S[13, 13] / ref[13, 13] - 1) < 1e-12
    with pytest.raises(nat.VmcpdeError, match="supported"):
        _gram_i8(JT, ldj, P, N, slices=5, bits=8)


def test_gram_i8_centred_and_weighted_match_fp64_kernels():
    """The centring shift and the per-sample weight (tdvp.py:42,47: SExp, rhoVar) applied while slicing: (J - c) w on the int8 pipe
    against the register-staged fp64 kernels."""
    P, N = 411, 5000
    rng = np.random.default_rng(2)
    J, JT, ldj = _jt(P, N, seed=9, mean=4.0)
    c = torch.as_tensor(J.mean(0)).to(DEV)
    w = torch.as_tensor(rng.standard_normal(N) * 3.0).to(DEV)
    for cc, ww in ((c, None), (None, w), (c, w)):
        ref = _unpack(_gram(JT, ldj, P, N, c=cc, w=ww), P)
        got = _unpack(_gram_i8(JT, ldj, P, N, c=cc, w=ww), P)
        assert np.linalg.norm(got - ref) / np.linalg.norm(ref) < 1e-12


def test_gram_i8_reports_heavy_tails():
    """stats[0] = the largest row maximum / row rms (within a factor two): ~5-10 for Gaussian rows, ~sqrt(N) (its upper bound) for a row
    that is one outlier."""
    P, N = 130, 20000
    J, JT, ldj = _jt(P, N, seed=4)
    st = torch.zeros(1, dtype=torch.float64, device=DEV)
    _gram_i8(JT, ldj, P, N, stats=st)
    assert 3.0 < float(st[0]) < 20.0
    JT[17, 5] = 1e6
    _gram_i8(JT, ldj, P, N, stats=st)
    assert 100.0 < float(st[0]) < 300.0            # sqrt(20000) = 141, times the head-room factor


@pytest.mark.parametrize("P,N", [(129, 1000), (411, 5000), (1030, 40000)])
def test_gram_i8_four_digits_on_an_fp32_jacobian(P, N):
    """BASELINE config 5 on the int8 pipe: an fp32 Jacobian sliced into four 8-bit digits (31 bits: every stored value to 2^-31 of its row
    maximum, 10 digit pairs).  Against the widening fp64 kernel on the same fp32 values: 1e-8 in correlation form (measured 6e-9 at
    4096 x 65536 -- the size of the fp32 storage error itself, which is what config 5 is about); bit-exact on integer data."""
    rng = np.random.default_rng(P)
    J, JT, ldj = _jt(P, N, seed=P + 5, mean=0.3)
    scale = 10.0 ** rng.uniform(-2, 2, P)
    J32 = (JT * torch.as_tensor(scale.tolist() + [1.0] * (JT.shape[0] - P), device=DEV)[:, None]).float().contiguous()
    ref = _unpack(_gram(J32, ldj, P, N, f32=1), P)
    got = _unpack(_gram_i8(J32, ldj, P, N, slices=4, bits=8, f32=1), P)
    d = np.sqrt(np.diag(ref))
    assert np.linalg.norm((got - ref) / np.outer(d, d)) / np.linalg.norm(ref / np.outer(d, d)) < 1e-8
    Ji = torch.zeros_like(J32)
    Ji[:P, :N] = torch.as_tensor(rng.integers(-100, 101, (P, N)).astype(np.float32)).to(DEV)
    assert np.array_equal(_unpack(_gram_i8(Ji, ldj, P, N, slices=4, bits=8, f32=1), P), _unpack(_gram(Ji, ldj, P, N, f32=1), P))
    with pytest.raises(nat.VmcpdeError, match="fp32 Jacobian"):
        _gram_i8(J32, ldj, P, N, slices=7, bits=7, f32=1)


@pytest.mark.parametrize("P,N,groups", [(4215, 4096, [(0, 17), (17, 24), (24, 29), (29, 33)]), (300, 1000, [(0, 1), (1, 3)]), (129, 70, [(1, 2), (0, 1)]),
                                        (16183, 512, [(0, 64), (64, 90), (90, 110), (110, 127)])])
def test_gram_by_tile_rows_is_the_whole_gram_bit_for_bit(P, N, groups):
    """vmcpde_gram_rows (ABI 5): the steady-state Gram for the tiles of a few tile rows -- one contiguous piece of the packed buffer.  A
    partition of the tile rows, in any order, reproduces vmcpde_gram's buffer exactly (same sample splits, same per-unit arithmetic):
    what lets a sharded step all-reduce finished row groups while the next is being formed (TDVP(overlapAllreduce=True))."""
    from vmc_pde_amd import _native as nat
    L = nat.lib()
    g = torch.Generator(device="cuda").manual_seed(P)
    ldj = (N + 15) // 16 * 16
    Ppad = (P + 127) // 128 * 128
    JT = torch.zeros((Ppad, ldj), dtype=torch.float64, device="cuda")
    JT[:P, :N] = torch.randn((P, N), generator=g, dtype=torch.float64, device="cuda")
    npk = L.vmcpde_gram_packed_doubles(P)
    ws = torch.empty(max(L.vmcpde_gram_workspace_bytes(P, N) // 8, 1), dtype=torch.float64, device="cuda")
    whole = torch.zeros(npk, dtype=torch.float64, device="cuda")
    nat.call("vmcpde_gram", nat.ptr(JT), ldj, 0, P, N, None, None, nat.ptr(whole), nat.ptr(ws), ws.numel() * 8, nat.stream())
    parts = torch.full((npk,), float("nan"), dtype=torch.float64, device="cuda")
    tri = lambda r: r * (r + 1) // 2
    for r0, r1 in groups:
        before = parts.clone()
        nat.call("vmcpde_gram_rows", nat.ptr(JT), ldj, P, N, r0, r1, nat.ptr(parts), nat.ptr(ws), ws.numel() * 8, nat.stream())
        lo, hi = tri(r0) * 128 * 128, tri(r1) * 128 * 128
        touched = ~((parts == before) | (torch.isnan(parts) & torch.isnan(before)))
        assert not bool(touched[:lo].any()) and not bool(touched[hi:].any())          # a group writes its own piece only
    assert torch.equal(parts, whole)
    with pytest.raises(nat.VmcpdeError, match="tile rows"):
        nat.call("vmcpde_gram_rows", nat.ptr(JT), ldj, P, N, 2, 2, nat.ptr(parts), nat.ptr(ws), ws.numel() * 8, nat.stream())
    with pytest.raises(nat.VmcpdeError, match="tile rows"):
        nat.call("vmcpde_gram_rows", nat.ptr(JT), ldj, P, N, 0, Ppad // 128 + 1, nat.ptr(parts), nat.ptr(ws), ws.numel() * 8, nat.stream())
