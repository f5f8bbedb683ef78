"""GPU parity tests of the individual HIP kernels (through the C ABI) against the CPU oracle
and numpy/scipy.  Run with ``pytest -m gpu`` on an MI355X."""
import numpy as np
import pytest
import scipy.linalg as la
import scipy.sparse as sp

import superscreen_oracle as orc

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from superscreen_amd import kernels

    return kernels


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.fixture(scope="module")
def disk():
    from superscreen_amd import synthetic

    sites, elements, dr = synthetic.ring_disk_mesh(14)  # n = 631 (odd, not a tile multiple)
    mesh = orc.make_mesh(sites, elements)
    return sites, elements, mesh


@pytest.mark.parametrize("dtype,tol", [("float64", 2e-14), ("float32", 1e-6)])
def test_q_assemble(K, disk, dtype, tol):
    sites, elements, mesh = disk
    n = len(sites)
    C = orc.C_vector(sites)
    Q, qd = K.q_assemble(dev(sites), dev(mesh.weights), dev(C), dtype)
    Qh = Q.cpu().numpy()[:, :n]
    assert relerr(qd.cpu().numpy(), np.diag(mesh.Q)) < 2e-14
    # off-diagonal entries are tiny next to the diagonal: compare them on their own scale
    off = ~np.eye(n, dtype=bool)
    assert np.max(np.abs(Qh[off] - mesh.Q[off]) / np.abs(mesh.Q[off])) < tol
    assert relerr(np.diag(Qh), np.diag(mesh.Q)) < tol


def test_q_assemble_golden(K, golden):
    d = golden("disk_K10.npz")
    Q, qd = K.q_assemble(dev(d["sites"]), dev(d["weights"]), dev(d["C"]), "float64")
    n = len(d["sites"])
    assert relerr(Q.cpu().numpy()[:, :n], d["Q"]) < 1e-13
    assert relerr(qd.cpu().numpy(), d["Q_diag"]) < 1e-13


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-13), ("float32", 2e-6)])
def test_system_assemble(K, disk, dtype, tol):
    sites, elements, mesh = disk
    n = len(sites)
    rng = np.random.default_rng(1)
    Lam = rng.uniform(0.05, 0.2, n)
    ix = np.sort(rng.choice(n, size=401, replace=False)).astype(np.int64)
    hole = np.arange(5, 42, dtype=np.int64)
    lap = mesh.laplacian.tocsr()
    lap.sort_indices()
    _, qd = K.q_assemble(dev(sites), dev(mesh.weights), dev(orc.C_vector(sites)), "float64", want_Q=False)
    args = (dev(sites), dev(mesh.weights), qd, dev(Lam), dev(lap.indptr.astype(np.int64)),
            dev(lap.indices.astype(np.int64)), dev(lap.data))
    npdt = np.dtype(dtype)
    Qc, wc, Lc, lapc = mesh.Q.astype(npdt), mesh.weights.astype(npdt), Lam.astype(npdt), lap.astype(npdt)
    A_ref = orc.build_system_2d(Qc, wc, Lc, lapc, ix)
    A = K.system_assemble(*args, dev(ix), dev(ix), sign=-1.0, dtype=dtype)
    assert relerr(A.cpu().numpy()[:, :len(ix)], -A_ref) < tol
    Ah_ref = orc.build_system_1d(Qc, wc, Lc, lapc, hole)
    Ah = K.system_assemble(*args, None, dev(hole), sign=1.0, dtype=dtype)
    assert relerr(Ah.cpu().numpy()[:, :len(hole)], Ah_ref) < tol
    # single column (ld = 1)
    one = np.array([77], dtype=np.int64)
    A1 = K.system_assemble(*args, None, dev(one), sign=1.0, dtype=dtype)
    assert relerr(A1.cpu().numpy()[:, 0], orc.build_system_1d(Qc, wc, Lc, lapc, one)[:, 0]) < tol


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-13), ("float32", 1e-5)])
@pytest.mark.parametrize("shape", [(300, 200, 64), (129, 257, 203), (1000, 1, 256), (77, 130, 5), (512, 384, 256)])
def test_gemm(K, dtype, tol, shape):
    M, N, Kd = shape
    rng = np.random.default_rng(2)
    A = rng.standard_normal((M, Kd)).astype(dtype)
    B = rng.standard_normal((Kd, N)).astype(dtype)
    C = rng.standard_normal((M, N)).astype(dtype)
    Cd = dev(C)
    K.gemm(dev(A), dev(B), Cd, M, N, Kd, alpha=-1.0, beta=1.0)
    ref = C.astype(np.float64) - A.astype(np.float64) @ B.astype(np.float64)
    assert relerr(Cd.cpu().numpy(), ref) < tol
    Cd2 = torch.full((M, N), float("nan"), dtype=Cd.dtype, device="cuda")
    K.gemm(dev(A), dev(B), Cd2, M, N, Kd, alpha=2.0, beta=0.0)
    assert relerr(Cd2.cpu().numpy(), 2.0 * (A.astype(np.float64) @ B.astype(np.float64))) < tol


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-11), ("float32", 2e-3)])
@pytest.mark.parametrize("n", [50, 64, 257, 300, 777, 1500])
def test_lu_factor_solve_random(K, dtype, tol, n):
    """General (not diagonally dominant) matrices: pivoting must match LAPACK's choices."""
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)).astype(dtype)
    ld = K.padded_ld(n, dtype)
    Ad = torch.zeros((n, ld), dtype=getattr(torch, dtype), device="cuda")
    Ad[:, :n] = dev(A)
    f = K.lu_factor(Ad, n)
    assert f.info == 0
    lu_ref, piv_ref = la.lu_factor(A)
    if dtype == "float64":
        assert np.array_equal(f.ipiv.cpu().numpy(), piv_ref)
        assert relerr(f.lu.cpu().numpy()[:, :n], lu_ref) < 1e-9
    for nrhs in (1, 3):
        b = rng.standard_normal((n, nrhs)).astype(dtype)
        bd = dev(b[:, 0]) if nrhs == 1 else dev(b)
        x = K.lu_solve(f, bd).cpu().numpy().reshape(n, nrhs)
        ref = np.linalg.solve(A.astype(np.float64), b.astype(np.float64))
        assert relerr(x, ref) < tol * max(1.0, np.linalg.cond(A.astype(np.float64)) / 1e3)


def test_lu_singular_info(K):
    n = 100
    A = np.random.default_rng(0).standard_normal((n, n))
    A[:, 40] = 0.0
    ld = K.padded_ld(n, "float64")
    Ad = torch.zeros((n, ld), dtype=torch.float64, device="cuda")
    Ad[:, :n] = dev(A)
    f = K.lu_factor(Ad, n)
    assert f.info == 41  # LAPACK: U[40, 40] is exactly zero


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-13), ("float32", 1e-5)])
def test_gemv_and_vector_kernels(K, dtype, tol):
    rng = np.random.default_rng(3)
    nr, nc = 517, 1031
    npdt = np.dtype(dtype)
    M = rng.standard_normal((nr, nc)).astype(npdt)
    ld = K.padded_ld(nc, dtype)
    Md = torch.zeros((nr, ld), dtype=getattr(torch, dtype), device="cuda")
    Md[:, :nc] = dev(M)
    x = rng.standard_normal(nc).astype(npdt)
    s = rng.uniform(0.5, 2, nc).astype(npdt)
    y = K.gemv(Md, nr, nc, dev(x), xscale=dev(s))
    assert relerr(y.cpu().numpy(), M.astype(float) @ (s.astype(float) * x)) < tol
    # unaligned ld + gathered x + accumulate
    big = rng.standard_normal(4000).astype(npdt)
    idx = np.sort(rng.choice(4000, nc, replace=False)).astype(np.int64)
    y0 = rng.standard_normal(nr).astype(npdt)
    yd = dev(y0)
    K.gemv(dev(M), nr, nc, dev(big), xidx=dev(idx), y=yd, alpha=-1.0, beta=1.0)
    assert relerr(yd.cpu().numpy(), y0 - M.astype(float) @ big[idx].astype(float)) < tol
    # rhs gather / scatter / index add / scale
    n = 900
    applied, other, ha = (rng.standard_normal(n).astype(npdt) for _ in range(3))
    ix = np.sort(rng.choice(n, 400, replace=False)).astype(np.int64)
    h = K.film_rhs(dev(applied), dev(other), dev(ha), dev(ix))
    assert np.array_equal(h.cpu().numpy(), (applied + other)[ix] - ha[ix])
    h2 = K.film_rhs(dev(applied), None, dev(ha), dev(ix))
    assert np.array_equal(h2.cpu().numpy(), applied[ix] - ha[ix])
    g = dev(applied.copy())
    K.scatter_add(g, dev(ix), h)
    ref = applied.copy(); ref[ix] += (applied + other)[ix] - ha[ix]
    assert np.array_equal(g.cpu().numpy(), ref)
    K.index_add_scalar(g, dev(ix[:10]), 2.5)
    ref[ix[:10]] += npdt.type(2.5)
    assert np.array_equal(g.cpu().numpy(), ref)
    assert np.array_equal(K.scale(g, 0.5).cpu().numpy(), (ref * npdt.type(0.5)))


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_current_density_and_self_field(K, disk, dtype):
    sites, elements, mesh = disk
    n = len(sites)
    rng = np.random.default_rng(4)
    g = rng.standard_normal(n).astype(dtype)
    gx, gy = mesh.gradient_x.tocsr(), mesh.gradient_y.tocsr()
    pattern = (abs(gx) + abs(gy)).tocsr()
    pattern.sort_indices()
    rows = np.repeat(np.arange(n), np.diff(pattern.indptr))
    vx = np.asarray(gx[rows, pattern.indices]).ravel()
    vy = np.asarray(gy[rows, pattern.indices]).ravel()
    J = K.current_density(dev(pattern.indptr.astype(np.int64)), dev(pattern.indices.astype(np.int64)),
                          dev(vx), dev(vy), dev(g))
    gd = g.astype(np.float64)
    ref = np.array([gy @ gd, -(gx @ gd)]).T
    assert relerr(J.cpu().numpy(), ref) < 1e-13
    _, qd = K.q_assemble(dev(sites), dev(mesh.weights), dev(orc.C_vector(sites)), "float64", want_Q=False)
    sf = K.self_field(dev(sites), dev(mesh.weights), qd, dev(g), alpha=0.25)
    ref_sf = 0.25 * (mesh.Q @ (mesh.weights * gd))
    assert relerr(sf.cpu().numpy(), ref_sf) < (1e-12 if dtype == "float64" else 1e-6)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_biot_savart_golden(K, golden, dtype):
    d = golden("biot_savart.npz")
    areas = d["areas"].astype(dtype)
    for tag in ("dz05", "dz0_disjoint", "dzneg"):
        za, zb, shift = d[f"args_{tag}"]
        tgt = d["sites2"] + np.array([shift, 0.0])
        out = torch.zeros(len(tgt), dtype=getattr(torch, dtype), device="cuda")
        K.biot_savart(dev(d["sites1"]), dev(areas), dev(d["J"]), dev(tgt), zb - za, out, accumulate=False)
        ref = orc.biot_savart_film_to_film(film1_sites=d["sites1"], film1_z0=za, film1_areas=areas.astype(float),
                                           film1_J=d["J"], film2_sites=tgt, film2_z0=zb)
        tol = 1e-12 if dtype == "float64" else 1e-6
        assert relerr(out.cpu().numpy(), ref) < tol
        if dtype == "float64":
            assert relerr(out.cpu().numpy(), d[f"H_{tag}"]) < 1e-12
        # source slices + accumulate == full sum
        ns = len(d["sites1"])
        acc = torch.zeros_like(out)
        for b, e in ((0, 100), (100, 101), (101, ns)):
            K.biot_savart(dev(d["sites1"]), dev(areas), dev(d["J"]), dev(tgt), zb - za, acc,
                          accumulate=True, src_begin=b, src_end=e)
        assert relerr(acc.cpu().numpy(), ref) < tol


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-13), ("float32", 1e-5)])
@pytest.mark.parametrize("case", [(0, 1, False, 300, 200, 64), (0, 1, True, 517, 517, 256), (0, 1, True, 1024, 1024, 64),
                                  (1, 0, False, 640, 3, 256), (1, 0, False, 256, 130, 203), (0, 0, False, 384, 256, 32),
                                  (1, 1, False, 130, 140, 50)])
def test_gemm_ex(K, dtype, tol, case):
    opA, opB, lower, M, N, Kd = case
    rng = np.random.default_rng(5)
    A = rng.standard_normal((M, Kd) if opA == 0 else (Kd, M)).astype(dtype)
    B = rng.standard_normal((Kd, N) if opB == 0 else (N, Kd)).astype(dtype)
    C = rng.standard_normal((M, N)).astype(dtype)
    Cd = dev(C)
    K.gemm_ex(opA, opB, lower, dev(A), dev(B), Cd, M, N, Kd, alpha=-1.0, beta=1.0)
    Am = A.astype(np.float64) if opA == 0 else A.astype(np.float64).T
    Bm = B.astype(np.float64) if opB == 0 else B.astype(np.float64).T
    ref = C.astype(np.float64) - Am @ Bm
    got = Cd.cpu().numpy().astype(np.float64)
    if lower:
        mask = np.tril(np.ones((M, N), dtype=bool))
        assert np.max(np.abs(got[mask] - ref[mask])) / np.max(np.abs(ref)) < tol
        # strictly above the 128-tile diagonal nothing may have been touched
        tm, tn = np.arange(M)[:, None] // 128, np.arange(N)[None, :] // 128
        untouched = tn > tm
        assert np.array_equal(got[untouched], C.astype(np.float64)[untouched])
    else:
        assert relerr(got, ref) < tol


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-12), ("float32", 2e-4)])
@pytest.mark.parametrize("n", [40, 64, 200, 257, 777, 1500])
def test_cholesky_factor_solve(K, dtype, tol, n):
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, n))
    S = (X @ X.T / n + np.eye(n) * 2.0).astype(dtype)
    npad = K.chol_padded_n(n)
    ld = K.padded_ld(npad, dtype)
    Sd = torch.full((npad, ld), float("nan"), dtype=getattr(torch, dtype), device="cuda")
    Sd[:n, :n] = dev(np.tril(S))  # only the lower triangle is given; the padding is the library's job
    f = K.chol_factor(Sd, n)
    assert f.info == 0
    L = np.tril(f.L.cpu().numpy()[:n, :n].astype(np.float64))
    assert relerr(L, np.linalg.cholesky(S.astype(np.float64))) < tol * 10
    for nrhs in (1, 5):
        b = rng.standard_normal((n, nrhs)).astype(dtype)
        bd = dev(b[:, 0].copy()) if nrhs == 1 else dev(b)
        x = K.chol_solve(f, bd).cpu().numpy().reshape(n, nrhs)
        assert relerr(x, np.linalg.solve(S.astype(np.float64), b.astype(np.float64))) < tol * 10
    # not positive definite -> info > 0
    Sbad = S.copy()
    Sbad[n // 2, n // 2] = -1.0
    Sd2 = torch.zeros((npad, ld), dtype=getattr(torch, dtype), device="cuda")
    Sd2[:n, :n] = dev(np.tril(Sbad))
    assert K.chol_factor(Sd2, n).info > 0


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-11), ("float32", 2e-3)])
@pytest.mark.parametrize("n", [5000, 9300])
def test_cholesky_solve_block_sizes_agree(K, dtype, tol, n):
    """ssa_chol_factor_batch_blk / ssa_chol_solve*_blk (ABI 6): the triangular solves on 4096-row diagonal blocks
    (default) or on smaller ones -- 2048 is what a factorization that serves few solves uses: the inverse levels from
    the block size upwards are then not built.  Same factor (bit for bit in float64: the schedule does not depend on
    it), same solutions to rounding for 1 and several right-hand sides, single and batched; a last block that is
    partial at either size (n = 5 000: 904 rows beyond 4096; n = 9 300: 1 108 beyond 8 192)."""
    tdt = torch.float64 if dtype == "float64" else torch.float32
    g = torch.Generator(device="cuda").manual_seed(n)
    U = torch.randn(n, 24, dtype=torch.float64, device="cuda", generator=g)
    S = U @ U.T / 24
    S.diagonal().add_(2.0 + torch.rand(n, dtype=torch.float64, device="cuda", generator=g))

    def factor(block):
        npad = K.chol_padded_n(n)
        t = torch.zeros((npad, K.padded_ld(npad, dtype)), dtype=tdt, device="cuda")
        t[:n, :n] = torch.tril(S).to(tdt)
        f = K.chol_factor(t, n, solve_block=block)
        assert f.info == 0 and f.solve_block == block
        return f

    ref = factor(4096)
    x = torch.randn(n, 3, dtype=torch.float64, device="cuda", generator=g)
    B = (S @ x).to(tdt)
    want1 = K.chol_solve(ref, B[:, 0].contiguous().clone()).double()
    want3 = K.chol_solve(ref, B.clone()).double()
    assert float((want3 - x).abs().max() / x.abs().max()) < tol
    for block in (2048, 1024, 256):
        f = factor(block)
        if dtype == "float64":
            assert torch.equal(f.L[:n, :n], ref.L[:n, :n])
        got1 = K.chol_solve(f, B[:, 0].contiguous().clone()).double()
        got3 = K.chol_solve(f, B.clone()).double()
        npad = K.chol_padded_n(n)
        padded = torch.zeros(npad, dtype=tdt, device="cuda")
        padded[:n] = B[:, 1]
        gotb = K.chol_solve_batch([f], [padded], padded=True)[0].double()
        scale = float(x.abs().max())
        assert float((got1 - want1).abs().max()) / scale < tol * 1e-2 + 1e-13
        assert float((got3 - want3).abs().max()) / scale < tol * 1e-2 + 1e-13
        assert float((gotb - want3[:, 1]).abs().max()) / scale < tol * 1e-2 + 1e-13
    with pytest.raises(Exception):
        factor(3000)           # not a power-of-two multiple of 256
    with pytest.raises(ValueError):
        K.chol_solve_batch([factor(2048), ref], [torch.zeros(K.chol_padded_n(n), dtype=tdt, device="cuda")] * 2, padded=True)


def test_system_assemble_symmetric_scaled(K, disk):
    """S = diag(w) A is symmetric; the lower_only assembly equals tril(w_i * A_ij)."""
    sites, elements, mesh = disk
    n = len(sites)
    rng = np.random.default_rng(9)
    ix = np.sort(rng.choice(n, size=333, replace=False)).astype(np.int64)
    lap = mesh.laplacian.tocsr()
    lap.sort_indices()
    Lam = 0.1 * np.ones(n)
    _, qd = K.q_assemble(dev(sites), dev(mesh.weights), dev(orc.C_vector(sites)), "float64", want_Q=False)
    A = orc.build_system_2d(mesh.Q, mesh.weights, Lam, lap, ix)
    S_ref = mesh.weights[ix][:, None] * A
    assert relerr(S_ref, S_ref.T) < 1e-13
    S = K.system_assemble(dev(sites), dev(mesh.weights), qd, dev(Lam), dev(lap.indptr.astype(np.int64)),
                          dev(lap.indices.astype(np.int64)), dev(lap.data), dev(ix), dev(ix), sign=1.0,
                          dtype="float64", row_scale=dev(mesh.weights), lower_only=True,
                          ld=K.padded_ld(K.chol_padded_n(len(ix)), "float64"), alloc_rows=K.chol_padded_n(len(ix)))
    got = S.cpu().numpy()[:len(ix), :len(ix)]
    mask = np.tril(np.ones_like(S_ref, dtype=bool))
    assert np.max(np.abs(got[mask] - S_ref[mask])) / np.max(np.abs(S_ref)) < 1e-13
    # and the Cholesky route reproduces the LU route: gf = lu_solve(lu_factor(-A), h) = -S^-1 (w h)
    h = rng.standard_normal(len(ix))
    f = K.chol_factor(S, len(ix))
    assert f.info == 0
    x = K.chol_solve(f, dev(-mesh.weights[ix] * h)).cpu().numpy()
    assert relerr(x, la.lu_solve(la.lu_factor(-A), h)) < 1e-12


@pytest.mark.parametrize("n", [4096, 5000, 9000])
def test_cholesky_block_solves(K, n):
    """Orders beyond one 4096-row solve block: level-batched block inverses, L^T mirrored into the
    upper triangle, triangular GEMV chain (single rhs) and GEMM chain (several)."""
    rng = np.random.default_rng(n)
    U = rng.standard_normal((n, 24))
    S = U @ U.T / 24 + np.diag(2.0 + rng.random(n))
    npad = K.chol_padded_n(n)
    Sd = torch.full((npad, K.padded_ld(npad, "float64")), float("nan"), dtype=torch.float64, device="cuda")
    Sd[:n, :n] = dev(np.tril(S))
    f = K.chol_factor(Sd, n)
    assert f.info == 0
    F = f.L.cpu().numpy()[:n, :n]
    Lref = np.linalg.cholesky(S)
    assert relerr(np.tril(F), Lref) < 1e-12
    assert np.array_equal(np.triu(F, 1), np.tril(F, -1).T)  # the mirror is a copy, not a recomputation
    for nrhs in (1, 3, 33, 64, 130):   # GEMV chain | skinny kernel (1, 3, 4 column blocks) | tiled GEMM + split-K
        b = rng.standard_normal((n, nrhs))
        x = K.chol_solve(f, dev(b[:, 0].copy()) if nrhs == 1 else dev(b)).cpu().numpy().reshape(n, nrhs)
        assert relerr(S @ x, b) < 1e-11
        assert relerr(x, la.cho_solve((Lref, True), b)) < 1e-11


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_cholesky_solve_batch_equals_single(K, dtype):
    """ssa_chol_solve_batch (the block steps of several single-right-hand-side solves side by side in one launch each)
    is bit-identical to separate ssa_chol_solve calls, for factors with different numbers of 4096-blocks."""
    rng = np.random.default_rng(11)
    tdt = getattr(torch, dtype)
    factors, rhs = [], []
    for n in (9000, 1500, 13111, 4096, 300):
        U = rng.standard_normal((n, 16))
        S = U @ U.T / 16 + np.diag(2.0 + rng.random(n))
        npad = K.chol_padded_n(n)
        Sd = torch.zeros((npad, K.padded_ld(npad, dtype)), dtype=tdt, device="cuda")
        Sd[:n, :n] = dev(np.tril(S).astype(dtype))
        f = K.chol_factor(Sd, n)
        assert f.info == 0
        factors.append(f)
        rhs.append(dev(rng.standard_normal(n).astype(dtype)))
    single = [K.chol_solve(f, b.clone()) for f, b in zip(factors, rhs)]
    batch = K.chol_solve_batch(factors, [b.clone() for b in rhs])
    for a, b in zip(single, batch):
        assert torch.equal(a, b)
    again = K.chol_solve_batch(factors[:2], [b.clone() for b in rhs[:2]])
    assert torch.equal(again[0], single[0]) and torch.equal(again[1], single[1])
    # right-hand sides that already sit in padded buffers (zero tail) are solved where they are
    padded = []
    for f, b in zip(factors, rhs):
        buf = torch.zeros(K.chol_padded_n(f.n), dtype=tdt, device="cuda")
        buf[:f.n] = b
        padded.append(buf)
    inplace = K.chol_solve_batch(factors, padded, padded=True)
    for a, b, buf, f in zip(single, inplace, padded, factors):
        assert torch.equal(a, b) and b.data_ptr() == buf.data_ptr()
        assert not buf[f.n:].any()                                   # the padding stays zero: the buffer can be reused


def test_cholesky_batch_equals_single(K):
    """ssa_chol_factor_batch (films interleaved on one schedule) is bit-identical to separate calls."""
    rng = np.random.default_rng(5)
    sizes = [1500, 4400, 700, 256, 500]   # 256 / 500 / 700: one, two, three panels (edges of the two-stream chain)
    mats = []
    for n in sizes:
        U = rng.standard_normal((n, 16))
        mats.append(np.tril(U @ U.T / 16 + np.diag(1.5 + rng.random(n))))

    def buf(S):
        n = len(S)
        npad = K.chol_padded_n(n)
        t = torch.zeros((npad, K.padded_ld(npad, "float64")), dtype=torch.float64, device="cuda")
        t[:n, :n] = dev(S)
        return t

    single = [K.chol_factor(buf(S), len(S)) for S in mats]
    batch = K.chol_factor_batch([(buf(S), len(S)) for S in mats])
    for S, a, b in zip(mats, single, batch):
        n = len(S)
        assert a.info == 0 and b.info == 0
        assert torch.equal(a.L[:n, :n], b.L[:n, :n])
        rhs = dev(rng.standard_normal(n))
        assert torch.equal(K.chol_solve(a, rhs.clone()), K.chol_solve(b, rhs.clone()))
    # orders above 8192 + a panel: trailing updates applied two panels at a time (K = 512), with a phase that
    # depends on the matrix' own size only -- still bit-identical alone and in a batch, in either order
    big = []
    for n in (9300, 8800):
        U = rng.standard_normal((n, 16))
        big.append(np.tril(U @ U.T / 16 + np.diag(1.5 + rng.random(n))))
    alone = [K.chol_factor(buf(S), len(S)) for S in big]
    together = K.chol_factor_batch([(buf(S), len(S)) for S in big])
    swapped = K.chol_factor_batch([(buf(S), len(S)) for S in big[::-1]])[::-1]
    for S, a, b, c in zip(big, alone, together, swapped):
        n = len(S)
        assert a.info == 0 and torch.equal(a.L[:n, :n], b.L[:n, :n]) and torch.equal(a.L[:n, :n], c.L[:n, :n])
        x = rng.standard_normal(n)
        full = S + np.tril(S, -1).T
        assert relerr(K.chol_solve(b, dev(full @ x)).cpu().numpy(), x) < 1e-11
    del alone, together, swapped
    # one indefinite matrix in the batch is reported for that matrix only
    bad = mats[2].copy()
    bad[300, 300] = -5.0
    res = K.chol_factor_batch([(buf(mats[0]), sizes[0]), (buf(bad), sizes[2])])
    assert res[0].info == 0 and res[1].info > 0


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-11), ("float32", 2e-3)])
def test_cholesky_schedule_parts_batch_equals_single(K, dtype, tol):
    """The two parts of the schedule (panel chains beside the updates while a trailing matrix exceeds 10 240 columns,
    single-stream rounds of batched launches after that) and the switch between them: matrices that start in the
    stream part, one that runs out of panels there, one that joins for the rounds only -- factored together, alone
    and in another order they give bit-identical factors in float64 (equal to rounding in float32), and the factors
    solve S x = b."""
    tdt = torch.float64 if dtype == "float64" else torch.float32
    g = torch.Generator(device="cuda").manual_seed(11)
    sizes = [12100, 600, 11400, 5000]

    def make(n):
        U = torch.randn(n, 24, dtype=torch.float64, device="cuda", generator=g)
        S = U @ U.T / 24
        S.diagonal().add_(2.0 + torch.rand(n, dtype=torch.float64, device="cuda", generator=g))
        return S.to(tdt)

    mats = [make(n) for n in sizes]

    def buf(S):
        n = S.shape[0]
        npad = K.chol_padded_n(n)
        t = torch.zeros((npad, K.padded_ld(npad, dtype)), dtype=tdt, device="cuda")
        t[:n, :n] = torch.tril(S)
        return t

    alone = [K.chol_factor(buf(S), S.shape[0]) for S in mats]
    together = K.chol_factor_batch([(buf(S), S.shape[0]) for S in mats])
    swapped = K.chol_factor_batch([(buf(S), S.shape[0]) for S in mats[::-1]])[::-1]
    for S, a, b, c in zip(mats, alone, together, swapped):
        n = S.shape[0]
        assert a.info == 0 and b.info == 0 and c.info == 0
        if dtype == "float64":
            assert torch.equal(a.L[:n, :n], b.L[:n, :n]) and torch.equal(a.L[:n, :n], c.L[:n, :n])
        else:
            # float32 tiles add C once per launch (csrc/gemm_ops.hip): a two-panel update and two one-panel updates
            # round differently, and where a matrix switches to rounds depends on the largest matrix of the batch
            scale = float(a.L[:n, :n].abs().max())
            for other in (b, c):
                assert float((a.L[:n, :n] - other.L[:n, :n]).abs().max()) < 1e-5 * scale
        x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
        rhs = (S.double() @ x).to(tdt)
        got = K.chol_solve(b, rhs.clone()).double()
        assert float((got - x).abs().max() / x.abs().max()) < tol


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_cholesky_finishing_passes_sliced_or_whole(K, dtype, monkeypatch):
    """The finishing passes (block inverses, their transposes) go out in SLICES beside the single-stream rounds
    (gemm.hip: gemm_batched_sliced, K in chunks of 512 accumulated in order) or as whole launches after the last
    round (chol.hip, CholDebug late=1).  float64: the same bits either way; float32: equal to rounding only (its
    tile adds C once per launch, so every chunk boundary rounds once more)."""
    tdt = torch.float64 if dtype == "float64" else torch.float32
    g = torch.Generator(device="cuda").manual_seed(5)
    sizes = [9000, 8500]        # rounds from the first panel on, two full 4096-blocks final while rounds still run

    def make(n):
        U = torch.randn(n, 24, dtype=torch.float64, device="cuda", generator=g)
        S = U @ U.T / 24
        S.diagonal().add_(2.0 + torch.rand(n, dtype=torch.float64, device="cuda", generator=g))
        return S.to(tdt)

    mats = [make(n) for n in sizes]

    def run():
        bufs = []
        for S in mats:
            n = S.shape[0]
            npad = K.chol_padded_n(n)
            t = torch.zeros((npad, K.padded_ld(npad, dtype)), dtype=tdt, device="cuda")
            t[:n, :n] = torch.tril(S)
            bufs.append((t, n))
        out = K.chol_factor_batch(bufs)
        torch.cuda.synchronize()
        return out

    monkeypatch.delenv("SSA_CHOL_DEBUG", raising=False)
    sliced = run()
    monkeypatch.setenv("SSA_CHOL_DEBUG", "late=1")
    whole = run()
    monkeypatch.delenv("SSA_CHOL_DEBUG", raising=False)
    for S, a, b in zip(mats, sliced, whole):
        n = S.shape[0]
        assert a.info == 0 and b.info == 0
        used = 2 * ((K.chol_padded_n(n) + 4095) // 4096) * 4096 * 4096       # inverse blocks and their transposes
        assert torch.equal(a.L[:n, :n], b.L[:n, :n])                          # the factor itself never depends on it
        if dtype == "float64":
            assert torch.equal(a.aux[:used], b.aux[:used])
        else:
            scale = float(b.aux[:used].abs().max())
            assert float((a.aux[:used] - b.aux[:used]).abs().max()) < 1e-5 * scale
        x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
        rhs = (S.double() @ x).to(tdt)
        for f in (a, b):
            got = K.chol_solve(f, rhs.clone()).double()
            assert float((got - x).abs().max() / x.abs().max()) < (1e-11 if dtype == "float64" else 2e-3)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_cholesky_lookahead_depth_does_not_change_the_factor(K, dtype, monkeypatch):
    """A single matrix is factored with a look-ahead of three block columns in the stream part of the schedule (the
    trailing updates leave the next two block columns to the panel chain, chol.hip ``look``), several matrices with
    one.  Every tile still takes its panels in ascending order, so in float64 the factor and the block inverses are
    the same bits for every depth -- with rounds (the look-ahead columns catch up at the switch) and without (the
    trailing region runs out before the columns do); float32 to rounding."""
    tdt = torch.float64 if dtype == "float64" else torch.float32
    g = torch.Generator(device="cuda").manual_seed(3)
    n = 17000
    U = torch.randn(n, 24, dtype=torch.float64, device="cuda", generator=g)
    S = U @ U.T / 24
    S.diagonal().add_(2.0 + torch.rand(n, dtype=torch.float64, device="cuda", generator=g))
    S = S.to(tdt)

    def run(debug):
        if debug:
            monkeypatch.setenv("SSA_CHOL_DEBUG", debug)
        else:
            monkeypatch.delenv("SSA_CHOL_DEBUG", raising=False)
        npad = K.chol_padded_n(n)
        t = torch.zeros((npad, K.padded_ld(npad, dtype)), dtype=tdt, device="cuda")
        t[:n, :n] = torch.tril(S)
        f = K.chol_factor(t, n)
        torch.cuda.synchronize()
        assert f.info == 0
        return f

    used = 2 * ((K.chol_padded_n(n) + 4095) // 4096) * 4096 * 4096
    for tail in ("", "tail=0"):
        ref = run(",".join(x for x in ("look=1", tail) if x))
        for look in ("", "look=2", "look=3", "look=5"):          # ("" = the default: 3 for a single matrix)
            f = run(",".join(x for x in (look, tail) if x))
            if dtype == "float64":
                assert torch.equal(f.L[:n, :n], ref.L[:n, :n]), (tail, look)
                assert torch.equal(f.aux[:used], ref.aux[:used]), (tail, look)
            else:
                scale = float(ref.L[:n, :n].abs().max())
                assert float((f.L[:n, :n] - ref.L[:n, :n]).abs().max()) < 1e-5 * scale, (tail, look)
    monkeypatch.delenv("SSA_CHOL_DEBUG", raising=False)
    x = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    got = K.chol_solve(run(""), (S.double() @ x).to(tdt)).double()
    assert float((got - x).abs().max() / x.abs().max()) < (1e-11 if dtype == "float64" else 2e-3)


def test_cholesky_full_size_residual(K):
    """BASELINE.json size (n_i = 20 419): S x = b to rounding, by a residual check that needs no
    O(n^3) host work (S = D + U U^T built on the GPU)."""
    n = 20419
    g = torch.Generator(device="cuda").manual_seed(3)
    U = torch.randn(n, 32, dtype=torch.float64, device="cuda", generator=g)
    d = 2.0 + torch.rand(n, dtype=torch.float64, device="cuda", generator=g)
    npad = K.chol_padded_n(n)
    Sd = torch.zeros((npad, K.padded_ld(npad, "float64")), dtype=torch.float64, device="cuda")
    Sd[:n, :n] = U @ U.T / 32
    Sd[:n, :n].diagonal().add_(d)
    Sfull = Sd[:n, :n].clone()
    f = K.chol_factor(Sd, n)
    assert f.info == 0
    b = torch.randn(n, dtype=torch.float64, device="cuda", generator=g)
    x = K.chol_solve(f, b.clone())
    r = (Sfull @ x - b).abs().max().item() / b.abs().max().item()
    assert r < 1e-12
    # linearity of the solve
    x2 = K.chol_solve(f, (2.5 * b).clone())
    assert (x2 - 2.5 * x).abs().max().item() <= 1e-13 * x.abs().max().item()


def test_sheet_field_golden(K, golden):
    """ssa_sheet_field against the reference's _biot_savart_2d_z / _biot_savart_2d_vector."""
    from scipy.constants import mu_0 as MU_0  # sources/current.py:5

    d = golden("sheet_field.npz")
    pref = MU_0 / (4 * np.pi) * 1.0  # uA/um == A/m
    args = (dev(d["sites"]), dev(d["areas"]), dev(d["J"]), float(d["z0"]), dev(d["eval_xyz"]), pref)
    assert relerr(K.sheet_field(*args, True).cpu().numpy(), d["B_tesla"]) < 1e-12
    assert relerr(K.sheet_field(*args, False).cpu().numpy(), d["Bz_tesla"]) < 1e-12


def test_sheet_field_ragged_and_large(K):
    """Ragged sizes (1 point, 257 points, source counts off the 256 tile) and an image-sized case
    against the vectorised oracle; linearity in J."""
    rng = np.random.default_rng(11)
    for ns, npts in ((1, 1), (255, 257), (1000, 3), (2049, 1025)):
        src = rng.uniform(-5, 5, (ns, 2))
        J = rng.standard_normal((ns, 2))
        a = rng.uniform(0.5, 1.5, ns)
        ev = np.column_stack([rng.uniform(-6, 6, npts), rng.uniform(-6, 6, npts), rng.uniform(0.3, 2.0, npts)])
        ref = orc.biot_savart_2d(ev[:, 0], ev[:, 1], ev[:, 2], positions=src, current_densities=J, z0=-0.1,
                                 areas=a, vector=True)
        from scipy.constants import mu_0 as MU_0
        got = K.sheet_field(dev(src), dev(a), dev(J), -0.1, dev(ev), MU_0 / (4 * np.pi), True).cpu().numpy()
        assert relerr(got, ref) < 1e-12
        got2 = K.sheet_field(dev(src), dev(a), dev(2.0 * J), -0.1, dev(ev), MU_0 / (4 * np.pi), True).cpu().numpy()
        assert relerr(got2, 2.0 * got) < 1e-15


@pytest.mark.parametrize("n", [4096, 4500, 9011])
def test_lu_block_solves(K, n):
    """Orders beyond one 4096-row solve block: level-batched inverses of the diagonal blocks of L
    and U (triangular K ranges), partial last block, GEMV chain (single rhs) and GEMM chain."""
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n)) / np.sqrt(n) + np.diag(2.0 + rng.random(n))   # well conditioned, no pivoting
    ld = K.padded_ld(n, "float64")
    Ad = torch.zeros((n, ld), dtype=torch.float64, device="cuda")
    Ad[:, :n] = dev(A)
    f = K.lu_factor(Ad, n)
    assert f.info == 0
    lu_ref, piv_ref = la.lu_factor(A)
    assert np.array_equal(f.ipiv.cpu().numpy(), piv_ref)
    assert relerr(f.lu.cpu().numpy()[:, :n], lu_ref) < 1e-11
    for nrhs in (1, 2, 20, 70):
        b = rng.standard_normal((n, nrhs))
        x = K.lu_solve(f, dev(b[:, 0].copy()) if nrhs == 1 else dev(b)).cpu().numpy().reshape(n, nrhs)
        assert relerr(A @ x, b) < 1e-11
        assert relerr(x, la.lu_solve((lu_ref, piv_ref), b)) < 1e-11


def test_c_abi_error_codes():
    """Argument checking of the C ABI: bad pointers / sizes / dtypes give SSA_ERR_INVALID_ARGUMENT (-1),
    short workspaces SSA_ERR_WORKSPACE_TOO_SMALL (-3); nothing is launched in those cases."""
    import ctypes

    from superscreen_amd import _hip

    lib = _hip.load_library()
    n = 64
    xy = torch.rand(n, 2, dtype=torch.float64, device="cuda")
    w = torch.ones(n, dtype=torch.float64, device="cuda")
    out = torch.zeros(n, dtype=torch.float64, device="cuda")
    ws = torch.zeros(1 << 16, dtype=torch.uint8, device="cuda")
    P = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    st = torch.cuda.current_stream().cuda_stream
    INVALID, TOO_SMALL = -1, -3
    # null pointer, non-positive size, bad dtype
    assert lib.ssa_q_assemble(None, P(w), P(w), n, None, 0, 1, P(out), st) == INVALID
    assert lib.ssa_q_assemble(P(xy), P(w), P(w), 0, None, 0, 1, P(out), st) == INVALID
    assert lib.ssa_q_assemble(P(xy), P(w), P(w), n, None, 0, 7, P(out), st) == INVALID
    assert lib.ssa_gemv(None, n, n, n, P(w), None, None, P(out), 1.0, 0.0, 1, st) == INVALID
    assert lib.ssa_gemm(n, n, n, 1.0, P(xy), 1, P(xy), n, 0.0, P(out), n, 1, st) == INVALID      # lda < K
    # factor / solve entry points
    A = torch.eye(n, dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda")
    ipiv = torch.zeros(n, dtype=torch.int32, device="cuda")
    aux = torch.zeros(lib.ssa_lu_aux_bytes(n, 1), dtype=torch.uint8, device="cuda")
    assert lib.ssa_lu_factor(P(A), n, n - 1, P(ipiv), P(info), P(aux), 1, P(ws), ws.numel(), st) == INVALID  # lda < n
    assert lib.ssa_lu_factor(P(A), n, n, P(ipiv), P(info), P(aux), 1, P(ws), 16, st) == TOO_SMALL
    assert lib.ssa_chol_factor(P(A), n, n, P(info), P(aux), 1, st) == INVALID          # lda < padded order (256)
    assert lib.ssa_chol_factor_batch(0, None, None, None, None, None, 1, st) == INVALID
    b = torch.ones(n, dtype=torch.float64, device="cuda")
    assert lib.ssa_chol_solve(P(A), n, n, P(aux), P(b), 1, 1, 1, P(ws), 8, st) == TOO_SMALL
    assert lib.ssa_lu_solve(P(A), n, n, P(aux), P(b), 0, 1, 1, P(ws), ws.numel(), st) == INVALID   # nrhs = 0
    # pairwise kernels
    assert lib.ssa_biot_savart(P(xy), P(w), P(xy), n, 5, 3, P(xy), n, 0.5, P(out), 0, 1, P(ws), ws.numel(), st) == INVALID
    assert lib.ssa_self_field(P(xy), P(w), P(w), P(w), n, P(out), 1.0, 1, P(ws), 8, st) == TOO_SMALL
    assert lib.ssa_sheet_field(P(xy), P(w), P(xy), n, 0.0, None, n, 1.0, 0, P(out), P(ws), ws.numel(), st) == INVALID
    assert lib.ssa_mfma_probe(0, P(out), None, st) == INVALID
    assert lib.ssa_error_string(INVALID).decode() and lib.ssa_error_string(TOO_SMALL).decode().startswith("workspace")
    torch.cuda.synchronize()
    assert float(out.abs().max()) == 0.0   # nothing ran


def test_tiny_and_degenerate_sizes(K):
    """n = 1 .. 3 systems and single-point pair sums (ragged edge of every tile loop)."""
    for n in (1, 2, 3):
        S = np.diag(2.0 + np.arange(n)) + 0.1 * np.ones((n, n))
        npad = K.chol_padded_n(n)
        Sd = torch.zeros((npad, K.padded_ld(npad, "float64")), dtype=torch.float64, device="cuda")
        Sd[:n, :n] = dev(np.tril(S))
        f = K.chol_factor(Sd, n)
        assert f.info == 0
        b = np.arange(1.0, n + 1)
        assert relerr(K.chol_solve(f, dev(b)).cpu().numpy(), np.linalg.solve(S, b)) < 1e-13
        Ad = torch.zeros((n, K.padded_ld(n, "float64")), dtype=torch.float64, device="cuda")
        Ad[:, :n] = dev(S)
        lf = K.lu_factor(Ad, n)
        assert lf.info == 0 and relerr(K.lu_solve(lf, dev(b)).cpu().numpy(), np.linalg.solve(S, b)) < 1e-13


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-13), ("float32", 2e-5)])
@pytest.mark.parametrize("nvec", [1, 4, 5, 9, 12, 13, 16, 19, 33, 48, 70])
def test_multi_vector_pairwise_kernels(K, disk, dtype, tol, nvec):
    """ssa_self_field_multi / ssa_biot_savart_multi column by column against the single-vector kernels."""
    sites, elements, mesh = disk
    n = len(sites)
    rng = np.random.default_rng(nvec)
    tdt = getattr(torch, dtype)
    xy, w = dev(sites), dev(mesh.weights)
    _, qd = K.q_assemble(xy, w, dev(orc.C_vector(sites)), "float64", want_Q=False)
    g = torch.from_numpy(rng.standard_normal((n, nvec))).to("cuda").to(tdt).contiguous()
    sf = K.self_field_multi(xy, w, qd, g, alpha=0.5).cpu().numpy().astype(np.float64)
    for v in range(nvec):
        ref = K.self_field(xy, w, qd, g[:, v].contiguous(), alpha=0.5).cpu().numpy().astype(np.float64)
        assert relerr(sf[:, v], ref) < tol
    # a list of target rows: only those rows are written, with the values of the full evaluation
    rows = torch.arange(2, n, 5, device="cuda")
    part = torch.full((n, nvec), float("nan"), dtype=tdt, device="cuda")
    K.self_field_multi_rows(xy, w, qd, g, rows, part, alpha=0.5)
    got = part.cpu().numpy().astype(np.float64)
    assert relerr(got[rows.cpu().numpy()], sf[rows.cpu().numpy()]) < tol
    keep = np.ones(n, dtype=bool)
    keep[rows.cpu().numpy()] = False
    assert np.isnan(got[keep]).all()
    # coupling between two different point sets, accumulate on top of an existing field
    tgt = dev(sites[: n // 2] * 0.9 + 0.05)
    J = torch.from_numpy(rng.standard_normal((n, nvec, 2))).to("cuda").contiguous()
    base = torch.from_numpy(rng.standard_normal((n // 2, nvec))).to("cuda").to(tdt).contiguous()
    out = base.clone()
    K.biot_savart_multi(xy, w.to(tdt), J, tgt, 0.7, out, accumulate=True)
    out = out.cpu().numpy().astype(np.float64)
    # a list of target rows: the other rows keep what they held
    trows = torch.arange(1, n // 2, 3, device="cuda")
    sub = base.clone()
    K.biot_savart_multi(xy, w.to(tdt), J, tgt, 0.7, sub, accumulate=True, rows=trows)
    sub = sub.cpu().numpy().astype(np.float64)
    tr = trows.cpu().numpy()
    assert relerr(sub[tr], out[tr]) < tol
    rest = np.ones(n // 2, dtype=bool)
    rest[tr] = False
    assert np.array_equal(sub[rest], base.cpu().numpy().astype(np.float64)[rest])
    for v in range(nvec):
        ref = base[:, v].contiguous().clone()
        K.biot_savart(xy, w.to(tdt), J[:, v, :].contiguous(), tgt, 0.7, ref, accumulate=True)
        assert relerr(out[:, v], ref.cpu().numpy().astype(np.float64)) < tol


def test_coupling_allreduce_c_abi_and_shutdown(K):
    """Section 7 of the C ABI on one GPU: a single-rank RCCL communicator made by ``ssa_rccl_*`` (the only
    communicator a 1-GPU box can hold), ``ssa_coupling_allreduce`` through it (a sum over one rank leaves
    the buffer unchanged, in float64 and float32), the same communicator driving a CouplingPlan inside
    ``solve``; then ``ssa_shutdown`` releases the library's side streams and the next factorization
    re-creates them."""
    import ctypes

    import superscreen_amd as sc
    from superscreen_amd import _hip, synthetic
    from superscreen_amd.parallel import CouplingPlan, RcclCommunicator

    lib = _hip.load_library()
    comm = RcclCommunicator(RcclCommunicator.new_unique_id(), 0, 1)
    for dt in (torch.float64, torch.float32):
        x = torch.randn(121204, dtype=dt, device="cuda")          # the 4 x 30 301 coupling vector of config 5
        y = x.clone()
        comm.all_reduce_sum_(y)
        torch.cuda.synchronize()
        assert torch.equal(x, y)
    assert lib.ssa_coupling_allreduce(None, 4, _hip.SSA_F64, comm.handle, None) == -1
    assert lib.ssa_coupling_allreduce(ctypes.c_void_p(x.data_ptr()), 4, 7, comm.handle, None) == -1
    assert lib.ssa_coupling_allreduce(ctypes.c_void_p(x.data_ptr()), 4, _hip.SSA_F64, None, None) == -1
    device = synthetic.make_stack_device(12, ("washer", "disk", "disk"), solve_dtype="float64")
    model = sc.factorize_model(device=device, current_units="uA", circulating_currents={"hole0": 1.0})
    base = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=2)
    dist_ = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=2, coupling=CouplingPlan(comm=comm))
    for a, b in zip(base, dist_):
        for nm in device.films:
            assert np.array_equal(a.film_solutions[nm].stream, b.film_solutions[nm].stream)
    comm.destroy()
    torch.cuda.synchronize()
    assert lib.ssa_shutdown() == 0
    assert lib.ssa_shutdown() == 0                                  # idempotent
    again = sc.solve(device, applied_field=sc.ConstantField(1.0), circulating_currents={"hole0": 1.0}, iterations=2)
    for a, b in zip(base, again):
        for nm in device.films:
            assert np.array_equal(a.film_solutions[nm].stream, b.film_solutions[nm].stream)


def _nopivot_buffer(K, A, dtype):
    n = len(A)
    npad = K.lu_padded_n(n)
    ld = K.padded_ld(npad, dtype)
    Ad = torch.full((npad, ld), float("nan"), dtype=getattr(torch, dtype), device="cuda")  # padding is set by the library
    Ad[:n, :n] = dev(A.astype(dtype))
    return Ad


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-11), ("float32", 2e-3)])
@pytest.mark.parametrize("n", [3, 64, 257, 500, 700, 1500, 4500, 9011])   # 2 / 3 panels: the two-stream chain's edges
def test_lu_nopivot_matches_lapack_on_dominant_matrices(K, dtype, tol, n):
    """The look-ahead route without interchanges (ssa_lu_factor_nopivot_batch) on row-diagonally-dominant,
    NON-symmetric matrices: LAPACK's getrf returns ipiv == arange for them and the factors must agree."""
    rng = np.random.default_rng(n)
    A = rng.standard_normal((n, n))
    A[np.arange(n), np.arange(n)] = np.abs(A).sum(axis=1) + 1.0          # strictly dominant by rows ...
    A = A * rng.uniform(0.5, 2.0, n)[None, :]                            # ... is not enough for |l| <= 1 by columns:
    A[np.arange(n), np.arange(n)] = np.maximum(np.abs(A).sum(axis=0), np.abs(A).sum(axis=1)) + 1.0
    A = A.astype(dtype)
    lu_ref, piv_ref = la.lu_factor(A)
    assert np.array_equal(piv_ref, np.arange(n))
    f = K.lu_factor_nopivot_batch([(_nopivot_buffer(K, A, dtype), n)])[0]
    assert f is not None and f.info == 0
    assert np.array_equal(f.ipiv.cpu().numpy(), piv_ref)
    assert relerr(f.lu.cpu().numpy()[:n, :n], lu_ref) < (1e-12 if dtype == "float64" else 1e-4)
    for nrhs in (1, 5):
        b = rng.standard_normal((n, nrhs)).astype(dtype)
        x = K.lu_solve(f, dev(b[:, 0]) if nrhs == 1 else dev(b)).cpu().numpy().reshape(n, nrhs)
        assert relerr(x, np.linalg.solve(A.astype(np.float64), b.astype(np.float64))) < tol


def test_lu_nopivot_rejects_matrices_that_need_interchanges_and_batches_are_reproducible(K):
    rng = np.random.default_rng(5)
    n1, n2 = 700, 1300
    general = rng.standard_normal((n1, n1))                              # LAPACK pivots here
    dom = rng.standard_normal((n2, n2)) + 2.0 * n2 * np.eye(n2)
    almost = dom.copy()
    almost[900, 100] = 1.5 * almost[100, 100]                            # ONE multiplier above 1, far below the diagonal block
    out = K.lu_factor_nopivot_batch([(_nopivot_buffer(K, general, "float64"), n1),
                                     (_nopivot_buffer(K, dom, "float64"), n2),
                                     (_nopivot_buffer(K, almost, "float64"), n2)])
    assert out[0] is None and out[2] is None and out[1] is not None
    assert not np.array_equal(la.lu_factor(almost)[1], np.arange(n2))    # LAPACK does swap rows for it
    alone = K.lu_factor_nopivot_batch([(_nopivot_buffer(K, dom, "float64"), n2)])[0]
    assert torch.equal(alone.lu[:n2, :n2], out[1].lu[:n2, :n2])         # a matrix' factor does not depend on its batch
    singular = dom.copy()
    singular[:, 300] = 0.0
    f = K.lu_factor_nopivot_batch([(_nopivot_buffer(K, singular, "float64"), n2)])[0]
    assert f is None or f.info == 301      # LAPACK: U[300, 300] is exactly zero, no interchange, info = 301


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-13), ("float32", 2e-5)])
@pytest.mark.parametrize("shape", [(2048, 128, 256), (96, 256, 512), (4096, 256, 64), (32, 128, 32)])
def test_gemm_nt_small_tile_path(K, dtype, tol, shape):
    """The 32 x 128 tile of the panel chain (few-tile NT launches): against numpy with beta = 1 and beta = 0, and
    the in-place panel product  A21[:, 128:256] = A21 W[128:256, :]^T  that relies on a workgroup reading and
    writing only its own rows."""
    M, N, Kd = shape
    rng = np.random.default_rng(M + N + Kd)
    A = rng.standard_normal((M, Kd)).astype(dtype)
    B = rng.standard_normal((N, Kd)).astype(dtype)
    C = rng.standard_normal((M, N)).astype(dtype)
    for beta in (1.0, 0.0):
        Cd = dev(C) if beta else torch.full((M, N), float("nan"), dtype=getattr(torch, dtype), device="cuda")
        K.gemm_ex(0, 1, False, dev(A), dev(B), Cd, M, N, Kd, alpha=-0.5, beta=beta)
        ref = beta * C.astype(np.float64) - 0.5 * A.astype(np.float64) @ B.astype(np.float64).T
        assert relerr(Cd.cpu().numpy(), ref) < tol
    if Kd == 256 and N == 128:
        A21 = dev(A)                                         # [M, 256]
        W = rng.standard_normal((256, 256)).astype(dtype)
        Wd = dev(W)
        from superscreen_amd import _hip
        lib = _hip.load_library()
        es = A21.element_size()
        _hip.check(lib.ssa_gemm_ex(0, 1, 0, M, 128, 256, 1.0, A21.data_ptr(), 256, Wd.data_ptr() + 128 * 256 * es, 256,
                                   0.0, A21.data_ptr() + 128 * es, 256, _hip.dtype_code(dtype), None), "ssa_gemm_ex")
        torch.cuda.synchronize()
        ref = A.astype(np.float64) @ W[128:, :].astype(np.float64).T
        got = A21.cpu().numpy()
        assert relerr(got[:, 128:], ref) < tol * 10 and np.array_equal(got[:, :128], A[:, :128])


def test_profile_kinds_mask(K):
    """ssa_profile_begin_kinds brackets only the selected kernel kinds (bench.py keeps the event pairs off the panel
    chains inside its timed region)."""
    import ctypes
    from superscreen_amd import _hip
    lib = _hip.load_library()
    rng = np.random.default_rng(0)
    n = 11264   # above the 10 240 trailing columns where the schedule switches to rounds: updates and chain products first
    U = dev(rng.standard_normal((n, 16)))
    S = torch.tril(U @ U.T / 16 + torch.diag(1.5 + dev(rng.random(n))))
    del U

    def factor():
        t = torch.zeros((n, K.padded_ld(n, "float64")), dtype=torch.float64, device="cuda")
        t[:n, :n] = S
        assert K.chol_factor(t, n).info == 0
        torch.cuda.synchronize()

    def counts():
        out = []
        for kind in range(5):
            ms, fl, cnt = ctypes.c_double(0), ctypes.c_double(0), ctypes.c_int64(0)
            _hip.check(lib.ssa_profile_read(kind, ctypes.byref(ms), ctypes.byref(fl), ctypes.byref(cnt)), "read")
            out.append(cnt.value)
        return out

    _hip.check(lib.ssa_profile_begin(), "begin")
    factor()
    everything = counts()
    _hip.check(lib.ssa_profile_end(), "end")
    assert everything[1] > 0 and everything[2] > 0            # trailing updates and chain products
    assert everything[3] > 0 and everything[4] > 0            # round launches and their batched products
    _hip.check(lib.ssa_profile_begin_kinds(0b010), "begin_kinds")
    factor()
    only_syrk = counts()
    _hip.check(lib.ssa_profile_end(), "end")
    assert only_syrk == [0, everything[1], 0, 0, 0]
