"""End-to-end GPU parity: ``superscreen_amd.solve`` (HIP path through the C ABI) against
(a) fixtures recorded from the reference itself and (b) the CPU oracle on the same inputs.
Tolerance for float64: stream function max-rel-error < 1e-9 (north_star asks < 1e-6)."""
import os

import numpy as np
import pytest

import superscreen_oracle as orc

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

TOL = 1e-9


@pytest.fixture(scope="module")
def sc():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import superscreen_amd

    return superscreen_amd


def relerr(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300)


@pytest.mark.parametrize("name,kind", [("disk_K10.npz", "disk"), ("disk_K26.npz", "disk"),
                                       ("washer_K17.npz", "washer")])
@pytest.mark.parametrize("mode", ["auto", "london", "matrix_free", "dense"])
def test_single_film_vs_reference_fixture(sc, golden, name, kind, mode):
    """Every self-field mode against the REFERENCE's recorded outputs, Lambda in {0, 0.1, 1}: "auto" is what
    ``factorize_model`` does by default (float64, uniform Lambda: the London equation on the unknowns' rows +
    the all-pairs sum on the rest, solver.py), so the headline's default path is pinned to the reference here
    and not to the product's own all-pairs kernel."""
    from superscreen_amd import synthetic

    d = golden(name)
    for li, Lam in enumerate(d["Lambdas"]):
        device = synthetic.make_stack_device(int(d["K"]), (kind,), Lambda=float(Lam), solve_dtype="float64")
        for ci, circ in enumerate(d["circs"]):
            cc = {"hole0": float(circ)} if kind == "washer" else None
            model = sc.factorize_model(device=device, current_units="uA", circulating_currents=cc,
                                       self_field=mode)
            film = f"{kind}0"
            assert np.array_equal(model.film_systems[film].indices, d["film_indices"])
            sols = sc.solve(model=model, applied_field=sc.ConstantField(1.0), field_units="mT",
                            iterations=3, check_inversion=True)
            assert len(sols) == 1  # single film: solve.py:486-489
            fs = sols[0].film_solutions[film]
            tag = f"L{li}_c{ci}"
            assert relerr(fs.stream, d[f"g_{tag}"]) < TOL
            assert relerr(fs.current_density, d[f"J_{tag}"]) < TOL
            assert relerr(fs.self_field, d[f"self_field_{tag}"]) < TOL
            assert relerr(fs.applied_field, d[f"applied_field_{tag}"]) < 1e-15
            assert fs.field_from_other_films is None
            if ci == 0 and mode == "matrix_free":
                sysm = model.film_systems[film]
                rows = d[f"A_rows_idx_L{li}"]
                assert relerr(sysm.A[rows], d[f"A_rows_L{li}"]) < 1e-12
                lu, piv = sysm.lu_piv
                assert np.array_equal(piv, d[f"piv_L{li}"])
                if kind == "washer":
                    Ah = model.hole_systems[film]["hole0"].A
                    assert relerr(Ah[d["sample_rows"]], d[f"A_hole_rows_L{li}"]) < 1e-12


def test_mesh_operators_Q_property(sc, golden):
    from superscreen_amd.mesh import Mesh

    d = golden("disk_K10.npz")
    mesh = Mesh.from_triangulation(d["sites"], d["elements"])
    assert relerr(mesh.operators.Q, d["Q"]) < 1e-13


@pytest.mark.parametrize("name,kinds", [("stack2_K12.npz", ("washer", "disk")),
                                        ("stack3_K8.npz", ("disk", "washer", "disk"))])
def test_coupled_films_jacobi_trace_and_fluxoid(sc, golden, name, kinds):
    from superscreen_amd import synthetic

    d = golden(name)
    K = int(d["K"])
    device = synthetic.make_stack_device(K, kinds, Lambda=float(d["Lambda"]), solve_dtype="float64",
                                         z_spacing=float(d["z0s"][1] - d["z0s"][0]))
    names = [str(s) for s in d["names"]]
    assert list(device.films) == names
    cc = {f"hole{i}": float(d["circ"]) for i, k in enumerate(kinds) if k == "washer"}
    iters = int(d["iterations"])
    sols = sc.solve(device, applied_field=sc.ConstantField(float(d["field_mT"])), field_units="mT",
                    current_units="uA", circulating_currents=cc, iterations=iters)
    assert len(sols) == iters + 1
    for it, sol in enumerate(sols):
        for nm in names:
            fs = sol.film_solutions[nm]
            assert relerr(fs.stream, d[f"g_{nm}_it{it}"]) < TOL
            assert relerr(fs.current_density, d[f"J_{nm}_it{it}"]) < TOL
            assert relerr(fs.self_field, d[f"self_field_{nm}_it{it}"]) < TOL
            if it == 0:
                assert fs.field_from_other_films is None
            else:
                assert relerr(fs.field_from_other_films, d[f"other_{nm}_it{it}"]) < TOL
    # fluxoid of a circle around the washer hole: both parts vs the reference's own arrays
    nm = str(d["fluxoid_film"])
    poly = d["fluxoid_poly"]
    raw_units = "mT * um**2"
    fl = sols[-1].polygon_fluxoid(poly, film=nm, units=raw_units, with_units=False)
    assert abs(fl.flux_part - float(d["flux_part_raw"])) < 1e-9 * abs(float(d["flux_part_raw"]))
    sc_raw = orc.MU_0 * float(d["int_J_raw"]) * 1e-12 / (1e-3 * 1e-12)  # mu_0 [uA um] -> mT um^2
    assert abs(fl.supercurrent_part - sc_raw) < 1e-9 * abs(sc_raw)
    hole = [h for h in device.holes if h.endswith(nm[-1])][0]
    fq = sols[-1].hole_fluxoid(hole, points=poly)  # Phi_0, Quantity
    ref_flux, ref_sc = orc.fluxoid_in_Phi0(float(d["flux_part_raw"]), float(d["int_J_raw"]))
    assert abs(float(fq.flux_part.magnitude) - ref_flux) < 1e-9 * abs(ref_flux)
    assert abs(float(fq.supercurrent_part.magnitude) - ref_sc) < 1e-9 * abs(ref_sc)
    # early stop on tolerance returns a prefix of the same trace
    short = sc.solve(device, applied_field=sc.ConstantField(float(d["field_mT"])), circulating_currents=cc,
                     iterations=iters, tolerance=1e-2)
    assert 2 <= len(short) <= iters + 1
    assert relerr(short[-1].film_solutions[names[0]].stream, d[f"g_{names[0]}_it{len(short) - 1}"]) < TOL


def _mixed_device(sc, spec=None, dtype="float64"):
    from superscreen_amd import synthetic

    spec = spec or synthetic.RINGS_MIXED
    device = synthetic.make_device(spec["films"], spec["layers"], solve_dtype=dtype)
    meshes = list(device.meshes.values())
    assert all(a is not b for i, a in enumerate(meshes) for b in meshes[i + 1:])   # NOT one shared Mesh object
    geos = {f["name"]: synthetic.film_geometry(f["kind"], f["K"], film_radius=f.get("film_radius", 5.0),
                                               center=f.get("center", (0.0, 0.0))) for f in spec["films"]}
    return device, geos


def _mixed_circ(d):
    return dict(zip((str(h) for h in d["circ_holes"]), (float(v) for v in d["circ_values"])))


@pytest.mark.parametrize("method", ["auto", "lu"])
def test_films_with_their_own_meshes_vs_reference_fixture(sc, golden, method):
    """The general case of the reference's coupling loop (solver/solve.py:495-515: ``meshes[source_film].sites``
    -> ``meshes[film].sites``) and the shape of its own multi-film test device (test/test_solve.py:40-93): three
    films on three DIFFERENT meshes (547 / 271 / 169 vertices), the little ring off the axis, Lambda = 0 below and
    0.2 above, a third film in the lower ring's layer (dz = 0), a field that is not uniform.  Every iterate, the
    fluxoid parts of both rings in every iterate, against what the reference produced
    (tests/golden/rings_mixed.npz)."""
    from superscreen_amd import synthetic

    d = golden("rings_mixed.npz")
    device, geos = _mixed_device(sc)
    names = [str(s) for s in d["names"]]
    assert list(device.films) == names
    assert [len(device.meshes[nm].sites) for nm in names] == [int(d[f"n_{nm}"]) for nm in names] == [547, 271, 169]
    iters = int(d["iterations"])
    field = sc.Parameter(synthetic.tilted_field, B0=float(d["field_mT"]))
    model = sc.factorize_model(device=device, current_units="uA", circulating_currents=_mixed_circ(d), method=method)
    for nm in names:
        assert np.array_equal(model.film_systems[nm].indices, d[f"film_indices_{nm}"])
    sols = sc.solve(model=model, applied_field=field, field_units="mT", iterations=iters)
    assert len(sols) == iters + 1
    raw_units = "mT * um**2"
    for it, sol in enumerate(sols):
        for nm in names:
            fs = sol.film_solutions[nm]
            assert fs.stream.shape == (int(d[f"n_{nm}"]),)
            assert relerr(fs.stream, d[f"g_{nm}_it{it}"]) < TOL
            assert relerr(fs.current_density, d[f"J_{nm}_it{it}"]) < TOL
            assert relerr(fs.self_field, d[f"self_field_{nm}_it{it}"]) < TOL
            if it == 0:
                assert fs.field_from_other_films is None
            else:
                assert relerr(fs.field_from_other_films, d[f"other_{nm}_it{it}"]) < TOL
            if geos[nm]["hole_polygon"] is not None:
                fl = sol.polygon_fluxoid(geos[nm]["fluxoid_polygon"], film=nm, units=raw_units, with_units=False)
                ref_flux = float(d[f"flux_part_raw_{nm}_it{it}"])
                ref_sc = orc.MU_0 * float(d[f"int_J_raw_{nm}_it{it}"]) * 1e-12 / (1e-3 * 1e-12)
                assert abs(fl.flux_part - ref_flux) < 1e-9 * abs(ref_flux)
                assert abs(fl.supercurrent_part - ref_sc) <= 1e-9 * abs(ref_sc)
    # the device form of solve() (factorization inside) gives the same iterates
    again = sc.solve(device, applied_field=field, field_units="mT", current_units="uA",
                     circulating_currents=_mixed_circ(d), iterations=iters)
    for a, b in zip(again, sols):
        for nm in names:
            assert relerr(a.film_solutions[nm].stream, b.film_solutions[nm].stream) < (0 if method == "auto" else TOL) + 1e-300


def test_vortices_and_lambda_xy_in_coupled_films_with_their_own_meshes(sc, golden):
    """The branches the single-film fixtures pin one at a time, TOGETHER in the Jacobi loop of three films on their own
    meshes (tests/golden/rings_mixed_extras.npz, recorded from the reference): Lambda(x, y) in the upper layer (that film
    takes the LU route, solve_film.py:181-185), a trapped vortex in the big ring (Lambda = 0) and one in the side disk
    (:541-554: one extra right-hand side each instead of the full inverse), circulating currents in both rings, a
    field that is not uniform.  Every iterate; the fluxoid of the ring with Lambda(x, y) takes Lambda at the polygon's
    vertices (solution.py:548-551)."""
    from superscreen_amd import synthetic

    d = golden("rings_mixed_extras.npz")
    spec = {"films": synthetic.RINGS_MIXED["films"],
            "layers": [dict(l, Lambda=(sc.Parameter(synthetic.lambda_ramp) if l["name"] == "layer1" else l["Lambda"]))
                       for l in synthetic.RINGS_MIXED["layers"]]}
    device, geos = _mixed_device(sc, spec)
    vortices = [sc.Vortex(x=x, y=y, film=film, nPhi0=n) for x, y, film, n in synthetic.RINGS_MIXED_VORTICES]
    names = [str(s) for s in d["names"]]
    iters = int(d["iterations"])
    sols = sc.solve(device, applied_field=sc.Parameter(synthetic.tilted_field, B0=float(d["field_mT"])), field_units="mT",
                    current_units="uA", circulating_currents=_mixed_circ(d), vortices=vortices, iterations=iters)
    assert len(sols) == iters + 1
    for it, sol in enumerate(sols):
        for nm in names:
            fs = sol.film_solutions[nm]
            assert relerr(fs.stream, d[f"g_{nm}_it{it}"]) < TOL
            assert relerr(fs.current_density, d[f"J_{nm}_it{it}"]) < TOL
            assert relerr(fs.self_field, d[f"self_field_{nm}_it{it}"]) < TOL
            if it > 0:
                assert relerr(fs.field_from_other_films, d[f"other_{nm}_it{it}"]) < TOL
            if geos[nm]["hole_polygon"] is not None:
                fl = sol.polygon_fluxoid(geos[nm]["fluxoid_polygon"], film=nm, units="mT * um**2", with_units=False)
                ref_flux = float(d[f"flux_part_raw_{nm}_it{it}"])
                ref_sc = orc.MU_0 * float(d[f"int_J_raw_{nm}_it{it}"]) * 1e-12 / (1e-3 * 1e-12)
                assert abs(fl.flux_part - ref_flux) < 1e-9 * abs(ref_flux)
                assert abs(fl.supercurrent_part - ref_sc) <= 1e-9 * abs(ref_sc)


@pytest.mark.parametrize("method", ["auto", "lu"])
def test_film_with_terminals_coupled_to_a_ring(sc, golden, method):
    """A strip carrying a transport current between two terminals under a ring on its own mesh (a field coil under a
    pickup loop, the shape of the reference's susceptometer notebooks): the terminal branch of solve_film
    (solve_film.py:505-524, 557-562) inside the Jacobi loop, every iterate of both films and the ring's fluxoid against
    what the reference produced (tests/golden/strip_ring.npz)."""
    from superscreen_amd import synthetic

    d = golden("strip_ring.npz")
    device = synthetic.make_strip_ring_device()
    geo = synthetic.strip_ring_geometry()
    assert np.array_equal(device.boundary_vertices("strip"), d["boundary_indices"])
    cur = float(d["current"])
    iters = int(d["iterations"])
    model = sc.factorize_model(device=device, current_units="uA", circulating_currents={"hole_ring": float(d["circ"])},
                               terminal_currents={"strip": {"source": cur, "drain": -cur}}, method=method)
    sols = sc.solve(model=model, applied_field=sc.ConstantField(float(d["field_mT"])), field_units="mT", iterations=iters)
    assert len(sols) == iters + 1
    for it, sol in enumerate(sols):
        for nm in ("strip", "ring"):
            fs = sol.film_solutions[nm]
            assert relerr(fs.stream, d[f"g_{nm}_it{it}"]) < TOL
            assert relerr(fs.current_density, d[f"J_{nm}_it{it}"]) < TOL
            assert relerr(fs.self_field, d[f"self_field_{nm}_it{it}"]) < TOL
            if it > 0:
                assert relerr(fs.field_from_other_films, d[f"other_{nm}_it{it}"]) < TOL
        fl = sol.polygon_fluxoid(geo["ring"]["fluxoid_polygon"], film="ring", units="mT * um**2", with_units=False)
        ref_flux = float(d[f"flux_part_raw_ring_it{it}"])
        ref_sc = orc.MU_0 * float(d[f"int_J_raw_ring_it{it}"]) * 1e-12 / (1e-3 * 1e-12)
        assert abs(fl.flux_part - ref_flux) < 1e-9 * abs(ref_flux)
        assert abs(fl.supercurrent_part - ref_sc) < 1e-9 * abs(ref_sc)
    with pytest.raises(NotImplementedError):
        sc.solve_sweep(model, [0.1, 0.2])


def test_solve_sweep_films_with_their_own_meshes(sc, golden):
    """solve_sweep (n x nvec operands per film, pair kernels between films of different size) on the mixed-mesh
    device: the column that carries the fixture's field reproduces the reference's iterates, every column equals
    the looped solve()."""
    from superscreen_amd import synthetic

    d = golden("rings_mixed.npz")
    device, _ = _mixed_device(sc)
    names = list(device.films)
    iters = int(d["iterations"])
    B0 = float(d["field_mT"])
    model = sc.factorize_model(device=device, current_units="uA", circulating_currents=_mixed_circ(d))
    fields = [sc.Parameter(synthetic.tilted_field, B0=b) for b in (0.3, B0, -1.2)] + [0.5, sc.ConstantField(0.0)]
    fields += [sc.Parameter(synthetic.tilted_field, B0=0.1 * k) for k in range(1, 15)]   # 19 columns
    sweep = sc.solve_sweep(model, fields, field_units="mT", iterations=iters)
    assert len(sweep) == 19 and all(len(s) == iters + 1 for s in sweep)
    for it, sol in enumerate(sweep[1]):
        for nm in names:
            fs = sol.film_solutions[nm]
            assert relerr(fs.stream, d[f"g_{nm}_it{it}"]) < TOL
            assert relerr(fs.current_density, d[f"J_{nm}_it{it}"]) < TOL
            assert relerr(fs.self_field, d[f"self_field_{nm}_it{it}"]) < TOL
            if it > 0:
                assert relerr(fs.field_from_other_films, d[f"other_{nm}_it{it}"]) < TOL
    for f, sols in zip(fields, sweep):
        ref = sc.solve(model=model, applied_field=f if callable(f) else sc.ConstantField(f), field_units="mT",
                       iterations=iters)
        for a, b in zip(sols, ref):
            for nm in names:
                fa, fb = a.film_solutions[nm], b.film_solutions[nm]
                scale = max(float(np.abs(fb.stream).max()), 1e-300)
                assert float(np.abs(fa.stream - fb.stream).max()) / scale < 1e-11
                assert np.array_equal(fa.applied_field, fb.applied_field)
    final = sc.solve_sweep(model, fields[:3], field_units="mT", iterations=iters, all_iterations=False)
    for (last,), sols in zip(final, sweep[:3]):
        for nm in names:
            assert relerr(last.film_solutions[nm].stream, sols[-1].film_solutions[nm].stream) < 1e-10
            assert relerr(last.film_solutions[nm].field_from_other_films,
                          sols[-1].film_solutions[nm].field_from_other_films) < 1e-10


def test_mutual_inductance_matrix_films_with_their_own_meshes(sc, golden):
    """Device.mutual_inductance_matrix (device/device.py:538-648) of the two rings of the mixed-mesh device (different
    meshes and Lambda, the side disk coupling to both) against the reference's raw fluxoid parts."""
    from superscreen_amd.units import MU_0

    d = golden("mutual_rings_mixed.npz")
    device, geos = _mixed_device(sc)
    hole_names = [str(h) for h in d["hole_names"]]
    mapping = {h: geos[h[len("hole_"):]]["fluxoid_polygon"] for h in hole_names}
    iterations = int(d["iterations"])
    I_A = float(d["I_circ_uA"]) * 1e-6
    M_ref = (d["flux_part_raw"] * 1e-3 * 1e-12 + MU_0 * d["int_J_raw"] * 1e-6 * 1e-6) / I_A / 1e-12
    Ms = device.mutual_inductance_matrix(hole_polygon_mapping=mapping, units="pH", all_iterations=True,
                                         iterations=iterations)
    assert len(Ms) == iterations + 1
    for it, M in enumerate(Ms):
        assert np.max(np.abs(np.asarray(M.magnitude) - M_ref[it])) < 1e-9 * np.max(np.abs(M_ref[it]))


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-9), ("float32", 5e-4)])
def test_films_with_their_own_meshes_vs_oracle_medium(sc, dtype, tol):
    """Two rings of 1 951 (K = 25) and 2 977 (K = 31) vertices -- sizes the reference under stubs does not reach
    -- the smaller one off the axis above the larger, different Lambda: every iterate against the oracle."""
    from superscreen_amd import synthetic

    spec = dict(layers=[dict(name="bottom", z0=0.0, Lambda=0.05), dict(name="top", z0=0.7, Lambda=0.3)],
                films=[dict(name="wide", kind="washer", K=31, layer="bottom", film_radius=6.0, center=(0.0, 0.0)),
                       dict(name="narrow", kind="washer", K=25, layer="top", film_radius=4.0, center=(-1.1, 0.6))])
    device, geos = _mixed_device(sc, spec, dtype)
    circ = {"hole_wide": 2.0}
    field = sc.Parameter(synthetic.tilted_field, B0=0.6)
    sols = sc.solve(device, applied_field=field, field_units="mT", current_units="uA", circulating_currents=circ,
                    iterations=4)
    films = orc.make_films(spec["layers"], spec["films"], geos, dtype=dtype)
    assert [len(f.mesh.sites) for f in films] == [2977, 1951]
    trace = orc.solve(films, lambda x, y, z: synthetic.tilted_field(x, y, z, 0.6), iterations=4,
                      circulating_currents=circ)
    for sol, ref in zip(sols, trace):
        for nm in device.films:
            assert relerr(sol.film_solutions[nm].stream, ref[nm].stream) < tol
            assert relerr(sol.film_solutions[nm].self_field, ref[nm].self_field) < tol
            assert relerr(sol.film_solutions[nm].current_density, ref[nm].current_density) < tol * 10
            if ref[nm].field_from_other_films is not None:
                assert relerr(sol.film_solutions[nm].field_from_other_films, ref[nm].field_from_other_films) < tol * 10


def _oracle_stack(K, kinds, z_spacing, Lambda, dtype):
    from matplotlib.path import Path

    from superscreen_amd import synthetic

    sites, elements, dr = synthetic.ring_disk_mesh(K)
    mesh = orc.make_mesh(sites, elements)
    Kf = synthetic.film_rings(K)
    film_poly = synthetic.circle_points((Kf + 0.5) * dr)
    hole_poly = synthetic.circle_points((Kf // 3 + 0.5) * dr, 201)
    films = []
    for i, kind in enumerate(kinds):
        holes = {f"hole{i}": Path(hole_poly, closed=True).contains_points(sites)} if kind == "washer" else {}
        films.append(orc.make_film(f"{kind}{i}", mesh, z0=i * z_spacing, Lambda=Lambda,
                                   in_film=Path(film_poly, closed=True).contains_points(sites),
                                   holes_mask=holes, dtype=dtype))
    return films


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-9), ("float32", 5e-4)])
def test_two_film_vs_oracle_medium(sc, dtype, tol):
    """n = 1951 per film (K = 25): a size the reference-under-stubs cannot reach quickly."""
    from superscreen_amd import synthetic

    K, kinds = 25, ("washer", "disk")
    device = synthetic.make_stack_device(K, kinds, solve_dtype=dtype)
    sols = sc.solve(device, applied_field=sc.ConstantField(0.7), circulating_currents={"hole0": "3 uA"},
                    iterations=4)
    films = _oracle_stack(K, kinds, 0.5, 0.1, dtype)
    trace = orc.solve(films, 0.7, iterations=4, circulating_currents={"hole0": 3.0})
    for sol, ref in zip(sols, trace):
        for nm in device.films:
            assert sol.film_solutions[nm].stream.dtype == np.dtype(dtype)
            assert relerr(sol.film_solutions[nm].stream, ref[nm].stream) < tol
            assert relerr(sol.film_solutions[nm].self_field, ref[nm].self_field) < tol
            assert relerr(sol.film_solutions[nm].current_density, ref[nm].current_density) < tol * 10


def test_mixed_precision_factorization_refined_to_float64(sc):
    """factorize_model(method="mixed"): diag(w) A factored in float32 (half the factorization time of a float64 device,
    half the bytes per triangular solve), every solve refined in float64 against the matrix-free system -- the
    stream functions, sheet currents, fields and fluxoids of a coupled two-film device (films with their own meshes)
    against the oracle's float64 LU at 1e-9, like the float64 routes."""
    from superscreen_amd import synthetic

    spec = dict(layers=[dict(name="bottom", z0=0.0, Lambda=0.05), dict(name="top", z0=0.7, Lambda=0.3)],
                films=[dict(name="wide", kind="washer", K=31, layer="bottom", film_radius=6.0, center=(0.0, 0.0)),
                       dict(name="narrow", kind="washer", K=25, layer="top", film_radius=4.0, center=(-1.1, 0.6))])
    device, geos = _mixed_device(sc, spec)
    circ = {"hole_wide": 2.0}
    field = sc.Parameter(synthetic.tilted_field, B0=0.6)
    model = sc.factorize_model(device=device, current_units="uA", circulating_currents=circ, method="mixed")
    assert all(s.chol is not None and s.chol.dtype == torch.float32 for s in model.film_systems.values())
    sols = sc.solve(model=model, applied_field=field, field_units="mT", iterations=4)
    films = orc.make_films(spec["layers"], spec["films"], geos)
    trace = orc.solve(films, lambda x, y, z: synthetic.tilted_field(x, y, z, 0.6), iterations=4, circulating_currents=circ)
    for sol, ref in zip(sols, trace):
        for f in films:
            nm = f.name
            assert sol.film_solutions[nm].stream.dtype == np.float64
            assert relerr(sol.film_solutions[nm].stream, ref[nm].stream) < TOL
            assert relerr(sol.film_solutions[nm].self_field, ref[nm].self_field) < TOL
            assert relerr(sol.film_solutions[nm].current_density, ref[nm].current_density) < 10 * TOL
            poly = geos[nm]["fluxoid_polygon"]
            got = sol.polygon_fluxoid(poly, film=nm, units="mT * um**2", with_units=False)
            want = orc.polygon_fluxoid_mT_um2(f, ref[nm], poly, geos[nm]["film_polygon"])
            scale = max(abs(want[0]), abs(want[1]))
            assert max(abs(got.flux_part - want[0]), abs(got.supercurrent_part - want[1])) < TOL * scale
    # one sweep fewer leaves the float32 error times ~ 1e-4: the refinement is what gives the digits
    from superscreen_amd import solver as solver_mod

    full = sols[-1].film_solutions["wide"].stream
    old = solver_mod.MIXED_REFINEMENT_SWEEPS
    try:
        solver_mod.MIXED_REFINEMENT_SWEEPS = 0
        raw = sc.solve(model=model, applied_field=field, field_units="mT", iterations=4)[-1].film_solutions["wide"].stream
    finally:
        solver_mod.MIXED_REFINEMENT_SWEEPS = old
    assert 1e-7 < relerr(raw, full) < 5e-3
    with pytest.raises(NotImplementedError):
        sc.solve_sweep(model, [0.1, 0.2])
    with pytest.raises(ValueError):
        sc.factorize_model(device=_mixed_device(sc, spec, "float32")[0], current_units="uA", method="mixed")


def test_cold_solve_uses_small_solve_blocks_and_agrees(sc, golden):
    """``solve(device=...)`` -- the reference's plain cold call -- knows that its factorization serves iterations + 1
    passes and has the triangular solves prepared on 2048-row blocks (``factorize_model(expected_passes=...)``); a
    model made for reuse keeps 4096.  Same Solutions to rounding, and the reference's fixture holds for both."""
    from superscreen_amd import synthetic

    device = synthetic.make_stack_device(34, ("washer", "disk"), solve_dtype="float64")      # 3 571 vertices per film
    cold = sc.solve(device, applied_field=sc.ConstantField(0.7), field_units="mT", current_units="uA", iterations=3)
    reuse = sc.factorize_model(device=device, current_units="uA")
    few = sc.factorize_model(device=device, current_units="uA", expected_passes=4)
    many = sc.factorize_model(device=device, current_units="uA", expected_passes=400)
    assert {s.chol.solve_block for s in reuse.film_systems.values()} == {4096}
    assert {s.chol.solve_block for s in few.film_systems.values()} == {2048}
    assert {s.chol.solve_block for s in many.film_systems.values()} == {4096}
    warm = sc.solve(model=reuse, applied_field=sc.ConstantField(0.7), field_units="mT", iterations=3)
    for a, b in zip(cold, warm):
        for nm in device.films:
            assert relerr(a.film_solutions[nm].stream, b.film_solutions[nm].stream) < 1e-12
            assert relerr(a.film_solutions[nm].self_field, b.film_solutions[nm].self_field) < 1e-11
    # the reference's own iterates through the cold call (films on their own meshes)
    d = golden("rings_mixed.npz")
    mixed, _ = _mixed_device(sc)
    sols = sc.solve(mixed, applied_field=sc.Parameter(synthetic.tilted_field, B0=float(d["field_mT"])), field_units="mT",
                    current_units="uA", circulating_currents=_mixed_circ(d), iterations=int(d["iterations"]))
    for it, sol in enumerate(sols):
        for nm in mixed.films:
            assert relerr(sol.film_solutions[nm].stream, d[f"g_{nm}_it{it}"]) < TOL


def test_translate_in_place_drops_the_cached_geometry(sc):
    """Device.translate(inplace=True) moves the mesh sites; the GPU copies of the geometry cached on the mesh operators
    and the cached point-in-polygon results follow (they are dropped), and a uniform field gives the same answer."""
    device, _ = _mixed_device(sc)
    before = sc.solve(device, applied_field=sc.ConstantField(0.4), iterations=2)[-1]
    assert all(m.operators._device_cache for m in device.meshes.values())
    device.translate(dx=3.0, dy=-1.5, inplace=True)
    assert not any(m.operators._device_cache for m in device.meshes.values())
    after = sc.solve(device, applied_field=sc.ConstantField(0.4), iterations=2)[-1]
    for nm in device.films:
        assert relerr(after.film_solutions[nm].stream, before.film_solutions[nm].stream) < 1e-10


@pytest.mark.parametrize("method", ["lu", "cholesky"])
def test_factorization_methods_agree(sc, method):
    """LU of -A (the reference's algorithm) and Cholesky of diag(w) A give the same solution."""
    from superscreen_amd import synthetic

    device = synthetic.make_stack_device(22, ("washer", "disk"), solve_dtype="float64")
    ref = sc.solve(model=sc.factorize_model(device=device, current_units="uA", method="lu"),
                   applied_field=sc.ConstantField(1.3), iterations=3, check_inversion=True)
    model = sc.factorize_model(device=device, current_units="uA", circulating_currents={"hole0": 2.0},
                               method=method)
    assert (model.film_systems["disk1"].chol is not None) == (method == "cholesky")
    model.set_circulating_currents({})
    got = sc.solve(model=model, applied_field=sc.ConstantField(1.3), iterations=3, check_inversion=True)
    for a, b in zip(ref, got):
        for nm in device.films:
            assert relerr(b.film_solutions[nm].stream, a.film_solutions[nm].stream) < 1e-11
            assert relerr(b.film_solutions[nm].current_density, a.film_solutions[nm].current_density) < 1e-10
    lu, piv = model.film_systems["disk1"].lu_piv  # available on demand for either route
    assert np.array_equal(piv, np.arange(len(piv)))


def test_model_reuse_and_linearity(sc):
    """A factorized model is reusable (solve never mutates it) and the response is linear in the
    applied field -- the size-independent property used at BASELINE scale."""
    from superscreen_amd import synthetic

    device = synthetic.make_stack_device(20, ("washer", "disk"), solve_dtype="float64")
    model = sc.factorize_model(device=device, current_units="uA")
    a = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=2)[-1]
    b = sc.solve(model=model, applied_field=sc.ConstantField(2.5), iterations=2)[-1]
    a2 = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=2)[-1]
    for nm in device.films:
        assert np.array_equal(a.film_solutions[nm].stream, a2.film_solutions[nm].stream)
        assert relerr(b.film_solutions[nm].stream, 2.5 * a.film_solutions[nm].stream) < 1e-12
    model.set_circulating_currents({"hole0": 1.0})
    c = sc.solve(model=model, applied_field=sc.ConstantField(0.0))[0]
    assert np.all(c.film_solutions["washer0"].stream[model.film_info["washer0"].hole_indices["hole0"]] == 1.0)
    with pytest.raises(KeyError):
        model.set_circulating_currents({"nope": 1.0})


def test_coupling_plan_matches_single_stream_path(sc):
    """The distributed coupling plan (source-slice partial sums) reproduces the default path:
    world = 1 through solve(), and an emulated world = 3 by running every rank's tasks in turn."""
    import itertools

    from superscreen_amd import kernels, synthetic
    from superscreen_amd.parallel import CouplingPlan

    device = synthetic.make_stack_device(14, ("washer", "disk", "disk"), solve_dtype="float64")
    model = sc.factorize_model(device=device, current_units="uA")
    base = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=2)
    plan = CouplingPlan(rank=0, world=1)
    dist = sc.solve(model=model, applied_field=sc.ConstantField(1.0), iterations=2, coupling=plan)
    for a, b in zip(base, dist):
        for nm in device.films:
            assert np.array_equal(a.film_solutions[nm].stream, b.film_solutions[nm].stream)
    # emulate 3 ranks on one GPU: partial sums over source slices add up to the full field
    films = list(device.films)
    fd = model.film_data
    J = {nm: torch.from_numpy(base[-1].film_solutions[nm].current_density).cuda() for nm in films}
    full = {nm: torch.zeros(fd[nm].n, dtype=torch.float64, device="cuda") for nm in films}
    for src, tgt in itertools.product(films, repeat=2):
        if src != tgt:
            kernels.biot_savart(fd[src].xy, fd[src].w_t, J[src], fd[tgt].xy,
                                model.film_info[tgt].z0 - model.film_info[src].z0, full[tgt], accumulate=True)
    part = {nm: torch.zeros_like(full[nm]) for nm in films}
    sizes = {nm: fd[nm].n for nm in films}
    for rank in range(3):
        for src, tgt, b, e in CouplingPlan.tasks(films, sizes, rank, 3):
            kernels.biot_savart(fd[src].xy, fd[src].w_t, J[src], fd[tgt].xy,
                                model.film_info[tgt].z0 - model.film_info[src].z0, part[tgt],
                                accumulate=True, src_begin=b, src_end=e)
    for nm in films:
        assert relerr(part[nm].cpu().numpy(), full[nm].cpu().numpy()) < 1e-13


def test_mutual_inductance_matrix_vs_reference(golden):
    """Device.mutual_inductance_matrix (device/device.py:538-648) against the raw fluxoid parts the
    reference produces for one circulating current at a time (tests/golden/mutual_K12.npz)."""
    import superscreen_amd as sc
    from superscreen_amd import synthetic
    from superscreen_amd.units import MU_0

    d = golden("mutual_K12.npz")
    device = synthetic.make_stack_device(int(d["K"]), tuple(str(k) for k in d["kinds"]),
                                         z_spacing=float(d["z0s"][1]), Lambda=float(d["Lambda"]))
    poly = d["fluxoid_poly"]
    mapping = {"hole0": poly, "hole1": poly}
    iterations = int(d["iterations"])
    # reference raw parts: flux [mT um^2], Lambda J dl [uA um]; fluxoid / I_circ -> H -> pH
    I_A = float(d["I_circ_uA"]) * 1e-6
    M_ref = (d["flux_part_raw"] * 1e-3 * 1e-12 + MU_0 * d["int_J_raw"] * 1e-6 * 1e-6) / I_A / 1e-12
    Ms = device.mutual_inductance_matrix(hole_polygon_mapping=mapping, units="pH", all_iterations=True,
                                         iterations=iterations)
    assert len(Ms) == iterations + 1
    for it, M in enumerate(Ms):
        assert np.max(np.abs(np.asarray(M.magnitude) - M_ref[it])) < 1e-9 * np.max(np.abs(M_ref[it]))
    M_last = device.mutual_inductance_matrix(hole_polygon_mapping=mapping, units="pH", iterations=iterations)
    assert np.array_equal(np.asarray(M_last.magnitude), np.asarray(Ms[-1].magnitude))
    # other units: Phi_0 / A
    M2 = device.mutual_inductance_matrix(hole_polygon_mapping=mapping, units="Phi_0 / A", iterations=iterations)
    assert np.allclose(np.asarray(M2.magnitude), np.asarray(M_last.to("Phi_0 / A").magnitude), rtol=1e-12)
    # physics: self inductances positive, coupling symmetric to discretisation accuracy
    m = np.asarray(M_last.magnitude)
    assert m[0, 0] > 0 and m[1, 1] > 0 and abs(m[0, 1] - m[1, 0]) < 0.05 * abs(m[0, 1])


def test_mutual_inductance_default_iterations_same_on_both_paths():
    """Without an ``iterations`` argument the reference forwards ``solve``'s default, 0 coupling
    iterations (device/device.py:593-627, solver/solve.py:303): the all-columns-at-once path and the
    one-solve-per-hole path must return the same (uncoupled) matrix."""
    import superscreen_amd as sc
    from superscreen_amd import synthetic

    device = synthetic.make_stack_device(12, ("washer", "washer"), z_spacing=0.4)
    M_sweep = np.asarray(device.mutual_inductance_matrix().magnitude)
    # an extra keyword that the fast path does not take forces the per-hole loop of solve() calls
    M_loop = np.asarray(device.mutual_inductance_matrix(check_inversion=False).magnitude)
    assert np.max(np.abs(M_sweep - M_loop)) < 1e-10 * np.max(np.abs(M_loop))
    M_coupled = np.asarray(device.mutual_inductance_matrix(iterations=1).magnitude)
    assert np.max(np.abs(M_coupled - M_loop)) > 1e-6 * np.max(np.abs(M_loop))  # coupling does change M
    # a device without holes: empty matrix, as the reference's loop over holes produces
    plain = synthetic.make_stack_device(8, ("disk",))
    assert np.asarray(plain.mutual_inductance_matrix().magnitude).shape == (0, 0)


def test_find_fluxoid_solution():
    """find_fluxoid_solution (fluxoid.py:55-119): the returned solution has the requested fluxoids."""
    import superscreen_amd as sc
    from superscreen_amd import synthetic

    device = synthetic.make_stack_device(12, ("washer", "washer"), z_spacing=0.4)
    model = sc.factorize_model(device=device, current_units="uA")
    polys = sc.make_fluxoid_polygons(device)
    target = {"hole0": 1.0, "hole1": -2.0}
    kw = dict(applied_field=sc.ConstantField(0.05), field_units="mT", iterations=40)
    sol = sc.find_fluxoid_solution(model, fluxoids=target, **kw)
    for name, want in target.items():
        got = float(sum(sol.hole_fluxoid(name, points=polys[name], with_units=False)))
        assert abs(got - want) < 1e-6
    assert all(v == 0 for v in model.circulating_currents.values())  # the model is restored
    # default target: zero fluxoid in every hole (Meissner state of the rings)
    sol0 = sc.find_fluxoid_solution(model, **kw)
    for name in target:
        assert abs(float(sum(sol0.hole_fluxoid(name, with_units=False)))) < 1e-6
    assert sol0.circulating_currents["hole0"] != 0


def test_field_at_position_and_sheet_sources(golden):
    """Solution.field_at_position / screening_field_at_position (solution.py:611-831) and
    sources.biot_savart_2d (sources/current.py:113-199) on a solved stack: the GPU all-pairs sum
    against the oracle's vectorised restatement of the reference kernels; in-plane points against
    the interpolated solution."""
    import superscreen_amd as sc
    from superscreen_amd import sources, synthetic
    from superscreen_amd.units import MU_0

    device = synthetic.make_stack_device(12, ("washer", "disk"), z_spacing=0.5)
    sols = sc.solve(device, applied_field=sc.ConstantField(1.0), field_units="mT", current_units="uA",
                    iterations=3)
    sol = sols[-1]
    # The Solution holds a COPY of the device, and Device.copy drops solve_dtype (reference quirk,
    # device/device.py:232-240), so the field arrays of these methods are float32 like the
    # reference's (dtype = device.solve_dtype, solution.py:653): compare at float32 resolution.
    assert sol.device.solve_dtype == np.float32
    TOL = 3e-7
    rng = np.random.default_rng(2)
    pts = np.column_stack([rng.uniform(-6, 6, 50), rng.uniform(-6, 6, 50)])
    z = 1.7
    # screening field above the device, per film, vector
    per_film = sol.screening_field_at_position(pts, zs=z, vector=True, units="tesla", with_units=False,
                                               return_sum=False)
    total = 0.0
    for name, film in device.films.items():
        layer = device.layers[film.layer]
        mesh = device.meshes[name]
        ref = orc.biot_savart_2d(pts[:, 0], pts[:, 1], z, positions=mesh.sites, areas=mesh.vertex_areas,
                                 current_densities=sol.film_solutions[name].current_density, z0=layer.z0,
                                 vector=True)
        assert relerr(per_film[name], ref) < TOL
        total = total + ref[:, 2]
    # total z field in field_units = applied + screening
    Hz = sol.field_at_position(np.column_stack([pts, np.full(len(pts), z)]), units="mT", with_units=False)
    assert relerr(Hz, 1.0 + total / 1e-3) < TOL
    q = sol.field_at_position(pts, zs=z)            # Quantity in field_units by default
    assert relerr(np.asarray(q.magnitude), Hz) < TOL
    # in the plane of the washer: inside the film the solved field is interpolated
    inside = np.array([[2.5, 0.3], [-3.0, 1.0]])
    got = sol.field_at_position(inside, zs=0.0, units="mT", with_units=False, return_sum=False)
    want = sol.interp_field(inside, film="washer0", dataset="self_field")
    assert relerr(got["washer0"], want) < TOL
    assert set(got) == {"washer0", "disk1", "applied_field"}
    with pytest.raises(ValueError):
        sol.field_at_position(np.zeros((2, 3)), zs=1.0)
    # sources.biot_savart_2d with its own triangulation of the sheet, and the Parameter wrapper
    mesh = device.meshes["disk1"]
    Jd = sol.film_solutions["disk1"].current_density
    B = sources.biot_savart_2d(pts[:5, 0], pts[:5, 1], 2.0, positions=mesh.sites, current_densities=Jd, z0=0.5,
                               areas=mesh.vertex_areas, vector=False)
    f = sources.SheetCurrentField(sheet_positions=mesh.sites, current_densities=Jd, z0=0.5)
    B2 = sources.biot_savart_2d(pts[:5, 0], pts[:5, 1], 2.0, positions=mesh.sites, current_densities=Jd, z0=0.5)
    assert relerr(B2[:, 2], B) < 1e-3   # Delaunay areas differ slightly from the mesh's at the rim
    assert B.shape == (5,) and B2.shape == (5, 3) and callable(f)
    assert MU_0 > 0


@pytest.mark.parametrize("name", ["vortex_disk_K13.npz", "vortex_washer_K13.npz"])
@pytest.mark.parametrize("method", ["auto", "lu"])
def test_vortices_vs_reference(golden, name, method):
    """Trapped vortices (solver/solve_film.py:541-554): one extra right-hand side per vortex instead
    of the full inverse the reference forms; both factorization routes."""
    import superscreen_amd as sc
    from superscreen_amd import synthetic

    d = golden(name)
    device = synthetic.make_stack_device(int(d["K"]), ("washer" if bool(d["washer"]) else "disk",), Lambda=0.25)
    film = list(device.films)[0]
    vortices = [sc.Vortex(x=float(x), y=float(y), film=film, nPhi0=float(n))
                for (x, y), n in zip(d["vortex_xy"], d["vortex_nPhi0"])]
    for tag in ("a", "b"):
        circ = {"hole0": float(d[f"circ_{tag}"])} if bool(d["washer"]) else None
        model = sc.factorize_model(device=device, current_units="uA", circulating_currents=circ,
                                   vortices=vortices, method=method)
        sol = sc.solve(model=model, applied_field=sc.ConstantField(float(d[f"field_mT_{tag}"])),
                       field_units="mT")[-1].film_solutions[film]
        assert relerr(sol.stream, d[f"g_{tag}"]) < 1e-9
        assert relerr(sol.current_density, d[f"J_{tag}"]) < 1e-9
        assert relerr(sol.self_field, d[f"self_field_{tag}"]) < 1e-9
    # set_vortices on an existing model: no re-factorization; removing them restores the Meissner answer
    model.set_vortices([])
    plain = sc.solve(model=model, applied_field=sc.ConstantField(0.7), field_units="mT")[-1].film_solutions[film]
    model.set_vortices(vortices)
    again = sc.solve(model=model, applied_field=sc.ConstantField(0.7), field_units="mT")[-1].film_solutions[film]
    assert relerr(again.stream, d["g_b"]) < 1e-9
    assert relerr(plain.stream, again.stream) > 1e-3
    with pytest.raises(ValueError):
        sc.factorize_model(device=device, current_units="uA", vortices=[sc.Vortex(x=40.0, y=0.0, film=film)])


def test_inhomogeneous_lambda_vs_reference(golden):
    """Layer.Lambda as a Parameter of (x, y): the grad(Lambda) term (solver/solve_film.py:181-185) is
    folded into the sparse part of the assembly kernel; diag(w) A is no longer symmetric, so the
    film goes through the LU route."""
    import superscreen_amd as sc
    from superscreen_amd import synthetic

    d = golden("inhomogeneous_washer_K11.npz")

    def lam(x, y):
        return 0.2 * (1.0 + 0.5 * x / 5.0 + 0.3 * (y / 5.0) ** 2)

    device = synthetic.make_stack_device(int(d["K"]), ("washer",))
    device.layers["layer0"].Lambda = sc.Parameter(lam)
    model = sc.factorize_model(device=device, current_units="uA", circulating_currents={"hole0": float(d["circ"])})
    system = model.film_systems["washer0"]
    assert system.chol is None and system.factors is not None          # LU route
    assert relerr(system.A[d["A_rows_idx"]], d["A_rows"]) < 1e-12
    hs = model.hole_systems["washer0"]["hole0"]
    n = len(device.meshes["washer0"].sites)
    rows = np.unique(np.linspace(0, n - 1, 8).astype(np.int64))
    assert relerr(hs.A[rows], d["A_hole_rows"]) < 1e-12
    sol = sc.solve(model=model, applied_field=sc.ConstantField(float(d["field_mT"])),
                   field_units="mT")[-1].film_solutions["washer0"]
    assert relerr(sol.stream, d["g"]) < 1e-9
    assert relerr(sol.current_density, d["J"]) < 1e-9
    assert relerr(sol.self_field, d["self_field"]) < 1e-9
    with pytest.raises(ValueError):
        sc.factorize_model(device=device, current_units="uA", method="cholesky")


@pytest.mark.parametrize("name", ["terminals_strip.npz", "terminals_strip_hole.npz"])
@pytest.mark.parametrize("method", ["auto", "lu"])
def test_terminal_currents_vs_reference(golden, name, method):
    """Transport currents through terminals (solver/solve_film.py:308-437, 505-524, 557-562)."""
    import superscreen_amd as sc
    from superscreen_amd import synthetic

    d = golden(name)
    device = synthetic.make_strip_device(int(d["nx"]), int(d["ny"]), Lambda=float(d["Lambda"]),
                                         hole_radius=float(d["hole_radius"]))
    assert np.array_equal(device.boundary_vertices("strip"), d["boundary_indices"])
    for tag in ("a", "b"):
        cur = float(d[f"current_{tag}"])
        circ = {"hole": float(d[f"circ_{tag}"])} if float(d["hole_radius"]) > 0 else None
        model = sc.factorize_model(device=device, current_units="uA", circulating_currents=circ,
                                   terminal_currents={"strip": {"source": cur, "drain": -cur}}, method=method)
        assert np.array_equal(model.film_systems["strip"].indices, d[f"film_indices_{tag}"])
        sol = sc.solve(model=model, applied_field=sc.ConstantField(float(d[f"field_mT_{tag}"])),
                       field_units="mT")[-1].film_solutions["strip"]
        assert relerr(sol.stream, d[f"g_{tag}"]) < 1e-9
        assert relerr(sol.current_density, d[f"J_{tag}"]) < 1e-9
        assert relerr(sol.self_field, d[f"self_field_{tag}"]) < 1e-9
    with pytest.raises(ValueError, match="not conserved"):
        sc.factorize_model(device=device, current_units="uA", terminal_currents={"strip": {"source": 1.0, "drain": 0.5}})
    # no terminal current: the transport part vanishes, the terminal film still uses the Biot-Savart self-field
    model0 = sc.factorize_model(device=device, current_units="uA")
    s0 = sc.solve(model=model0, applied_field=sc.ConstantField(0.4), field_units="mT")[-1].film_solutions["strip"]
    assert np.isfinite(s0.stream).all() and np.abs(s0.stream).max() > 0


def test_vector_potential_and_polygon_flux():
    """Solution.vector_potential_at_position (solution.py:833-934) on a solved device against its
    cdist/einsum formula, and polygon_flux (:430-482) against the flux part of polygon_fluxoid (the
    reference's own outputs: test_vector_potential_and_polygon_flux_vs_reference)."""
    import superscreen_amd as sc
    from superscreen_amd import synthetic
    from superscreen_amd.units import MU_0

    device = synthetic.make_stack_device(12, ("washer", "disk"), z_spacing=0.5)
    model = sc.factorize_model(device=device, current_units="uA", circulating_currents={"hole0": 3.0})
    sol = sc.solve(model=model, applied_field=sc.ConstantField(0.0), field_units="mT", iterations=2)[-1]
    rng = np.random.default_rng(4)
    pts = np.column_stack([rng.uniform(-6, 6, 40), rng.uniform(-6, 6, 40), rng.uniform(0.8, 2.0, 40)])
    got = sol.vector_potential_at_position(pts, units="mT * um", with_units=False, return_sum=False)
    for name, film in device.films.items():
        mesh = device.meshes[name]
        z0 = device.layers[film.layer].z0
        J = sol.film_solutions[name].current_density
        rho = np.sqrt(((pts[:, None, :2] - mesh.sites[None, :, :]) ** 2).sum(axis=2) + (pts[:, 2, None] - z0) ** 2)
        Axy = np.einsum("ijk, j -> ik", J[None, :, :] / rho[:, :, None], mesh.vertex_areas)   # uA
        ref = MU_0 / (4 * np.pi) * Axy * 1e-6 / (1e-3 * 1e-6)                                  # T m -> mT um
        assert relerr(got[name][:, :2], ref) < 1e-12 and not got[name][:, 2].any()
    total = sol.vector_potential_at_position(pts[:, :2], zs=1.5)
    assert total.magnitude.shape == (40, 3)
    with pytest.raises(ValueError, match="inside the film"):
        sol.vector_potential_at_position(np.array([[0.5, 0.5]]) * 4.0, zs=0.0)
    # polygon_flux of the hole == flux part of the fluxoid of the hole's own outline
    hole = device.holes["hole0"]
    flux = sol.polygon_flux("hole0", units="Phi_0", with_units=False)
    fl = sol.polygon_fluxoid(hole.points, film="washer0", units="Phi_0", with_units=False)
    assert abs(flux - fl.flux_part) <= 1e-12 * abs(fl.flux_part)
    assert sol.polygon_flux("washer0").units.dims == sc.units.parse_units("mT * um**2").dims
    with pytest.raises(ValueError, match="Unknown polygon"):
        sol.polygon_flux("nope")


def test_vector_potential_and_polygon_flux_vs_reference(golden):
    """``Solution.vector_potential_at_position`` (solution.py:833-934) and ``Solution.polygon_flux``
    (:430-482) against the outputs of the reference's own methods (tests/golden/potential_flux.npz,
    recorded by oracle/make_golden.py::potential_and_flux_fixture)."""
    import superscreen_amd as sc
    from superscreen_amd import synthetic
    from superscreen_amd.solution import FilmSolution, Solution

    d = golden("potential_flux.npz")
    names = [str(x) for x in d["names"]]
    device = synthetic.make_stack_device(int(d["K"]), ("washer", "disk"), z_spacing=float(d["z0s"][1]))
    assert list(device.films) == names
    n = len(device.meshes[names[0]].sites)
    fs = {nm: FilmSolution(stream=np.zeros(n), current_density=d[f"J_{nm}"], applied_field=d[f"total_field_{nm}"],
                           self_field=np.zeros(n)) for nm in names}
    sol = Solution(device=device, film_solutions=fs, applied_field_func=sc.ConstantField(0.0), field_units="mT",
                   current_units="uA")
    got = sol.vector_potential_at_position(d["eval_xyz"], units="mT * um", with_units=False, return_sum=False)
    for nm in names:
        assert got[nm].shape == d[f"A_{nm}"].shape
        assert relerr(got[nm], d[f"A_{nm}"]) < 1e-12
    total = sol.vector_potential_at_position(d["eval_xyz"][:, :2], zs=float(d["zs_plane"]), units="mT * um",
                                             with_units=False)
    assert relerr(total, d["A_sum_plane"]) < 1e-12
    for poly in ("washer0", "disk1", "hole0"):
        a = sol.polygon_flux(poly, with_units=False)
        b = sol.polygon_flux(poly, units="T * m**2", with_units=False)
        assert abs(a - float(d[f"flux_{poly}_mT_um2"])) < 1e-12 * abs(float(d[f"flux_{poly}_mT_um2"]))
        assert abs(b - float(d[f"flux_{poly}_T_m2"])) < 1e-12 * abs(float(d[f"flux_{poly}_T_m2"]))


@pytest.mark.parametrize("which", ["stack", "mixed"])
def test_film_placement_two_ranks(which):
    """parallel.FilmPlacement (owner-computes films, broadcast of the O(n) result vectors): two
    ranks share this GPU, gloo carries the broadcasts, results equal the single-process solve.
    ``mixed``: the three films have their own meshes of different size (and the iterates are also held against
    the reference's, tests/golden/rings_mixed.npz)."""
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    worker = os.path.join(here, "workers", "placement_worker.py")
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), worker] + (["mixed"] if which == "mixed" else [])
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("owner-computes == single process") == 2


def test_film_placement_helper_groups_four_ranks():
    """parallel.FilmPlacement with more ranks than films (BASELINE config 5 on 8 GPUs: groups of an owner and its
    helpers): four ranks share this GPU, two films -- the owner of a film factors and solves it, owner and helper
    each take half of the sources of the film's coupling sums, summed inside the group; equal to the single-process
    solve to 1e-12 (the source slices change the order of the sums), one cross-group collective per pass."""
    import socket
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    worker = os.path.join(here, "workers", "placement_helpers_worker.py")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4",
           "--master-addr", "127.0.0.1", "--master-port", str(port), worker]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("helper groups == single process") == 4


def test_nccl_backend_single_rank():
    """The ``nccl`` (= RCCL) backend itself: one rank on this GPU (RCCL does not admit two ranks on one
    device, so the 2-rank test above stays on gloo).  The worker runs the coupling plan, the C-ABI
    communicator, the placement exchange and the sharded sweep on RCCL."""
    import socket
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    worker = os.path.join(here, "workers", "nccl_worker.py")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(port), worker]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "coupling plan, C-ABI communicator, placement and sharded sweep ok" in out.stdout


@pytest.mark.parametrize("method,dtype,tol", [("auto", "float64", 1e-11), ("lu", "float64", 1e-11),
                                              ("auto", "float32", 2e-3)])
def test_solve_sweep_equals_looped_solve(method, dtype, tol):
    """solve_sweep (all fields of a scan as columns of one multi-RHS solve) == solve() per field."""
    import superscreen_amd as sc
    from superscreen_amd import synthetic

    device = synthetic.make_stack_device(12, ("washer", "disk"), z_spacing=0.6, solve_dtype=dtype)
    model = sc.factorize_model(device=device, current_units="uA", circulating_currents={"hole0": 0.7}, method=method)
    values = [0.0, 0.3, -1.1, 2.0, 0.05] + [0.1 * k for k in range(14)]     # 19 fields: one full chunk + 3
    fields = values[:10] + [sc.ConstantField(v) for v in values[10:]]
    sweep = sc.solve_sweep(model, fields, field_units="mT", iterations=3)
    assert len(sweep) == len(values) and all(len(s) == 4 for s in sweep)
    for v, sols in zip(values, sweep):
        ref = sc.solve(model=model, applied_field=sc.ConstantField(v), field_units="mT", iterations=3)
        for a, b in zip(sols, ref):
            for name in device.films:
                fa, fb = a.film_solutions[name], b.film_solutions[name]
                scale = max(float(np.abs(fb.stream).max()), 1e-300)
                assert float(np.abs(fa.stream - fb.stream).max()) / scale < tol
                assert relerr(fa.current_density, fb.current_density) < tol
                assert relerr(fa.self_field, fb.self_field) < tol
                assert np.array_equal(fa.applied_field, fb.applied_field)
                if fb.field_from_other_films is not None:
                    assert relerr(fa.field_from_other_films, fb.field_from_other_films) < tol
    final = sc.solve_sweep(model, fields, field_units="mT", iterations=3, all_iterations=False)
    assert all(len(s) == 1 for s in final)
    for sols, (last,) in zip(sweep, final):
        for name in device.films:
            fa, fb = last.film_solutions[name], sols[-1].film_solutions[name]
            # (not bitwise: iterates that are not returned get their coupling field on the unknowns' rows
            # only, which may group the source slices differently)
            assert relerr(fa.stream, fb.stream) < 10 * tol and relerr(fa.self_field, fb.self_field) < 10 * tol
            assert relerr(fa.field_from_other_films, fb.field_from_other_films) < 10 * tol
    # one set of circulating currents per column (the model's own are left alone)
    per_column = [{"hole0": 0.5}, {"hole0": "2 uA"}, {}]
    swept = sc.solve_sweep(model, [0.1, 0.1, 0.3], field_units="mT", iterations=2, all_iterations=False,
                           circulating_currents=per_column)
    assert model.circulating_currents == {"hole0": 0.7}
    for (sol,), currents, v in zip(swept, per_column, [0.1, 0.1, 0.3]):
        other = sc.factorize_model(device=device, current_units="uA", circulating_currents=currents, method=method)
        ref = sc.solve(model=other, applied_field=sc.ConstantField(v), field_units="mT", iterations=2)[-1]
        assert sol.circulating_currents == other.circulating_currents
        for name in device.films:
            assert relerr(sol.film_solutions[name].stream, ref.film_solutions[name].stream) < tol
            assert relerr(sol.film_solutions[name].self_field, ref.film_solutions[name].self_field) < tol
    with pytest.raises(ValueError):
        sc.solve_sweep(model, [0.1, 0.2], circulating_currents=[{}])
    with pytest.raises(KeyError):
        sc.solve_sweep(model, [0.1], circulating_currents=[{"nope": 1.0}])
    assert sc.solve_sweep(model, [], iterations=1) == []
    single = synthetic.make_stack_device(10, ("disk",))
    m1 = sc.factorize_model(device=single, current_units="uA")
    assert [len(s) for s in sc.solve_sweep(m1, [1.0, 2.0], iterations=5)] == [1, 1]   # one film: no Jacobi loop


@pytest.mark.gpu
@pytest.mark.parametrize("pre_factorize", [False, True])
def test_circulating_current_value(pre_factorize):
    """The reference's own physics test (``test/test_solve.py:96-183``): a ring carrying a circulating
    current of 1 mA, no applied field -- the current crossing any radial cut of the ring is 1000 uA to
    5 %, from ``Solution.current_through_path`` and from the interpolated sheet current."""
    import superscreen_amd as sc
    from superscreen_amd import synthetic

    K = 40
    device = synthetic.make_stack_device(K, ("washer",), Lambda=0.5)
    Kf = synthetic.film_rings(K)
    dr = 5.0 / (Kf + 0.5)
    r_hole, r_film = (Kf // 3 + 0.5) * dr, 5.0
    cc = {"hole0": "1 mA"}
    if pre_factorize:
        model = sc.factorize_model(device=device, circulating_currents=cc, current_units="uA")
        sols = sc.solve(model=model, applied_field=sc.ConstantField(0), field_units="mT", iterations=1)
    else:
        sols = sc.solve(device=device, applied_field=sc.ConstantField(0), circulating_currents=cc,
                        field_units="mT", current_units="uA", iterations=1)
    assert isinstance(sols, list) and len(sols) == 1
    solution = sols[0]
    xs = np.linspace(r_hole - 0.1, r_film + 0.1, 1001)
    positions = np.stack([xs, np.zeros_like(xs)], axis=1)
    for angle, axis in [(0, 1), (90, 0), (180, 1), (270, 0)]:
        c, s = np.cos(np.radians(angle)), np.sin(np.radians(angle))
        coords = positions @ np.array([[c, -s], [s, c]]).T
        current = solution.current_through_path(coords, film="washer0", units="uA", with_units=False)
        assert np.isclose(abs(current), 1000, rtol=5e-2)
        j = solution.interp_current_density(coords, film="washer0", units="uA / um", with_units=False)
        seg = np.linalg.norm(np.diff(coords, axis=0), axis=1)
        assert np.isclose(abs(np.sum(j[1:, axis] * seg)), 1000, rtol=5e-2)


@pytest.mark.gpu
@pytest.mark.parametrize("hole_radius,field", [(0.0, 0.0), (0.0, 0.002), (0.0, -0.001), (0.8, 0.0)])
def test_transport_current_through_cross_sections(hole_radius, field):
    """Physics check in the manner of the reference's ``test/test_transport.py:203-249``: the current
    crossing any full cross-section of a strip equals the terminal current to 5 %, with an applied
    field of the reference test's size (uT; screening currents integrate to zero across the strip) and with a hole carrying a
    circulating current (it adds to one side of the hole and subtracts from the other)."""
    import superscreen_amd as sc
    from superscreen_amd import synthetic

    device = synthetic.make_strip_device(60, 24, Lambda=0.3, hole_radius=hole_radius)
    circ = {"hole": 1.0} if hole_radius > 0 else None
    sol = sc.solve(device, terminal_currents={"strip": {"source": "2 uA", "drain": "-2 uA"}},
                   circulating_currents=circ, applied_field=sc.ConstantField(field), field_units="mT",
                   current_units="uA")[-1]
    ys = np.linspace(-2.0, 2.0, 401)
    for x0 in (-3.0, 0.0 if hole_radius == 0 else 2.0, 3.5):
        section = np.stack([np.full_like(ys, x0), ys], axis=1)
        current = sol.current_through_path(section, film="strip", units="uA", with_units=False)
        assert np.isclose(abs(current), 2.0, rtol=5e-2)
    if hole_radius > 0:
        # cuts from the centre of the hole to either edge of the strip at x = 0: I/2 -+ the circulating current
        up = np.stack([np.zeros(201), np.linspace(0.0, 2.0, 201)], axis=1)
        down = np.stack([np.zeros(201), np.linspace(-2.0, 0.0, 201)], axis=1)
        i_up = sol.current_through_path(up, film="strip", units="uA", with_units=False)
        i_down = sol.current_through_path(down, film="strip", units="uA", with_units=False)
        assert np.isclose(abs(i_up + i_down), 2.0, rtol=5e-2)
        assert np.isclose(abs(i_up - i_down), 2.0, rtol=8e-2)      # the 1 uA circulating current, twice


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["circle", "box"])
@pytest.mark.parametrize("center", [(0.3, 0.2), (2.0, 1.0), (-4.0, 0.0)])
def test_fluxoid_of_simply_connected_regions(shape, center):
    """The reference's ``test_fluxoid_simply_connected`` (``test/test_solution.py:181-241``): in a film
    with one trapped vortex the fluxoid of a simply connected region is Phi_0 times the number of
    vortices inside it (8 %), zero without one (8 % of its flux part); a region that leaves the film
    raises ValueError.  (Regions are kept off-centre: a circle concentric with the synthetic ring mesh
    takes whole rings of vertices in or out of the flux-part sum, a quadrature artefact of that mesh.)"""
    import superscreen_amd as sc
    from superscreen_amd import synthetic

    device = synthetic.make_stack_device(45, ("disk",), Lambda=10.0)   # lambda = 1 um, d = 0.1 um as in the reference test
    sol = sc.solve(device=device, applied_field=sc.ConstantField(1.0), field_units="mT", current_units="uA",
                   vortices=[sc.Vortex(x=0.0, y=0.0, film="disk0")])[-1]
    if shape == "circle":
        coords = synthetic.circle_points(1.5, 201) + np.asarray(center)
    else:
        t = np.linspace(0.0, 1.0, 101)[:-1]
        w, h = 3.0, 2.0
        box = np.concatenate([np.stack([-w / 2 + w * t, np.full_like(t, -h / 2)], 1),
                              np.stack([np.full_like(t, w / 2), -h / 2 + h * t], 1),
                              np.stack([w / 2 - w * t, np.full_like(t, h / 2)], 1),
                              np.stack([np.full_like(t, -w / 2), h / 2 - h * t], 1)])
        coords = np.concatenate([box, box[:1]]) + np.asarray(center)
    if center == (-4.0, 0.0):
        with pytest.raises(ValueError):
            sol.polygon_fluxoid(coords, film="disk0")
        return
    flux_part, supercurrent_part = sol.polygon_fluxoid(coords, film="disk0", units="Phi_0", with_units=False)
    total = flux_part + supercurrent_part
    if center == (0.3, 0.2):
        assert abs(total - 1.0) < 8e-2
    else:
        assert abs(total) / abs(flux_part) < 8e-2


@pytest.mark.gpu
def test_bz_from_vector_potential():
    """The reference's ``test_bz_from_vector_potential`` (``test/test_solution.py:292-341``): above the
    device, B_z = applied + (dA_y/dx - dA_x/dy) with the mesh's own gradient operators, to 5 % of
    max|B_z| -- ties ``vector_potential_at_position`` (``ssa_sheet_potential``) to ``field_at_position``
    (``ssa_sheet_field``) through physics, not through a shared formula."""
    import superscreen_amd as sc
    from superscreen_amd import synthetic

    device = synthetic.make_stack_device(30, ("disk", "washer"), Lambda=0.5)
    sol = sc.solve(device=device, applied_field=sc.ConstantField(0.0), circulating_currents={"hole1": "1 mA"},
                   field_units="mT", current_units="uA", iterations=5)[-1]
    mesh = device.meshes["disk0"]
    positions, z0 = mesh.sites, 1.5
    gx, gy = mesh.operators.gradient_x, mesh.operators.gradient_y
    Bz = np.asarray(sol.field_at_position(positions, zs=z0, units="mT", with_units=False), dtype=np.float64)
    A = sol.vector_potential_at_position(positions, zs=z0, units="mT * um", with_units=False)
    parts = sol.vector_potential_at_position(positions, zs=z0, units="mT * um", with_units=False, return_sum=False)
    assert set(parts) == set(device.films) and np.allclose(sum(parts.values()), A)
    A = np.asarray(A, dtype=np.float64)
    Bz_from_A = gx @ A[:, 1] - gy @ A[:, 0]
    interior = np.linalg.norm(positions, axis=1) < 0.9 * np.linalg.norm(positions, axis=1).max()
    assert np.all(np.abs(Bz_from_A - Bz)[interior] < 5e-2 * np.abs(Bz).max())
    assert np.abs(Bz).max() > 0


@pytest.mark.gpu
@pytest.mark.parametrize("return_solutions", [False, True])
def test_save_path_and_model_persistence(tmp_path, return_solutions):
    """``solve(save_path=...)`` writes the device once and every iterate under its index
    (``solver/solve.py:474-483, 539-547``), readable with ``Solution.load_solutions``; a saved
    ``FactorizedModel`` comes back as an equivalent model (``test/test_solve.py:122-136``: save, load,
    solve).  Container: superscreen_amd.io (.npz), h5py when it is installed."""
    import superscreen_amd as sc
    from superscreen_amd import io, synthetic

    device = synthetic.make_stack_device(10, ("washer", "disk"), z_spacing=0.7)
    model = sc.factorize_model(device=device, current_units="uA", circulating_currents={"hole0": "2 uA"})
    model_path = tmp_path / "model.npz"
    with io.open_file(model_path, "x") as f:
        model.to_hdf5(f)
    with io.open_file(model_path, "r") as f:
        loaded = sc.FactorizedModel.from_hdf5(f)
    assert isinstance(loaded, sc.FactorizedModel) and loaded.circulating_currents == {"hole0": 2.0}
    assert loaded.device == device and loaded.current_units == "uA"

    save_path = tmp_path / "solutions.npz"
    out = sc.solve(model=loaded, applied_field=sc.ConstantField(0.4), field_units="mT", iterations=3,
                   return_solutions=return_solutions, save_path=save_path)
    ref = sc.solve(model=model, applied_field=sc.ConstantField(0.4), field_units="mT", iterations=3)
    assert (out is None) != return_solutions
    on_disk = sc.Solution.load_solutions(save_path)
    assert len(on_disk) == 4
    for k, (a, b) in enumerate(zip(on_disk, ref)):
        assert a.equals(b) and a.device == device and a.circulating_currents == {"hole0": 2.0}
        for name in device.films:
            fa, fb = a.film_solutions[name], b.film_solutions[name]
            assert np.array_equal(fa.stream, fb.stream) and np.array_equal(fa.current_density, fb.current_density)
            assert (fa.field_from_other_films is None) == (k == 0)
        if return_solutions:
            assert a == out[k]          # same time stamp as the Solution that was returned
    with pytest.raises(FileExistsError):
        sc.solve(model=model, applied_field=sc.ConstantField(0.4), iterations=1, save_path=save_path)


@pytest.mark.gpu
def test_self_field_from_the_london_equation():
    """``self_field="london"`` (the default where it applies): on the rows that are unknowns the self
    field is Laplacian(Lambda g) - H_applied - H_other, elsewhere the all-pairs sum; it must agree with
    the all-pairs evaluation on every row to the residual of the linear solve.  Vortices, Lambda(x, y),
    terminals and float32 keep the all-pairs sum."""
    import superscreen_amd as sc
    from superscreen_amd import kernels, synthetic

    device = synthetic.make_stack_device(20, ("washer", "disk", "washer"), z_spacing=1.0)
    cc = {"hole0": 3.0, "hole2": "-1 uA"}
    sols = {}
    for mode in ("matrix_free", "london", "auto"):
        model = sc.factorize_model(device=device, current_units="uA", circulating_currents=cc, self_field=mode)
        sols[mode] = sc.solve(model=model, applied_field=sc.ConstantField(0.7), field_units="mT", iterations=4)
    for a, b, c in zip(sols["matrix_free"], sols["london"], sols["auto"]):
        for name in device.films:
            fa, fb, fc = (s.film_solutions[name] for s in (a, b, c))
            assert np.array_equal(fa.stream, fb.stream) and np.array_equal(fb.self_field, fc.self_field)
            assert not np.array_equal(fa.self_field, fb.self_field)      # a different evaluation ...
            assert relerr(fb.self_field, fa.self_field) < 1e-11          # ... of the same quantity
    # kernel level: rows variant == full kernel on those rows
    fd = model.film_data["washer0"]
    g = torch.from_numpy(sols["auto"][-1].film_solutions["washer0"].stream).cuda()
    full = kernels.self_field(fd.xy, fd.w, fd.qdiag, g)
    rows = torch.arange(3, fd.n, 7, device="cuda")
    part = torch.full_like(full, float("nan"))
    kernels.self_field_rows(fd.xy, fd.w, fd.qdiag, g, rows, part)
    assert relerr(part[rows].cpu().numpy(), full[rows].cpu().numpy()) < 1e-13
    mask = torch.ones(fd.n, dtype=torch.bool, device="cuda")
    mask[rows] = False
    assert torch.isnan(part[mask]).all()
    # where the identity does not hold as written the all-pairs sum stays
    with_vortex = sc.solve(device=device, applied_field=sc.ConstantField(0.1), current_units="uA",
                           vortices=[sc.Vortex(x=0.0, y=3.0, film="disk1")], iterations=1)[-1]
    ref = sc.solve(model=sc.factorize_model(device=device, current_units="uA", self_field="matrix_free",
                                            vortices=[sc.Vortex(x=0.0, y=3.0, film="disk1")]),
                   applied_field=sc.ConstantField(0.1), iterations=1)[-1]
    assert np.array_equal(with_vortex.film_solutions["disk1"].self_field, ref.film_solutions["disk1"].self_field)
    with pytest.raises(ValueError):
        sc.factorize_model(device=device, current_units="uA", self_field="nope")


@pytest.mark.gpu
def test_hole_without_mesh_vertices_is_tolerated():
    """A hole polygon that contains no mesh vertex gives an empty hole system (the reference tolerates empty index
    sets, solver/solve_film.py:209-218, 498-503): factorization and solve go through and equal the hole-free film."""
    import superscreen_amd as sc
    from superscreen_amd import synthetic
    from superscreen_amd.device import Device, Layer, Polygon

    base = synthetic.make_stack_device(12, ("disk",), solve_dtype="float64")
    mesh = base.meshes["disk0"]
    sites = mesh.sites
    # a tiny square between the centre vertex and the first ring: no vertex inside
    c = 0.5 * (sites[0] + sites[1]) + np.array([0.0, 1e-3])
    tiny = c + 1e-4 * np.array([[-1, -1], [1, -1], [1, 1], [-1, 1], [-1, -1]])
    film = base.films["disk0"]
    device = Device("tiny_hole", layers=[Layer("layer0", Lambda=0.1, z0=0.0)],
                    films=[Polygon("disk0", layer="layer0", points=film.points)],
                    holes=[Polygon("pinhole", layer="layer0", points=tiny)], length_units="um", solve_dtype="float64")
    device.meshes = {"disk0": mesh}
    model = sc.factorize_model(device=device, current_units="uA", circulating_currents={"pinhole": 1.0})
    assert len(model.hole_systems["disk0"]["pinhole"].indices) == 0
    assert model.hole_systems["disk0"]["pinhole"].A.shape == (len(sites), 0)
    got = sc.solve(model=model, applied_field=sc.ConstantField(1.0))[0].film_solutions["disk0"]
    ref = sc.solve(base, applied_field=sc.ConstantField(1.0))[0].film_solutions["disk0"]
    assert np.array_equal(got.stream, ref.stream)


@pytest.mark.gpu
def test_lu_route_falls_back_to_partial_pivoting(monkeypatch):
    """factorize_linear_systems tries the LU without interchanges first and, for a film whose result fails the
    LAPACK pivot check, assembles the matrix again and factors it with partial pivoting (one stream per film).
    Here the check is made to fail for every film: the answer must not change."""
    import superscreen_amd as sc
    from superscreen_amd import kernels, synthetic

    device = synthetic.make_stack_device(16, ("washer", "disk"), solve_dtype="float64")
    kw = dict(applied_field=sc.ConstantField(0.6), iterations=2)
    fast = sc.solve(model=sc.factorize_model(device=device, current_units="uA", method="lu"), **kw)
    calls = []
    real = kernels.lu_factor_nopivot_batch

    def rejecting(systems):
        calls.append(len(systems))
        real(systems)                       # runs (and clobbers the buffers), but its verdict is overruled
        return [None] * len(systems)

    monkeypatch.setattr(kernels, "lu_factor_nopivot_batch", rejecting)
    model = sc.factorize_model(device=device, current_units="uA", method="lu")
    assert calls == [2]
    slow = sc.solve(model=model, **kw)
    for a, b in zip(fast, slow):
        for nm in device.films:
            assert relerr(b.film_solutions[nm].stream, a.film_solutions[nm].stream) < 1e-12
    lu, piv = model.film_systems["disk1"].lu_piv
    assert np.array_equal(piv, np.arange(len(piv))) and lu.shape == (len(piv), len(piv))


@pytest.mark.gpu
@pytest.mark.parametrize("films", [("washer", "disk"), ("disk", "washer", "disk")])
def test_output_only_rows_batched_equal_per_pass_route(tmp_path, films):
    """What only goes into the returned Solutions - the self field and the coupling field on the rows that are not
    unknowns - is evaluated for all iterates at once after the last pass (``solver._enqueue_exterior_*``); with
    ``save_path`` the iterates are written as they come and every pass evaluates all rows itself.  Both routes give
    the same Solutions (the stream functions bit for bit: the iteration itself does not change)."""
    import superscreen_amd as sc
    from superscreen_amd import synthetic

    device = synthetic.make_stack_device(24, films, z_spacing=0.6)
    currents = {f"hole{films.index('washer')}": "1.5 uA"}
    model = sc.factorize_model(device=device, current_units="uA", circulating_currents=currents)
    batched = sc.solve(model=model, applied_field=sc.ConstantField(0.7), field_units="mT", iterations=4)
    per_pass = sc.solve(model=model, applied_field=sc.ConstantField(0.7), field_units="mT", iterations=4,
                        save_path=tmp_path / "iterates.npz")
    assert len(batched) == len(per_pass) == 5
    for k, (a, b) in enumerate(zip(batched, per_pass)):
        for name in device.films:
            fa, fb = a.film_solutions[name], b.film_solutions[name]
            assert np.array_equal(fa.stream, fb.stream) and np.array_equal(fa.current_density, fb.current_density)
            scale = np.abs(fb.self_field).max()
            assert np.abs(fa.self_field - fb.self_field).max() < 1e-12 * scale
            if k == 0:
                assert fa.field_from_other_films is None and fb.field_from_other_films is None
            else:
                scale = np.abs(fb.field_from_other_films).max()
                assert np.abs(fa.field_from_other_films - fb.field_from_other_films).max() < 1e-12 * scale
                assert np.count_nonzero(fa.field_from_other_films) == len(fa.field_from_other_films)   # every row filled in


@pytest.mark.gpu
def test_chain_streams_are_calibrated():
    """The panel chains of the factorization schedules run on the streams the library measured as cheap to launch on
    beside the stream of the trailing updates (chain_streams.hip): after a factorization the costs are known, the
    streams in use come first and are within the limit that separates the two kinds of stream."""
    import torch
    from superscreen_amd import kernels

    n = 1500
    rng = np.random.default_rng(5)
    M = rng.standard_normal((n, n))
    S_host = M @ M.T + n * np.eye(n)
    npad = kernels.chol_padded_n(n)
    S = torch.zeros((npad, npad), dtype=torch.float64, device="cuda")
    S[:n, :n] = torch.from_numpy(S_host).cuda()
    f = kernels.chol_factor(S, n)
    torch.cuda.synchronize()
    assert int(f.info_device.item()) == 0
    L = torch.tril(f.L[:n, :n]).cpu().numpy()
    assert np.abs(L @ L.T - S_host).max() < 1e-9 * np.abs(S_host).max()
    costs, groups = kernels.chol_chain_stream_costs()
    assert len(costs) == len(groups) == 16
    limit = 2.0 * max(min(costs), 0.0) + 12.0
    for c, g in zip(costs, groups):
        assert (g == 0) == (c > limit), (costs, groups)     # group 0 = the streams that are slow beside the caller's
    n_good = sum(g > 0 for g in groups)
    pipes = max(groups)
    assert n_good >= 2 and 1 <= pipes <= 8, (costs, groups)   # (12 of 16 with 4 hardware queues per priority, 3 with 8)
    assert all(g > 0 for g in groups[:n_good]), (costs, groups)      # the good ones first ...
    assert sorted(groups[:pipes]) == list(range(1, pipes + 1)), (costs, groups)   # ... one of every pipe to begin with,
    assert costs[:pipes] == sorted(costs[:pipes]), (costs, groups)                # the cheapest pipe first
    assert costs[0] == min(costs), (costs, groups)
    # the measurements can be dropped (a caller that creates / destroys streams between factorizations): nothing is
    # known until the next factorization, which measures again -- with four matrices in one schedule this time
    from superscreen_amd import _hip
    _hip.check(_hip.load_library().ssa_chol_chain_streams_invalidate(), "ssa_chol_chain_streams_invalidate")
    assert kernels.chol_chain_stream_costs() == ([], [])
    four = []
    for k in range(4):
        t = torch.zeros((npad, npad), dtype=torch.float64, device="cuda")
        t[:n, :n] = torch.from_numpy(S_host + k * np.eye(n)).cuda()
        four.append((t, n))
    fs = kernels.chol_factor_batch(four)
    torch.cuda.synchronize()
    assert all(int(x.info_device.item()) == 0 for x in fs)
    for k, x in enumerate(fs):
        Lk = torch.tril(x.L[:n, :n]).cpu().numpy()
        assert np.abs(Lk @ Lk.T - (S_host + k * np.eye(n))).max() < 1e-9 * np.abs(S_host).max()
    costs2, groups2 = kernels.chol_chain_stream_costs()
    assert len(costs2) == 16 and sum(g > 0 for g in groups2) >= 2, (costs2, groups2)


@pytest.mark.gpu
def test_lu_factor_batch_keeps_cooperative_panels_coresident():
    """The pivoting route's exact sub-panel kernel is cooperative (ceil(n / 256) workgroups of one CU each that
    spin-wait on one another).  Three GENERAL matrices (every sub-panel needs interchanges, so the cooperative
    kernel really runs) whose panel kernels together need more workgroups than the chip has CUs: factored side by
    side they could all be resident only in part and time out (info = -1); ``lu_factor_batch`` runs them group
    after group instead.  Checked by the residual of a solve with each factor."""
    from superscreen_amd import kernels

    cus = kernels.device_info()[0]
    n = 256 * (cus // 3 + 2)                      # 3 x ceil(n / 256) > CUs, 2 x ceil(n / 256) <= CUs
    orders = [n, n - 77, n - 300]
    groups = kernels.lu_concurrency_groups(orders, cus)
    assert groups == [[0, 1], [2]]
    gen = torch.Generator(device="cuda").manual_seed(11)
    mats, copies = [], []
    for m in orders:
        ld = kernels.padded_ld(m, "float64")
        A = torch.zeros((m, ld), dtype=torch.float64, device="cuda")
        A[:, :m] = torch.randn((m, m), dtype=torch.float64, device="cuda", generator=gen)
        copies.append(A.clone())
        mats.append(A)
    factors = kernels.lu_factor_batch([(A, m) for A, m in zip(mats, orders)])
    for f, A0, m in zip(factors, copies, orders):
        assert f.info == 0
        piv = f.ipiv.cpu().numpy()
        assert (piv != np.arange(m)).sum() > m // 2        # interchanges nearly everywhere
        b = torch.randn(m, dtype=torch.float64, device="cuda", generator=gen)
        x = kernels.lu_solve(f, b.clone())
        r = kernels.gemv(A0, m, m, x) - b
        assert float(r.abs().max() / b.abs().max()) < 1e-7
    del mats, copies, factors
    torch.cuda.empty_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("which", ["stack", "mixed"])
def test_sweep_grid_two_ranks_on_one_gpu(which):
    """parallel.SweepGrid (film owner x field shard, BASELINE config 4): two ranks share this GPU as the two film
    owners of one shard; each factors ONE film and exchanges the [n, nvec] result arrays with one all-reduce per
    pass (gloo here); the scan equals the single-process ``solve_sweep`` to 1e-12.  ``mixed``: the two films have
    their own meshes of different size."""
    import socket
    import subprocess
    import sys

    here = os.path.dirname(os.path.abspath(__file__))
    worker = os.path.join(here, "workers", "sweep_grid_worker.py")
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), worker] + (["mixed"] if which == "mixed" else [])
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.count("sweep grid == single process") == 2
