"""Pins the CPU oracle (oracle/superscreen_oracle.py) against fixtures recorded from the
reference itself (oracle/make_golden.py).  Runs anywhere (no GPU, no /root/reference)."""
import numpy as np
import pytest
import scipy.sparse as sp
from matplotlib.path import Path

import superscreen_oracle as orc

RTOL = 1e-11  # fp64 both sides; differences = summation order + LAPACK rounding


def csr(d, prefix, shape):
    return sp.csr_array((d[f"{prefix}_data"], d[f"{prefix}_indices"], d[f"{prefix}_indptr"]), shape=shape)


def relerr(a, b):
    return np.max(np.abs(np.asarray(a) - np.asarray(b))) / max(np.max(np.abs(b)), 1e-300)


def contains(poly, pts):
    return Path(poly, closed=True).contains_points(pts)


@pytest.mark.parametrize("name", ["disk_K10.npz", "disk_K26.npz", "washer_K17.npz"])
def test_mesh_operators(golden, name):
    d = golden(name)
    mesh = orc.make_mesh(d["sites"], d["elements"])
    n, m = len(d["sites"]), len(d["elements"])
    assert np.array_equal(mesh.boundary_indices, d["boundary_indices"])
    assert relerr(mesh.triangle_areas, d["triangle_areas"]) < 1e-14
    assert relerr(mesh.weights, d["weights"]) < 1e-14
    assert relerr(orc.C_vector(d["sites"]), d["C"]) < 1e-14
    assert relerr(np.diag(mesh.Q), d["Q_diag"]) < 1e-13
    assert relerr(mesh.Q[d["sample_rows"]], d["Q_rows"]) < 1e-13
    if "Q" in d:
        assert relerr(mesh.Q, d["Q"]) < 1e-13
    for prefix, op, shape in [
        ("lap", mesh.laplacian, (n, n)), ("gx", mesh.gradient_x, (n, n)),
        ("gy", mesh.gradient_y, (n, n)), ("Gx", mesh.gradient_tri_x, (m, n)),
        ("Gy", mesh.gradient_tri_y, (m, n)),
    ]:
        ref = csr(d, prefix, shape)
        diff = abs(op - ref)
        assert diff.max() <= 1e-11 * abs(ref).max(), prefix


@pytest.mark.parametrize("name", ["disk_K10.npz", "disk_K26.npz", "washer_K17.npz"])
def test_film_system_and_solve(golden, name):
    d = golden(name)
    mesh = orc.make_mesh(d["sites"], d["elements"])
    in_film = contains(d["film_poly"], d["sites"])
    holes = {"hole": contains(d["hole_poly"], d["sites"])} if bool(d["washer"]) else {}
    conv = float(d["field_conversion"])
    for li, Lam in enumerate(d["Lambdas"]):
        film = orc.make_film("film", mesh, z0=0.0, Lambda=Lam, in_film=in_film, holes_mask=holes)
        assert np.array_equal(film.film_indices, d["film_indices"])
        if holes:
            assert np.array_equal(film.hole_indices["hole"], d["hole_indices"])
            assert relerr(film.A_holes["hole"][d["sample_rows"]], d[f"A_hole_rows_L{li}"]) < 1e-12
        assert relerr(film.A[d[f"A_rows_idx_L{li}"]], d[f"A_rows_L{li}"]) < 1e-12
        assert relerr(np.diag(film.A), d[f"A_diag_L{li}"]) < 1e-12
        assert np.array_equal(film.lu_piv[1], d[f"piv_L{li}"])
        for ci, circ in enumerate(d["circs"]):
            applied = conv * np.ones(len(d["sites"]))
            sol = orc.solve_film(film, applied, field_conversion=conv,
                                 circulating_currents={"hole": float(circ)})
            tag = f"L{li}_c{ci}"
            assert relerr(sol.stream, d[f"g_{tag}"]) < RTOL
            assert relerr(sol.current_density, d[f"J_{tag}"]) < RTOL
            assert relerr(sol.self_field, d[f"self_field_{tag}"]) < RTOL
            assert relerr(sol.applied_field, d[f"applied_field_{tag}"]) < 1e-15


def test_biot_savart(golden):
    d = golden("biot_savart.npz")
    for tag in ("dz05", "dz0_disjoint", "dzneg"):
        za, zb, shift = d[f"args_{tag}"]
        H = orc.biot_savart_film_to_film(
            film1_sites=d["sites1"], film1_z0=za, film1_areas=d["areas"], film1_J=d["J"],
            film2_sites=d["sites2"] + np.array([shift, 0.0]), film2_z0=zb)
        assert relerr(H, d[f"H_{tag}"]) < 1e-12


def _cpu_kernels():
    """The OpenMP C port of the reference's two numba kernels (oracle/csrc/oracle_kernels.c), built on demand."""
    import shutil

    if shutil.which("gcc") is None:
        pytest.skip("gcc not available: the C port of the oracle kernels cannot be built")
    import build_oracle
    import cpu_kernels

    build_oracle.build(verbose=False)
    return cpu_kernels


def test_c_port_q_matrix_pinned_to_reference(golden):
    """cpu_kernels.q_matrix (distance.py:87-115; what the headline-size oracle and bench.py's cpu_baseline use) against
    the reference's own Q of the 331-vertex disk: Q_ij = -q_ij off the diagonal (device/mesh.py:435-458), q_ii = 0."""
    ck = _cpu_kernels()
    d = golden("disk_K10.npz")
    q = ck.q_matrix(d["sites"])
    off = ~np.eye(len(q), dtype=bool)
    assert not np.diag(q).any()
    assert relerr(-q[off], d["Q"][off]) < 1e-13
    # and the numpy restatement of the same function
    assert relerr(q, orc.q_matrix(d["sites"])) < 1e-13
    # the diagonal the reference derives from it: Q_ii = (C_i + sum_l q_il w_l) / w_i
    w = d["weights"]
    assert relerr((d["C"] + q @ w) / w, d["Q_diag"]) < 1e-13


def test_c_port_biot_savart_pinned_to_reference(golden):
    """cpu_kernels.biot_savart_film_to_film (solver/solve.py:28-73) against the reference's outputs: stacked films,
    disjoint films in one plane (dz = 0) and a source above its target."""
    ck = _cpu_kernels()
    d = golden("biot_savart.npz")
    for tag in ("dz05", "dz0_disjoint", "dzneg"):
        za, zb, shift = d[f"args_{tag}"]
        H = ck.biot_savart_film_to_film(
            film1_sites=d["sites1"], film1_z0=za, film1_areas=d["areas"], film1_J=d["J"],
            film2_sites=d["sites2"] + np.array([shift, 0.0]), film2_z0=zb)
        assert relerr(H, d[f"H_{tag}"]) < 1e-13


def _stack(d):
    import importlib.util, os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_syn", os.path.join(here, "superscreen_amd", "synthetic.py"))
    syn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(syn)
    K = int(d["K"])
    sites, elements, dr = syn.ring_disk_mesh(K)
    mesh = orc.make_mesh(sites, elements)
    Kf = syn.film_rings(K)
    film_poly = syn.circle_points((Kf + 0.5) * dr)
    hole_poly = syn.circle_points((Kf // 3 + 0.5) * dr, 201)
    films = []
    for nm, kind, z in zip(d["names"], d["kinds"], d["z0s"]):
        holes = {f"hole_{nm}": contains(hole_poly, sites)} if kind == "washer" else {}
        films.append(orc.make_film(str(nm), mesh, z0=float(z), Lambda=float(d["Lambda"]),
                                   in_film=contains(film_poly, sites), holes_mask=holes))
    return films, film_poly


@pytest.mark.parametrize("name", ["stack2_K12.npz", "stack3_K8.npz"])
def test_jacobi_trace_and_fluxoid(golden, name):
    d = golden(name)
    films, film_poly = _stack(d)
    circ = {f"hole_{nm}": float(d["circ"]) for nm in d["names"]}
    trace = orc.solve(films, float(d["field_mT"]), iterations=int(d["iterations"]),
                      circulating_currents=circ, field_conversion=float(d["field_conversion"]))
    assert len(trace) == int(d["iterations"]) + 1
    for it, sols in enumerate(trace):
        for nm in d["names"]:
            nm = str(nm)
            assert relerr(sols[nm].stream, d[f"g_{nm}_it{it}"]) < RTOL
            assert relerr(sols[nm].current_density, d[f"J_{nm}_it{it}"]) < RTOL
            assert relerr(sols[nm].self_field, d[f"self_field_{nm}_it{it}"]) < RTOL
            if it > 0:
                assert relerr(sols[nm].field_from_other_films, d[f"other_{nm}_it{it}"]) < RTOL
    nm = str(d["fluxoid_film"])
    film = next(f for f in films if f.name == nm)
    poly = d["fluxoid_poly"]
    flux, int_J = orc.polygon_fluxoid_raw(film, trace[-1][nm], poly,
                                          contains(poly, film.mesh.sites), contains(film_poly, poly))
    assert abs(flux - float(d["flux_part_raw"])) < 1e-10 * abs(float(d["flux_part_raw"]))
    assert abs(int_J - float(d["int_J_raw"])) < 1e-10 * abs(float(d["int_J_raw"]))


def test_mutual_inductance_raw_parts(golden):
    """The raw fluxoid parts behind Device.mutual_inductance_matrix (device/device.py:538-648),
    recorded from the reference for a circulating current in one hole at a time."""
    d = golden("mutual_K12.npz")
    films, film_poly = _stack(d)
    poly = d["fluxoid_poly"]
    hole_names = [str(h) for h in d["hole_names"]]
    film_of = {f"hole_{f.name}": f for f in films}
    for j, src in enumerate(hole_names):
        circ = {h: (float(d["I_circ_uA"]) if h == src else 0.0) for h in hole_names}
        trace = orc.solve(films, 0.0, iterations=int(d["iterations"]), circulating_currents=circ,
                          field_conversion=float(d["field_conversion"]))
        for it, sols in enumerate(trace):
            for i, hole in enumerate(hole_names):
                film = film_of[hole]
                flux, int_J = orc.polygon_fluxoid_raw(film, sols[film.name], poly,
                                                      contains(poly, film.mesh.sites), contains(film_poly, poly))
                assert abs(flux - d["flux_part_raw"][it, i, j]) <= 1e-10 * abs(d["flux_part_raw"][:, :, j]).max()
                assert abs(int_J - d["int_J_raw"][it, i, j]) <= 1e-10 * abs(d["int_J_raw"][:, :, j]).max()


def _synthetic():
    import importlib.util, os
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_syn", os.path.join(here, "superscreen_amd", "synthetic.py"))
    syn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(syn)
    return syn


def _mixed(syn):
    spec = syn.RINGS_MIXED
    geos = {f["name"]: syn.film_geometry(f["kind"], f["K"], film_radius=f["film_radius"], center=f["center"])
            for f in spec["films"]}
    return spec, geos, orc.make_films(spec["layers"], spec["films"], geos)


def test_jacobi_trace_and_fluxoids_films_with_their_own_meshes(golden):
    """Three films on three DIFFERENT meshes (547 / 271 / 169 vertices), the little ring off the axis, Lambda = 0 in
    the lower layer, two films in one layer, a field that is not uniform: every iterate and the fluxoid parts of
    both rings against the reference's (tests/golden/rings_mixed.npz; solver/solve.py:495-515)."""
    d = golden("rings_mixed.npz")
    syn = _synthetic()
    spec, geos, films = _mixed(syn)
    assert [f.name for f in films] == [str(s) for s in d["names"]]
    assert len({len(f.mesh.sites) for f in films}) == 3
    for f in films:
        assert len(f.mesh.sites) == int(d[f"n_{f.name}"])
        assert np.array_equal(f.film_indices, d[f"film_indices_{f.name}"])
    circ = dict(zip((str(h) for h in d["circ_holes"]), d["circ_values"]))
    B0 = float(d["field_mT"])
    trace = orc.solve(films, lambda x, y, z: syn.tilted_field(x, y, z, B0), iterations=int(d["iterations"]),
                      circulating_currents=circ, field_conversion=float(d["field_conversion"]))
    assert len(trace) == int(d["iterations"]) + 1
    for it, sols in enumerate(trace):
        for f in films:
            nm = f.name
            assert relerr(sols[nm].stream, d[f"g_{nm}_it{it}"]) < RTOL
            assert relerr(sols[nm].current_density, d[f"J_{nm}_it{it}"]) < RTOL
            assert relerr(sols[nm].self_field, d[f"self_field_{nm}_it{it}"]) < RTOL
            if it > 0:
                assert relerr(sols[nm].field_from_other_films, d[f"other_{nm}_it{it}"]) < RTOL
            if geos[nm]["hole_polygon"] is not None:
                poly = geos[nm]["fluxoid_polygon"]
                flux, int_J = orc.polygon_fluxoid_raw(f, sols[nm], poly, contains(poly, f.mesh.sites),
                                                      contains(geos[nm]["film_polygon"], poly))
                assert abs(flux - float(d[f"flux_part_raw_{nm}_it{it}"])) <= 1e-10 * abs(float(d[f"flux_part_raw_{nm}_it{it}"]))
                assert abs(int_J - float(d[f"int_J_raw_{nm}_it{it}"])) <= 1e-10 * abs(float(d[f"int_J_raw_{nm}_it{it}"]))


def test_vortices_and_lambda_xy_in_coupled_films_with_their_own_meshes(golden):
    """The branches of solve_film that the single-film fixtures pin one at a time, together inside the Jacobi loop of
    three films on their own meshes (tests/golden/rings_mixed_extras.npz): Lambda(x, y) in the upper layer (the
    grad-Lambda term, solve_film.py:181-185; the fluxoid's Lambda at the polygon's vertices, solution.py:548-551),
    one trapped vortex in the big ring (Lambda = 0) and one in the side disk (solve_film.py:541-554), circulating
    currents in both rings."""
    d = golden("rings_mixed_extras.npz")
    syn = _synthetic()
    spec = syn.RINGS_MIXED
    geos = {f["name"]: syn.film_geometry(f["kind"], f["K"], film_radius=f["film_radius"], center=f["center"])
            for f in spec["films"]}
    lam = {"layer1": syn.lambda_ramp}
    films = orc.make_films(spec["layers"], spec["films"], geos, lambda_funcs=lam)
    vort = {}
    for x, y, film, n in syn.RINGS_MIXED_VORTICES:
        vort.setdefault(film, []).append((x, y, n))
    circ = dict(zip((str(h) for h in d["circ_holes"]), d["circ_values"]))
    B0 = float(d["field_mT"])
    trace = orc.solve(films, lambda x, y, z: syn.tilted_field(x, y, z, B0), iterations=int(d["iterations"]),
                      circulating_currents=circ, field_conversion=float(d["field_conversion"]), vortices=vort)
    layer_of = {f["name"]: f["layer"] for f in spec["films"]}
    for it, sols in enumerate(trace):
        for f in films:
            nm = f.name
            assert relerr(sols[nm].stream, d[f"g_{nm}_it{it}"]) < RTOL
            assert relerr(sols[nm].current_density, d[f"J_{nm}_it{it}"]) < RTOL
            assert relerr(sols[nm].self_field, d[f"self_field_{nm}_it{it}"]) < RTOL
            if it > 0:
                assert relerr(sols[nm].field_from_other_films, d[f"other_{nm}_it{it}"]) < RTOL
            if geos[nm]["hole_polygon"] is not None:
                poly = geos[nm]["fluxoid_polygon"]
                flux, int_J = orc.polygon_fluxoid_raw(f, sols[nm], poly, contains(poly, f.mesh.sites),
                                                      contains(geos[nm]["film_polygon"], poly),
                                                      lambda_func=lam.get(layer_of[nm]))
                assert abs(flux - float(d[f"flux_part_raw_{nm}_it{it}"])) <= 1e-10 * abs(float(d[f"flux_part_raw_{nm}_it{it}"]))
                assert abs(int_J - float(d[f"int_J_raw_{nm}_it{it}"])) <= 1e-10 * abs(float(d[f"int_J_raw_{nm}_it{it}"])) + 1e-300


def test_film_with_terminals_coupled_to_a_ring(golden):
    """tests/golden/strip_ring.npz (recorded from the reference): a strip carrying a transport current between two
    terminals under a ring on its own mesh -- the terminal branch of solve_film (solve_film.py:505-524, 557-562) inside
    the Jacobi loop (solve.py:491-536): every iterate of both films, the ring's fluxoid."""
    d = golden("strip_ring.npz")
    syn = _synthetic()
    spec = syn.STRIP_RING
    geo = syn.strip_ring_geometry(spec)
    strip, ring = geo["strip"], geo["ring"]
    sites = strip["sites"]
    loop = np.asarray(d["boundary_indices"])
    assert np.array_equal(np.sort(loop), np.sort(orc.boundary_vertices(sites, strip["elements"])))
    smesh = orc.make_mesh(sites, strip["elements"])
    film_s = orc.make_film("strip", smesh, z0=0.0, Lambda=spec["strip_Lambda"], in_film=contains(strip["film_polygon"], sites),
                           boundary_indices=loop,
                           terminal_masks={t: contains(p, sites[loop]) for t, p in strip["terminals"].items()})
    rmesh = orc.make_mesh(ring["sites"], ring["elements"])
    film_r = orc.make_film("ring", rmesh, z0=spec["ring_z0"], Lambda=spec["ring_Lambda"],
                           in_film=contains(ring["film_polygon"], ring["sites"]),
                           holes_mask={"hole_ring": contains(ring["hole_polygon"], ring["sites"])})
    cur = float(d["current"])
    trace = orc.solve([film_s, film_r], float(d["field_mT"]), iterations=int(d["iterations"]),
                      circulating_currents={"hole_ring": float(d["circ"])}, field_conversion=float(d["field_conversion"]),
                      terminal_currents={"strip": {"source": cur, "drain": -cur}})
    for it, sols in enumerate(trace):
        for nm in ("strip", "ring"):
            assert relerr(sols[nm].stream, d[f"g_{nm}_it{it}"]) < RTOL
            assert relerr(sols[nm].current_density, d[f"J_{nm}_it{it}"]) < RTOL
            assert relerr(sols[nm].self_field, d[f"self_field_{nm}_it{it}"]) < RTOL
            if it > 0:
                assert relerr(sols[nm].field_from_other_films, d[f"other_{nm}_it{it}"]) < RTOL
        poly = ring["fluxoid_polygon"]
        flux, int_J = orc.polygon_fluxoid_raw(film_r, sols["ring"], poly, contains(poly, ring["sites"]),
                                              contains(ring["film_polygon"], poly))
        assert abs(flux - float(d[f"flux_part_raw_ring_it{it}"])) <= 1e-10 * abs(float(d[f"flux_part_raw_ring_it{it}"]))
        assert abs(int_J - float(d[f"int_J_raw_ring_it{it}"])) <= 1e-10 * abs(float(d[f"int_J_raw_ring_it{it}"]))


def test_mutual_inductance_raw_parts_films_with_their_own_meshes(golden):
    d = golden("mutual_rings_mixed.npz")
    syn = _synthetic()
    spec, geos, films = _mixed(syn)
    hole_names = [str(h) for h in d["hole_names"]]
    film_of = {"hole_" + f.name: f for f in films}
    for j, src in enumerate(hole_names):
        circ = {h: (float(d["I_circ_uA"]) if h == src else 0.0) for h in hole_names}
        trace = orc.solve(films, 0.0, iterations=int(d["iterations"]), circulating_currents=circ,
                          field_conversion=float(d["field_conversion"]))
        for it, sols in enumerate(trace):
            for i, hole in enumerate(hole_names):
                film = film_of[hole]
                poly = geos[film.name]["fluxoid_polygon"]
                flux, int_J = orc.polygon_fluxoid_raw(film, sols[film.name], poly, contains(poly, film.mesh.sites),
                                                      contains(geos[film.name]["film_polygon"], poly))
                assert abs(flux - d["flux_part_raw"][it, i, j]) <= 1e-10 * abs(d["flux_part_raw"][:, :, j]).max()
                assert abs(int_J - d["int_J_raw"][it, i, j]) <= 1e-10 * abs(d["int_J_raw"][:, :, j]).max()


def test_sheet_field(golden):
    """sources/current.py:13-110 (numba kernels of biot_savart_2d) recorded from the reference."""
    d = golden("sheet_field.npz")
    ev = d["eval_xyz"]
    kw = dict(positions=d["sites"], current_densities=d["J"], z0=float(d["z0"]), areas=d["areas"])
    B = orc.biot_savart_2d(ev[:, 0], ev[:, 1], ev[:, 2], vector=True, **kw)
    Bz = orc.biot_savart_2d(ev[:, 0], ev[:, 1], ev[:, 2], vector=False, **kw)
    assert relerr(B, d["B_tesla"]) < RTOL
    assert relerr(Bz, d["Bz_tesla"]) < RTOL


@pytest.mark.parametrize("name", ["vortex_disk_K13.npz", "vortex_washer_K13.npz"])
def test_vortices(golden, name):
    """solve_film with trapped vortices (solver/solve_film.py:541-554) recorded from the reference."""
    import importlib.util, os
    d = golden(name)
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_syn", os.path.join(here, "superscreen_amd", "synthetic.py"))
    syn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(syn)
    K = int(d["K"])
    sites, elements, dr = syn.ring_disk_mesh(K)
    mesh = orc.make_mesh(sites, elements)
    Kf = syn.film_rings(K)
    film_poly = syn.circle_points((Kf + 0.5) * dr)
    holes = {"hole": contains(syn.circle_points((Kf // 3 + 0.5) * dr, 201), sites)} if bool(d["washer"]) else {}
    film = orc.make_film("film", mesh, z0=0.0, Lambda=0.25, in_film=contains(film_poly, sites), holes_mask=holes)
    vortices = [(x, y, n) for (x, y), n in zip(d["vortex_xy"], d["vortex_nPhi0"])]
    conv = float(d["field_conversion"])
    for tag in ("a", "b"):
        sol = orc.solve_film(film, float(d[f"field_mT_{tag}"]) * conv * np.ones(len(sites)), field_conversion=conv,
                             circulating_currents={"hole": float(d[f"circ_{tag}"])}, vortices=vortices,
                             vortex_flux=float(d["vortex_flux"]))
        assert relerr(sol.stream, d[f"g_{tag}"]) < RTOL
        assert relerr(sol.current_density, d[f"J_{tag}"]) < RTOL
        assert relerr(sol.self_field, d[f"self_field_{tag}"]) < RTOL


def test_inhomogeneous_lambda(golden):
    """Lambda(x, y): grad(Lambda) term of solver/solve_film.py:181-185, recorded from the reference."""
    import importlib.util, os
    d = golden("inhomogeneous_washer_K11.npz")
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("_syn", os.path.join(here, "superscreen_amd", "synthetic.py"))
    syn = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(syn)
    K = int(d["K"])
    sites, elements, dr = syn.ring_disk_mesh(K)
    mesh = orc.make_mesh(sites, elements)
    Kf = syn.film_rings(K)
    film_poly = syn.circle_points((Kf + 0.5) * dr)
    holes = {"hole": contains(syn.circle_points((Kf // 3 + 0.5) * dr, 201), sites)}
    film = orc.make_film("film", mesh, z0=0.0, Lambda=d["Lambda"], in_film=contains(film_poly, sites),
                         holes_mask=holes)
    assert relerr(film.A[d["A_rows_idx"]], d["A_rows"]) < RTOL
    assert relerr(np.diag(film.A), d["A_diag"]) < RTOL
    conv = float(d["field_conversion"])
    sol = orc.solve_film(film, float(d["field_mT"]) * conv * np.ones(len(sites)), field_conversion=conv,
                         circulating_currents={"hole": float(d["circ"])})
    assert relerr(sol.stream, d["g"]) < RTOL
    assert relerr(sol.current_density, d["J"]) < RTOL
    assert relerr(sol.self_field, d["self_field"]) < RTOL


@pytest.mark.parametrize("name", ["terminals_strip.npz", "terminals_strip_hole.npz"])
def test_terminal_currents(golden, name):
    """Transport currents (solver/solve_film.py:308-437, 505-524, 557-562) recorded from the reference."""
    import importlib.util, os
    d = golden(name)
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

    def load(mod):
        spec = importlib.util.spec_from_file_location("_" + mod, os.path.join(here, "superscreen_amd", mod + ".py"))
        m = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(m)
        return m

    syn, geo = load("synthetic"), load("geometry")
    nx, ny, length, width = int(d["nx"]), int(d["ny"]), 10.0, 4.0
    sites, elements = syn.strip_mesh(nx, ny, length, width)
    mesh = orc.make_mesh(sites, elements)
    eps = 1e-3 * min(length / nx, width / ny)
    dx = length / nx
    film_poly = geo.box(length + 2 * eps, width + 2 * eps, points=401)
    terms = {"source": geo.box(dx, width + 4 * eps, points=41, center=(-length / 2, 0.0)),
             "drain": geo.box(dx, width + 4 * eps, points=41, center=(length / 2, 0.0))}
    loop = orc.boundary_vertices(sites, elements)
    loop = orc.roll_boundary_outside_terminals(
        loop, [lambda ix, p=p: np.where(contains(p, sites[ix]))[0] for p in terms.values()])
    assert np.array_equal(loop, d["boundary_indices"])
    hr = float(d["hole_radius"])
    holes = {"hole": contains(syn.circle_points(hr, 101), sites)} if hr > 0 else {}
    film = orc.make_film("film", mesh, z0=0.0, Lambda=float(d["Lambda"]), in_film=contains(film_poly, sites),
                         holes_mask=holes, boundary_indices=loop,
                         terminal_masks={t: contains(p, sites[loop]) for t, p in terms.items()})
    conv = float(d["field_conversion"])
    for tag in ("a", "b"):
        cur = float(d[f"current_{tag}"])
        tc = {"source": cur, "drain": -cur}
        assert np.array_equal(film.film_indices, d[f"film_indices_{tag}"])
        assert relerr(orc.terminal_current_stream(film, tc), d[f"g_transport_{tag}"]) < RTOL
        sol = orc.solve_film(film, float(d[f"field_mT_{tag}"]) * conv * np.ones(len(sites)), field_conversion=conv,
                             circulating_currents={"hole": float(d[f"circ_{tag}"])}, terminal_currents=tc)
        assert relerr(sol.stream, d[f"g_{tag}"]) < RTOL
        assert relerr(sol.current_density, d[f"J_{tag}"]) < RTOL
        assert relerr(sol.self_field, d[f"self_field_{tag}"]) < RTOL


def test_london_identity_holds_on_the_reference_outputs(golden):
    """The identity behind ``self_field="london"`` (include/superscreen_hip.h, ssa_london_field_rows),
    checked on what the REFERENCE itself produced: on the rows that are unknowns of the film system its
    ``self_field = Q @ (w g)`` equals ``Laplacian(Lambda g) - H_applied - H_other`` to the residual of its
    LAPACK solve -- single washer with and without a circulating current (the reference's own sparse
    Laplacian from the fixture), and every Jacobi iterate of the coupled two-film stack."""
    d = golden("washer_K17.npz")
    n = len(d["sites"])
    lap = csr(d, "lap", (n, n))
    ix, conv = d["film_indices"], float(d["field_conversion"])
    for L, Lam in enumerate(d["Lambdas"]):
        for c in range(len(d["circs"])):
            g = d[f"g_L{L}_c{c}"]
            sf, Hz = d[f"self_field_L{L}_c{c}"] * conv, d[f"applied_field_L{L}_c{c}"] * conv
            london = (lap @ (float(Lam) * g))[ix] - Hz[ix]
            assert np.abs(sf[ix] - london).max() < 1e-12 * np.abs(sf).max()
    d = golden("stack2_K12.npz")
    films, _ = _stack(d)
    conv = float(d["field_conversion"])
    for it in range(int(d["iterations"]) + 1):
        for film in films:
            nm = film.name
            g, sf = d[f"g_{nm}_it{it}"], d[f"self_field_{nm}_it{it}"] * conv
            Hz = float(d["field_mT"]) * conv * np.ones_like(g)
            if it > 0:
                Hz = Hz + d[f"other_{nm}_it{it}"] * conv
            ix = film.film_indices
            london = (film.mesh.laplacian @ (film.Lambda * g))[ix] - Hz[ix]
            assert np.abs(sf[ix] - london).max() < 1e-12 * np.abs(sf).max()
            # ... and it is NOT an identity on the other rows (holes, boundary, vacuum buffer)
            rest = np.setdiff1d(np.arange(len(g)), ix)
            assert np.abs(sf[rest] - ((film.mesh.laplacian @ (film.Lambda * g))[rest] - Hz[rest])).max() \
                > 1e-3 * np.abs(sf).max()


def test_vector_potential_and_polygon_flux_oracle_vs_reference(golden):
    """Oracle restatements of Solution.vector_potential_at_position / polygon_flux (solution.py:833-934,
    430-482) against the reference's own methods (potential_flux.npz)."""
    from matplotlib.path import Path

    from superscreen_amd import synthetic

    d = golden("potential_flux.npz")
    K = int(d["K"])
    sites, elements, dr = synthetic.ring_disk_mesh(K)
    mesh = orc.make_mesh(sites, elements, build_Q=False)
    Kf = synthetic.film_rings(K)
    masks = {"film": Path(synthetic.circle_points((Kf + 0.5) * dr), closed=True).contains_points(sites),
             "hole": Path(synthetic.circle_points((Kf // 3 + 0.5) * dr, 201), closed=True).contains_points(sites)}
    total = 0.0
    for nm, z0 in zip([str(x) for x in d["names"]], d["z0s"]):
        A = orc.vector_potential(d["eval_xyz"], sites=sites, z0=float(z0), areas=mesh.weights, J=d[f"J_{nm}"])
        assert np.max(np.abs(A - d[f"A_{nm}"])) < 1e-12 * np.max(np.abs(d[f"A_{nm}"]))
        plane = np.column_stack([d["eval_xyz"][:, :2], np.full(len(d["eval_xyz"]), float(d["zs_plane"]))])
        total = total + orc.vector_potential(plane, sites=sites, z0=float(z0), areas=mesh.weights, J=d[f"J_{nm}"])
    assert np.max(np.abs(total - d["A_sum_plane"])) < 1e-12 * np.max(np.abs(d["A_sum_plane"]))
    for poly, film, mask in (("washer0", "washer0", "film"), ("disk1", "disk1", "film"), ("hole0", "washer0", "hole")):
        raw = orc.polygon_flux_raw(d[f"total_field_{film}"], mesh.weights, masks[mask])
        assert abs(raw - float(d[f"flux_{poly}_mT_um2"])) < 1e-12 * abs(raw)
        assert abs(raw * 1e-3 * 1e-12 - float(d[f"flux_{poly}_T_m2"])) < 1e-12 * abs(raw) * 1e-15
