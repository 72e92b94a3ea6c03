"""world_size-2 gloo tests (CPU) of the multi-rank host logic: column sharding, unique-id broadcast and the
max-over-ranks timing reduction bench.py uses; plus a two-shard oracle run showing that a column-sharded grid
reproduces the unsharded answer (columns are independent: no data-path collective is needed)."""
import os
import socket

import numpy as np
import pytest

import problems as P
from mpp_amd import constants as C
from mpp_amd import parallel as PAR


def test_shard_columns_partition():
    for ncols, world in ((1048576, 8), (10, 3), (7, 8), (65536, 4)):
        spans = [PAR.shard_columns(ncols, world, r) for r in range(world)]
        assert spans[0][0] == 0
        for (a0, an), (b0, _bn) in zip(spans, spans[1:]):
            assert a0 + an == b0
        assert spans[-1][0] + spans[-1][1] == ncols
        sizes = [n for _, n in spans]
        assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        PAR.shard_columns(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        uid = PAR.broadcast_unique_id(dist, rank, lambda: bytes(range(128)))
        slowest = PAR.max_over_ranks(dist, 1.0 + rank)
        total = PAR.sum_over_ranks(dist, 10.0 * (rank + 1))
        # each rank solves its own shard of a 5-column celia-extruded grid with the CPU oracle
        ncols = 5
        mult = np.exp(0.5 * np.random.default_rng(12345).standard_normal(ncols))
        c0, nloc = PAR.shard_columns(ncols, world, rank)
        prob = P.celia1990(ncols=nloc, nz=20, perm_mult=mult[c0:c0 + nloc])
        o = P.to_oracle(prob, with_blocks=False)
        o.set_solver(C.KSP_GMRES, C.PC_ILU0, snes_rtol=1e-13, snes_stol=1e-15)  # converge far below the comparison
        o.step_dt(prob.dt, 1)
        pr = o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
        q.put((rank, uid, slowest, total, c0, nloc, pr))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_host_logic():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == bytes(range(128))
    assert res[0][2] == res[1][2] == 2.0
    assert res[0][3] == res[1][3] == 30.0
    # the sharded answer equals the unsharded one (every column converges on its own: independent solves here)
    ncols = 5
    mult = np.exp(0.5 * np.random.default_rng(12345).standard_normal(ncols))
    for r in res:
        _rank, _uid, _s, _t, c0, nloc, pr = r
        for j in range(nloc):
            one = P.celia1990(ncols=1, nz=20, perm_mult=mult[c0 + j:c0 + j + 1])
            oo = P.to_oracle(one, with_blocks=False)
            oo.set_solver(C.KSP_GMRES, C.PC_ILU0, snes_rtol=1e-13, snes_stol=1e-15)
            oo.step_dt(one.dt, 1)
            np.testing.assert_allclose(pr[j * 20:(j + 1) * 20], oo.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE), rtol=1e-8)


def _halo_worker(rank, world, port, q):
    """one rank of a column-partitioned 7-point mesh: its block built directly (workloads.vchannel_rank), the halo plan executed with
    gloo point-to-point messages exactly as mppgpu_set_halo describes it (send lists of owned cells per neighbour, ghosts received in
    the order of the receive counts), then the owned rows of y = J x from the refreshed local vector"""
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        nx, ny, nz = 5, 2 * world, 4
        part, lp = P.vchannel_rank(nx, ny, nz, world, rank)
        n_glob = nx * ny * nz
        xg = np.random.default_rng(99).standard_normal(n_glob)           # the same global vector on every rank
        x = np.zeros(part.l2g.size)
        x[:part.nowned] = xg[part.l2g[:part.nowned]]                      # owned values; ghosts not yet known
        reqs, bufs, off = [], [], part.nowned
        for k, peer in enumerate(part.peer):
            send = torch.from_numpy(x[part.send_cells[part.send_ptr[k]:part.send_ptr[k + 1]] - 1].copy())
            recv = torch.empty(int(part.recv_count[k]), dtype=torch.float64)
            reqs += [dist.isend(send, int(peer)), dist.irecv(recv, int(peer))]
            bufs.append((off, recv))
            off += int(part.recv_count[k])
        for r in reqs:
            r.wait()
        for o, recv in bufs:
            x[o:o + recv.numel()] = recv.numpy()
        # the rank's Jacobian rows from the oracle on its LOCAL problem state would need ghost support in the oracle; the global
        # oracle's rows restricted to the owned cells serve: columns mapped through l2g
        glob = P.vchannel(nx, ny, nz)
        o = P.to_oracle(glob)
        o.set_dtime(glob.dt); o.pre_solve()
        rp, cj, val = o.jacobian(glob.press_ic - np.random.default_rng(5).uniform(0, 2e5, n_glob))
        g2l = np.full(n_glob, -1)
        g2l[part.l2g] = np.arange(part.l2g.size)
        y = np.zeros(part.nowned)
        for li in range(part.nowned):
            gi = part.l2g[li]
            cols = cj[rp[gi]:rp[gi + 1]]
            assert (g2l[cols] >= 0).all()                                  # every coupling of an owned row is owned or a ghost
            y[li] = val[rp[gi]:rp[gi + 1]] @ x[g2l[cols]]
        q.put((rank, part.l2g[:part.nowned].copy(), y, x.copy(), part.l2g.copy(), int(part.nghost)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gloo_halo_exchange_of_the_whole_column_partition(world):
    """SURVEY 8(e) on CPU processes: the halo plan of the whole-column partition run with real messages (gloo, world 2 and 3) -- after
    the exchange every ghost holds its owner's value, and the owned rows of J x formed rank by rank equal the global product"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_halo_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=180) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    nx, ny, nz = 5, 2 * world, 4
    n = nx * ny * nz
    xg = np.random.default_rng(99).standard_normal(n)
    glob = P.vchannel(nx, ny, nz)
    o = P.to_oracle(glob)
    o.set_dtime(glob.dt); o.pre_solve()
    rp, cj, val = o.jacobian(glob.press_ic - np.random.default_rng(5).uniform(0, 2e5, n))
    from oracle import vsfm_oracle as VO
    y_ref = VO.spmv(rp, cj, val, xg)
    seen = np.zeros(n, dtype=bool)
    for rank, own, y, x, l2g, nghost in res:
        assert np.array_equal(x, xg[l2g])                                  # ghosts hold their owners' values
        assert nghost == (1 if rank in (0, world - 1) else 2) * nz * nx    # nz x nx cells per neighbour
        assert np.abs(y - y_ref[own]).max() <= 1e-13 * np.abs(y_ref).max()
        seen[own] = True
    assert seen.all()


def _hg_worker(rank, world, port, q):
    g = PAR.HostGroup(rank=rank, world=world, addr="127.0.0.1", port=port, timeout=60.0)
    uid = g.broadcast_bytes(bytes(range(128)) if rank == 0 else b"")
    g.barrier()
    mx = g.allreduce([1.0 + rank, -rank], op="max")
    sm = g.allreduce([10.0 * (rank + 1)], op="sum")
    g.barrier()
    g.close()
    q.put((rank, uid, mx, sm))


@pytest.mark.parametrize("world", [2, 3])
def test_tcp_host_group(world):
    """the rendezvous bench.py uses under torch.distributed.run (no torch in the rank processes)"""
    import multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hg_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r, uid, mx, sm in res:
        assert uid == bytes(range(128))
        assert mx == [float(world), 0.0]
        assert sm == [10.0 * world * (world + 1) / 2]


def test_host_group_single_rank_is_trivial():
    g = PAR.HostGroup(rank=0, world=1)
    assert g.broadcast_bytes(b"abc") == b"abc"
    assert g.allreduce([2.0, 3.0], op="max") == [2.0, 3.0]
    g.barrier()
    g.close()


@pytest.mark.parametrize("mode", ["columns", "ranges"])
def test_partition_mesh_halo_plans_are_consistent(mode):
    """mpp_amd.parallel.partition_mesh, both modes: every rank's ghosts from rank q are, in order, the cells of q's send list to it;
    owned cells partition the mesh; every kept connection has an owned end and no connection joins two ghosts"""
    import numpy as np
    import problems as P
    from mpp_amd import parallel as PAR
    prob = P.vchannel(5, 4, 7)
    n = prob.mesh.ncells_local
    for nranks in (2, 3, 4):
        parts = PAR.partition_mesh(prob.mesh, nranks, mode)
        assert all(pt.mode == mode for pt in parts)
        owned = np.concatenate([pt.l2g[:pt.nowned] for pt in parts])
        assert np.array_equal(np.sort(owned), np.arange(n))
        nconn_owned_both = 0
        for r, pr in enumerate(parts):
            assert np.array_equal(pr.l2g[:pr.nowned], np.sort(pr.l2g[:pr.nowned]))       # owned cells in ascending global id
            off = pr.nowned
            for k, q in enumerate(pr.peer):
                gl = pr.l2g[off:off + pr.recv_count[k]]
                off += pr.recv_count[k]
                pq = parts[q]
                kk = list(pq.peer).index(r)
                sl = pq.l2g[pq.send_cells[pq.send_ptr[kk]:pq.send_ptr[kk + 1]] - 1]
                assert np.array_equal(gl, sl)
            assert off == pr.mesh.ncells_local
            cs = pr.mesh.internal_connections()
            up_owned, dn_owned = cs.id_up <= pr.nowned, cs.id_dn <= pr.nowned
            assert (up_owned | dn_owned).all()
            nconn_owned_both += int((up_owned & dn_owned).sum()) * 2 + int((up_owned ^ dn_owned).sum())
        assert nconn_owned_both == 2 * prob.mesh.internal_connections().num_connections


def test_partition_mesh_by_default_gives_every_rank_whole_soil_columns():
    """SURVEY 8(e) / MeshCreateFromCLMCols (src/mpp/dtypes/MeshType.F90:293-460; :362 ncells_ghost = ncols_ghost*nlevgrnd): ranks
    own x-y blocks of WHOLE columns, ghosts are whole neighbour columns, and the halo of a 7-point nx x ny x nz grid cut in y is
    nz x nx cells per neighbour (SURVEY.md:381-382: nz*nx*8 B per neighbour per SpMV)"""
    import numpy as np
    import problems as P
    from mpp_amd import parallel as PAR
    nx, ny, nz = 5, 6, 7
    prob = P.vchannel(nx, ny, nz)
    col = PAR.column_of_cells(prob.mesh)
    ids = np.arange(nx * ny * nz)
    assert np.array_equal(col, ids % (nx * ny))                  # x-fastest ids: column = (jj, ii), numbered by the lowest cell
    for nranks in (2, 3, 6):
        parts = PAR.partition_mesh(prob.mesh, nranks)            # mode "auto": the mesh has vertical connections -> columns
        rows = ny // nranks
        for r, pt in enumerate(parts):
            assert pt.mode == "columns" and pt.ncols_owned == rows * nx and pt.nowned == rows * nx * nz
            own_cols = np.unique(col[pt.l2g[:pt.nowned]])
            assert np.array_equal(own_cols, np.arange(r * rows * nx, (r + 1) * rows * nx))          # a contiguous range of jj
            for c in own_cols:                                    # no column is cut
                assert np.isin(ids[col == c], pt.l2g[:pt.nowned]).all()
            ghost_cols = np.unique(col[pt.l2g[pt.nowned:]])
            for c in ghost_cols:                                  # ghosts are whole columns too
                assert np.isin(ids[col == c], pt.l2g[pt.nowned:]).all()
            assert pt.recv_count.tolist() == [nz * nx] * len(pt.peer)
            assert np.diff(pt.send_ptr).tolist() == [nz * nx] * len(pt.peer)
            assert pt.peer.tolist() == [q for q in (r - 1, r + 1) if 0 <= q < nranks]
            # the local numbering of the owned block is again x-fastest: natural-order ILU(0) of a rank = PETSc's on that block
            jj0 = r * rows
            kk, rem = np.divmod(pt.l2g[:pt.nowned], nx * ny)
            assert np.array_equal(np.arange(pt.nowned), kk * rows * nx + (rem - jj0 * nx))
    # a mesh of independent soil columns (vertical-only): the same call shards the columns, no ghosts
    cols = P.celia1990(ncols=10, nz=6)
    for pt in PAR.partition_mesh(cols.mesh, 3):
        assert pt.nghost == 0 and pt.mode == "columns" and pt.nowned % 6 == 0
    # no vertical connection at all (the 1-D-in-x MMS meshes): id ranges
    flat = P.th_mms(12)
    assert all(pt.mode == "ranges" for pt in PAR.partition_mesh(flat.mesh, 2))
    with pytest.raises(ValueError):
        PAR.partition_mesh(prob.mesh, nx * ny + 1)


# ------------------------------------------------------------------------------------------------------------
# bench.py's own rank launcher (`python bench.py --gpus N` invoked directly): the host path up to, not including,
# mppgpu_create -- rendezvous, column sharding, id broadcast, barrier, max-over-ranks; and failure propagation
# ------------------------------------------------------------------------------------------------------------
def _run_bench(*argv, timeout=120):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + list(argv), env=env, capture_output=True,
                          text=True, timeout=timeout)


@pytest.mark.parametrize("world", [2, 4])
def test_bench_launcher_starts_its_own_ranks(world):
    import json
    r = _run_bench("--gpus", str(world), "--dry-run-ranks", "--nx", "10", "--ny", "7")
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout          # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["dry_run"] and out["n_gpus"] == world and out["unique_id_ok"]
    assert out["columns"] == 70.0 and out["elapsed_max"] == float(world)
    assert out["first_column_of_rank"] == [float(PAR.shard_columns(70, world, k)[0]) for k in range(world)]


def test_bench_launcher_thermal_workload_builds_every_ranks_slab():
    """`bench.py --workload thermal --gpus 4` (BASELINE config 4's rank count) up to, not including, the first GPU call: every rank
    builds its own z-slab (owned planes + one ghost plane per neighbour) without the global mesh; the slabs tile the grid, what is
    sent equals what is received, every boundary face belongs to exactly one rank"""
    import json
    n, world = 8, 4
    r = _run_bench("--gpus", str(world), "--dry-run-ranks", "--workload", "thermal", "--n", str(n))
    assert r.returncode == 0, r.stderr
    out = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
    assert out["workload"] == "thermal" and out["n_gpus"] == world and out["unique_id_ok"]
    assert out["cells"] == out["owned_cells"] == float(n ** 3)
    assert out["ghost_cells"] == out["halo_values_sent"] == float(2 * (world - 1) * n * n)
    assert out["boundary_faces"] == float(6 * n * n)
    assert out["first_cell_of_rank"] == [float(k * n ** 3 // world) for k in range(world)]


def test_bench_launcher_th_workload_shards_the_columns():
    """`bench.py --workload th --gpus 8` (config 5's rank count): the coupled problem's columns shard like the VSFM grid's"""
    import json
    r = _run_bench("--gpus", "8", "--dry-run-ranks", "--workload", "th", "--nx", "6", "--ny", "5", "--nz", "7")
    assert r.returncode == 0, r.stderr
    out = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
    assert out["workload"] == "th" and out["n_gpus"] == 8 and out["columns"] == 30.0
    assert out["cells"] == 210.0 and out["unknowns"] == 420.0


@pytest.mark.parametrize("workload,world", [("vsfm3d", 4), ("vsfm3d", 8), ("th", 4)])
def test_bench_launcher_3d_workloads_give_every_rank_whole_columns_and_their_ghost_columns(workload, world):
    """`bench.py --workload vsfm3d --gpus N` / `--workload th --mesh 3d --gpus N` up to, not including, the first GPU call: every rank
    builds its own x-y block of WHOLE soil columns (workloads.vchannel_rank / th_vchannel_rank: no global mesh) + the neighbours' edge
    columns as ghosts; the blocks tile the grid, what is sent equals what is received, the halo per neighbour is nz x nx cells
    (nz * nx * 8 B per neighbour per SpMV), every rank's first column is the first of its rows"""
    import json
    nx, ny, nz = 6, 8, 5
    argv = ["--gpus", str(world), "--dry-run-ranks", "--workload", workload, "--nx", str(nx), "--ny", str(ny), "--nz", str(nz)]
    if workload == "th":
        argv += ["--mesh", "3d"]
    r = _run_bench(*argv)
    assert r.returncode == 0, r.stderr
    out = json.loads([l for l in r.stdout.splitlines() if l.strip()][0])
    assert out["workload"] == workload and out["n_gpus"] == world and out["unique_id_ok"] and out["mesh"] == "3d"
    assert out["partition"] == "columns"
    assert out["cells"] == out["owned_cells"] == float(nx * ny * nz) and out["columns"] == float(nx * ny)
    assert out["ghost_columns"] == float(2 * (world - 1) * nx)
    assert out["ghost_cells"] == out["halo_values_sent"] == float(2 * (world - 1) * nx * nz)
    assert out["halo_bytes_per_neighbour_per_exchange"] == float(nz * nx * 8 * (2 if workload == "th" else 1))
    assert out["first_column_of_rank"] == [float(k * (ny // world) * nx) for k in range(world)]
    assert out["unknowns"] == float(nx * ny * nz * (2 if workload == "th" else 1))


@pytest.mark.parametrize("dims,world", [((5, 4, 6), 2), ((4, 6, 5), 3), ((3, 8, 4), 4), ((5, 5, 4), 2)])
def test_3d_blocks_built_directly_equal_the_split_of_the_global_problem(dims, world):
    """workloads.vchannel_rank / th_vchannel_rank (what a rank of `bench.py --workload vsfm3d` / `--workload th --mesh 3d` builds: its
    rows and ghost rows only) against split_problem / split_th_problem of the whole problem: the same cells in the same order,
    connections, halo plan, soils, start and sources, array by array ((5, 5, 4) over 2 ranks cuts a row: the generic route is taken)"""
    def same(a, b):
        return (a is None and b is None) or (np.asarray(a).shape == np.asarray(b).shape and np.array_equal(np.asarray(a), np.asarray(b)))

    ref = P.split_problem(P.vchannel(*dims), world)
    tref = P.split_th_problem(P.th_vchannel(*dims), world)
    for rank in range(world):
        part, lp = P.vchannel_rank(*dims, world, rank)
        tpart, tlp = P.th_vchannel_rank(*dims, world, rank)
        rpart, rlp = ref[rank]
        assert part.nghost == rpart.nghost and same(part.l2g, rpart.l2g) and same(tpart.l2g, rpart.l2g)
        assert part.mode == rpart.mode == "columns" and part.ncols_owned == rpart.ncols_owned
        for f in ("peer", "send_ptr", "send_cells", "recv_count"):
            assert same(getattr(part, f), getattr(rpart, f)), f
        for f in ("x", "y", "z", "dx", "dy", "dz", "area_xy", "vol", "is_active"):
            assert same(getattr(lp.mesh, f), getattr(rlp.mesh, f)), f
        a, b = lp.mesh.internal_connections(), rlp.mesh.internal_connections()
        for f in ("id_up", "id_dn", "dist_up", "dist_dn", "area", "unit_vec", "itype"):
            assert same(getattr(a, f), getattr(b, f)), f
        for f in ("perm", "por", "alpha", "lam", "sat_res", "press_ic", "satfunc_type"):
            assert same(getattr(lp, f), getattr(rlp, f)), f
        for f in ("perm", "por", "press_ic", "temp_ic", "heat_cap", "tcond_wet", "tcond_dry", "talpha", "den_soil"):
            assert same(getattr(tlp, f), getattr(tref[rank][1], f)), f
        (t1, c1, v1), (t2, c2, v2) = tlp.energy_sss[0], tref[rank][1].energy_sss[0]
        assert t1 == t2 and same(v1, v2) and same(c1.id_dn, c2.id_dn)


@pytest.mark.parametrize("dims,world", [((5, 4, 6), 2), ((5, 4, 6), 3), ((7, 3, 8), 4), ((6, 5, 9), 3), ((4, 3, 5), 2)])
def test_thermal_slab_built_directly_equals_the_split_of_the_global_problem(dims, world):
    """workloads.thermal_mms_rank (what a rank of `bench.py --workload thermal` builds: its slab only) against
    split_thermal_problem of the whole problem: the same cells in the same order, connections, halo plan, boundary faces, soil,
    sources and exact solution, array by array ((4, 3, 5) over 2 ranks cuts a plane: the generic route is taken)"""
    nx, ny, nz = dims
    ref = P.split_thermal_problem(P.thermal_mms(3, nx, ny, nz, replicate_reference_soil_indexing=False), world)

    def same(a, b):
        return (a is None and b is None) or (np.asarray(a).shape == np.asarray(b).shape and np.array_equal(np.asarray(a), np.asarray(b)))

    for rank in range(world):
        part, lp = P.thermal_mms_rank(nx, ny, nz, world, rank)
        rpart, rlp = ref[rank]
        assert part.nghost == rpart.nghost and same(part.l2g, rpart.l2g)
        for f in ("peer", "send_ptr", "send_cells", "recv_count"):
            assert same(getattr(part, f), getattr(rpart, f)), f
        for f in ("x", "y", "z", "dx", "dy", "dz", "area_xy", "vol", "is_active"):
            assert same(getattr(lp.mesh, f), getattr(rlp.mesh, f)), f
        assert (lp.mesh.ncells_local, lp.mesh.nlev, lp.mesh.ncols) == (rlp.mesh.ncells_local, rlp.mesh.nlev, rlp.mesh.ncols)
        a, b = lp.mesh.internal_connections(), rlp.mesh.internal_connections()
        for f in ("id_up", "id_dn", "dist_up", "dist_dn", "area", "unit_vec", "itype"):
            assert same(getattr(a, f), getattr(b, f)), f
        for f in ("id_up", "id_dn", "dist_up", "dist_dn", "area", "unit_vec"):
            assert same(getattr(lp.bc, f), getattr(rlp.bc, f)), f
        for f in ("tkdry", "bc_temperature", "heat_source", "exact"):
            assert same(getattr(lp, f), getattr(rlp, f)), f


def test_bench_launcher_reports_a_failed_rank_and_stops_the_others():
    r = _run_bench("--gpus", "2", "--dry-run-ranks", "--nx", "4", "--ny", "4", "--fail-rank", "1")
    assert r.returncode != 0
    assert "rank 1 exited with code" in r.stderr and "injected failure" in r.stderr
    assert r.stdout.strip() == ""             # no JSON line from a failed run


def test_bench_launcher_time_limit():
    """a rank that never arrives: the other one is stopped at the launcher's time limit (exit code 124), nothing hangs"""
    import time
    t0 = time.time()
    r = _run_bench("--gpus", "2", "--dry-run-ranks", "--nx", "4", "--ny", "4", "--fail-rank", "-2", "--launch-timeout", "3")
    assert time.time() - t0 < 60
    assert r.returncode == 124 and "launcher time limit" in r.stderr


def test_rccl_binding_guard_runs_without_a_gpu():
    """the run-time half of the RCCL binding guard (the build-time half are the static_asserts of mpp_amd/csrc/runtime.cpp against
    <rccl/rccl.h>: enumerator values and prototypes): the library that dlopen finds must report a 2.x version of the same major
    as that header.  A one-rank all-reduce is the identity for every datatype and operation, so this -- not a single-GPU solve --
    is what catches a wrong constant"""
    import ctypes as ct
    from mpp_amd import _lib
    L = _lib.lib()
    v = ct.c_int(0)
    rc = L.mppgpu_comm_runtime_version(ct.byref(v))
    if rc != 0 and b"cannot load RCCL" in L.mppgpu_last_error(None):
        pytest.skip("no RCCL on this machine")
    assert rc == 0, L.mppgpu_last_error(None)
    assert 20000 <= v.value < 30000
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mpp_amd", "csrc", "runtime.cpp")).read()
    for what in ("ncclDouble == kNcclFloat64", "ncclSum == kNcclSum", "ncclMax == kNcclMax", "decltype(&ncclCommInitRank)",
                 "decltype(&ncclAllReduce)"):
        assert what in src
