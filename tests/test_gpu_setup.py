"""Set-up structures built on the device (mpp_amd/csrc/setup_kernels.hip) against the host-built ones.

VERDICT r03, item 8: "patterns array_equal to the host-built ones".  Option setup_verify = 1 makes mppgpu_setup run both routes and
compare them entry by entry inside the library (an MPPError if anything differs; assembly_info()["setup_structures_verified"]
counts the comparisons that passed).  On top of that every problem is solved with setup_device = 1 and 0: the same bits."""
import numpy as np
import pytest

import problems as P
from mpp_amd import constants as C

pytestmark = pytest.mark.gpu


def _verified(soe):
    return soe.assembly_info()["setup_structures_verified"]


@pytest.mark.parametrize("assembly", ["general", "columns"])
def test_incidence_lists_of_a_3d_vsfm_mesh_with_lateral_connections(assembly):
    """vchannel (src/driver/standalone/vsfm/vsfm_vchannel_problem.F90): vertical + horizontal connections in one set, the ELL copy
    of the general flux kernel included"""
    prob = P.vchannel(8, 6, 9)
    res = {}
    for dev in (1, 0):
        g = P.to_gpu(prob, {"assembly": assembly, "setup_device": dev, "setup_verify": dev, "pc_type": "ilu", "ksp_type": "gmres"})
        if dev:
            assert _verified(g.soe) >= 1
        conv, _ = g.soe.StepDT(prob.dt, 1)
        assert conv
        res[dev] = g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)
    assert np.array_equal(res[0], res[1])


def test_incidence_lists_with_a_repeated_pair_of_cells_and_shuffled_connection_order():
    """two connections between the same pair of cells (the ELL copy must be refused: one matrix slot met twice by a row) and the
    connection list in a random order (a row's list is ordered by the caller's connection number, whatever order the device's
    atomics filled it in)"""
    from mpp_amd import mesh as M
    from test_gpu_parity import _hetero, _operator_parity
    prob, x = _hetero(ncols=9, nz=7, seed=5)
    cs = prob.mesh.intrn_conn_sets[0]
    k = 17
    area = np.asarray(cs.area, dtype=float).copy()
    area[k] *= 0.5
    dup = lambda a: np.concatenate([np.asarray(a), np.asarray(a)[k:k + 1]])
    order = np.random.default_rng(3).permutation(np.asarray(cs.id_up).size + 1)
    pick = lambda a: dup(a)[order]
    prob.mesh.intrn_conn_sets = [M.ConnectionSet(pick(cs.id_up), pick(cs.id_dn), pick(cs.dist_up), pick(cs.dist_dn), pick(area),
                                                 pick(cs.unit_vec), None if cs.itype is None else pick(cs.itype))]
    o, g, _, _ = _operator_parity(prob, x, {"setup_verify": 1})
    assert _verified(g.soe) >= 1
    _operator_parity(prob, x, {"setup_device": 0})


@pytest.mark.parametrize("ptype", [1, 2, 3])
def test_incidence_lists_of_the_thermal_mms_meshes(ptype):
    tp = P.thermal_mms(ptype, 6, 5, 7)
    res = {}
    for dev in (1, 0):
        g = P.to_thermal_gpu(tp, {"setup_device": dev, "setup_verify": dev, "ksp_rtol": 1e-12})
        if dev:
            assert _verified(g.soe) >= 1
        conv, _ = g.soe.StepDT(tp.dt, 1)
        assert conv
        res[dev] = np.array(g.soe.GetSoln())
    assert np.array_equal(res[0], res[1])


def test_incidence_lists_of_the_coupled_th_vchannel():
    tp = P.th_vchannel(6, 5, 7)
    res = {}
    for dev in (1, 0):
        g = P.to_th_gpu(tp, {"setup_device": dev, "setup_verify": dev})
        P.th_gpu_set_conditions(tp, g)
        if dev:
            assert _verified(g.soe) >= 1
        conv, _ = g.soe.StepDT(tp.dt, 1)
        assert conv
        res[dev] = (g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE), g.soe.GetDataForCLM(C.AUXVAR_INTERNAL, 605))
    assert np.array_equal(res[0][0], res[1][0]) and np.array_equal(res[0][1], res[1][1])


def test_device_scan_across_many_tiles():
    """a mesh large enough for the prefix sums to span several scan tiles (2048 rows each) with a ragged last one: thermal 3-D,
    37 x 29 x 23 = 24679 cells"""
    tp = P.thermal_mms(3, 37, 29, 23)
    g = P.to_thermal_gpu(tp, {"setup_verify": 1})
    assert _verified(g.soe) >= 1


@pytest.mark.parametrize("assembly", ["columns", "general"])
def test_pattern_of_soil_columns_written_down_directly_or_built_on_the_device(assembly):
    """vertical-only soil columns (ragged last tile: 70 columns): with the column march the CSR pattern is written down from the
    layout, with the general assembly it comes from the device-built incidence lists; both against the host's general route, then
    the operators against the oracle"""
    from test_gpu_parity import _hetero, _operator_parity
    prob, x = _hetero(ncols=70, nz=11, seed=2)
    o, g, _, _ = _operator_parity(prob, x, {"assembly": assembly, "setup_verify": 1})
    assert _verified(g.soe) >= (1 if assembly == "columns" else 2)
    _operator_parity(prob, x, {"assembly": assembly, "setup_device": 0})


def test_meshes_with_ghost_cells_keep_the_identity_layout_on_either_route():
    """a rank's share of a split 3-D mesh (ghost cells last, mppgpu_set_ghost_cells): no column layout is allowed there, cells and
    connections keep the caller's numbering -- the device route must decide the same and build the same lists and pattern as the
    host's; thermal (z-slabs of the MMS grid) and VSFM (v-channel)"""
    tp = P.thermal_mms(3, 7, 6, 9, replicate_reference_soil_indexing=False)
    for _, lp in P.split_thermal_problem(tp, 3):
        g = P.to_thermal_gpu(lp, {"ksp_type": "gmres", "pc_type": "ilu", "setup_verify": 1})
        assert _verified(g.soe) >= 3          # layout, pattern, incidence lists
    prob = P.vchannel(8, 6, 9)
    for _, lp in P.split_problem(prob, 2):
        g = P.to_gpu(lp, {"setup_verify": 1})
        assert _verified(g.soe) >= 3
