"""Wire format of the regression files and the comparator semantics, pinned on the reference's own
data files: re-writing the parsed golden values must reproduce the baseline text byte for byte."""
import os

import numpy as np
import pytest

import problems as P
from mpp_amd import constants as C
from mpp_amd import regression as R


def test_fortran_e_format():
    assert R.fortran_e(3535.5) == "  0.3535500000000E+04"
    assert R.fortran_e(0.2987655977304) == "  0.2987655977304E+00"
    assert R.fortran_e(-1.5e-7) == " -0.1500000000000E-06"
    assert R.fortran_e(0.0) == "  0.0000000000000E+00"
    assert R.fortran_e(0.99999999999999999) == "  0.1000000000000E+01"
    assert R.fortran_e(1.0) == "  0.1000000000000E+01"


@pytest.mark.parametrize("name", ["vsfm_celia1990", "vsfm_vchannel"])
def test_oracle_run_rewrites_the_baseline_file_verbatim(golden_dir, name):
    gold_text = open(os.path.join(golden_dir, name + ".regression.baseline")).read()
    prob = P.celia1990() if name == "vsfm_celia1990" else P.vchannel()
    o = P.to_oracle(prob)
    for s in range(prob.nstep):
        o.step_dt(prob.dt, s + 1)
    w = R.RegressionWriter(None, num_cells=5)
    w.WriteData("liquid_pressure", "pressure", o.get(C.AUXVAR_INTERNAL, C.VAR_PRESSURE))
    w.WriteData("liquid_saturation", "general", o.get(C.AUXVAR_INTERNAL, C.VAR_LIQ_SAT))
    ours, gold = R.parse_regression(w.text()), R.parse_regression(gold_text)
    assert list(ours) == list(gold)
    for sec in gold:
        assert list(ours[sec]) == list(gold[sec])       # same keys ("cell    1", "cell 1201", ...) in the same order
    # the comparator with the reference's own cfg tolerances (pressure 1e-10 absolute for celia1990): the CPU
    # oracle differs from the PETSc run only in the last printed digit
    tol = R.Tolerances()
    tol.update_from_name("pressure", "2.0e-8 absolute")
    tol.update_from_name("general", "1.0e-12 absolute")
    status, msgs = R.compare(w.text(), gold_text, tol)
    assert status == "pass", msgs
    # lines whose values agree to the last digit are byte-identical to the reference file
    same = sum(1 for a, b in zip(w.text().splitlines(), gold_text.splitlines()) if a == b)
    assert same >= 0.6 * len(gold_text.splitlines())


def test_comparator_semantics(golden_dir):
    gold_text = open(os.path.join(golden_dir, "vsfm_celia1990.regression.baseline")).read()
    cfg = open(os.path.join(golden_dir, "vsfm.cfg")).read()
    tol = R.Tolerances.from_cfg(cfg, "vsfm_celia1990")
    assert tol.t["pressure"] == {"value": 1.0e-10, "type": "absolute"}
    assert tol.t["general"] == {"value": 1.0e-16, "type": "absolute"}
    assert R.Tolerances.from_cfg(cfg, "vsfm_vchannel").t["pressure"] == {"value": 1.0e-12, "type": "absolute"}
    assert R.compare(gold_text, gold_text, tol)[0] == "pass"
    # a value off by more than the tolerance, a missing key and a missing section all fail
    bad = gold_text.replace("0.9398362808479E+05", "0.9398362808489E+05")
    assert R.compare(bad, gold_text, tol)[0] == "fail"
    assert R.compare(gold_text.replace("cell   21 =   0.3535500000000E+04\n", "", 1), gold_text, tol)[0] == "fail"
    assert R.compare(gold_text.split("[liquid_saturation]")[0], gold_text, tol)[0] == "fail"
    rel = R.Tolerances()
    rel.update_from_name("pressure", "1.0e-12 relative")
    rel.update_from_name("general", "1.0e-3 percent")
    assert R.compare(bad, gold_text, rel)[0] == "fail"       # 1e-7 Pa of 9.4e4 Pa = 1.06e-12 relative
    rel.update_from_name("pressure", "2.0e-12 relative")
    assert R.compare(bad, gold_text, rel)[0] == "pass"
    with pytest.raises(RuntimeError):
        rel.update_from_name("salinity", "1 absolute")


# ------------------------------------------------------------------------------------------------------------
# namelist inputs (standalone_mpp.F90:69-70, 93-122; thermal_mms_problem.F90:59, 99-119), pinned on the reference's own
# .namelist files (tests/golden/namelists: data files, verbatim)
# ------------------------------------------------------------------------------------------------------------
def _nl(golden_dir, name):
    return open(os.path.join(golden_dir, "namelists", name + ".namelist")).read()


def test_namelist_reader_on_the_reference_inputs(golden_dir):
    from mpp_amd import namelist as NL
    from mpp_amd import standalone as S
    expect = {"vsfm_celia1990": ("vsfm_celia1990", 5), "vsfm_vchannel": ("vsfm_vchannel", 5), "thermal_mms": ("thermal_mms", 5),
              "thermal_mms_2d": ("thermal_mms", 2), "thermal_mms_3d": ("thermal_mms", 3), "mass_and_heat": ("mass_and_heat", 5),
              "th_mms": ("th_mms", 5), "vsfm_spac_campbell_multi_goveqns": ("vsfm_spac_campbell", 5)}
    for name, (ptype, ncell) in expect.items():
        nl = NL.read_driver_namelist(_nl(golden_dir, name))
        assert nl["problem_type"] == ptype and nl["num_cells"] == ncell and nl["write_regression_output"] is True
    # &problem_options of the thermal problems: defaults by problem type, then the file's values
    assert S._thermal_mms_dims(NL.parse(_nl(golden_dir, "thermal_mms"))) == (1, (20, 1, 1))
    assert S._thermal_mms_dims(NL.parse(_nl(golden_dir, "thermal_mms_2d"))) == (2, (3, 3, 1))
    assert S._thermal_mms_dims(NL.parse(_nl(golden_dir, "thermal_mms_3d"))) == (3, (3, 3, 3))
    # "/" directly followed by the next group, a logical in a group this path does not read
    g = NL.parse(_nl(golden_dir, "vsfm_spac_campbell_multi_goveqns"))
    assert g["problem_options"]["multi_goveqns_formulation"] is True
    with pytest.raises(S.DriverError, match="outside the GPU solve path"):
        S.run_problem("vsfm_spac_campbell", g)


def test_namelist_syntax():
    from mpp_amd import namelist as NL
    text = """! a comment line
 &A  x = 1.5d0, y=-2 , name = 'it''s' ! trailing comment with a / and a &
   flag=.FALSE.  list = 1, 2,3   big = 1.0E+3
 /
&b
  s = "a / in a string"
  t = T
&end
"""
    g = NL.parse(text)
    assert g["a"] == {"x": 1.5, "y": -2, "name": "it's", "flag": False, "list": [1, 2, 3], "big": 1000.0}
    assert g["b"] == {"s": "a / in a string", "t": True}
    assert NL.read_group(g, "a", {"x": 0.0, "y": 0, "name": "", "flag": True, "list": [], "big": 0.0})["y"] == -2
    assert NL.read_group(g, "zz", {"k": 3}) == {"k": 3}                  # a missing optional group leaves the defaults
    with pytest.raises(NL.NamelistError):
        NL.read_group(g, "zz", {"k": 3}, required=True)
    with pytest.raises(NL.NamelistError, match="not a member"):
        NL.read_group(g, "b", {"s": ""})                                 # `t` is not declared
    with pytest.raises(NL.NamelistError, match="expects int"):
        NL.read_group(g, "a", dict(x=0, y=0, name="", flag=True, list=[], big=0.0))
    with pytest.raises(NL.NamelistError, match="not terminated"):
        NL.parse("&a x = 1")
    with pytest.raises(NL.NamelistError):
        NL.parse("&a x = 1.2.3 /")
    with pytest.raises(NL.NamelistError, match="mpp_driver"):
        NL.read_driver_namelist("&regression_test num_cells = 2 /")
