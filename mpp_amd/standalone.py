"""The standalone driver of the reference, for the problems on this path: read the namelist, run the problem
on the GPU through the reference-shaped host API, write `<namelist base>.regression`.

    python -m mpp_amd.standalone -namelist regression_tests/vsfm/vsfm_celia1990.namelist [-ksp_type bcgs -pc_type bjacobi ...]

Mirrors src/driver/standalone/standalone_mpp.F90: `&mpp_driver problem_type` selects the problem (:124-200),
`&regression_test write_regression_output, num_cells` the output (:69-70), `&problem_options` belongs to the problem module
(thermal_mms: nx, ny, nz, problem_type with the defaults and checks of thermal_mms_problem.F90:59-205).  Sections and
categories of the regression file are those of the drivers' `output_regression_*` routines
(vsfm_celia1990_problem.F90:427-436, vsfm_vchannel_problem.F90:1085-1094, thermal_mms_problem.F90:871-872,
mass_and_heat_model_problem.F90:682-694).  Solver options not given are PETSc's defaults, as in the reference runs
(GMRES(30) + ILU(0)); `-name value` pairs on the command line play the role of the PETSc options database.

th_mms forms its manufactured mass and heat sources on the host from the constitutive laws, as the reference driver does
(th_mms_problem.F90:901-934, 1262-1452; mpp_amd.workloads.th_mms with mpp_amd.laws).  Problems of the reference that are
outside this path (SURVEY 2 OUT OF SCOPE: vsfm_spac*, the canopy models) are refused by name.
"""
from __future__ import annotations

import os
import sys

import numpy as np

from . import constants as C
from . import namelist as NL
from . import regression as R
from . import workloads as P

OUT_OF_SCOPE = ("vsfm_spac", "vsfm_spac_campbell", "heat_transport_1D", "mlc", "leafbndlyr", "longwave", "shortwave", "photosynthesis")
DEFAULT_OPTIONS = {"ksp_type": "gmres", "pc_type": "ilu"}


class DriverError(RuntimeError):
    pass


def _thermal_mms_dims(groups):
    """thermal_mms_problem.F90:59-205: defaults by problem_type, then &problem_options, then the shape checks"""
    first = NL.read_group(groups, "problem_options", {"nx": 0, "ny": 0, "nz": 0, "problem_type": 1})
    ptype = first["problem_type"]
    if ptype not in (1, 2, 3):
        raise DriverError("Unknown -problem_type")
    defaults = {1: (20, 1, 1), 2: (20, 20, 1), 3: (20, 20, 20)}[ptype]
    g = groups.get("problem_options", {})
    nx, ny, nz = (g.get(k, d) for k, d in zip(("nx", "ny", "nz"), defaults))
    if ptype == 1 and (nx <= 1 or ny != 1 or nz != 1):
        raise DriverError("For STEADY_STATE_1D problem, need nx > 1, ny = 1, nz = 1")
    if ptype == 2 and (nx <= 1 or ny < 1 or nz != 1):
        raise DriverError("For STEADY_STATE_2D problem, need nx > 1, ny >= 1, nz = 1")
    if ptype == 3 and (nx < 1 or ny < 1 or nz < 1):
        raise DriverError("For STEADY_STATE_3D problem, need nx, ny, nz >= 1")
    return ptype, (nx, ny, nz)


def run_problem(problem_type, groups, options=None, device_id=0):
    """run one problem; returns [(section name, category, data)] in the order the reference driver writes them"""
    opts = dict(DEFAULT_OPTIONS)
    opts.update(options or {})
    if problem_type in ("vsfm_celia1990", "vsfm_vchannel"):
        prob = P.celia1990() if problem_type == "vsfm_celia1990" else P.vchannel()
        mpp = P.to_gpu(prob, opts, device_id)
        for s in range(prob.nstep):
            # the drivers set the boundary values before every step (set_bondary_conditions, celia1990_problem.F90:86-100)
            for b, (_t, _cs, vals) in enumerate(prob.bcs):
                mpp.soe.SetDataFromCLM(C.AUXVAR_BC, C.VAR_BC_SS_CONDITION, b + 1, vals)
            conv, reason = mpp.soe.StepDT(prob.dt, s + 1)
            if not conv:
                raise DriverError("%s: StepDT %d did not converge (reason %d)" % (problem_type, s + 1, reason))
        return [("liquid_pressure", "pressure", mpp.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE, -1)),
                ("liquid_saturation", "general", mpp.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_LIQ_SAT, -1))]
    if problem_type == "thermal_mms":
        ptype, dims = _thermal_mms_dims(groups)
        tp = P.thermal_mms(ptype, *dims)
        # a regression run is compared with files the REFERENCE wrote: its two thermal defects (stale `area` in the Dirichlet right-hand
        # side, GoveqnThermalKSPTemperatureSoilType.F90:906-911; `dhsdT**area`, :1215) are switched on unless the caller says otherwise,
        # so that like is compared with like whatever the face areas are (the library's own default is the corrected arithmetic)
        mpp = P.to_thermal_gpu(tp, dict({"thermal_reference_defects": 1}, **(opts or {})), device_id)
        conv, _ = mpp.soe.StepDT(tp.dt, 1)
        if not conv:
            raise DriverError("thermal_mms: the linear solve did not converge")
        return [("temperature", "general", mpp.soe.GetSoln())]
    if problem_type in ("mass_and_heat", "th_mms"):
        tp = P.mass_and_heat() if problem_type == "mass_and_heat" else P.th_mms()   # (th_mms: nx = 20, th_mms_problem.F90:943)
        mpp = P.to_th_gpu(tp, opts, device_id)
        for s in range(tp.nstep):
            P.th_gpu_set_conditions(tp, mpp)
            conv, reason = mpp.soe.StepDT(tp.dt, s + 1)
            if not conv:
                raise DriverError("%s: StepDT did not converge (reason %d)" % (problem_type, reason))
        return [("liquid_pressure", "pressure", mpp.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_PRESSURE)),
                ("temperature", "temperature", mpp.soe.GetDataForCLM(C.AUXVAR_INTERNAL, C.VAR_TEMPERATURE))]
    if problem_type in OUT_OF_SCOPE:
        raise DriverError("problem_type '%s' is outside the GPU solve path (plant hydraulics / canopy models)" % problem_type)
    raise DriverError("Unknown problem_type '%s'" % problem_type)   # standalone_mpp.F90:196-199


def run_namelist(text, filename_base=None, options=None, device_id=0):
    """standalone_mpp.F90 from the namelist buffer on: returns the regression text (None when write_regression_output is
    false) and writes `<filename_base>.regression` when a base name is given"""
    nl = NL.read_driver_namelist(text)
    sections = run_problem(nl["problem_type"], nl["groups"], options, device_id)
    if not nl["write_regression_output"]:
        return None
    w = R.RegressionWriter(filename_base, nl["num_cells"])
    for name, category, data in sections:
        w.WriteData(name, category, np.asarray(data))
    w.CloseOutput()
    return w.text()


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if len(argv) % 2:
        raise SystemExit("usage: python -m mpp_amd.standalone -namelist <filename> [-option value ...]")
    pairs = {argv[i].lstrip("-"): argv[i + 1] for i in range(0, len(argv), 2)}
    path = pairs.pop("namelist", "")
    if not path:
        raise SystemExit("ERROR: -namelist <filename> was not defined. Bailing out.")   # standalone_mpp.F90:84-87
    try:
        text = open(path).read()
    except OSError:
        raise SystemExit("ERROR: Unable to open namelist file: %s" % path)
    base = os.path.splitext(path)[0]                                                    # remove_filename_extension
    device = int(pairs.pop("device", 0))
    try:
        run_namelist(text, base, pairs, device)
    except (DriverError, NL.NamelistError) as e:
        raise SystemExit("ERROR: %s" % e)


if __name__ == "__main__":
    main()
