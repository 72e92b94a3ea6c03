"""Regression output of the standalone drivers and its comparison against a baseline.

Writer  : regression_mod%WriteData            src/driver/standalone/util/regression.F90:76-124
          ("[name]", "category = c", "min/max/mean = e21.13", "cell i4 = e21.13" for every k-th cell, blank line;
           |value| < 1e-50 is written as 0)
Compare : regression_tests/regression_tests.py:604-770 (sections and keys must match both ways; one tolerance per
          *category*: absolute | relative | percent; relative differences divide by the first value unless it is
          zero) and :757-870 (defaults 1e-16 absolute for general / temperature / pressure / velocity; a .cfg gives
          `[default_tolerances]` and per-test overrides as "value type").
Host-side bookkeeping only (no device work).
"""
from __future__ import annotations

import configparser
from collections import OrderedDict

import numpy as np

KNOWN_CATEGORIES = ("general", "temperature", "pressure", "velocity")
ABSOLUTE, RELATIVE, PERCENT = "absolute", "relative", "percent"
DEFAULT_EPSILON = 1.0e-16


def fortran_e(value: float, width: int = 21, digits: int = 13) -> str:
    """Fortran `e21.13` edit descriptor: 0.dddddddddddddE+xx, right-justified."""
    v = float(value)
    if v == 0.0:
        mant, exp = 0.0, 0
    else:
        exp = int(np.floor(np.log10(abs(v)))) + 1
        mant = v / 10.0 ** exp
        if abs(round(mant, digits)) >= 1.0:   # 0.99999.. rounds up to 1.0
            exp += 1
            mant = v / 10.0 ** exp
    body = "%.*f" % (digits, abs(mant))
    s = ("-" if v < 0 else "") + body + "E%+03d" % exp
    return s.rjust(width)


class RegressionWriter:
    """regression_type: Init(filename_base, num_cells) / OpenOutput / WriteData / CloseOutput."""

    def __init__(self, filename_base=None, num_cells=0):
        self.filename = None if filename_base is None else filename_base + ".regression"
        self.num_cells = int(num_cells)
        self.lines = []

    def WriteData(self, name, category, data):
        data = np.asarray(data, dtype=np.float64).ravel()

        def clean(v):
            return 0.0 if abs(v) < 1.0e-50 else float(v)

        out = self.lines
        out.append("[%s]" % name.strip())
        out.append("category = %s" % category.strip())
        out.append("min = " + fortran_e(clean(data.min())))
        out.append("max = " + fortran_e(clean(data.max())))
        out.append("mean = " + fortran_e(clean(data.sum() / data.size)))
        if self.num_cells > 0:
            ncell = min(self.num_cells, data.size)
            inc = data.size // ncell
            for cell in range(1, data.size + 1, inc):
                out.append("cell %4d = %s" % (cell, fortran_e(clean(data[cell - 1]))))
            out.append("")

    def text(self):
        return "\n".join(self.lines) + "\n"

    def CloseOutput(self):
        if self.filename:
            with open(self.filename, "w") as f:
                f.write(self.text())


def parse_regression(text: str) -> "OrderedDict[str, OrderedDict]":
    out, cur = OrderedDict(), None
    for line in text.splitlines():
        line = line.strip()
        if not line:
            continue
        if line.startswith("[") and line.endswith("]"):
            cur = line[1:-1]
            out[cur] = OrderedDict()
        else:
            k, v = [t.strip() for t in line.split("=", 1)]
            out[cur][" ".join(k.split())] = v
    return out


class Tolerances:
    def __init__(self):
        self.t = {c: {"value": DEFAULT_EPSILON, "type": ABSOLUTE} for c in KNOWN_CATEGORIES}

    def update_from_name(self, category, data):
        if category not in self.t:
            raise RuntimeError("Unknown tolerance type '%s'." % category)
        value, tol_type = data.split()
        value = float(value)
        if tol_type not in (ABSOLUTE, RELATIVE, PERCENT):
            raise RuntimeError("invalid tolerance type '%s'" % tol_type)
        if value < 0.0 or (tol_type == PERCENT and value > 100.0):
            raise RuntimeError("tolerance '%s = %s [%s]' out of range" % (category, value, tol_type))
        self.t[category] = {"value": value, "type": tol_type}

    @classmethod
    def from_cfg(cls, cfg_text: str, test_name: str):
        """[default_tolerances] first, then the section of the test (regression_tests.py:394-420)."""
        tol = cls()
        cp = configparser.ConfigParser()
        cp.read_string(cfg_text)
        for section in ("default_tolerances", test_name):
            if cp.has_section(section):
                for key, val in cp.items(section):
                    if key in KNOWN_CATEGORIES:
                        tol.update_from_name(key, val)
        return tol


def compare(regression_text: str, baseline_text: str, tolerances: Tolerances | None = None):
    """Returns (status, messages) with status 'pass' or 'fail', same rules as the reference harness."""
    tol = tolerances or Tolerances()
    a_data, b_data = parse_regression(baseline_text), parse_regression(regression_text)
    msgs = []
    for section in a_data:
        if section not in b_data:
            msgs.append('section "%s" present in baseline but missing from regression' % section)
            continue
        a, b = a_data[section], b_data[section]
        for name, d in (("baseline", a), ("regression", b)):
            if "category" not in d:
                raise RuntimeError('section "%s" missing required option "category" in %s' % (section, name))
            if d["category"] not in KNOWN_CATEGORIES:
                raise RuntimeError('invalid tolerance category "%s"' % d["category"])
        if a["category"] != b["category"]:
            msgs.append("%s : data category does not match: %s vs %s" % (section, a["category"], b["category"]))
        category = a["category"]
        for key in a:
            if key not in b:
                msgs.append('%s : key "%s" present in baseline missing in regression' % (section, key))
            elif key != "category":
                av, bv = float(a[key]), float(b[key])
                diff = abs(av - bv)
                if av == 0.0 and bv != 0.0:
                    den = bv
                elif av != 0.0 and bv == 0.0:
                    den = av
                elif av == 0.0 and bv == 0.0:
                    den = 1.0
                else:
                    den = av
                t = tol.t[category]
                if t["type"] == RELATIVE:
                    diff = diff / den
                elif t["type"] == PERCENT:
                    diff = 100.0 * diff / den
                if diff > t["value"]:
                    msgs.append("%s : %s : tolerance failure : %g > %g [%s]" % (section, key, diff, t["value"], t["type"]))
        for key in b:
            if key not in a:
                msgs.append('%s : key "%s" present in regression missing in baseline' % (section, key))
    for section in b_data:
        if section not in a_data:
            msgs.append('section "%s" present in regression but missing from baseline' % section)
    return ("fail" if msgs else "pass"), msgs
