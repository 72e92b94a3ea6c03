"""Fortran namelist input of the standalone drivers.

The reference reads the whole file into a buffer and then the groups it knows from it, each with a Fortran
`read(buffer, nml=group)` (src/driver/standalone/standalone_mpp.F90:69-70, 93-122: `&mpp_driver problem_type`,
`&regression_test write_regression_output, num_cells`; the problem modules read `&problem_options`, e.g.
src/driver/standalone/thermal/thermal_mms_problem.F90:59, 99-119: `nx, ny, nz, problem_type`).  This module parses the
subset of namelist syntax those files use and a little more: `&group ... /` blocks, `name = value` pairs separated by
newlines, blanks or commas, `!` comments, character values in single or double quotes (doubled quotes inside), logicals
(`.true.`, `.false.`, `T`, `F`, `.t.`, `.f.`), integers and reals with `e` or `d` exponents, and comma- or blank-separated
value lists (returned as Python lists).  Group and variable names are case-insensitive (stored lower-case), as in Fortran.
Not handled (nothing in the reference uses it): array sections `a(2:3) =`, repeat counts `3*0.0`, derived types.
Host-side bookkeeping only.
"""
from __future__ import annotations

import re
from collections import OrderedDict


class NamelistError(ValueError):
    pass


_TOKEN = re.compile(r"""
    \s*(?:
      (?P<str>'(?:[^']|'')*'|"(?:[^"]|"")*")   |
      (?P<eq>=)                                 |
      (?P<comma>,)                              |
      (?P<word>[^\s=,'"]+)
    )""", re.X)


def _strip_comment(line: str) -> str:
    out, quote = [], None
    for ch in line:
        if quote:
            out.append(ch)
            if ch == quote:
                quote = None
        elif ch in "'\"":
            quote = ch
            out.append(ch)
        elif ch == "!":
            break
        else:
            out.append(ch)
    return "".join(out)


def _value(tok: str):
    t = tok.lower()
    if t in (".true.", ".t.", "t", "true"):
        return True
    if t in (".false.", ".f.", "f", "false"):
        return False
    if re.fullmatch(r"[+-]?\d+", t):
        return int(t)
    num = re.sub(r"[dD]", "e", tok)
    if re.fullmatch(r"[+-]?(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?", num):
        return float(num)
    raise NamelistError("cannot read value '%s'" % tok)


def parse(text: str) -> "OrderedDict[str, OrderedDict]":
    """every group of the buffer: {group: {variable: value or [values]}} (a group that appears twice: the later
    assignments win, as successive reads would)"""
    body = "\n".join(_strip_comment(l) for l in text.splitlines())
    groups: "OrderedDict[str, OrderedDict]" = OrderedDict()
    pos = 0
    while True:
        m = re.compile(r"&\s*([A-Za-z_]\w*)").search(body, pos)
        if not m:
            break
        gname = m.group(1).lower()
        # the group ends at the first "/" (or &end) outside a character value
        i, quote, end = m.end(), None, None
        while i < len(body):
            ch = body[i]
            if quote:
                if ch == quote:
                    quote = None
            elif ch in "'\"":
                quote = ch
            elif ch == "/":
                end = i
                break
            elif body[i:i + 4].lower() == "&end":
                end = i
                break
            i += 1
        if end is None:
            raise NamelistError("group '%s' is not terminated by '/'" % gname)
        toks = [(k, v) for mm in _TOKEN.finditer(body[m.end():end]) for k, v in mm.groupdict().items() if v is not None]
        g = groups.setdefault(gname, OrderedDict())
        k = 0
        while k < len(toks):
            kind, val = toks[k]
            if kind == "comma":
                k += 1
                continue
            if kind != "word" or k + 1 >= len(toks) or toks[k + 1][0] != "eq":
                raise NamelistError("group '%s': expected 'name =' at '%s'" % (gname, val))
            name = val.lower()
            k += 2
            vals = []
            while k < len(toks):
                kind, val = toks[k]
                if kind == "comma":
                    k += 1
                    continue
                if kind == "word" and k + 1 < len(toks) and toks[k + 1][0] == "eq":
                    break          # the next assignment
                if kind == "str":
                    q = val[0]
                    vals.append(val[1:-1].replace(q + q, q))
                elif kind == "word":
                    vals.append(_value(val))
                else:
                    raise NamelistError("group '%s', variable '%s': unexpected '%s'" % (gname, name, val))
                k += 1
            if not vals:
                raise NamelistError("group '%s': variable '%s' has no value" % (gname, name))
            g[name] = vals[0] if len(vals) == 1 else vals
        pos = end + 1
    return groups


def read_group(groups, group: str, known: dict, required=False) -> dict:
    """what `read(buffer, nml=group)` does with the declared variables `known` (name -> default): unknown variables are an
    error (iostat /= 0 in Fortran), a missing group leaves the defaults (or is an error when `required`), value types must
    match the declaration"""
    out = dict(known)
    g = groups.get(group.lower())
    if g is None:
        if required:
            raise NamelistError('Unable to read "%s" in namelist file' % group)
        return out
    for name, val in g.items():
        if name not in known:
            raise NamelistError('group "%s": variable "%s" is not a member of the namelist' % (group, name))
        want = type(known[name])
        if want is float and isinstance(val, int) and not isinstance(val, bool):
            val = float(val)
        if not isinstance(val, want) or (want is int and isinstance(val, bool)):
            raise NamelistError('group "%s": variable "%s" expects %s, got %r' % (group, name, want.__name__, val))
        out[name] = val
    return out


def read_driver_namelist(text: str) -> dict:
    """the two groups standalone_mpp.F90 reads (:69-70; mpp_driver is required :110-114, regression_test optional :119)"""
    groups = parse(text)
    drv = read_group(groups, "mpp_driver", {"problem_type": ""}, required=True)
    reg = read_group(groups, "regression_test", {"write_regression_output": False, "num_cells": 0})
    return {"problem_type": drv["problem_type"].strip(), "write_regression_output": reg["write_regression_output"],
            "num_cells": reg["num_cells"], "groups": groups}
