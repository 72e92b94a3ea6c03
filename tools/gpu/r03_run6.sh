#!/bin/bash
# round 3, run 6: after the algebraic reduction of real powers (Brooks-Corey kr, van Genuchten derivative terms, IFC-67 temperature terms on the
# host): the whole GPU suite, then the march per constitutive law again (same plan as run 5)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run6; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest -m gpu"; timeout -k 10 1100 python3 -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "rc=$?"; tail -8 $O/pytest.log
