#!/bin/bash
# round 3, run 5: the column march off the van Genuchten m = 1/2 specialisation -- residual, Jacobian and residual + Jacobian passes
# per constitutive law at 1024x1024x100 (tools/kbench.py --satfunc / --density / --per-cell-soils)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-r03_run5}; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 python3 tools/kbench.py --reps 10 --only residual assemble_both jacobian --out $O/$name.json "$@" > $O/$name.log 2>&1; echo "rc=$?"; grep -v "^\s*$" $O/$name.log | tail -4; }
run vg_tgdpb01
run vg_tgdpb01_percell --per-cell-soils
run vgm_tgdpb01 --satfunc vg_m
run bc_tgdpb01 --satfunc bc
run sbc2_tgdpb01 --satfunc sbc_bz2
run sbc3_tgdpb01 --satfunc sbc_bz3
run sbc3_tgdpb01_percell --satfunc sbc_bz3 --per-cell-soils
run sbc3_tgdpb01_4096cls --satfunc sbc_bz3 --soil-classes 4096
run sbc3_ifc67 --satfunc sbc_bz3 --density ifc67
run vg_ifc67 --density ifc67
run mixed_tgdpb01 --satfunc mixed
