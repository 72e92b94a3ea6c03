#!/bin/bash
# round 3: HBM bytes moved by the column march for every constitutive law: for each law one untimed kbench
# run for the times, then the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, nothing else traced) of the same command.
# Output: gpurun_out/r03_laws/<law>.json (times) and <law>_pmc.json (bytes per launch per kernel).
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_laws; mkdir -p $O
run() {   # name, kbench flags...
  local name=$1; shift
  echo "== $name"
  timeout -k 10 300 python3 tools/kbench.py --only copy residual assemble_both --reps 10 "$@" --out $O/$name.json > $O/$name.log 2>&1 || { echo "kbench failed"; tail -5 $O/$name.log; return 1; }
  timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE -d $O/f_$name -o f --output-format csv -- python3 tools/kbench.py --only copy residual assemble_both --reps 2 "$@" > $O/${name}_f.log 2>&1 || { echo "fetch pass failed"; tail -5 $O/${name}_f.log; return 1; }
  timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE -d $O/w_$name -o w --output-format csv -- python3 tools/kbench.py --only copy residual assemble_both --reps 2 "$@" > $O/${name}_w.log 2>&1 || { echo "write pass failed"; tail -5 $O/${name}_w.log; return 1; }
  python3 tools/pmc_summary.py $O/f_$name $O/w_$name k_assemble_cols k_copy > $O/${name}_pmc.json 2> $O/${name}_pmc.err
  rm -rf $O/f_$name $O/w_$name
  tail -3 $O/$name.log | cut -c1-200
}
run vg_tgdpb01 --satfunc vg &&
run vg_tgdpb01_percell --satfunc vg --per-cell-soils &&
run vgm_tgdpb01 --satfunc vg_m &&
run bc_tgdpb01 --satfunc bc &&
run sbc2_tgdpb01 --satfunc sbc_bz2 &&
run sbc3_tgdpb01 --satfunc sbc_bz3 &&
run sbc3_tgdpb01_percell --satfunc sbc_bz3 --per-cell-soils &&
run sbc3_ifc67 --satfunc sbc_bz3 --density ifc67 &&
run vg_ifc67 --satfunc vg --density ifc67 &&
run mixed_tgdpb01 --satfunc mixed
ls $O | head -40
