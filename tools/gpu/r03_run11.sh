#!/bin/bash
# round 3, run 11: coupled TH at config-5 size (512x512x80, BiCGStab + block-Jacobi) with fpow and the Kersten-number quotient;
# smoothed Brooks-Corey with one pow call site; the VSFM bench lines (full grid, config 2, one eighth) on the current code
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run11; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== th bench 512x512x80"; timeout -k 10 600 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --out $O/th_512x512x80.json > $O/th.log 2>&1; echo "rc=$?"; tail -1 $O/th.log | cut -c1-900
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 python3 tools/kbench.py --reps 10 --only residual assemble_both jacobian --out $O/$name.json "$@" > $O/$name.log 2>&1; echo "rc=$?"; grep -v "^\s*$" $O/$name.log | tail -3; }
run sbc3_tgdpb01 --satfunc sbc_bz3
run sbc3_tgdpb01_percell --satfunc sbc_bz3 --per-cell-soils
B="--no-cpu-baseline --no-verify --steps 8 --warmup 2"
echo "== c2"; timeout -k 10 300 python3 bench.py --nx 256 --ny 256 --nz 50 $B --profile spmv > $O/c2.json 2>> $O/err.log; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/c2.json'));print(j['value'],j['newton_iterations'],j['host_readbacks_per_newton_iteration'])"
echo "== s8"; timeout -k 10 300 python3 bench.py --nx 1024 --ny 128 $B --profile spmv > $O/s8.json 2>> $O/err.log; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/s8.json'));print(j['value'],j['newton_iterations'],j['host_readbacks_per_newton_iteration'])"
echo "== full"; timeout -k 10 400 python3 bench.py --no-cpu-baseline --steps 4 --warmup 1 > $O/full.json 2>> $O/err.log; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/full.json'));print(j['value'],j['newton_iterations'],j['setup_s'],j['host_readbacks_per_newton_iteration'],j['answer_check']['max_rel_err'])"
tail -3 $O/err.log
