#!/bin/bash
# round 3, run 9: 16-bit soil class ids, halved smoothed-Brooks-Corey coefficient stream: tests, then the laws again
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run9; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest -m gpu (vsfm)"; timeout -k 10 1100 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_fused.py tests/test_gpu_more.py tests/test_gpu_r02.py -q -m gpu > $O/pytest.log 2>&1; echo "rc=$?"; tail -8 $O/pytest.log
run() { name=$1; shift; echo "== $name"; timeout -k 10 300 python3 tools/kbench.py --reps 10 --only residual assemble_both jacobian --out $O/$name.json "$@" > $O/$name.log 2>&1; echo "rc=$?"; grep -v "^\s*$" $O/$name.log | tail -3; }
run vg_tgdpb01
run vg_tgdpb01_percell --per-cell-soils
run sbc3_tgdpb01 --satfunc sbc_bz3
run sbc3_tgdpb01_percell --satfunc sbc_bz3 --per-cell-soils
run sbc3_tgdpb01_4096cls --satfunc sbc_bz3 --soil-classes 4096
run sbc3_tgdpb01_60000cls --satfunc sbc_bz3 --soil-classes 60000
run vg_tgdpb01_4096cls --soil-classes 4096
