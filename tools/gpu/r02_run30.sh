#!/bin/bash
# round 2, GPU run 30: four waves per box (16 x 16 lines, inner faces through LDS, a barrier per four steps): parity, thermal and 3-D VSFM timings
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run30; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest box tests"; timeout -k 10 300 python3 -m pytest tests/test_gpu_r02.py -q -m gpu -k "box" > $O/pytest_box.log 2>&1; rc=$?; echo "rc=$rc"; tail -12 $O/pytest_box.log | cut -c1-300; [ $rc -eq 0 ] || exit 1
for n in 128 256; do for w in 2 1; do echo "== thermal $n waves=$w"; timeout -k 10 600 python3 tools/thermal_bench.py --n $n --cpu-n 0 --opt ilu_box_waves=$w --out $O/therm_${n}_w$w.json > $O/therm_${n}_w$w.log 2>&1; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/therm_${n}_w$w.json'));k=j['kernels'];print(j['step_s'],j['linear_iterations'],{m:(v['launches'],round(v['total_ms']/v['launches'],3)) for m,v in k.items() if m in ('pc_apply','pc_setup')}, j['roofline']['kernel'])"; done; done
for w in 2 1; do echo "== vsfm3d gmres ilu waves=$w"; timeout -k 10 600 python3 tools/vsfm3d_bench.py --nx 128 --ny 128 --nz 64 --ksp gmres --pc ilu --opt ilu_box_waves=$w --out $O/v3d_w$w.json > $O/v3d_w$w.log 2>&1; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/v3d_w$w.json'));print(j['newton_its_per_s'],{m:(v['launches'],round(v['avg_ms'],3)) for m,v in j['kernels'].items() if m in ('pc_apply','pc_setup')})"; done
echo "== thermal 512 waves=2"; timeout -k 10 900 python3 tools/thermal_bench.py --n 512 --cpu-n 0 --opt ilu_box_waves=2 --out $O/therm_512_w2.json > $O/therm_512_w2.log 2>&1; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/therm_512_w2.json'));k=j['kernels'];print(j['step_s'],j['linear_iterations'],{m:(v['launches'],round(v['total_ms']/v['launches'],3)) for m,v in k.items() if m in ('pc_apply','pc_setup')})"
