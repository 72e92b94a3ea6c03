#!/bin/bash
# round 2, evidence B for the end-of-round code: the "next" rows -- thermal (config 4 size, with the error each tolerance buys),
# coupled TH (config 5 size), 3-D VSFM with both solver choices.  Output under gpurun_out/r02_evidence_b/.
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_evidence_b; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== thermal 256 sweep"; timeout -k 10 600 python3 tools/thermal_bench.py --n 256 --cpu-n 64 --rtol-sweep 1e-5,1e-7,1e-9 --out $O/thermal_256_rtol_sweep.json > $O/thermal_256.log 2>&1; echo "rc=$?"; tail -1 $O/thermal_256.log | cut -c1-300
echo "== thermal 512 sweep"; timeout -k 10 1000 python3 tools/thermal_bench.py --n 512 --cpu-n 0 --rtol-sweep 1e-5,1e-8 --out $O/thermal_512_rtol_sweep.json > $O/thermal_512.log 2>&1; echo "rc=$?"; tail -1 $O/thermal_512.log | cut -c1-300
echo "== th 512x512x80"; timeout -k 10 600 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 --out $O/th_512x512x80_bcgs_bjacobi.json > $O/th.log 2>&1; echo "rc=$?"; tail -1 $O/th.log | cut -c1-300
echo "== vsfm3d bjacobi"; timeout -k 10 600 python3 tools/vsfm3d_bench.py --nx 256 --ny 256 --nz 64 --out $O/vsfm3d_256x256x64_bcgs_bjacobi.json > $O/v3d_bj.log 2>&1; echo "rc=$?"; tail -1 $O/v3d_bj.log | cut -c1-300
echo "== vsfm3d gmres ilu"; timeout -k 10 600 python3 tools/vsfm3d_bench.py --nx 128 --ny 128 --nz 64 --ksp gmres --pc ilu --out $O/vsfm3d_128x128x64_gmres_ilu.json > $O/v3d_ilu.log 2>&1; echo "rc=$?"; tail -1 $O/v3d_ilu.log | cut -c1-300
echo "== vsfm3d gmres ilu, level-scheduled"; timeout -k 10 600 python3 tools/vsfm3d_bench.py --nx 128 --ny 128 --nz 64 --ksp gmres --pc ilu --opt ilu_box=0 --out $O/vsfm3d_128x128x64_gmres_ilu_levels.json > $O/v3d_ilu0.log 2>&1; echo "rc=$?"; tail -1 $O/v3d_ilu0.log | cut -c1-300
ls $O
