#!/bin/bash
# round 2, GPU run 33: rocprofv3 kernel stats of the "next" rows on the end-of-round code: thermal 256^3 (GMRES + ILU(0) by boxes),
# coupled TH 512x512x80 (BiCGStab + block-Jacobi), 3-D VSFM 128x128x64 (GMRES + ILU(0) by boxes)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02_run33; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
cd /tmp
echo "== thermal 256"; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/thermal -o t -- python3 $GRAFT_REPO_ROOT/tools/thermal_bench.py --n 256 --cpu-n 0 > $O/thermal.log 2>&1; echo "rc=$?"
echo "== th"; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/th -o t -- python3 $GRAFT_REPO_ROOT/tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 > $O/th.log 2>&1; echo "rc=$?"
echo "== vsfm3d"; timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/v3d -o t -- python3 $GRAFT_REPO_ROOT/tools/vsfm3d_bench.py --nx 128 --ny 128 --nz 64 --ksp gmres --pc ilu > $O/v3d.log 2>&1; echo "rc=$?"
cd $O; for d in thermal th v3d; do f=$(find $d -name "*kernel_stats.csv" | head -1); cp "$f" ${d}_kernel_stats.csv; rm -rf $d; head -6 ${d}_kernel_stats.csv | cut -c1-160; done
