#!/bin/bash
# round 2, GPU run 13: box-wavefront ILU(0) solves (parity with the level-scheduled ones; thermal 128^3 / 256^3 timings)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run13; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
echo "== pytest box ilu + thermal"; timeout -k 10 900 python3 -m pytest tests/test_gpu_r02.py tests/test_gpu_thermal.py -q -m gpu -k "box or ilu or thermal or vchannel" > $O/pytest.log 2>&1; echo "rc=$?"; tail -15 $O/pytest.log
for n in 128 256; do
  for box in 0 1; do
    echo "== thermal $n ilu_box=$box"; timeout -k 10 600 python3 tools/thermal_bench.py --n $n --cpu-n 0 --opt ilu_box=$box --out $O/therm_${n}_box$box.json > $O/therm_${n}_box$box.log 2>&1; echo "rc=$?"; tail -3 $O/therm_${n}_box$box.log | cut -c1-300
  done
done
for len in 16 64; do
  echo "== thermal 256 ilu_box_len=$len"; timeout -k 10 600 python3 tools/thermal_bench.py --n 256 --cpu-n 0 --opt ilu_box_len=$len --out $O/therm_256_len$len.json > $O/therm_256_len$len.log 2>&1; echo "rc=$?"; tail -2 $O/therm_256_len$len.log | cut -c1-300
done
