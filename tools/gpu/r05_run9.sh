#!/bin/bash
# round 5, run 9: occupancy hints (amdgpu_waves_per_eu) on three kernels that sit just above a register boundary -- the residual-only march
# of the benchmark's soil (102 -> 96 registers, 5 waves, 32 B of scratch), the windowed SpMV (103 -> 96, 5 waves, 36 B of scratch), the
# coupled system's block-Thomas solves (130 -> 128, 4 waves, no scratch): library built without / with the hints, one box
source "$(dirname "$0")/_common.sh"
for v in old new old new; do
  if [ $v = old ]; then F="-DMPP_RESIDUAL_MARCH_WAVES=1 -DMPP_SPMV_MB_WAVES=1 -DMPP_TH_SOLVE_WAVES=1"; else F=""; fi
  rm -f mpp_amd/_build/kernels.o mpp_amd/_build/th_kernels.o mpp_amd/_build/libmppgpu.so
  make -C mpp_amd/csrc CXXFLAGS="-O3 -std=c++17 -fPIC -Wall -Wno-unused-result $F" > $O/build_$v.log 2>&1 || { tail -5 $O/build_$v.log; exit 0; }
  echo "== $v"
  timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-verify --no-extra-configs --steps 20 --warmup 5 > $O/bench_$v.json 2>> $O/err.log
  python3 -c "import json; d=json.load(open('$O/bench_$v.json')); print('headline', round(d['value'],1), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
  timeout -k 10 400 python3 bench.py --workload vsfm3d --nx 512 --ny 512 --nz 100 --steps 2 --warmup 1 > $O/v3d_$v.json 2>> $O/err.log
  python3 -c "import json; d=json.load(open('$O/v3d_$v.json')); print('vsfm3d', round(d['value'],2), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
  timeout -k 10 400 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 > $O/th_$v.log 2>&1
  tail -1 $O/th_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('th', round(d['newton_its_per_s'],1), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done
exit 0
