#!/bin/bash
# round 5, run 15: coupled TH, the combined factorisation + first solve with its level's rows and right-hand side requested one level
# ahead (as k_th_spmv_tri), without the slot-table checks of the stand-alone kernel, asked for four waves per SIMD; library of the commit
# before / this one, three times each on one box; then the TH tests
source "$(dirname "$0")/_common.sh"
cp mpp_amd/_build/libmppgpu.so mpp_amd/_build/libmppgpu_new.so
for v in old new old new old new; do
  cp mpp_amd/_build/libmppgpu_$v.so mpp_amd/_build/libmppgpu.so
  timeout -k 10 300 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 > $O/th_$v.log 2>&1 || { tail -5 $O/th_$v.log; cp mpp_amd/_build/libmppgpu_new.so mpp_amd/_build/libmppgpu.so; exit 0; }
  tail -1 $O/th_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('th $v', round(d['newton_its_per_s'],1), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done
cp mpp_amd/_build/libmppgpu_new.so mpp_amd/_build/libmppgpu.so
timeout -k 10 600 python3 -m pytest tests/test_gpu_th.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
exit 0
