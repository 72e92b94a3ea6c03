#!/bin/bash
# round 5, run 13: coupled TH, the back sweeps' factor G with a cell's four entries side by side (one 2 KB piece a wave and level instead
# of four planes), and the block-Thomas kernels asked for four waves per SIMD on top; library of the commit before / AoS / AoS + 4 waves
source "$(dirname "$0")/_common.sh"
cp mpp_amd/_build/libmppgpu.so mpp_amd/_build/libmppgpu_new.so
for v in old new aos4 old new aos4; do
  cp mpp_amd/_build/libmppgpu_$v.so mpp_amd/_build/libmppgpu.so
  timeout -k 10 300 python3 tools/th_bench.py --nx 512 --ny 512 --nz 80 --ksp bcgs --pc bjacobi --cpu-cols 0 > $O/th_$v.log 2>&1 || { tail -5 $O/th_$v.log; cp mpp_amd/_build/libmppgpu_new.so mpp_amd/_build/libmppgpu.so; exit 0; }
  tail -1 $O/th_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('th $v', round(d['newton_its_per_s'],1), {k:round(v['avg_ms'],4) for k,v in d['kernels'].items()})"
done
cp mpp_amd/_build/libmppgpu_new.so mpp_amd/_build/libmppgpu.so
timeout -k 10 600 python3 -m pytest tests/test_gpu_th.py -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
exit 0
