#!/bin/bash
# round 5, run 1: the whole-column partition on the GPU (split tests back on GMRES(30) + ILU(0)), and the 7-point VSFM grid through
# bench.py --workload vsfm3d on the round-4 kernels (general assembly) as the starting point of the 3-D column work
source "$(dirname "$0")/_common.sh"
build_all
echo "== pytest split tests"; timeout -k 10 900 python3 -m pytest tests/test_gpu_more.py tests/test_gpu_th.py tests/test_gpu_fuzz.py tests/test_gpu_r02.py -q -m gpu -k "split or ranks" > $O/pytest_split.log 2>&1; rc=$?; echo "rc=$rc"; tail -15 $O/pytest_split.log
[ $rc -ne 0 ] && keep_failure $O/pytest_split.log
for dims in "256 256 64" "512 512 100"; do
  set -- $dims
  echo "== vsfm3d $1 x $2 x $3"
  timeout -k 10 500 python3 bench.py --workload vsfm3d --nx $1 --ny $2 --nz $3 --steps 2 --warmup 1 > $O/vsfm3d_$1x$2x$3.json 2> $O/vsfm3d_$1x$2x$3.err; echo "rc=$?"
  python3 -c "import json; d=json.load(open('$O/vsfm3d_$1x$2x$3.json')); print(d['value'], d['newton_iterations'], d['linear_iterations'], d['setup_s'], d['answer_check'], d.get('roofline',{}).get('frac'), {k:(round(v['avg_ms'],4), v['launches'], v.get('GBps_algorithmic')) for k,v in d['kernels'].items()})" || tail -5 $O/vsfm3d_$1x$2x$3.err
done
echo "== rocprofv3 kernel stats, vsfm3d 512x512x100"
timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/prof -o v3d --output-format csv -- python3 bench.py --workload vsfm3d --nx 512 --ny 512 --nz 100 --steps 2 --warmup 1 --opt ilu_graph=0 > $O/prof.log 2>&1; echo "rc=$?"
rm -f $O/prof/*kernel_trace.csv $O/prof/*/*kernel_trace.csv
find $O/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} sh -c 'cut -d, -f1-8 {} | head -25'
exit 0
