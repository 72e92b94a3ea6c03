#!/bin/bash
# round 5, run 7: the driver's own bench command on the final sources (complete-row march back at 3 waves per SIMD), then twice GPU only
source "$(dirname "$0")/_common.sh"
build_all
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_form_steps20_warmup5.json 2> $O/bench_driver_form.err; echo "rc=$?"
for r in 1 2; do timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-verify --no-extra-configs --steps 20 --warmup 5 > $O/bench_gpu_only_$r.json 2> $O/bench_gpu_only_$r.err; echo "rc=$?"; done
python3 - <<PY
import json
for f in ("bench_driver_form_steps20_warmup5","bench_gpu_only_1","bench_gpu_only_2"):
    d=json.load(open("$O/%s.json"%f)); print(f, round(d["value"],1), d["newton_iterations"], d["roofline"]["frac"], {k:round(v["avg_ms"],3) for k,v in d["kernels"].items()})
PY
exit 0
