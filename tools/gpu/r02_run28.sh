#!/bin/bash
# round 2, GPU run 28: box length along the fastest direction at 512^3 (padding against box time): 96 and 128 against the default 64
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02_run28; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
for len in 128 96; do echo "== thermal 512 ilu_box_len=$len"; timeout -k 10 900 python3 tools/thermal_bench.py --n 512 --cpu-n 0 --opt ilu_box_len=$len --out $O/therm_512_len$len.json > $O/therm_512_len$len.log 2>&1; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/therm_512_len$len.json'));k=j['kernels'];print(j['step_s'],j['linear_iterations'],{m:(v['launches'],round(v['total_ms']/v['launches'],3)) for m,v in k.items()})"; done
