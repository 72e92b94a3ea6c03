#!/bin/bash
# round 3, run 2: config 2 with the complete-row march forced (nseg=1) against the default (nseg=2: separate factor / norms / trial point)
set -o pipefail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03_run2; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.build()" > $O/build.log 2>&1 || { tail -20 $O/build.log; exit 1; }
B="--no-cpu-baseline --no-verify --steps 8 --warmup 2 --profile spmv"
for g in "256 256 50" "128 128 50" "512 256 50" "256 256 100"; do set -- $g
for o in "nseg=0" "nseg=1"; do
echo "== $g $o"; timeout -k 10 300 python3 bench.py --nx $1 --ny $2 --nz $3 $B --opt $o > $O/b_$1x$2x$3_$o.json 2>> $O/err.log; echo "rc=$?"; python3 -c "import json;j=json.load(open('$O/b_$1x$2x$3_$o.json'));print(j['value'],j['newton_iterations'])"
done; done
