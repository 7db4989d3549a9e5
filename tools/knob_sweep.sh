#!/bin/bash
# whole-step A/B of tuning knobs: tools/knob_sweep.sh "VAR=val" "VAR2=val" ...   (each argument is one run; A=1 = baseline)
run() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-also 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for a in "$@"; do run $a; done
