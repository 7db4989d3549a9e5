#!/bin/bash
# whole-step A/B of tuning knobs on configs[2]: tools/knob_sweep_isd.sh "VAR=val" ...   (each argument is one run; A=1 = baseline)
run() { echo "== $*"; env "$@" timeout -k 10 300 python3 bench.py --config instance_styled --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"; }
for a in "$@"; do run $a; done
