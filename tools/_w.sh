timeout -k 10 300 python -m pytest tests -q -m gpu -x -k "winograd" 2>&1 | tail -2
for i in 1 2 3; do timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', round(d['ms_per_step'],3), d['config']['loss'])"; done
for i in 1 2 3; do I2V_WINO_ROWS=0 timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench rows=0', round(d['ms_per_step'],3), d['config']['loss'])"; done
