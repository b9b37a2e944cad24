run() { printf "%-44s " "$*"; env "$@" timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json
try:
    d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'])
except Exception as e: print('fail', e)"; }
for a in "$@"; do run $(echo $a | tr ',' ' '); done
