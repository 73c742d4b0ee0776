mkdir -p gpurun_out/r4k
timeout 1200 python -m pytest tests/ -m gpu -q -x -k "team or weighted or needleman or smith or golden or straddle or reach or config3 or config4 or cost" 2>&1 | tail -6
python bench.py --steps 50 --warmup 5 --extra-configs 3,4,7,8 --no-cpu-baseline --extra-seconds 3 > gpurun_out/r4k/bench_teams.json 2> gpurun_out/r4k/bench_teams.err
python - <<'P'
import json
d = json.loads(open('gpurun_out/r4k/bench_teams.json').read().strip().splitlines()[-1])
for c in d['configs']:
    print(c['config'], c['value'], c['ms_per_step'], c['kernel_gcups'], c['results_checksum'])
P
