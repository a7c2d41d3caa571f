cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_sams_gpu.py -x -q -k "full_size_generator" 2>&1 | tail -2
timeout 400 python bench.py > gpurun_out/r03_u_bench_c4.json 2> gpurun_out/r03_u_bench_c4.log; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_u_bench_c4.json').read().strip().splitlines()[-1])
r=d['roofline']
print(d['value'], d['ms_per_step'], r['kernel'], r['achieved'], r['frac'], r['executed_frac'], r['traffic'], r['step_frac'], d['cpu_baseline']['value'])
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
