cd /root/repo
python -m pytest tests/test_ops_gpu.py -x -q -k "norm" 2>&1 | tail -2
for c in c4 c2; do
python bench.py --config $c --steps 40 --warmup 5 --no-cpu-baseline --no-hbm-table 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', round(d['value'],1), round(d['ms_per_step'],3))"
done
