python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for c in c3 c2 c4 c5; do python bench.py --quick --config $c --steps 3 --warmup 1 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$c', round(j['value']), round(j['ms_per_step'],1), {k: round(v,1) for k,v in j['roofline']['kernel_ms_per_step'].items()})"; done
bash tools/r3_shape_stats.sh "c4 c5" | grep "launch order\|=="
