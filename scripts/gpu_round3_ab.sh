# One box, one call: the bench line with the feed-forward as (0) three kernels, (1) LayerNorm+GELU in ffn.0's epilogue, (3, default) one kernel;
# repeated once in reverse order (the first run of a call also warms the box).
cd $GRAFT_REPO_ROOT
for v in 3 0 1 3 1 0; do
  python bench.py --no-cpu-baseline --main-region-only --tune 11=$v 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(json.dumps({'tune11': $v, 'pairs_per_s': round(d['value'],1), 'ms_per_step': round(d['ms_per_step'],2), 'clock_mhz': round(d.get('sustained_clock_mhz',0))}))"
done
