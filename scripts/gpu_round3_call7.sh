cd $GRAFT_REPO_ROOT
python scripts/gpu_ffn_ln_check.py 2>&1 | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print({k:(v['bad_rows'],v['bit_equal_rerun'],v['max_err']) for k,v in d.items()})"
for t in 0 1 0 1; do python bench.py --no-cpu-baseline --main-region-only --tune 11=$t 2>/dev/null | python -c "import json,sys;d=json.loads(sys.stdin.read());print('ln=$t', round(d['value'],1), round(d['ms_per_step'],2), round(d['sustained_clock_mhz']))"; done
timeout 900 python -m pytest tests/test_tile_matching_gpu.py tests/test_lightglue_gpu.py -m gpu -q -x 2>&1 | tail -4
