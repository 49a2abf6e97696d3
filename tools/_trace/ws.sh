python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d 2>&1 | tail -1 | cut -c1-260
PASE_WGRAD_STREAM=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-h2d 2>&1 | tail -1 | cut -c1-260
timeout 900 python -m pytest tests/test_pase_step.py tests/test_encoder.py tests/test_ddp_gloo.py tests/test_dsp.py -m gpu -q -x 2>&1 | tail -3
timeout 900 python -m pytest tests/test_bench_config.py -m gpu -q -x -k "x6 and not cfg1 and not emb256" 2>&1 | tail -3
