python tools/step_breakdown.py gpurun_out/bd_q.json > gpurun_out/bd_q.txt 2>&1
tail -n 1 gpurun_out/bd_q.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-330
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
