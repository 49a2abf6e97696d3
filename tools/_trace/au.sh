timeout 600 python -m pytest tests/test_conv_x6c.py tests/test_wgrad_x6c.py tests/test_bs32_launches.py tests/test_conv_gemm.py -m gpu -q -x 2>&1 | tail -3
python tools/step_breakdown.py gpurun_out/bd_asm.json > gpurun_out/bd_asm.txt 2>&1; tail -n 1 gpurun_out/bd_asm.txt
python bench.py --no-cpu-baseline --no-h2d 2>/dev/null | tail -1 | cut -c1-230
