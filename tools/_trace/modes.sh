timeout 300 python -m pytest tests/test_wgrad_x6c.py -m gpu -q 2>&1 | tail -3
PASE_X6C_WGRAD_MODE=3 python tools/trace_x6c.py wg5 wg7 2>&1 | grep -v amdgpu.ids
PASE_X6C_WGRAD_FLAT=1 python tools/trace_x6c.py wglps wgqrnn 2>&1 | grep -v amdgpu.ids
PASE_X6C_WGRAD_MODE=3 PASE_X6C_WGRAD_FLAT=1 python tools/step_breakdown.py gpurun_out/bd_m3.json > gpurun_out/bd_m3.txt 2>&1
tail -n 1 gpurun_out/bd_m3.txt
