PASE_X6=0 python tools/step_breakdown.py gpurun_out/bd_fp32.json > gpurun_out/bd_fp32.txt 2>&1
PASE_X6C_FORCE=1 PASE_X6C_WGRAD_FLAT=1 python tools/step_breakdown.py gpurun_out/bd_force.json > gpurun_out/bd_force.txt 2>&1
python tools/step_breakdown.py gpurun_out/bd_cur.json > gpurun_out/bd_cur.txt 2>&1
tail -1 gpurun_out/bd_fp32.txt gpurun_out/bd_force.txt gpurun_out/bd_cur.txt
