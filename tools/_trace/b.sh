python tools/step_breakdown.py gpurun_out/bd_r.json > gpurun_out/bd_r.txt 2>&1
tail -n 1 gpurun_out/bd_r.txt
