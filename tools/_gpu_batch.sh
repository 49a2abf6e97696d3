cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python tools/step_breakdown.py gpurun_out/gl_i.json > gpurun_out/gl_i.txt 2>&1; tail -1 gpurun_out/gl_i.txt
PASE_LIB=$PWD/tools/_ab/libpase_prev.so timeout 300 python tools/step_breakdown.py gpurun_out/gl_iprev.json > gpurun_out/gl_iprev.txt 2>&1; tail -1 gpurun_out/gl_iprev.txt
timeout 300 python tools/step_breakdown.py gpurun_out/gl_i2.json > gpurun_out/gl_i2.txt 2>&1; tail -1 gpurun_out/gl_i2.txt
PASE_LIB=$PWD/tools/_ab/libpase_prev.so timeout 300 python tools/step_breakdown.py gpurun_out/gl_iprev2.json > gpurun_out/gl_iprev2.txt 2>&1; tail -1 gpurun_out/gl_iprev2.txt
