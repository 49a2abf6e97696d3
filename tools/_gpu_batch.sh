cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 300 python tools/step_breakdown.py gpurun_out/gl_a.json > gpurun_out/gl_a.txt 2>&1; tail -1 gpurun_out/gl_a.txt
PASE_X6C_XP=1 timeout 300 python tools/step_breakdown.py gpurun_out/gl_xpall.json > gpurun_out/gl_xpall.txt 2>&1; tail -1 gpurun_out/gl_xpall.txt
PASE_X6C_XP=1 PASE_X6C_FORCE=1 timeout 300 python tools/step_breakdown.py gpurun_out/gl_xpforce.json > gpurun_out/gl_xpforce.txt 2>&1; tail -1 gpurun_out/gl_xpforce.txt
