cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
( echo BASE; timeout 300 python tools/trace_x6c.py blk5 blk3 2>&1 | grep "workgroup 0\|==" | cut -c1-130
for v in noa nob noab; do echo "VARIANT $v"; PASE_TRACE_FLAGS=x PASE_TRACE_TAG=_$v timeout 300 python tools/trace_x6c.py blk5 blk3 2>&1 | grep "workgroup 0\|==" | cut -c1-130; done ) > gpurun_out/trace_ablate.txt 2>&1
