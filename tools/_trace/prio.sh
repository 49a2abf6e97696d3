for pr in 0 1 3 5 12; do
  echo "=== PASE_X6C_PRIO=$pr"
  PASE_X6C_PRIO=$pr python tools/trace_x6c.py blk5 blk7 lps wg5 2>&1 | grep -v amdgpu.ids | grep -v "workgroup 131"
done
