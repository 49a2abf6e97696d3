for cfg in "PASE_X6_CONV=0" "PASE_X6_WGRAD=0" "PASE_X6_WGRAD=0 PASE_X6_ONLY=fwd" "PASE_X6_WGRAD=0 PASE_X6_ONLY=bwd"; do
  echo "=== $cfg"
  env $cfg python -m pytest tests/test_pase_step.py -k "full_width_golden and pase_plus and x6" -m gpu -q -s 2>&1 | grep -A8 "gradients vs the fp64" | head -10
done
