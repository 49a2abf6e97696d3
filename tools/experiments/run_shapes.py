"""Run a few PASE+ bs32 launch shapes of tools/trace_x6c.py on the PRODUCT library (for rocprofv3 --pmc passes):
python tools/experiments/run_shapes.py blk5 qrnn lps ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import trace_x6c as T  # noqa: E402

dev = torch.device("cuda:0")
for name in sys.argv[1:]:
    fn = T.run_shape(name, dev)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
