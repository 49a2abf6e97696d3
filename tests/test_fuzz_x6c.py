"""A fixed-seed slice of tools/fuzz_x6c.py in the GPU suite: random shapes (ragged channels / rows / columns, 1-5 sequences,
strides 1 / 2 / 4, forced pre-split / symmetric / DUO forms, persistent grids capped at 1-7 workgroups) through the split-bf16
convolution, context-MSE and weight-gradient kernels against fp64.  The full sweep (thousands of cases, several seeds) is the
tool; its round-6 results are in profiles/fuzz_r06.txt."""
import importlib.util
import os
import random

import pytest
import torch


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [101, 102, 103])
def test_random_shapes_against_fp64(seed):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from pase_amd import _lib
    from pase_amd import kernels as K
    _lib.use_library(None, "cuda")
    _lib.lib()
    saved_env, saved_x6 = dict(os.environ), K.X6
    try:
        spec = importlib.util.spec_from_file_location(
            "fuzz_x6c", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "fuzz_x6c.py"))
        fz = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(fz)
        rng = random.Random(seed)
        torch.manual_seed(seed)
        worst = []
        for i in range(45):
            e, tag = (fz.one_conv, fz.one_wgrad, fz.one_mse)[i % 3](rng)
            if not (e < 1e-6):
                worst.append((e, tag))
        assert not worst, worst
    finally:
        os.environ.clear()
        os.environ.update(saved_env)
        K.X6 = saved_x6
