"""full-size gradient errors of both pipes and of the torch-fp32 comparator against a torch-fp64 evaluation on the same GPU"""
import sys, random, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import test_bench_config as TB
from pase_amd import kernels as K
variant = sys.argv[1] if len(sys.argv) > 1 else "pase"
res = {}
for x6 in (True, False):
    K.X6 = x6
    st = TB._make_setup(variant)
    tr, P, fe, raw, dev = st["tr"], st["P"], st["fe"], st["raw"], st["dev"]
    batch = TB._batch(4321, raw, dev, st["B"], st["T"])
    m = tr.model; m.train()
    for opt in tr.optimizers(): opt.zero_grad()
    random.seed(77)
    m.loss_and_grads(batch)
    res[x6] = {n: p.grad.detach().double().clone() for n, p in m.named_parameters()}
    if x6:
        lo, _ = TB._oracle_step(P, fe, raw, batch, seed=77)
        ref32 = {n: P[n].grad.detach().double().clone() for n in st["names"]}
        P64 = {k: (v.detach().double() if v.is_floating_point() else v.detach().clone()) for k, v in P.items()}
        for n in st["names"]: P64[n].requires_grad_(True)
        b64 = {k: v.double() for k, v in batch.items()}
        t0 = time.time()
        lo64, _ = TB._oracle_step(P64, fe, raw, b64, seed=77)
        torch.cuda.synchronize(); print("fp64 oracle step %.1f s" % (time.time() - t0))
        ref64 = {n: P64[n].grad.detach().clone() for n in st["names"]}
        print({k: (lo[k], lo64[k]) for k in list(lo)[:4]})
    del st, tr, m
def rel(a, b): return float((a - b).norm() / b.norm().clamp_min(1e-300))
print("%-40s %10s %10s %10s" % ("tensor", "x6", "fp32pipe", "torch32"))
for n in ref64:
    if TB._noise(variant, n): continue
    print("%-40s %10.2e %10.2e %10.2e" % (n[-40:], rel(res[True][n], ref64[n]), rel(res[False][n], ref64[n]), rel(ref32[n], ref64[n])))
