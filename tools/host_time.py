import contextlib, io, sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
from pase_amd.trainer import trainer
dev = torch.device("cuda", 0)
fe_cfg, wk_cfg, raw = bench.load_cfgs()
torch.manual_seed(2)
with contextlib.redirect_stdout(io.StringIO()):
    tr = trainer(frontend_cfg=dict(fe_cfg), minions_cfg=wk_cfg, cfg=dict(epoch=1, bpe=100), lr_mode="poly", device=dev)
batch = bench.synthetic_batch(1234, 32, 32000, raw, dev)
for _ in range(3):
    tr.train_step(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    tr.train_step(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host enqueue ms/step %.2f; total ms/step %.2f" % ((t1 - t0) / 10 * 1e3, (t2 - t0) / 10 * 1e3))
