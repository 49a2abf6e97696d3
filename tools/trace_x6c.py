"""Where does a conv_x6c workgroup spend its time?  Builds pase_amd/csrc with -DPASE_X6C_TRACE into a side library
(gpurun_out/libpase_trace.so; the product library is untouched), runs a few PASE+ bs32 launch shapes and prints, per
shape, the per-item phase durations (shader clock cycles) of workgroup 0 and workgroup 131:
  compute wave 0:  wait = item start -> first stage visible;  mfma = main loop;  epi = epilogue
  staging wave 4:  pro = next item's prologue (position setup, loads, first stage);  loop = stage loop
usage (GPU box):  python tools/trace_x6c.py [shape ...]
"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from pase_amd import _lib, build  # noqa: E402
from pase_amd import engine as E  # noqa: E402
from pase_amd import kernels as K  # noqa: E402
from pase_amd.engine import Act  # noqa: E402


def build_trace_lib():
    # built in the CPU container (hipcc cross-compiles) so that no GPU minutes go into compiling: `python tools/trace_x6c.py build`
    # PASE_TRACE_FLAGS: extra -D flags of an A/B variant (e.g. -DPASE_ABL_NOA), PASE_TRACE_TAG: its file name suffix
    extra = os.environ.get("PASE_TRACE_FLAGS", "").split()
    out = os.path.join(ROOT, "tools", "_trace", "libpase_trace%s.so" % os.environ.get("PASE_TRACE_TAG", ""))
    stamp = out + ".digest"
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if os.path.exists(out) and (extra == ["x"] or (os.path.exists(stamp) and open(stamp).read() == build.hip_digest() + " ".join(extra))):
        return out
    srcs = build._sources()
    flags = [f for f in build._hip_flags()] + ["-DPASE_X6C_TRACE"] + extra
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + flags + ["-o", out] + srcs)
    open(stamp, "w").write(build.hip_digest() + " ".join(extra))
    return out


def run_shape(name, dev):
    S = 96
    if name in ("blk1", "blk5", "blk7", "blk3", "blk6"):
        Cin, Cout, k, st, Tin = {"blk1": (64, 64, 20, 10, 32000), "blk3": (128, 128, 11, 1, 1600), "blk5": (256, 256, 11, 1, 800),
                                 "blk6": (256, 512, 11, 2, 800), "blk7": (512, 512, 11, 2, 400)}[name]
        pL, pR = E.reflect_pads(k, st)
        x = torch.randn(S, Cin, Tin, device=dev)
        w = torch.randn(Cout, Cin, k, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        a = Act(x, C=Cin, scale=torch.ones(Cin, device=dev), shift=torch.zeros(Cin, device=dev),
                alpha=torch.full((Cin,), 0.1, device=dev))
        return lambda: E.conv_fwd(a, w.view(Cout, -1), b, Cout=Cout, taps=k, stride=st, padL=pL, padR=pR,
                                  pad_mode=K.PAD_REFLECT, want_stats=True)
    if name == "cat":        # dense-skip + W projection: (96, 1920, 200) -> 256 rows, 1x1, BatchNorm statistics
        x = torch.randn(S, 1920, 200, device=dev)
        w = torch.randn(256, 1920, device=dev) * 0.05
        b = torch.randn(256, device=dev)
        return lambda: E.conv_fwd(Act(x, C=1920), w, b, Cout=256, taps=1, want_stats=True, Tout=200)
    if name == "dec3d":      # decoder output layer's data gradient: Conv1d(128 -> 256, k 30, stride 10) over 32 x 32000
        x = torch.randn(32, 128, 32000, device=dev)
        w = torch.randn(256, 128 * 30, device=dev) * 0.05
        return lambda: E.conv_fwd(Act(x, C=128), w, None, Cout=256, taps=30, stride=10, padL=10, padR=10, pad_mode=K.PAD_ZERO,
                                  Tout=3200)
    if name in ("qrnn", "qrnn_nb"):
        x = torch.randn(S, 512, 200, device=dev)
        lin = torch.randn(1536, 1024, device=dev) * 0.05
        b = torch.randn(1536, device=dev) if name == "qrnn" else None
        y = torch.empty(S, 1536, 200, device=dev)
        return lambda: K.conv_gemm(x, lin, y, S=S, Cin=512, Tin=200, M=1536, K=1024, taps=2, Ncols=200, Tout=200, bias=b,
                                   tap_major=1, tapstep=-1)
    if name == "lps":
        B, F_, D, r = 32, 200, 3075, 7
        h = torch.randn(B, 256, F_, device=dev)
        W = torch.randn(D * r, 256, device=dev) * 0.05
        bb = torch.zeros(D * r, device=dev)
        lab = torch.randn(B, D, F_, device=dev)
        g = torch.empty(B, D * r, F_, device=dev)
        acc = torch.zeros(1, dtype=torch.float64, device=dev)
        al = torch.full((256,), 0.25, device=dev)
        return lambda: K.conv_gemm(h, W, None, S=B, Cin=256, Tin=F_, M=D * r, K=256, taps=1, Ncols=F_, Tout=F_, bias=bb,
                                   in_alpha=al, epilogue=K.EPI_MSE_CTX, label=lab, grad_out=g, loss_acc=acc,
                                   grad_scale=1e-6, r_ctx=r, label_D=D)
    if name == "dec3":       # decoder output layer: ConvTranspose1d(256 -> 128, k 30, stride 10)
        x = torch.randn(32, 256, 3200, device=dev)
        w = torch.randn(256, 128, 30, device=dev) * 0.05
        b = torch.randn(128, device=dev)
        a = Act(x, C=256, alpha=torch.full((256,), 0.1, device=dev))
        return lambda: E.deconv_fwd(a, w, b, Cout=128, k=30, stride=10)
    if name in ("wg5", "wg7", "wglps", "wgqrnn"):     # weight gradients: block 5 / block 7 conv, LPS head, QRNN tap
        if name in ("wg5", "wg7"):
            Cin, Cout, k, st, Tin = {"wg5": (256, 256, 11, 1, 800), "wg7": (512, 512, 11, 2, 400)}[name]
            pL, pR = E.reflect_pads(k, st)
            x = torch.randn(S, Cin, Tin, device=dev)
            a = Act(x, C=Cin, scale=torch.ones(Cin, device=dev), shift=torch.zeros(Cin, device=dev),
                    alpha=torch.full((Cin,), 0.1, device=dev))
            Tout = (Tin + pL + pR - k) // st + 1
            dy = torch.randn(S, Cout, Tout, device=dev)
            dw = torch.zeros(Cout, Cin * k, device=dev)
            db = torch.zeros(Cout, device=dev)
            return lambda: E.conv_wgrad(dy, a, dw, db, taps=k, stride=st, padL=pL, pad_mode=K.PAD_REFLECT)
        if name == "wglps":
            g = torch.randn(32, 21525, 200, device=dev)
            h = torch.randn(32, 256, 200, device=dev)
            dw = torch.zeros(21525, 256, device=dev)
            db = torch.zeros(21525, device=dev)
            return lambda: E.conv_wgrad(g, Act(h, C=256, alpha=torch.full((256,), 0.25, device=dev)), dw, db, taps=1)
        g = torch.randn(S, 1536, 200, device=dev)
        h = torch.randn(S, 512, 200, device=dev)
        dw = torch.zeros(1536, 512, device=dev)
        db = torch.zeros(1536, device=dev)
        return lambda: E.conv_wgrad(g, Act(h, C=512), dw, db, taps=1)
    if name == "dgrad21525":
        B, F_ = 32, 200
        g = torch.randn(B, 21525, F_, device=dev)
        W = torch.randn(21525, 256, device=dev) * 0.05
        return lambda: E.conv_dgrad(g, W, R=21525, O=256, k=1, stride=1, Tin=F_, padL=0, padR=0, s_red=256, s_out=1, s_k=1)
    raise SystemExit("unknown shape " + name)


def main():
    shapes = sys.argv[1:] or ["blk5", "blk7", "qrnn", "lps", "dec3", "wg5", "wg7", "wglps", "wgqrnn"]
    so = build_trace_lib()
    if shapes == ["build"]:
        print(so)
        return
    _lib.use_library(so, "cuda")
    lib = _lib.lib()
    lib.pase_x6c_trace_read.argtypes = [C.c_void_p]
    lib.pase_x6c_trace_reset.argtypes = []
    dev = torch.device("cuda:0")
    NI = 64
    buf = (C.c_ulonglong * (2 * NI * 20))()
    for name in shapes:
        fn = run_shape(name, dev)
        fn()
        fn()
        torch.cuda.synchronize()
        lib.pase_x6c_trace_reset()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        assert lib.pase_x6c_trace_read(buf) == 0
        print("== %s: %.3f ms per call (pack launches included), plan kind %s, %s" % (name, ms, K.LAST_PLAN_KIND, K.LAST_KERNEL))
        for wg in (0, 1):
            rows = []
            for i in range(NI):
                t = [buf[(wg * NI + i) * 20 + s] for s in range(20)]
                if t[3] == 0 or t[3] < t[0]:
                    break
                rows.append(t)
            if len(rows) < 1:
                print("   workgroup %s: no items traced" % ("0" if wg == 0 else "131"))
                continue
            mid = rows[1:-1] if len(rows) >= 3 else rows
            def avg(f):
                v = [f(r) for r in mid]
                return sum(v) / len(v)
            print("   workgroup %-3s items %2d | compute: wait %7.0f  mfma %7.0f (in barriers %7.0f)  epi %7.0f (setup %5.0f, rows %6.0f, "
                  "tail %6.0f)  item-to-item %7.0f"
                  " | staging: pro %7.0f  loop %7.0f (busy %7.0f: vmwait %7.0f, convert+store %7.0f)" % (
                      "0" if wg == 0 else "131", len(rows), avg(lambda r: r[1] - r[0]), avg(lambda r: r[2] - r[1]),
                      avg(lambda r: r[9]), avg(lambda r: r[3] - r[2]), avg(lambda r: r[12] - r[2]), avg(lambda r: r[13] - r[12]),
                      avg(lambda r: r[3] - r[13]), (rows[-1][0] - rows[0][0]) / max(1, len(rows) - 1),
                      avg(lambda r: r[5] - r[4]), avg(lambda r: r[6] - r[5]), avg(lambda r: r[8]), avg(lambda r: r[10]),
                      avg(lambda r: r[11])))
            if mid[0][14]:                      # lean store epilogue: bias wait + first row / rows 1-7 / rows 8-15
                print("        rows: first %6.0f, next seven %6.0f, last eight %6.0f" % (
                    avg(lambda r: r[14] - r[12]), avg(lambda r: r[15] - r[14]), avg(lambda r: r[13] - r[15])))
                if mid[0][18]:
                    print("        first: to the store branch %5.0f, offsets + interior vote %5.0f, bias %5.0f, row 0 %5.0f" % (
                        avg(lambda r: r[16] - r[12]), avg(lambda r: r[17] - r[16]), avg(lambda r: r[18] - r[17]),
                        avg(lambda r: r[14] - r[18])))


if __name__ == "__main__":
    main()
