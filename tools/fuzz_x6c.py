"""Random-shape sweep of the split-bf16 kernels against fp64 torch references (GPU box):  python tools/fuzz_x6c.py [N] [seed]
Convolutions (fp32-staged, pre-split staging-wave, both symmetric forms) and weight gradients (fp32-staged, pre-split planes,
symmetric) at shapes the PASE+ step never launches: ragged channels / rows / columns, few sequences, odd strides.  Prints one
line per failure and a summary; exit code 1 on any failure.  The fixed-shape tests are tests/test_conv_x6c.py / test_wgrad_x6c.py."""
import os
import random
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pase_amd import kernels as K  # noqa: E402

K.X6 = True
dev = torch.device("cuda:0")


def rel(a, ref):
    return float((a.cpu().double() - ref).norm() / ref.norm().clamp_min(1e-300))


def xf(x, sc, sh, al):
    v = x.double() * sc.double()[None, :, None] + sh.double()[None, :, None]
    return torch.where(v > 0, v, v * al.double()[None, :, None])


def one_conv(rng):
    Cin = rng.choice([16, 24, 40, 64, 100, 128, 256, 300])
    Cout = rng.choice([64, 100, 128, 130, 256, 300, 512, 640, 1024, 1100, 1536])
    k = rng.choice([1, 1, 2, 3, 5, 11])
    st = 1 if k <= 2 else rng.choice([1, 1, 2, 4])
    T = rng.randint(max(40, 3 * k), 420)
    S = rng.randint(1, 5)
    xp = rng.choice(["", "1", "0"])
    sym = rng.choice(["1", "1", "8", "duo", "0"])
    maxwg = rng.choice([0, 0, 1, 3, 7])
    os.environ["PASE_X6C_FORCE"] = "1"
    os.environ["PASE_X6C_XP"] = xp
    os.environ["PASE_X6C_SYM"] = sym
    if maxwg:
        os.environ["PASE_X6C_MAXWG"] = str(maxwg)
    else:
        os.environ.pop("PASE_X6C_MAXWG", None)
    x = torch.randn(S, Cin, T)
    w = torch.randn(Cout, Cin, k) * 0.2
    b = torch.randn(Cout)
    sc, sh, al = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.rand(Cin) * 0.5
    P = (k // 2 - 1, k // 2) if (st > 1 or k % 2 == 0) else (k // 2, k // 2)
    P = (max(P[0], 0), P[1])
    ref = F.conv1d(F.pad(xf(x, sc, sh, al), P, mode="reflect") if k > 1 else xf(x, sc, sh, al), w.double(), b.double(), stride=st)
    Tout = ref.shape[2]
    y = torch.full((S, Cout, Tout), float("nan"), device=dev)
    K.conv_gemm(x.to(dev), w.reshape(Cout, -1).contiguous().to(dev), y, S=S, Cin=Cin, Tin=T, M=Cout, K=Cin * k, taps=k,
                Ncols=Tout, Tout=Tout, bias=b.to(dev), stride=st, padL=P[0], pad_mode=K.PAD_REFLECT, in_scale=sc.to(dev),
                in_shift=sh.to(dev), in_alpha=al.to(dev), splitk=1)
    e = rel(y, ref)
    tag = "conv Cin%d Cout%d k%d s%d T%d S%d xp=%r sym=%s maxwg=%d kind %s %s" % (Cin, Cout, k, st, T, S, xp, sym, maxwg,
                                                                                   K.LAST_PLAN_KIND, K.LAST_KERNEL)
    return e, tag


def one_mse(rng):
    """1x1 projection with the fused r-context MSE epilogue (the LPS-style heads): loss sum, prediction, d(loss)/d(prediction)"""
    Cin = rng.choice([64, 128, 256, 272, 768])
    D = rng.choice([3, 21, 40, 123, 300])
    r = rng.choice([3, 5, 7])
    Fr = rng.randint(20, 260)
    B = rng.randint(1, 6)
    outs = rng.choice(["both", "grad", "pred"])
    os.environ["PASE_X6C_FORCE"] = "1"
    os.environ["PASE_X6C_XP"] = rng.choice(["", "1", "0"])
    os.environ["PASE_X6C_SYM"] = rng.choice(["1", "1", "8", "duo", "0"])
    maxwg = rng.choice([0, 0, 1, 3])
    if maxwg:
        os.environ["PASE_X6C_MAXWG"] = str(maxwg)
    else:
        os.environ.pop("PASE_X6C_MAXWG", None)
    M = D * r
    h = torch.randn(B, Cin, Fr)
    w = torch.randn(M, Cin) * 0.2
    b = torch.randn(M)
    al = torch.rand(Cin) * 0.5
    lab = torch.randn(B, D, Fr)
    hin = torch.where(h > 0, h, h * al[None, :, None]).double()
    pred = torch.einsum("mk,bkt->bmt", w.double(), hin) + b.double()[None, :, None]
    padded = F.pad(lab.double(), (r // 2, r // 2))
    tgt = torch.stack([padded[:, :, t:t + r].reshape(B, -1) for t in range(Fr)], 2)
    ref_loss = ((pred - tgt) ** 2).sum()
    y = torch.full((B, M, Fr), float("nan"), device=dev) if outs != "grad" else None
    g = torch.full((B, M, Fr), float("nan"), device=dev) if outs != "pred" else None
    acc = torch.zeros(1, dtype=torch.float64, device=dev)
    K.conv_gemm(h.to(dev), w.to(dev), y, S=B, Cin=Cin, Tin=Fr, M=M, K=Cin, taps=1, Ncols=Fr, Tout=Fr, bias=b.to(dev),
                in_alpha=al.to(dev), epilogue=K.EPI_MSE_CTX, label=lab.to(dev), grad_out=g, loss_acc=acc, grad_scale=0.5, r_ctx=r,
                label_D=D)
    e = abs(float(acc) - float(ref_loss)) / float(ref_loss)
    if y is not None:
        e = max(e, rel(y, pred))
    if g is not None:
        e = max(e, 0.5 * rel(g, 0.5 * (pred - tgt)))      # (the gradient is a difference of two O(1) numbers: 2e-6 bound)
    tag = "mse Cin%d D%d r%d F%d B%d outs=%s xp=%r sym=%s maxwg=%d kind %s %s" % (
        Cin, D, r, Fr, B, outs, os.environ["PASE_X6C_XP"], os.environ["PASE_X6C_SYM"], maxwg, K.LAST_PLAN_KIND, K.LAST_KERNEL)
    return e, tag


def one_wgrad(rng):
    Cin = rng.choice([12, 16, 24, 40, 64, 100, 128])
    Cout = rng.choice([96, 130, 256, 260, 300, 512, 520])
    k = rng.choice([3, 4, 5, 8, 11, 30])
    st = rng.choice([1, 1, 2, 4]) if k >= 4 else 1
    T = rng.randint(max(60, 4 * k), 500)
    S = rng.randint(1, 5)
    mode = rng.choice(["", "", "1", "3"])
    sym = rng.choice(["1", "1", "0"])
    maxwg = rng.choice([0, 0, 1, 2, 5])
    bias = rng.random() < 0.6
    os.environ["PASE_X6C_WGRAD_FLAT"] = "1"
    if mode:
        os.environ["PASE_X6C_WGRAD_MODE"] = mode
    else:
        os.environ.pop("PASE_X6C_WGRAD_MODE", None)
    os.environ["PASE_X6C_WGRAD_SYM"] = sym
    if maxwg:
        os.environ["PASE_X6C_MAXWG"] = str(maxwg)
    else:
        os.environ.pop("PASE_X6C_MAXWG", None)
    x = torch.randn(S, Cin, T)
    sc, sh, al = torch.rand(Cin) + 0.5, torch.randn(Cin) * 0.1, torch.rand(Cin) * 0.5
    P = (k // 2 - 1, k // 2) if (st > 1 or k % 2 == 0) else (k // 2, k // 2)
    w = torch.randn(Cout, Cin, k, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
    y = F.conv1d(F.pad(xf(x, sc, sh, al), P, mode="reflect"), w, b, stride=st)
    g = torch.randn(y.shape)
    (y * g.double()).sum().backward()
    dw = torch.zeros(Cout, Cin * k, device=dev)
    db = torch.zeros(Cout, device=dev) if bias else None
    K.wgrad_gemm(g.to(dev), x.to(dev), dw, S=S, M=Cout, Tg=y.shape[2], Ncols=y.shape[2], Cin=Cin, Tz=T, taps=k, dbias=db,
                 in_scale=sc.to(dev), in_shift=sh.to(dev), in_alpha=al.to(dev), stride=st, padL=P[0], pad_mode=K.PAD_REFLECT)
    e = rel(dw.view(Cout, Cin, k), w.grad)
    if bias:
        e = max(e, rel(db, b.grad))
    tag = "wgrad Cin%d Cout%d k%d s%d T%d S%d mode=%r sym=%s maxwg=%d bias=%d x6 %s kind %s" % (
        Cin, Cout, k, st, T, S, mode, sym, maxwg, bias, K.LAST_WGRAD_X6, K.LAST_WGRAD_KIND)
    return e, tag


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = random.Random(seed)
    torch.manual_seed(seed)
    bad, kinds, worst = 0, {}, 0.0
    for i in range(n):
        fn = (one_conv, one_wgrad, one_mse)[i % 3]
        try:
            e, tag = fn(rng)
        except Exception as ex:      # a refusal (-11 / -12 ...) of a forced combination is reported, not fatal
            print("EXC", fn.__name__, repr(ex)[:200])
            bad += 1
            continue
        key = tag.split(" kind ")[1] if " kind " in tag else "?"
        kinds[key] = kinds.get(key, 0) + 1
        worst = max(worst, e)
        if not (e < 1e-6):
            bad += 1
            print("FAIL %.3e  %s" % (e, tag))
        elif e > 6e-7:
            print("high %.3e  %s" % (e, tag))
    print("cases %d, failures %d, worst rel. error %.3e" % (n, bad, worst))
    for k_, v in sorted(kinds.items(), key=lambda kv: -kv[1]):
        print("   %4d  %s" % (v, k_))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
