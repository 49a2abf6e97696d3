"""Is the split-bf16 contraction BIASED?  (round-3 investigation: the golden step's encoder gradients sit 1e-3 from
the fp64 reference on the split pipe, 1e-4 on the fp32 MFMA pipe, although single launches are 1e-7 from fp64 in L2.)

Random rounding errors average out in the long sums a training step makes of a layer's outputs (BatchNorm statistics,
weight gradients over 1e5 positions); a systematic error -- truncation in the operand split, or in the matrix core's
internal accumulation -- does not.  This probe measures the SIGNED error of single conv_gemm / wgrad launches against
fp64 on both pipes: mean(err) / rms(err) ~ 1/sqrt(n) for unbiased noise, O(1) for a bias.

usage: python tools/x6_bias_probe.py            (needs a GPU)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from pase_amd import kernels as K


def conv_case(name, Cin, Cout, k, T, S, positive, dev, scale_w=0.2):
    torch.manual_seed(5)
    x = torch.randn(S, Cin, T)
    w = torch.randn(Cout, Cin, k) * scale_w
    if positive:
        x, w = x.abs(), w.abs()
    pad = (k // 2, k // 2) if k > 1 else (0, 0)
    ref = F.conv1d(F.pad(x.double(), pad), w.double()).to(dev)
    Tout = ref.shape[2]
    kw = dict(S=S, Cin=Cin, Tin=T, M=Cout, K=Cin * k, taps=k, Ncols=Tout, Tout=Tout, stride=1, padL=pad[0],
              pad_mode=K.PAD_ZERO, splitk=1)
    w2 = w.reshape(Cout, -1).contiguous().to(dev)
    xd = x.to(dev)
    out = dict(case=name, K=Cin * k, positive=positive, n=int(ref.numel()))
    saved = K.X6
    try:
        for mode, tag in ((True, "x6"), (False, "f32")):
            K.X6 = mode
            y = torch.zeros(S, Cout, Tout, device=dev)
            K.conv_gemm(xd, w2, y, **kw)
            e = y.double() - ref
            # scale of the summands of one output: rms over outputs of sum |a b| is what an ulp-level error is relative to
            out[tag] = dict(rel_l2=float(e.norm() / ref.norm()),
                            mean_over_rms=float(e.mean() / e.pow(2).mean().sqrt().clamp_min(1e-300)),
                            mean_rel=float((e / ref.abs().clamp_min(1e-30)).mean()) if positive else None,
                            frac_negative=float((e < 0).double().mean()))
    finally:
        K.X6 = saved
    return out


def main():
    dev = torch.device("cuda:0")
    res = []
    for positive in (True, False):
        res.append(conv_case("1x1 K=16", 16, 128, 1, 4096, 2, positive, dev))
        res.append(conv_case("1x1 K=256", 256, 128, 1, 4096, 2, positive, dev))
        res.append(conv_case("1x1 K=2048", 2048, 128, 1, 2048, 2, positive, dev))
        res.append(conv_case("11 taps Cin=128", 128, 128, 11, 1600, 2, positive, dev))
    for r in res:
        print(json.dumps(r))


if __name__ == "__main__":
    main()
