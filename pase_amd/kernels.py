"""Thin ctypes bindings: one Python function per C-ABI entry point of include/pase_amd.h.

Tensors are torch tensors used purely as device buffers (data_ptr + the current HIP stream);
every wrapper checks dtype / contiguity / device and raises on a non-zero return code.
"""
import ctypes as C
import os

import torch

from . import _lib

PAD_ZERO, PAD_REFLECT = 0, 1
EPI_STORE, EPI_MSE_CTX = 0, 1
POST_NONE, POST_POW, POST_LOGPOW, POST_LOG, POST_MAG, POST_RELU, POST_SQRTPOS = 0, 1, 2, 3, 4, 5, 6

_fp = C.c_void_p


class PaseConvGemm(C.Structure):
    _fields_ = [
        ("x", _fp), ("w", _fp), ("wt", _fp), ("y", _fp), ("bias", _fp),
        ("in_scale", _fp), ("in_shift", _fp), ("in_alpha", _fp),
        ("stat_part", _fp), ("label", _fp), ("grad_out", _fp), ("loss_acc", _fp),
        ("grad_scale", C.c_float),
        ("S", C.c_int), ("Cin", C.c_int), ("Tin", C.c_int), ("x_ctot", C.c_int), ("x_coff", C.c_int),
        ("M", C.c_int), ("K", C.c_int), ("ldw", C.c_int), ("ldwt", C.c_int), ("taps", C.c_int),
        ("tap_major", C.c_int),
        ("stride", C.c_int), ("tapstep", C.c_int), ("padL", C.c_int), ("pad_mode", C.c_int),
        ("Ncols", C.c_int),
        ("y_ctot", C.c_int), ("y_coff", C.c_int), ("Cout_store", C.c_int), ("ps", C.c_int),
        ("poff", C.c_int), ("Tout", C.c_int),
        ("epilogue", C.c_int), ("r_ctx", C.c_int), ("label_D", C.c_int),
        ("tile_hint", C.c_int), ("post_op", C.c_int), ("post_scale", C.c_float), ("post_eps", C.c_float),
        ("splitk", C.c_int),
        ("wx6", C.c_void_p), ("xp6", C.c_void_p),
        ("x6_ctl", C.c_int), ("max_wg", C.c_int),
    ]


def declare(l):
    l.pase_conv_gemm.argtypes = [C.POINTER(PaseConvGemm), C.c_void_p]
    l.pase_conv_gemm.restype = C.c_int
    l.pase_conv_gemm_stat_tiles.argtypes = [C.POINTER(PaseConvGemm)]
    l.pase_conv_gemm_stat_tiles.restype = C.c_int
    l.pase_conv_gemm_splitk.argtypes = [C.POINTER(PaseConvGemm)]
    l.pase_conv_gemm_splitk.restype = C.c_int
    l.pase_conv_gemm_x6_bytes.argtypes = [C.POINTER(PaseConvGemm)]
    l.pase_conv_gemm_x6_bytes.restype = C.c_long
    l.pase_conv_gemm_plan_kind.argtypes = [C.POINTER(PaseConvGemm)]
    l.pase_conv_gemm_plan_kind.restype = C.c_int
    l.pase_conv_gemm_kernel_id.argtypes = [C.POINTER(PaseConvGemm)]
    l.pase_conv_gemm_kernel_id.restype = C.c_int
    l.pase_pack_x6.argtypes = [C.POINTER(PaseConvGemm), C.c_void_p]
    l.pase_pack_x6.restype = C.c_int
    l.pase_conv_gemm_xp_bytes.argtypes = [C.POINTER(PaseConvGemm)]
    l.pase_conv_gemm_xp_bytes.restype = C.c_long
    l.pase_pack_xp.argtypes = [C.POINTER(PaseConvGemm), C.c_void_p]
    l.pase_pack_xp.restype = C.c_int
    l.pase_wgrad_x6_bytes.argtypes = [C.POINTER(PaseWgrad)]
    l.pase_wgrad_x6_bytes.restype = C.c_long
    l.pase_wgrad_plan_kind.argtypes = [C.POINTER(PaseWgrad)]
    l.pase_wgrad_plan_kind.restype = C.c_int
    l.pase_abi_sizeof.argtypes = [C.c_int]
    l.pase_abi_sizeof.restype = C.c_int
    if l.pase_abi_sizeof(0) != C.sizeof(PaseConvGemm):
        raise _lib.PaseLibraryError("ABI mismatch: PaseConvGemm")
    for name, args in _SIMPLE.items():
        fn = getattr(l, name)
        fn.argtypes = args
        fn.restype = C.c_int
    abi_check(l)


# name -> argtypes for the flat (non-struct) entry points; filled in below
_SIMPLE = {}


def _stream():
    if _lib.device_type() == "cuda":
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)
    return C.c_void_p(0)


def _ptr(t, dtype=torch.float32):
    if t is None:
        return None
    if t.dtype != dtype:
        raise TypeError("pase_amd kernel: expected %s, got %s" % (dtype, t.dtype))
    if t.device.type != _lib.device_type():
        raise _lib.PaseLibraryError(
            "pase_amd kernel: tensor on %s but the kernel library runs on %s (no CPU fallback)"
            % (t.device.type, _lib.device_type()))
    if not t.is_contiguous():
        raise ValueError("pase_amd kernel: tensor must be contiguous")
    return t.data_ptr()


def _check(rc, name):
    if rc != 0:
        raise RuntimeError("%s failed with code %d" % (name, rc))


def _conv_desc(x, w, y, *, S, Cin, Tin, M, K, taps, Ncols, Tout, ldw=None, bias=None,
               in_scale=None, in_shift=None, in_alpha=None, stat_part=None,
               x_ctot=None, x_coff=0, tap_major=0, stride=1, tapstep=1, padL=0, pad_mode=PAD_ZERO,
               y_ctot=None, y_coff=0, Cout_store=None, ps=1, poff=0,
               epilogue=EPI_STORE, label=None, grad_out=None, loss_acc=None, grad_scale=0.0,
               r_ctx=0, label_D=0, tile_hint=0, splitk=0, post_op=0, post_scale=1.0, post_eps=0.0, wt=None, max_wg=0):
    d = PaseConvGemm()
    if wt is not None:
        d.wt, d.ldwt = _ptr(wt), wt.shape[1]
    d.post_op, d.post_scale, d.post_eps = post_op, post_scale, post_eps
    d.x, d.w, d.y, d.bias = _ptr(x), _ptr(w), _ptr(y), _ptr(bias)
    d.in_scale, d.in_shift, d.in_alpha = _ptr(in_scale), _ptr(in_shift), _ptr(in_alpha)
    d.stat_part, d.label, d.grad_out = _ptr(stat_part), _ptr(label), _ptr(grad_out)
    d.loss_acc = _ptr(loss_acc, torch.float64)
    d.grad_scale = grad_scale
    d.S, d.Cin, d.Tin = S, Cin, Tin
    d.x_ctot = Cin if x_ctot is None else x_ctot
    d.x_coff = x_coff
    d.M, d.K, d.ldw, d.taps, d.tap_major = M, K, (K if ldw is None else ldw), taps, tap_major
    d.stride, d.tapstep, d.padL, d.pad_mode = stride, tapstep, padL, pad_mode
    d.Ncols = Ncols
    cs = M if Cout_store is None else Cout_store
    d.y_ctot = cs if y_ctot is None else y_ctot
    d.y_coff, d.Cout_store, d.ps, d.poff, d.Tout = y_coff, cs, ps, poff, Tout
    d.epilogue, d.r_ctx, d.label_D = epilogue, r_ctx, label_D
    d.tile_hint = tile_hint
    d.splitk = splitk
    # measurement / test controls travel in the descriptor (the C library reads no environment variables)
    d.x6_ctl = ((1 if os.environ.get("PASE_X6C_FORCE") else 0) | {"1": 2, "0": 4}.get(os.environ.get("PASE_X6C_XP", ""), 0)
                | (8 if os.environ.get("PASE_SINC_X6", "1") == "0" else 0)
                | (16 if os.environ.get("PASE_X6C_NARROW", "1") == "0" else 0)
                | (32 if os.environ.get("PASE_X6C_LEANEPI", "1") == "0" else 0)
                | (64 if os.environ.get("PASE_X6C_BIASINIT", "1") == "0" else 0)
                | (128 if os.environ.get("PASE_X6C_SYM", "1") == "0" else 0)
                | {"8": 0x20000, "duo": 0x40000}.get(os.environ.get("PASE_X6C_SYM", ""), 0)
                | (0x10000 if os.environ.get("PASE_X6C_PAIRS", "1") == "0" else 0)
                | ((int(os.environ.get("PASE_X6C_STAGGER", "0")) & 255) << 8))
    d.max_wg = _max_wg(max_wg)
    return d


def _max_wg(max_wg):
    """Cap on the persistent grid of a split-bf16 launch (PaseConvGemm::max_wg / PaseWgrad::max_wg; 0 = one workgroup per
    CU).  It travels with the CALL -- the data-parallel trainer passes 256 - reserved CUs down the encoder backward so that
    RCCL's channel kernels find free CUs beside the GEMMs -- never through module state.  PASE_X6C_MAXWG (tests: several items
    per workgroup on small shapes) overrides it."""
    e = os.environ.get("PASE_X6C_MAXWG")
    return int(e) if e else int(max_wg or 0)


def stat_tiles(*, M, S, Ncols, Cin, taps, stride=1, padL=0, tapstep=1, tile_hint=0):
    """first dim of the stat_part buffer a conv_gemm launch with these dims writes"""
    d = PaseConvGemm()
    d.M, d.S, d.Ncols, d.Cin, d.taps, d.stride, d.padL, d.tapstep, d.tile_hint = (M, S, Ncols, Cin, taps, stride,
                                                                                  padL, tapstep, tile_hint)
    d.K = Cin * taps
    d.splitk = 1
    return _lib.lib().pase_conv_gemm_stat_tiles(C.byref(d))


class GemmTimer(object):
    """Measurement hook (bench.py): brackets every MFMA-kernel launch with HIP events recorded on the
    stream the kernel is launched on, together with the launch's algorithmic FLOPs."""

    def __init__(self):
        self.records = []
        self.tags = []
        self.pipes = []          # per launch: "x6" (split-bf16 kernel) or "f32" (exact-fp32 MFMA kernels)
        self.kernels = []        # per launch: the kernel instantiation as rocprofv3 names it (prefix)

    def start(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        return ev

    def stop(self, family, flops, ev0, tag=None, pipe="f32", kernel=None):
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record(torch.cuda.current_stream())
        self.records.append((family, flops, ev0, ev1))
        self.tags.append(tag)
        self.pipes.append(pipe)
        self.kernels.append(kernel)

    def per_launch(self):
        """[(family, tag, flops, ms)] in launch order (tools/step_breakdown.py)."""
        torch.cuda.synchronize()
        return [(f, t, fl, e0.elapsed_time(e1)) for (f, fl, e0, e1), t in zip(self.records, self.tags)]

    def summary(self, by_pipe=False, by_kernel=False):
        """{family: launches / flops / ms}; by_pipe: {(family, pipe): ...}; by_kernel: {kernel instantiation: ...}"""
        torch.cuda.synchronize()
        fam = {}
        for (family, flops, e0, e1), pipe, kern in zip(self.records, self.pipes, self.kernels):
            f = fam.setdefault(kern if by_kernel else ((family, pipe) if by_pipe else family), dict(launches=0, flops=0.0, ms=0.0))
            f["launches"] += 1
            f["flops"] += flops
            f["ms"] += e0.elapsed_time(e1)
        return fam


def conv_kernel_name(kid):
    """pase_conv_gemm_kernel_id -> the prefix rocprofv3 shows for that instantiation"""
    if kid == 0:
        return "conv_gemm_kernel<"
    if kid == 1:
        return "sinc_x6_fwd_kernel"
    b = lambda v: "true" if v else "false"
    fl = kid % 100
    return "conv_x6c_kernel<%d, %d, false, %s, %s, %s, %s>" % (kid // 1000, (kid // 100) % 10, b(fl & 2), b(fl & 1), b(fl & 4), b(fl & 8))


WGRAD_KERNEL_NAMES = {0: "wgrad_", 1: "conv_x6c_kernel<128, 4, true, false, false, false, false>", 2: "conv_x6c_kernel<128, 4, true, false, false, false, false>",
                      3: "conv_x6c_kernel<128, 4, true, false, false, false, false>", 4: "conv_x6c_kernel<128, 5, true, true, false, false, false>",
                      5: "sinc_x6_wgrad_kernel<", 7: "x6c_wgrad_sym_kernel"}

GEMM_TIMER = None
LAST_WGRAD_X6 = None       # did the most recent wgrad_gemm launch run on the split-bf16 kernel
LAST_WGRAD_KIND = None     # ... and in which orientation (pase_wgrad_plan_kind: 0 fp32 pipe, 1 / 2 / 3)
LAST_XP = None             # did the most recent conv_gemm launch stage a pre-split activation (pase_pack_xp)
LAST_PLAN_KIND = None      # plan kind of the most recent conv_gemm launch (0 fp32 pipe, 2 split-bf16 x6c): tests / reports
LAST_KERNEL = None         # ... and the kernel instantiation it ran, as rocprofv3 names it (conv_kernel_name)


def pack_wt(w, *, M, K, Cin, taps, ldw=None, tap_major=0):
    """K-major pack (K, ldwt) of the logical A operand (M, K) held row-major in `w` (pase_pack_wt)."""
    ldwt = (M + 3) // 4 * 4
    wt = torch.empty(K, ldwt, device=w.device, dtype=torch.float32)
    _check(_lib.lib().pase_pack_wt(_ptr(w), _ptr(wt), M, K, Cin, taps, K if ldw is None else ldw, tap_major, ldwt,
                                   _stream()), "pase_pack_wt")
    return wt


# PASE_X6=0 keeps every contraction on the fp32 matrix pipe (A/B measurements, tools/)
X6 = os.environ.get("PASE_X6", "1") != "0"


def _x6_conv_ok(kw):
    # debugging filter: PASE_X6_ONLY=fwd|bwd|taps<N> restricts the split-bf16 conv launches
    f = os.environ.get("PASE_X6_ONLY")
    if not f:
        return True
    if f == "fwd":
        return kw.get("tapstep", 1) == 1 and kw.get("ps", 1) == 1
    if f == "bwd":
        return not (kw.get("tapstep", 1) == 1 and kw.get("ps", 1) == 1)
    if f.startswith("taps"):
        return kw["taps"] == int(f[4:])
    return True


ZERO_ALLOC = None      # set by pase_amd.engine: (shape, like) -> zero-initialised tensor out of the step's zero arena


def conv_gemm_out(x, w, y_shape, **kw):
    """conv_gemm into a freshly allocated output.  A launch the library will split along K adds into a zeroed output: that
    one comes out of the step's zero arena (ONE memset per step) instead of a torch.empty + a fill launch of its own."""
    d = _conv_desc(x, w, None, **kw)
    split = False
    if d.splitk != 1:
        if X6 and os.environ.get("PASE_X6_CONV", "1") != "0" and _x6_conv_ok(kw) and _lib.lib().pase_conv_gemm_x6_bytes(C.byref(d)) > 0:
            d.wx6 = 1          # non-NULL marker: the split-K factor is the split-bf16 plan's
        split = _lib.lib().pase_conv_gemm_splitk(C.byref(d)) > 1
    if split and ZERO_ALLOC is not None:
        y = ZERO_ALLOC(tuple(y_shape), x)
        conv_gemm(x, w, y, y_zeroed=True, **kw)
    else:
        y = torch.empty(tuple(y_shape), device=x.device, dtype=torch.float32)
        conv_gemm(x, w, y, **kw)
    return y


def conv_gemm(x, w, y, want_stats=False, y_zeroed=False, **kw):
    """see include/pase_amd.h PaseConvGemm.  With splitk > 1 the output is zero-filled here first.
    The fp32-pipe kernels read the K-major pack of the weight: pass it as wt= (e.g. straight from pack_dgrad, or a
    weight that already is K-major) or it is produced here from `w`; a split-bf16 launch packs its operand from wt= if
    given, else straight from `w`.
    want_stats=True: the (column tiles, M, 2) partial-sum buffer for pase_bn_finalize is allocated here (the tile count
    is a function of the plan the library picks for the COMPLETE descriptor, split-bf16 pack included) and returned."""
    d = _conv_desc(x, w, y, **kw)
    global LAST_XP
    LAST_XP = False
    # (per-launch timing brackets the launch's own operand packs too: weight pack, pre-split activation)
    ev0 = GEMM_TIMER.start() if GEMM_TIMER is not None else None
    if want_stats:
        # the plan (pixel-shuffle row order, automatic split-K) depends on whether BatchNorm partial sums are written: the pack
        # and the launch must see the SAME descriptor, so the statistics buffer exists before the pack is sized.  Its tile count
        # is a function of the plan WITH the split-bf16 pack, hence sized against a descriptor that carries a (dummy) pack.
        d.stat_part = 1          # non-NULL marker for the sizing queries below; replaced by the real buffer
    if X6 and os.environ.get("PASE_X6_CONV", "1") != "0" and _x6_conv_ok(kw):
        # contraction on the bf16 matrix pipe with both operands split into three bf16 pieces (fp32-grade result,
        # see PaseConvGemm::wx6) for the launch shapes the library has a split-bf16 plan for
        nbytes = _lib.lib().pase_conv_gemm_x6_bytes(C.byref(d))
        if nbytes > 0:
            wx6 = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            d.wx6 = wx6.data_ptr()
            _check(_lib.lib().pase_pack_x6(C.byref(d), _stream()), "pase_pack_x6")
            # ... and, where the library asks for it, the activation pre-split once for this launch (staging = copy)
            xbytes = _lib.lib().pase_conv_gemm_xp_bytes(C.byref(d))
            if xbytes > 0:
                xp6 = torch.empty(xbytes, dtype=torch.uint8, device=x.device)
                d.xp6 = xp6.data_ptr()
                _check(_lib.lib().pase_pack_xp(C.byref(d), _stream()), "pase_pack_xp")
                LAST_XP = True
    if not d.wx6 and not d.wt:
        # the fp32-pipe kernels read the K-major pack of the weight (a split-bf16 launch packs straight from `w`)
        wt = pack_wt(w, M=kw["M"], K=kw["K"], Cin=kw["Cin"], taps=kw["taps"], ldw=kw.get("ldw"), tap_major=kw.get("tap_major", 0))
        d.wt, d.ldwt = wt.data_ptr(), wt.shape[1]
    stat = None
    if want_stats:
        stat = torch.empty(_lib.lib().pase_conv_gemm_stat_tiles(C.byref(d)), d.M, 2, device=x.device, dtype=torch.float32)
        d.stat_part = stat.data_ptr()
    elif kw.get("stat_part") is not None:
        need = _lib.lib().pase_conv_gemm_stat_tiles(C.byref(d))
        if kw["stat_part"].shape[0] != need:
            raise ValueError("pase_conv_gemm: stat_part has %d tile rows, the launch writes %d (use want_stats=True)"
                             % (kw["stat_part"].shape[0], need))
    if d.splitk != 1 and not y_zeroed:
        if _lib.lib().pase_conv_gemm_splitk(C.byref(d)) > 1:
            y.zero_()
    global LAST_PLAN_KIND, LAST_KERNEL
    LAST_PLAN_KIND = _lib.lib().pase_conv_gemm_plan_kind(C.byref(d))
    LAST_KERNEL = conv_kernel_name(_lib.lib().pase_conv_gemm_kernel_id(C.byref(d)))
    _check(_lib.lib().pase_conv_gemm(C.byref(d), _stream()), "pase_conv_gemm")
    if ev0 is not None:
        GEMM_TIMER.stop("conv_gemm", 2.0 * d.S * d.Ncols * d.M * d.K, ev0,
                        "M%d K%d(Cin%d x %d) N%dx%d s%d ps%d epi%d" % (d.M, d.K, d.Cin, d.taps, d.S, d.Ncols, d.stride,
                                                                    d.ps, d.epilogue), pipe="x6" if LAST_PLAN_KIND else "f32",
                        kernel=LAST_KERNEL)
    return stat


# ======================================================================================
# wgrad
# ======================================================================================
class PaseWgrad(C.Structure):
    _fields_ = [
        ("g", _fp), ("z", _fp), ("dw", _fp), ("dbias", _fp),
        ("in_scale", _fp), ("in_shift", _fp), ("in_alpha", _fp), ("g_alpha", _fp),
        ("S", C.c_int), ("M", C.c_int), ("Tg", C.c_int), ("g_ctot", C.c_int), ("g_coff", C.c_int),
        ("Ncols", C.c_int),
        ("Cin", C.c_int), ("Tz", C.c_int), ("z_ctot", C.c_int), ("z_coff", C.c_int), ("taps", C.c_int),
        ("tap_major", C.c_int), ("stride", C.c_int), ("tapstep", C.c_int), ("padL", C.c_int),
        ("pad_mode", C.c_int), ("ldw", C.c_int), ("splitk", C.c_int), ("x6", C.c_int),
        ("gx6", C.c_void_p), ("max_wg", C.c_int),
    ]


class PaseActBwd(C.Structure):
    _fields_ = [
        ("y", _fp), ("dsrc", _fp), ("dpool", _fp),
        ("scale", _fp), ("shift", _fp), ("alpha", _fp), ("mean", _fp), ("rstd", _fp),
        ("sums", _fp), ("dy", _fp),
        ("S", C.c_int), ("C", C.c_int), ("T", C.c_int), ("y_ctot", C.c_int), ("y_coff", C.c_int),
        ("dsrc_ctot", C.c_int), ("dsrc_coff", C.c_int), ("Tp", C.c_int), ("padL", C.c_int),
        ("pad_mode", C.c_int),
        ("dpool_ctot", C.c_int), ("dpool_coff", C.c_int), ("pool_F", C.c_int), ("pool_d", C.c_int),
        ("pool_inv", C.c_float), ("has_bn", C.c_int),
    ]


class PaseAddBlock(C.Structure):
    _fields_ = [("src", _fp), ("dst", _fp), ("rows", C.c_int), ("width", C.c_int), ("src_ld", C.c_int), ("dst_ld", C.c_int)]


class PaseAddBlocks(C.Structure):
    _fields_ = [("n", C.c_int), ("seg", PaseAddBlock * 16)]


class PaseMlpHead1(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("y", "alpha0", "w1", "b1", "alpha1", "w2", "b2", "target", "pred", "dy", "loss_acc",
                                          "sums0", "sums1", "dw1")] + \
               [(n, C.c_int) for n in ("S", "C", "T", "H", "loss_type")] + [("grad_scale", C.c_float), ("max_wg", C.c_int)]


_i, _f, _d, _l = C.c_int, C.c_float, C.c_double, C.c_long
_SIMPLE.update({
    "pase_mlp_head1_supported": [C.POINTER(PaseMlpHead1)],
    "pase_mlp_head1_step": [C.POINTER(PaseMlpHead1), _fp],
    "pase_wgrad_gemm": [C.POINTER(PaseWgrad), _fp],
    "pase_wgrad_gemm_act_bwd": [C.POINTER(PaseWgrad), C.POINTER(PaseActBwd), _fp],
    "pase_wgrad_gemm_act_bwd_ok": [C.POINTER(PaseWgrad), C.POINTER(PaseActBwd)],
    "pase_bn_finalize": [_fp, _i, _i, _d, _fp, _fp, _f, _f, _fp, _fp, _fp, _fp, _fp, _fp, _fp],
    "pase_bn_act_pool": [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _fp],
    "pase_bn_act_apply": [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _fp],
    "pase_act_bwd_reduce": [C.POINTER(PaseActBwd), _fp],
    "pase_act_bwd_apply": [C.POINTER(PaseActBwd), _fp],
    "pase_rownorm_act_fwd": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _f, _i, _fp],
    "pase_rownorm_act_bwd": [C.POINTER(PaseActBwd), _fp],
    "pase_qrnn_scan_fwd": [_fp, _fp, _fp, _i, _i, _i, _i, _i, _fp],
    "pase_qrnn_scan_bwd": [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp],
    "pase_head1_fwd": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _f, _fp],
    "pase_head1_bwd": [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _fp],
    "pase_ctx_loss": [_fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _f, _fp],
    "pase_sinc_filters": [_fp, _fp, _fp, _fp, _fp, _i, _i, _f, _f, _f, _fp],
    "pase_sinc_filters_bwd": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _f, _f, _f, _fp],
    "pase_pack_dgrad": [_fp, _fp, _i, _i, _i, _i, _l, _l, _l, _fp],
    "pase_pack_dgrad_t": [_fp, _fp, _i, _i, _i, _i, _l, _l, _l, _i, _fp],
    "pase_chunk_gather": [_fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _fp],
    "pase_peak_scale": [_fp, _fp, _i, _i, _fp],
    "pase_reverb": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _fp],
    "pase_fir_distort": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _fp],
    "pase_clip": [_fp, _fp, _i, _i, _fp],
    "pase_overlap_gather": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _fp],
    "pase_zero_front": [_fp, _fp, _i, _i, _fp],
    "pase_add_noise": [_fp, _fp, _fp, _fp, _fp, _fp, _fp, _i, _i, _fp],
    "pase_gammatone_blocks": [_fp, _fp, _fp, _i, _i, _i, _i, _fp],
    "pase_gammatone_frames": [_fp, _fp, _i, _i, _i, _i, _i, _i, _f, _fp],
    "pase_commit_cols": [_fp, _i, _i, _fp, _i, _fp, _i, _fp, _i, _fp],
    "pase_add_blocks": [C.POINTER(PaseAddBlocks), _fp],
    "pase_pack_wt": [_fp, _fp, _i, _i, _i, _i, _i, _i, _i, _fp],
    "pase_adam_step": [_fp, _fp, _fp, _fp, _l, _fp, _fp, _f, _f, _f, _f, _fp],
    "pase_step_tick": [_fp, _fp],
    "pase_delta_znorm": [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _i, _i, _fp],
    "pase_power_to_db": [_fp, _fp, _fp, _l, _i, _f, _f, _f, _fp],
    "pase_frame_prep": [_fp, _fp, _i, _i, _i, _i, _i, _i, _f, _fp],
    "pase_zcr_rms": [_fp, _fp, _i, _i, _i, _i, _i, _i, _i, _fp],
    "pase_lf0_interp": [_fp, _fp, _i, _i, _i, _i, _i, _f, _fp],
    "pase_swipe_accumulate": [_fp, _fp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _f, _fp],
    "pase_swipe_pick": [_fp, _fp, _fp, _i, _i, _i, _f, _f, _f, _f, _fp],
})

LOSS_NONE, LOSS_L1, LOSS_MSE, LOSS_BCE = 0, 1, 2, 3


def abi_check(l):
    if l.pase_abi_sizeof(1) != C.sizeof(PaseWgrad):
        raise _lib.PaseLibraryError("ABI mismatch: PaseWgrad")
    if l.pase_abi_sizeof(2) != C.sizeof(PaseActBwd):
        raise _lib.PaseLibraryError("ABI mismatch: PaseActBwd")
    if l.pase_abi_sizeof(3) != C.sizeof(PaseAddBlocks):
        raise _lib.PaseLibraryError("ABI mismatch: PaseAddBlocks")
    if l.pase_abi_sizeof(4) != C.sizeof(PaseMlpHead1):
        raise _lib.PaseLibraryError("ABI mismatch: PaseMlpHead1")


def wgrad_gemm(g, z, dw, *, S, M, Tg, Ncols, Cin, Tz, taps, ldw=None, dbias=None, g_ctot=None, g_coff=0,
               z_ctot=None, z_coff=0, in_scale=None, in_shift=None, in_alpha=None, tap_major=0, stride=1,
               tapstep=1, padL=0, pad_mode=PAD_ZERO, splitk=0, g_alpha=None, dw_col_off=0, max_wg=0, g_bwd=None):
    """g_bwd: a dict of act_bwd_apply's keyword arguments (its `y` included) -- the gradient operand is then that apply pass
    evaluated on load (pase_wgrad_gemm_act_bwd; g is not read and may be None).  Only the one-channel SincNet plan has it:
    returns False, with nothing enqueued, when the launch would run on another kernel (the caller then materialises dy)."""
    d = PaseWgrad()
    d.g_alpha = _ptr(g_alpha)
    d.g, d.z, d.dw, d.dbias = _ptr(g), _ptr(z), _ptr(dw) + 4 * dw_col_off, _ptr(dbias)
    d.in_scale, d.in_shift, d.in_alpha = _ptr(in_scale), _ptr(in_shift), _ptr(in_alpha)
    d.S, d.M, d.Tg, d.Ncols = S, M, Tg, Ncols
    d.g_ctot = M if g_ctot is None else g_ctot
    d.g_coff = g_coff
    d.Cin, d.Tz = Cin, Tz
    d.z_ctot = Cin if z_ctot is None else z_ctot
    d.z_coff = z_coff
    d.taps, d.tap_major, d.stride, d.tapstep, d.padL, d.pad_mode = taps, tap_major, stride, tapstep, padL, pad_mode
    d.ldw = Cin * taps if ldw is None else ldw
    d.splitk = splitk
    d.x6 = 1 if (X6 and os.environ.get("PASE_X6_WGRAD", "1") != "0") else 0
    if d.x6:      # orientation forcing / 1x1 layers on the split kernel / no row-coalesced staging: tests and A/B tools
        d.x6 |= (int(os.environ.get("PASE_X6C_WGRAD_MODE", "0")) & 15) << 4
        d.x6 |= 256 if os.environ.get("PASE_X6C_WGRAD_FLAT") else 0
        d.x6 |= 512 if os.environ.get("PASE_X6C_NOVEC") else 0
        d.x6 |= 1024 if os.environ.get("PASE_SINC_X6", "1") == "0" else 0
        d.x6 |= 2048 if os.environ.get("PASE_X6C_WGRAD_SYM", "1") == "0" else 0
        d.x6 |= (int(os.environ.get("PASE_X6C_TMKGS", "0")) & 7) << 12
    d.max_wg = _max_wg(max_wg)
    global LAST_WGRAD_X6, LAST_WGRAD_KIND
    LAST_WGRAD_X6 = False
    LAST_WGRAD_KIND = 0
    if d.x6:
        # split-bf16 contraction (conv_x6c.hip, T-mode): one operand is packed into this scratch by the launch itself
        nbytes = _lib.lib().pase_wgrad_x6_bytes(C.byref(d))
        if nbytes > 0:
            gx6 = torch.empty(nbytes, dtype=torch.uint8, device=z.device)
            d.gx6 = gx6.data_ptr()
            LAST_WGRAD_X6 = True
            LAST_WGRAD_KIND = _lib.lib().pase_wgrad_plan_kind(C.byref(d))
    ab = None
    if g_bwd is not None:
        # the library's own answer (plan kind AND what the on-load kernel can evaluate while staging, e.g. a dense-skip pooled
        # branch needs pool_d >= 16): anything it would refuse is the caller's to materialise -- nothing is enqueued
        kw = dict(g_bwd)
        ab = _act_bwd_desc(kw.pop("y"), **kw)
        if LAST_WGRAD_KIND != 5 or not _lib.lib().pase_wgrad_gemm_act_bwd_ok(C.byref(d), C.byref(ab)):
            return False
    ev0 = GEMM_TIMER.start() if GEMM_TIMER is not None else None
    if g_bwd is not None:
        _check(_lib.lib().pase_wgrad_gemm_act_bwd(C.byref(d), C.byref(ab), _stream()), "pase_wgrad_gemm_act_bwd")
    else:
        _check(_lib.lib().pase_wgrad_gemm(C.byref(d), _stream()), "pase_wgrad_gemm")
    if ev0 is not None:
        GEMM_TIMER.stop("wgrad_gemm", 2.0 * S * Ncols * M * (Cin * taps + (1 if dbias is not None else 0)), ev0,
                        "M%d Kw%d(Cin%d x %d) N%dx%d s%d" % (M, Cin * taps, Cin, taps, S, Ncols, d.stride),
                        pipe="x6" if LAST_WGRAD_X6 else "f32", kernel=WGRAD_KERNEL_NAMES.get(LAST_WGRAD_KIND))
    return True


def bn_finalize(stat_part, C_, count, gamma, beta, eps, momentum, running_mean, running_var, scale, shift,
                mean_out, rstd_out):
    _check(_lib.lib().pase_bn_finalize(_ptr(stat_part), stat_part.shape[0], C_, float(count), _ptr(gamma),
                                       _ptr(beta), eps, momentum, _ptr(running_mean), _ptr(running_var),
                                       _ptr(scale), _ptr(shift), _ptr(mean_out), _ptr(rstd_out), _stream()),
           "pase_bn_finalize")


def bn_act_pool(y, out, scale, shift, alpha, *, S, C_, T, F, d, o_ctot, o_coff):
    _check(_lib.lib().pase_bn_act_pool(_ptr(y), _ptr(out), _ptr(scale), _ptr(shift), _ptr(alpha), S, C_, T, F, d,
                                       o_ctot, o_coff, _stream()), "pase_bn_act_pool")


def bn_act_apply(y, out, scale, shift, alpha, *, S, C_, T):
    _check(_lib.lib().pase_bn_act_apply(_ptr(y), _ptr(out), _ptr(scale), _ptr(shift), _ptr(alpha), S, C_, T,
                                        _stream()), "pase_bn_act_apply")


def _act_bwd_desc(y, *, S, C_, T, dsrc=None, dsrc_ctot=None, dsrc_coff=0, Tp=None, padL=0, pad_mode=PAD_ZERO,
                  dpool=None, dpool_ctot=0, dpool_coff=0, pool_F=0, pool_d=1, scale=None, shift=None, alpha=None,
                  mean=None, rstd=None, sums=None, dy=None, has_bn=0, y_ctot=None, y_coff=0):
    d = PaseActBwd()
    d.y, d.dsrc, d.dpool = _ptr(y), _ptr(dsrc), _ptr(dpool)
    d.scale, d.shift, d.alpha, d.mean, d.rstd = _ptr(scale), _ptr(shift), _ptr(alpha), _ptr(mean), _ptr(rstd)
    d.sums = _ptr(sums, torch.float64)
    d.dy = _ptr(dy)
    d.S, d.C, d.T = S, C_, T
    d.y_ctot = C_ if y_ctot is None else y_ctot
    d.y_coff = y_coff
    d.dsrc_ctot = C_ if dsrc_ctot is None else dsrc_ctot
    d.dsrc_coff = dsrc_coff
    d.Tp = T if Tp is None else Tp
    d.padL, d.pad_mode = padL, pad_mode
    d.dpool_ctot, d.dpool_coff, d.pool_F, d.pool_d = dpool_ctot, dpool_coff, pool_F, max(1, pool_d)
    d.pool_inv = 1.0 / max(1, pool_d)
    d.has_bn = has_bn
    return d


def act_bwd_reduce(y, **kw):
    d = _act_bwd_desc(y, **kw)
    _check(_lib.lib().pase_act_bwd_reduce(C.byref(d), _stream()), "pase_act_bwd_reduce")


def act_bwd_apply(y, **kw):
    d = _act_bwd_desc(y, **kw)
    _check(_lib.lib().pase_act_bwd_apply(C.byref(d), _stream()), "pase_act_bwd_apply")


NORM_INSTANCE, NORM_LAYER = 0, 1


def rownorm_act_fwd(y, out, gamma, beta, alpha, mean_out, rstd_out, *, S, C_, T, eps, mode):
    _check(_lib.lib().pase_rownorm_act_fwd(_ptr(y), _ptr(out), _ptr(gamma), _ptr(beta), _ptr(alpha), _ptr(mean_out),
                                           _ptr(rstd_out), S, C_, T, eps, mode, _stream()), "pase_rownorm_act_fwd")


def rownorm_act_bwd(y, **kw):
    d = _act_bwd_desc(y, **kw)
    _check(_lib.lib().pase_rownorm_act_bwd(C.byref(d), _stream()), "pase_rownorm_act_bwd")


def qrnn_scan_fwd(gates, h_out, c_out, *, S, H, F, h_ctot, h_coff):
    _check(_lib.lib().pase_qrnn_scan_fwd(_ptr(gates), _ptr(h_out), _ptr(c_out), S, H, F, h_ctot, h_coff, _stream()),
           "pase_qrnn_scan_fwd")


def qrnn_scan_bwd(gates, c_saved, dh, dgates, *, S, H, F, dh_ctot, dh_coff):
    _check(_lib.lib().pase_qrnn_scan_bwd(_ptr(gates), _ptr(c_saved), _ptr(dh), _ptr(dgates), S, H, F, dh_ctot,
                                         dh_coff, _stream()), "pase_qrnn_scan_bwd")


def head1_fwd(z, w, bias, *, S, C_, T, in_scale=None, in_shift=None, in_alpha=None, target=None, y=None, dy=None,
              loss_acc=None, loss_type=LOSS_NONE, grad_scale=0.0):
    _check(_lib.lib().pase_head1_fwd(_ptr(z), _ptr(in_scale), _ptr(in_shift), _ptr(in_alpha), _ptr(w), _ptr(bias),
                                     _ptr(target), _ptr(y), _ptr(dy), _ptr(loss_acc, torch.float64), S, C_, T,
                                     loss_type, grad_scale, _stream()), "pase_head1_fwd")


def head1_bwd(z, in_alpha, w, dy, dz, sums, *, S, C_, T):
    _check(_lib.lib().pase_head1_bwd(_ptr(z), _ptr(in_alpha), _ptr(w), _ptr(dy), _ptr(dz),
                                     _ptr(sums, torch.float64), S, C_, T, _stream()), "pase_head1_bwd")


def _mlp_head1_desc(y, alpha0, w1, b1, alpha1, w2, b2, target, pred, dy, loss_acc, sums0, sums1, dw1, *, S, C_, T, H,
                    loss_type, grad_scale, max_wg=0):
    d = PaseMlpHead1()
    d.y, d.alpha0, d.w1, d.b1, d.alpha1, d.w2, d.b2 = (_ptr(v) for v in (y, alpha0, w1, b1, alpha1, w2, b2))
    d.target, d.pred, d.dy, d.dw1 = _ptr(target), _ptr(pred), _ptr(dy), _ptr(dw1)
    d.loss_acc, d.sums0, d.sums1 = (_ptr(v, torch.float64) for v in (loss_acc, sums0, sums1))
    d.S, d.C, d.T, d.H, d.loss_type, d.grad_scale = S, C_, T, H, loss_type, grad_scale
    d.max_wg = _max_wg(max_wg)
    return d


def mlp_head1_supported(*, S, C_, T, H):
    d = PaseMlpHead1()
    d.S, d.C, d.T, d.H = S, C_, T, H
    return bool(_lib.lib().pase_mlp_head1_supported(C.byref(d)))


MLP_HEAD1_CALLS = 0        # launches of the one-pass decoder tail so far (tests assert the path they mean to exercise ran)


def mlp_head1_step(y, alpha0, w1, b1, alpha1, w2, b2, target, pred, dy, loss_acc, sums0, sums1, dw1, **kw):
    global MLP_HEAD1_CALLS
    MLP_HEAD1_CALLS += 1
    d = _mlp_head1_desc(y, alpha0, w1, b1, alpha1, w2, b2, target, pred, dy, loss_acc, sums0, sums1, dw1, **kw)
    ev0 = GEMM_TIMER.start() if GEMM_TIMER is not None else None
    _check(_lib.lib().pase_mlp_head1_step(C.byref(d), _stream()), "pase_mlp_head1_step")
    if ev0 is not None:
        GEMM_TIMER.stop("mlp_head1", 2.0 * d.S * d.T * (3 * d.C * d.H + 2 * d.H), ev0,
                        "C%d H%d N%dx%d fwd + bwd" % (d.C, d.H, d.S, d.T), pipe="f32", kernel="mlp_head1_kernel<")


def ctx_loss(pred, label, dpred, loss_acc, *, B, M, F, r_ctx, label_D, loss_type, grad_scale):
    _check(_lib.lib().pase_ctx_loss(_ptr(pred), _ptr(label), _ptr(dpred), _ptr(loss_acc, torch.float64), B, M, F,
                                    r_ctx, label_D, loss_type, grad_scale, _stream()), "pase_ctx_loss")


def sinc_filters(low, band, n_, window_, filt, *, C_, Kw, min_low, min_band, sr):
    _check(_lib.lib().pase_sinc_filters(_ptr(low), _ptr(band), _ptr(n_), _ptr(window_), _ptr(filt), C_, Kw,
                                        min_low, min_band, sr, _stream()), "pase_sinc_filters")


def sinc_filters_bwd(low, band, n_, window_, dfilt, dlow, dband, *, C_, Kw, min_low, min_band, sr):
    _check(_lib.lib().pase_sinc_filters_bwd(_ptr(low), _ptr(band), _ptr(n_), _ptr(window_), _ptr(dfilt),
                                            _ptr(dlow), _ptr(dband), C_, Kw, min_low, min_band, sr, _stream()),
           "pase_sinc_filters_bwd")


def pack_dgrad(src, dst, *, R, O, k, st, s_red, s_out, s_k):
    _check(_lib.lib().pase_pack_dgrad(_ptr(src), _ptr(dst), R, O, k, st, s_red, s_out, s_k, _stream()),
           "pase_pack_dgrad")


def pack_dgrad_t(src, *, R, O, k, st, s_red, s_out, s_k):
    """K-major data-gradient / transposed-conv weight pack, ready to be conv_gemm's wt= operand.
    A 1x1 weight stored (R, O) row-major already IS that pack: it is returned as is when aligned."""
    taps_p = -(-k // st)
    if (k == 1 and st == 1 and s_red == O and s_out == 1 and O % 4 == 0 and src.data_ptr() % 16 == 0
            and src.is_contiguous()):
        return src.view(R, O)
    ldt = (st * O + 3) // 4 * 4
    dst = torch.empty(R * taps_p, ldt, device=src.device, dtype=torch.float32)
    _check(_lib.lib().pase_pack_dgrad_t(_ptr(src), _ptr(dst), R, O, k, st, s_red, s_out, s_k, ldt, _stream()),
           "pase_pack_dgrad_t")
    return dst


def adam_step(p, g, m, v, lr, step, *, beta1=0.9, beta2=0.999, eps=1e-8, grad_mul=1.0):
    _check(_lib.lib().pase_adam_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel(), _ptr(lr),
                                     _ptr(step, torch.int32), beta1, beta2, eps, grad_mul, _stream()),
           "pase_adam_step")


def step_tick(step):
    _check(_lib.lib().pase_step_tick(_ptr(step, torch.int32), _stream()), "pase_step_tick")


def delta_znorm(x, coef, mean, istd, out, *, B, D, F, Fo, order, x_ctot=None, x_coff=0):
    _check(_lib.lib().pase_delta_znorm(_ptr(x), _ptr(coef), _ptr(mean), _ptr(istd), _ptr(out), B, D, F, Fo, order,
                                       D if x_ctot is None else x_ctot, x_coff, _stream()), "pase_delta_znorm")


def power_to_db(x, y, umax, *, per_utt, B, amin=1e-10, ref_db=0.0, top_db=80.0):
    _check(_lib.lib().pase_power_to_db(_ptr(x), _ptr(y), _ptr(umax, torch.int32), per_utt, B, amin, ref_db, top_db,
                                       _stream()), "pase_power_to_db")


def zcr_rms(x, out, *, B, T, F, hop, win, out_ctot, out_coff):
    _check(_lib.lib().pase_zcr_rms(_ptr(x), _ptr(out), B, T, F, hop, win, out_ctot, out_coff, _stream()), "pase_zcr_rms")


def lf0_interp(f0, out, *, B, Fin, F, out_ctot, out_coff, f0_min):
    _check(_lib.lib().pase_lf0_interp(_ptr(f0), _ptr(out), B, Fin, F, out_ctot, out_coff, f0_min, _stream()),
           "pase_lf0_interp")


def swipe_accumulate(num, den2, mu, cand, S, *, B, nj, nfr, NC, F, frames_per_out):
    _check(_lib.lib().pase_swipe_accumulate(_ptr(num), _ptr(den2), _ptr(mu), _ptr(cand, torch.int32), _ptr(S), B, nj, nfr,
                                            NC, F, frames_per_out, _stream()), "pase_swipe_accumulate")


def swipe_pick(S, f0, strength, *, B, NC, F, log2_fmin, dlog2p, polyv, st):
    _check(_lib.lib().pase_swipe_pick(_ptr(S), _ptr(f0), _ptr(strength), B, NC, F, log2_fmin, dlog2p, polyv, st,
                                      _stream()), "pase_swipe_pick")


def frame_prep(x, y, *, B, T, hop, Q, padL, pad_mode, preemph=0.0):
    _check(_lib.lib().pase_frame_prep(_ptr(x), _ptr(y), B, T, hop, Q, padL, pad_mode, preemph, _stream()),
           "pase_frame_prep")


# ======================================================================================
# batch producer
# ======================================================================================
def chunk_gather(pool, off, length, src, beg, out, *, N, T):
    _check(_lib.lib().pase_chunk_gather(_ptr(pool), _ptr(off, torch.int64), _ptr(length, torch.int32),
                                        _ptr(src, torch.int32), _ptr(beg, torch.int32), _ptr(out), N, T, _stream()),
           "pase_chunk_gather")


def peak_scale(x, u, *, N, T):
    _check(_lib.lib().pase_peak_scale(_ptr(x), _ptr(u), N, T, _stream()), "pase_peak_scale")


def reverb(x, irs, ir_off, ir_len, ir_pmax, ir_idx, full, energies, *, B, T, max_ir_len):
    _check(_lib.lib().pase_reverb(_ptr(x), _ptr(irs), _ptr(ir_off, torch.int64), _ptr(ir_len, torch.int32),
                                  _ptr(ir_pmax, torch.int32), _ptr(ir_idx, torch.int32), _ptr(full),
                                  _ptr(energies, torch.float64), B, T, max_ir_len, _stream()), "pase_reverb")


def add_noise(x, npool, noff, nlen, nidx, nbeg, snr, *, B, T):
    _check(_lib.lib().pase_add_noise(_ptr(x), _ptr(npool), _ptr(noff, torch.int64), _ptr(nlen, torch.int32),
                                     _ptr(nidx, torch.int32), _ptr(nbeg, torch.int32), _ptr(snr), B, T, _stream()),
           "pase_add_noise")


def gammatone_blocks(x, coef, blocks, *, B, C_, T, g):
    _check(_lib.lib().pase_gammatone_blocks(_ptr(x), _ptr(coef, torch.float64), _ptr(blocks), B, C_, T, g, _stream()),
           "pase_gammatone_blocks")


def gammatone_frames(blocks, out, *, rows, T, g, nwin, hop, ncol, eps):
    _check(_lib.lib().pase_gammatone_frames(_ptr(blocks), _ptr(out), rows, T, g, nwin, hop, ncol, eps, _stream()),
           "pase_gammatone_frames")


def fir_distort(x, irs, ir_off, ir_len, ir_shift, ir_idx, full, energies, *, B, T, max_ir_len, trimmed_energy):
    _check(_lib.lib().pase_fir_distort(_ptr(x), _ptr(irs), _ptr(ir_off, torch.int64), _ptr(ir_len, torch.int32),
                                       _ptr(ir_shift, torch.int32), _ptr(ir_idx, torch.int32), _ptr(full),
                                       _ptr(energies, torch.float64), B, T, max_ir_len, trimmed_energy, _stream()),
           "pase_fir_distort")


def clip(x, factor, *, B, T):
    _check(_lib.lib().pase_clip(_ptr(x), _ptr(factor), B, T, _stream()), "pase_clip")


def overlap_gather(pool, off, length, src, beg, shift, out, *, B, T):
    _check(_lib.lib().pase_overlap_gather(_ptr(pool), _ptr(off, torch.int64), _ptr(length, torch.int32),
                                          _ptr(src, torch.int32), _ptr(beg, torch.int32), _ptr(shift, torch.int32),
                                          _ptr(out), B, T, _stream()), "pase_overlap_gather")


def zero_front(x, shift, *, B, T):
    _check(_lib.lib().pase_zero_front(_ptr(x), _ptr(shift, torch.int32), B, T, _stream()), "pase_zero_front")


def add_blocks(pairs):
    """dst += src for (dst, src) pairs of 2-D row-major blocks (unit inner stride; a row stride on either side) in ONE launch
    per 16 pairs (pase_add_blocks).  Returns False when a pair does not have that form (the caller falls back to torch)."""
    segs = []
    for dst, src in pairs:
        if dst.dim() != 2 or src.dim() != 2 or dst.shape != src.shape or dst.dtype != torch.float32 or src.dtype != torch.float32:
            return False
        if (dst.stride(1) != 1 and dst.shape[1] > 1) or (src.stride(1) != 1 and src.shape[1] > 1):
            return False
        if dst.device.type != _lib.device_type() or src.device.type != _lib.device_type():
            return False
        segs.append((src.data_ptr(), dst.data_ptr(), dst.shape[0], dst.shape[1], max(src.stride(0), src.shape[1]),
                     max(dst.stride(0), dst.shape[1])))
    for i in range(0, len(segs), 16):
        d = PaseAddBlocks()
        chunk = segs[i:i + 16]
        d.n = len(chunk)
        for k, (sp, dp, r, w_, sl, dl) in enumerate(chunk):
            d.seg[k].src, d.seg[k].dst, d.seg[k].rows, d.seg[k].width, d.seg[k].src_ld, d.seg[k].dst_ld = sp, dp, r, w_, sl, dl
        _check(_lib.lib().pase_add_blocks(C.byref(d), _stream()), "pase_add_blocks")
    return True


def commit_cols(sums, ld, C_, pairs):
    """pairs: up to three (grad buffer or None, column) -- g[c] += sums[c*ld + column]"""
    pairs = [(g, c) for g, c in pairs if g is not None]
    pairs += [(None, 0)] * (3 - len(pairs))
    (g0, c0), (g1, c1), (g2, c2) = pairs[:3]
    _check(_lib.lib().pase_commit_cols(_ptr(sums, torch.float64), ld, C_, _ptr(g0), c0, _ptr(g1), c1, _ptr(g2), c2,
                                       _stream()), "pase_commit_cols")
