"""Thin ctypes bindings: one Python function per C-ABI entry point of include/pase_amd.h.

Tensors are torch tensors used purely as device buffers (data_ptr + the current HIP stream);
every wrapper checks dtype / contiguity / device and raises on a non-zero return code.
"""
import ctypes as C

import torch

from . import _lib

PAD_ZERO, PAD_REFLECT = 0, 1
EPI_STORE, EPI_MSE_CTX = 0, 1

_fp = C.c_void_p


class PaseConvGemm(C.Structure):
    _fields_ = [
        ("x", _fp), ("w", _fp), ("y", _fp), ("bias", _fp),
        ("in_scale", _fp), ("in_shift", _fp), ("in_alpha", _fp),
        ("stat_part", _fp), ("label", _fp), ("grad_out", _fp), ("loss_acc", _fp),
        ("grad_scale", C.c_float),
        ("S", C.c_int), ("Cin", C.c_int), ("Tin", C.c_int), ("x_ctot", C.c_int), ("x_coff", C.c_int),
        ("M", C.c_int), ("K", C.c_int), ("ldw", C.c_int), ("taps", C.c_int), ("tap_major", C.c_int),
        ("stride", C.c_int), ("tapstep", C.c_int), ("padL", C.c_int), ("pad_mode", C.c_int),
        ("Ncols", C.c_int),
        ("y_ctot", C.c_int), ("y_coff", C.c_int), ("Cout_store", C.c_int), ("ps", C.c_int),
        ("poff", C.c_int), ("Tout", C.c_int),
        ("epilogue", C.c_int), ("r_ctx", C.c_int), ("label_D", C.c_int),
        ("tile_hint", C.c_int),
    ]


def declare(l):
    l.pase_conv_gemm.argtypes = [C.POINTER(PaseConvGemm), C.c_void_p]
    l.pase_conv_gemm.restype = C.c_int
    l.pase_conv_gemm_stat_tiles.argtypes = [C.c_int] * 4
    l.pase_conv_gemm_stat_tiles.restype = C.c_int
    l.pase_abi_sizeof.argtypes = [C.c_int]
    l.pase_abi_sizeof.restype = C.c_int
    if l.pase_abi_sizeof(0) != C.sizeof(PaseConvGemm):
        raise _lib.PaseLibraryError("ABI mismatch: PaseConvGemm")
    for name, args in _SIMPLE.items():
        fn = getattr(l, name)
        fn.argtypes = args
        fn.restype = C.c_int


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


def stat_tiles(M, S, Ncols, tile_hint=0):
    return _lib.lib().pase_conv_gemm_stat_tiles(M, S, Ncols, tile_hint)


def conv_gemm(x, w, y, *, S, Cin, Tin, M, K, taps, Ncols, Tout, ldw=None, bias=None,
              in_scale=None, in_shift=None, in_alpha=None, stat_part=None,
              x_ctot=None, x_coff=0, tap_major=0, stride=1, tapstep=1, padL=0, pad_mode=PAD_ZERO,
              y_ctot=None, y_coff=0, Cout_store=None, ps=1, poff=0,
              epilogue=EPI_STORE, label=None, grad_out=None, loss_acc=None, grad_scale=0.0,
              r_ctx=0, label_D=0, tile_hint=0):
    d = PaseConvGemm()
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
    _check(_lib.lib().pase_conv_gemm(C.byref(d), _stream()), "pase_conv_gemm")
