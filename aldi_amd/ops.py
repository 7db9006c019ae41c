"""Thin torch-tensor wrappers over the C ABI (device memory + stream plumbing only).

No arithmetic happens here: every function marshals pointers/sizes into one C-ABI call of
libaldi_hip.so on torch's current HIP stream.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Sequence

import torch

from . import _lib as L


def dtype_code(t: torch.dtype) -> int:
    if t == torch.float32:
        return L.F32
    if t == torch.bfloat16:
        return L.BF16
    raise TypeError(f"unsupported dtype {t}")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


_RAW_STREAM = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr() -> int:
    """raw HIP stream of torch's current stream (every launch asks: the C hook is ~10x cheaper than building a Stream object)"""
    if _RAW_STREAM is not None:
        return _RAW_STREAM(torch.cuda.current_device())
    return torch.cuda.current_stream().cuda_stream


def upload_packed(tensors, device):
    """several small host tensors -> ONE pinned staging buffer -> one asynchronous copy; returns device views.
    (A `.to(device)` of pageable memory is a synchronous hipMemcpy per tensor.)"""
    metas, off = [], 0
    for t in tensors:
        nb = t.numel() * t.element_size()
        metas.append((off, nb))
        off += (nb + 15) // 16 * 16
    host = torch.empty(max(off, 16), dtype=torch.uint8).pin_memory()
    for t, (o, nb) in zip(tensors, metas):
        if nb:
            host[o:o + nb] = t.contiguous().view(-1).view(torch.uint8)
    dev = host.to(device, non_blocking=True)
    return [dev[o:o + nb].view(t.dtype).view(t.shape) for t, (o, nb) in zip(tensors, metas)]


# ------------------------------------------------------------------------------- dense path
def conv2d(x: torch.Tensor, w: torch.Tensor, *, stride: int = 1, pad: int = 0,
           scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None,
           res: Optional[torch.Tensor] = None, res_mode: int = 0, relu: bool = False,
           mask: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None,
           out_f32: Optional[torch.Tensor] = None, want_f32: bool = False,
           out_scale: int = 1, out_hw=None, ksplit: Optional[int] = None, mask_bits: Optional[torch.Tensor] = None,
           bits_out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [N,H,W,Cin] (NHWC), w [Cout,KH,KW,Cin] -> y [N,Ho,Wo,Cout] (or the scattered
    [N,OH,OW,Cout] tensor when out_scale > 1, which must be pre-zeroed by the caller).
    ksplit: K slices of a long-K linear layer (None: chosen here -- the box head's FC1 is 128 tiles of 128 x 128 over K = 12544)."""
    N, H, W_, Cin = x.shape
    Cout, KH, KW, Cin2 = w.shape
    assert Cin == Cin2 and x.is_contiguous() and w.is_contiguous() and x.dtype == w.dtype, (x.shape, w.shape, x.dtype, w.dtype)
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W_ + 2 * pad - KW) // stride + 1
    if out_scale > 1:
        OH, OW = out_hw
        shape = (N, OH, OW, Cout)
    else:
        OH = OW = 0
        shape = (N, Ho, Wo, Cout)
    if out is None and not want_f32:
        out = torch.empty(shape, dtype=x.dtype, device=x.device)
    if want_f32 and out_f32 is None:
        out_f32 = torch.empty(shape, dtype=torch.float32, device=x.device)
    ws = None
    if ksplit is None:
        ksplit = _auto_ksplit(x, N * Ho * Wo, Cout, KH * KW * Cin, KH * KW == 1 and stride == 1 and pad == 0 and res is None and mask is None
                              and not want_f32 and out_f32 is None and out_scale == 1 and mask_bits is None and bits_out is None)
    if ksplit and ksplit > 1:
        ws = torch.empty(ksplit * N * Ho * Wo * Cout, dtype=torch.float32, device=x.device)
    a = L.ConvArgs(_p(x), _p(w), _p(out), _p(out_f32), _p(scale), _p(shift), _p(res), _p(mask),
                   N, H, W_, Cin, Cout, KH, KW, stride, pad, Ho, Wo,
                   int(relu), res_mode, out_scale, OH, OW, dtype_code(x.dtype), _p(ws), int(ksplit or 0), _p(mask_bits), _p(bits_out))
    L.call("aldi_conv_igemm", C.byref(a), stream_ptr())
    return out_f32 if want_f32 and out is None else out


def _auto_ksplit(x, M, Cout, K, plain: bool) -> int:
    """K slices for a bf16 linear layer with a very long K and fewer output tiles than CUs (ALDI_SPLITK=0: never): as many as give
    the 256 x 128-tile launch about one 8-wave workgroup per CU (tools/fc1_splitk_sweep.py: FC1 at 2048 rows 98 us unsplit, 72 us with
    4 slices; at 1024 rows 92 -> 49 us with 7)"""
    if not plain or x.dtype != torch.bfloat16 or K < 4096 or K % 256 or os.environ.get("ALDI_SPLITK", "1") != "1":
        return 0
    tiles = ((M + 127) // 128) * ((Cout + 127) // 128)
    if not (16 <= tiles <= 160) or Cout % 4:
        return 0
    tiles256 = ((M + 255) // 256) * ((Cout + 127) // 128)
    for ks in (2, 4, 7, 8, 14, 16):
        if K % (64 * ks) == 0 and tiles256 * ks >= 216:
            return ks
    return 4 if K % 256 == 0 else 0


def _conv_args(x, w, *, stride=1, pad=0, scale=None, shift=None, res=None, res_mode=0, relu=False, mask=None, out=None, out_f32=None,
               want_f32=False, out_scale=1, out_hw=None, mask_bits=None, bits_out=None):
    """(ConvArgs, result tensor) of one conv2d call -- shared by the single and the grouped launch"""
    N, H, W_, Cin = x.shape
    Cout, KH, KW, Cin2 = w.shape
    assert Cin == Cin2 and x.is_contiguous() and w.is_contiguous() and x.dtype == w.dtype, (x.shape, w.shape, x.dtype, w.dtype)
    Ho = (H + 2 * pad - KH) // stride + 1
    Wo = (W_ + 2 * pad - KW) // stride + 1
    if out_scale > 1:
        OH, OW = out_hw
        shape = (N, OH, OW, Cout)
    else:
        OH = OW = 0
        shape = (N, Ho, Wo, Cout)
    if out is None and not want_f32:
        out = torch.empty(shape, dtype=x.dtype, device=x.device)
    if want_f32 and out_f32 is None:
        out_f32 = torch.empty(shape, dtype=torch.float32, device=x.device)
    a = L.ConvArgs(_p(x), _p(w), _p(out), _p(out_f32), _p(scale), _p(shift), _p(res), _p(mask),
                   N, H, W_, Cin, Cout, KH, KW, stride, pad, Ho, Wo,
                   int(relu), res_mode, out_scale, OH, OW, dtype_code(x.dtype), None, 0, _p(mask_bits), _p(bits_out))
    return a, (out_f32 if want_f32 and out is None else out)


def conv2d_group(calls) -> List[torch.Tensor]:
    """calls = [(x, w, kwargs-of-conv2d)]: the same layer applied to several batches with several weight sets (student / teacher)
    in ONE launch (aldi_conv_igemm_group; anything else than one common layer shape degrades to single launches inside)"""
    arr = (L.ConvArgs * len(calls))()
    outs = []
    for i, (x, w, kw) in enumerate(calls):
        arr[i], y = _conv_args(x, w, **kw)
        outs.append(y)
    L.call("aldi_conv_igemm_group", arr, len(calls), stream_ptr())
    return outs


_WGRAD_WS = {}          # (device, stream handle) -> fp32 workspace of the ordered weight-gradient epilogue (grow only)
WGRAD_WS_EPOCH = 0      # bumped whenever a workspace is REPLACED by a larger one: hipGraphs recorded before hold the old buffer's address
WGRAD_ORDERED = os.environ.get("ALDI_WGRAD_ORDERED", "1") != "0"


def _wgrad_workspace(arr, n: int, device) -> None:
    """give the call's first problem the workspace its ordered epilogue needs (aldi_conv_wgrad_group_workspace); launches on one
    stream are ordered, so one buffer per stream serves all of them"""
    if not WGRAD_ORDERED:
        return
    need = L.lib.aldi_conv_wgrad_group_workspace(arr, n)
    if need < 0:
        raise RuntimeError("aldi_conv_wgrad_group_workspace: " + L.lib.aldi_last_error().decode())
    if need == 0:
        need = 256          # a non-NULL workspace selects the ordered epilogue (plain read-modify-write of unsplit tiles)
    key = (device, stream_ptr())
    buf = _WGRAD_WS.get(key)
    if buf is None or buf.numel() * 4 < need:
        global WGRAD_WS_EPOCH
        if buf is not None:
            # a larger problem set (another image size, a new chunk layout) outgrew the buffer: graphs recorded with the old one would
            # replay into freed memory -- the fused step compares this epoch with the one its graphs were recorded at and re-records
            WGRAD_WS_EPOCH += 1
            if torch.cuda.is_current_stream_capturing():
                # (the eager warm-up steps size the buffer; outgrowing it DURING a recording means the shapes changed under the capture)
                raise RuntimeError("the weight-gradient workspace has to grow during a hipGraph capture: run the new shapes eagerly once first")
        else:
            # a stream's first buffer (e.g. the capture stream of a step graph) starts at the size the other streams' buffers have reached: the
            # eager warm-up steps have seen the step's largest problem there, the first call on this stream need not be it
            need = max([need] + [b.numel() * 4 for (d_, _), b in _WGRAD_WS.items() if d_ == device])
        buf = torch.empty((need + (need >> 2) + 3) // 4, dtype=torch.float32, device=device)
        _WGRAD_WS[key] = buf
    arr[0].ws = buf.data_ptr()
    arr[0].ws_bytes = buf.numel() * 4


def conv_wgrad(x: torch.Tensor, g: torch.Tensor, dw: torch.Tensor, *, KH: int, KW: int, stride: int = 1, pad: int = 0,
               scale: Optional[torch.Tensor] = None, db: Optional[torch.Tensor] = None) -> None:
    """dw [Cout,KH,KW,Cin] fp32 += scale * (g^T . im2col(x)); x [N,H,W,Cin], g [N,Ho,Wo,Cout]; db [Cout] fp32 += column sums of g."""
    N, H, W_, Cin = x.shape
    _, Ho, Wo, Cout = g.shape
    assert dw.dtype == torch.float32 and dw.numel() == Cout * KH * KW * Cin and x.dtype == g.dtype, (dw.shape, x.shape, g.shape)
    arr = (L.WgradArgs * 1)()
    arr[0] = L.WgradArgs(_p(x), _p(g), _p(dw), _p(scale), N, H, W_, Cin, Cout, KH, KW, stride, pad, Ho, Wo, dtype_code(x.dtype), _p(db), None, 0)
    if x.dtype == torch.bfloat16:
        _wgrad_workspace(arr, 1, x.device)
    L.call("aldi_conv_wgrad", arr, stream_ptr())


def conv_wgrad_group(problems) -> None:
    """several weight gradients in one launch (aldi_conv_wgrad_group); problems = [(x, g, dw, dict(KH, KW, stride, pad, scale))]"""
    arr = (L.WgradArgs * len(problems))()
    for i, (x, g, dw, kw) in enumerate(problems):
        N, H, W_, Cin = x.shape
        _, Ho, Wo, Cout = g.shape
        KH, KW = kw["KH"], kw["KW"]
        assert dw.dtype == torch.float32 and dw.numel() == Cout * KH * KW * Cin and x.dtype == g.dtype, (dw.shape, x.shape, g.shape)
        arr[i] = L.WgradArgs(_p(x), _p(g), _p(dw), _p(kw.get("scale")), N, H, W_, Cin, Cout, KH, KW, kw.get("stride", 1), kw.get("pad", 0), Ho, Wo,
                             dtype_code(x.dtype), _p(kw.get("db")), None, 0)
    _wgrad_workspace(arr, len(problems), problems[0][0].device)
    L.call("aldi_conv_wgrad_group", arr, len(problems), stream_ptr())


def noop() -> None:
    """an empty one-workgroup launch on the current stream (aldi_noop): a marker in kernel traces"""
    L.call("aldi_noop", stream_ptr())


def bias_grad(g: torch.Tensor, db: torch.Tensor) -> None:
    Cc = g.shape[-1]
    L.call("aldi_bias_grad", _p(g), _p(db), g.numel() // Cc, Cc, dtype_code(g.dtype), stream_ptr())


def dgrad_weights(w_master: torch.Tensor, scale: Optional[torch.Tensor], dtype: torch.dtype,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """w_master fp32 [Cout,KH,KW,Cin] -> [Cin,KH,KW,Cout] rotated/transposed (x scale[co]) in `dtype`."""
    Cout, KH, KW, Cin = w_master.shape
    if out is None:
        out = torch.empty((Cin, KH, KW, Cout), dtype=dtype, device=w_master.device)
    L.call("aldi_dgrad_weights", _p(w_master), _p(scale), _p(out), Cout, KH, KW, Cin, dtype_code(dtype), stream_ptr())
    return out


class DgradWeightsPlan:
    """All data-gradient weights of a model in one launch (aldi_dgrad_weights_batch).  `entries` = [(w_master, scale|None)];
    outputs live in one persistent buffer (`out[i]` = [Cin,KH,KW,Cout] view).  Inputs must keep their storage (they are
    views of the flat master / scale buffers)."""

    def __init__(self, entries, dtype: torch.dtype):
        dev = entries[0][0].device
        self.dtype = dtype
        self.keep = list(entries)
        sizes = [w.numel() for w, _ in entries]
        offs, tot = [], 0
        for n in sizes:
            offs.append(tot)
            tot += (n + 63) // 64 * 64
        self.buf = torch.empty(tot, dtype=dtype, device=dev)
        self.out = []
        items = (L.DgwItem * len(entries))()
        tile = 0
        for i, (w, sc) in enumerate(entries):
            Cout, KH, KW, Cin = w.shape
            o = self.buf[offs[i]:offs[i] + sizes[i]].view(Cin, KH, KW, Cout)
            self.out.append(o)
            it = items[i]
            it.w_master, it.scale, it.wt = w.data_ptr(), (sc.data_ptr() if sc is not None else None), o.data_ptr()
            it.Cout, it.KH, it.KW, it.Cin, it.tile_begin, it.reserved = Cout, KH, KW, Cin, tile, 0
            tile += KH * KW * ((Cout + 31) // 32) * ((Cin + 31) // 32)
        self.total_tiles = tile
        self.n = len(entries)
        raw = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8)
        self.items = raw.to(dev)

    def run(self):
        L.call("aldi_dgrad_weights_batch", _p(self.items), self.n, self.total_tiles, dtype_code(self.dtype), stream_ptr())


class FoldWeightsPlan:
    """bf16 copies of conv weights with the FrozenBN scale folded into their rows, all in one launch (aldi_fold_weights_batch).
    `entries` = [(w_master fp32 view [Cout, ...], scale [Cout] | None)]; `out[i]` has w_master's shape.  Inputs keep their storage."""

    def __init__(self, entries):
        dev = entries[0][0].device
        self.keep = list(entries)
        sizes = [w.numel() for w, _ in entries]
        offs, tot = [], 0
        for n in sizes:
            assert n % 8 == 0
            offs.append(tot)
            tot += (n + 63) // 64 * 64
        self.buf = torch.empty(tot, dtype=torch.bfloat16, device=dev)
        self.out = []
        items = (L.FoldItem * len(entries))()
        chunk = 0
        for i, (w, sc) in enumerate(entries):
            assert w.dtype == torch.float32 and w.is_contiguous() and w.data_ptr() % 16 == 0, "fold_weights: fp32, contiguous, 16-byte aligned"
            o = self.buf[offs[i]:offs[i] + sizes[i]].view(w.shape)
            self.out.append(o)
            it = items[i]
            it.w, it.scale, it.out = w.data_ptr(), (sc.data_ptr() if sc is not None else None), o.data_ptr()
            it.rows, it.cols, it.chunk_begin, it.reserved = w.shape[0], sizes[i] // w.shape[0], chunk, 0
            chunk += sizes[i] // 8
        self.total_chunks, self.n = chunk, len(entries)
        self.items = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(dev)

    def run(self):
        L.call("aldi_fold_weights_batch", _p(self.items), self.n, self.total_chunks, stream_ptr())


def bottleneck_fused(x: torch.Tensor, res: torch.Tensor, w1, w2, w3, b1, b2, b3, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """one ResNet bottleneck without saved activations in one kernel (aldi_bottleneck_fused): x [N,H,W,Cin] bf16, res [N,H,W,Cout]
    (x itself for an identity block), folded bf16 weights w1 [mid,1,1,Cin], w2 [mid,3,3,mid], w3 [Cout,1,1,mid], fp32 shifts"""
    N, H, W_, Cin = x.shape
    mid, Cout = w1.shape[0], w3.shape[0]
    assert x.dtype == torch.bfloat16 and x.is_contiguous() and res.is_contiguous() and res.shape == (N, H, W_, Cout), (x.shape, res.shape)
    assert w1.numel() == mid * Cin and w2.numel() == mid * 9 * mid and w3.numel() == Cout * mid and w1.dtype == torch.bfloat16
    if out is None:
        out = torch.empty((N, H, W_, Cout), dtype=x.dtype, device=x.device)
    a = L.BottleneckArgs(_p(x), _p(res), _p(out), _p(w1), _p(w2), _p(w3), _p(b1), _p(b2), _p(b3), N, H, W_, Cin, mid, Cout)
    L.call("aldi_bottleneck_fused", C.byref(a), stream_ptr())
    return out


# ------------------------------------------------------------------------------- stem / glue
def stem_forward(img_u8: torch.Tensor, sizes: Sequence[Sequence[int]], w: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor,
                 mean: Sequence[float], std: Sequence[float], dtype: torch.dtype) -> torch.Tensor:
    """img_u8 [N,3,Hs,Ws] uint8 staging (Hs,Ws already padded to /32) -> [N,Hs/2,Ws/2,64]."""
    N, _, Hs, Ws = img_u8.shape
    Hc, Wc = Hs // 2, Ws // 2
    y = torch.empty((N, Hc, Wc, 64), dtype=dtype, device=img_u8.device)
    a = L.StemArgs()
    a.img, a.w, a.scale, a.shift, a.y = _p(img_u8), _p(w), _p(scale), _p(shift), _p(y)
    a.N, a.Hs, a.Ws, a.Hc, a.Wc = N, Hs, Ws, Hc, Wc
    for i, (h, w_) in enumerate(sizes):
        a.h[i], a.w_img[i] = int(h), int(w_)
    for c in range(3):
        a.mean[c], a.std[c] = float(mean[c]), float(std[c])
    a.dtype = dtype_code(dtype)
    L.call("aldi_stem_forward", C.byref(a), stream_ptr())
    return y


def stem_pack_weights(w: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp32 [64,7,7,3] stem kernel -> bf16 [64,200] in the MFMA reduction order (aldi_stem_pack_weights)"""
    if out is None:
        out = torch.empty((64, 200), dtype=torch.bfloat16, device=w.device)
    L.call("aldi_stem_pack_weights", _p(w), _p(out), stream_ptr())
    return out


def stem_pool_forward(img_u8: torch.Tensor, sizes: Sequence[Sequence[int]], w_packed: torch.Tensor, scale: torch.Tensor, shift: torch.Tensor,
                      mean: Sequence[float], std: Sequence[float]) -> torch.Tensor:
    """stem conv + FrozenBN + ReLU + max_pool2d(3, 2, 1) in one launch (bf16): [N,3,Hs,Ws] uint8 -> [N,Hs/4,Ws/4,64]"""
    N, _, Hs, Ws = img_u8.shape
    Hc, Wc = Hs // 2, Ws // 2
    y = torch.empty((N, (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1, 64), dtype=torch.bfloat16, device=img_u8.device)
    a = L.StemArgs()
    a.img, a.w, a.scale, a.shift, a.y = _p(img_u8), None, _p(scale), _p(shift), None
    a.N, a.Hs, a.Ws, a.Hc, a.Wc = N, Hs, Ws, Hc, Wc
    for i, (h, w_) in enumerate(sizes):
        a.h[i], a.w_img[i] = int(h), int(w_)
    for c in range(3):
        a.mean[c], a.std[c] = float(mean[c]), float(std[c])
    a.dtype = L.BF16
    L.call("aldi_stem_pool_forward", C.byref(a), _p(w_packed), _p(y), stream_ptr())
    return y


def maxpool3s2(x: torch.Tensor) -> torch.Tensor:
    N, H, W_, Cc = x.shape
    y = torch.empty((N, (H - 1) // 2 + 1, (W_ - 1) // 2 + 1, Cc), dtype=x.dtype, device=x.device)
    L.call("aldi_maxpool3s2", _p(x), _p(y), N, H, W_, Cc, dtype_code(x.dtype), stream_ptr())
    return y


def subsample2(x: torch.Tensor) -> torch.Tensor:
    N, H, W_, Cc = x.shape
    y = torch.empty((N, (H - 1) // 2 + 1, (W_ - 1) // 2 + 1, Cc), dtype=x.dtype, device=x.device)
    L.call("aldi_subsample2", _p(x), _p(y), N, H, W_, Cc, 0, dtype_code(x.dtype), stream_ptr())
    return y


def subsample2_bwd(g_small: torch.Tensor, g_big: torch.Tensor) -> None:
    N, H, W_, Cc = g_big.shape
    L.call("aldi_subsample2", _p(g_small), _p(g_big), N, H, W_, Cc, 1, dtype_code(g_big.dtype), stream_ptr())


def upsample2_bwd(g: torch.Tensor, out: torch.Tensor, accumulate: bool) -> None:
    N, Hc, Wc, Cc = out.shape
    assert g.shape == (N, Hc * 2, Wc * 2, Cc)
    L.call("aldi_upsample2_bwd", _p(g), _p(out), N, Hc, Wc, Cc, int(accumulate), dtype_code(g.dtype), stream_ptr())


def upsample2_bwd_chain(g2: torch.Tensor, o3: torch.Tensor, o4: torch.Tensor, o5: torch.Tensor) -> None:
    """o3 += blocks(g2); o4 += blocks(o3); o5 += blocks(o4) in one launch (aldi_upsample2_bwd_chain)"""
    N, H5, W5, Cc = o5.shape
    assert o4.shape == (N, H5 * 2, W5 * 2, Cc) and o3.shape == (N, H5 * 4, W5 * 4, Cc) and g2.shape == (N, H5 * 8, W5 * 8, Cc)
    L.call("aldi_upsample2_bwd_chain", _p(g2), _p(o3), _p(o4), _p(o5), N, H5, W5, Cc, dtype_code(g2.dtype), stream_ptr())


def add_f32(a: Optional[torch.Tensor], b: Optional[torch.Tensor], out: torch.Tensor, relu_src: Optional[torch.Tensor] = None) -> torch.Tensor:
    L.call("aldi_add_f32", _p(a), _p(b), _p(relu_src), _p(out), out.numel(), dtype_code(out.dtype), stream_ptr())
    return out


def cast_from_f32(src: torch.Tensor, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if out is None:
        out = torch.empty(src.shape, dtype=dtype, device=src.device)
    L.call("aldi_cast_from_f32", _p(src), _p(out), src.numel(), dtype_code(dtype), stream_ptr())
    return out


def sgd_step(p, g, buf, p_compute, n, lr, momentum, weight_decay, grad_scale, first_step, dtype) -> None:
    L.call("aldi_sgd_step", _p(p), _p(g), _p(buf), _p(p_compute), n, lr, momentum, weight_decay, grad_scale, int(first_step),
           dtype_code(dtype), stream_ptr())


def sgd_step_dev(p, g, buf, p_compute, lo: int, hi: int, hyper: torch.Tensor, dtype) -> None:
    """SGD over elements [lo, hi) of the flat buffers, scalars (lr, momentum, weight decay, gradient scale) from the device tensor `hyper`"""
    pc = None if p_compute is None else p_compute.data_ptr() + lo * p_compute.element_size()
    L.call("aldi_sgd_step_dev", p.data_ptr() + 4 * lo, g.data_ptr() + 4 * lo, buf.data_ptr() + 4 * lo, pc, hi - lo, _p(hyper), dtype_code(dtype), stream_ptr())


def ema_update(teacher, student, teacher_compute, n, alpha, copy_only, dtype, n_compute=None) -> None:
    nc = 0 if teacher_compute is None else (teacher_compute.numel() if n_compute is None else n_compute)
    L.call("aldi_ema_update", _p(teacher), _p(student), _p(teacher_compute), n, nc, alpha, int(copy_only), dtype_code(dtype), stream_ptr())


def bn_fold(w, b, mean, var, scale, shift, Cc) -> None:
    L.call("aldi_bn_fold", _p(w), _p(b), _p(mean), _p(var), _p(scale), _p(shift), Cc, stream_ptr())


# ------------------------------------------------------------------------------- RPN
def make_geom(shapes: Sequence[Sequence[int]], A: int, Cc: int) -> L.RpnGeom:
    g = L.RpnGeom()
    g.num_levels, g.A, g.C = len(shapes), A, Cc
    off = 0
    for i, (h, w) in enumerate(shapes):
        g.H[i], g.W[i], g.off[i] = h, w, off
        off += h * w * A
    for i in range(len(shapes), L.MAX_LEVELS + 1):
        g.off[i] = off
    return g


def ptrs(ts: Optional[Sequence[Optional[torch.Tensor]]]):
    if ts is None:
        return None
    arr = L.PtrArray5()
    for i, t in enumerate(ts):
        arr[i] = _p(t)
    return arr


def box_match(boxes, box_stride_n, box_count, Lb, gt_boxes, gt_count, Gmax, N, lo, hi, lowq, best_iou, best_idx, scratch, labels):
    """scratch: >= N * Gmax int32 words; N * Gmax * 32 words give every GT's best-IoU word its own 128-byte line (box_match_scratch)"""
    L.call("aldi_box_match", _p(boxes), box_stride_n, _p(box_count), Lb, _p(gt_boxes), _p(gt_count), Gmax, N, lo, hi, int(lowq),
           _p(best_iou), _p(best_idx), _p(scratch), scratch.numel() * scratch.element_size(), _p(labels), stream_ptr())


def box_match_scratch(N: int, Gmax: int, device) -> torch.Tensor:
    """the matcher's per-GT scratch in its padded layout (one 128-byte line per GT)"""
    return torch.empty((N, Gmax * 32), dtype=torch.int32, device=device)


def stage_images(images, batch: torch.Tensor) -> None:
    """uint8 CHW device images (their own sizes) -> batch[i, :, :h, :w], one launch (aldi_stage_images)"""
    n = len(images)
    ptr = (C.c_void_p * n)(*[_p(im) for im in images])
    hs = (C.c_int * n)(*[int(im.shape[1]) for im in images])
    ws = (C.c_int * n)(*[int(im.shape[2]) for im in images])
    L.call("aldi_stage_images", ptr, hs, ws, n, batch.shape[1], batch.shape[2], batch.shape[3], _p(batch), stream_ptr())


def compact_labels(labels, Lb, N, bg, lists, counts):
    ws = torch.empty(max(int(L.lib.aldi_compact_labels_workspace(Lb, N)), 16), dtype=torch.uint8, device=labels.device)
    L.call("aldi_compact_labels", _p(labels), Lb, N, bg, _p(lists), _p(counts), _p(ws), stream_ptr())


def rpn_apply_sample(labels, Lb, N, lists, sel, nsel, S):
    L.call("aldi_rpn_apply_sample", _p(labels), Lb, N, _p(lists), _p(sel), _p(nsel), S, stream_ptr())


def rpn_loss(geom, head, grad, anchors, labels, matched, gt_boxes, gt_count, Gmax, N, inv_norm, gs_cls, gs_loc, loss2):
    L.call("aldi_rpn_loss", C.byref(geom), ptrs(head), ptrs(grad), _p(anchors), _p(labels), _p(matched), _p(gt_boxes), _p(gt_count), Gmax, N,
           inv_norm, gs_cls, gs_loc, _p(loss2), stream_ptr())


def rpn_proposals_workspace(N, nl) -> int:
    return int(L.lib.aldi_rpn_proposals_workspace(N, nl))


def rpn_proposals(geom, head, anchors, img_hw, N, pre, post, thr, workspace, out_boxes, out_scores, out_count, err):
    L.call("aldi_rpn_proposals", C.byref(geom), ptrs(head), _p(anchors), _p(img_hw), N, pre, post, thr, _p(workspace),
           _p(out_boxes), _p(out_scores), _p(out_count), _p(err), stream_ptr())


def rpn_active_pixels_workspace(geom, N) -> int:
    return int(L.lib.aldi_rpn_active_pixels_workspace(C.byref(geom), N))


def rpn_active_pixels(geom, ghead, N, cap, idx, count, workspace, err):
    L.call("aldi_rpn_active_pixels", C.byref(geom), ptrs(ghead), N, cap, _p(idx), _p(count), _p(workspace), _p(err), stream_ptr())


def rpn_sparse_gather(geom, ghead, hidden, feat, N, Cf, cap, idx, count, G, Tm, X9):
    L.call("aldi_rpn_sparse_gather", C.byref(geom), ptrs(ghead), ptrs(hidden), ptrs(feat), N, Cf, cap, _p(idx), _p(count), _p(G), _p(Tm), _p(X9),
           dtype_code(G.dtype), stream_ptr())


def rpn_sparse_scatter(geom, gfeat, Y, N, Cf, cap, idx, count):
    L.call("aldi_rpn_sparse_scatter", C.byref(geom), ptrs(gfeat), _p(Y), N, Cf, cap, _p(idx), _p(count), dtype_code(Y.dtype),
           dtype_code(gfeat[0].dtype), stream_ptr())


# ------------------------------------------------------------------------------- ROI heads
def make_roi_feats(feats: Sequence[torch.Tensor], grads: Optional[Sequence[torch.Tensor]], scales: Sequence[float]) -> L.RoiFeats:
    f = L.RoiFeats()
    for i, t in enumerate(feats):
        f.feat[i] = _p(t)
        f.grad[i] = _p(grads[i]) if grads is not None else None
        f.H[i], f.W[i] = t.shape[1], t.shape[2]
        f.scale[i] = scales[i]
    f.C = feats[0].shape[3]
    return f


def roi_prepare(props, pcount, P, gt_boxes, gt_classes, gt_count, Gmax, N, K, thr, cand, ccount, best_iou, best_idx, scratch, labels, cls):
    L.call("aldi_roi_prepare", _p(props), _p(pcount), P, _p(gt_boxes), _p(gt_classes), _p(gt_count), Gmax, N, K, thr, _p(cand), _p(ccount),
           _p(best_iou), _p(best_idx), _p(scratch), _p(labels), _p(cls), stream_ptr())


def roi_prepare_lists(props, pcount, P, gt_boxes, gt_classes, gt_count, Gmax, N, K, thr, cand, ccount, best_iou, best_idx, labels, cls, lists, counts, tickets,
                      tail_a=None, tail_b=None):
    """roi_prepare + compact_labels(cls, bg = K) in one launch (aldi_roi_prepare_lists); tickets: N zero uint32 words, left zero;
    tail_a / tail_b: two int32 device words copied behind the counts (counts then holds 2N + 2 words)"""
    L.call("aldi_roi_prepare_lists", _p(props), _p(pcount), P, _p(gt_boxes), _p(gt_classes), _p(gt_count), Gmax, N, K, thr, _p(cand), _p(ccount),
           _p(best_iou), _p(best_idx), _p(labels), _p(cls), _p(lists), _p(counts), _p(tickets), _p(tail_a), _p(tail_b), stream_ptr())


def roi_gather(cand, cls, best_idx, Lb, lists, sel, nsel, S, row_off, gt_boxes, gt_count, Gmax, N, rois, r_cls, r_gt, r_idx):
    L.call("aldi_roi_gather", _p(cand), _p(cls), _p(best_idx), Lb, _p(lists), _p(sel), _p(nsel), S, _p(row_off), _p(gt_boxes), _p(gt_count),
           Gmax, N, _p(rois), _p(r_cls), _p(r_gt), _p(r_idx), stream_ptr())


def rois_from_proposals(props, pcount, P, N, rois):
    L.call("aldi_rois_from_proposals", _p(props), _p(pcount), P, N, _p(rois), stream_ptr())


def roialign(feats: L.RoiFeats, rois, R, P, pooled, backward: bool):
    L.call("aldi_roialign", C.byref(feats), _p(rois), R, P, _p(pooled), int(backward), dtype_code(pooled.dtype), stream_ptr())


def roialign_backward(feats: L.RoiFeats, rois, R, P, g_pooled, N, rois_sorted: bool = False, grad_dtype=torch.float32):
    """gather form: overwrites the gradient maps of `feats` (fp32, or bf16 with bf16 pooled gradients; each element written once, no atomics)"""
    L.call("aldi_roialign_backward", C.byref(feats), _p(rois), R, P, _p(g_pooled), N, int(rois_sorted), dtype_code(g_pooled.dtype),
           dtype_code(grad_dtype), stream_ptr())


def box_loss(pred, Cp, K, R, rois, cls, gt_boxes, weights4, gs_cls, gs_box, grad, loss2):
    w = (C.c_float * 4)(*weights4)
    L.call("aldi_box_loss", _p(pred), Cp, K, R, _p(rois), _p(cls), _p(gt_boxes), w, gs_cls, gs_box, _p(grad), _p(loss2), stream_ptr())


def detections_workspace(N) -> int:
    return int(L.lib.aldi_detections_workspace(N))


def detections(pred, Cp, K, props, pcount, P, N, img_hw, weights4, score_thresh, nms_thresh, topk, pl_thresh, workspace,
               det_boxes, det_scores, det_cls, det_count, pl_boxes, pl_cls, pl_scores, pl_count, err):
    w = (C.c_float * 4)(*weights4)
    L.call("aldi_detections", _p(pred), Cp, K, _p(props), _p(pcount), P, N, _p(img_hw), w, score_thresh, nms_thresh, topk, pl_thresh,
           _p(workspace), _p(det_boxes), _p(det_scores), _p(det_cls), _p(det_count), _p(pl_boxes), _p(pl_cls), _p(pl_scores), _p(pl_count),
           pl_boxes.shape[1], _p(err), stream_ptr())


# ------------------------------------------------------------------------------- ALDI losses
def rpn_distill_loss(geom, s_head, t_head, grad, labels, N, obj_T, n_valid, n_fg, do_obj, do_reg, grad_scale, loss2, counts_dev=None):
    """counts_dev: device int32[2] {n_valid, n_fg} overriding the two host ints (graph-captured steps)"""
    L.call("aldi_rpn_distill_loss", C.byref(geom), ptrs(s_head), ptrs(t_head), ptrs(grad), _p(labels), N, obj_T, int(n_valid), int(n_fg),
           _p(counts_dev), int(do_obj), int(do_reg), grad_scale, _p(loss2), stream_ptr())


def box_losses_fused(pred, Cp, K, rois, cls, gt_boxes, weights4, chunks, grad, grad_lo):
    """aldi_box_losses_fused: chunks = [dict(r0, r1, gs_cls, gs_box, loss_box, t_pred=None, cls_T=1.0, kl=False, do_cls=False, do_reg=False,
    gs_dcls=0.0, gs_dreg=0.0, loss_d=None)] -- box_loss per chunk + roih_distill_loss where t_pred is given + the bf16 copy of the rows"""
    arr = (L.BoxLossChunk * len(chunks))()
    for i, q in enumerate(chunks):
        arr[i] = L.BoxLossChunk(q["r0"], q["r1"], q["gs_cls"], q["gs_box"], _p(q["loss_box"]), _p(q.get("t_pred")), float(q.get("cls_T", 1.0)),
                                int(bool(q.get("kl"))), int(bool(q.get("do_cls"))), int(bool(q.get("do_reg"))), float(q.get("gs_dcls", 0.0)),
                                float(q.get("gs_dreg", 0.0)), _p(q.get("loss_d")))
    w = (C.c_float * 4)(*weights4)
    L.call("aldi_box_losses_fused", _p(pred), Cp, K, _p(rois), _p(cls), _p(gt_boxes), w, arr, len(chunks), _p(grad), _p(grad_lo), stream_ptr())


def roih_distill_loss(s_pred, t_pred, Cp, K, R, cls_T, kl, do_cls, do_reg, grad_scale, grad, loss2):
    L.call("aldi_roih_distill_loss", _p(s_pred), _p(t_pred), Cp, K, R, cls_T, int(kl), int(do_cls), int(do_reg), grad_scale, _p(grad),
           _p(loss2), stream_ptr())


def domain_bce(pred, ld, R, label, weight, grad_scale, grad, loss):
    L.call("aldi_domain_bce", _p(pred), ld, R, label, weight, grad_scale, _p(grad), _p(loss),
           dtype_code(grad.dtype) if grad is not None else L.F32, stream_ptr())


def avgpool(x: torch.Tensor) -> torch.Tensor:
    N, H, W_, Cc = x.shape
    y = torch.empty((N, 1, 1, Cc), dtype=x.dtype, device=x.device)
    ws = torch.empty(max(int(L.lib.aldi_avgpool_workspace(N, Cc)), 16), dtype=torch.uint8, device=x.device)
    L.call("aldi_avgpool", _p(x), _p(y), N, H * W_, Cc, dtype_code(x.dtype), _p(ws), stream_ptr())
    return y


def avgpool_bwd(gy: torch.Tensor, act: torch.Tensor) -> torch.Tensor:
    N, H, W_, Cc = act.shape
    gx = torch.empty_like(act)
    L.call("aldi_avgpool_bwd", _p(gy), _p(act), _p(gx), N, H * W_, Cc, dtype_code(act.dtype), stream_ptr())
    return gx
