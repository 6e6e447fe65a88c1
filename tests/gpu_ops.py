"""Thin test-side callers of the per-block C-ABI entry points (tests only)."""
import ctypes as C

import torch

from l3ac_amd import _capi


def _stream(dev):
    return torch.cuda.current_stream(dev).cuda_stream


def to_frames(x_bct: torch.Tensor) -> torch.Tensor:
    """reference layout (B, C, T) -> frame-major (B, T, C) contiguous on the GPU."""
    return x_bct.permute(0, 2, 1).contiguous().cuda()


def from_frames(y_btc: torch.Tensor) -> torch.Tensor:
    return y_btc.permute(0, 2, 1).contiguous().cpu()


def op_block(ctx, fn_name, block, x_btc, out_shape):
    y = torch.empty(out_shape, dtype=torch.float32, device=x_btc.device)
    b, frames = x_btc.shape[0], x_btc.shape[1]
    fn = getattr(ctx.lib, fn_name)
    _capi.check(fn(ctx.handle, block.encode(), x_btc.data_ptr(), b, frames, y.data_ptr(), _stream(x_btc.device)))
    torch.cuda.synchronize()
    return y


def op_block2(ctx, fn_name, block_a, block_b, x_btc, out_shape, out=None):
    y = torch.empty(out_shape, dtype=torch.float32, device=x_btc.device) if out is None else out
    b, frames = x_btc.shape[0], x_btc.shape[1]
    fn = getattr(ctx.lib, fn_name)
    _capi.check(fn(ctx.handle, block_a.encode(), block_b.encode(), x_btc.data_ptr(), b, frames, y.data_ptr(), _stream(x_btc.device)))
    torch.cuda.synchronize()
    return y


def op_plain(ctx, fn_name, x, d1, d2, out_shape, out_dtype=torch.float32):
    y = torch.empty(out_shape, dtype=out_dtype, device=x.device)
    fn = getattr(ctx.lib, fn_name)
    _capi.check(fn(ctx.handle, x.data_ptr(), d1, d2, y.data_ptr(), _stream(x.device)))
    torch.cuda.synchronize()
    return y


def fsq_forward(x, levels, w_in, b_in, w_out, b_out, latents_in=None, want_latents=False):
    lib = _capi.load_library()
    dev = w_out.device
    feat = w_out.shape[0]
    d = len(levels)
    n = (x if x is not None else latents_in).reshape(-1, feat if x is not None else d).shape[0]
    q = torch.empty((n, feat), dtype=torch.float32, device=dev)
    idx = torch.empty((n,), dtype=torch.int32, device=dev)
    li = torch.empty((n, d), dtype=torch.float32, device=dev)
    lv = (C.c_int32 * d)(*levels)
    if x is not None:
        lat = torch.empty((n, d), dtype=torch.float32, device=dev) if want_latents else None
        x = x.reshape(n, feat).contiguous()
        xp = x.data_ptr()
    else:
        lat = latents_in.reshape(n, d).contiguous().clone()
        xp = None
    _capi.check(lib.l3ac_fsq_forward(xp, n, feat, lv, d, w_in.data_ptr() if w_in is not None else None,
                                     b_in.data_ptr() if b_in is not None else None, w_out.data_ptr(), b_out.data_ptr(),
                                     q.data_ptr(), idx.data_ptr(), li.data_ptr(), lat.data_ptr() if lat is not None else None,
                                     _stream(dev)))
    torch.cuda.synchronize()
    return q, idx, li, lat


def fsq_quantize_act(act, levels, w_out, b_out):
    """SuperFSQ.quantize_act_value onwards (vq/fsq.py:56-68), from activation values in [0, 1]."""
    lib = _capi.load_library()
    dev = w_out.device
    feat, d = w_out.shape[0], len(levels)
    act = act.reshape(-1, d).contiguous()
    n = act.shape[0]
    q = torch.empty((n, feat), dtype=torch.float32, device=dev)
    idx = torch.empty((n,), dtype=torch.int32, device=dev)
    li = torch.empty((n, d), dtype=torch.float32, device=dev)
    _capi.check(lib.l3ac_fsq_quantize_act(act.data_ptr(), n, feat, (C.c_int32 * d)(*levels), d, w_out.data_ptr(), b_out.data_ptr(),
                                          q.data_ptr(), idx.data_ptr(), li.data_ptr(), _stream(dev)))
    torch.cuda.synchronize()
    return q, idx, li


def fsq_decode(indices, levels, w_out, b_out):
    lib = _capi.load_library()
    dev = w_out.device
    feat = w_out.shape[0]
    idx = indices.reshape(-1).to(torch.int32).contiguous()
    n = idx.shape[0]
    q = torch.empty((n, feat), dtype=torch.float32, device=dev)
    lv = (C.c_int32 * len(levels))(*levels)
    _capi.check(lib.l3ac_fsq_decode(idx.data_ptr(), n, feat, lv, len(levels), w_out.data_ptr(), b_out.data_ptr(),
                                    q.data_ptr(), _stream(dev)))
    torch.cuda.synchronize()
    return q


def vq_argmin(queries, codebook, form=0, info=None):
    """form 1: the direct-form scan where the screened form would run (an argument of l3ac_vq_argmin).  info: dict that receives
    'listed' = how many queries the screened form sent to the full direct-form search (first int32 of its scratch)."""
    lib = _capi.load_library()
    n, dim = queries.shape
    out = torch.empty((n,), dtype=torch.int32, device=queries.device)
    nbytes = lib.l3ac_vq_argmin_scratch_bytes(n, codebook.shape[0], form)
    scratch = torch.zeros((max(nbytes, 4),), dtype=torch.uint8, device=queries.device)
    _capi.check(lib.l3ac_vq_argmin(queries.data_ptr(), n, codebook.data_ptr(), codebook.shape[0], dim, out.data_ptr(),
                                   scratch.data_ptr(), nbytes, form, _stream(queries.device)))
    torch.cuda.synchronize()
    if info is not None:
        info["listed"] = int(scratch[:4].view(torch.int32).item()) if (form == 0 and n >= 5120) else None
    return out


def gemm(a, w, bias):
    lib = _capi.load_library()
    m, k = a.shape
    n = w.shape[0]
    c = torch.empty((m, n), dtype=torch.float32, device=a.device)
    _capi.check(lib.l3ac_gemm_f32(a.data_ptr(), a.stride(0), w.data_ptr(), bias.data_ptr() if bias is not None else None,
                                  c.data_ptr(), n, m, n, k, _stream(a.device)))
    torch.cuda.synchronize()
    return c


def gemm_split(a, w, bias):
    """bf16x3 split-operand GEMM through the C ABI: builds the weight image on the device, then multiplies."""
    lib = _capi.load_library()
    m, k = a.shape
    n = w.shape[0]
    nbytes = lib.l3ac_gemm_split_image_bytes(n, k)
    assert nbytes > 0, f"shape n={n} k={k} is not eligible for the split kernel"
    img = torch.empty((nbytes,), dtype=torch.uint8, device=a.device)
    c = torch.empty((m, n), dtype=torch.float32, device=a.device)
    _capi.check(lib.l3ac_gemm_split_image(w.data_ptr(), n, k, img.data_ptr(), _stream(a.device)))
    _capi.check(lib.l3ac_gemm_split_f32(a.data_ptr(), a.stride(0), img.data_ptr(), bias.data_ptr() if bias is not None else None,
                                        c.data_ptr(), n, m, n, k, _stream(a.device)))
    torch.cuda.synchronize()
    return c


def snake(x, alpha, mode=0):
    """l3ac_op_snake: x [rows][c], alpha [c]; mode bit 0 = packed form, bit 1 = sin(x)^2 only."""
    lib = _capi.load_library()
    rows, c = x.shape
    y = torch.empty_like(x)
    _capi.check(lib.l3ac_op_snake(x.data_ptr(), y.data_ptr(), rows, c, alpha.data_ptr(), mode, _stream(x.device)))
    torch.cuda.synchronize()
    return y
