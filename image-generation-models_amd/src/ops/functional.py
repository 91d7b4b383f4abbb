"""Tensor-level wrappers over the C ABI (include/mi_ddpm.h).

Activations are fp32 NHWC tensors [N, H, W, C] on a HIP device; a channel slice
``t[..., a:b]`` of a contiguous tensor is a legal activation (pixel stride ld = t.stride(2)).
Every function launches on torch's current stream and never synchronises.  There is no
CPU path: a tensor that is not on a GPU raises.
"""
from __future__ import annotations

import ctypes as C
import functools
import os
from typing import Optional, Tuple

import torch

from .lib import ROWSUM_MAX, MiConvDesc, MiGnDesc, MiRowSum, MiWgradDesc, check, load_library

MODE_FP32, MODE_BF16 = 0, 1


def debug_knob(name: str, default: str) -> str:
    """Experiment / A-B switches (kernel picks, copies on or off): honoured only when MI_DEBUG_KNOBS=1 is set as well (the tools/
    that A/B them, e.g. tools/sample_steps.py, set it), so that a stray variable cannot put a production run on a path the test suite does not cover.  The mode selectors a
    user is meant to set are read directly: MI_DDPM_MODE, MI_DDPM_STORAGE, MI_DDPM_FUSE_GN, MI_DDPM_GRAPH, MI_DDPM_LIB,
    MI_CONV_AUTO (halo-only fallback, covered by a subprocess test), MI_DDP_BUCKET_MB, MI_DIST_BACKEND."""
    if os.environ.get("MI_DEBUG_KNOBS") == "1":
        return os.environ.get(name, default)
    return default
PROBE = None      # bench.py sets this to a list: (kernel symbol, algorithmic FLOPs, start event, end event, shape note, algorithmic
                  # HBM bytes) per launch, events recorded on the launch stream


# MI_CONV_AUTO=0: the register-staged halo kernel (round 1 / 2) everywhere -- the fallback of the per-layer pick below, kept under test
# by a subprocess run of the conv and UNet tests.
CONV_AUTO = os.environ.get("MI_CONV_AUTO", "1") == "1"


# The private-weight-stream kernel (csrc/conv_pw.hip): 128-pixel tiles, one barrier per 64-channel chunk, two workgroups per CU.
USE_CONV_PW = CONV_AUTO and debug_knob("MI_CONV_PW", "1") == "1"     # MI_CONV_AUTO=0: the halo kernel everywhere


def _pw_sym(d, out16, var=0):
    """conv_pw_kernel's symbol as rocprofv3 prints it."""
    pt = _query("mi_conv3x3_pw_tile", d) if var < 2 else 128
    pt = _pw_tile_now(d, var, False, pt)
    return f"conv_pw_kernel<{'true' if out16 else 'false'}, {var}, 0, {pt or 128}>"


def _pw_tile_now(d, var, in32, pt):
    """The tile a launch runs on right now: a split-K launch (mi_conv3x3_pw_splitk) takes 128-pixel tiles where the unsplit rule takes 64."""
    q = _query("mi_conv3x3_pw_splitk", d, var, int(in32))
    return (q >> 4) if (q & 15) > 1 else pt


def _pick_pw(N, H, W, K, Nc, d=None):
    """The private-weight-stream kernel takes a layer that offers it at least about one tile per CU: 128-pixel x 128-channel tiles,
    or (d given: the plain conv and the variant with GroupNorm sums; the library decides, mi_conv3x3_pw_tile) 64-pixel tiles when
    the 128-pixel ones would be too few.  Smaller layers stay with round 2's kernels."""
    pt = 128
    if d is not None:
        pt = _query("mi_conv3x3_pw_tile", d)
        if not pt:
            return False
    return (N * H * W // pt) * ((Nc + 127) // 128) >= PW_MIN_TILES


PW_MIN_TILES = int(debug_knob("MI_CONV_PW_MIN_TILES", "64"))


USE_WGRAD_TR = debug_knob("MI_W3_TR", "1") != "0"      # A/B switch for the LDS-DMA weight-gradient kernel (csrc/wgrad_tr.hip)


_QUERY_CACHE = {}


def _desc_key(d):
    return tuple(getattr(d, f) for f, _ in d._fields_)


def _query(name, d, *extra):
    """Cached yes / no (or size) answer of a descriptor query of the library: the answers are pure functions of the descriptor, and
    one FFI call per launch for them was ~15 % of the step's enqueue time."""
    key = (name, _desc_key(d)) + extra
    r = _QUERY_CACHE.get(key)
    if r is None:
        r = getattr(load_library(), name)(C.byref(d), *extra)
        _QUERY_CACHE[key] = r
    return r


@functools.lru_cache(maxsize=None)
def _linattn_ws(N, n, heads):
    return load_library().mi_linattn_workspace(N, n, heads)


@functools.lru_cache(maxsize=None)
def _cvt_colsum_ws(M, Cc):
    return load_library().mi_f32_to_bf16_colsum_workspace(M, Cc)


def _probe_open():
    if PROBE is None:
        return None
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record()
    return e0


def _probe_close(e0, sym, flops, desc, nbytes=0.0):
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record()
    PROBE.append((sym, float(flops), e0, e1, desc, float(nbytes)))


def _esz(t) -> int:
    return 0 if t is None else t.element_size()


def _stream() -> C.c_void_p:
    """torch's current stream on the CURRENT device: kernels launch on the current HIP device, so every wrapper first checks
    (`_need_gpu`) that its tensors live there -- a rank that forgot `torch.cuda.set_device(LOCAL_RANK)` raises instead of
    launching on GPU 0 against another GPU's memory."""
    global LAUNCHES
    LAUNCHES += 1
    return C.c_void_p(_raw_stream(_cur_dev()))


LAUNCHES = 0      # library launches so far (every wrapper fetches the stream once per launch): lets a stream capture know whether a
                  # segment it is about to close recorded anything (src/runtime/graphed.py)


# the raw forms of torch.cuda.current_device() / current_stream(): same answers (they follow torch.cuda.stream(...) contexts and
# graph capture), without the Python-side device bookkeeping that cost ~3 us per launch
_cur_dev = torch._C._cuda_getDevice
_raw_stream = torch._C._cuda_getCurrentRawStream


def _p(t: Optional[torch.Tensor]) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def _need_gpu(t: torch.Tensor):
    if not t.is_cuda:
        raise RuntimeError("libmi_ddpm kernels need tensors on an MI355X (HIP) device; there is no CPU fallback")
    if t.dtype not in (torch.float32, torch.bfloat16):
        raise RuntimeError(f"expected float32 (or bf16 block-internal storage), got {t.dtype}")
    if t.device.index != _cur_dev():
        raise RuntimeError(f"tensor is on {t.device} but the current HIP device is cuda:{torch.cuda.current_device()}; "
                           "call torch.cuda.set_device(tensor.device) first (one process per GPU)")
    if t.device.index not in _SPLITK_WS:
        _splitk_workspace(t.device)


# Split-K launches of the 3x3 tile kernel (the layers whose tiles would leave CUs without a workgroup: the sampler at B = 64, cfg 3 at 32 images
# per GPU) exchange fp32 partial tiles through a per-device workspace the library is told about once: zeroed here, never freed (captured graphs
# hold its address).  SPLITK_MB = 0 (MI_CONV_PW_SPLITK_MB with MI_DEBUG_KNOBS=1) leaves every launch unsplit.
_SPLITK_WS = {}
SPLITK_MB = int(debug_knob("MI_CONV_PW_SPLITK_MB", "64"))


def _splitk_workspace(device):
    ws = None
    if SPLITK_MB > 0 and USE_CONV_PW and not torch.cuda.is_current_stream_capturing():
        ws = torch.zeros(SPLITK_MB << 20, dtype=torch.uint8, device=device)
        torch.cuda.current_stream().synchronize()
        check(load_library().mi_conv_pw_set_splitk_workspace(ws.data_ptr(), ws.numel()), "mi_conv_pw_set_splitk_workspace")
    elif torch.cuda.is_current_stream_capturing():
        return                                   # decided by the first eager call on this device
    _SPLITK_WS[device.index] = ws


def ld_of(t: torch.Tensor) -> int:
    """Pixel stride of an NHWC activation (or row stride of a 2-D [M, C] matrix)."""
    if t.dim() == 2:
        assert t.stride(1) == 1
        return t.stride(0)
    n, h, w, c = t.shape
    ld = t.stride(2)
    assert t.stride(3) == 1 and ld >= c, (t.shape, t.stride())
    assert (w == 1 or True) and t.stride(1) == w * ld and (n == 1 or t.stride(0) == h * w * ld), (t.shape, t.stride())
    return ld


def new_act(n, h, w, c, like: torch.Tensor, dtype=torch.float32) -> torch.Tensor:
    return torch.empty((n, h, w, c), device=like.device, dtype=dtype)


def _b16(t) -> int:
    return int(t is not None and t.dtype == torch.bfloat16)


# --------------------------------------------------------------------------- conv family
@functools.lru_cache(maxsize=None)      # pure function of the shape: one FFI query per distinct layer, not per step
def igemm_bf16_in_supported(K, Nc, k, stride, transposed, mode, out_hw):
    """Can conv_igemm read bf16-stored activations for this layer (the stride-2 conv / transposed conv family, dense tensors)?"""
    d = MiConvDesc(N=1, IH=1, IW=1, OH=out_hw[0], OW=out_hw[1], K=K, Nc=Nc, KH=k, KW=k, stride=stride, pad=1, transposed=int(transposed),
                   w_kn=0, mode=mode, K1=K, ldx=K, ldx2=0, ldy=Nc, ldr=0, accumulate=0)
    return bool(load_library().mi_conv_igemm_bf16w_io_supported(C.byref(d)))


USE_CONV_GT = debug_knob("MI_CONV_GT", "1") != "0"      # A/B switch: the tap-gather kernel for the stride-2 / transposed convs


def conv_gt(x, wq, *, kh, kw, stride, pad, transposed, K, Nc, out_hw, bias=None, residual=None, out=None, accumulate=False,
            out_dtype=torch.float32, mode=MODE_BF16, want16=False):
    """The stride-2 / transposed convs and their data gradients through the tap-gather kernel (mi_conv_gt): x bf16, wq = the layer's
    fragment-order weights (wfq: contraction over the master layout's ci, wdq: over co).  Returns None when the kernel does not take
    the layer (the caller falls back to conv_igemm).  mode = MODE_FP32: the exact-fp32 instantiation (x fp32, wq = wfq32 / wdq32)."""
    if not USE_CONV_GT or wq is None or x.dtype != (torch.bfloat16 if mode == MODE_BF16 else torch.float32):
        return None
    _need_gpu(x)
    N, IH, IW, _ = x.shape
    OH, OW = out_hw
    d = MiConvDesc(N=N, IH=IH, IW=IW, OH=OH, OW=OW, K=K, Nc=Nc, KH=kh, KW=kw, stride=stride, pad=pad, transposed=int(transposed), w_kn=0,
                   mode=mode, K1=K, ldx=ld_of(x), ldx2=0, ldy=Nc if out is None else ld_of(out),
                   ldr=ld_of(residual) if residual is not None else 0, accumulate=int(accumulate))
    if d.ldy % 8 or not _query("mi_conv_gt_supported", d):
        return None
    if out is None:
        assert not accumulate
        out = new_act(N, OH, OW, Nc, x, out_dtype)
        d.ldy = ld_of(out)
    e0 = _probe_open()
    y16 = None
    if want16:          # -> (y, y16): the bf16 copy of the fp32 output from the same epilogue
        assert out.dtype == torch.float32 and not accumulate and mode == MODE_BF16
        y16 = new_act(N, OH, OW, Nc, x, torch.bfloat16)
        check(load_library().mi_conv_gt_dual(C.byref(d), _p(x), _p(wq), _p(bias), _p(residual), _p(out), _p(y16), ld_of(y16), _stream()),
              "mi_conv_gt_dual")
    else:
        check(load_library().mi_conv_gt(C.byref(d), _p(x), _p(wq), _p(bias), _p(residual), _p(out), _b16(out), _stream()), "mi_conv_gt")
    if e0 is not None:
        pxt, ncls = C.c_int(), C.c_int()
        load_library().mi_conv_gt_tile(C.byref(d), C.byref(pxt), C.byref(ncls))
        flops = 2.0 * N * OH * OW * Nc * K * (kh * kw if not (transposed and stride > 1) else kh * kw / (stride * stride))
        nb = N * IH * IW * K * _esz(x) + N * OH * OW * Nc * _esz(out) * (2 if accumulate else 1) + kh * kw * K * Nc * _esz(x)
        _probe_close(e0, f"conv_gt_kernel<{'true' if _b16(out) else 'false'}, {pxt.value}{', true' if mode == MODE_FP32 else ''}>", flops,
                     f"N{N} {IH}x{IW}->{OH}x{OW} K{K}->{Nc} k{kh} s{stride} T{int(transposed)} acc{int(accumulate)}", nb)
    return (out, y16) if want16 else out


def conv_igemm(x, w, *, kh, kw, stride, pad, transposed, w_kn, K, Nc, out_hw, mode, x2=None,
               bias=None, residual=None, out=None, accumulate=False, wb=None):
    """y = conv(x [| x2]) per MiConvDesc.  x: [N,IH,IW,K1], x2: [N,IH,IW,K-K1] or None."""
    _need_gpu(x)
    N, IH, IW, K1 = x.shape
    if x2 is None:
        K1 = K
    OH, OW = out_hw
    if out is None:
        assert not accumulate
        out = new_act(N, OH, OW, Nc if Nc % 4 == 0 else (Nc + 3) // 4 * 4, x)
        if out.shape[3] != Nc:
            out.zero_()
            out = out[..., :Nc]
    d = MiConvDesc(N=N, IH=IH, IW=IW, OH=OH, OW=OW, K=K, Nc=Nc, KH=kh, KW=kw, stride=stride, pad=pad,
                   transposed=int(transposed), w_kn=int(w_kn), mode=mode, K1=K1, ldx=ld_of(x),
                   ldx2=ld_of(x2) if x2 is not None else 0, ldy=ld_of(out),
                   ldr=ld_of(residual) if residual is not None else 0, accumulate=int(accumulate))
    if PROBE is not None:
        bm, bn = C.c_int(), C.c_int()
        load_library().mi_conv_igemm_tile(C.byref(d), C.byref(bm), C.byref(bn))
        flops = 2.0 * N * OH * OW * Nc * K * (kh * kw if not (transposed and stride > 1) else kh * kw / (stride * stride))
        e0 = _probe_open()
    in16 = _b16(x)
    if in16:                                                     # bf16-stored activations: ring kernel with 64-channel stages
        if wb is None or mode != MODE_BF16 or (x2 is not None and not _b16(x2)) or not load_library().mi_conv_igemm_bf16w_io_supported(C.byref(d)):
            raise RuntimeError("bf16-stored activations: layer not supported (check igemm_bf16_in_supported first)")
        check(load_library().mi_conv_igemm_bf16w_io(C.byref(d), _p(x), _p(x2), _p(wb), _p(bias), _p(residual), _p(out), 1, _stream()),
              "mi_conv_igemm_bf16w_io")
    elif wb is not None and mode == MODE_BF16 and K % 8 == 0:    # bf16 weight copy [tap][Nc][K]
        check(load_library().mi_conv_igemm_bf16w(C.byref(d), _p(x), _p(x2), _p(wb), _p(bias), _p(residual), _p(out), _stream()),
              "mi_conv_igemm_bf16w")
    else:
        check(load_library().mi_conv_igemm(C.byref(d), _p(x), _p(x2), _p(w), _p(bias), _p(residual), _p(out), _stream()),
              "mi_conv_igemm")
    if PROBE is not None:
        fast = wb is not None and mode == MODE_BF16 and K % 8 == 0
        nb = N * IH * IW * K * _esz(x) + N * OH * OW * Nc * _esz(out) * (2 if accumulate else 1) + kh * kw * K * Nc * (2 if fast else 4)
        _probe_close(e0, f"igemm{'_fast' if fast else ''}_kernel<{mode},{bm.value},{bn.value}{',in16' if in16 else ''}>", flops,
                     f"N{N} {IH}x{IW}->{OH}x{OW} K{K}->{Nc} k{kh} s{stride} T{int(transposed)} acc{int(accumulate)}", nb)
    return out


def conv3x3_bf16w(x, wsh, *, K, Nc, flip, ksize=3, x2=None, bias=None, residual=None, out=None, accumulate=False,
                  out_dtype=torch.float32, gn_sums=None, want16=False, wq=None):
    """3x3/s1/p1 (or 1x1) conv (flip=False) or its data gradient (flip=True) through the pipelined
    LDS-tile kernel.  wsh: bf16 weights [k][k][Nc][K].  Returns None when the shape is not supported.
    want16: -> (y, y16), the fp32 output and its bf16 copy written by the same epilogue.
    wq: the same weights in MFMA-fragment order (pack_weights_bf16's wfq / wdq slice) -- lets the private-weight-stream kernel
    (mi_conv3x3_pw) take the layer."""
    _need_gpu(x)
    N, H, W, K1 = x.shape
    if x2 is None:
        K1 = K
    d = MiConvDesc(N=N, IH=H, IW=W, OH=H, OW=W, K=K, Nc=Nc, KH=ksize, KW=ksize, stride=1, pad=ksize // 2, transposed=int(flip),
                   w_kn=0, mode=MODE_BF16, K1=K1, ldx=ld_of(x), ldx2=ld_of(x2) if x2 is not None else 0, ldy=0,
                   ldr=ld_of(residual) if residual is not None else 0, accumulate=int(accumulate))
    lib = load_library()
    if not _query("mi_conv3x3_bf16w_supported", d):
        return None
    if out is None:
        assert not accumulate
        out = new_act(N, H, W, Nc, x, out_dtype)
    d.ldy = ld_of(out)
    io = _b16(x) | (_b16(out) << 1)
    assert x2 is None or x2.dtype == x.dtype
    pick = "halo"
    if (ksize == 1 and wq is not None and USE_CONV_PW and _b16(x) and gn_sums is None and K % 128 == 0
            and _query("mi_conv1x1_pw_supported", d)):
        # the streaming 1x1 kernel: tile by LDS-DMA per 128-channel chunk, per-wave weight fragments, whole-row stores (conv1x1_pw_kernel)
        y16 = new_act(N, H, W, Nc, x, torch.bfloat16) if want16 else None
        assert not (want16 and (accumulate or _b16(out)))
        e0 = _probe_open()
        check(lib.mi_conv1x1_pw(C.byref(d), _p(x), _p(x2), _p(wq), _p(bias), _p(residual), _p(out), _b16(out), _p(y16),
                                ld_of(y16) if y16 is not None else 0, _stream()), "mi_conv1x1_pw")
        if e0 is not None:
            nb = (N * H * W * K * 2 + N * H * W * Nc * (_esz(out) * (2 if accumulate else 1) + _esz(residual) + (2 if want16 else 0)) + K * Nc * 2)
            px = 64 if (N * H * W // 128) * ((Nc + 127) // 128) < 256 else 128
            # (the channel-tile loop instantiation: mi_conv1x1_pw's condition, restated for the symbol name only)
            sw = lib.mi_debug_conv1x1_pw_nloop(-1)
            nloop = (_b16(out) and residual is None and not accumulate and px == 128 and K == 128 and x2 is None and Nc > 128
                     and (sw == 2 or (sw == 1 and N * H * W // 128 >= 1024)))
            sym = (f"conv1x1_pw_kernel<true, false, 128, false, false, true>" if nloop else
                   f"conv1x1_pw_kernel<{'true' if _b16(out) else 'false'}, {'true' if want16 else 'false'}, {px}>")
            _probe_close(e0, sym, 2.0 * N * H * W * Nc * K,
                         f"N{N} {H}x{W} K{K}{'(2src)' if x2 is not None else ''}->{Nc} flip{int(flip)} acc{int(accumulate)}", nb)
        return (out, y16) if want16 else out
    if (ksize == 1 and not _b16(x) and USE_CONV_PW and wq is not None and not want16 and gn_sums is None and K % 128 == 0
            and _query("mi_conv1x1_pw_x32_supported", d)):
        # fp32-stored input of a 1x1 conv (the residual-stream gradient of to_out / res_conv's data gradients, res_conv in inference)
        e0 = _probe_open()
        check(lib.mi_conv1x1_pw_x32(C.byref(d), _p(x), _p(x2), _p(wq), _p(bias), _p(residual), _p(out), _b16(out), _stream()), "mi_conv1x1_pw_x32")
        if e0 is not None:
            nb = (N * H * W * K * 4 + N * H * W * Nc * (_esz(out) * (2 if accumulate else 1) + _esz(residual)) + K * Nc * 2)
            _probe_close(e0, f"conv1x1_pw_kernel<{'true' if _b16(out) else 'false'}, false, 64, false, true>", 2.0 * N * H * W * Nc * K,
                         f"N{N} {H}x{W} K{K}{'(2src)' if x2 is not None else ''}->{Nc} fp32 in flip{int(flip)} acc{int(accumulate)}", nb)
        return out
    if (ksize == 3 and not _b16(x) and USE_CONV_PW and wq is not None and not want16 and d.ldx % 4 == 0 and (x2 is None or d.ldx2 % 4 == 0)):
        # fp32-stored input (the residual stream: the sampler's block1 convs): the same kernel, pieces rounded to bf16 while staged
        pt = _query("mi_conv3x3_pw_x32_tile", d)
        if pt and (N * H * W // pt) * ((Nc + 127) // 128) >= PW_MIN_TILES:
            assert gn_sums is None or (gn_sums.dtype == torch.int64 and gn_sums.numel() == N * (Nc // 16) * 2)
            e0 = _probe_open()
            check(lib.mi_conv3x3_pw_x32(C.byref(d), _p(x), _p(x2), _p(wq), _p(bias), _p(residual), _p(out), _b16(out), _p(gn_sums), _stream()),
                  "mi_conv3x3_pw_x32")
            if e0 is not None:
                nb = (N * H * W * K * 4 + N * H * W * Nc * (_esz(out) * (2 if accumulate else 1) + _esz(residual)) + 9 * K * Nc * 2)
                _probe_close(e0, f"conv_pw_kernel<{'true' if _b16(out) else 'false'}, {0 if gn_sums is None else 1}, 0, {_pw_tile_now(d, 0, True, pt)}, true>",
                             2.0 * N * H * W * Nc * K * 9, f"N{N} {H}x{W} K{K}{'(2src)' if x2 is not None else ''}->{Nc} fp32 in flip{int(flip)} acc{int(accumulate)}", nb)
            return out
    if (gn_sums is not None and ksize == 3 and _b16(x) and USE_CONV_PW and wq is not None and _pick_pw(N, H, W, K, Nc, d)
            and _query("mi_conv3x3_pw_supported", d)):
        # the next layer's GroupNorm sums from the private-weight-stream kernel's epilogue
        assert gn_sums.dtype == torch.int64 and gn_sums.numel() == N * (Nc // 16) * 2
        e0 = _probe_open()
        check(lib.mi_conv3x3_pw_gnsums(C.byref(d), _p(x), _p(x2), _p(wq), _p(bias), _p(residual), _p(out), _b16(out), _p(gn_sums), _stream()),
              "mi_conv3x3_pw_gnsums")
        if e0 is not None:
            nb = (N * H * W * K * 2 + N * H * W * Nc * (_esz(out) * (2 if accumulate else 1) + _esz(residual)) + 9 * K * Nc * 2)
            _probe_close(e0, _pw_sym(d, _b16(out), 1), 2.0 * N * H * W * Nc * K * 9,
                         f"N{N} {H}x{W} K{K}->{Nc} + GroupNorm sums", nb)
        return out
    if gn_sums is None and not want16 and ksize == 3 and _b16(x):
        if USE_CONV_PW and wq is not None and _pick_pw(N, H, W, K, Nc, d):
            pick = "pw"
    if pick == "pw":
        e0 = _probe_open()
        check(lib.mi_conv3x3_pw(C.byref(d), _p(x), _p(x2), _p(wq), _p(bias), _p(residual), _p(out), _b16(out), _stream()), "mi_conv3x3_pw")
        if e0 is not None:
            nb = (N * H * W * K * 2 + N * H * W * Nc * (_esz(out) * (2 if accumulate else 1) + _esz(residual)) + 9 * K * Nc * 2)
            _probe_close(e0, _pw_sym(d, _b16(out)), 2.0 * N * H * W * Nc * K * 9,
                         f"N{N} {H}x{W} K{K}{'(2src)' if x2 is not None else ''}->{Nc} flip{int(flip)} acc{int(accumulate)}", nb)
        return out
    e0 = _probe_open()
    if want16:                    # fp32 output + its bf16 copy from one epilogue
        assert out.dtype == torch.float32 and gn_sums is None and not accumulate
        y16 = new_act(N, H, W, Nc, x, torch.bfloat16)
        check(lib.mi_conv3x3_bf16w_io_dual(C.byref(d), _p(x), _p(x2), _p(wsh), _p(bias), _p(residual), _p(out), _p(y16), ld_of(y16), io, _stream()),
              "mi_conv3x3_bf16w_io_dual")
    elif gn_sums is not None:     # the next layer's GroupNorm statistics ride in this conv's epilogue
        assert ksize == 3 and gn_sums.dtype == torch.int64 and gn_sums.numel() == N * (Nc // 16) * 2
        check(lib.mi_conv3x3_bf16w_io_gnsums(C.byref(d), _p(x), _p(x2), _p(wsh), _p(bias), _p(residual), _p(out), io, _p(gn_sums), _stream()),
              "mi_conv3x3_bf16w_io_gnsums")
    elif io:
        check(lib.mi_conv3x3_bf16w_io(C.byref(d), _p(x), _p(x2), _p(wsh), _p(bias), _p(residual), _p(out), io, _stream()),
              "mi_conv3x3_bf16w_io")
    else:
        check(lib.mi_conv3x3_bf16w(C.byref(d), _p(x), _p(x2), _p(wsh), _p(bias), _p(residual), _p(out), _stream()),
              "mi_conv3x3_bf16w")
    if e0 is not None:
        bm, ck, sk = C.c_int(), C.c_int(), C.c_int()
        lib.mi_conv3x3_bf16w_tile(C.byref(d), io, C.byref(bm), C.byref(ck), C.byref(sk))
        nb = (N * H * W * K * _esz(x) + N * H * W * Nc * (_esz(out) * (2 if accumulate else 1) + _esz(residual))
              + ksize * ksize * K * Nc * 2)
        _probe_close(e0, f"conv3x3_halo_kernel<{bm.value}, {ck.value}, {ksize}, {'true' if sk.value else 'false'}, {io}, {8 if bm.value == 256 or (bm.value == 128 and ck.value == 64) else 4}>",
                     2.0 * N * H * W * Nc * K * ksize * ksize,
                     f"N{N} {H}x{W} K{K}{'(2src)' if x2 is not None else ''}->{Nc} flip{int(flip)} acc{int(accumulate)} io{io}", nb)
    return (out, y16) if want16 else out


def gn_stats_coef(x, gamma, beta, *, groups=8, eps=1e-5, temb=None):
    """GroupNorm statistics of x only: -> (stats [N,G,2] = mean, rstd; coef [3,N,C] = scale, shift, time bias) for
    conv3x3_gn_mish, which applies GroupNorm + Mish (+ time bias) while it stages its input (ddpm.py:112-120,139-140)."""
    _need_gpu(x)
    N, H, W, Cc = x.shape
    stats = torch.empty((N, groups, 2), device=x.device, dtype=torch.float32)
    coef = torch.empty((3, N, Cc), device=x.device, dtype=torch.float32)
    d = MiGnDesc(N=N, HW=H * W, C=Cc, G=groups, eps=eps, ldx=ld_of(x), ldy=Cc, ldr=0)
    e0 = _probe_open()
    check(load_library().mi_gn_stats_coef(C.byref(d), _p(x), _p(gamma), _p(beta), _p(temb), ld_of(temb) if temb is not None else 0,
                                          _p(stats), _p(coef), _b16(x), _stream()), "mi_gn_stats_coef")
    if e0 is not None:
        _probe_close(e0, f"gn_mish_fwd_kernel<io{_b16(x)}> (statistics only)", 0.0, f"N{N} HW{H * W} C{Cc}", N * H * W * Cc * _esz(x))
    return stats, coef


GSUM_SCALE = float(1 << 20)      # csrc/common.h MI_GSUM_SCALE: the epilogue sums are 64-bit fixed point (integer atomics: order-independent)


def gn_sums_buffer(N, Nc, device):
    """Zeroed [N][Nc / 16][2] buffer for the (sum, sum of squares) a conv's epilogue accumulates (gn_sums=...)."""
    return torch.zeros(N * (Nc // 16) * 2, device=device, dtype=torch.int64)


def gn_sums_encode(sums_f64):
    """float sums [N][Nc / 16][2] -> the fixed-point buffer the kernels read (tests, benchmarks)."""
    return torch.round(sums_f64.double() * GSUM_SCALE).to(torch.int64).contiguous().view(-1)


def gn_sums_decode(buf):
    return buf.double() / GSUM_SCALE


def gn_coef_from_sums(sums, N, HW, gamma, beta, *, groups=8, eps=1e-5, temb=None):
    """-> (stats [N][G][2], coef [3][N][C]) from the (sum, sum of squares) per sample and 16-channel slab that the producing conv's
    epilogue accumulated (conv3x3_bf16w(..., gn_sums=sums)): what gn_stats_coef computes with a pass over the tensor."""
    Cc = gamma.numel()
    stats = torch.empty((N, groups, 2), device=sums.device, dtype=torch.float32)
    coef = torch.empty((3, N, Cc), device=sums.device, dtype=torch.float32)
    e0 = _probe_open()
    check(load_library().mi_gn_coef_from_sums(N, Cc, groups, HW, eps, _p(sums), _p(gamma), _p(beta), _p(temb),
                                              ld_of(temb) if temb is not None else 0, _p(stats), _p(coef), _stream()), "mi_gn_coef_from_sums")
    if e0 is not None:
        _probe_close(e0, "gn_coef_from_sums_kernel", 0.0, f"N{N} C{Cc}", N * Cc * 16.0)
    return stats, coef


def gn_mish_apply_sums(x, sums, gamma, beta, *, groups=8, eps=1e-5, temb=None, residual=None, out_dtype=torch.float32, want16=False):
    """GroupNorm + Mish (+ time bias) (+ residual) as one streaming pass over x (bf16, the output of a conv that ran with gn_sums=sums):
    the statistics come from the conv's epilogue sums, so there is no statistics phase.  -> (y, stats[, y16]) like gn_mish_fwd, or None when
    the library does not take the shape (the caller then runs gn_mish_fwd)."""
    _need_gpu(x)
    if x.dtype != torch.bfloat16:
        return None
    N, H, W, Cc = x.shape
    y16 = None
    y = new_act(N, H, W, Cc, x, out_dtype)
    if want16:
        assert out_dtype == torch.float32
        y16 = new_act(N, H, W, Cc, x, torch.bfloat16)
    stats = torch.empty((N, groups, 2), device=x.device, dtype=torch.float32)
    d = MiGnDesc(N=N, HW=H * W, C=Cc, G=groups, eps=eps, ldx=ld_of(x), ldy=ld_of(y), ldr=ld_of(residual) if residual is not None else 0)
    e0 = _probe_open()
    rc = load_library().mi_gn_mish_apply_sums(C.byref(d), _p(x), _p(sums), _p(gamma), _p(beta), _p(temb), ld_of(temb) if temb is not None else 0,
                                              _p(residual), _p(y), _b16(y), _p(y16), ld_of(y16) if y16 is not None else 0, _p(stats), _stream())
    if rc == 1:
        return None
    check(rc, "mi_gn_mish_apply_sums")
    if e0 is not None:
        _probe_close(e0, f"gn_apply_sums_kernel<io{1 + 2 * _b16(y)}>", 0.0, f"N{N} HW{H * W} C{Cc}",
                     N * H * W * Cc * (_esz(x) + _esz(y) + _esz(residual) + _esz(y16)))
    return (y, stats, y16) if want16 else (y, stats)


@functools.lru_cache(maxsize=None)      # pure function of the shape: one FFI query per distinct layer, not per step
def conv3x3_gn_mish_supported(N, H, W, K, Nc):
    d = MiConvDesc(N=N, IH=H, IW=W, OH=H, OW=W, K=K, Nc=Nc, KH=3, KW=3, stride=1, pad=1, transposed=0, w_kn=0, mode=MODE_BF16, K1=K,
                   ldx=K, ldx2=K, ldy=Nc, ldr=0, accumulate=0)
    return bool(load_library().mi_conv3x3_gn_mish_supported(C.byref(d)))


@functools.lru_cache(maxsize=None)
def conv3x3_pw_gn_mish_picked(N, H, W, K, Nc):
    """Does conv3x3_gn_mish route a bf16-stored layer of this shape to the private-weight-stream fused kernel (given wq)?"""
    d = MiConvDesc(N=N, IH=H, IW=W, OH=H, OW=W, K=K, Nc=Nc, KH=3, KW=3, stride=1, pad=1, transposed=0, w_kn=0, mode=MODE_BF16, K1=K,
                   ldx=K, ldx2=K, ldy=Nc, ldr=0, accumulate=0)
    # Where it pays (measured in the B = 64 sampler, round 4): every 64-channel chunk of K costs the fused kernel one transform block
    # (~1 us that the matrix pipe of that workgroup waits for), the GroupNorm pass it replaces costs ~4 us + its bytes -- the fusion
    # wins up to K = 256 (level 0: -2.3 us per block, 256 channels @16x16: -4 us) and loses on the 512-channel 8x8 layers (+2 us)
    # (round 6, the fused variants on the pinned loop: K = 512 is break-even or slightly ahead -- sampler 994-1 003 -> 1 002-1 004 steps/s)
    return bool(USE_CONV_PW and K <= int(debug_knob("MI_FUSE_GN_MAXK", "512")) and _pick_pw_fused(N, H, W, Nc, d, False))


def _pick_pw_fused(N, H, W, Nc, d, in32):
    """Tile (128 / 64 pixels; 0 = no) the fused GroupNorm + Mish + conv variants of the private-weight-stream kernel would run this
    layer with -- the library's pick, subject to the same minimum grid as the plain conv."""
    pt = _query("mi_conv3x3_pw_x32_gn_mish_tile" if in32 else "mi_conv3x3_pw_gn_mish_tile", d)
    return pt if pt and (N * H * W // pt) * ((Nc + 127) // 128) >= PW_MIN_TILES else 0


def conv3x3_gn_mish(x, coef, wsh, *, K, Nc, bias=None, out_dtype=None, gn=None, wq=None):
    """The fused kernel BASELINE.json names: y = conv3x3(mish(x * scale + shift) + tb) + bias with x the RAW previous conv output
    (fp32 -> fp32 y, or bf16 -> bf16 y), coef from gn_stats_coef -- or coef = None and gn = (sums, gamma, beta, temb, groups, eps):
    the statistics come from the sums the producing conv's epilogue left (conv3x3_bf16w(..., gn_sums=sums)) and are resolved inside
    the kernel.  Returns None when the shape is not supported."""
    _need_gpu(x)
    N, H, W, _ = x.shape
    d = MiConvDesc(N=N, IH=H, IW=W, OH=H, OW=W, K=K, Nc=Nc, KH=3, KW=3, stride=1, pad=1, transposed=0, w_kn=0, mode=MODE_BF16, K1=K,
                   ldx=ld_of(x), ldx2=ld_of(x), ldy=Nc, ldr=0, accumulate=0)
    lib = load_library()
    ptf = _pick_pw_fused(N, H, W, Nc, d, x.dtype == torch.float32) if (wq is not None and USE_CONV_PW) else 0
    if ptf and x.dtype == torch.bfloat16:
        # the private-weight-stream kernel: the transform runs once per staged element, in place in LDS
        y = new_act(N, H, W, Nc, x, x.dtype if out_dtype is None else out_dtype)
        d.ldy = Nc
        e0 = _probe_open()
        if coef is None and (K // gn[4]) not in (16, 32, 64):
            _, coef = gn_coef_from_sums(gn[0], N, H * W, gn[1], gn[2], groups=gn[4], eps=gn[5], temb=gn[3])
        if coef is None:          # statistics, affine and time bias resolved inside the kernel from the producer's sums
            sums, gamma, beta, temb, groups, eps = gn
            check(lib.mi_conv3x3_pw_gn_mish_sums(C.byref(d), _p(x), _p(sums), _p(gamma), _p(beta), _p(temb),
                                                 ld_of(temb) if temb is not None else 0, groups, eps, _p(wq), _p(bias), _p(y), _b16(y),
                                                 _stream()), "mi_conv3x3_pw_gn_mish_sums")
        else:
            check(lib.mi_conv3x3_pw_gn_mish(C.byref(d), _p(x), _p(coef), _p(wq), _p(bias), _p(y), _b16(y), _stream()), "mi_conv3x3_pw_gn_mish")
        if e0 is not None:
            _probe_close(e0, f"conv_pw_kernel<{'true' if _b16(y) else 'false'}, {3 if coef is None else 2}, 0, {_pw_tile_now(d, 2, False, ptf)}>", 2.0 * N * H * W * Nc * K * 9,
                         f"N{N} {H}x{W} K{K}->{Nc} fused GN+Mish", N * H * W * (K * 2 + Nc * _esz(y)) + 9 * K * Nc * 2)
        return y
    if ptf and x.dtype == torch.float32:
        # fp32-stored activations on the same kernel (round 4): the transform is applied to the fp32 values in registers, between
        # their load and the one rounding to bf16
        y = new_act(N, H, W, Nc, x, x.dtype if out_dtype is None else out_dtype)
        d.ldy = Nc
        e0 = _probe_open()
        if coef is None and (K // gn[4]) not in (16, 32, 64):
            _, coef = gn_coef_from_sums(gn[0], N, H * W, gn[1], gn[2], groups=gn[4], eps=gn[5], temb=gn[3])
        if coef is None:
            sums, gamma, beta, temb, groups, eps = gn
            check(lib.mi_conv3x3_pw_x32_gn_mish_sums(C.byref(d), _p(x), _p(sums), _p(gamma), _p(beta), _p(temb),
                                                     ld_of(temb) if temb is not None else 0, groups, eps, _p(wq), _p(bias), _p(y), _b16(y),
                                                     _stream()), "mi_conv3x3_pw_x32_gn_mish_sums")
        else:
            check(lib.mi_conv3x3_pw_x32_gn_mish(C.byref(d), _p(x), _p(coef), _p(wq), _p(bias), _p(y), _b16(y), _stream()),
                  "mi_conv3x3_pw_x32_gn_mish")
        if e0 is not None:
            _probe_close(e0, f"conv_pw_kernel<{'true' if _b16(y) else 'false'}, {3 if coef is None else 2}, 0, {_pw_tile_now(d, 2, True, ptf)}, true>",
                         2.0 * N * H * W * Nc * K * 9, f"N{N} {H}x{W} K{K}->{Nc} fused GN+Mish, fp32 in",
                         N * H * W * (K * 4 + Nc * _esz(y)) + 9 * K * Nc * 2)
        return y
    if not lib.mi_conv3x3_gn_mish_supported(C.byref(d)):
        return None
    out_dtype = x.dtype if out_dtype is None else out_dtype
    if out_dtype != x.dtype:
        raise RuntimeError("the fused kernel writes y in x's storage type (fp32 -> fp32 or bf16 -> bf16)")
    y = new_act(N, H, W, Nc, x, out_dtype)
    io = 3 if x.dtype == torch.bfloat16 else 0
    e0 = _probe_open()
    if gn is not None:
        sums, gamma, beta, temb, groups, eps = gn
        check(lib.mi_conv3x3_gn_mish_sums(C.byref(d), _p(x), _p(sums), _p(gamma), _p(beta), _p(temb), ld_of(temb) if temb is not None else 0,
                                          groups, eps, _p(wsh), _p(bias), _p(y), io, _stream()), "mi_conv3x3_gn_mish_sums")
    else:
        check(lib.mi_conv3x3_gn_mish(C.byref(d), _p(x), _p(coef), _p(wsh), _p(bias), _p(y), io, _stream()), "mi_conv3x3_gn_mish")
    if e0 is not None:
        bm, ck = C.c_int(), C.c_int()
        lib.mi_conv3x3_gn_mish_tile(C.byref(d), C.byref(bm), C.byref(ck))
        _probe_close(e0, f"conv3x3_halo_kernel<{bm.value}, {ck.value}, 3, false, {io}, {8 if bm.value == 256 else 4}, true>",
                     2.0 * N * H * W * Nc * K * 9, f"N{N} {H}x{W} K{K}->{Nc} fused GN+Mish",
                     N * H * W * (K * _esz(x) + Nc * _esz(y)) + 9 * K * Nc * 2)
    return y


def pack_weights_tile():
    """Tile edge mi_pack_weights_bf16 counts an entry's tiles in (tile0 = running sum of taps * ceil(ci / T) * ceil(co / T))."""
    return int(load_library().mi_pack_weights_tile())


PACK_ENTRY = [("off", "<i8"), ("taps", "<i4"), ("ci", "<i4"), ("co", "<i4"), ("tile0", "<i4"), ("frag", "<i4"), ("pad", "<i4")]


def pack_table(entries, device):
    """Device table for mi_pack_weights_bf16 from (float offset, taps, ci, co) per conv weight -> (table, nent, total tiles).
    3x3 and 1x1 layers with ci % 64 == 0 and co % 64 == 0 are flagged for the MFMA-fragment-order copies (the operands of
    mi_conv3x3_pw / mi_conv1x1_pw)."""
    import numpy as np
    rec = np.zeros(len(entries), dtype=np.dtype(PACK_ENTRY))
    tile, T = 0, pack_weights_tile()
    for i, (off, taps, ci, co) in enumerate(entries):
        rec[i] = (off, taps, ci, co, tile, int(taps in (1, 9, 16) and ci % 64 == 0 and co % 64 == 0), 0)
        tile += taps * ((ci + T - 1) // T) * ((co + T - 1) // T)
    return torch.from_numpy(rec.view(np.uint8).copy()).to(device), len(entries), tile


def pack_weights_bf16(table_dev, nent, total_tiles, master, wd, wf, wdq=None, wfq=None):
    """wdq / wfq: the fragment-order copies (same offsets as the master buffer) of the entries the table flags."""
    e0 = _probe_open()
    check(load_library().mi_pack_weights_bf16(nent, _p(table_dev), total_tiles, _p(master), _p(wd), _p(wf), _p(wdq), _p(wfq), _stream()),
          "mi_pack_weights_bf16")
    if e0 is not None:
        _probe_close(e0, "pack_weights_kernel", 0.0, f"{master.numel()} params", master.numel() * (8.0 if wdq is None else 12.0))


def pack_weights_f32frag(table_dev, nent, total_tiles, master, wdq32, wfq32):
    """fp32 fragment-order copies (forward / data-gradient operand of mi_conv3x3_pw_f32) of the entries pack_table flags."""
    e0 = _probe_open()
    check(load_library().mi_pack_weights_f32frag(nent, _p(table_dev), total_tiles, _p(master), _p(wdq32), _p(wfq32), _stream()),
          "mi_pack_weights_f32frag")
    if e0 is not None:
        _probe_close(e0, "pack_f32frag_kernel", 0.0, f"{master.numel()} params", master.numel() * 12.0)


def conv3x3_f32(x, wq32, *, K, Nc, flip, x2=None, bias=None, residual=None, out=None, accumulate=False):
    """3x3/s1/p1 conv (flip=False) or its data gradient (flip=True) in exact-fp32 mode through the private-weight-stream kernel on
    v_mfma_f32_32x32x2_f32 (mi_conv3x3_pw_f32).  wq32: the layer's slice of the fp32 fragment-order copy.  None: not supported."""
    _need_gpu(x)
    if x.dtype != torch.float32 or (x2 is not None and x2.dtype != torch.float32) or not USE_CONV_PW:
        return None
    N, H, W, K1 = x.shape
    if x2 is None:
        K1 = K
    d = MiConvDesc(N=N, IH=H, IW=W, OH=H, OW=W, K=K, Nc=Nc, KH=3, KW=3, stride=1, pad=1, transposed=int(flip),
                   w_kn=0, mode=MODE_FP32, K1=K1, ldx=ld_of(x), ldx2=ld_of(x2) if x2 is not None else 0, ldy=Nc,
                   ldr=ld_of(residual) if residual is not None else 0, accumulate=int(accumulate))
    pt = _query("mi_conv3x3_pw_f32_tile", d)
    if not pt:
        return None
    if out is None:
        assert not accumulate
        out = new_act(N, H, W, Nc, x, torch.float32)
    d.ldy = ld_of(out)
    if d.ldy % 8:
        return None
    e0 = _probe_open()
    check(load_library().mi_conv3x3_pw_f32(C.byref(d), _p(x), _p(x2), _p(wq32), _p(bias), _p(residual), _p(out), _stream()), "mi_conv3x3_pw_f32")
    if e0 is not None:
        nb = (N * H * W * K * 4 + N * H * W * Nc * (4 * (2 if accumulate else 1) + _esz(residual)) + 9 * K * Nc * 4)
        _probe_close(e0, f"conv_pw_kernel<false, 0, 0, {pt}, false, true>", 2.0 * N * H * W * Nc * K * 9,
                     f"N{N} {H}x{W} K{K}{'(2src)' if x2 is not None else ''}->{Nc} fp32 flip{int(flip)} acc{int(accumulate)}", nb)
    return out


def conv1x1_f32(x, wq32, *, K, Nc, flip, x2=None, bias=None, residual=None, out=None, accumulate=False):
    """1x1 conv (flip=False) or its data gradient (flip=True) in exact-fp32 mode through the streaming kernel (mi_conv1x1_pw_f32).
    None: not supported."""
    _need_gpu(x)
    if x.dtype != torch.float32 or (x2 is not None and x2.dtype != torch.float32) or not USE_CONV_PW:
        return None
    N, H, W, K1 = x.shape
    if x2 is None:
        K1 = K
    d = MiConvDesc(N=N, IH=H, IW=W, OH=H, OW=W, K=K, Nc=Nc, KH=1, KW=1, stride=1, pad=0, transposed=int(flip),
                   w_kn=0, mode=MODE_FP32, K1=K1, ldx=ld_of(x), ldx2=ld_of(x2) if x2 is not None else 0, ldy=Nc,
                   ldr=ld_of(residual) if residual is not None else 0, accumulate=int(accumulate))
    if not _query("mi_conv1x1_pw_f32_supported", d):
        return None
    if out is None:
        assert not accumulate
        out = new_act(N, H, W, Nc, x, torch.float32)
    d.ldy = ld_of(out)
    if d.ldy % 8 or (residual is not None and d.ldr % 4):
        return None
    e0 = _probe_open()
    check(load_library().mi_conv1x1_pw_f32(C.byref(d), _p(x), _p(x2), _p(wq32), _p(bias), _p(residual), _p(out), _stream()), "mi_conv1x1_pw_f32")
    if e0 is not None:
        nb = (N * H * W * K * 4 + N * H * W * Nc * (4 * (2 if accumulate else 1) + _esz(residual)) + K * Nc * 4)
        px = 64 if (N * H * W // 128) * ((Nc + 127) // 128) < 256 else 128
        _probe_close(e0, f"conv1x1_pw_kernel<false, false, {px}, true>", 2.0 * N * H * W * Nc * K,
                     f"N{N} {H}x{W} K{K}{'(2src)' if x2 is not None else ''}->{Nc} fp32 flip{int(flip)} acc{int(accumulate)}", nb)
    return out


@functools.lru_cache(maxsize=None)
def small_cin_supported(ks, Cin, Cout, wgrad=False):
    """The 3-channel-input kernels (mi_conv_small_cin_*): which (kernel size, Cin, Cout) they take."""
    if ks not in (1, 3) or not 1 <= Cin <= 4:
        return False
    if wgrad:
        return Cout in (64, 128, 256) and 3 * ks * ks * Cin * (Cout // 4) * 16 <= 48 * 1024
    return Cout % 4 == 0 and Cout <= 1024 and 256 % (Cout // 4) == 0


@functools.lru_cache(maxsize=None)
def small_cin_bf16_supported(ks, N, H, W, Cin, Cout, ldx):
    """The first Block's conv (Cin <= 4) can write its output as bf16 AND its weight gradient can read a bf16 dy (round 4)."""
    return bool(load_library().mi_conv_small_cin_bf16_supported(ks, N, H, W, Cin, Cout, ldx))


def conv_small_cin_fwd(x, w, bias, Cout, ks, out_dtype=torch.float32):
    """Conv2d(Cin<=4, Cout, ks, padding=ks//2) on an NHWC image tensor (first convs of the UNet); out_dtype bf16: the output rounded
    once on the way out (small_cin_bf16_supported)."""
    _need_gpu(x)
    N, H, W, Cin = x.shape
    y = (torch.empty((N, H, W, Cout), device=x.device, dtype=torch.bfloat16) if out_dtype == torch.bfloat16 else new_act(N, H, W, Cout, x))
    check(load_library().mi_conv_small_cin_fwd_io(ks, N, H, W, Cin, Cout, _p(x), ld_of(x), _p(w), _p(bias), _p(y), ld_of(y), _b16(y), _stream()),
          "mi_conv_small_cin_fwd")
    return y


def conv1x1_small_cout_bwd(x, dy, w, dW, dx, accumulate=False):
    """Backward of the C -> Cs (<= 4) 1x1 conv in one pass over (x, dy): dW[C][Cs] += x^T dy and dx (+)= dy w^T (what conv1x1_small_cout op 2 and
    op 1 compute in two launches; bitwise the same)."""
    _need_gpu(x)
    N, H, W, Cc = x.shape
    Cs = dy.shape[3]
    lib = load_library()
    need = lib.mi_conv_small_wgrad_workspace(4 * Cc)
    ws = _workspace(x.device, need)
    check(lib.mi_conv1x1_small_cout_bwd(N * H * W, Cc, Cs, _p(x), ld_of(x), _b16(x), _p(dy), ld_of(dy), _p(w), _p(dW), _p(dx), ld_of(dx), _b16(dx),
                                        int(accumulate), _p(ws), need, _stream()), "mi_conv1x1_small_cout_bwd")


@functools.lru_cache(maxsize=None)
def small_cout_gn_supported(C, Cs, G):
    return bool(load_library().mi_conv1x1_small_cout_gn_supported(C, Cs, G))


def conv1x1_small_cout_gn(x, sums, gamma, beta, w, bias, Cs, groups=8, eps=1e-5):
    """Inference, final_conv: y = conv1x1(mish(groupnorm(x))) + bias with x the Block conv's bf16 output and its GroupNorm statistics taken from
    the sums that conv's epilogue left (conv3x3_bf16w(..., gn_sums=sums)) -- one launch, the normalised tensor is never written."""
    _need_gpu(x)
    assert x.dtype == torch.bfloat16 and sums.dtype == torch.int64
    N, H, W, Cc = x.shape
    y = torch.empty((N, H, W, 4), device=x.device, dtype=torch.float32)       # padded 4-channel pixels; the kernel writes the padding (zero) too
    e0 = _probe_open()
    check(load_library().mi_conv1x1_small_cout_gn_fwd(N * H * W, H * W, Cc, Cs, _p(x), ld_of(x), _p(sums), _p(gamma), _p(beta), groups, eps,
                                                      _p(w), _p(bias), _p(y), ld_of(y), _stream()), "mi_conv1x1_small_cout_gn_fwd")
    if e0 is not None:
        _probe_close(e0, f"small_cout_fwd_gn_kernel<{Cc // 32}>", 2.0 * N * H * W * Cc * Cs, f"M{N * H * W} C{Cc}->{Cs} GN+Mish in the load",
                     N * H * W * (Cc * 2 + 16))
    return y[..., :Cs]


@functools.lru_cache(maxsize=None)
def small_cin_dual_supported(N, H, W, Cin, Cout, ldx):
    return bool(load_library().mi_conv_small_cin_fwd_dual_supported(N, H, W, Cin, Cout, ldx))


def conv_small_cin_fwd_dual(x, w3, bias3, w1, bias1, Cout, out_dtype=torch.float32, zero=None, gather=None):
    """The first ResnetBlock's block1 conv (3x3) and its res_conv (1x1) on the same image in one launch -> (y3 in out_dtype, y1 fp32).
    This is a forward's first launch; it can do the forward's chores whose consumers come later -- zero: an int64 tensor it clears (the pool of
    GroupNorm sums); gather = (table [T, R], idx int64 [B], out [B, R]): out[b] = table[idx[b]] (the sampler's time-bias rows)."""
    _need_gpu(x)
    N, H, W, Cin = x.shape
    y3 = (torch.empty((N, H, W, Cout), device=x.device, dtype=torch.bfloat16) if out_dtype == torch.bfloat16 else new_act(N, H, W, Cout, x))
    y1 = new_act(N, H, W, Cout, x)
    e0 = _probe_open()
    if zero is not None or gather is not None:
        assert zero is None or (zero.dtype == torch.int64 and zero.is_contiguous())
        gt, gi, go = gather if gather is not None else (None, None, None)
        assert gather is None or (gt.is_contiguous() and go.is_contiguous() and gi.dtype == torch.int64 and go.shape == (gi.shape[0], gt.shape[1]))
        check(load_library().mi_conv_small_cin_fwd_dual_chores(N, H, W, Cin, Cout, _p(x), ld_of(x), _p(w3), _p(bias3), _p(y3), ld_of(y3), _b16(y3),
                                                             _p(w1), _p(bias1), _p(y1), ld_of(y1), _p(zero), zero.numel() * 8 if zero is not None else 0,
                                                             _p(gt), _p(gi), _p(go), gt.shape[1] if gather is not None else 0,
                                                             gi.shape[0] if gather is not None else 0, _stream()),
              "mi_conv_small_cin_fwd_dual_chores")
    else:
        check(load_library().mi_conv_small_cin_fwd_dual(N, H, W, Cin, Cout, _p(x), ld_of(x), _p(w3), _p(bias3), _p(y3), ld_of(y3), _b16(y3),
                                                        _p(w1), _p(bias1), _p(y1), ld_of(y1), _stream()), "mi_conv_small_cin_fwd_dual")
    if e0 is not None:
        _probe_close(e0, f"small_cin3x3_fwd_tiled_kernel<{Cin}, {'true' if _b16(y3) else 'false'}, true>", 2.0 * N * H * W * Cout * Cin * 10,
                     f"N{N} {H}x{W} K{Cin}->{Cout} 3x3 + res_conv", N * H * W * (16 + Cout * (_esz(y3) + 4)))
    return y3, y1


def conv_small_cin_wgrad(x, dy, dW, ks):
    N, H, W, Cin = x.shape
    lib = load_library()
    need = lib.mi_conv_small_wgrad_workspace(ks * ks * Cin * dy.shape[3])
    ws = _workspace(x.device, need)
    check(lib.mi_conv_small_cin_wgrad_io(ks, N, H, W, Cin, dy.shape[3], _p(x), ld_of(x), _p(dy), ld_of(dy), _b16(dy), _p(dW), _p(ws), need,
                                         _stream()), "mi_conv_small_cin_wgrad")


@functools.lru_cache(maxsize=None)      # pure function of the shape: one FFI query per distinct layer, not per step
def small_cout_supported(op, C, Cs):
    """Conv2d(C, Cs<=4, 1) kernels (mi_conv1x1_small_cout): op 0 forward, 1 dgrad, 2 wgrad."""
    if not 1 <= Cs <= 4 or C not in (32, 64, 128, 256):
        return False
    return C <= 128 if op == 0 else (C >= 64 if op == 2 else True)


def conv1x1_small_cout(op, a, w, *, b=None, bias=None, out=None, Cs=None, accumulate=False):
    """op 0: y[.,Cs] = x w + bias; op 1: dx[.,C] (+)= dy w^T; op 2: dW[C][Cs] += x^T dy.  w: [C][Cs]."""
    _need_gpu(a)
    N, H, W = a.shape[:3]
    M = N * H * W
    if op == 0:
        C, Cs = a.shape[3], Cs
        if out is None:
            buf = torch.empty((N, H, W, (Cs + 3) // 4 * 4), device=a.device, dtype=torch.float32)      # (the kernel writes the padding channels: zeros)
            out = buf[..., :Cs]
    elif op == 1:
        Cs = a.shape[3]; C = out.shape[3]
    else:
        C, Cs = a.shape[3], b.shape[3]
    lib = load_library()
    need = lib.mi_conv_small_wgrad_workspace(4 * C) if op == 2 else 0
    ws = _workspace(a.device, need) if op == 2 else None
    wide16 = _b16(out) if op == 1 else _b16(a)       # the C-channel tensor: x (op 0 / 2) or dx (op 1)
    check(lib.mi_conv1x1_small_cout_io(op, M, C, Cs, _p(a), ld_of(a), _p(b), ld_of(b) if b is not None else 0, _p(w),
                                       _p(bias), _p(out), ld_of(out) if out.dim() == 4 else Cs, int(accumulate), wide16, _p(ws), need, _stream()),
          "mi_conv1x1_small_cout")
    return out


def conv_wgrad(P, Q, dW, *, kh, kw, stride, pad, gather_i, Ci, Cj, grid_g, grid_d, mode, P2=None, dbias=None):
    """dW[tap][i][j] += sum P*Q (see MiWgradDesc).  dW: flat fp32 buffer of kh*kw*Ci*Cj.
    dbias (optional): dbias[j] += sum over pixels of Q -- fused when the fast kernel runs, else a colsum pass."""
    _need_gpu(P)
    N = P.shape[0]
    I1 = P.shape[3] if P2 is not None else Ci
    d = MiWgradDesc(N=N, GH=grid_g[0], GW=grid_g[1], DH=grid_d[0], DW=grid_d[1], Ci=Ci, Cj=Cj, KH=kh, KW=kw,
                    stride=stride, pad=pad, gather_i=int(gather_i), mode=mode, I1=I1, ldp=ld_of(P),
                    ldp2=ld_of(P2) if P2 is not None else 0, ldq=ld_of(Q))
    lib = load_library()
    fast = bool(lib.mi_conv3x3_wgrad_supported(C.byref(d)))
    use_tr = bool(USE_WGRAD_TR and dbias is None and kh == 3 and _b16(P) and _b16(Q) and (P2 is None or _b16(P2))
                  and _query("mi_conv3x3_wgrad_tr_supported", d))
    if not fast and not use_tr and (_b16(P) or _b16(Q)):
        raise RuntimeError("bf16-stored operands need the fast wgrad kernel (caller must check wgrad_supported)")
    flops = 2.0 * N * grid_d[0] * grid_d[1] * Ci * Cj * kh * kw
    nb_in = N * grid_g[0] * grid_g[1] * Ci * _esz(P) + N * grid_d[0] * grid_d[1] * Cj * _esz(Q)
    desc = f"N{N} {grid_d[0]}x{grid_d[1]} Ci{Ci}{'(2src)' if P2 is not None else ''} Cj{Cj} k{kh} s{stride} g{int(gather_i)}"
    if use_tr:
        # both operands bf16-stored: LDS-DMA + transposing-read kernel, all nine taps per workgroup
        need = lib.mi_conv3x3_wgrad_tr_workspace(C.byref(d))
        ws = _workspace(P.device, need)

        def go_tr():
            check(lib.mi_conv3x3_wgrad_tr(C.byref(d), _p(P), _p(P2), _p(Q), _p(dW), _p(ws), ws.numel() * 4, _stream()), "mi_conv3x3_wgrad_tr")
        if PROBE is None:
            go_tr()
        else:
            sp = lib.mi_conv3x3_wgrad_tr_splits(C.byref(d))
            part = 9.0 * 64 * 128 * 4 * sp * (Ci // 64) * ((Cj + 127) // 128)
            try:
                lib.mi_debug_wgrad_tr_phase(1)
                e0 = _probe_open(); go_tr(); _probe_close(e0, "wgrad_tr_kernel", flops, desc, nb_in + part)
                lib.mi_debug_wgrad_tr_phase(2)
                e0 = _probe_open(); go_tr()
                _probe_close(e0, "wgrad_tr_reduce_kernel", 0.0, desc + f" splits{sp}", part + 9.0 * Ci * Cj * 8)
            finally:
                lib.mi_debug_wgrad_tr_phase(0)
        return
    if fast:
        need = lib.mi_conv3x3_wgrad_workspace(C.byref(d))
        ws = _workspace(P.device, need)
        io = _b16(P) | (_b16(Q) << 1)

        def go():
            if io:
                check(lib.mi_conv3x3_wgrad_io(C.byref(d), _p(P), _p(P2), _p(Q), _p(dW), _p(dbias), _p(ws), ws.numel() * 4, io, _stream()),
                      "mi_conv3x3_wgrad_io")
            else:
                check(lib.mi_conv3x3_wgrad_bias(C.byref(d), _p(P), _p(P2), _p(Q), _p(dW), _p(dbias), _p(ws), ws.numel() * 4, _stream()),
                      "mi_conv3x3_wgrad_bias")
        if PROBE is None:
            go()
        else:                         # time the contraction kernel and the partial-tile reduce under separate events
            nj, sp = C.c_int(), C.c_int()
            lib.mi_conv3x3_wgrad_tile(C.byref(d), C.byref(nj), C.byref(sp))
            sym = f"wgrad3x3_kernel<{nj.value}, {kh}, {io}>"
            part = kh * kw * Ci * Cj * 4.0 * sp.value
            if sp.value > 1:
                try:
                    lib.mi_debug_wgrad3x3_phase(1)
                    e0 = _probe_open(); go(); _probe_close(e0, sym, flops, desc, nb_in + part)
                    lib.mi_debug_wgrad3x3_phase(2)
                    e0 = _probe_open(); go()
                    _probe_close(e0, f"wgrad_reduce_kernel<{16 if sp.value >= 128 else 4}>", 0.0, desc + f" splits{sp.value}", part + kh * kw * Ci * Cj * 8.0)
                finally:
                    lib.mi_debug_wgrad3x3_phase(0)
            else:
                e0 = _probe_open(); go(); _probe_close(e0, sym, flops, desc, nb_in + kh * kw * Ci * Cj * 8.0)
    else:
        e0 = _probe_open()
        check(lib.mi_conv_wgrad(C.byref(d), _p(P), _p(P2), _p(Q), _p(dW), _stream()), "mi_conv_wgrad")
        if e0 is not None:
            _probe_close(e0, f"wgrad_kernel<{mode}>", flops, desc, nb_in + kh * kw * Ci * Cj * 8.0)
        if dbias is not None:
            colsum(Q, dbias)


@functools.lru_cache(maxsize=None)      # pure function of the shape: one FFI query per distinct layer, not per step
def conv_s2_wgrad_f32(P, Q, dW, *, k, Ci, Cj, gather_i, grid_g, grid_d):
    """Exact-fp32 mode: the weight gradient of Downsample (3x3 / stride 2, gather_i: P is the big tensor) or Upsample (ConvTranspose 4x4 /
    stride 2: Q is) as k x k gathered problems of the fp32 1x1 weight-gradient kernel (mi_conv_s2_wgrad_f32).  dW [k][k][Ci][Cj] +=.
    Returns True when the library took the layer, False when the caller has to use conv_wgrad."""
    if P.dtype != torch.float32 or Q.dtype != torch.float32:
        return False
    _need_gpu(P)
    N = P.shape[0]
    d = MiWgradDesc(N=N, GH=grid_g[0], GW=grid_g[1], DH=grid_d[0], DW=grid_d[1], Ci=Ci, Cj=Cj, KH=k, KW=k, stride=2, pad=1,
                    gather_i=int(gather_i), mode=MODE_FP32, I1=Ci, ldp=ld_of(P), ldp2=0, ldq=ld_of(Q))
    if not _query("mi_conv_s2_wgrad_f32_supported", d):
        return False
    lib = load_library()
    need = lib.mi_conv_s2_wgrad_f32_workspace(C.byref(d))
    ws = _workspace(P.device, need)
    e0 = _probe_open()
    check(lib.mi_conv_s2_wgrad_f32(C.byref(d), _p(P), _p(Q), _p(dW), _p(ws), ws.numel() * 4, _stream()), "mi_conv_s2_wgrad_f32")
    if e0 is not None:
        _probe_close(e0, "wgrad1x1_f32_kernel[s2 taps]", 2.0 * N * grid_d[0] * grid_d[1] * Ci * Cj * k * k, f"N{N} {grid_d[0]}x{grid_d[1]} Ci{Ci} Cj{Cj} k{k}",
                     N * (grid_g[0] * grid_g[1] * (Ci if gather_i else Cj) + grid_d[0] * grid_d[1] * (Cj if gather_i else Ci)) * 4.0)
    return True


def s2_wgrad_supported(N, k, Ci, Cj, transposed, big, small, mode):
    """Can the stride-2 LDS-DMA weight-gradient kernel take this Downsample (3x3 conv) / Upsample (4x4 transposed conv) layer
    (dense bf16 operands)?  big / small: the (H, W) of the 2x-resolution tensor and of the other one."""
    d = MiWgradDesc(N=N, GH=big[0], GW=big[1], DH=small[0], DW=small[1], Ci=Ci, Cj=Cj, KH=k, KW=k, stride=2, pad=1,
                    gather_i=int(not transposed), mode=mode, I1=Ci, ldp=Ci, ldp2=0, ldq=Cj)
    return bool(USE_WGRAD_TR and load_library().mi_conv_s2_wgrad_tr_supported(C.byref(d)))


class WgradQueue:
    """Deferred weight gradients of the Block convs (3x3) and of the 1x1 convs.  Backward pushes (X, dY, dW) here instead of
    launching one full-chip kernel per layer; every `group` layers of a kind go out as ONE launch (mi_conv3x3_wgrad_tr_batch /
    mi_conv1x1_wgrad_tr_batch), each layer on its share of the CUs (see include/mi_ddpm.h: an eighth of the partial-tile traffic,
    an eighth of the launches).  Layers the LDS-DMA kernels cannot take run at once through conv_wgrad.  `pushed` numbers the deferred
    layers (1, 2, ...); `flushed` = the largest n such that layers 1..n have ALL been issued (the two kinds flush independently, so a
    count of issued layers would not say that); `on_flush()` is called after every flush.
    Kind 4 (round 4): reductions that leave one partial row per workgroup (channel LayerNorm's dg / db) push (rows, dst) pairs; ONE
    mi_rowsum_batch launch adds all of them (at the final flush, or right away when a caller flushes kind 4: data parallelism)."""

    def __init__(self, group: int = 8, on_flush=None, group_s2=None):
        self.group, self.on_flush = max(1, min(8, int(group))), on_flush
        self.group_s2 = self.group if group_s2 is None else max(1, min(8, int(group_s2)))    # stride-2 layers per launch
        self.items3, self.items1, self.items2, self.items4 = [], [], [], []
        self.pushed = 0
        self._seq3, self._seq1, self._seq2, self._seq4 = [], [], [], []        # sequence numbers of the queued layers, per kind
        self._arena_top = 0                                    # partial rows of this backward pass (see rows_buffer)

    @property
    def flushed(self) -> int:
        pending = self._seq3[:1] + self._seq1[:1] + self._seq2[:1] + self._seq4[:1]
        return min(pending) - 1 if pending else self.pushed

    def _desc(self, P, Q, k, Ci, Cj, hw, mode, P2):
        I1 = P.shape[3] if P2 is not None else Ci
        return MiWgradDesc(N=P.shape[0], GH=hw[0], GW=hw[1], DH=hw[0], DW=hw[1], Ci=Ci, Cj=Cj, KH=k, KW=k, stride=1, pad=k // 2, gather_i=1,
                           mode=mode, I1=I1, ldp=ld_of(P), ldp2=ld_of(P2) if P2 is not None else 0, ldq=ld_of(Q))

    def push(self, P, Q, dW, *, Ci, Cj, hw, mode, P2=None):
        """3x3 / stride 1 / pad 1"""
        d = self._desc(P, Q, 3, Ci, Cj, hw, mode, P2)
        same = (lambda dt: P.dtype == dt and Q.dtype == dt and (P2 is None or P2.dtype == dt))      # noqa: E731
        ok = (USE_WGRAD_TR and self.group > 1 and (same(torch.bfloat16) if mode == MODE_BF16 else same(torch.float32))
              and _query("mi_conv3x3_wgrad_tr_supported", d))      # (mode 0: the exact-fp32 instantiation, fp32 operands)
        if not ok:
            conv_wgrad(P, Q, dW, kh=3, kw=3, stride=1, pad=1, gather_i=True, Ci=Ci, Cj=Cj, grid_g=hw, grid_d=hw, mode=mode, P2=P2)
            return
        self.items3.append((d, P, P2, Q, dW))
        self.pushed += 1
        self._seq3.append(self.pushed)
        if len(self.items3) >= self.group:
            self.flush(kinds=(3,))

    def push1x1(self, P, Q, dW, *, Ci, Cj, hw, mode, P2=None, dbias=None):
        """1x1; dbias (optional) += column sums of Q"""
        d = self._desc(P, Q, 1, Ci, Cj, hw, mode, P2)
        q32 = int(Q.dtype == torch.float32)
        if mode == MODE_FP32:       # exact-fp32 instantiation (round 4): fp32 X and dY
            dt_ok = P.dtype == torch.float32 and (P2 is None or P2.dtype == torch.float32) and q32
        else:
            dt_ok = _b16(P) and (P2 is None or _b16(P2))
        ok = (USE_WGRAD_TR and self.group > 1 and dt_ok and (dbias is None or q32)
              and _query("mi_conv1x1_wgrad_tr_supported", d, q32))
        if not ok:
            conv_wgrad(P, Q, dW, kh=1, kw=1, stride=1, pad=0, gather_i=True, Ci=Ci, Cj=Cj, grid_g=hw, grid_d=hw, mode=mode, P2=P2, dbias=dbias)
            return
        self.items1.append((d, P, P2, Q, dW, dbias, q32))
        self.pushed += 1
        self._seq1.append(self.pushed)
        if len(self.items1) >= self.group:
            self.flush(kinds=(1,))

    def push_s2(self, P, Q, dW, *, k, Ci, Cj, gather_i, grid_g, grid_d, mode):
        """Downsample (3x3 / stride 2 / pad 1 conv: gather_i, P = X on the 2x grid) and Upsample (4x4 / stride 2 / pad 1 transposed
        conv: not gather_i, Q = dY on the 2x grid) weight gradients; both operands bf16.  Returns False when the LDS-DMA kernel
        cannot take the layer (the caller then runs conv_wgrad)."""
        d = MiWgradDesc(N=P.shape[0], GH=grid_g[0], GW=grid_g[1], DH=grid_d[0], DW=grid_d[1], Ci=Ci, Cj=Cj, KH=k, KW=k, stride=2, pad=1,
                        gather_i=int(gather_i), mode=mode, I1=Ci, ldp=ld_of(P), ldp2=0, ldq=ld_of(Q))
        if not (USE_WGRAD_TR and _b16(P) and _b16(Q) and _query("mi_conv_s2_wgrad_tr_supported", d)):
            return False
        self.items2.append((d, P, None, Q, dW))
        self.pushed += 1
        self._seq2.append(self.pushed)
        if len(self.items2) >= self.group_s2:
            self.flush(kinds=(2,))
        return True

    def rows_buffer(self, device, nfloats):
        """nfloats of the persistent partial-row arena (static address: capturable); valid until the next backward pass"""
        n = (int(nfloats) + 63) // 64 * 64
        arena = _rows_arena(device, (self._arena_top + n) * 4)
        out = arena[self._arena_top:self._arena_top + n]
        self._arena_top += n
        return out

    def push_rowsum(self, src, rows, cols, ld, dst):
        """dst[c] += sum_r src[r * ld + c] (deferred)"""
        self.items4.append((src, int(rows), int(cols), int(ld), dst))
        self.pushed += 1
        self._seq4.append(self.pushed)

    def flush(self, kinds=(3, 1, 2, 4)):
        lib = load_library()
        if 4 in kinds and self.items4:
            items, self.items4, self._seq4 = self.items4, [], []
            _need_gpu(items[0][0])
            for i0 in range(0, len(items), ROWSUM_MAX):
                chunk = items[i0:i0 + ROWSUM_MAX]
                arr4 = (MiRowSum * len(chunk))(*[MiRowSum(src=it[0].data_ptr(), dst=it[4].data_ptr(), rows=it[1], cols=it[2], ld=it[3], pad_=0)
                                                 for it in chunk])
                e0 = _probe_open()
                check(lib.mi_rowsum_batch(len(chunk), arr4, _stream()), "mi_rowsum_batch")
                if e0 is not None:
                    _probe_close(e0, "rowsum_batch_kernel", 0.0, f"{len(chunk)} items", sum(it[1] * it[2] * 4.0 for it in chunk))
        arr = lambda items, k: (C.c_void_p * len(items))(*[(it[k].data_ptr() if it[k] is not None else 0) for it in items])   # noqa: E731
        if 3 in kinds and self.items3:
            items, self.items3, self._seq3 = self.items3, [], []
            _need_gpu(items[0][1])
            n = len(items)
            descs = (MiWgradDesc * n)(*[it[0] for it in items])
            need = lib.mi_conv3x3_wgrad_tr_batch_workspace(n, descs)
            ws = _workspace(items[0][1].device, need)

            def go():
                check(lib.mi_conv3x3_wgrad_tr_batch(n, descs, arr(items, 1), arr(items, 2), arr(items, 3), arr(items, 4), _p(ws), ws.numel() * 4,
                                                    _stream()), "mi_conv3x3_wgrad_tr_batch")
            flops = sum(2.0 * it[0].N * it[0].DH * it[0].DW * it[0].Ci * it[0].Cj * 9 for it in items)
            f32 = items[0][0].mode == MODE_FP32
            nb = sum(it[0].N * it[0].DH * it[0].DW * (it[0].Ci + it[0].Cj) * (4.0 if f32 else 2.0) for it in items)
            self._run(go, lib.mi_debug_wgrad_tr_phase, "wgrad_tr32" if f32 else "wgrad_tr", n, flops, nb, need)
        if 1 in kinds and self.items1:
            items, self.items1, self._seq1 = self.items1, [], []
            _need_gpu(items[0][1])
            n = len(items)
            descs = (MiWgradDesc * n)(*[it[0] for it in items])
            q32 = (C.c_int * n)(*[it[6] for it in items])
            need = lib.mi_conv1x1_wgrad_tr_batch_workspace(n, descs, q32)
            ws = _workspace(items[0][1].device, need)

            def go1():
                check(lib.mi_conv1x1_wgrad_tr_batch(n, descs, q32, arr(items, 1), arr(items, 2), arr(items, 3), arr(items, 4), arr(items, 5),
                                                    _p(ws), ws.numel() * 4, _stream()), "mi_conv1x1_wgrad_tr_batch")
            flops = sum(2.0 * it[0].N * it[0].DH * it[0].DW * it[0].Ci * it[0].Cj for it in items)
            f32 = items[0][0].mode == MODE_FP32
            nb = sum(it[0].N * it[0].DH * it[0].DW * (it[0].Ci * (4.0 if f32 else 2.0) + it[0].Cj * (4.0 if it[6] else 2.0)) for it in items)
            self._run(go1, lib.mi_debug_wgrad1x1_tr_phase, "wgrad1x1_f32" if f32 else "wgrad1x1_tr", n, flops, nb, need)
        if 2 in kinds and self.items2:
            items, self.items2, self._seq2 = self.items2, [], []
            _need_gpu(items[0][1])
            n = len(items)
            descs = (MiWgradDesc * n)(*[it[0] for it in items])
            need = lib.mi_conv_s2_wgrad_tr_batch_workspace(n, descs)
            ws = _workspace(items[0][1].device, need)

            def go2():
                check(lib.mi_conv_s2_wgrad_tr_batch(n, descs, arr(items, 1), arr(items, 3), arr(items, 4), _p(ws), ws.numel() * 4, _stream()),
                      "mi_conv_s2_wgrad_tr_batch")
            flops = sum(2.0 * it[0].N * it[0].DH * it[0].DW * it[0].Ci * it[0].Cj * it[0].KH * it[0].KW for it in items)
            nb = sum(it[0].N * it[0].DH * it[0].DW * 2.0 * ((4 if it[0].gather_i else 1) * it[0].Ci + (1 if it[0].gather_i else 4) * it[0].Cj)
                     for it in items)
            self._run(go2, lib.mi_debug_wgrad_s2_tr_phase, "wgrad_s2_tr", n, flops, nb, need)
        if self.on_flush is not None:
            self.on_flush()

    @staticmethod
    def _run(go, phase, name, n, flops, nb, need):
        if PROBE is None:
            go()
            return
        try:                                   # contraction and partial-tile reduce under separate events
            phase(1)
            e0 = _probe_open(); go(); _probe_close(e0, name + "_kernel", flops, f"{n} layers", nb + need)
            phase(2)
            e0 = _probe_open(); go(); _probe_close(e0, name.replace("tr32", "tr").replace("1x1_f32", "1x1_tr") + "_reduce_kernel", 0.0, f"{n} layers", float(need))
        finally:
            phase(0)


_WS = {}
_WS_RETIRED = []      # outgrown workspaces stay allocated: a captured hipGraph may still write its partial tiles there
_ROWS = {}


def _rows_arena(device, nbytes):
    """Persistent per-device buffer of the partial rows a backward pass defers (WgradQueue.rows_buffer); grows like _workspace."""
    cur = _ROWS.get(device)
    if cur is None or cur.numel() * 4 < nbytes:
        if cur is not None:
            _WS_RETIRED.append(cur)
        cur = torch.empty((max(int(nbytes) * 2, 16 << 20) + 3) // 4, device=device, dtype=torch.float32)
        _ROWS[device] = cur
    return cur


def _workspace(device, nbytes):
    """Persistent per-device scratch.  The address a launch saw stays valid for the life of the process (captured hipGraphs
    keep referencing it): when a later call needs more, a larger buffer is added and the old one is retired, never freed."""
    cur = _WS.get(device)
    if cur is None or cur.numel() * 4 < nbytes:
        if cur is not None:
            _WS_RETIRED.append(cur)
        cur = torch.empty((max(int(nbytes), 64 << 20) + 3) // 4, device=device, dtype=torch.float32)
        _WS[device] = cur
    return cur


def _rows(x):
    return x.shape[0] * x.shape[1] * x.shape[2] if x.dim() == 4 else x.shape[0]


def colsum(x, out, defer=None):
    """out[c] += sum over pixels of x[..., c]; defer (a WgradQueue): as partial rows, added by the queue's batched row sum"""
    Cc = x.shape[-1]
    M = _rows(x)
    if defer is not None and Cc % 4 == 0 and 4 <= Cc <= 1024 and _colsum_deferred(x, None, out, defer):
        return
    e0 = _probe_open()
    vec = Cc % 4 == 0 and 4 <= Cc <= 1024 and ld_of(x) % 4 == 0 and x.data_ptr() % 16 == 0 and x.dtype == torch.float32
    lib = load_library()
    if vec:       # float4 rows on a full grid, per-workgroup partial sums through the scratch buffer
        ws = _workspace(x.device, _cvt_colsum_ws(M, Cc))
        check(lib.mi_f32_to_bf16_colsum(M, Cc, _p(x), ld_of(x), None, 0, _p(out), _p(ws), ws.numel() * 4, _stream()), "mi_f32_to_bf16_colsum")
    else:
        check(lib.mi_colsum(M, Cc, _p(x), ld_of(x), _p(out), _stream()), "mi_colsum")
    if e0 is not None:
        _probe_close(e0, "cvt_colsum_kernel<false, true>" if vec else "colsum_kernel", 0.0, f"M{M} C{Cc}", M * Cc * _esz(x))


# --------------------------------------------------------------------------- norms
def _colsum_deferred(x, y16, out, defer):
    """partial rows of x's column sums (and the bf16 copy y16, if given) now, their sum into `out` with the queue's batched row sum;
    False: the tensor is too small for that (the caller takes the direct path)"""
    M, Cc = _rows(x), x.shape[-1]
    need = _cvt_colsum_ws(M, Cc)
    if M < 4096 or not need or x.dtype != torch.float32 or ld_of(x) % 4 or x.data_ptr() % 16:
        return False
    part = defer.rows_buffer(x.device, need // 4)
    rows = C.c_int(0)
    e0 = _probe_open()
    check(load_library().mi_f32_to_bf16_colsum_part(M, Cc, _p(x), ld_of(x), _p(y16), Cc, _p(part), need, C.byref(rows), _stream()),
          "mi_f32_to_bf16_colsum_part")
    if e0 is not None:
        _probe_close(e0, "cvt_colsum_kernel<true, true>" if y16 is not None else "cvt_colsum_kernel<false, true>", 0.0, f"M{M} C{Cc} (rows)",
                     M * Cc * (6.0 if y16 is not None else 4.0))
    assert rows.value > 0
    defer.push_rowsum(part, rows.value, Cc, Cc, out)
    return True


def to_bf16(x, colsum_out=None, defer=None):
    """bf16 copy (round-to-nearest-even) of an fp32 NHWC activation or channel slice: the MFMA operand of the next conv /
    weight gradient, rounded once for all of its consumers.  colsum_out (optional)[c] += sum over pixels of x[..., c] from the
    same pass (the bias gradient when x is a conv's output gradient); defer (a WgradQueue): the sums leave as partial rows and are
    added by the queue's batched row sum (no second pass per tensor)."""
    _need_gpu(x)
    N, H, W, Cc = x.shape
    y = new_act(N, H, W, Cc, x, torch.bfloat16)
    if colsum_out is not None and defer is not None and _colsum_deferred(x, y, colsum_out, defer):
        return y
    e0 = _probe_open()
    if colsum_out is not None:
        lib = load_library()
        ws = _workspace(x.device, _cvt_colsum_ws(N * H * W, Cc))
        check(lib.mi_f32_to_bf16_colsum(N * H * W, Cc, _p(x), ld_of(x), _p(y), Cc, _p(colsum_out), _p(ws), ws.numel() * 4, _stream()),
              "mi_f32_to_bf16_colsum")
    else:
        check(load_library().mi_f32_to_bf16(N * H * W, Cc, _p(x), ld_of(x), _p(y), Cc, _stream()), "mi_f32_to_bf16")
    if e0 is not None:
        _probe_close(e0, "cvt_colsum_kernel<true, true>" if colsum_out is not None else "f32_to_bf16_kernel", 0.0,
                     f"M{N * H * W} C{Cc}", N * H * W * Cc * 6.0)
    return y


def gn_mish_fwd(x, gamma, beta, *, groups=8, eps=1e-5, temb=None, residual=None, out_dtype=torch.float32, want16=False):
    """-> (y, stats), or (y, stats, y16) with want16: y16 = y rounded to bf16, written by the same pass."""
    _need_gpu(x)
    N, H, W, Cc = x.shape
    y = new_act(N, H, W, Cc, x, out_dtype)
    if want16:
        y16 = new_act(N, H, W, Cc, x, torch.bfloat16)
        stats = torch.empty((N, groups, 2), device=x.device, dtype=torch.float32)
        d = MiGnDesc(N=N, HW=H * W, C=Cc, G=groups, eps=eps, ldx=ld_of(x), ldy=ld_of(y),
                     ldr=ld_of(residual) if residual is not None else 0)
        io = _b16(x) | (_b16(y) << 1)
        e0 = _probe_open()
        check(load_library().mi_gn_mish_fwd_dual(C.byref(d), _p(x), _p(gamma), _p(beta), _p(temb), ld_of(temb) if temb is not None else 0,
                                                 _p(residual), _p(y), _p(y16), Cc, _p(stats), io, _stream()), "mi_gn_mish_fwd_dual")
        if e0 is not None:
            _probe_close(e0, f"gn_mish_fwd_kernel<io{io}>", 0.0, f"N{N} HW{H * W} C{Cc} res{int(residual is not None)} +bf16 copy",
                         N * H * W * Cc * (_esz(x) + _esz(y) + _esz(residual) + 2))
        return y, stats, y16
    stats = torch.empty((N, groups, 2), device=x.device, dtype=torch.float32)
    d = MiGnDesc(N=N, HW=H * W, C=Cc, G=groups, eps=eps, ldx=ld_of(x), ldy=ld_of(y),
                 ldr=ld_of(residual) if residual is not None else 0)
    io = _b16(x) | (_b16(y) << 1)
    ldt = ld_of(temb) if temb is not None else 0
    e0 = _probe_open()
    if io:
        check(load_library().mi_gn_mish_fwd_io(C.byref(d), _p(x), _p(gamma), _p(beta), _p(temb), ldt, _p(residual), _p(y),
                                               _p(stats), io, _stream()), "mi_gn_mish_fwd_io")
    else:
        check(load_library().mi_gn_mish_fwd(C.byref(d), _p(x), _p(gamma), _p(beta), _p(temb), ldt, _p(residual), _p(y),
                                            _p(stats), _stream()), "mi_gn_mish_fwd")
    if e0 is not None:
        _probe_close(e0, f"gn_mish_fwd_kernel<io{io}>", 0.0, f"N{N} HW{H * W} C{Cc} res{int(residual is not None)}",
                     N * H * W * Cc * (_esz(x) + _esz(y) + _esz(residual)))
    return y, stats


def gn_mish_bwd(x, stats, gamma, beta, dout, *, groups=8, eps=1e-5, dgamma=None, dbeta=None, dtemb=None, dbias=None,
                out_dtype=torch.float32):
    N, H, W, Cc = x.shape
    dx = new_act(N, H, W, Cc, x, out_dtype)
    d = MiGnDesc(N=N, HW=H * W, C=Cc, G=groups, eps=eps, ldx=ld_of(x), ldy=0, ldr=0)
    io = _b16(x) | (_b16(dx) << 1) | (_b16(dout) << 2)
    ldt = ld_of(dtemb) if dtemb is not None else 0
    e0 = _probe_open()
    if io:
        check(load_library().mi_gn_mish_bwd_io(C.byref(d), _p(x), _p(stats), _p(gamma), _p(beta), _p(dout), ld_of(dout),
                                               _p(dx), ld_of(dx), _p(dgamma), _p(dbeta), _p(dtemb), ldt, _p(dbias), io,
                                               _stream()), "mi_gn_mish_bwd_io")
    else:
        check(load_library().mi_gn_mish_bwd(C.byref(d), _p(x), _p(stats), _p(gamma), _p(beta), _p(dout), ld_of(dout),
                                            _p(dx), ld_of(dx), _p(dgamma), _p(dbeta), _p(dtemb), ldt, _p(dbias), _stream()),
              "mi_gn_mish_bwd")
    if e0 is not None:
        _probe_close(e0, f"gn_mish_bwd_kernel<io{io}>", 0.0, f"N{N} HW{H * W} C{Cc}", N * H * W * Cc * (_esz(x) + _esz(dout) + _esz(dx)))
    return dx


def chan_layernorm_fwd(x, g, b, eps=1e-5, out_dtype=torch.float32):
    _need_gpu(x)
    N, H, W, Cc = x.shape
    y = new_act(N, H, W, Cc, x, out_dtype)
    e0 = _probe_open()
    check(load_library().mi_chan_layernorm_fwd_io(N * H * W, Cc, _p(x), ld_of(x), _p(g), _p(b), eps, _p(y), ld_of(y), _b16(y),
                                                  _stream()), "mi_chan_layernorm_fwd")
    if e0 is not None:
        _probe_close(e0, f"chan_ln_fwd_kernel<io{_b16(y)}>", 0.0, f"M{N * H * W} C{Cc}", N * H * W * Cc * (_esz(x) + _esz(y)))
    return y


def ln_conv1x1_supported(N, H, W, K, Nc, ldx=None):
    """Does the LayerNorm + 1x1 conv kernel (mi_ln_conv1x1_pw) take the layer?"""
    d = MiConvDesc(N=N, IH=H, IW=W, OH=H, OW=W, K=K, Nc=Nc, KH=1, KW=1, stride=1, pad=0, transposed=0, w_kn=0, mode=MODE_BF16, K1=K,
                   ldx=K if ldx is None else ldx, ldx2=0, ldy=Nc, ldr=0, accumulate=0)
    return bool(_query("mi_ln_conv1x1_pw_supported", d))


def ln_conv1x1(x, g, b, wq, *, Nc, eps=1e-5, bias=None, want_ln=False):
    """y = conv1x1(chan_layernorm(x)) as bf16 in ONE launch: PreNorm + to_qkv, reference src/models/ddpm.py:85-106,151.  x: the fp32
    residual stream [N, H, W, K], wq: the conv's slice of pack_weights_bf16's wfq.  Inference: the normalised tensor is never written;
    want_ln (training: to_qkv's weight gradient reads it): -> (y, ln), ln bf16, written by the same launch."""
    _need_gpu(x)
    assert x.dtype == torch.float32
    N, H, W, K = x.shape
    y = new_act(N, H, W, Nc, x, torch.bfloat16)
    ln = new_act(N, H, W, K, x, torch.bfloat16) if want_ln else None
    d = MiConvDesc(N=N, IH=H, IW=W, OH=H, OW=W, K=K, Nc=Nc, KH=1, KW=1, stride=1, pad=0, transposed=0, w_kn=0, mode=MODE_BF16, K1=K,
                   ldx=ld_of(x), ldx2=0, ldy=ld_of(y), ldr=0, accumulate=0)
    e0 = _probe_open()
    if want_ln:
        check(load_library().mi_ln_conv1x1_pw_dual(C.byref(d), _p(x), _p(g), _p(b), eps, _p(wq), _p(bias), _p(y), _p(ln), ld_of(ln), _stream()),
              "mi_ln_conv1x1_pw_dual")
    else:
        check(load_library().mi_ln_conv1x1_pw(C.byref(d), _p(x), _p(g), _p(b), eps, _p(wq), _p(bias), _p(y), _stream()), "mi_ln_conv1x1_pw")
    if e0 is not None:
        px, nch = (128 if (K == 128 and N * H * W // 128 >= 512) else 64), K // 128
        g2 = "true" if (px == 64 and N * H * W // 64 < 512) else "false"      # mi_ln_conv1x1_pw's pick, restated for the symbol name only
        _probe_close(e0, f"ln_conv1x1_pw_kernel<{px}, {nch}, {g2}>", 2.0 * N * H * W * Nc * K, f"N{N} {H}x{W} K{K}->{Nc} fp32 in + LayerNorm",
                     N * H * W * (K * 4 + Nc * 2 + (2 * K if want_ln else 0)) + K * Nc * 2)
    return (y, ln) if want_ln else y


def chan_layernorm_bwd(x, g, dy, dx, accumulate, dg, db, eps=1e-5, defer=None):
    """defer (a WgradQueue): dg / db leave as one partial row per workgroup and are added by the queue's batched row sum -- the
    2 C atomics of every workgroup on the same addresses cost ~10 us per launch."""
    N, H, W, Cc = x.shape
    e0 = _probe_open()
    lib = load_library()
    if defer is not None and dg is not None and db is not None:
        rows = lib.mi_chan_layernorm_bwd_part_rows(N * H * W, Cc)
        part = defer.rows_buffer(x.device, rows * 2 * Cc)
        check(lib.mi_chan_layernorm_bwd_part(N * H * W, Cc, _p(x), ld_of(x), _p(g), eps, _p(dy), ld_of(dy), _p(dx), ld_of(dx),
                                             int(accumulate), _p(part), _b16(dy), _stream()), "mi_chan_layernorm_bwd_part")
        defer.push_rowsum(part, rows, Cc, 2 * Cc, dg)
        defer.push_rowsum(part[Cc:], rows, Cc, 2 * Cc, db)
    else:
        check(lib.mi_chan_layernorm_bwd_io(N * H * W, Cc, _p(x), ld_of(x), _p(g), eps, _p(dy), ld_of(dy), _p(dx),
                                           ld_of(dx), int(accumulate), _p(dg), _p(db), _b16(dy), _stream()),
              "mi_chan_layernorm_bwd")
    if e0 is not None:
        _probe_close(e0, f"chan_ln_bwd_kernel<io{_b16(dy)}>", 0.0, f"M{N * H * W} C{Cc} acc{int(accumulate)}",
                     N * H * W * Cc * (_esz(x) + _esz(dy) + _esz(dx) * (2 if accumulate else 1)))


# --------------------------------------------------------------------------- nn.Linear of the time MLP (exact fp32)
def small_gemm(ta, tb, A, B, *, bias=None, out=None, accumulate=False, allow_split=False):
    """out[i][j] (+)= bias[j] + sum_k opA(i,k) opB(k,j) (mi_small_gemm); returns None when the shape is not supported.
    allow_split: long contractions may be split and combined with atomics (gradients; not bit-reproducible)."""
    _need_gpu(A)
    I = A.shape[1] if ta else A.shape[0]
    Kc = A.shape[0] if ta else A.shape[1]
    J = B.shape[0] if tb else B.shape[1]
    lib = load_library()
    if (A.dtype != torch.float32 or B.dtype != torch.float32 or A.stride(1) != 1 or B.stride(1) != 1 or
            (A.data_ptr() | B.data_ptr()) & 15 or not lib.mi_small_gemm_supported(int(ta), int(tb), I, J, Kc, A.stride(0), B.stride(0))):
        return None
    if out is None:
        assert not accumulate
        out = torch.empty((I, J), device=A.device, dtype=torch.float32)
    check(lib.mi_small_gemm(int(ta), int(tb), I, J, Kc, _p(A), A.stride(0), _p(B), B.stride(0), _p(bias), _p(out), out.stride(0),
                            int(accumulate), int(allow_split), _stream()), "mi_small_gemm")
    return out


# --------------------------------------------------------------------------- VQ-VAE codebook step (vqvae.py:24-43)
def vq_nearest(z_rows, codebook):
    """z_rows [M, D] (row stride % 4 == 0), codebook [K, D] -> (idx int32 [M], zq [M, D], sum ||z - zq||^2 as a 0-dim tensor)."""
    _need_gpu(z_rows)
    M, D = z_rows.shape
    Kc = codebook.shape[0]
    assert z_rows.stride(1) == 1 and codebook.is_contiguous() and codebook.shape[1] == D
    idx = torch.empty(M, device=z_rows.device, dtype=torch.int32)
    zq = torch.empty((M, D), device=z_rows.device, dtype=torch.float32)
    part = torch.empty(load_library().mi_vq_partials(M), device=z_rows.device, dtype=torch.float32)
    check(load_library().mi_vq_nearest_fwd(M, D, Kc, _p(z_rows), z_rows.stride(0), _p(codebook), _p(idx), _p(zq), D, _p(part), _stream()),
          "mi_vq_nearest_fwd")
    return idx, zq, part.sum()


def vq_backward(z_rows, codebook, idx, g_vq, g_commit, dz=None, accumulate=False, dcodebook=None, g_dev=None):
    """Adds the gradients of g_vq * vq_loss + g_commit * mean((z - sg(q))^2) into dz / dcodebook (either may be None);
    g_dev: optional 2-float device tensor multiplied into (g_vq, g_commit)."""
    M, D = z_rows.shape
    check(load_library().mi_vq_bwd(M, D, codebook.shape[0], _p(z_rows), z_rows.stride(0), _p(codebook), _p(idx), float(g_vq), float(g_commit),
                                   _p(g_dev), _p(dz), dz.stride(0) if dz is not None else 0, int(accumulate), _p(dcodebook), _stream()), "mi_vq_bwd")


def vq_scatter_rows(src_rows, idx, table):
    """table[idx[m]] += src_rows[m]."""
    M, D = src_rows.shape
    check(load_library().mi_vq_scatter_rows(M, D, table.shape[0], _p(src_rows), src_rows.stride(0), _p(idx), _p(table), _stream()),
          "mi_vq_scatter_rows")


# --------------------------------------------------------------------------- attention
def linattn_fwd(qkv, heads=4):
    _need_gpu(qkv)
    N, H, W, C3 = qkv.shape
    assert C3 == 3 * heads * 32 and qkv.is_contiguous()
    out = new_act(N, H, W, heads * 32, qkv, qkv.dtype)        # bf16 qkv -> bf16 output (attention-internal storage)
    ctx = torch.empty((N, heads, 32, 32), device=qkv.device, dtype=torch.float32)
    kstat = torch.empty((N, heads, 32, 2), device=qkv.device, dtype=torch.float32)
    e0 = _probe_open()
    lib = load_library()
    need = _linattn_ws(N, H * W, heads)                      # > 0: the pixel axis is cut into slices (several workgroups per image)
    ws = _workspace(qkv.device, need) if need else None
    check(lib.mi_linattn_fwd_ws(N, H * W, heads, _p(qkv), _p(out), _p(ctx), _p(kstat), _b16(qkv), _p(ws), ws.numel() * 4 if need else 0,
                                _stream()), "mi_linattn_fwd")
    if e0 is not None:
        _probe_close(e0, f"linattn_fwd_kernel<io{_b16(qkv)}>", 4.0 * N * heads * 32 * 32 * H * W, f"N{N} n{H * W}",
                     N * H * W * heads * 32 * 4 * _esz(qkv))
    return out, ctx, kstat


def linattn_to_out_folded(qkv, w_out16, Cc, bias, residual, heads=4, want16=False):
    """Inference: LinearAttention + its to_out conv + the Residual add (reference ddpm.py:146-165, Residual) as TWO launches that never form the
    attention output: (1) mi_linattn_fold_fwd -- softmax statistics and context per (sample, head) as in linattn_fwd, then the per-sample
    effective weights W_eff[b] = W_out blockdiag(ctx_h^T) in fragment order; (2) mi_conv1x1_pw_batched -- y = W_eff[b] q + bias + residual,
    q read straight out of qkv.  qkv bf16 [N, H, W, 3 * heads * 32], w_out16: to_out's bf16 weight rows [C][heads * 32] (pack_weights_bf16's wf slice).
    -> y fp32 (want16: (y, its bf16 copy)); None when the shape is not taken."""
    _need_gpu(qkv)
    N, H, W, C3 = qkv.shape
    hid = heads * 32
    if (qkv.dtype != torch.bfloat16 or C3 != 3 * hid or not qkv.is_contiguous() or hid % 128 or Cc % 64 or (H * W) % 64 or (N * H * W) % 128
            or _linattn_ws(N, H * W, heads)):
        return None
    lib = load_library()
    weff = torch.empty((N, Cc * hid), device=qkv.device, dtype=torch.bfloat16)
    e0 = _probe_open()
    check(lib.mi_linattn_fold_fwd(N, H * W, heads, _p(qkv), _p(w_out16), Cc, _p(weff), _stream()), "mi_linattn_fold_fwd")
    if e0 is not None:
        _probe_close(e0, "linattn_fwd_kernel<io1, fold>", 2.0 * N * heads * 32 * 32 * (H * W + Cc), f"N{N} n{H * W} C{Cc}",
                     N * H * W * 2 * hid * 2 + N * Cc * hid * 2)
    y = new_act(N, H, W, Cc, qkv, torch.float32)
    y16 = new_act(N, H, W, Cc, qkv, torch.bfloat16) if want16 else None
    d = MiConvDesc(N=N, IH=H, IW=W, OH=H, OW=W, K=hid, Nc=Cc, KH=1, KW=1, stride=1, pad=0, transposed=0, w_kn=0, mode=MODE_BF16, K1=hid,
                   ldx=C3, ldx2=0, ldy=ld_of(y), ldr=ld_of(residual) if residual is not None else 0, accumulate=0)
    e0 = _probe_open()
    check(lib.mi_conv1x1_pw_batched(C.byref(d), _p(qkv), _p(weff), Cc * hid, _p(bias), _p(residual), _p(y), 0, _p(y16),
                                    ld_of(y16) if y16 is not None else 0, _stream()), "mi_conv1x1_pw_batched")
    if e0 is not None:
        px = 64 if ((N * H * W // 128) * ((Cc + 127) // 128) < 256 or (H * W) % 128) else 128
        _probe_close(e0, f"conv1x1_pw_kernel<false, {'true' if want16 else 'false'}, {px}>", 2.0 * N * H * W * Cc * hid,
                     f"N{N} {H}x{W} K{hid}->{Cc} per-sample weights", N * H * W * (hid * 2 + Cc * (4 + _esz(residual) + (2 if want16 else 0))) + N * Cc * hid * 2)
    return (y, y16) if want16 else y


def linattn_bwd(qkv, ctx, kstat, dout, heads=4):
    N, H, W, C3 = qkv.shape
    assert dout.is_contiguous() and dout.dtype == qkv.dtype
    dqkv = torch.empty_like(qkv)
    e0 = _probe_open()
    lib = load_library()
    need = _linattn_ws(N, H * W, heads)
    ws = _workspace(qkv.device, need) if need else None
    check(lib.mi_linattn_bwd_ws(N, H * W, heads, _p(qkv), _p(ctx), _p(kstat), _p(dout), _p(dqkv), _b16(qkv), _p(ws),
                                ws.numel() * 4 if need else 0, _stream()), "mi_linattn_bwd")
    if e0 is not None:
        _probe_close(e0, f"linattn_bwd_kernel<io{_b16(qkv)}>", 12.0 * N * heads * 32 * 32 * H * W, f"N{N} n{H * W}",
                     N * H * W * heads * 32 * 7 * _esz(qkv))
    return dqkv


# --------------------------------------------------------------------------- element-wise
def time_embed(t, dim):
    if not t.is_cuda:
        raise RuntimeError("time_embed needs a GPU tensor")
    t = t.to(torch.int64).contiguous()
    out = torch.empty((t.shape[0], dim), device=t.device, dtype=torch.float32)
    check(load_library().mi_time_embed(t.shape[0], dim, _p(t), _p(out), _stream()), "mi_time_embed")
    return out


def mish_fwd(x):
    _need_gpu(x)
    y = torch.empty_like(x)
    check(load_library().mi_mish_fwd(x.numel(), _p(x), _p(y), _stream()), "mi_mish_fwd")
    return y


def mish_bwd(x, dy):
    dx = torch.empty_like(x)
    check(load_library().mi_mish_bwd(x.numel(), _p(x), _p(dy), _p(dx), _stream()), "mi_mish_bwd")
    return dx


def relu_fwd(x, inplace=False):
    """max(x, 0) on a dense tensor (any shape); inplace mirrors nn.ReLU(True)."""
    _need_gpu(x)
    assert x.is_contiguous() and x.dtype == torch.float32
    y = x if inplace else torch.empty_like(x)
    check(load_library().mi_relu_fwd(x.numel(), _p(x), _p(y), _stream()), "mi_relu_fwd")
    return y


def relu_bwd(y, dy, out=None, accumulate=False):
    """dx (+)= dy where the ReLU OUTPUT y is positive; out=None allocates, out=dy works in place."""
    assert y.is_contiguous() and dy.is_contiguous()
    dx = torch.empty_like(dy) if out is None else out
    check(load_library().mi_relu_bwd(y.numel(), _p(y), _p(dy), _p(dx), int(accumulate), _stream()), "mi_relu_bwd")
    return dx


# --------------------------------------------------------------------------- VAE operators (vae_ops.hip)
_BN_WS = {}


def _bn_ws(device, C):
    ws = _BN_WS.get(device)
    if ws is None or ws.numel() < 2 * C:
        ws = torch.zeros(max(2 * C, 4096), device=device, dtype=torch.float64)     # kept zero by the kernels
        _BN_WS[device] = ws
    return ws


def batchnorm_fwd(x, gamma, beta, running_mean=None, running_var=None, momentum=0.1, eps=1e-5, training=True):
    """nn.BatchNorm2d on a dense NHWC tensor -> (y, mean[C], rstd[C]); updates the running statistics in training mode."""
    _need_gpu(x); _dense(x)
    Cc = x.shape[-1]
    M = x.numel() // Cc
    y = torch.empty_like(x)
    mean = torch.empty(Cc, device=x.device, dtype=torch.float32)
    rstd = torch.empty(Cc, device=x.device, dtype=torch.float32)
    check(load_library().mi_batchnorm_fwd(M, Cc, _p(x), _p(gamma), _p(beta), _p(y), _p(mean), _p(rstd), _p(running_mean), _p(running_var),
                                          momentum, eps, int(training), _p(_bn_ws(x.device, Cc)), _stream()), "mi_batchnorm_fwd")
    return y, mean, rstd


def batchnorm_bwd(x, mean, rstd, gamma, dy, dgamma=None, dbeta=None, out=None):
    _dense(x, dy, out)
    Cc = x.shape[-1]
    M = x.numel() // Cc
    dx = torch.empty_like(x) if out is None else out
    check(load_library().mi_batchnorm_bwd(M, Cc, _p(x), _p(mean), _p(rstd), _p(gamma), _p(dy), _p(dx), _p(dgamma), _p(dbeta),
                                          _p(_bn_ws(x.device, Cc)), _stream()), "mi_batchnorm_bwd")
    return dx


def vae_latent_fwd(h, eps):
    """h [N, 2L] = [mu | log_sigma], eps [N, L] -> (z [N, L], kld scalar)."""
    _need_gpu(h)
    N, L2 = h.shape
    L = L2 // 2
    z = torch.empty((N, L), device=h.device, dtype=torch.float32)
    kld = torch.zeros((), device=h.device, dtype=torch.float32)
    eps = eps.float().contiguous()
    check(load_library().mi_vae_latent_fwd(N, L, _p(h), h.stride(0), _p(eps), _p(z), _p(kld), _stream()), "mi_vae_latent_fwd")
    return z, kld


def vae_latent_bwd(h, eps, dz, g_kld, g_dev=None):
    N, L2 = h.shape
    dh = torch.empty((N, L2), device=h.device, dtype=torch.float32)
    check(load_library().mi_vae_latent_bwd(N, L2 // 2, _p(h), h.stride(0), _p(eps.float().contiguous()), _p(dz.contiguous()), float(g_kld), _p(g_dev),
                                           _p(dh), L2, _stream()), "mi_vae_latent_bwd")
    return dh


# --------------------------------------------------------------------------- WGAN-GP operators (critic_ops.hip)
def _dense(*ts):
    for t in ts:
        assert t is None or (t.is_contiguous() and t.dtype == torch.float32), "dense fp32 tensor expected"


def sample_norm_fwd(x, gamma, beta, eps=1e-5):
    """nn.GroupNorm(1, C) on a dense NHWC tensor -> (y, stats[N,2])."""
    _need_gpu(x); _dense(x)
    N, H, W, Cc = x.shape
    y = torch.empty_like(x)
    stats = torch.empty((N, 2), device=x.device, dtype=torch.float32)
    check(load_library().mi_sample_norm_fwd(N, H * W, Cc, _p(x), _p(gamma), _p(beta), _p(y), _p(stats), eps, _stream()), "mi_sample_norm_fwd")
    return y, stats


def sample_norm_bwd(x, stats, gamma, dy, dgamma=None, dbeta=None, extra=None, out=None):
    _dense(x, dy, extra, out)
    N, H, W, Cc = x.shape
    dx = torch.empty_like(x) if out is None else out
    check(load_library().mi_sample_norm_bwd(N, H * W, Cc, _p(x), _p(stats), _p(gamma), _p(dy), _p(extra), _p(dx), _p(dgamma), _p(dbeta),
                                            _stream()), "mi_sample_norm_bwd")
    return dx


def sample_norm_bwd2(x, stats, gamma, dy, u, dgamma=None):
    """Backward of sample_norm_bwd: (adjoint wrt dy, adjoint wrt x); dgamma accumulated."""
    _dense(x, dy, u)
    N, H, W, Cc = x.shape
    adj_dy, adj_x = torch.empty_like(x), torch.empty_like(x)
    check(load_library().mi_sample_norm_bwd2(N, H * W, Cc, _p(x), _p(stats), _p(gamma), _p(dy), _p(u), _p(adj_dy), _p(adj_x), _p(dgamma),
                                             _stream()), "mi_sample_norm_bwd2")
    return adj_dy, adj_x


def leaky_relu_fwd(x, slope=0.2, inplace=False):
    _need_gpu(x); _dense(x)
    y = x if inplace else torch.empty_like(x)
    check(load_library().mi_leaky_relu_fwd(x.numel(), _p(x), _p(y), slope, _stream()), "mi_leaky_relu_fwd")
    return y


def leaky_relu_bwd(y, dy, slope=0.2, out=None):
    _dense(y, dy, out)
    dx = torch.empty_like(dy) if out is None else out
    check(load_library().mi_leaky_relu_bwd(y.numel(), _p(y), _p(dy), _p(dx), slope, _stream()), "mi_leaky_relu_bwd")
    return dx


def tanh_fwd(x, inplace=False):
    _need_gpu(x); _dense(x)
    y = x if inplace else torch.empty_like(x)
    check(load_library().mi_tanh_fwd(x.numel(), _p(x), _p(y), _stream()), "mi_tanh_fwd")
    return y


def tanh_bwd(y, dy, out=None):
    _dense(y, dy, out)
    dx = torch.empty_like(dy) if out is None else out
    check(load_library().mi_tanh_bwd(y.numel(), _p(y), _p(dy), _p(dx), _stream()), "mi_tanh_bwd")
    return dx


def lerp_rows(a, b, e):
    """e[s] * a[s] + (1 - e[s]) * b[s] over the leading dimension."""
    _need_gpu(a); _dense(a, b)
    out = torch.empty_like(a)
    check(load_library().mi_lerp_rows(a.shape[0], a[0].numel(), _p(a), _p(b), _p(e), _p(out), _stream()), "mi_lerp_rows")
    return out


def gp_penalty(g, want_grad=True, scale=1.0, scale_dev=None):
    """(mean_s (||g_s|| - 1)^2, scale * d penalty / d g) for a dense [N, ...] gradient tensor."""
    _need_gpu(g); _dense(g)
    pen = torch.zeros((), device=g.device, dtype=torch.float32)
    u = torch.empty_like(g) if want_grad else None
    check(load_library().mi_gp_penalty(g.shape[0], g[0].numel(), _p(g), _p(pen), _p(u), scale, _p(scale_dev), _stream()), "mi_gp_penalty")
    return pen, u


def nchw_to_nhwc(x, ld=None):
    _need_gpu(x)
    B, Cc, H, W = x.shape
    ld = ld or (Cc + 3) // 4 * 4
    x = x.contiguous()
    y = torch.empty((B, H, W, ld), device=x.device, dtype=torch.float32)
    check(load_library().mi_nchw_to_nhwc(B, Cc, H * W, _p(x), _p(y), ld, _stream()), "mi_nchw_to_nhwc")
    return y[..., :Cc]


def nhwc_to_nchw(x):
    B, H, W, Cc = x.shape
    y = torch.empty((B, Cc, H, W), device=x.device, dtype=torch.float32)
    check(load_library().mi_nhwc_to_nchw(B, Cc, H * W, _p(x), ld_of(x), _p(y), _stream()), "mi_nhwc_to_nchw")
    return y


def q_sample(x0, noise, t, sqrt_ac, sqrt_1mac, want_nchw=False):
    _need_gpu(x0)
    B, Cc, H, W = x0.shape
    ld = (Cc + 3) // 4 * 4
    x0 = x0.contiguous(); noise = noise.contiguous()
    xt = torch.empty((B, H, W, ld), device=x0.device, dtype=torch.float32)
    xt_nchw = torch.empty_like(x0) if want_nchw else None
    check(load_library().mi_q_sample(B, Cc, H * W, _p(x0), _p(noise), _p(t), _p(sqrt_ac), _p(sqrt_1mac), _p(xt), ld,
                                     _p(xt_nchw), _stream()), "mi_q_sample")
    return xt[..., :Cc], xt_nchw


def eps_loss(pred_nhwc, target_nchw, loss_type=0, want_grad=True, gscale=1.0):
    B, H, W, Cc = pred_nhwc.shape
    target_nchw = target_nchw.contiguous()
    loss = torch.zeros((), device=pred_nhwc.device, dtype=torch.float32)
    ld = ld_of(pred_nhwc)
    dpred = torch.empty((B, H, W, ld), device=pred_nhwc.device, dtype=torch.float32) if want_grad else None
    check(load_library().mi_eps_loss(B, Cc, H * W, _p(pred_nhwc), ld, _p(target_nchw), loss_type, _p(loss), _p(dpred),
                                     gscale, _stream()), "mi_eps_loss")
    return loss, (dpred[..., :Cc] if want_grad else None)


def p_sample_update(x, eps_nhwc, z, t, tab, clip=True, want_nhwc=True, out=None, out_nhwc=None):
    """x_{t-1} from (x_t, eps prediction, z).  out: where the NCHW result goes (may be x itself: in place); out_nhwc: a [B, H, W, ldo]
    buffer (padding channels zero) that receives the NHWC copy -- the graph sampler's static buffers."""
    B, Cc, H, W = x.shape
    ldo = (Cc + 3) // 4 * 4
    xp = torch.empty_like(x) if out is None else out
    if out_nhwc is not None:
        assert out_nhwc.shape == (B, H, W, ldo) and out_nhwc.is_contiguous() and out_nhwc.dtype == torch.float32
        xp_nhwc = out_nhwc
    else:
        xp_nhwc = torch.zeros((B, H, W, ldo), device=x.device, dtype=torch.float32) if want_nhwc else None
    check(load_library().mi_p_sample_update(
        B, Cc, H * W, _p(x), _p(eps_nhwc), ld_of(eps_nhwc), _p(z), _p(t), _p(tab["sqrt_recip_alphas_cumprod"]),
        _p(tab["sqrt_recipm1_alphas_cumprod"]), _p(tab["posterior_mean_coef1"]), _p(tab["posterior_mean_coef2"]),
        _p(tab["posterior_log_variance_clipped"]), int(clip), _p(xp), _p(xp_nhwc), ldo, _stream()), "mi_p_sample_update")
    return xp, (xp_nhwc[..., :Cc] if want_nhwc else None)


def adam_step(p, g, m, v, lr, b1, b2, eps, step, gscale=1.0):
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    e0 = _probe_open()
    check(load_library().mi_adam_step(p.numel(), _p(p), _p(g), _p(m), _p(v), lr, b1, b2, eps, bc1, bc2, gscale, _stream()),
          "mi_adam_step")
    if e0 is not None:
        _probe_close(e0, "adam_kernel", 0.0, f"{p.numel()} params", p.numel() * 28.0)


def adam_tick(state):
    check(load_library().mi_adam_tick(_p(state), _stream()), "mi_adam_tick")


def adam_step_dev(p, g, m, v, state, b1, b2, eps, gscale=1.0):
    """Adam with step count / learning rate in `state` (device float[2]): graph-capturable."""
    check(load_library().mi_adam_step_dev(p.numel(), _p(p), _p(g), _p(m), _p(v), _p(state), b1, b2, eps, gscale, _stream()), "mi_adam_step_dev")


def axpby(a, x, y, accumulate):
    """y = a*x (+ y).  x, y: same logical shape; strided channel slices allowed for 4-D."""
    if x.is_contiguous() and y.is_contiguous():
        check(load_library().mi_axpby(x.numel(), a, _p(x), int(accumulate), _p(y), _stream()), "mi_axpby")
    else:
        Cc = x.shape[-1]
        M = x.shape[0] * x.shape[1] * x.shape[2] if x.dim() == 4 else x.shape[0]
        check(load_library().mi_axpby2d(M, Cc, a, _p(x), ld_of(x), int(accumulate), _p(y), ld_of(y), _stream()), "mi_axpby2d")


def scale_by_device_scalar(x, s):
    """x *= s where s is a 0-d/1-element fp32 tensor on the device."""
    s = s.reshape(1).float()
    check(load_library().mi_scale_by_device_scalar(_rows(x), x.shape[-1], _p(x), ld_of(x), _p(s), _stream()),
          "mi_scale_by_device_scalar")


def gather_rows(table, idx):
    """table[T, C] fp32, idx int64 [B] on the device -> [B, C]."""
    out = torch.empty((idx.shape[0], table.shape[1]), device=table.device, dtype=torch.float32)
    check(load_library().mi_gather_rows(idx.shape[0], table.shape[1], _p(table), _p(idx), _p(out), _stream()), "mi_gather_rows")
    return out


def clock_probe(device, usec=300, blocks=256):
    """Shader clock (MHz) workgroup 0 sustained while `blocks` workgroups issue back-to-back bf16 MFMAs for `usec` microseconds
    (mi_debug_clock_probe: s_memtime ticks over 100 MHz wall-clock ticks).  A measurement aid for bench.py; synchronises."""
    out = torch.zeros(3, device=device, dtype=torch.int64)
    check(load_library().mi_debug_clock_probe(blocks, usec, _p(out), _stream()), "mi_debug_clock_probe")
    torch.cuda.synchronize()
    s, w = int(out[0]), int(out[1])
    return round(100.0 * s / max(w, 1), 1)


def u8_gather_normalize(data_u8, idx, flip=None, normalize=True):
    """data_u8: uint8 [N,H,W,C] on the device, idx: int64 [B], flip: uint8/bool [B] or None -> fp32 NCHW batch [B,C,H,W] with the
    reference transform chain applied (ToTensor, flip, Normalize(0.5, 0.5))."""
    if not data_u8.is_cuda or data_u8.dtype != torch.uint8 or not data_u8.is_contiguous():
        raise RuntimeError("u8_gather_normalize needs a contiguous uint8 dataset tensor on the HIP device")
    if data_u8.device.index != torch.cuda.current_device():
        raise RuntimeError("dataset tensor is not on the current HIP device")
    N, H, W, Cc = data_u8.shape
    idx = idx.to(device=data_u8.device, dtype=torch.int64).contiguous()
    if flip is not None:
        flip = flip.to(device=data_u8.device, dtype=torch.uint8).contiguous()
    out = torch.empty((idx.shape[0], Cc, H, W), device=data_u8.device, dtype=torch.float32)
    check(load_library().mi_u8_gather_normalize(idx.shape[0], Cc, H, W, _p(data_u8), _p(idx), _p(flip), int(normalize), _p(out), _stream()),
          "mi_u8_gather_normalize")
    return out


@functools.lru_cache(maxsize=None)      # pure function of the shape: one FFI query per distinct layer, not per step
def fast3x3_supported(N, H, W, K, Nc, K1=None):
    """(conv via the LDS-tile kernel?, wgrad via the image-major kernel?) for a 3x3/s1/p1 layer in bf16 mode."""
    lib = load_library()
    dc = MiConvDesc(N=N, IH=H, IW=W, OH=H, OW=W, K=K, Nc=Nc, KH=3, KW=3, stride=1, pad=1, transposed=0, w_kn=0,
                    mode=MODE_BF16, K1=K1 or K, ldx=4, ldx2=4, ldy=Nc, ldr=0, accumulate=0)
    dw = MiWgradDesc(N=N, GH=H, GW=W, DH=H, DW=W, Ci=K, Cj=Nc, KH=3, KW=3, stride=1, pad=1, gather_i=1, mode=MODE_BF16,
                     I1=K1 or K, ldp=4, ldp2=4, ldq=4)
    return bool(lib.mi_conv3x3_bf16w_supported(C.byref(dc))), bool(lib.mi_conv3x3_wgrad_supported(C.byref(dw)))


@functools.lru_cache(maxsize=None)      # pure function of the shape: one FFI query per distinct layer, not per step
def fast1x1_supported(N, H, W, K, Nc):
    """(conv + dgrad via the LDS-tile kernel?, wgrad via the image-major kernel?) for a 1x1 layer in bf16 mode."""
    lib = load_library()
    ok = True
    for k, n in ((K, Nc), (Nc, K)):             # forward and data gradient
        dc = MiConvDesc(N=N, IH=H, IW=W, OH=H, OW=W, K=k, Nc=n, KH=1, KW=1, stride=1, pad=0, transposed=0, w_kn=0,
                        mode=MODE_BF16, K1=k, ldx=8, ldx2=8, ldy=n, ldr=0, accumulate=0)
        ok = ok and bool(lib.mi_conv3x3_bf16w_supported(C.byref(dc)))
    dw = MiWgradDesc(N=N, GH=H, GW=W, DH=H, DW=W, Ci=K, Cj=Nc, KH=1, KW=1, stride=1, pad=0, gather_i=1, mode=MODE_BF16,
                     I1=K, ldp=8, ldp2=8, ldq=8)
    return ok, bool(lib.mi_conv3x3_wgrad_supported(C.byref(dw)))


@functools.lru_cache(maxsize=None)      # pure function of the shape: one FFI query per distinct layer, not per step
def conv3x3_uses_splitk(N, H, W, K, Nc, K1=None):
    """Would the 3x3 LDS-tile kernel run this layer with the split-K plan (fp32 output only)?"""
    dc = MiConvDesc(N=N, IH=H, IW=W, OH=H, OW=W, K=K, Nc=Nc, KH=3, KW=3, stride=1, pad=1, transposed=0, w_kn=0,
                    mode=MODE_BF16, K1=K1 or K, ldx=4, ldx2=4, ldy=Nc, ldr=0, accumulate=0)
    return bool(load_library().mi_conv3x3_bf16w_uses_splitk(C.byref(dc)))
