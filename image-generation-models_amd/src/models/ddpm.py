"""DDPM on MI355X: `Unet`, `GaussianDiffusion`, `DDPM` with the reference's constructor
signatures, method names and state_dict keys (reference: src/models/ddpm.py:169-521), but
executed by the hand-written gfx950 kernels of libmi_ddpm.so (include/mi_ddpm.h).

Design (DESIGN.md has the long form):
  * every UNet parameter is a VIEW into ONE flat fp32 buffer (`Unet.flat_params`); conv weights
    are stored tap-major [kh][kw][Cin][Cout] and exposed as permuted views with PyTorch's logical
    shapes, so state_dict()/load_state_dict() round-trip reference checkpoints while forward,
    dgrad, wgrad, the optimizer and the RCCL all-reduce stream contiguous memory.
  * activations are NHWC fp32; NCHW only at the `Unet.forward` boundary.
  * `Unet.forward` is one autograd node: the forward pass records a tape of kernel-level
    backward steps; `backward` replays it, writing into the flat gradient buffer
    (`Unet.flat_grads`, which every `param.grad` is a view of).
  * there is no CPU / ATen fallback: tensors that are not on a HIP device raise.
"""
from __future__ import annotations

import math
import os
from functools import partial
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import nn

from ..ops import functional as K
from .base import BaseModel, ValidationResult

_HEADS, _DHEAD = 4, 32            # LinearAttention(dim, heads=4, dim_head=32)   ddpm.py:147
_GN_GROUPS = 8                    # Block always builds GroupNorm(8, .)           ddpm.py:113,132-133


def _mode_id(name: str) -> int:
    if name not in ("fp32", "bf16"):
        raise ValueError(f"compute mode must be 'fp32' or 'bf16', got {name!r}")
    return K.MODE_BF16 if name == "bf16" else K.MODE_FP32


# --------------------------------------------------------------------------------------------
# network description
# --------------------------------------------------------------------------------------------
class _Entry:
    """One parameter: logical (PyTorch) shape, storage layout, init rule."""
    __slots__ = ("key", "shape", "layout", "init", "fan_in", "rank", "offset", "numel")

    def __init__(self, key, shape, layout, init, fan_in, rank):
        self.key, self.shape, self.layout, self.init, self.fan_in, self.rank = key, tuple(shape), layout, init, fan_in, rank
        self.numel = int(np.prod(shape))
        self.offset = -1

    def storage_view(self, flat: torch.Tensor) -> torch.Tensor:
        seg = flat[self.offset:self.offset + self.numel]
        if self.layout == "conv":            # logical [Cout,Cin,kh,kw], stored [kh,kw,Cin,Cout]
            co, ci, kh, kw = self.shape
            return seg.view(kh, kw, ci, co)
        if self.layout == "convT":           # logical [Cin,Cout,kh,kw], stored [kh,kw,Cin,Cout]
            ci, co, kh, kw = self.shape
            return seg.view(kh, kw, ci, co)
        return seg.view(self.shape)

    def logical_view(self, flat: torch.Tensor) -> torch.Tensor:
        sv = self.storage_view(flat)
        if self.layout == "conv":
            return sv.permute(3, 2, 0, 1)
        if self.layout == "convT":
            return sv.permute(2, 3, 0, 1)
        return sv


class _Arch:
    """Static layer list of the reference Unet (ddpm.py:182-236) + parameter table."""

    def __init__(self, dim, dim_mults, channels, out_dim, with_time_emb=True):
        self.dim, self.channels, self.out_dim = dim, channels, out_dim or channels
        self.with_time = bool(with_time_emb)               # False: no time MLP, ResnetBlocks without their time Linear (ddpm.py:186-198,126-130)
        self.dims = [channels] + [dim * m for m in dim_mults]
        pairs = list(zip(self.dims[:-1], self.dims[1:]))
        self.entries: List[_Entry] = []
        self.res_blocks: List[dict] = []
        rank = {"time_mlp": 0, "downs": 1, "ups": 2, "mid_block1": 3, "mid_attn": 4, "mid_block2": 5, "final_conv": 6}

        def add(key, shape, layout="plain", init="zero", fan_in=0):
            self.entries.append(_Entry(key, shape, layout, init, fan_in, rank[key.split(".")[0]]))

        def linear(pre, i, o):
            add(pre + "weight", (o, i), "plain", "kaiming", i); add(pre + "bias", (o,), "plain", "ubias", i)

        def conv(pre, i, o, k, bias=True, transposed=False):
            shape = (i, o, k, k) if transposed else (o, i, k, k)
            fan = shape[1] * k * k
            add(pre + "weight", shape, "convT" if transposed else "conv", "kaiming", fan)
            if bias:
                add(pre + "bias", (o,), "plain", "ubias", fan)

        def norm_affine(wkey, bkey, shape):
            add(wkey, shape, "plain", "one"); add(bkey, shape, "plain", "zero")

        def resblock(pre, i, o):
            if self.with_time:
                linear(pre + "mlp.1.", dim, o)
            for blk, ci in (("block1.", i), ("block2.", o)):
                conv(pre + blk + "block.0.", ci, o, 3)
                norm_affine(pre + blk + "block.1.weight", pre + blk + "block.1.bias", (o,))
            if i != o:
                conv(pre + "res_conv.", i, o, 1)
            blk = {"pre": pre, "cin": i, "cout": o, "res": i != o}
            self.res_blocks.append(blk)
            return blk

        def attention(pre, c):
            conv(pre + "fn.fn.to_qkv.", c, 3 * _HEADS * _DHEAD, 1, bias=False)
            conv(pre + "fn.fn.to_out.", _HEADS * _DHEAD, c, 1)
            norm_affine(pre + "fn.norm.g", pre + "fn.norm.b", (1, c, 1, 1))
            return {"pre": pre, "c": c}

        if self.with_time:
            linear("time_mlp.1.", dim, dim * 4)
            linear("time_mlp.3.", dim * 4, dim)
        self.downs, self.ups = [], []
        for L, (i, o) in enumerate(pairs):
            lvl = {"res1": resblock(f"downs.{L}.0.", i, o), "res2": resblock(f"downs.{L}.1.", o, o),
                   "attn": attention(f"downs.{L}.2.", o), "down": None}
            if L < len(pairs) - 1:
                conv(f"downs.{L}.3.conv.", o, o, 3)
                lvl["down"] = {"pre": f"downs.{L}.3.conv.", "c": o}
            self.downs.append(lvl)
        mid = self.dims[-1]
        self.mid1 = resblock("mid_block1.", mid, mid)
        self.mid_attn = attention("mid_attn.", mid)
        self.mid2 = resblock("mid_block2.", mid, mid)
        for L, (i, o) in enumerate(reversed(pairs[1:])):
            lvl = {"res1": resblock(f"ups.{L}.0.", o * 2, i), "res2": resblock(f"ups.{L}.1.", i, i),
                   "attn": attention(f"ups.{L}.2.", i)}
            conv(f"ups.{L}.3.conv.", i, i, 4, transposed=True)      # `is_last` never fires (ddpm.py:222)
            lvl["up"] = {"pre": f"ups.{L}.3.conv.", "c": i}
            self.ups.append(lvl)
        c1 = self.dims[1]
        conv("final_conv.0.block.0.", c1, c1, 3)
        norm_affine("final_conv.0.block.1.weight", "final_conv.0.block.1.bias", (c1,))
        conv("final_conv.1.", c1, self.out_dim, 1)

        # flat storage order: all time-bias Linear weights adjacent (one GEMM feeds every
        # ResnetBlock), then their biases, then everything else; 64-float alignment per entry.
        mlp_w = [e for e in self.entries if e.key.endswith(".mlp.1.weight")]
        mlp_b = [e for e in self.entries if e.key.endswith(".mlp.1.bias")]
        rest = [e for e in self.entries if e not in mlp_w and e not in mlp_b]
        off = 0
        for e in mlp_w:
            e.offset = off; off += e.numel
        off = (off + 63) // 64 * 64
        self.mlp_w_off, self.mlp_rows = (mlp_w[0].offset if mlp_w else 0), sum(e.shape[0] for e in mlp_w)
        self.mlp_b_off = off
        for e in mlp_b:
            e.offset = off; off += e.numel
        for e in rest:
            off = (off + 63) // 64 * 64
            e.offset = off; off += e.numel
        self.flat_numel = (off + 63) // 64 * 64
        # flat-buffer range owned by each backward record (for bucketed gradient all-reduce)
        def span(prefixes):
            es = [e for e in rest if any(e.key.startswith(p) for p in prefixes)]
            return (min(e.offset for e in es), max(e.offset + e.numel for e in es))
        for rec in self.res_blocks:
            rec["range"] = span([rec["pre"] + "block", rec["pre"] + "res_conv"])
        for lvl in self.downs + self.ups:
            lvl["attn"]["range"] = span([lvl["attn"]["pre"]])
            for kind in ("down", "up"):
                if lvl.get(kind):
                    lvl[kind]["range"] = span([lvl[kind]["pre"]])
        self.mid_attn["range"] = span([self.mid_attn["pre"]])
        self.final_range = span(["final_conv."])
        self.time_range = (0, span(["time_mlp."])[1]) if self.with_time else (0, 0)
        col = 0
        for blk in self.res_blocks:
            blk["tcol"] = 0
        for blk, e in zip(self.res_blocks, mlp_w):
            assert e.key == blk["pre"] + "mlp.1.weight"
            blk["tcol"] = col; col += blk["cout"]


class _Node(nn.Module):
    """Anonymous container; the module tree exists only to reproduce the reference's keys."""


# --------------------------------------------------------------------------------------------
# gradient bookkeeping for the tape
# --------------------------------------------------------------------------------------------
class _GradMap:
    def __init__(self):
        self._g: Dict[int, torch.Tensor] = {}
        self._keep: List[torch.Tensor] = []

    def take(self, t: torch.Tensor) -> torch.Tensor:
        return self._g.pop(id(t))

    def add(self, t: torch.Tensor, g: torch.Tensor):
        """grad[t] += g, by alias when it is the first contribution."""
        cur = self._g.get(id(t))
        if cur is None:
            self._g[id(t)] = g; self._keep.append(t)
        else:
            K.axpby(1.0, g, cur, True)

    def target(self, t: torch.Tensor) -> Tuple[torch.Tensor, bool]:
        """(buffer, accumulate) for a kernel that writes a contribution to grad[t]."""
        cur = self._g.get(id(t))
        if cur is not None:
            return cur, True
        buf = torch.empty(t.shape, device=t.device, dtype=t.dtype)     # bf16 for block-internal tensors, else fp32
        self._g[id(t)] = buf; self._keep.append(t)
        return buf, False


class _UnetFunction(torch.autograd.Function):
    """One autograd node for the whole UNet; parameter gradients are written straight into
    `Unet.flat_grads` (what every param.grad views), so nothing is returned for them."""

    @staticmethod
    def forward(ctx, x, time, anchor, net):
        y, tape = net._execute(x, time, record=True)
        ctx.net, ctx.tape = net, tape
        ctx.need_dx = x.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        net, tape = ctx.net, ctx.tape
        ctx.tape = None
        dx = net._backward(tape, dy.contiguous(), need_dx=ctx.need_dx)
        return dx, None, None, None


class Unet(nn.Module):
    """Drop-in for the reference `Unet` (ddpm.py:169-261): same ctor, `forward(x, time)`,
    parameters()/state_dict() keys and shapes.  `groups` is accepted and ignored exactly like
    the reference (its blocks always use 8 groups)."""

    def __init__(self, dim, out_dim=None, dim_mults=(1, 2, 4, 8), groups=8, channels=3, with_time_emb=True):
        super().__init__()
        if dim % 8 or dim < 8:
            raise ValueError("dim must be a multiple of 8")
        self.channels = channels
        self.dim = dim
        # with_time_emb=False (round 5; the reference's flag that DDPM never sets): no time_mlp, no per-block mlp -- the state_dict has
        # neither, forward ignores `time` as the reference does (t = None, ddpm.py:241)
        self.with_time_emb = bool(with_time_emb)
        arch = _Arch(dim, tuple(dim_mults), channels, out_dim, with_time_emb)
        object.__setattr__(self, "_arch", arch)
        self.compute_mode = os.environ.get("MI_DDPM_MODE", "fp32")
        # bf16 mode only: storage of the ResnetBlock-internal tensors (conv output -> GroupNorm -> conv input, and
        # their gradients).  "bf16" (default): wherever every kernel touching them has a bf16 path; "auto": same,
        # except layers whose conv would run the split-K plan (fp32 atomics) stay fp32; "fp32": never.  The residual
        # stream, attention path and every parameter/statistic stay fp32.  Measured on cfg 2 (B=128): bf16 12.8k,
        # auto 12.6k, fp32 12.5k images/s.
        self.block_storage = os.environ.get("MI_DDPM_STORAGE", "bf16")
        # inference: GroupNorm-apply + Mish (+ time bias) folded into the staging of block2's 3x3 conv (BASELINE.json's named fused
        # kernel; h1 is never materialised).  "auto" (default): wherever the private-weight-stream kernel takes the layer
        # (mi_conv3x3_pw_gn_mish_sums: the transform runs once per staged element between the MFMAs, statistics from conv1's
        # epilogue -- two launches per Block -> Block; sampler at B = 64: 722 vs 722 denoise steps/s with the 8x8 level included through
        # round 2's halo kernel, which loses there and is therefore not taken).  "0": never; "1": everywhere a fused kernel exists,
        # statistics by a pass over c1 (mi_gn_stats_coef); "2": everywhere, statistics from conv1's epilogue.
        self.fuse_gn_conv = os.environ.get("MI_DDPM_FUSE_GN", "auto")
        # Downsample / Upsample weight gradients through the LDS-DMA kernel (csrc/wgrad_s2_tr.hip); 0 = round 1's ring kernel
        self.s2_wgrad_tr = K.debug_knob("MI_DDPM_S2_TR", "1") != "0"
        # final_conv.0's conv output stored like the other Blocks' (bf16 in bf16 mode); 0 = fp32 as in round 1
        self.final_block16 = K.debug_knob("MI_DDPM_FINAL16", "1") != "0"
        # round 4: the two tensors at the 3-channel ends stored like every other block-internal tensor of bf16 mode -- the first Block's conv
        # output (and its gradient) and final_conv.0's output (and its gradient); 0 = fp32 as before
        self.ends16 = K.debug_knob("MI_DDPM_ENDS16", "1") != "0"
        # LinearAttention's to_out conv writes the bf16 copy of its (residual-stream) output along; 0 = separate conversion launches
        self.dual_out = K.debug_knob("MI_DDPM_DUAL_OUT", "1") != "0"
        # PreNorm's LayerNorm applied while to_qkv's input is staged (mi_ln_conv1x1_pw; training: _dual -- the same launch writes the
        # normalised tensor for to_qkv's weight gradient)
        self.fuse_ln_qkv = K.debug_knob("MI_DDPM_FUSE_LN", "1") != "0"
        self.fuse_ln_qkv_train = K.debug_knob("MI_DDPM_FUSE_LN_TRAIN", "1") != "0"
        self.small_cin_dual = K.debug_knob("MI_DDPM_CIN_DUAL", "1") != "0"      # the first block's 3x3 conv + res_conv in one launch
        self.cin_dual_chores = K.debug_knob("MI_DDPM_CIN_CHORES", "1") != "0"       # ... which also does the forward's chores (sums pool, time-bias rows)
        # inference: LinearAttention folded into to_out's (per-sample) weights -- built, tested, measured 1 % SLOWER in the replayed sampler step
        # (1 057-1 071 against 1 075-1 080 steps/s: per-sample weights are not shared between workgroups): off
        self.fold_attn = K.debug_knob("MI_DDPM_FOLD_ATTN", "0") == "1"
        self.small_cout_bwd1 = K.debug_knob("MI_DDPM_COUT_BWD1", "1") != "0"    # the C -> 3 conv's weight and data gradient in one launch
        self.fuse_final = K.debug_knob("MI_DDPM_FUSE_FINAL", "1") != "0"        # inference: final_conv.0's GroupNorm + Mish inside final_conv.1's load
        self.accumulate_grads = False
        self.grad_ready_hook = None        # callable(lo, hi): flat_grads[lo:hi) is final (set by the DDP reducer)

        flat = torch.zeros(arch.flat_numel)
        # default torch init drawn in the reference's construction order => same seeded weights
        for e in arch.entries:
            if e.init == "kaiming":
                w = torch.empty(e.shape)
                nn.init.kaiming_uniform_(w, a=math.sqrt(5))
                e.logical_view(flat).copy_(w)
            elif e.init == "ubias":
                bound = 1 / math.sqrt(e.fan_in) if e.fan_in > 0 else 0
                e.logical_view(flat).copy_(torch.empty(e.shape).uniform_(-bound, bound))
            elif e.init == "one":
                e.logical_view(flat).fill_(1.0)
        object.__setattr__(self, "_flat", flat)
        object.__setattr__(self, "_gflat", None)
        object.__setattr__(self, "_dirty", 0)
        object.__setattr__(self, "_shadow", None)
        object.__setattr__(self, "_shadow_key", None)
        object.__setattr__(self, "_shadow32", None)
        object.__setattr__(self, "_shadow32_key", None)
        object.__setattr__(self, "_anchor", torch.zeros(1, requires_grad=True))
        # module tree in the reference's registration order (time_mlp, downs, ups, mid_*, final_conv)
        self._plist: List[Tuple[_Entry, nn.Parameter]] = []
        for e in sorted(arch.entries, key=lambda q: q.rank):   # stable: keeps construction order inside a group
            parts = e.key.split(".")
            node = self
            for name in parts[:-1]:
                if name not in node._modules:
                    node.add_module(name, _Node())
                node = node._modules[name]
            p = nn.Parameter(e.logical_view(flat))
            node.register_parameter(parts[-1], p)
            self._plist.append((e, p))
        self._bind(flat)

    # ------------------------------------------------------------------ storage management
    def _bind(self, flat: torch.Tensor):
        object.__setattr__(self, "_flat", flat)
        sv = {}
        for e, p in self._plist:
            p.data = e.logical_view(flat)
            sv[e.key] = e.storage_view(flat)
        object.__setattr__(self, "_sv", sv)
        object.__setattr__(self, "_offs", {e.key: e.offset for e, _ in self._plist})
        object.__setattr__(self, "_shadow_key", None)
        object.__setattr__(self, "_shadow32_key", None)
        g = self._gflat
        if g is not None and (g.device != flat.device or g.dtype != flat.dtype):
            object.__setattr__(self, "_gflat", None)
            for _, p in self._plist:
                p.grad = None

    def _apply(self, fn, recurse=True):
        new = fn(self._flat)
        if new.dtype != torch.float32:
            raise RuntimeError("the MI355X UNet keeps fp32 master weights; use compute_mode='bf16' for bf16 math")
        if new is not self._flat:
            self._bind(new)
        return self

    @property
    def flat_params(self) -> torch.Tensor:
        return self._flat

    def mark_params_dirty(self):
        """Tell the UNet its flat buffer was written behind torch's back (fused Adam, RCCL broadcast)."""
        object.__setattr__(self, "_dirty", self._dirty + 1)

    def _shadows(self):
        """bf16 copies of every conv weight for the bf16-MFMA kernels: wd = master layout
        [tap][Cin][Cout] (dgrad operand), wf = [tap][Cout][Cin] (forward operand).  Rebuilt by one
        kernel launch whenever the fp32 master buffer changed.  wdq / wfq: the same two operands of the 3x3 layers in
        MFMA-fragment order (mi_conv3x3_pw streams them 1 KB per instruction)."""
        flat = self._flat
        key = (flat.data_ptr(), flat._version, self._dirty)
        if self._shadow_key != key:
            if self._shadow is None or self._shadow[0].device != flat.device:
                ents = []
                for e in self._arch.entries:
                    if e.layout in ("conv", "convT"):
                        kh, kw, ci, co = e.storage_view(flat).shape
                        ents.append((e.offset, kh * kw, ci, co))
                table, nent, tiles = K.pack_table(ents, flat.device)
                bufs = [torch.zeros(flat.numel(), device=flat.device, dtype=torch.bfloat16) for _ in range(4)]
                object.__setattr__(self, "_shadow", (*bufs, table, nent, tiles))
            wd, wf, wdq, wfq, table, nent, tiles = self._shadow
            K.pack_weights_bf16(table, nent, tiles, flat, wd, wf, wdq, wfq)
            object.__setattr__(self, "_shadow_key", key)
        return self._shadow[:4]

    def _shadows32(self):
        """fp32 mode: fragment-order fp32 copies (data-gradient, forward operand) of the 3x3 / 1x1 conv weights for
        mi_conv3x3_pw_f32, rebuilt by one launch whenever the master buffer changed (the table is the bf16 pack's)."""
        flat = self._flat
        key = (flat.data_ptr(), flat._version, self._dirty)
        if getattr(self, "_shadow32_key", None) != key:
            if getattr(self, "_shadow32", None) is None or self._shadow32[0].device != flat.device:
                ents = []
                for e in self._arch.entries:
                    if e.layout in ("conv", "convT"):
                        kh, kw, ci, co = e.storage_view(flat).shape
                        ents.append((e.offset, kh * kw, ci, co))
                table, nent, tiles = K.pack_table(ents, flat.device)
                bufs = [torch.zeros(flat.numel(), device=flat.device, dtype=torch.float32) for _ in range(2)]
                object.__setattr__(self, "_shadow32", (*bufs, table, nent, tiles))
            wdq32, wfq32, table, nent, tiles = self._shadow32
            K.pack_weights_f32frag(table, nent, tiles, flat, wdq32, wfq32)
            object.__setattr__(self, "_shadow32_key", key)
        return self._shadow32[:2]

    @property
    def flat_grads(self) -> torch.Tensor:
        if self._gflat is None:
            g = torch.zeros_like(self._flat)
            object.__setattr__(self, "_gflat", g)
            gv = {}
            for e, p in self._plist:
                p.grad = e.logical_view(g)
                gv[e.key] = e.storage_view(g)
            object.__setattr__(self, "_gv", gv)
        return self._gflat

    def zero_grad(self, set_to_none: bool = False):
        if self._gflat is not None:
            self._gflat.zero_()

    def load_state_dict(self, state_dict, strict: bool = True, assign: bool = False):
        out = super().load_state_dict(state_dict, strict=strict, assign=False)
        return out

    # ------------------------------------------------------------------ public forward
    def forward(self, x, time):
        if not x.is_cuda:
            raise RuntimeError("Unet.forward: input is not on a HIP device; this implementation has no CPU path")
        if self._flat.device != x.device:
            raise RuntimeError("Unet parameters and input are on different devices; call .to(device) first")
        if torch.is_grad_enabled() and (self.training or x.requires_grad):
            if self._anchor.device != x.device:
                object.__setattr__(self, "_anchor", torch.zeros(1, device=x.device, requires_grad=True))
            return _UnetFunction.apply(x, time, self._anchor, self)
        y, _ = self._execute(x, time, record=False)
        return y

    def _execute(self, x, time, record):
        """NCHW in -> NCHW out; returns (y, tape)."""
        x_in = K.nchw_to_nhwc(x.float())
        eps, tape = self.forward_nhwc(x_in, time, record)
        return K.nhwc_to_nchw(eps), tape

    # ------------------------------------------------------------------ kernel-level forward
    @torch.no_grad()
    def time_bias_table(self, timesteps: int) -> torch.Tensor:
        """Every ResnetBlock's time bias for t = 0..T-1, [T, sum Cout]: the time MLP (ddpm.py:186-193) and the
        per-block Linear(Mish(.)) (ddpm.py:126-130,139-140) depend on t only, so a T-step sampler computes them
        once instead of T times (inference only: the table is stale once the weights change)."""
        A, sv, flat = self._arch, self._sv, self._flat
        if not A.with_time:
            return None                                    # nothing depends on t
        T = int(timesteps)
        t = torch.arange(T, device=flat.device, dtype=torch.long)

        def lin(inp, w, b):
            o, i = w.shape
            y = K.small_gemm(False, True, inp, w, bias=b)
            if y is not None:
                return y
            return K.conv_igemm(inp.view(T, 1, 1, i), w, kh=1, kw=1, stride=1, pad=0, transposed=False, w_kn=False,
                                K=i, Nc=o, out_hw=(1, 1), mode=K.MODE_FP32, bias=b).view(T, o)
        temb = lin(K.mish_fwd(lin(K.time_embed(t, A.dim), sv["time_mlp.1.weight"], sv["time_mlp.1.bias"])),
                   sv["time_mlp.3.weight"], sv["time_mlp.3.bias"])
        w_all = flat[A.mlp_w_off:A.mlp_w_off + A.mlp_rows * A.dim].view(A.mlp_rows, A.dim)
        b_all = flat[A.mlp_b_off:A.mlp_b_off + A.mlp_rows]
        return lin(K.mish_fwd(temb), w_all, b_all).contiguous()

    def forward_nhwc(self, x, time, record=False, time_bias_table=None):
        """x: NHWC activation [B,H,W,channels] (pixel stride 4-aligned) -> NHWC eps prediction + tape.
        time_bias_table (inference only): rows of `time_bias_table(T)` replace the time MLP."""
        A, sv, mode = self._arch, self._sv, _mode_id(self.compute_mode)
        tape: Optional[list] = [] if record else None
        B, H, W, _ = x.shape
        flat = self._flat
        tb_pending: list = []

        def lin(inp, wkey, rows=None, w=None, b=None):
            w = sv[wkey + "weight"] if w is None else w
            b = sv[wkey + "bias"] if b is None else b
            o, i = w.shape
            y = K.small_gemm(False, True, inp, w, bias=b)                  # y = x W^T + b, exact fp32
            if y is not None:
                return y
            y = K.conv_igemm(inp.view(B, 1, 1, i), w, kh=1, kw=1, stride=1, pad=0, transposed=False, w_kn=False,
                             K=i, Nc=o, out_hw=(1, 1), mode=K.MODE_FP32, bias=b)
            return y.view(B, o)

        # ---- time embedding MLP (ddpm.py:186-193) and every block's time bias (ddpm.py:126-130) in one GEMM
        if not A.with_time:
            tb_all = None
        elif time_bias_table is not None and not record:
            # sampler: precomputed per timestep.  Where the image -> features block takes the dual small-Cin kernel (the forward's first launch), that
            # launch gathers the rows too (tb_pending); else its own launch
            b0 = A.downs[0]["res1"]
            if (self.cin_dual_chores and self.small_cin_dual and b0["res"] and time_bias_table.is_contiguous() and time_bias_table.shape[1] % 4 == 0
                    and K.small_cin_supported(3, b0["cin"], b0["cout"]) and K.small_cin_dual_supported(B, H, W, b0["cin"], b0["cout"], K.ld_of(x))):
                tb_all = torch.empty((B, time_bias_table.shape[1]), device=x.device, dtype=torch.float32)
                tb_pending.append((time_bias_table, time, tb_all))
            else:
                tb_all = K.gather_rows(time_bias_table, time)
        else:
            te = K.time_embed(time, A.dim)
            t1 = lin(te, "time_mlp.1.")
            a1 = K.mish_fwd(t1)
            temb = lin(a1, "time_mlp.3.")
            mt = K.mish_fwd(temb)
            w_all = flat[A.mlp_w_off:A.mlp_w_off + A.mlp_rows * A.dim].view(A.mlp_rows, A.dim)
            b_all = flat[A.mlp_b_off:A.mlp_b_off + A.mlp_rows]
            tb_all = lin(mt, None, w=w_all, b=b_all)                       # [B, sum Cout]
            if record:
                tape.append(("time", te, t1, a1, temb, mt))

        if mode == K.MODE_BF16:
            wd_sh, wf_sh, wdq_sh, wfq_sh = self._shadows()
        else:
            wdq32_sh, wfq32_sh = self._shadows32()
        offs = self._offs

        BF = torch.bfloat16
        # bf16 copies of residual-stream tensors (fp32) that feed a Block conv: the conv and its weight gradient read the copy --
        # the rounding they would apply while staging, applied once (bit-identical results, half the bytes, and both kernels
        # can take their operands by LDS-DMA).  A ResnetBlock output is copied by the GroupNorm kernel that writes it
        # (want16), everything else by mi_f32_to_bf16 on first use.
        sh: Dict[int, tuple] = {}          # id(tensor) -> (tensor, copy): holding the tensor keeps its id from being reused
        # (training only: in inference there is no weight gradient to feed and the copies cost 2.5 % of a denoise step)
        use_sh = (record and mode == K.MODE_BF16 and self.block_storage in ("bf16", "auto")
                  and K.debug_knob("MI_DDPM_SHADOW", "1") == "1")

        # inference: the GroupNorm kernel of a ResnetBlock whose output goes straight into the next Block's conv can write the bf16 copy
        # along (no conversion launch).  Off: measured 680 vs 720 denoise steps/s at B = 64 -- the dual-output GroupNorm costs
        # +4 us per launch and the convs that switch to bf16 input were the cheap ones
        eval16 = (not record and mode == K.MODE_BF16 and self.block_storage in ("bf16", "auto")
                  and K.debug_knob("MI_DDPM_EVAL16", "0") == "1")

        def s2_copy(c, k, transposed):      # Downsample / Upsample read (and their weight gradients want) a bf16 copy of their input
            return (use_sh and self.s2_wgrad_tr and K.igemm_bf16_in_supported(c, c, k, 2, transposed, mode, (8, 8)))

        def s2_eval_copy(t, k, transposed):  # inference: the copy to_out's epilogue wrote along (attention(..., feeds_s2=True)), if any
            if record or id(t) not in sh or not K.igemm_bf16_in_supported(t.shape[3], t.shape[3], k, 2, transposed, mode, (8, 8)):
                return None
            return sh[id(t)][1]

        zpool = [None, 0]                   # fused eval path: zeroed pool for the per-block GroupNorm sums, next free float

        def take_sums(nc, device):    # [B][nc / 16][2] of the pool: ONE zero fill per forward for every block's sums (+ final_conv.0's)
            nsum = B * (nc // 16) * 2
            if zpool[0] is None:
                zpool[0] = torch.zeros(2 * B * (sum(rb["cout"] // 16 + 1 for rb in A.res_blocks) + A.dim // 16 + 1), device=device, dtype=torch.int64)
            assert zpool[1] + nsum <= zpool[0].numel()
            sl = zpool[0][zpool[1]:zpool[1] + nsum]
            zpool[1] += (nsum + 3) // 4 * 4
            return sl

        def shadow(t):
            ent = sh.get(id(t))
            if ent is None:
                ent = (t, K.to_bf16(t))
                sh[id(t)] = ent
            return ent[1]

        def conv(inp, pre, k, stride=1, pad=0, x2=None, residual=None, transposed_conv=False, bias=True, out_dtype=torch.float32,
                 gn_sums=None, want16=False, soft=False):
            w = sv[pre + "weight"]
            kh, kw, ci, co = w.shape
            if x2 is None and residual is None and stride == 1 and not transposed_conv:
                if K.small_cin_supported(k, ci, co):                                    # image -> features
                    od = BF if (out_dtype == BF and K.small_cin_bf16_supported(k, B, inp.shape[1], inp.shape[2], ci, co, K.ld_of(inp))) else torch.float32
                    return K.conv_small_cin_fwd(inp, w, sv[pre + "bias"] if bias else None, co, k, out_dtype=od)
                if k == 1 and K.small_cout_supported(0, ci, co):                        # features -> image
                    return K.conv1x1_small_cout(0, inp, w, bias=sv[pre + "bias"] if bias else None, Cs=co)
            if mode == K.MODE_BF16 and k in (1, 3) and stride == 1 and not transposed_conv:
                y = K.conv3x3_bf16w(inp, wf_sh[offs[pre + "weight"]:], K=ci, Nc=co, flip=False, ksize=k, x2=x2,
                                    bias=sv[pre + "bias"] if bias else None, residual=residual, out_dtype=out_dtype, gn_sums=gn_sums,
                                    want16=want16, wq=wfq_sh[offs[pre + "weight"]:])
                if y is not None:
                    if want16:                      # the epilogue wrote the bf16 copy along: register it for the consumers
                        sh[id(y[0])] = y
                        return y[0]
                    return y
                if soft:                            # the caller has another operand to offer (res_conv: the fp32 tensors)
                    return None
            assert not want16, "the bf16 copy rides in the tile kernel's epilogue only"
            assert gn_sums is None, "GroupNorm sums ride in the tile kernel's epilogue only"
            if mode == K.MODE_FP32 and k == 3 and stride == 1 and not transposed_conv and out_dtype == torch.float32:
                y = K.conv3x3_f32(inp, wfq32_sh[offs[pre + "weight"]:], K=ci, Nc=co, flip=False, x2=x2,
                                  bias=sv[pre + "bias"] if bias else None, residual=residual)
                if y is not None:
                    return y
            if mode == K.MODE_FP32 and k == 1 and stride == 1 and not transposed_conv and out_dtype == torch.float32:
                y = K.conv1x1_f32(inp, wfq32_sh[offs[pre + "weight"]:], K=ci, Nc=co, flip=False, x2=x2,
                                  bias=sv[pre + "bias"] if bias else None, residual=residual)
                if y is not None:
                    return y
            ih, iw = inp.shape[1], inp.shape[2]
            if transposed_conv:
                oh, ow = ih * stride, iw * stride
            else:
                oh, ow = (ih + 2 * pad - kh) // stride + 1, (iw + 2 * pad - kw) // stride + 1
            if mode == K.MODE_BF16 and stride == 2 and x2 is None and inp.dtype == BF and out_dtype == torch.float32:
                # Downsample / Upsample on the tap-gather kernel (round 4): bf16 copy of the input, fragment-order weights
                y = K.conv_gt(inp, wfq_sh[offs[pre + "weight"]:], kh=kh, kw=kw, stride=stride, pad=pad, transposed=transposed_conv, K=ci, Nc=co,
                              out_hw=(oh, ow), bias=sv[pre + "bias"] if bias else None, residual=residual, want16=use_sh)
                if y is not None:
                    if use_sh:                      # training: the next Block's conv and its weight gradient read the copy
                        sh[id(y[0])] = y
                        return y[0]
                    return y
            if mode == K.MODE_FP32 and stride == 2 and x2 is None and inp.dtype == torch.float32 and out_dtype == torch.float32:
                y = K.conv_gt(inp, wfq32_sh[offs[pre + "weight"]:], kh=kh, kw=kw, stride=stride, pad=pad, transposed=transposed_conv, K=ci, Nc=co,
                              out_hw=(oh, ow), bias=sv[pre + "bias"] if bias else None, residual=residual, mode=mode)
                if y is not None:
                    return y
            assert out_dtype == torch.float32 and (inp.dtype == torch.float32 or (
                stride == 2 and K.igemm_bf16_in_supported(ci, co, k, stride, transposed_conv, mode, (oh, ow)))), \
                "bf16 block storage needs the tile kernel"
            y = K.conv_igemm(inp, w, kh=kh, kw=kw, stride=stride, pad=pad, transposed=transposed_conv, w_kn=True,
                             K=ci, Nc=co, out_hw=(oh, ow), mode=mode, x2=x2,
                             bias=sv[pre + "bias"] if bias else None, residual=residual,
                             wb=wf_sh[offs[pre + "weight"]:] if mode == K.MODE_BF16 else None)
            return y

        def resblock(blk, inp, x2=None, want_out16=False, out16_in_eval=True):
            pre, co, ci = blk["pre"], blk["cout"], blk["cin"]
            # bf16 storage of c1 / h1 / c2 when every kernel touching them has a bf16 path for this shape
            lo16 = False
            auto = self.block_storage == "auto"
            if mode == K.MODE_BF16 and self.block_storage in ("bf16", "auto") and co % 32 == 0 and B % 8 == 0:
                ok2 = K.fast3x3_supported(B, inp.shape[1], inp.shape[2], co, co)
                lo16 = ok2[0] and ok2[1] and not (auto and K.conv3x3_uses_splitk(B, inp.shape[1], inp.shape[2], co, co))
            k1 = inp.shape[3] if x2 is not None else None
            c1_16 = (lo16 and ci % 32 == 0 and all(K.fast3x3_supported(B, inp.shape[1], inp.shape[2], ci, co, k1))
                     and not (auto and K.conv3x3_uses_splitk(B, inp.shape[1], inp.shape[2], ci, co, k1)))
            if (not c1_16 and lo16 and self.ends16 and x2 is None and K.small_cin_supported(3, ci, co, wgrad=True)
                    and K.small_cin_bf16_supported(3, B, inp.shape[1], inp.shape[2], ci, co, K.ld_of(inp))):
                c1_16 = True                      # the image -> features conv: its VALU kernels write c1 / read its gradient as bf16
            inp_c, x2_c = inp, x2
            if c1_16 and use_sh and inp.dtype == torch.float32 and inp.shape[3] % 8 == 0:
                inp_c, x2_c = shadow(inp), (shadow(x2) if x2 is not None else None)
            elif c1_16 and eval16 and x2 is None and id(inp) in sh:
                inp_c = sh[id(inp)][1]            # inference: the copy the producing GroupNorm kernel wrote along (no conversion launches)
            tb = tb_all[:, blk["tcol"]:blk["tcol"] + co] if tb_all is not None else None
            hw = inp.shape[1] * inp.shape[2]
            fmode = str(self.fuse_gn_conv)
            sums_ok = ((co // _GN_GROUPS) % 16 == 0 and hw % 32 == 0 and ci % 32 == 0
                       and all(K.fast3x3_supported(B, inp.shape[1], inp.shape[2], ci, co, k1)))
            fuse = (not record and mode == K.MODE_BF16 and fmode != "0" and c1_16 == lo16
                    and K.conv3x3_gn_mish_supported(B, inp.shape[1], inp.shape[2], co, co))
            if fuse and fmode == "auto":
                fuse = lo16 and sums_ok and K.conv3x3_pw_gn_mish_picked(B, inp.shape[1], inp.shape[2], co, co)
            # ... with block1's GroupNorm statistics taken from conv1's epilogue when the tile kernel runs it (no pass over c1 at all)
            sums = None
            if fuse and fmode in ("2", "auto") and sums_ok:
                sums = take_sums(co, inp.device)
            r_pre = None
            if (blk["res"] and x2 is None and sums is None and self.small_cin_dual and K.small_cin_supported(3, ci, co)
                    and K.small_cin_dual_supported(B, inp.shape[1], inp.shape[2], ci, co, K.ld_of(inp))):
                # the image -> features block: its 3x3 conv and its res_conv read the same 3-channel image -- one launch (the 1x1's input is the centre tap)
                od = BF if (c1_16 and K.small_cin_bf16_supported(3, B, inp.shape[1], inp.shape[2], ci, co, K.ld_of(inp))) else torch.float32
                # inference: this is the forward's first launch -- it clears the pool of GroupNorm sums the later conv epilogues add into
                if not record and mode == K.MODE_BF16 and zpool[0] is None and str(self.fuse_gn_conv) != "0" and self.cin_dual_chores:
                    zpool[0] = torch.empty(2 * B * (sum(rb["cout"] // 16 + 1 for rb in A.res_blocks) + A.dim // 16 + 1), device=inp.device, dtype=torch.int64)
                    zfill = zpool[0]
                else:
                    zfill = None
                c1, r_pre = K.conv_small_cin_fwd_dual(inp, sv[pre + "block1.block.0.weight"], sv[pre + "block1.block.0.bias"],
                                                      sv[pre + "res_conv.weight"], sv[pre + "res_conv.bias"], co, out_dtype=od, zero=zfill,
                                                      gather=tb_pending.pop() if tb_pending else None)
            else:
                c1 = conv(inp_c, pre + "block1.block.0.", 3, 1, 1, x2=x2_c, out_dtype=BF if c1_16 else torch.float32, gn_sums=sums)
            c2 = None
            if fuse:
                # inference / sampling: GroupNorm-apply + Mish + time bias ride in block2's conv staging (the named fused kernel);
                # h1 is never materialised.  Training keeps h1: the weight gradient of block2's conv reads it.
                if sums is not None:
                    st1 = None
                    c2 = K.conv3x3_gn_mish(c1, None, wf_sh[offs[pre + "block2.block.0.weight"]:], K=co, Nc=co, bias=sv[pre + "block2.block.0.bias"],
                                           gn=(sums, sv[pre + "block1.block.1.weight"], sv[pre + "block1.block.1.bias"], tb, _GN_GROUPS, 1e-5),
                                           wq=wfq_sh[offs[pre + "block2.block.0.weight"]:])
                else:
                    st1, coef = K.gn_stats_coef(c1, sv[pre + "block1.block.1.weight"], sv[pre + "block1.block.1.bias"], temb=tb)
                    c2 = K.conv3x3_gn_mish(c1, coef, wf_sh[offs[pre + "block2.block.0.weight"]:], K=co, Nc=co, bias=sv[pre + "block2.block.0.bias"],
                                           wq=wfq_sh[offs[pre + "block2.block.0.weight"]:])
                h1 = None
            if c2 is None:
                h1, st1 = K.gn_mish_fwd(c1, sv[pre + "block1.block.1.weight"], sv[pre + "block1.block.1.bias"], temb=tb,
                                        out_dtype=BF if lo16 else torch.float32)
                c2 = conv(h1, pre + "block2.block.0.", 3, 1, 1, out_dtype=BF if lo16 else torch.float32)
            # res_conv reads the bf16 copies where block1's conv does (the tile kernels round an fp32 input to bf16 while staging it: the
            # same numbers, half the bytes
            # -- where the 1x1 tile kernel takes the layer; otherwise the fp32 tensors go to the generic kernel as before)
            r = inp
            if r_pre is not None:
                r = r_pre
            elif blk["res"]:
                r = None
                if inp_c is not inp and inp_c.dtype == BF and (x2 is None or x2_c is not None):
                    r = conv(inp_c, pre + "res_conv.", 1, x2=x2_c, soft=True)
                if r is None:
                    r = conv(inp, pre + "res_conv.", 1, x2=x2)
            if want_out16 and (use_sh or (eval16 and out16_in_eval)) and co % 32 == 0:
                out, st2, out16 = K.gn_mish_fwd(c2, sv[pre + "block2.block.1.weight"], sv[pre + "block2.block.1.bias"], residual=r, want16=True)
                sh[id(out)] = (out, out16)
            else:
                out, st2 = K.gn_mish_fwd(c2, sv[pre + "block2.block.1.weight"], sv[pre + "block2.block.1.bias"], residual=r)
            if record:
                tape.append(("res", blk, inp, x2, c1, st1, h1, c2, st2, out, inp_c, x2_c))
            return out

        def attention(at, inp, feeds_s2=False):
            pre = at["pre"]
            # bf16 storage of the attention-internal tensors (LayerNorm output, qkv, attention output and their
            # gradients) when the 1x1 tile kernels take both projections; the residual stream stays fp32
            c = inp.shape[3]
            a16 = (mode == K.MODE_BF16 and self.block_storage in ("bf16", "auto") and B % 8 == 0
                   and all(K.fast1x1_supported(B, inp.shape[1], inp.shape[2], c, 3 * _HEADS * _DHEAD))
                   and all(K.fast1x1_supported(B, inp.shape[1], inp.shape[2], _HEADS * _DHEAD, c)))
            dt = BF if a16 else torch.float32
            nq = 3 * _HEADS * _DHEAD
            if (a16 and self.fuse_ln_qkv and (not record or self.fuse_ln_qkv_train) and inp.dtype == torch.float32
                    and K.ln_conv1x1_supported(B, inp.shape[1], inp.shape[2], c, nq, K.ld_of(inp))):
                # the LayerNorm rides in to_qkv's staging; inference never materialises the normalised tensor, training has the same
                # launch write it along (to_qkv's weight gradient reads it)
                r_ = K.ln_conv1x1(inp, sv[pre + "fn.norm.g"], sv[pre + "fn.norm.b"], wfq_sh[offs[pre + "fn.fn.to_qkv.weight"]:], Nc=nq,
                                  want_ln=record)
                qkv, ln = r_ if record else (r_, None)
            else:
                ln = K.chan_layernorm_fwd(inp, sv[pre + "fn.norm.g"], sv[pre + "fn.norm.b"], out_dtype=dt)
                qkv = conv(ln, pre + "fn.fn.to_qkv.", 1, bias=False, out_dtype=dt)
            if not record and a16 and self.fold_attn and mode == K.MODE_BF16:
                # inference: the attention output is never formed -- to_out runs on q with per-sample weights W_out blockdiag(ctx^T)
                w16 = bool(feeds_s2 and self.dual_out and inp.shape[3] % 32 == 0)
                r_ = K.linattn_to_out_folded(qkv, wf_sh[offs[pre + "fn.fn.to_out.weight"]:], c, sv[pre + "fn.fn.to_out.bias"], inp,
                                             heads=_HEADS, want16=w16)
                if r_ is not None:
                    if w16:
                        sh[id(r_[0])] = r_
                        return r_[0]
                    return r_
            ao, ctx, kstat = K.linattn_fwd(qkv, _HEADS)
            # training: the block's output is a residual-stream tensor whose bf16 copy is wanted by the skip connection's consumer,
            # Downsample / Upsample and the next Block's conv: written by to_out's epilogue instead of a conversion pass
            # (inference: only where a Downsample / Upsample reads it -- their ring kernel takes 64-channel stages of a bf16 input)
            out = conv(ao, pre + "fn.fn.to_out.", 1, residual=inp,
                       want16=bool((use_sh or (feeds_s2 and not record and mode == K.MODE_BF16)) and a16 and self.dual_out
                                   and inp.shape[3] % 32 == 0))
            if record:
                tape.append(("attn", at, inp, ln, qkv, ctx, kstat, ao, out))
            return out

        skips = []
        h = x
        for lvl in A.downs:
            h = resblock(lvl["res1"], h, want_out16=True)        # feeds res2's first conv
            assert not tb_pending, "the time-bias rows were left to a launch that did not happen"
            h = resblock(lvl["res2"], h)
            h = attention(lvl["attn"], h, feeds_s2=lvl["down"] is not None)
            skips.append(h)
            if lvl["down"] is not None:
                inp = h
                # training: the skip tensor's bf16 copy (its consumer in the up path wants it too) feeds Downsample and its weight gradient
                inp_c = shadow(inp) if s2_copy(inp.shape[3], 3, False) else s2_eval_copy(inp, 3, False)
                h = conv(inp if inp_c is None else inp_c, lvl["down"]["pre"], 3, 2, 1)
                if record:
                    tape.append(("down", lvl["down"], inp, h, inp_c))
        h = resblock(A.mid1, h)
        h = attention(A.mid_attn, h)
        h = resblock(A.mid2, h, want_out16=True, out16_in_eval=False)    # feeds the first up block's conv (with the skip: training only)
        for lvl in A.ups:
            h = resblock(lvl["res1"], h, x2=skips.pop(), want_out16=True)          # cat((x, skip)) read in place (ddpm.py:255)
            h = resblock(lvl["res2"], h)
            h = attention(lvl["attn"], h, feeds_s2=True)
            inp = h
            inp_c = shadow(inp) if s2_copy(inp.shape[3], 4, True) else s2_eval_copy(inp, 4, True)
            h = conv(inp if inp_c is None else inp_c, lvl["up"]["pre"], 4, 2, 1, transposed_conv=True)
            if record:
                tape.append(("up", lvl["up"], inp, h, inp_c))
        # final_conv.0 is a Block too (ddpm.py:232-234): its conv output is block-internal and stored like every other Block's
        # (bf16 in bf16 mode); the GroupNorm output feeds the 128 -> 3 conv, an fp32 VALU kernel, and stays fp32
        cdim = h.shape[3]
        f16 = False
        if mode == K.MODE_BF16 and self.block_storage in ("bf16", "auto") and cdim % 64 == 0 and B % 8 == 0 and self.final_block16:
            okf = K.fast3x3_supported(B, h.shape[1], h.shape[2], cdim, cdim)
            f16 = okf[0] and okf[1] and not K.conv3x3_uses_splitk(B, h.shape[1], h.shape[2], cdim, cdim)
        h_c = shadow(h) if (f16 and use_sh) else h
        ncs = sv["final_conv.1.weight"].shape[3]
        # inference: final_conv.0's GroupNorm-apply + Mish ride in final_conv.1's load, the statistics come from final_conv.0's conv epilogue
        fuse_final = (f16 and not record and self.fuse_final and str(self.fuse_gn_conv) != "0" and (cdim // _GN_GROUPS) % 16 == 0
                      and (h.shape[1] * h.shape[2]) % 32 == 0 and K.small_cout_gn_supported(cdim, ncs, _GN_GROUPS)
                      and K.conv3x3_pw_gn_mish_picked(B, h.shape[1], h.shape[2], cdim, cdim))
        if fuse_final:
            sumsF = take_sums(cdim, h.device) if cdim == A.dim else K.gn_sums_buffer(B, cdim, h.device)
            cF = conv(h_c, "final_conv.0.block.0.", 3, 1, 1, out_dtype=BF, gn_sums=sumsF)
            return K.conv1x1_small_cout_gn(cF, sumsF, sv["final_conv.0.block.1.weight"], sv["final_conv.0.block.1.bias"],
                                           sv["final_conv.1.weight"], sv["final_conv.1.bias"], ncs, groups=_GN_GROUPS), tape
        cF = conv(h_c, "final_conv.0.block.0.", 3, 1, 1, out_dtype=BF if f16 else torch.float32)
        # round 4: hF as bf16 too where the 128 -> 3 kernels take it (they widen on load; the gradient wrt hF is written as bf16)
        hF16 = f16 and self.ends16 and all(K.small_cout_supported(op, cdim, sv["final_conv.1.weight"].shape[3]) for op in (0, 1, 2))
        hF, stF = K.gn_mish_fwd(cF, sv["final_conv.0.block.1.weight"], sv["final_conv.0.block.1.bias"], out_dtype=BF if hF16 else torch.float32)
        eps = conv(hF, "final_conv.1.", 1)
        if record:
            tape.append(("final", h, cF, stF, hF, eps, h_c))
            tape.append(("input", x))
        return eps, tape

    # ------------------------------------------------------------------ kernel-level backward
    def _backward(self, tape, dy_nchw, need_dx=False):
        d_eps = K.nchw_to_nhwc(dy_nchw)
        return self.backward_nhwc(tape, d_eps, need_dx)

    def backward_nhwc(self, tape, d_eps, need_dx=False):
        """Replays the tape in reverse.  d_eps: NHWC gradient of the eps prediction."""
        A, sv, mode = self._arch, self._sv, _mode_id(self.compute_mode)
        gflat = self.flat_grads
        if not self.accumulate_grads:
            gflat.zero_()                                  # wgrad / norm kernels accumulate atomically
        gv = self._gv
        offs = self._offs
        if mode == K.MODE_BF16:
            wd_sh, wf_sh, wdq_sh, wfq_sh = self._shadows()
        else:
            wdq32_sh, wfq32_sh = self._shadows32()
        G = _GradMap()
        # data parallel (a grad-ready hook is set): the two Upsample layers come early in backward, the two Downsample layers last; with
        # all four in one launch at the very end their 8 MB of gradients would be all-reduced after backward, exposed -- two per launch
        wq = K.WgradQueue(group=int(K.debug_knob("MI_DDPM_WGRAD_GROUP", "8")),
                          group_s2=2 if self.grad_ready_hook is not None else None)
        x_in = tape[-1][1]
        B = x_in.shape[0]
        # (every column is WRITTEN by its block's GroupNorm backward -- one workgroup per (sample, channel) stores its sum -- before the time MLP's
        #  backward reads it: no fill)
        dtb_all = torch.empty((B, A.mlp_rows), device=x_in.device, dtype=torch.float32)

        def s2_push(dy, inp, pre, k, ci, co, transposed_conv, ihw, ohw, bias):
            """Downsample / Upsample weight gradient through the LDS-DMA kernel: both operands as bf16 (the input's copy is the one
            the skip connection's consumer reads when there is one; the output gradient is rounded by the pass that also sums
            it into the bias gradient)."""
            big, small = (ohw, ihw) if transposed_conv else (ihw, ohw)
            if dy.dtype != torch.float32 or not K.s2_wgrad_supported(inp.shape[0], k, ci, co, transposed_conv, big, small, mode):
                return None
            x16 = inp if inp.dtype == torch.bfloat16 else K.to_bf16(inp)
            dy16 = K.to_bf16(dy, colsum_out=gv[pre + "bias"] if bias == "colsum" else None, defer=wq)
            ok = wq.push_s2(x16, dy16, gv[pre + "weight"], k=k, Ci=ci, Cj=co, gather_i=not transposed_conv, grid_g=big, grid_d=small,
                            mode=mode)
            assert ok
            return dy16

        def conv_bwd(dy, inp, pre, k, stride=1, pad=0, x2=None, transposed_conv=False, bias="colsum", want_dx=True, winp=None, wx2=None):
            """Gradients of y = conv(inp [|x2]); dy may be a channel slice.  The gradient wrt inp has inp's dtype
            (bf16 for the block-internal h1, fp32 for everything on the residual stream).  winp / wx2: the tensors the forward
            conv actually read (bf16 copies of inp / x2), used as the weight-gradient operand."""
            w = sv[pre + "weight"]
            kh, kw, ci, co = w.shape
            ih, iw = inp.shape[1], inp.shape[2]
            oh, ow = dy.shape[1], dy.shape[2]
            winp = inp if winp is None else winp
            wx2 = x2 if wx2 is None else wx2
            dy16 = None
            if (x2 is None and stride == 1 and not transposed_conv and k == 1 and K.small_cout_supported(2, ci, co)
                    and K.small_cout_supported(1, ci, co)):
                if want_dx and self.small_cout_bwd1 and ci in (64, 128, 256):
                    buf, acc = G.target(inp)                      # weight gradient and data gradient from one pass over (inp, dy)
                    K.conv1x1_small_cout_bwd(inp, dy, w, gv[pre + "weight"], buf, accumulate=acc)
                    want_dx = False
                else:
                    K.conv1x1_small_cout(2, inp, None, b=dy, out=gv[pre + "weight"])
                if bias == "colsum":
                    K.colsum(dy, gv[pre + "bias"], defer=wq)
                if want_dx:
                    buf, acc = G.target(inp)
                    K.conv1x1_small_cout(1, dy, w, out=buf, accumulate=acc)
                return
            small_cin = x2 is None and stride == 1 and not transposed_conv and K.small_cin_supported(k, ci, co, wgrad=True)
            if small_cin:
                K.conv_small_cin_wgrad(inp, dy, gv[pre + "weight"], k)
            elif (stride == 2 and mode == K.MODE_BF16 and self.s2_wgrad_tr and
                  (dy16 := s2_push(dy, winp, pre, kh, ci, co, transposed_conv, (ih, iw), (oh, ow), bias)) is not None):
                bias = None                                           # Downsample / Upsample: deferred, bias gradient rode along
            elif (stride == 2 and mode == K.MODE_FP32 and x2 is None and pad == 1 and
                  K.conv_s2_wgrad_f32(inp, dy, gv[pre + "weight"], k=kh, Ci=ci, Cj=co, gather_i=not transposed_conv,
                                      grid_g=(oh, ow) if transposed_conv else (ih, iw), grid_d=(ih, iw) if transposed_conv else (oh, ow))):
                pass                    # fp32 mode (round 6): Downsample / Upsample as gathered 1x1 problems of the exact-fp32 kernel (bias: column sum below)
            elif transposed_conv:   # dW[tap][ci][co] = sum over input pixels  x[j] * dy[gather(j, tap)]
                K.conv_wgrad(inp, dy, gv[pre + "weight"], kh=kh, kw=kw, stride=stride, pad=pad, gather_i=False,
                             Ci=ci, Cj=co, grid_g=(oh, ow), grid_d=(ih, iw), mode=mode)
            elif k == 3 and stride == 1 and bias != "colsum" and mode == K.MODE_BF16:
                # Block conv: deferred, several layers per launch (K.WgradQueue)
                wq.push(winp, dy, gv[pre + "weight"], Ci=ci, Cj=co, hw=(ih, iw), mode=mode, P2=wx2)
            elif (k == 3 and stride == 1 and not transposed_conv and mode == K.MODE_FP32 and winp.dtype == torch.float32
                  and dy.dtype == torch.float32 and ci % 64 == 0):
                # fp32 mode: the same queue, the exact-fp32 instantiation of the kernel (the bias gradient stays a column sum below)
                wq.push(winp, dy, gv[pre + "weight"], Ci=ci, Cj=co, hw=(ih, iw), mode=mode, P2=wx2)
            elif k == 1 and stride == 1 and mode == K.MODE_BF16 and winp.dtype == torch.bfloat16:
                # to_qkv / to_out / res_conv: deferred too; the bias gradient rides along when dy is the fp32 stream gradient
                fuse_b = bias == "colsum" and dy.dtype == torch.float32
                wq.push1x1(winp, dy, gv[pre + "weight"], Ci=ci, Cj=co, hw=(ih, iw), mode=mode, P2=wx2,
                           dbias=gv[pre + "bias"] if fuse_b else None)
                if fuse_b:
                    bias = None
            elif (k == 1 and stride == 1 and not transposed_conv and mode == K.MODE_FP32 and winp.dtype == torch.float32
                  and dy.dtype == torch.float32 and ci % 64 == 0 and co % 32 == 0):
                # fp32 mode (round 4): the same queue, the exact-fp32 instantiation of the 1x1 kernel; the bias gradient rides along
                fuse_b = bias == "colsum"
                wq.push1x1(winp, dy, gv[pre + "weight"], Ci=ci, Cj=co, hw=(ih, iw), mode=mode, P2=wx2,
                           dbias=gv[pre + "bias"] if fuse_b else None)
                if fuse_b:
                    bias = None
            else:
                if stride == 2:
                    winp = inp                                        # round 1's ring kernel converts while it stages
                K.conv_wgrad(winp, dy, gv[pre + "weight"], kh=kh, kw=kw, stride=stride, pad=pad, gather_i=True,
                             Ci=ci, Cj=co, grid_g=(ih, iw), grid_d=(oh, ow), mode=mode, P2=wx2,
                             dbias=gv[pre + "bias"] if bias == "colsum" else None)
                bias = None                                           # handled (fused or by conv_wgrad's fallback)
            if bias == "colsum":
                K.colsum(dy, gv[pre + "bias"], defer=wq)
            if not want_dx:
                return
            fast = mode == K.MODE_BF16 and k in (1, 3) and stride == 1 and not transposed_conv
            if x2 is None:
                buf, acc = G.target(inp)
                if fast and K.conv3x3_bf16w(dy, wd_sh[offs[pre + "weight"]:], K=co, Nc=ci, flip=True, ksize=k, out=buf,
                                            accumulate=acc, wq=wdq_sh[offs[pre + "weight"]:]) is not None:
                    return
                if (mode == K.MODE_FP32 and k in (1, 3) and stride == 1 and not transposed_conv and buf.dtype == torch.float32
                        and (K.conv3x3_f32 if k == 3 else K.conv1x1_f32)(dy, wdq32_sh[offs[pre + "weight"]:], K=co, Nc=ci, flip=True, out=buf,
                                                                          accumulate=acc) is not None):
                    return
                if small_cin and dy.dtype != torch.float32:
                    dy = dy.float()               # the input gradient of the image -> features conv (rarely wanted): the generic kernel reads fp32
                assert dy.dtype == torch.float32 and buf.dtype == torch.float32, "bf16 block storage needs the tile kernel"
                if (mode == K.MODE_FP32 and stride == 2 and K.conv_gt(dy, wdq32_sh[offs[pre + "weight"]:], kh=kh, kw=kw, stride=stride, pad=pad,
                                                                     transposed=not transposed_conv, K=co, Nc=ci, out_hw=(ih, iw), out=buf,
                                                                     accumulate=acc, mode=mode) is not None):
                    return
                if dy16 is not None and stride == 2 and K.conv_gt(dy16, wdq_sh[offs[pre + "weight"]:], kh=kh, kw=kw, stride=stride, pad=pad,
                                                                  transposed=not transposed_conv, K=co, Nc=ci, out_hw=(ih, iw), out=buf,
                                                                  accumulate=acc) is not None:
                    return                                            # Downsample / Upsample data gradient on the tap-gather kernel
                if dy16 is not None and K.igemm_bf16_in_supported(co, ci, k, stride, not transposed_conv, mode, (ih, iw)):
                    dy = dy16                                         # the copy the weight gradient reads: half the bytes, 64-channel stages
                K.conv_igemm(dy, w, kh=kh, kw=kw, stride=stride, pad=pad, transposed=not transposed_conv, w_kn=False,
                             K=co, Nc=ci, out_hw=(ih, iw), mode=mode, out=buf, accumulate=acc,
                             wb=wd_sh[offs[pre + "weight"]:] if mode == K.MODE_BF16 else None)
            else:
                # gradient of the (virtual) concat: one buffer, the two sources take channel slices of it
                cat = G._g.get(("cat", id(inp)))
                acc = cat is not None
                if cat is None:
                    cat = torch.empty((B, ih, iw, ci), device=dy.device, dtype=torch.float32)
                    G._g[("cat", id(inp))] = cat
                if fast and K.conv3x3_bf16w(dy, wd_sh[offs[pre + "weight"]:], K=co, Nc=ci, flip=True, ksize=k, out=cat,
                                            accumulate=acc, wq=wdq_sh[offs[pre + "weight"]:]) is not None:
                    return
                if (mode == K.MODE_FP32 and k in (1, 3) and stride == 1 and not transposed_conv
                        and (K.conv3x3_f32 if k == 3 else K.conv1x1_f32)(dy, wdq32_sh[offs[pre + "weight"]:], K=co, Nc=ci, flip=True, out=cat,
                                                                          accumulate=acc) is not None):
                    return
                K.conv_igemm(dy, w, kh=kh, kw=kw, stride=stride, pad=pad, transposed=True, w_kn=False,
                             K=co, Nc=ci, out_hw=(ih, iw), mode=mode, out=cat, accumulate=acc)

        def res_bwd(rec):
            _, blk, inp, x2, c1, st1, h1, c2, st2, out, inp_c, x2_c = rec
            pre = blk["pre"]
            dout = G.take(out)
            # residual branch first: its gradient is dout itself (read before anything accumulates into it)
            is_first = inp is x_in
            want_dx = (not is_first) or need_dx
            if blk["res"]:
                conv_bwd(dout, inp, pre + "res_conv.", 1, x2=x2, want_dx=want_dx, winp=inp_c, wx2=x2_c)
            else:
                G.add(inp, dout)
            dc2 = K.gn_mish_bwd(c2, st2, sv[pre + "block2.block.1.weight"], sv[pre + "block2.block.1.bias"], dout,
                                dgamma=gv[pre + "block2.block.1.weight"], dbeta=gv[pre + "block2.block.1.bias"],
                                dbias=gv[pre + "block2.block.0.bias"], out_dtype=c2.dtype)
            conv_bwd(dc2, h1, pre + "block2.block.0.", 3, 1, 1, bias=None)
            dh1 = G.take(h1)
            dtb = dtb_all[:, blk["tcol"]:blk["tcol"] + blk["cout"]] if A.with_time else None
            dc1 = K.gn_mish_bwd(c1, st1, sv[pre + "block1.block.1.weight"], sv[pre + "block1.block.1.bias"], dh1,
                                dgamma=gv[pre + "block1.block.1.weight"], dbeta=gv[pre + "block1.block.1.bias"],
                                dtemb=dtb, dbias=gv[pre + "block1.block.0.bias"], out_dtype=c1.dtype)
            conv_bwd(dc1, inp, pre + "block1.block.0.", 3, 1, 1, x2=x2, bias=None, want_dx=want_dx, winp=inp_c, wx2=x2_c)
            if x2 is not None:
                cat = G._g.pop(("cat", id(inp)))
                k1 = inp.shape[3]
                G.add(inp, cat[..., :k1])
                G.add(x2, cat[..., k1:])

        def attn_bwd(rec):
            _, at, inp, ln, qkv, ctx, kstat, ao, out = rec
            pre = at["pre"]
            dout = G.take(out)
            G.add(inp, dout)                                           # Residual: fn(x) + x  (ddpm.py:45)
            conv_bwd(dout, ao, pre + "fn.fn.to_out.", 1)
            dao = G.take(ao)
            dqkv = K.linattn_bwd(qkv, ctx, kstat, dao, _HEADS)
            conv_bwd(dqkv, ln, pre + "fn.fn.to_qkv.", 1, bias=None)
            dln = G.take(ln)
            # the LayerNorm gradient accumulates INTO dout (grad[inp] aliases it, above): the deferred to_out weight gradient
            # that reads dout has to be issued first
            wq.flush(kinds=(1,))
            buf, acc = G.target(inp)
            K.chan_layernorm_bwd(inp, sv[pre + "fn.norm.g"], dln, buf, acc, gv[pre + "fn.norm.g"], gv[pre + "fn.norm.b"], defer=wq)

        hook = self.grad_ready_hook
        dx_in = None
        # a record's flat-gradient range is final once the weight gradients it deferred have been issued: ranges are reported
        # in tape order, each after the flush that covers it
        fifo: List[tuple] = []

        def drain():
            while fifo and fifo[0][1] <= wq.flushed:
                hook(*fifo.pop(0)[0])
        wq.on_flush = drain if hook is not None else None
        for rec in reversed(tape):
            kind = rec[0]
            if kind == "input":
                continue
            if hook is not None:
                rng = (A.final_range if kind == "final" else A.time_range if kind == "time" else rec[1]["range"])
            if kind == "final":
                _, h, cF, stF, hF, eps, h_c = rec
                conv_bwd(d_eps, hF, "final_conv.1.", 1)
                dhF = G.take(hF)
                dcF = K.gn_mish_bwd(cF, stF, sv["final_conv.0.block.1.weight"], sv["final_conv.0.block.1.bias"], dhF,
                                    dgamma=gv["final_conv.0.block.1.weight"], dbeta=gv["final_conv.0.block.1.bias"],
                                    dbias=gv["final_conv.0.block.0.bias"], out_dtype=cF.dtype)
                conv_bwd(dcF, h, "final_conv.0.block.0.", 3, 1, 1, bias=None, winp=h_c)
            elif kind == "res":
                res_bwd(rec)
            elif kind == "attn":
                attn_bwd(rec)
            elif kind == "down":
                _, dn, inp, out, inp_c = rec
                conv_bwd(G.take(out), inp, dn["pre"], 3, 2, 1, winp=inp_c)
            elif kind == "up":
                _, up, inp, out, inp_c = rec
                conv_bwd(G.take(out), inp, up["pre"], 4, 2, 1, transposed_conv=True, winp=inp_c)
            elif kind == "time":
                _, te, t1, a1, temb, mt = rec
                dim = A.dim

                def lin_bwd(dy, inp, gw, gb, w, want_dx=True):
                    o, i = w.shape
                    if K.small_gemm(True, False, dy, inp, out=gw.view(o, i), accumulate=True, allow_split=True) is None:     # dW += dy^T x
                        K.conv_wgrad(dy.view(B, 1, 1, o), inp.view(B, 1, 1, i), gw, kh=1, kw=1, stride=1, pad=0,
                                     gather_i=True, Ci=o, Cj=i, grid_g=(1, 1), grid_d=(1, 1), mode=K.MODE_FP32)
                    K.colsum(dy, gb)
                    if not want_dx:
                        return None
                    dx = K.small_gemm(False, False, dy, w, allow_split=True)                               # dx = dy W
                    if dx is not None:
                        return dx
                    dx = K.conv_igemm(dy.view(B, 1, 1, o), w, kh=1, kw=1, stride=1, pad=0, transposed=False, w_kn=True,
                                      K=o, Nc=i, out_hw=(1, 1), mode=K.MODE_FP32)
                    return dx.view(B, i)
                w_all = self._flat[A.mlp_w_off:A.mlp_w_off + A.mlp_rows * dim].view(A.mlp_rows, dim)
                gw_all = gflat[A.mlp_w_off:A.mlp_w_off + A.mlp_rows * dim]
                gb_all = gflat[A.mlp_b_off:A.mlp_b_off + A.mlp_rows]
                dmt = lin_bwd(dtb_all, mt, gw_all, gb_all, w_all)
                dtemb = K.mish_bwd(temb, dmt)
                da1 = lin_bwd(dtemb, a1, gv["time_mlp.3.weight"], gv["time_mlp.3.bias"], sv["time_mlp.3.weight"])
                dt1 = K.mish_bwd(t1, da1)
                lin_bwd(dt1, te, gv["time_mlp.1.weight"], gv["time_mlp.1.bias"], sv["time_mlp.1.weight"], want_dx=False)
            if hook is not None:
                wq.flush(kinds=(4,))        # data parallel: the record's deferred row sums go out now, its range must be final when reported
                fifo.append((rng, wq.pushed))
                drain()
        wq.flush()
        if hook is not None:
            drain()
        if need_dx:
            dx_in = K.nhwc_to_nchw(G.take(x_in))
        return dx_in


# --------------------------------------------------------------------------------------------
# diffusion process
# --------------------------------------------------------------------------------------------
def extract(a, t, x_shape):
    """a[t] broadcast to x (ddpm.py:263-266); kept for API compatibility."""
    out = a.gather(-1, t)
    return out.reshape(t.shape[0], *((1,) * (len(x_shape) - 1)))


def cosine_beta_schedule(timesteps, s=0.008):
    """Cosine schedule exactly as the reference computes it (ddpm.py:281-291): float64 numpy,
    T+1 grid points over [0, T+1]."""
    n = timesteps + 1
    grid = np.linspace(0, n, n)
    f = np.cos(((grid / n) + s) / (1 + s) * np.pi * 0.5) ** 2
    f = f / f[0]
    return np.clip(1 - (f[1:] / f[:-1]), a_min=0, a_max=0.999)


def linear_beta_schedule(timesteps):
    scale = 1000 / timesteps
    return torch.linspace(scale * 0.0001, scale * 0.02, timesteps, dtype=torch.float64)


# The reference module's three small helpers (ddpm.py:25-36, 268-273): none of them is on the training / sampling path, they are here so
# that `from src.models.ddpm import ...` of a script written against the reference keeps working.
def cycle(dl):
    """Endless iteration over a data loader, epoch after epoch."""
    while True:
        yield from dl


def num_to_groups(num, divisor):
    """`num` split into full groups of `divisor` and, if anything is left, one smaller group."""
    full, rest = divmod(int(num), int(divisor))
    return [int(divisor)] * full + ([rest] if rest else [])


def noise_like(shape, device, repeat=False):
    """Standard normal noise of `shape`; repeat=True draws ONE sample and repeats it along the batch axis."""
    if repeat:
        one = torch.randn((1, *shape[1:]), device=device)
        return one.repeat(shape[0], *((1,) * (len(shape) - 1)))
    return torch.randn(shape, device=device)


class _LossFunction(torch.autograd.Function):
    """q_sample -> UNet -> L1/L2 loss as one node: all three stay NHWC and on the HIP kernels."""

    @staticmethod
    def forward(ctx, x0, t, noise, anchor, gd):
        net = gd.denoise_fn
        xt, _ = K.q_sample(x0, noise, t, gd.sqrt_alphas_cumprod, gd.sqrt_one_minus_alphas_cumprod)
        eps, tape = net.forward_nhwc(xt, t, record=ctx.needs_input_grad[3])
        loss, dpred = K.eps_loss(eps, noise, 0 if gd.loss_type == "l1" else 1, want_grad=ctx.needs_input_grad[3])
        ctx.net, ctx.tape, ctx.dpred = net, tape, dpred
        return loss

    @staticmethod
    def backward(ctx, dloss):
        net, tape, dpred = ctx.net, ctx.tape, ctx.dpred
        ctx.tape = ctx.dpred = None
        # d loss is 1.0 in training_step; honour other scales without a host sync
        d_eps = dpred
        if dloss is not None:
            K.scale_by_device_scalar(d_eps, dloss)
        net.backward_nhwc(tape, d_eps, need_dx=False)
        return None, None, None, None, None


class GaussianDiffusion(nn.Module):
    """Reference `GaussianDiffusion` (ddpm.py:294-466): same ctor, buffers and methods."""

    def __init__(self, denoise_fn, *, image_size, channels=3, timesteps=1000, loss_type="l1", betas=None):
        super().__init__()
        self.channels = channels
        self.image_size = image_size
        self.denoise_fn = denoise_fn
        if betas is not None:
            betas = betas.detach().cpu().numpy() if isinstance(betas, torch.Tensor) else np.asarray(betas)
        else:
            betas = cosine_beta_schedule(timesteps)
        betas = betas.astype(np.float64)
        alphas = 1.0 - betas
        ac = np.cumprod(alphas, axis=0)
        ac_prev = np.append(1.0, ac[:-1])
        self.num_timesteps = int(betas.shape[0])
        if loss_type not in ("l1", "l2"):
            raise NotImplementedError(loss_type)
        self.loss_type = loss_type
        post_var = betas * (1.0 - ac_prev) / (1.0 - ac)
        tables = (
            ("betas", betas), ("alphas_cumprod", ac), ("alphas_cumprod_prev", ac_prev),
            ("sqrt_alphas_cumprod", np.sqrt(ac)), ("sqrt_one_minus_alphas_cumprod", np.sqrt(1.0 - ac)),
            ("log_one_minus_alphas_cumprod", np.log(1.0 - ac)), ("sqrt_recip_alphas_cumprod", np.sqrt(1.0 / ac)),
            ("sqrt_recipm1_alphas_cumprod", np.sqrt(1.0 / ac - 1)), ("posterior_variance", post_var),
            ("posterior_log_variance_clipped", np.log(np.maximum(post_var, 1e-20))),
            ("posterior_mean_coef1", betas * np.sqrt(ac_prev) / (1.0 - ac)),
            ("posterior_mean_coef2", (1.0 - ac_prev) * np.sqrt(alphas) / (1.0 - ac)),
        )
        for name, val in tables:                      # same 12 fp32 buffers, same order (ddpm.py:329-350)
            self.register_buffer(name, torch.tensor(val, dtype=torch.float32))
        self.noise_source = None                      # optional callable(shape, device) -> N(0,1) tensor (parity tests)
        self._graph = None

    # ---- helpers
    def _tables(self):
        return {k: getattr(self, k) for k in ("sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
                                              "posterior_mean_coef1", "posterior_mean_coef2",
                                              "posterior_log_variance_clipped")}

    def _randn(self, shape, device):
        if self.noise_source is not None:
            return self.noise_source(shape, device)
        return torch.randn(shape, device=device)

    def _is_native(self):
        return isinstance(self.denoise_fn, Unet)

    # ---- q(x_t | x_0)
    def q_mean_variance(self, x_start, t):
        mean = extract(self.sqrt_alphas_cumprod, t, x_start.shape) * x_start
        variance = extract(1.0 - self.alphas_cumprod, t, x_start.shape)
        log_variance = extract(self.log_one_minus_alphas_cumprod, t, x_start.shape)
        return mean, variance, log_variance

    def q_sample(self, x_start, t, noise=None):
        if noise is None:
            noise = torch.randn_like(x_start)
        _, xt = K.q_sample(x_start.float(), noise.float(), t, self.sqrt_alphas_cumprod,
                           self.sqrt_one_minus_alphas_cumprod, want_nchw=True)
        return xt

    def predict_start_from_noise(self, x_t, t, noise):
        return (extract(self.sqrt_recip_alphas_cumprod, t, x_t.shape) * x_t
                - extract(self.sqrt_recipm1_alphas_cumprod, t, x_t.shape) * noise)

    def q_posterior(self, x_start, x_t, t):
        mean = (extract(self.posterior_mean_coef1, t, x_t.shape) * x_start
                + extract(self.posterior_mean_coef2, t, x_t.shape) * x_t)
        return mean, extract(self.posterior_variance, t, x_t.shape), extract(self.posterior_log_variance_clipped, t, x_t.shape)

    def p_mean_variance(self, x, t, clip_denoised: bool):
        x_recon = self.predict_start_from_noise(x, t=t, noise=self.denoise_fn(x, t))
        if clip_denoised:
            x_recon.clamp_(-1.0, 1.0)
        return self.q_posterior(x_start=x_recon, x_t=x, t=t)

    # ---- reverse process
    @torch.no_grad()
    def p_sample(self, x, t, clip_denoised=True, repeat_noise=False):
        """One reverse step (ddpm.py:390-397): UNet + fused posterior update."""
        x = x.float().contiguous()
        eps, _ = self.denoise_fn.forward_nhwc(K.nchw_to_nhwc(x), t, record=False)
        if repeat_noise:
            z = self._randn((1, *x.shape[1:]), x.device).repeat(x.shape[0], 1, 1, 1)
        else:
            z = self._randn(x.shape, x.device)
        xp, _ = K.p_sample_update(x, eps, z, t, self._tables(), clip=clip_denoised, want_nhwc=False)
        return xp

    @torch.no_grad()
    def p_sample_loop(self, shape, use_graph: Optional[bool] = None):
        """x_T ~ N(0, I); T sequential denoise steps (ddpm.py:399-409).  With `use_graph` the
        denoise iteration is captured once in a hipGraph and replayed T times (device-side t)."""
        device = self.betas.device
        from ..runtime.sampler import GraphSampler
        if use_graph is None:
            use_graph = self.noise_source is None and os.environ.get("MI_DDPM_GRAPH", "1") == "1"
        if use_graph:
            # a captured denoise step bakes in the kernel picks: the numeric mode, the storage of the internal tensors, the fusion
            net = self.denoise_fn
            key = (tuple(shape), getattr(net, "compute_mode", None), getattr(net, "block_storage", None), str(getattr(net, "fuse_gn_conv", None)))
            if self._graph is None or getattr(self, "_graph_key", None) != key:
                self._graph = GraphSampler(self, tuple(shape))
                self._graph_key = key
            return self._graph.run()
        b = shape[0]
        img = self._randn(shape, device)
        for i in reversed(range(self.num_timesteps)):
            img = self.p_sample(img, torch.full((b,), i, device=device, dtype=torch.long))
        return img

    @torch.no_grad()
    def sample(self, batch_size=16):
        return self.p_sample_loop((batch_size, self.channels, *self.image_size))

    @torch.no_grad()
    def interpolate(self, x1, x2, t=None, weight=0.5):
        b, device = x1.shape[0], x1.device
        t = self.num_timesteps - 1 if t is None else t
        assert x1.shape == x2.shape
        tb = torch.full((b,), t, device=device, dtype=torch.long)
        img = (1 - weight) * self.q_sample(x1, tb) + weight * self.q_sample(x2, tb)
        for i in reversed(range(0, t)):
            img = self.p_sample(img, torch.full((b,), i, device=device, dtype=torch.long))
        return img

    # ---- training objective
    def p_losses(self, x_start, t, noise=None):
        """L1/L2 between the noise and the UNet's prediction at x_t (ddpm.py:446-460)."""
        if noise is None:
            noise = torch.randn_like(x_start)
        net = self.denoise_fn
        if net._anchor.device != x_start.device:
            object.__setattr__(net, "_anchor", torch.zeros(1, device=x_start.device, requires_grad=True))
        anchor = net._anchor if (torch.is_grad_enabled() and net.training) else net._anchor.detach()
        return _LossFunction.apply(x_start.float().contiguous(), t, noise.float().contiguous(), anchor, self)

    def forward(self, x, *args, **kwargs):
        b = x.shape[0]
        t = torch.randint(0, self.num_timesteps, (b,), device=x.device).long()     # t first, then eps (RNG order)
        return self.p_losses(x, t, *args, **kwargs)

    @torch.no_grad()
    def loss_and_backward(self, x):
        """forward(x) followed by loss.backward(), without autograd: the same draws (t, then eps), the same kernels in the same
        order, the flat gradient buffer written by the UNet's tape replay -- all on the CALLING thread (autograd runs a CUDA node's
        backward on its device worker thread), which is what lets the data-parallel graph step cut its stream capture into segments
        from inside the gradient-ready hook (src/runtime/graphed.py).  -> the loss as a device scalar."""
        net = self.denoise_fn
        b = x.shape[0]
        t = torch.randint(0, self.num_timesteps, (b,), device=x.device).long()
        x0 = x.float().contiguous()
        noise = torch.randn_like(x0)
        xt, _ = K.q_sample(x0, noise, t, self.sqrt_alphas_cumprod, self.sqrt_one_minus_alphas_cumprod)
        eps, tape = net.forward_nhwc(xt, t, record=True)
        loss, dpred = K.eps_loss(eps, noise, 0 if self.loss_type == "l1" else 1, want_grad=True)
        net.backward_nhwc(tape, dpred, need_dx=False)
        return loss


# --------------------------------------------------------------------------------------------
# LightningModule-shaped wrapper
# --------------------------------------------------------------------------------------------
class DDPM(BaseModel):
    """Reference `DDPM` (ddpm.py:469-521): same ctor and hooks; `configure_optimizers` returns the
    fused flat-buffer Adam (same update rule as torch.optim.Adam)."""

    def __init__(self, datamodule, hidden_dim: int = 64, timesteps: int = 1000, loss_type: str = "l1",
                 dim_mults: Sequence[int] = (1, 2, 4, 8), lr: float = 0.0002, b1: float = 0.5, b2: float = 0.999,
                 optim="adam", **kwargs):
        super().__init__(datamodule)
        self.save_hyperparameters()
        self.denoising_model = Unet(dim=hidden_dim, channels=self.channels, dim_mults=tuple(dim_mults))
        self.diffusion_model = GaussianDiffusion(
            self.denoising_model, image_size=(self.height, self.width), timesteps=timesteps,
            loss_type=loss_type, channels=self.channels)

    def training_step(self, batch, batch_idx):
        imgs, _ = batch
        loss = self.diffusion_model(imgs)
        self.log("train_loss/loss", loss)          # logged as a device scalar: no per-step host sync
        return loss

    def training_step_and_backward(self, batch, batch_idx):
        """training_step + loss.backward() in one autograd-free call (GaussianDiffusion.loss_and_backward)."""
        imgs, _ = batch
        loss = self.diffusion_model.loss_and_backward(imgs)
        self.log("train_loss/loss", loss)
        return loss

    def configure_optimizers(self):
        from ..runtime.optim import FlatAdam
        hp = self.hparams
        return FlatAdam(self.denoising_model, lr=hp.lr, betas=(hp.b1, hp.b2))

    def validation_step(self, batch, batch_idx):
        imgs, labels = batch
        n = imgs.shape[0]
        t_last = torch.full((n,), self.hparams.timesteps - 1, device=imgs.device, dtype=torch.long)
        diffusion_imgs = self.diffusion_model.q_sample(imgs, t=t_last)
        fake_imgs = self.diffusion_model.sample(64) if batch_idx == 0 else None
        return ValidationResult(real_image=imgs, fake_image=fake_imgs, others={"diffusion": diffusion_imgs})
