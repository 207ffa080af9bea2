"""ModifiedResNet image tower behind the reference's attribute surface (reference clip/model.py:10-150) -- SURVEY §8(f) f-4.

Parameter containers carry the checkpoint's names (``visual.conv1.weight``, ``visual.bn1.running_mean``,
``visual.layer3.2.downsample.0.weight``, ``visual.attnpool.q_proj.bias`` ...); none of their torch forwards is ever
called.  ``forward`` is a sequence of C-ABI launches on NHWC fp16 activations:

* BatchNorm (eval: running statistics) is folded into the preceding convolution at bind time:
  ``s = gamma / sqrt(var + eps)``, weights ``W * s`` rounded once to fp16, bias ``beta - mean * s`` in fp32;
* 1x1 convolutions are GEMMs on the ``[B*H*W, C]`` rows, 3x3 convolutions an im2col + GEMM, ReLU and the
  Bottleneck's residual add + ReLU are GEMM epilogues, the anti-aliasing ``AvgPool2d`` and the attention pool are
  small kernels (csrc/resnet.hip).
"""
from __future__ import annotations

import threading
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import check, lib

EPI_BIAS, EPI_BIAS_RELU, EPI_BIAS_RES16_RELU = 1, 4, 5


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.stride = stride
        self.downsample = None
        if stride > 1 or inplanes != planes * self.expansion:   # clip/model.py:33-40: avgpool, 1x1 conv, bn under keys -1 / 0 / 1
            self.downsample = nn.Sequential(OrderedDict([("-1", nn.AvgPool2d(stride)),
                                                         ("0", nn.Conv2d(inplanes, planes * self.expansion, 1, stride=1, bias=False)),
                                                         ("1", nn.BatchNorm2d(planes * self.expansion))]))


class AttentionPool2d(nn.Module):
    def __init__(self, spacial_dim: int, embed_dim: int, num_heads: int, output_dim: int):
        super().__init__()
        self.positional_embedding = nn.Parameter(torch.randn(spacial_dim ** 2 + 1, embed_dim) / embed_dim ** 0.5)
        self.k_proj = nn.Linear(embed_dim, embed_dim)
        self.q_proj = nn.Linear(embed_dim, embed_dim)
        self.v_proj = nn.Linear(embed_dim, embed_dim)
        self.c_proj = nn.Linear(embed_dim, output_dim)
        self.num_heads = num_heads


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


class ModifiedResNet(nn.Module):
    """``clip_model.visual`` for the RN backbones: image [B,3,R,R] -> [B, output_dim]."""

    def __init__(self, layers: Tuple[int, int, int, int], output_dim: int, heads: int, input_resolution: int = 224, width: int = 64):
        super().__init__()
        if width % 64:
            raise ValueError("ModifiedResNet: width must be a multiple of 64 (GEMM K granularity; the RN50/101/x4.. checkpoints are)")
        self.output_dim, self.input_resolution, self.width, self.layers_cfg = output_dim, input_resolution, width, tuple(layers)
        self.conv1 = nn.Conv2d(3, width // 2, 3, stride=2, padding=1, bias=False)
        self.bn1 = nn.BatchNorm2d(width // 2)
        self.conv2 = nn.Conv2d(width // 2, width // 2, 3, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(width // 2)
        self.conv3 = nn.Conv2d(width // 2, width, 3, padding=1, bias=False)
        self.bn3 = nn.BatchNorm2d(width)
        inplanes = width
        for li, (mult, n, stride) in enumerate(zip((1, 2, 4, 8), layers, (1, 2, 2, 2)), start=1):
            blocks = [Bottleneck(inplanes, width * mult, stride)]
            inplanes = width * mult * Bottleneck.expansion
            blocks += [Bottleneck(inplanes, width * mult) for _ in range(1, n)]
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        self.attnpool = AttentionPool2d(input_resolution // 32, width * 32, heads, output_dim)
        self._packed = None
        self._packed_elsewhere: Dict[int, dict] = {}    # device index -> the packed operands copied there (nn.DataParallel replicas)
        self._pack_lock = threading.Lock()

    # ---- binding: fold BatchNorm, pack weights ------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._packed, self._packed_elsewhere = None, {}
        return out

    def load_state_dict(self, *a, **k):
        res = super().load_state_dict(*a, **k)
        self._packed, self._packed_elsewhere = None, {}
        return res

    def __getstate__(self):
        state = dict(self.__dict__)
        state.update(_packed=None, _packed_elsewhere={}, _pack_lock=None)
        state.pop("_origin", None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self._pack_lock = threading.Lock()

    def _replicate_for_data_parallel(self):
        """nn.DataParallel clones this module per device on every forward (model.py "several GPUs in one process"): a clone takes the
        owner's packed operands from a per-device copy kept by the owner instead of re-folding its re-broadcast parameters per call."""
        replica = super()._replicate_for_data_parallel()
        replica.__dict__.update(_packed=None, _packed_elsewhere={})
        object.__setattr__(replica, "_origin", self.__dict__.get("_origin") or self)
        return replica

    def _packed_on(self, device: torch.device) -> dict:
        """The owner's packed operands on ``device`` (copied once per device, dropped when the owner re-packs)."""
        with self._pack_lock:                   # (several replica threads ask at once: the pack runs once)
            own = self._ensure_packed()
        if device == self.conv1.weight.device:
            return own

        def move(v):
            if isinstance(v, torch.Tensor):
                return v.to(device)
            if isinstance(v, (list, tuple)):
                return type(v)(move(e) for e in v)
            if isinstance(v, dict):
                return {k: move(e) for k, e in v.items()}
            return v
        with self._pack_lock:
            p = self._packed_elsewhere.get(device.index)
            if p is None:
                p = self._packed_elsewhere[device.index] = move(own)
        return p

    @staticmethod
    def _fold(conv: nn.Conv2d, bn: nn.BatchNorm2d, order: str, cin_pad: Optional[int] = None, cout_pad: Optional[int] = None):
        """(fp16 weight [Cout_p, Kpad], fp32 bias [Cout_p]) of conv followed by eval-mode bn.  ``order``: 'ckk' keeps the
        checkpoint's (c, ky, kx) column order (stem conv1, fed by the NCHW im2col), 'kkc' puts the tap first (NHWC operands).
        ``cin_pad`` / ``cout_pad`` add zero input / output channels (the stem's 32-channel activations are carried as 64 so that
        the implicit-GEMM convolution applies: a padded output channel is relu(0 + 0) = 0 and meets zero weights downstream)."""
        w = conv.weight.detach().float()
        s = bn.weight.detach().float() / torch.sqrt(bn.running_var.detach().float() + bn.eps)
        w = w * s[:, None, None, None]
        bias = bn.bias.detach().float() - bn.running_mean.detach().float() * s
        cout, cin = w.shape[0], w.shape[1]
        ci, co = cin_pad or cin, cout_pad or cout
        wp = torch.zeros(co, ci, w.shape[2], w.shape[3], dtype=torch.float32, device=w.device)
        wp[:cout, :cin] = w
        bp = torch.zeros(co, dtype=torch.float32, device=w.device)
        bp[:cout] = bias
        w2 = (wp.permute(0, 2, 3, 1) if order == "kkc" else wp).reshape(co, -1)
        k = w2.shape[1]
        kp = (k + 63) // 64 * 64
        out = torch.zeros(co, kp, dtype=torch.float16, device=w.device)
        out[:, :k] = w2.to(torch.float16)
        return out.contiguous(), bp.contiguous()

    def _ensure_packed(self):
        if self._packed is not None:
            return self._packed
        if self.conv1.weight.device.type != "cuda":
            raise RuntimeError("clipmi: the model must be on a ROCm GPU before it is run; there is no CPU path")
        p: Dict[str, object] = {}
        hp = _round_up(self.width // 2, 64)      # stem activations carried with zero channels up to a multiple of 64
        p["stem"] = [self._fold(self.conv1, self.bn1, "ckk", None, hp), self._fold(self.conv2, self.bn2, "kkc", hp, hp),
                     self._fold(self.conv3, self.bn3, "kkc", hp, None)]
        p["stem_c"] = hp
        blocks = []
        for li in range(1, 5):
            for b in getattr(self, f"layer{li}"):
                d = {"c1": self._fold(b.conv1, b.bn1, "kkc"), "c2": self._fold(b.conv2, b.bn2, "kkc"), "c3": self._fold(b.conv3, b.bn3, "kkc"),
                     "stride": b.stride, "down": None}
                if b.downsample is not None:
                    d["down"] = self._fold(b.downsample[1], b.downsample[2], "kkc")
                blocks.append(d)
        p["blocks"] = blocks
        ap = self.attnpool
        p["pos"] = ap.positional_embedding.detach().float().contiguous()
        p["wq"], p["bq"] = ap.q_proj.weight.detach().half().contiguous(), ap.q_proj.bias.detach().float().contiguous()
        p["wkv"] = torch.cat([ap.k_proj.weight.detach(), ap.v_proj.weight.detach()]).half().contiguous()
        p["bkv"] = torch.cat([ap.k_proj.bias.detach(), ap.v_proj.bias.detach()]).float().contiguous()
        p["wc"], p["bc"] = ap.c_proj.weight.detach().half().contiguous(), ap.c_proj.bias.detach().float().contiguous()
        self._packed = p
        return p

    # ---- launches ------------------------------------------------------------------------------------------------
    @staticmethod
    def _conv3x3(x: torch.Tensor, B: int, H: int, W: int, C: int, wb, epilogue=EPI_BIAS_RELU) -> torch.Tensor:
        w, bias = wb
        if C % 64 == 0 and w.shape[1] == 9 * C and w.shape[0] % 8 == 0:      # implicit GEMM: no im2col matrix
            out = torch.empty(B * H * W, w.shape[0], dtype=torch.float16, device=x.device)
            check(lib.clipmi_conv3x3_nhwc(x.data_ptr(), w.data_ptr(), bias.data_ptr(), out.data_ptr(), B, H, W, C, w.shape[0],
                                          1 if epilogue == EPI_BIAS_RELU else 0, ops._stream()), "clipmi_conv3x3_nhwc")
            return out
        col = torch.empty(B * H * W, w.shape[1], dtype=torch.float16, device=x.device)
        check(lib.clipmi_im2col3x3_nhwc(x.data_ptr(), col.data_ptr(), B, H, W, C, w.shape[1], ops._stream()), "clipmi_im2col3x3_nhwc")
        return ops.gemm_f16(col, w, bias, None, epilogue, torch.float16)

    @staticmethod
    def _avgpool(x: torch.Tensor, B: int, H: int, W: int, C: int, k: int) -> torch.Tensor:
        y = torch.empty(B * (H // k) * (W // k), C, dtype=torch.float16, device=x.device)
        check(lib.clipmi_avgpool_nhwc(x.data_ptr(), y.data_ptr(), B, H, W, C, k, ops._stream()), "clipmi_avgpool_nhwc")
        return y

    @torch.no_grad()
    def features_f32(self, image: torch.Tensor) -> torch.Tensor:
        """ModifiedResNet.forward (clip/model.py:136-150) with fp32 output."""
        origin = self.__dict__.get("_origin")
        if isinstance(image, torch.Tensor) and image.is_cuda and (origin is not None or image.device != self.conv1.weight.device):
            p = (origin or self)._packed_on(image.device)
        else:
            p = self._ensure_packed()
        image = ops._dev(image, "image", (torch.float16, torch.float32))
        R, w = self.input_resolution, self.width
        if image.dim() != 4 or tuple(image.shape[1:]) != (3, R, R):
            raise ValueError(f"encode_image: expected [B,3,{R},{R}], got {tuple(image.shape)}")
        B = image.shape[0]
        if B == 0:
            return torch.empty(0, self.output_dim, dtype=torch.float32, device=image.device)
        # stem: conv1 (stride 2) from NCHW, conv2, conv3, avgpool 2
        (w1, b1), c2, c3 = p["stem"]
        H = W = (R + 2 - 3) // 2 + 1
        col = torch.empty(B * H * W, w1.shape[1], dtype=torch.float16, device=image.device)
        check(lib.clipmi_im2col3x3_nchw(image.data_ptr(), _lib.F32 if image.dtype == torch.float32 else _lib.F16, col.data_ptr(), B, 3, R, R,
                                        2, w1.shape[1], ops._stream()), "clipmi_im2col3x3_nchw")
        x = ops.gemm_f16(col, w1, b1, None, EPI_BIAS_RELU, torch.float16)
        x = self._conv3x3(x, B, H, W, p["stem_c"], c2)
        x = self._conv3x3(x, B, H, W, p["stem_c"], c3)
        x = self._avgpool(x, B, H, W, w, 2)
        H, W, C = H // 2, W // 2, w
        # bottlenecks (clip/model.py:42-56)
        for blk in p["blocks"]:
            s = blk["stride"]
            planes = blk["c1"][0].shape[0]
            out = ops.gemm_f16(x, blk["c1"][0], blk["c1"][1], None, EPI_BIAS_RELU, torch.float16)
            out = self._conv3x3(out, B, H, W, planes, blk["c2"])
            identity = x
            if s > 1:
                out = self._avgpool(out, B, H, W, planes, s)
                identity = self._avgpool(x, B, H, W, C, s)
                H, W = H // s, W // s
            if blk["down"] is not None:
                identity = ops.gemm_f16(identity, blk["down"][0], blk["down"][1], None, EPI_BIAS, torch.float16)
            x = ops.gemm_f16(out, blk["c3"][0], blk["c3"][1], identity, EPI_BIAS_RES16_RELU, torch.float16)
            C = planes * Bottleneck.expansion
        # attention pool (clip/model.py:58-90)
        T, heads = H * W + 1, self.attnpool.num_heads
        tok = torch.empty(B * T, C, dtype=torch.float16, device=x.device)
        check(lib.clipmi_attnpool_tokens(x.data_ptr(), p["pos"].data_ptr(), tok.data_ptr(), B, H * W, C, ops._stream()), "clipmi_attnpool_tokens")
        kv = ops.gemm_f16(tok, p["wkv"], p["bkv"], None, EPI_BIAS, torch.float16)
        q = ops.gemm_f16(tok.view(B, T, C)[:, 0, :].contiguous(), p["wq"], p["bq"], None, EPI_BIAS, torch.float16)
        att = torch.empty(B, C, dtype=torch.float16, device=x.device)
        check(lib.clipmi_attnpool(q.data_ptr(), kv.data_ptr(), att.data_ptr(), B, T, heads, ops._stream()), "clipmi_attnpool")
        return ops.gemm_f16(att, p["wc"], p["bc"], None, EPI_BIAS, torch.float32)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.features_f32(x).to(self.conv1.weight.dtype)
