"""Drop-in boundary: ``build_model(state_dict, design_details)`` and the attribute surface the reference's trainers
consume from it (reference clip/model.py:656-699).

The returned object is an ``nn.Module`` whose parameters carry the OpenAI checkpoint key names and the reference's
dtype policy (``convert_weights``, clip/model.py:632-653), so ``state_dict()`` / ``load_state_dict`` / ``.float()`` /
``.to(device)`` / ``named_parameters()`` behave as trainers expect -- but no forward arithmetic runs in torch: every
``forward`` launches the HIP towers of ``libclipmi.so`` through the C ABI.  torch owns memory and streams only.
Running any forward on a CPU tensor raises; there is no fallback.
"""
from __future__ import annotations

import ctypes as C
import threading
import weakref
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import F16, F32, check, lib
from .resnet import ModifiedResNet
from .synthetic import ClipGeometry, geometry_from_state_dict, resnet_config_from_state_dict

_DT = {torch.float16: F16, torch.float32: F32}


# --------------------------------------------------------------------------------------------------------------------
# parameter holders with the checkpoint's names
# --------------------------------------------------------------------------------------------------------------------
class LayerNorm(nn.Module):
    """Callable LN with fp32 statistics whatever the activation dtype (reference clip/model.py:153-159)."""

    def __init__(self, width: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(width))
        self.bias = nn.Parameter(torch.zeros(width))
        self.eps = 1e-5

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.layernorm(x, self.weight.detach().float(), self.bias.detach().float(), self.eps)


class _Linear(nn.Module):
    def __init__(self, n_in: int, n_out: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(n_out, n_in))
        self.bias = nn.Parameter(torch.zeros(n_out))


class _Attn(nn.Module):
    def __init__(self, width: int):
        super().__init__()
        self.in_proj_weight = nn.Parameter(torch.empty(3 * width, width))
        self.in_proj_bias = nn.Parameter(torch.zeros(3 * width))
        self.out_proj = _Linear(width, width)


class _Mlp(nn.Module):
    def __init__(self, width: int):
        super().__init__()
        self.c_fc = _Linear(width, 4 * width)
        self.c_proj = _Linear(4 * width, width)


class _Block(nn.Module):
    """Parameters of one ResidualAttentionBlock (clip/model.py:167-188); the arithmetic is in csrc/capi.hip run_block."""

    def __init__(self, width: int):
        super().__init__()
        self.attn = _Attn(width)
        self.ln_1 = LayerNorm(width)
        self.mlp = _Mlp(width)
        self.ln_2 = LayerNorm(width)


class _Conv(nn.Module):
    def __init__(self, width: int, patch: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(width, 3, patch, patch))


_FP16_SUFFIXES = ("attn.in_proj_weight", "attn.in_proj_bias", "out_proj.weight", "out_proj.bias", "c_fc.weight",
                  "c_fc.bias", "c_proj.weight", "c_proj.bias", "conv1.weight",
                  # ModifiedResNet: every Conv2d and the attention pool's Linears (clip/model.py:636-639)
                  "conv2.weight", "conv3.weight", "downsample.0.weight", "q_proj.weight", "q_proj.bias", "k_proj.weight",
                  "k_proj.bias", "v_proj.weight", "v_proj.bias")
_FP16_NAMES = ("visual.proj", "text_projection")


def _policy_dtype(name: str) -> torch.dtype:
    """convert_weights (clip/model.py:632-653): fp16 for Conv/Linear/MHA weights+biases and the two projections,
    fp32 for LayerNorm, class/positional/token embeddings and logit_scale."""
    if name in _FP16_NAMES or name.endswith(_FP16_SUFFIXES):
        return torch.float16
    return torch.float32


class TextTransformer(nn.Module):
    """``clip_model.transformer``: callable on LND activations (coop.py:58-60) or on MaPLe's ``[x, deep_prompts, counter]``
    list (maple.py:64-66); causal mask inside (clip/model.py:585-591).

    ``live_rows`` (opt-in, Level 1): the blocks return every token row, so they run every row -- they cannot know where the caller's
    prompts end.  A caller that reads only the EOT rows afterwards (every ``TextEncoder`` of the reference does: coop.py:65, maple.py:72) may
    say so with ONE line where it holds the tokenised prompts, e.g. in ``CustomCLIP.__init__``::

        clip_model.transformer.live_rows = tokenized_prompts          # or an int; None (default) = every row

    Then only the rows up to the last prompt's EOT are computed (include/clipmi.h ``seq_rows``; the causal mask makes the rows behind them
    irrelevant to any row in front), those rows come back bit for bit as before, and the rows behind come back as ZEROS."""

    live_rows = None

    def __init__(self, owner: "CLIP", width: int, layers: int):
        super().__init__()
        self.width, self.layers = width, layers
        self.resblocks = nn.ModuleList([_Block(width) for _ in range(layers)])
        object.__setattr__(self, "_owner", owner)  # not a sub-module (avoids a reference cycle in module traversal)

    def _rows(self, n_ctx: int) -> int:
        hint, owner = self.live_rows, self._owner
        if hint is None:
            return 0
        if isinstance(hint, torch.Tensor):
            return owner.live_rows(hint, n_ctx)
        return max(int(hint), 1 + int(n_ctx))

    def forward(self, x):
        owner: CLIP = self._owner
        if isinstance(x, (list, tuple)):
            xt, deep, counter = x
            n_ctx = owner.design_details.get("maple_length", deep[0].shape[0] if len(deep) else 0)
            y = owner._text_blocks(xt, list(deep), n_ctx, rows=self._rows(n_ctx if len(deep) else 0))
            used = min(len(deep), self.layers - 1)
            return [y, deep, counter + used]
        deep, n_ctx = owner.ivlp_text_prompts()
        return owner._text_blocks(x, deep, n_ctx, rows=self._rows(n_ctx if deep else 0))


class VisionTransformer(nn.Module):
    """``clip_model.visual``: image [B,3,R,R] -> [B,E] (clip/model.py:394-424); with (shared_ctx, deep_prompts) the MaPLe
    variant (clip/model.py:447-478)."""

    def __init__(self, owner: "CLIP", g: ClipGeometry):
        super().__init__()
        self.input_resolution = g.image_resolution
        self.output_dim = g.embed_dim
        self.conv1 = _Conv(g.vision_width, g.vision_patch_size)
        self.class_embedding = nn.Parameter(torch.empty(g.vision_width))
        self.positional_embedding = nn.Parameter(torch.empty(g.vision_tokens, g.vision_width))
        self.ln_pre = LayerNorm(g.vision_width)
        self.transformer = nn.Module()
        self.transformer.resblocks = nn.ModuleList([_Block(g.vision_width) for _ in range(g.vision_layers)])
        self.transformer.width, self.transformer.layers = g.vision_width, g.vision_layers
        self.ln_post = LayerNorm(g.vision_width)
        self.proj = nn.Parameter(torch.empty(g.vision_width, g.embed_dim))
        object.__setattr__(self, "_owner", owner)

    def forward(self, x: torch.Tensor, shared_ctx: Optional[torch.Tensor] = None,
                compound_deeper_prompts: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
        owner: CLIP = self._owner
        return owner.image_features_f32(x, shared_ctx, compound_deeper_prompts).to(owner.dtype)


# --------------------------------------------------------------------------------------------------------------------
class CLIP(nn.Module):
    """Attribute surface of the reference ``CLIP`` (clip/model.py:481-629), HIP inside."""

    def __init__(self, geom: ClipGeometry, design_details: Optional[dict] = None, resnet: Optional[dict] = None):
        super().__init__()
        self.geometry = geom
        self.design_details = dict(design_details or {"trainer": "CoOp"})
        self.context_length = geom.context_length
        self.vocab_size = geom.vocab_size
        self.is_resnet = resnet is not None
        if resnet is not None:      # clip/model.py:502-510: vision_layers is a tuple -> ModifiedResNet, heads = width * 32 // 64
            self.visual = ModifiedResNet(resnet["layers"], geom.embed_dim, resnet["width"] * 32 // 64, geom.image_resolution, resnet["width"])
        else:
            self.visual = VisionTransformer(self, geom)
        self.transformer = TextTransformer(self, geom.transformer_width, geom.transformer_layers)
        self.token_embedding = nn.Embedding(geom.vocab_size, geom.transformer_width)
        self.positional_embedding = nn.Parameter(torch.empty(geom.context_length, geom.transformer_width))
        self.ln_final = LayerNorm(geom.transformer_width)
        self.text_projection = nn.Parameter(torch.empty(geom.transformer_width, geom.embed_dim))
        self.logit_scale = nn.Parameter(torch.ones([]) * float(np.log(1 / 0.07)))
        self._init_ivlp_prompts()
        self._resnet_cfg = dict(resnet) if resnet is not None else None
        self._handle: Optional[int] = None
        self._bound = None          # keeps the packed tensors + ctypes arrays alive
        self._ws: Dict[tuple, torch.Tensor] = {}   # (kind, stream) -> workspace, LRU-bounded (_workspace)
        self._options: Dict[str, int] = {}         # per-handle settings, replayed on the per-device copies
        self._device_copies: Dict[int, "CLIP"] = {}   # device index -> resident copy (see _resident)
        self._live_rows: Dict[tuple, tuple] = {}      # tokenised prompt set -> (the tensor, rows the text tower must compute)
        self._copies_lock = threading.Lock()
        # Held while ONE tower call enqueues its launches: two host threads that drive this model on the SAME stream (they then share the
        # stream's workspace) must not interleave their launch sequences; on different streams the workspaces differ and the lock costs a few us.
        self._launch_lock = threading.Lock()

    # ---- IVLP / VPT design (clip/model.py:191-256, 334-346, 361-381) -------------------------------------------
    def _init_ivlp_prompts(self) -> None:
        """Per-layer prompt tokens owned by the model itself, under the reference's parameter names: ``visual.VPT``
        (appended after the positional embedding), ``visual.transformer.resblocks.{i}.VPT_shallow`` and
        ``transformer.resblocks.{i}.VPT_shallow`` for 1 <= i < depth (block 0 never carries one, model.py:209-226)."""
        dd, g = self.design_details, self.geometry
        if dd.get("trainer") not in ("IVLP", "VPT"):
            return
        v_depth, v_ctx = int(dd.get("vision_depth", 0)), int(dd.get("vision_ctx", 0))
        t_depth, t_ctx = int(dd.get("language_depth", 0)), int(dd.get("language_ctx", 0))

        def tokens(n, width):
            return nn.Parameter(torch.empty(n, width).normal_(std=0.02))
        if v_depth > 0:
            self.visual.VPT = tokens(v_ctx, g.vision_width)
            for i in range(1, min(v_depth, g.vision_layers)):
                self.visual.transformer.resblocks[i].VPT_shallow = tokens(v_ctx, g.vision_width)
        for i in range(1, min(t_depth, g.transformer_layers)):
            self.transformer.resblocks[i].VPT_shallow = tokens(t_ctx, g.transformer_width)

    def ivlp_vision_prompts(self):
        """(shallow, deep list) for the image tower, (None, None) for every other design."""
        if self.is_resnet:
            return None, None
        vpt = getattr(self.visual, "VPT", None)
        if vpt is None:
            return None, None
        return vpt, [b.VPT_shallow for b in self.visual.transformer.resblocks[1:] if hasattr(b, "VPT_shallow")]

    def ivlp_text_prompts(self):
        """(deep list, n_ctx) for the text tower, (None, 0) when the design has none."""
        deep = [b.VPT_shallow for b in self.transformer.resblocks[1:] if hasattr(b, "VPT_shallow")]
        return (deep, deep[0].shape[0]) if deep else (None, 0)

    # ---- nn.Module protocol -----------------------------------------------------------------------------------
    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._bound = None          # tensors moved / re-typed: re-pack lazily
        self._ws = {}
        self._device_copies = {}
        return out

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        res = super().load_state_dict(state_dict, strict=strict, **kw)
        self.rebind()
        return res

    def rebind(self) -> None:
        """Call after modifying tower parameters in place (the packed fp16/fp32 copies are otherwise reused)."""
        self._bound = None
        self._device_copies = {}
        if self.is_resnet:          # the ModifiedResNet tower keeps its own folded operands (resnet.py)
            self.visual._packed, self.visual._packed_elsewhere = None, {}

    def __getstate__(self):
        """copy.deepcopy / pickle: the C handle, the packed operands, the workspaces and the per-device copies belong to THIS object
        (``__del__`` destroys the handle); a copy binds its own lazily."""
        state = dict(self.__dict__)
        state.update(_handle=None, _bound=None, _ws={}, _device_copies={}, _copies_lock=None, _launch_lock=None, _live_rows={})
        state.pop("_origin", None)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self._copies_lock, self._launch_lock = threading.Lock(), threading.Lock()
        for tower in (self.visual, self.transformer):      # the towers of a copy call into the copy
            if "_owner" in tower.__dict__:
                object.__setattr__(tower, "_owner", self)

    # ---- several GPUs in one process: nn.DataParallel around the unchanged trainers -------------------------------
    # The reference wraps its models in nn.DataParallel whenever torch.cuda.device_count() > 1 (trainers/classification/coop.py:268-272:
    # the TextEncoder; trainers/calibration/tempscaling.py:117-120: the whole calibration model).  DataParallel clones the module tree per
    # device on EVERY forward (``_replicate_for_data_parallel``: a shallow ``__dict__`` copy, parameters re-broadcast) and runs the clones
    # in threads under ``torch.cuda.device(k)``.  A clone of a tower proxy therefore still points at the model that owns the C handle,
    # whose operands live on ITS device.  The owner answers a call whose activations live on another GPU from a copy of itself that is
    # resident there: built once per device (weights, packed operands, handle, workspaces), kept until the owner's weights are re-bound --
    # not re-sent per call as DataParallel's own broadcast is.  One process per GPU (torchrun, INTEGRATION.md "Multi-GPU") stays the
    # recommended deployment; this makes the unchanged single-process callers correct instead of silently reading cuda:0's weights.
    def _replicate_for_data_parallel(self):
        replica = super()._replicate_for_data_parallel()
        replica.__dict__.update(_handle=None, _bound=None, _ws={}, _device_copies={}, _live_rows={})   # never shares (or frees) the owner's handle
        object.__setattr__(replica, "_origin", self._home())
        return replica

    def _home(self) -> "CLIP":
        return self.__dict__.get("_origin") or self

    def _copy_to(self, device: torch.device) -> "CLIP":
        """A model of the same geometry, design and per-handle options with this model's weights copied to ``device``."""
        with torch.no_grad():
            twin = CLIP(self.geometry, self.design_details, self._resnet_cfg)
            src = dict(self.named_parameters())
            src.update(dict(self.named_buffers()))
            for name, t in list(twin.named_parameters()) + list(twin.named_buffers()):
                t.data = src[name].detach().to(device, copy=True)
            for p in twin.parameters():
                p.requires_grad_(False)
        twin.eval()
        for name, value in self._options.items():
            twin.set_option(name, value)
        return twin

    def _resident(self, device: torch.device) -> "CLIP":
        """The model that owns the weights on ``device``: the owner itself, or its resident copy there."""
        home = self._home()
        if device == home.device or device.type != "cuda" or home.device.type != "cuda":   # (a CPU tensor / model is refused further down)
            return home
        sig = home._param_versions()
        with home._copies_lock:
            twin = home._device_copies.get(device.index)
            seen = home.__dict__.setdefault("_copy_versions", {})
            if twin is not None and sig is not None and seen.get(device.index) != sig:
                twin = None                  # a parameter of the owner was updated in place (IVLP / VPT prompts under training): a copy is a snapshot
            if twin is None:
                with torch.cuda.device(device):
                    twin = home._copy_to(device)
                home._device_copies[device.index] = twin
                seen[device.index] = sig
        return twin

    def _param_versions(self):
        """Version counters of every parameter and buffer (in-place updates move them); None when some tensor has none (made under inference_mode)."""
        try:
            return tuple(t._version for t in list(self.parameters()) + list(self.buffers()))
        except RuntimeError:
            return None

    def _elsewhere(self, t) -> Optional["CLIP"]:
        """The resident copy a call on tensor ``t`` must run on, None when this model is the right one."""
        if not isinstance(t, torch.Tensor) or not t.is_cuda:
            return None
        m = self._resident(t.device)
        return None if m is self else m

    def __del__(self):
        try:
            if self._handle is not None:
                lib.clipmi_destroy(self._handle)
        except Exception:
            pass

    @property
    def dtype(self) -> torch.dtype:
        return self.visual.conv1.weight.dtype

    @property
    def device(self) -> torch.device:
        return self.visual.conv1.weight.device

    # ---- weight binding ---------------------------------------------------------------------------------------
    def _ensure_handle(self):
        if self._handle is None:
            g = self.geometry
            geo = _lib.Geometry(g.embed_dim, g.image_resolution, g.vision_patch_size, g.vision_width, g.vision_layers,
                                g.context_length, g.vocab_size, g.transformer_width, g.transformer_layers,
                                g.transformer_heads)
            h = C.c_void_p()
            check(lib.clipmi_create(C.byref(geo), C.byref(h)), "clipmi_create")
            self._handle = h.value

    def _ensure_bound(self):
        if self._bound is not None:
            return
        with self._launch_lock:             # two threads' first calls: one packs, the other finds it done
            if self._bound is None:
                self._bind()

    def _bind(self):
        dev = self.device
        if dev.type != "cuda":
            raise RuntimeError("clipmi: the model must be on a ROCm GPU before it is run (model.to('cuda')); "
                               "there is no CPU path")
        g = self.geometry
        self._ensure_handle()
        keep: List[torch.Tensor] = []

        def f16(t):  # GEMM operand: fp16, contiguous (aliases the parameter when it already is)
            t = t.detach().to(torch.float16).contiguous()
            keep.append(t)
            return t.data_ptr()

        def f32(t):
            t = t.detach().to(torch.float32).contiguous()
            keep.append(t)
            return t.data_ptr()

        def fold(lin_w, lin_b, ln):
            """LayerNorm folded into the following Linear (csrc/gemm.hip "LayerNorm folded into the GEMMs"):
            w_f = fp16(gamma * W), g = row sums of that fp16 w_f (what the MFMA multiplies), c = W beta + b."""
            w32 = lin_w.detach().float()
            wf = (w32 * ln.weight.detach().float()[None, :]).to(torch.float16).contiguous()
            gsum = wf.float().sum(dim=1).contiguous()
            c = (w32 @ ln.bias.detach().float() + lin_b.detach().float()).contiguous()
            keep.extend([wf, gsum, c])
            return wf.data_ptr(), gsum.data_ptr(), c.data_ptr()

        def blocks(mods) -> "C.Array":
            arr = (_lib.BlockWeights * len(mods))()
            for i, b in enumerate(mods):
                arr[i] = _lib.BlockWeights(
                    f32(b.ln_1.weight), f32(b.ln_1.bias), f16(b.attn.in_proj_weight), f32(b.attn.in_proj_bias),
                    f16(b.attn.out_proj.weight), f32(b.attn.out_proj.bias), f32(b.ln_2.weight), f32(b.ln_2.bias),
                    f16(b.mlp.c_fc.weight), f32(b.mlp.c_fc.bias), f16(b.mlp.c_proj.weight), f32(b.mlp.c_proj.bias),
                    *fold(b.attn.in_proj_weight, b.attn.in_proj_bias, b.ln_1),
                    *fold(b.mlp.c_fc.weight, b.mlp.c_fc.bias, b.ln_2))
            return arr

        v = self.visual
        vb = None
        if not self.is_resnet:      # the ModifiedResNet tower packs its own operands (resnet.py) and runs op by op
            k = 3 * g.vision_patch_size ** 2
            kpad = (k + 63) // 64 * 64
            conv = torch.zeros(g.vision_width, kpad, dtype=torch.float16, device=dev)
            conv[:, :k] = v.conv1.weight.detach().reshape(g.vision_width, k).to(torch.float16)
            keep.append(conv)
            vb = blocks(v.transformer.resblocks)
            vw = _lib.VisionWeights(conv.data_ptr(), f32(v.class_embedding), f32(v.positional_embedding),
                                    f32(v.ln_pre.weight), f32(v.ln_pre.bias), f32(v.ln_post.weight), f32(v.ln_post.bias),
                                    f16(v.proj.detach().t()), vb)
            check(lib.clipmi_set_vision_weights(self._handle, C.byref(vw)), "clipmi_set_vision_weights")
        tb = blocks(self.transformer.resblocks)
        tw = _lib.TextWeights(f32(self.token_embedding.weight), f32(self.positional_embedding),
                              f32(self.ln_final.weight), f32(self.ln_final.bias),
                              f16(self.text_projection.detach().t()), tb)
        check(lib.clipmi_set_text_weights(self._handle, C.byref(tw)), "clipmi_set_text_weights")
        self._bound = (keep, vb, tb)

    # ---- per-model settings (include/clipmi.h, clipmi_model_set_option) ------------------------------------------
    _MODEL_OPTIONS = ("residual_f16", "ln_fold", "cls_only_last_block")

    def set_option(self, name: str, value: int) -> None:
        """Precision / variant selection of THIS model: the reference picks precision per model (cfg.TRAINER.<X>.PREC,
        trainers/classification/coop.py:243-245) and builds a second CLIP in the same process (base_learner.py:262-272).
        -1 = follow the process-wide default again."""
        if name not in self._MODEL_OPTIONS:
            raise KeyError(f"unknown model option {name!r}; one of {self._MODEL_OPTIONS}")
        self._ensure_handle()
        check(lib.clipmi_model_set_option(self._handle, name.encode(), int(value)), "clipmi_model_set_option")
        self._options[name] = int(value)
        for twin in list(self._device_copies.values()):
            twin.set_option(name, value)

    def get_option(self, name: str) -> int:
        self._ensure_handle()
        v = C.c_int(0)
        check(lib.clipmi_model_get_option(self._handle, name.encode(), C.byref(v)), "clipmi_model_get_option")
        return v.value

    def _workspace(self, kind: str, nbytes: int) -> torch.Tensor:
        """Tower workspace of the CURRENT stream: launches on one stream run in order and may share a buffer; two tower calls in flight on
        different streams (batches pipelined over two streams, the text tower beside the image tower) must not.
        Eviction (more than 8 (kind, stream) keys) drops a buffer that may still have a pass in flight on ITS stream.  That is safe because of
        one property of torch's caching allocator this code relies on: a freed block is handed out again only to allocations made under the
        stream it was allocated under (other streams get it after an event recorded at the free has passed), and a workspace is always allocated
        under the stream that keys it -- so a re-use is stream-ordered behind the evicted buffer's last launch.  Cycling through more than 8
        keys re-allocates (possibly a hipMalloc) on the hot path: keep to a handful of streams per model."""
        key = (kind, torch.cuda.current_stream(self.device).cuda_stream if self.device.type == "cuda" else 0)
        ws = self._ws.pop(key, None)
        if ws is None or ws.numel() < nbytes or ws.device != self.device:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._ws[key] = ws                       # (re-inserted last: the dict is the LRU order)
        while len(self._ws) > 8:                 # short-lived streams must not pin workspaces for ever
            self._ws.pop(next(iter(self._ws)))
        return ws

    def _hook(self, n_ctx: int, shallow: Optional[torch.Tensor], deep: Optional[Sequence[torch.Tensor]], max_deep: int):
        """clipmi_prompt_hook; prompts go through .half() exactly as the reference does (clip/model.py:306,323,459)."""
        keep = []
        ps = pd = None
        if shallow is not None:
            s = shallow.detach().to(self.device).half().float().contiguous()
            keep.append(s)
            ps = s.data_ptr()
        deep = list(deep or [])[:max_deep]
        if deep:
            d = torch.stack([t.detach().to(self.device).half().float() for t in deep]).contiguous()
            keep.append(d)
            pd = d.data_ptr()
        return _lib.PromptHook(n_ctx, len(deep), ps, pd), keep

    # ---- towers -------------------------------------------------------------------------------------------------
    def image_features_f32(self, image: torch.Tensor, shared_ctx: Optional[torch.Tensor] = None,
                           deep_prompts: Optional[Sequence[torch.Tensor]] = None, flags: int = _lib.CALL_DEFAULT) -> torch.Tensor:
        """VisionTransformer.forward / ModifiedResNet.forward with fp32 output (un-normalised).  ``flags``: per-call
        stream precision (``_lib.CALL_STREAM_F32`` / ``_F16``); default = the model's setting."""
        twin = self._elsewhere(image)
        if twin is not None:
            with torch.cuda.device(image.device):
                return twin.image_features_f32(image, shared_ctx, deep_prompts, flags)
        if self.is_resnet:
            if shared_ctx is not None:
                raise ValueError("prompt tokens apply to the ViT towers only")
            return self.visual.features_f32(image)
        self._ensure_bound()
        g = self.geometry
        image = ops._dev(image, "image", (torch.float16, torch.float32))
        if image.dim() != 4 or tuple(image.shape[1:]) != (3, g.image_resolution, g.image_resolution):
            raise ValueError(f"encode_image: expected [B,3,{g.image_resolution},{g.image_resolution}], got {tuple(image.shape)}")
        B = image.shape[0]
        out = torch.empty(B, g.embed_dim, dtype=torch.float32, device=image.device)
        hook_ref, keep, n_ctx = None, None, 0
        if shared_ctx is None:                      # IVLP / VPT models carry their own prompt tokens
            shared_ctx, deep_prompts = self.ivlp_vision_prompts()
        if shared_ctx is not None:
            n_ctx = shared_ctx.shape[0]
            hook, keep = self._hook(n_ctx, shared_ctx, deep_prompts, g.vision_layers - 1)
            hook_ref = C.byref(hook)
        nbytes = lib.clipmi_vision_workspace_bytes(self._handle, B, n_ctx)
        with self._launch_lock:
            ws = self._workspace("vision", nbytes)
            check(lib.clipmi_encode_image(self._handle, image.data_ptr(), _DT[image.dtype], B, hook_ref, out.data_ptr(),
                                          ws.data_ptr(), ws.numel(), int(flags), ops._stream()), "clipmi_encode_image")
        return out

    def encode_image(self, image: torch.Tensor) -> torch.Tensor:
        """clip/model.py:597-598."""
        return self.visual(image.type(self.dtype))

    def _text_blocks(self, x_lnd: torch.Tensor, deep: Optional[List[torch.Tensor]], n_ctx: int, flags: int = _lib.CALL_DEFAULT, rows: int = 0) -> torch.Tensor:
        twin = self._elsewhere(x_lnd)
        if twin is not None:
            with torch.cuda.device(x_lnd.device):
                return twin._text_blocks(x_lnd, deep, n_ctx, flags, rows)
        self._ensure_bound()
        g = self.geometry
        x_lnd = ops._dev(x_lnd, "x", (torch.float16, torch.float32))
        if x_lnd.dim() != 3 or x_lnd.shape[0] != g.context_length or x_lnd.shape[2] != g.transformer_width:
            raise ValueError(f"transformer: expected LND [{g.context_length}, C, {g.transformer_width}], got {tuple(x_lnd.shape)}")
        Cn = x_lnd.shape[1]
        x = x_lnd.permute(1, 0, 2).contiguous()       # layout change only; the library is token-major
        y = torch.empty_like(x)
        hook_ref, keep = None, None
        if deep:
            hook, keep = self._hook(n_ctx, None, deep, g.transformer_layers - 1)
            hook_ref = C.byref(hook)
        rows = int(rows) if 0 < int(rows) < g.context_length else 0
        with self._launch_lock:
            ws = self._workspace("text", lib.clipmi_text_workspace_bytes(self._handle, Cn, rows))
            check(lib.clipmi_text_blocks(self._handle, x.data_ptr(), y.data_ptr(), _DT[x.dtype], Cn, rows, hook_ref, ws.data_ptr(),
                                         ws.numel(), int(flags), ops._stream()), "clipmi_text_blocks")
        return y.permute(1, 0, 2)

    # ---- dead-row elimination in the causal text tower (include/clipmi.h, clipmi_text_encoder `seq_rows`) ---------------------
    text_dead_row_elimination: bool = True   # False: always run all context_length rows of every prompt

    def live_rows(self, tokenized_prompts: torch.Tensor, n_ctx: int = 0) -> int:
        """Token rows per prompt the text tower has to compute for this set of tokenised prompts: the blocks mask causally
        (clip/model.py:585-591) and only the EOT row = ``argmax(ids)`` leaves the tower (clip/model.py:611, coop.py:65), so nothing behind the
        last prompt's EOT can reach an output.  ``max(EOT) + 1`` rounded up to a multiple of 8 (few distinct shapes), at least the prompt
        tokens 1..n_ctx a hook overwrites, at most the context.  Costs one read-back of a scalar per NEW prompt set: the integer is kept under
        the tensor's (storage, version, shape) with a WEAK reference to the tensor -- while that tensor lives its storage cannot be anybody else's --
        and re-used while its version counter stands.  Tensors made under ``torch.inference_mode()`` have no version counter: computed per call.
        (Contract: token ids edited IN PLACE through ``tensor.data`` or another route that bypasses autograd's version counter are not seen;
        pass a new tensor, or set ``text_dead_row_elimination = False``.  An EOT index at or beyond the bound is clamped by the library, as an
        index at or beyond the context always was.)"""
        L = self.context_length
        if not self.text_dead_row_elimination:
            return L
        t = tokenized_prompts
        try:
            version = t._version
        except RuntimeError:                         # a tensor made under torch.inference_mode() has no version counter: nothing to key a cached
            version = None                           # answer on -- the bound is computed per call (one scalar read-back)
        key = (t.data_ptr(), version, tuple(t.shape), tuple(t.stride()), str(t.device))
        hit = None
        if version is not None:
            with self._copies_lock:                  # (nn.DataParallel's clones ask the owner from their threads)
                hit = self._live_rows.pop(key, None)
            if hit is not None and hit[0]() is None:     # the tensor the entry was made for is gone: its storage may have been handed to another one
                hit = None
        if hit is None:
            last = int(t.reshape(-1, t.shape[-1]).argmax(dim=-1).max()) if t.numel() else 0
            hit = (weakref.ref(t), min(L, max((last + 1 + 7) // 8 * 8, 1)))
        if version is not None:
            with self._copies_lock:
                self._live_rows[key] = hit           # (re-inserted last: the dict is the LRU order); only the integer and a weak reference are kept
                while len(self._live_rows) > 16:
                    self._live_rows.pop(next(iter(self._live_rows)))
        return max(hit[1], min(L, 1 + int(n_ctx)))

    def text_encoder_f32(self, prompts: torch.Tensor, tokenized_prompts: torch.Tensor,
                         deep_prompts: Optional[Sequence[torch.Tensor]] = None, n_ctx: int = 0,
                         flags: int = _lib.CALL_DEFAULT, seq_rows: Optional[int] = None) -> torch.Tensor:
        """TextEncoder.forward fused (coop.py:56-67; maple.py:60-74): prompts [C,77,D] (no pos-emb) -> fp32 [C,E].
        ``flags``: per-call stream precision (CoCoOp's per-image passes ask for ``_lib.CALL_STREAM_F16``).  ``seq_rows``: the caller's own
        bound on the live token rows (``live_rows`` of the prompt set it tiles or repeats per call); None = derived from ``tokenized_prompts``."""
        twin = self._elsewhere(prompts)
        if twin is not None:
            with torch.cuda.device(prompts.device):
                return twin.text_encoder_f32(prompts, tokenized_prompts, deep_prompts, n_ctx, flags, seq_rows)
        self._ensure_bound()
        g = self.geometry
        prompts = ops._dev(prompts, "prompts", (torch.float16, torch.float32))
        if tuple(prompts.shape[1:]) != (g.context_length, g.transformer_width):
            raise ValueError(f"text_encoder: expected [C,{g.context_length},{g.transformer_width}], got {tuple(prompts.shape)}")
        Cn = prompts.shape[0]
        eot = tokenized_prompts.to(prompts.device).argmax(dim=-1).to(torch.int32).contiguous()  # index plumbing
        out = torch.empty(Cn, g.embed_dim, dtype=torch.float32, device=prompts.device)
        hook_ref, keep = None, None
        if not deep_prompts:
            deep_prompts, n_ctx_own = self.ivlp_text_prompts()
            n_ctx = n_ctx_own if deep_prompts else n_ctx
        if deep_prompts:
            hook, keep = self._hook(n_ctx, None, deep_prompts, g.transformer_layers - 1)
            hook_ref = C.byref(hook)
        rows = self.live_rows(tokenized_prompts, n_ctx if deep_prompts else 0) if seq_rows is None else int(seq_rows)
        if deep_prompts and 0 < rows < g.context_length:
            rows = max(rows, min(g.context_length, 1 + int(n_ctx)))      # a caller's bound never cuts the prompt tokens 1 .. n_ctx a hook overwrites
        with self._launch_lock:
            ws = self._workspace("text", lib.clipmi_text_workspace_bytes(self._handle, Cn, rows))
            check(lib.clipmi_text_encoder(self._handle, prompts.data_ptr(), _DT[prompts.dtype], eot.data_ptr(), Cn, rows, hook_ref,
                                          out.data_ptr(), ws.data_ptr(), ws.numel(), int(flags), ops._stream()), "clipmi_text_encoder")
        return out

    def text_features_f32(self, text: torch.Tensor, flags: int = _lib.CALL_DEFAULT) -> torch.Tensor:
        twin = self._elsewhere(text)
        if twin is not None:
            with torch.cuda.device(text.device):
                return twin.text_features_f32(text, flags)
        self._ensure_bound()
        g = self.geometry
        text = ops._dev(text, "text", (torch.int64,))
        if text.dim() != 2 or text.shape[1] != g.context_length:
            raise ValueError(f"encode_text: expected ids [C,{g.context_length}], got {tuple(text.shape)}")
        if self.ivlp_text_prompts()[0]:      # IVLP text blocks splice their own tokens: embeddings in, hook on
            return self.text_encoder_f32(self.token_embedding(text), text, flags=flags)
        Cn = text.shape[0]
        out = torch.empty(Cn, g.embed_dim, dtype=torch.float32, device=text.device)
        rows = self.live_rows(text)
        with self._launch_lock:
            ws = self._workspace("text", lib.clipmi_text_workspace_bytes(self._handle, Cn, rows))
            check(lib.clipmi_encode_text(self._handle, text.data_ptr(), Cn, rows, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                         int(flags), ops._stream()), "clipmi_encode_text")
        return out

    def encode_text(self, text: torch.Tensor) -> torch.Tensor:
        """clip/model.py:600-613."""
        return self.text_features_f32(text).to(self.dtype)

    def forward(self, image: torch.Tensor, text: torch.Tensor):
        """clip/model.py:615-629: (logits_per_image, logits_per_text)."""
        img_n = ops.l2_normalize(self.image_features_f32(image))
        txt_n = ops.l2_normalize(self.text_features_f32(text))
        scale = float(self.logit_scale.detach().exp())
        lpi, _, _ = ops.logits_fused(img_n, txt_n, scale, None, want_conf_pred=False)
        lpt, _, _ = ops.logits_fused(txt_n, img_n, scale, None, want_conf_pred=False)
        return lpi.to(self.dtype), lpt.to(self.dtype)

    BLOCK_KERNELS = ("in_proj", "attention", "out_proj", "c_fc", "c_proj")

    def profile_block_ms(self, batch: int, iters: int = 20, only: int = -1) -> Dict[str, float]:
        """Mean launch time (ms, hipEvents on the launch stream) of the five per-layer kernels of the image tower as
        the tower launches them (clipmi_profile_block).  Call ``image_features_f32`` on ``batch`` images first so that
        the workspace holds real activations."""
        self._ensure_bound()
        ws = self._workspace("vision", lib.clipmi_vision_workspace_bytes(self._handle, batch, 0))
        ms = (C.c_float * 5)()
        check(lib.clipmi_profile_block(self._handle, batch, iters, only, ws.data_ptr(), ws.numel(), ms, ops._stream()),
              "clipmi_profile_block")
        return {name: float(ms[i]) for i, name in enumerate(self.BLOCK_KERNELS)}

    def image_tower_launch_us(self, image: torch.Tensor) -> Dict[str, object]:
        """Device time of every launch of ONE real image-tower pass, each kernel in place behind its real predecessor
        (clipmi_encode_image_timed): {"embed": [us per embedding launch], "blocks": [[in_proj, attention, out_proj, c_fc, c_proj] per
        layer], "post": [ln_post, proj], "total_us": their sum, "features": the pass's fp32 features}."""
        self._ensure_bound()
        g = self.geometry
        image = ops._dev(image, "image", (torch.float16, torch.float32))
        B = image.shape[0]
        out = torch.empty(B, g.embed_dim, dtype=torch.float32, device=image.device)
        ws = self._workspace("vision", lib.clipmi_vision_workspace_bytes(self._handle, B, 0))
        cap = 16 + 5 * g.vision_layers + 2
        us = (C.c_float * cap)()
        n_pre = C.c_int(0)
        n = lib.clipmi_encode_image_timed(self._handle, image.data_ptr(), _DT[image.dtype], B, out.data_ptr(), ws.data_ptr(), ws.numel(),
                                          int(_lib.CALL_DEFAULT), us, cap, C.byref(n_pre), ops._stream())
        if n < 0:
            check(n, "clipmi_encode_image_timed")
        v = [float(us[i]) for i in range(n)]
        p = n_pre.value
        return {"embed": v[:p], "blocks": [v[p + 5 * i:p + 5 * i + 5] for i in range(g.vision_layers)], "post": v[n - 2:], "total_us": sum(v),
                "features": out}


def convert_weights(model: nn.Module) -> None:
    """Apply the reference dtype policy in place (clip/model.py:632-653)."""
    for name, p in model.named_parameters():
        p.data = p.data.to(_policy_dtype(name))


def build_model(state_dict: Dict[str, torch.Tensor], design_details: Optional[dict] = None) -> CLIP:
    """Same contract as the reference factory (clip/model.py:656-699): geometry from tensor shapes, fp16 weight
    conversion, strict load with a printed non-strict fallback, eval mode.  ViT towers, or a ModifiedResNet image tower
    when the checkpoint has no ``visual.proj`` (clip/model.py:659-672)."""
    resnet = None if "visual.proj" in state_dict else resnet_config_from_state_dict(state_dict)
    geom = geometry_from_state_dict(state_dict)
    model = CLIP(geom, design_details, resnet)
    for key in ("input_resolution", "context_length", "vocab_size"):
        if key in state_dict:
            del state_dict[key]
    convert_weights(model)
    try:
        model.load_state_dict(state_dict)
    except Exception:
        missing, _ = model.load_state_dict(state_dict, strict=False)
        print("Weights not found for some missing keys: ", missing)
    return model.eval()
