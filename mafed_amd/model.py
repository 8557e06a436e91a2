"""VLPythia (image-patch prefix + GPT-NeoX decoder) on hand-written gfx950 kernels.

Mirrors the reference model surface for the training hot path:
  * ``model_architecture["vlpythia"]`` registry (mafed/model/__init__.py:3-5)
  * state-dict names/shapes of ``VLCLIPGPTNeoXForCausalLM`` (mafed/model/vl_pythia.py:209-237) so reference
    checkpoints load: ``gpt_neox.embed_in.weight``, ``gpt_neox.layers.{i}.*``, ``gpt_neox.final_layer_norm.*``,
    ``embed_out.weight``, ``vision_embed_tokens.{0,2}.*``
  * ``model(input_ids=, pixel_values=, attention_mask=, labels=, output_hidden_states=, return_dict=True, **kw)``
    -> object with ``.loss``, ``.logits``, ``.hidden_states`` (mafed/model/vl_pythia.py:247-326)

Differences kept deliberately small and documented in DESIGN.md: the frozen vision encoder is the path's input
boundary (``pixel_values`` may be pre-computed ``[B,1+P,Dv]``/``[B,P,Dv]`` features, or a user-supplied frozen
``vision_encoder`` module is called on images); ``.logits`` covers the T text positions only (the reference computes
all S positions and uses the last T, vl_pythia.py:89,310).

The forward/backward of the whole model is ONE autograd node with a hand-scheduled backward (no per-op autograd
graph): activations live in plain buffers, parameter gradients are accumulated by the kernels straight into a flat
fp32 gradient buffer (the RCCL bucket source), hidden-state gradients coming from the distillation loss are injected
at the layer boundaries.
"""
from __future__ import annotations

import copy
import math
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence, Tuple

import os as _os

import torch
from torch import nn

from mafed_amd import ops
from mafed_amd._lib import EPI_GELU, EPI_GELU_BWD, EPI_NONE


# ----------------------------------------------------------------------------------------------------------------
@dataclass
class VLPythiaConfig:
    """Fields of config/vlpythia-base.json that the path reads."""

    vocab_size: int = 50304
    hidden_size: int = 1024
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    intermediate_size: int = 4096
    rotary_pct: float = 0.25
    rotary_emb_base: float = 10000.0
    layer_norm_eps: float = 1e-5
    initializer_range: float = 0.02
    vision_hidden_size: int = 1024     # EVA02-L / CLIP-L feature width
    num_vision_tokens: int = 256       # patches kept by feature_select("patch")
    use_parallel_residual: bool = True
    select_feature: str = "patch"

    PRESETS = {"160m": (768, 12, 12), "410m": (1024, 24, 16), "1b": (2048, 16, 8), "1.4b": (2048, 24, 16)}

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads

    @property
    def rotary_ndims(self) -> int:
        return int(self.head_dim * self.rotary_pct)

    @classmethod
    def preset(cls, name: str, **kw) -> "VLPythiaConfig":
        h, L, H = cls.PRESETS[name]
        return cls(hidden_size=h, num_hidden_layers=L, num_attention_heads=H, intermediate_size=4 * h, **kw)

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "VLPythiaConfig":
        keys = {f for f in cls.__dataclass_fields__}
        return cls(**{k: v for k, v in d.items() if k in keys})


@dataclass
class CausalLMOutput:
    """Stand-in for transformers' CausalLMOutputWithPast (mafed/model/vl_pythia.py:320-326)."""

    loss: Optional[torch.Tensor] = None
    logits: Optional[torch.Tensor] = None
    past_key_values: Any = None
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None
    attentions: Any = None
    mafed_ctx: Any = None  # (activation record, hook tensor) for the fused distillation path of mafed_amd's own plugin

    def __getitem__(self, i):
        return tuple(v for v in (self.loss, self.logits, self.hidden_states) if v is not None)[i]


# parameter holders: modules without a forward; the tree only exists so that state-dict names match the reference
class _Affine(nn.Module):
    def __init__(self, w: nn.Parameter, b: Optional[nn.Parameter]):
        super().__init__()
        self.weight = w
        if b is not None:
            self.bias = b


class _FrozenVision(nn.Module):
    """Placeholder for the frozen encoder (vqa_cont_learner.py:202-203 freezes ``model.vision_encoder.parameters()``)."""

    def __init__(self, encoder: Optional[nn.Module] = None):
        super().__init__()
        if encoder is not None:
            self.encoder = encoder

    def forward(self, x):
        enc = getattr(self, "encoder", None)
        if enc is None:
            return x
        with torch.no_grad():
            return enc.forward_features(x) if hasattr(enc, "forward_features") else enc(x)


NO_DECAY_KEYS = ("bias", "LayerNorm.bias", "LayerNorm.weight", "vqa_output_distill_loss_params")


def is_no_decay(name: str) -> bool:
    """Name test of BaseModule.configure_optimizers (vqa_cont_learner.py:73-78): GPT-NeoX LayerNorm parameters are
    called ``*layernorm.*`` (lower case), so only names containing ``bias`` escape weight decay."""
    return any(k in name for k in NO_DECAY_KEYS)


def _param_specs(cfg: VLPythiaConfig) -> List[Tuple[str, Tuple[int, ...]]]:
    h, ff, V, dv = cfg.hidden_size, cfg.intermediate_size, cfg.vocab_size, cfg.vision_hidden_size
    out: List[Tuple[str, Tuple[int, ...]]] = [("gpt_neox.embed_in.weight", (V, h))]
    for i in range(cfg.num_hidden_layers):
        p = f"gpt_neox.layers.{i}."
        out += [(p + "input_layernorm.weight", (h,)), (p + "input_layernorm.bias", (h,)),
                (p + "post_attention_layernorm.weight", (h,)), (p + "post_attention_layernorm.bias", (h,)),
                (p + "attention.query_key_value.weight", (3 * h, h)), (p + "attention.query_key_value.bias", (3 * h,)),
                (p + "attention.dense.weight", (h, h)), (p + "attention.dense.bias", (h,)),
                (p + "mlp.dense_h_to_4h.weight", (ff, h)), (p + "mlp.dense_h_to_4h.bias", (ff,)),
                (p + "mlp.dense_4h_to_h.weight", (h, ff)), (p + "mlp.dense_4h_to_h.bias", (h,))]
    out += [("gpt_neox.final_layer_norm.weight", (h,)), ("gpt_neox.final_layer_norm.bias", (h,)),
            ("embed_out.weight", (V, h)),
            ("vision_embed_tokens.0.weight", (h, dv)), ("vision_embed_tokens.0.bias", (h,)),
            ("vision_embed_tokens.2.weight", (h, h)), ("vision_embed_tokens.2.bias", (h,))]
    return out


class VLPythiaForCausalLM(nn.Module):
    """MI355X-native counterpart of ``VLCLIPGPTNeoXForCausalLM`` for the training hot path."""

    def __init__(self, config: VLPythiaConfig, compute_dtype: torch.dtype = torch.bfloat16, device: Any = None,
                 vision_encoder: Optional[nn.Module] = None, seed: Optional[int] = None):
        super().__init__()
        assert compute_dtype in (torch.bfloat16, torch.float32)
        assert config.use_parallel_residual, "GPT-NeoX sequential residual is not on the MAFED path (config/vlpythia-base.json:30)"
        self.config = config
        self.compute_dtype = compute_dtype
        dev = torch.device(device) if device is not None else torch.device("cuda" if torch.cuda.is_available() else "cpu")
        specs = _param_specs(config)
        # flat layout: [decayed ...][non-decayed ...], every tensor starts on a 64-element boundary
        order = [s for s in specs if not is_no_decay(s[0])] + [s for s in specs if is_no_decay(s[0])]
        offs, off = {}, 0
        n_decay_end = 0
        for name, shape in order:
            n = int(math.prod(shape))
            offs[name] = (off, n, shape)
            off += (n + 63) // 64 * 64
            if not is_no_decay(name):
                n_decay_end = off
        self._offsets = offs
        self._n_flat = off
        self._n_decay = n_decay_end
        self.flat_params = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_grads = torch.zeros(off, dtype=torch.float32, device=dev)
        self.flat_shadow = torch.zeros(off, dtype=torch.bfloat16, device=dev) if compute_dtype == torch.bfloat16 else None
        self._shadow_dirty = True
        self._build_tree(specs)
        from mafed_amd.vision import ClipVisionTower
        if isinstance(vision_encoder, ClipVisionTower):
            # native frozen tower: registered directly so that its tensors appear as ``vision_encoder.vision_model.*``, the names
            # of the reference's state dict (CLIPVisionModel under ``vision_encoder``, vl_pythia.py:215)
            if vision_encoder.config.hidden_size != config.vision_hidden_size or vision_encoder.config.num_patches != config.num_vision_tokens:
                raise ValueError("vision tower (hidden %d, %d patches) does not match the config (vision_hidden_size %d, num_vision_tokens %d)" % (
                    vision_encoder.config.hidden_size, vision_encoder.config.num_patches, config.vision_hidden_size, config.num_vision_tokens))
            self.vision_encoder = vision_encoder
        else:
            self.vision_encoder = _FrozenVision(vision_encoder)
        self._anchor = torch.zeros(1, device=dev, requires_grad=True)
        self._rot_cache: Dict[Tuple[int, str], Tuple[torch.Tensor, torch.Tensor]] = {}
        # callable(i): fired when layer i's parameter gradients are final for this backward; i = L for the LM head /
        # final LayerNorm, -1 when everything (embeddings, projector) is done.  Used by the DDP bucket reducer.
        self.grad_ready_hook = None
        # callable(l, x): fired in a training forward as soon as hidden_states[l] (fp32 [rows, h]) is final; the MAFED plugin
        # uses it to start that layer's distillation sums on a side stream, under the following layers' GEMMs
        self.hidden_ready_hook = None
        # {"pre" | ("layer", i) | "head": event} left by FlatAdamW.apply_pipelined: the forward waits chunk by chunk
        self._param_events = None
        self._side = None
        self._view_cache: Dict[Tuple[int, str], torch.Tensor] = {}
        self.overlap_param_grads = True  # run dW / bias-gradient kernels on side_stream() concurrently with the dX chain
        # bf16: the four weight gradients dW += dY^T.X of `dw_group_layers` consecutive layers are deferred and launched as ONE grouped
        # persistent GEMM (mafed_gemm_grouped) on the main stream: 2 layers = 768 tiles of 128 x 256 = three whole rounds of the
        # 256 CUs without split-K (one layer's products alone leave a third to seven eighths of the chip idle); 0 = one launch each
        self.dw_group_layers = 2
        self.sparse_lm_head = True          # batches that carry ``max_label_rows`` get the row-sparse LM head in training
        self.defer_ln_param_reduce = True   # LayerNorm parameter-gradient reduction on a side stream (needs overlap_param_grads)
        self.reset_parameters(seed)
        self.register_load_state_dict_post_hook(lambda m, ik: setattr(m, "_shadow_dirty", True))

    # ---- construction ---------------------------------------------------------------------------------------
    def _build_tree(self, specs):
        params: Dict[str, nn.Parameter] = {}
        for name, shape in specs:
            o, n, _ = self._offsets[name]
            p = nn.Parameter(self.flat_params[o:o + n].view(shape))
            p.grad = self.flat_grads[o:o + n].view(shape)
            params[name] = p
        self._params_by_name = params
        cfg = self.config
        neox = nn.Module()
        neox.embed_in = _Affine(params["gpt_neox.embed_in.weight"], None)
        layers = nn.ModuleList()
        for i in range(cfg.num_hidden_layers):
            pre = f"gpt_neox.layers.{i}."
            lyr = nn.Module()
            lyr.input_layernorm = _Affine(params[pre + "input_layernorm.weight"], params[pre + "input_layernorm.bias"])
            lyr.post_attention_layernorm = _Affine(params[pre + "post_attention_layernorm.weight"], params[pre + "post_attention_layernorm.bias"])
            att = nn.Module()
            att.query_key_value = _Affine(params[pre + "attention.query_key_value.weight"], params[pre + "attention.query_key_value.bias"])
            att.dense = _Affine(params[pre + "attention.dense.weight"], params[pre + "attention.dense.bias"])
            lyr.attention = att
            mlp = nn.Module()
            mlp.dense_h_to_4h = _Affine(params[pre + "mlp.dense_h_to_4h.weight"], params[pre + "mlp.dense_h_to_4h.bias"])
            mlp.dense_4h_to_h = _Affine(params[pre + "mlp.dense_4h_to_h.weight"], params[pre + "mlp.dense_4h_to_h.bias"])
            lyr.mlp = mlp
            layers.append(lyr)
        neox.layers = layers
        neox.final_layer_norm = _Affine(params["gpt_neox.final_layer_norm.weight"], params["gpt_neox.final_layer_norm.bias"])
        self.gpt_neox = neox
        self.embed_out = _Affine(params["embed_out.weight"], None)
        vt = nn.Module()
        setattr(vt, "0", _Affine(params["vision_embed_tokens.0.weight"], params["vision_embed_tokens.0.bias"]))
        setattr(vt, "2", _Affine(params["vision_embed_tokens.2.weight"], params["vision_embed_tokens.2.bias"]))
        self.vision_embed_tokens = vt

    def reset_parameters(self, seed: Optional[int] = None):
        """HF init distribution: Linear/Embedding N(0, initializer_range), LayerNorm (1, 0), biases 0."""
        g = torch.Generator(device="cpu")
        g.manual_seed(0 if seed is None else int(seed))
        with torch.no_grad():
            for name, p in self._params_by_name.items():
                if "layernorm" in name or "layer_norm" in name:
                    p.fill_(1.0 if name.endswith("weight") else 0.0)
                elif name.endswith("bias"):
                    p.zero_()
                else:
                    p.copy_(torch.randn(p.shape, generator=g) * self.config.initializer_range)
        self._shadow_dirty = True

    @classmethod
    def from_pretrained(cls, pretrained_model_name: str, vision_encoder_name: str = "", select_layer: int = -2,
                        select_feature: str = "patch", use_flash_attention_2: bool = False, state_dict=None, **kw):
        """Signature of the reference loader (mafed/model/vl_pythia.py:385-394) for a LOCAL directory holding
        ``config.json`` and ``model.safetensors`` / ``pytorch_model.bin``; hub names need network and are refused."""
        import json
        import os
        if not os.path.isdir(pretrained_model_name):
            raise ValueError(f"{pretrained_model_name!r} is not a local directory (no network access on this path)")
        with open(os.path.join(pretrained_model_name, "config.json")) as fp:
            cfg = VLPythiaConfig.from_dict(json.load(fp))
        model = cls(cfg, **kw)
        st = os.path.join(pretrained_model_name, "model.safetensors")
        if os.path.exists(st):
            from safetensors.torch import load_file
            sd = load_file(st)
        else:
            sd = torch.load(os.path.join(pretrained_model_name, "pytorch_model.bin"), map_location="cpu")
        model.load_state_dict({k: v for k, v in sd.items() if not k.startswith("vision_encoder.")}, strict=False)
        if state_dict is not None:
            model.load_state_dict(state_dict, strict=False)
        return model

    def __deepcopy__(self, memo):
        """Teacher snapshot (mafed/methods/distillation.py:211-213): one flat device copy instead of a per-tensor walk."""
        # (the frozen tower is shared, not copied: upstream's deepcopy duplicates ~0.3 B frozen parameters per teacher)
        enc = self.vision_encoder if not isinstance(self.vision_encoder, _FrozenVision) else getattr(self.vision_encoder, "encoder", None)
        new = VLPythiaForCausalLM(self.config, self.compute_dtype, self.flat_params.device, vision_encoder=enc)
        with torch.no_grad():
            new.flat_params.copy_(self.flat_params)
        new._shadow_dirty = True
        new.train(self.training)
        return new

    def _apply(self, fn, recurse=True):
        """``.to(device)`` / ``.cuda()``: move the flat buffers and re-point every parameter view at them."""
        probe = fn(self.flat_params)
        if probe.device != self.flat_params.device or probe.dtype != self.flat_params.dtype:
            if probe.dtype != torch.float32:
                raise TypeError("master weights stay fp32; choose compute_dtype at construction")
            self.flat_params = probe
            self.flat_grads = fn(self.flat_grads)
            if self.flat_shadow is not None:
                self.flat_shadow = self.flat_shadow.to(probe.device)
            self._anchor = torch.zeros(1, device=probe.device, requires_grad=True)
            for name, p in self._params_by_name.items():
                o, n, shape = self._offsets[name]
                p.data = self.flat_params[o:o + n].view(shape)
                p.grad = self.flat_grads[o:o + n].view(shape)
            self._rot_cache.clear()
            self._view_cache.clear()
            self._side = None
            self._shadow_dirty = True
            enc = self.vision_encoder if not isinstance(self.vision_encoder, _FrozenVision) else getattr(self.vision_encoder, "encoder", None)
            if enc is not None:
                enc._apply(fn)
            return self
        return self

    # ---- streams ---------------------------------------------------------------------------------------------------
    N_SIDE = int(_os.environ.get("MAFED_N_SIDE", "3"))   # parameter-gradient streams; same-box A/B on the round-2 build: 1 -> 33.4, 2 -> 34.0, 3 -> 33.4, 4 -> 33.8 ms

    def side_streams(self):
        """Extra HIP streams of this replica: parameter-gradient GEMMs (dW = dY^T.X, bias column sums) run here, off the
        critical dX chain.  Their grids are small (64 / 192 / 256 / 256 tiles per layer at 410M): spread over three
        streams they are co-resident and fill the 512 block slots together with the main stream's kernels."""
        if self._side is None:
            self._side = [torch.cuda.Stream(device=self.flat_params.device) for _ in range(self.N_SIDE)]
        return self._side

    def side_stream(self):
        return self.side_streams()[0]

    def _hook_zero(self) -> torch.Tensor:
        """A zero scalar of this replica's device, filled once (value of the 0-dim hook outputs / gradients nobody reads)."""
        z = getattr(self, "_hook_zero_t", None)
        if z is None or z.device != self.flat_params.device:
            z = self._hook_zero_t = torch.zeros(1, device=self.flat_params.device)
        return z

    # ---- parameter views ---------------------------------------------------------------------------------------
    def sync_shadow(self):
        """Refresh the bf16 copy of the weights read by the MFMA GEMMs (the optimiser kernel keeps it current)."""
        if self.flat_shadow is not None and self.flat_params.is_cuda:
            ops.cast(self.flat_params, torch.bfloat16, out=self.flat_shadow)
        self._shadow_dirty = False

    def _view(self, cache_id: int, src: torch.Tensor, name: str) -> torch.Tensor:
        """Cached view of one tensor inside a flat buffer (the host path looks these up ~40 times per layer per step)."""
        key = (cache_id, name)
        v = self._view_cache.get(key)
        if v is None or v._base is not src:
            o, n, shape = self._offsets[name]
            v = src[o:o + n].view(shape)
            self._view_cache[key] = v
        return v

    def _w(self, name: str) -> torch.Tensor:
        """Weight in compute dtype."""
        return self._view(0, self.flat_shadow if self.compute_dtype == torch.bfloat16 else self.flat_params, name)

    def _p(self, name: str) -> torch.Tensor:
        return self._view(1, self.flat_params, name)

    def _g(self, name: str) -> torch.Tensor:
        return self._view(2, self.flat_grads, name)

    def zero_grad(self, set_to_none: bool = False):  # gradients are views of the flat buffer: always zero in place
        self.flat_grads.zero_()
        self._dw_stale = False

    def layer_matrix_range(self, i: int) -> Tuple[int, int]:
        """Flat range of layer i's four weight matrices (query_key_value, dense, dense_h_to_4h, dense_4h_to_h: contiguous, behind the layer's
        two LayerNorm weights in the decayed segment) -- the part of the gradient buffer that the grouped weight-gradient GEMMs write whole."""
        pre = f"gpt_neox.layers.{i}."
        lo = self._offsets[pre + "attention.query_key_value.weight"][0]
        o, n, _ = self._offsets[pre + "mlp.dense_4h_to_h.weight"]
        names = [pre + "attention.query_key_value.weight", pre + "attention.dense.weight", pre + "mlp.dense_h_to_4h.weight", pre + "mlp.dense_4h_to_h.weight"]
        assert all(lo <= self._offsets[k][0] < o + n for k in names)
        return lo, o + (n + 63) // 64 * 64

    def _zero_layer_matrices(self, layers) -> None:
        for i in layers:
            lo, hi = self.layer_matrix_range(i)
            self.flat_grads[lo:hi].zero_()

    def decay_split(self) -> int:
        """flat[:n] is weight-decayed, flat[n:] is not."""
        return self._n_decay

    def rotary_tables(self, S: int) -> Tuple[torch.Tensor, torch.Tensor]:
        """cos/sin [S, rot/2] fp32 (GPTNeoXRotaryEmbedding, tf:52-108; positions = arange(S), pads not skipped)."""
        key = (S, str(self.flat_params.device))
        if key not in self._rot_cache:
            rd = self.config.rotary_ndims
            inv = 1.0 / (self.config.rotary_emb_base ** (torch.arange(0, rd, 2, dtype=torch.float32) / rd))
            fr = torch.arange(S, dtype=torch.float32)[:, None] * inv[None, :]
            self._rot_cache[key] = (fr.cos().contiguous().to(self.flat_params.device), fr.sin().contiguous().to(self.flat_params.device))
        return self._rot_cache[key]

    # ---- public forward ------------------------------------------------------------------------------------------
    def get_patch_embeddings(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """get_patch_embeddings + feature_select (mafed/model/vl_pythia.py:453-475): images [B,3,H,W] go through the frozen
        tower (the native CLIP tower of mafed_amd.vision, or a user-supplied module); features pass through."""
        P, dv = self.config.num_vision_tokens, self.config.vision_hidden_size
        x = pixel_values
        if x.dim() == 4:
            x = self.vision_encoder(x)
        if x.dim() != 3 or x.shape[-1] != dv:
            raise ValueError(f"expected vision features [B,{P}(+1),{dv}], got {tuple(x.shape)}")
        if x.shape[1] == P + 1 and self.config.select_feature == "patch":
            x = x[:, 1:]
        if x.shape[1] != P:
            raise ValueError(f"expected {P} patch tokens, got {x.shape[1]}")
        return x

    def forward(self, input_ids: Optional[torch.Tensor] = None, pixel_values: Optional[torch.Tensor] = None,
                attention_mask: Optional[torch.Tensor] = None, position_ids=None, inputs_embeds=None, head_mask=None,
                past_key_values=None, labels: Optional[torch.Tensor] = None, use_cache=None, output_attentions=None,
                output_hidden_states: Optional[bool] = None, return_dict: Optional[bool] = None,
                allow_input_gradients: bool = False, patch_embeddings: Optional[torch.Tensor] = None, **kwargs):
        if input_ids is None or (pixel_values is None and patch_embeddings is None):
            raise ValueError("the training path needs input_ids and pixel_values / patch_embeddings")
        if position_ids is not None or inputs_embeds is not None or past_key_values is not None or use_cache or output_attentions:
            raise NotImplementedError("position_ids / inputs_embeds / KV cache / attentions are outside the MAFED training path")
        feats = patch_embeddings if patch_embeddings is not None else self.get_patch_embeddings(pixel_values)
        if attention_mask is None:
            attention_mask = torch.ones_like(input_ids)
        want_h = bool(output_hidden_states)
        dev = self.flat_params.device
        input_ids = input_ids.to(dev, torch.int64).contiguous()
        attention_mask = attention_mask.to(dev, torch.int64).contiguous()
        labels = labels.to(dev, torch.int64).contiguous() if labels is not None else None
        feats = feats.to(dev).contiguous()
        if torch.is_grad_enabled():
            ctx_box: List[Any] = []
            # ``max_label_rows`` (an int the replay buffer attaches to its batches): upper bound on the labelled positions of a sample ->
            # row-sparse LM head (loss and gradients unchanged; ``.logits`` is None then: nothing on the training path reads it)
            hint = kwargs.get("max_label_rows") if (self.sparse_lm_head and labels is not None) else None
            outs = _ModelFn.apply(self._anchor, self, feats, input_ids, attention_mask, labels, want_h, ctx_box, hint)
            loss = outs[0] if labels is not None else None
            logits, hs = (outs[1] if outs[1].numel() else None), tuple(outs[3:]) if want_h else None
            mctx = (ctx_box[0], outs[2]) if want_h else None
        else:
            mctx = None
            st = self._engine_forward(feats, input_ids, attention_mask, labels, want_h, train=False)
            loss = st["loss"].reshape(()) if st["loss"] is not None else None
            logits, hs = st["logits"], tuple(st["hidden"]) if want_h else None
        out = CausalLMOutput(loss=loss, logits=logits, hidden_states=hs, mafed_ctx=mctx)
        if return_dict is False:
            return tuple(v for v in (out.loss, out.logits, out.hidden_states) if v is not None)
        return out

    @torch.no_grad()
    def hidden_states_upto(self, input_ids, attention_mask, pixel_values=None, patch_embeddings=None, n_hidden: Optional[int] = None):
        """Frozen-teacher fast path (mafed/methods/distillation.py:218-224): hidden_states[0 .. n_hidden-1] only --
        the stack stops after layer n_hidden-2, no LM head, no saved activations."""
        feats = patch_embeddings if patch_embeddings is not None else self.get_patch_embeddings(pixel_values)
        dev = self.flat_params.device
        st = self._engine_forward(feats.to(dev).contiguous(), input_ids.to(dev, torch.int64).contiguous(),
                                  attention_mask.to(dev, torch.int64).contiguous(), None, True, train=False, n_hidden=n_hidden)
        return tuple(st["hidden"])

    # ---- greedy decode -----------------------------------------------------------------------
    @torch.no_grad()
    def generate(self, input_ids: torch.Tensor, attention_mask: Optional[torch.Tensor] = None, pixel_values: Optional[torch.Tensor] = None,
                 patch_embeddings: Optional[torch.Tensor] = None, max_new_tokens: int = 10, use_cache: bool = True,
                 pad_token_id: Optional[int] = None, eos_token_id: Optional[int] = 0, do_sample: bool = False,
                 return_step_logits: bool = False, use_graph: bool = False, **kwargs):
        """Greedy search with the call signature the reference validation uses (mafed/model/vqa_cont_learner.py:260-267,
        mafed/utils/eval_utils.py:170-177: ``generate(input_ids=, attention_mask=, pixel_values=, max_new_tokens=10,
        use_cache=False, pad_token_id=eos)``) and HF ``greedy_search`` semantics (transformers 4.37.1): next token = argmax of
        the last position, finished rows keep emitting ``pad_token_id``, the attention mask grows by ones, generation stops
        when every row has produced ``eos_token_id`` (GPT-NeoX / Pythia: 0) or after ``max_new_tokens``.  Positions are
        ``arange`` over [image | text | generated].

        ``use_cache=False`` is the reference's literal behaviour -- the whole 256 + T + t prefix is pushed through the stack
        again for every token.  ``use_cache=True`` (default here) runs the prefix once, keeps each layer's fused-QKV output as
        the K/V cache and then moves ONE row per sample through the stack per token (``mafed_attn_decode``); both produce the
        same tokens.  Returns [B, T + n_generated] like HF; with ``return_step_logits`` also the fp32 last-position logits of
        every step [n, B, V]."""
        if do_sample or kwargs.get("num_beams", 1) != 1:
            raise NotImplementedError("only greedy search is on the validation path")
        if input_ids is None or (pixel_values is None and patch_embeddings is None):
            raise ValueError("generate needs input_ids and pixel_values / patch_embeddings")
        dev = self.flat_params.device
        feats = (patch_embeddings if patch_embeddings is not None else self.get_patch_embeddings(pixel_values)).to(dev).contiguous()
        ids = input_ids.to(dev, torch.int64).contiguous()
        am = (attention_mask if attention_mask is not None else torch.ones_like(input_ids)).to(dev, torch.int64).contiguous()
        if eos_token_id is not None and pad_token_id is None:
            pad_token_id = eos_token_id  # HF's fallback for open-end generation
        B, T = ids.shape
        unfinished = torch.ones(B, dtype=torch.int64, device=dev)
        new_tokens, step_logits = [], []
        if not hasattr(self, "_decode_graphs"):
            self._decode_graphs = {}

        def pick(last_logits):
            nonlocal unfinished
            nxt = last_logits.float().argmax(dim=-1)
            if eos_token_id is not None:
                nxt = nxt * unfinished + pad_token_id * (1 - unfinished)
                unfinished = unfinished * (nxt != eos_token_id).to(torch.int64)
            new_tokens.append(nxt)
            if return_step_logits:
                step_logits.append(last_logits.float())
            return nxt

        if not use_cache:
            cur_ids, cur_am = ids, am
            for _ in range(max_new_tokens):
                st = self._engine_forward(feats, cur_ids, cur_am, None, False, train=False)
                nxt = pick(st["logits"][:, -1, :])
                cur_ids = torch.cat([cur_ids, nxt[:, None]], dim=1)
                cur_am = torch.cat([cur_am, torch.ones_like(nxt)[:, None]], dim=1)
        elif use_graph and not return_step_logits and max_new_tokens > 1:
            # opt-in: the nine one-row-per-sample steps (~150 launches of 5-20 us kernels each) captured once per (B, T, max_new)
            # into a hipGraph whose K/V cache lives at fixed addresses (the prefill's QKV GEMMs write straight into it).  It
            # takes the host out of the loop; on an idle host it measures the same as eager launches (24.1 vs 24.2 ms at
            # 410M / B = 32): the steps are bound by the GPU-side cost of that many small kernels.
            gd = self._decode_graphs.get((B, T, max_new_tokens, eos_token_id, pad_token_id))
            if gd is None:
                gd = self._decode_graphs[(B, T, max_new_tokens, eos_token_id, pad_token_id)] = _GraphedDecode(
                    self, B, T, max_new_tokens, eos_token_id, pad_token_id)
            gen_all = gd.run(feats, ids, am)
            new_tokens = list(gen_all.unbind(1))
        else:
            # every layer's fused-QKV output lands in one tensor (the K/V cache's prefix): its keys are then rotated by ONE launch
            cfg_ = self.config
            S_ = cfg_.num_vision_tokens + T
            store = torch.empty((cfg_.num_hidden_layers, B * S_, 3 * cfg_.num_attention_heads * cfg_.head_dim), dtype=self.compute_dtype, device=dev)
            st = self._engine_forward(feats, ids, am, None, False, train=False, qkv_out=list(store.unbind(0)), last_only=True)
            cache = _DecodeCache(self, list(store.unbind(0)), B, st["S"], max_new_tokens, am, fused=getattr(self, "fused_decode", True),
                                 prefix_storage=store)
            nxt = pick(st["logits"][:, -1, :])
            for t in range(max_new_tokens - 1):
                nxt = pick(self._engine_decode_step(nxt, t, cache))
        gen = torch.stack(new_tokens, dim=1)
        if eos_token_id is not None:
            # HF leaves the loop as soon as every row has finished: the output is as long as the slowest row needed
            done = (gen == eos_token_id).to(torch.int64).cumsum(1).clamp_(max=1)       # 1 from the first eos on
            first = (done.shape[1] - done.sum(1)) + done[:, -1]                        # tokens up to and including the first eos
            n_keep = int(first.max().clamp_(max=gen.shape[1]))
            gen = gen[:, :n_keep]
            step_logits = step_logits[:n_keep]
        out = torch.cat([ids, gen], dim=1)
        if return_step_logits:
            return out, torch.stack(step_logits, dim=0)
        return out

    def _engine_decode_step(self, tokens: torch.Tensor, t: int, cache: "_DecodeCache") -> torch.Tensor:
        """One token per sample through the stack: ``tokens`` [B] sit at position S0 + t; returns the logits [B, V]."""
        cfg, cd = self.config, self.compute_dtype
        h, H, D, L = cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim, cfg.num_hidden_layers
        B, S0, rot = cache.B, cache.S0, cfg.rotary_ndims
        cos, sin = self.rotary_tables(S0 + cache.cap)
        w = self._w
        x = self._p("gpt_neox.embed_in.weight").index_select(0, tokens)  # fp32 residual stream row
        if cache.flow is not None:
            # the whole step as one launch (csrc/decode_flow.hip): work items hand over through arrival counters
            fl = cache.flow
            torch.index_select(self._p("gpt_neox.embed_in.weight"), 0, tokens, out=fl.x[:B])
            return fl.step(S0, cache.cap, t, rot, cfg.num_vision_tokens, cos, sin, cache.attention_mask, cfg.layer_norm_eps)
        if cache.pair is not None:
            # two launches per layer: [LN1 | LN2] + QKV + fc1/GELU (csrc/decode.hip), then attention + dense + fc2 + residuals as one grid
            # whose dense K-slices wait for the attention slices' arrival counters (csrc/decode_flow.hip: decode_attn_out_kernel)
            pr = cache.pair
            pr.begin_step(t)
            for i in range(L):
                pre = f"gpt_neox.layers.{i}."
                ops.decode_ln_qkv_fc1(x, self._p(pre + "input_layernorm.weight"), self._p(pre + "input_layernorm.bias"),
                                      self._p(pre + "post_attention_layernorm.weight"), self._p(pre + "post_attention_layernorm.bias"),
                                      cfg.layer_norm_eps, w(pre + "attention.query_key_value.weight"),
                                      self._p(pre + "attention.query_key_value.bias"), cache.new[i][:, t, :],
                                      w(pre + "mlp.dense_h_to_4h.weight"), self._p(pre + "mlp.dense_h_to_4h.bias"), a_out=pr.act)
                pr.run(i, t, x, S0, rot, cfg.num_vision_tokens, cos, sin, cache.attention_mask)
            lnf, _, _, _ = ops.layernorm_fwd(x, self._p("gpt_neox.final_layer_norm.weight"), self._p("gpt_neox.final_layer_norm.bias"),
                                             None, None, cfg.layer_norm_eps, cd, save_stats=False)
            return ops.gemm(lnf, w("embed_out.weight"), False, True)
        if cache.fused:
            # three launches per layer (csrc/decode.hip): [LN1 | LN2] + QKV + fc1/GELU, attention over the pre-rotated cache, and
            # dense + fc2 + both residuals as one product over the concatenated K
            for i in range(L):
                pre = f"gpt_neox.layers.{i}."
                a = ops.decode_ln_qkv_fc1(x, self._p(pre + "input_layernorm.weight"), self._p(pre + "input_layernorm.bias"),
                                          self._p(pre + "post_attention_layernorm.weight"), self._p(pre + "post_attention_layernorm.bias"),
                                          cfg.layer_norm_eps, w(pre + "attention.query_key_value.weight"),
                                          self._p(pre + "attention.query_key_value.bias"), cache.new[i][:, t, :],
                                          w(pre + "mlp.dense_h_to_4h.weight"), self._p(pre + "mlp.dense_h_to_4h.bias"))
                ao = ops.attn_decode(cache.prefix[i], S0, cache.new[i], t, B, H, D, rot, cos, sin, cache.attention_mask, prerot=True)
                x = ops.decode_out(x, ao, a, w(pre + "attention.dense.weight"), self._p(pre + "attention.dense.bias"),
                                   w(pre + "mlp.dense_4h_to_h.weight"), self._p(pre + "mlp.dense_4h_to_h.bias"), cache.workspace, out=x)
            if B <= 32 and h == 1024 and cfg.vocab_size % 32 == 0 and cfg.vocab_size >= 16384:
                # final LayerNorm + LM head as one persistent launch (decode_head_kernel: rows normalised once per CU, the vocabulary's
                # weight strips streamed through LDS): 24 us against 48 for LayerNorm + the skinny product at V = 50k
                return ops.decode_ln_linear(x, self._p("gpt_neox.final_layer_norm.weight"), self._p("gpt_neox.final_layer_norm.bias"),
                                            cfg.layer_norm_eps, w("embed_out.weight"))
            # (other shapes: the one-slab-per-block forms of ops.decode_ln_linear are no faster than the two launches below)
            lnf, _, _, _ = ops.layernorm_fwd(x, self._p("gpt_neox.final_layer_norm.weight"), self._p("gpt_neox.final_layer_norm.bias"),
                                             None, None, cfg.layer_norm_eps, cd, save_stats=False)
            return ops.gemm(lnf, w("embed_out.weight"), False, True)
        for i in range(L):
            pre = f"gpt_neox.layers.{i}."
            ln1, ln2, _, _ = ops.layernorm_fwd(x, self._p(pre + "input_layernorm.weight"), self._p(pre + "input_layernorm.bias"),
                                               self._p(pre + "post_attention_layernorm.weight"), self._p(pre + "post_attention_layernorm.bias"),
                                               cfg.layer_norm_eps, cd, save_stats=False)
            # the new token's q | k | v row goes straight into the cache (row t of the per-layer [B, cap, 3*H*D] tensor)
            ops.gemm(ln1, w(pre + "attention.query_key_value.weight"), False, True, bias=self._p(pre + "attention.query_key_value.bias"),
                     out=cache.new[i][:, t, :])
            ao = ops.attn_decode(cache.prefix[i], S0, cache.new[i], t, B, H, D, rot, cos, sin, cache.attention_mask, prerot=cache.prerot)
            attn = ops.gemm(ao, w(pre + "attention.dense.weight"), False, True, bias=self._p(pre + "attention.dense.bias"), out_dtype=cd)
            a = ops.gemm(ln2, w(pre + "mlp.dense_h_to_4h.weight"), False, True, bias=self._p(pre + "mlp.dense_h_to_4h.bias"), epilogue=EPI_GELU)
            x = ops.gemm(a, w(pre + "mlp.dense_4h_to_h.weight"), False, True, bias=self._p(pre + "mlp.dense_4h_to_h.bias"),
                         out_dtype=torch.float32, res1=attn, res2=x)
        lnf, _, _, _ = ops.layernorm_fwd(x, self._p("gpt_neox.final_layer_norm.weight"), self._p("gpt_neox.final_layer_norm.bias"),
                                         None, None, cfg.layer_norm_eps, cd, save_stats=False)
        return ops.gemm(lnf, w("embed_out.weight"), False, True)

    # ---- engine ------------------------------------------------------------------------------------------------------
    def _engine_forward(self, feats, input_ids, attention_mask, labels, want_hidden, train, n_hidden: Optional[int] = None,
                        keep_qkv: bool = False, qkv_out: Optional[Sequence[torch.Tensor]] = None, label_rows_hint: Optional[int] = None,
                        last_only: bool = False):
        if not self.flat_params.is_cuda:
            raise RuntimeError("mafed_amd runs on the GPU only (no CPU fallback); move the model with .cuda()")
        pe, main_st = self._param_events, torch.cuda.current_stream()
        if pe is not None:
            main_st.wait_event(pe["pre"])
        if self._shadow_dirty:
            self.sync_shadow()
        cfg, cd = self.config, self.compute_dtype
        B, T = input_ids.shape
        P, h, H, D, L = cfg.num_vision_tokens, cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim, cfg.num_hidden_layers
        S = P + T
        rows = B * S
        rot = cfg.rotary_ndims
        cos, sin = self.rotary_tables(S)
        w = self._w
        sv: Dict[str, Any] = {"B": B, "T": T, "P": P, "S": S, "input_ids": input_ids, "attention_mask": attention_mask, "labels": labels,
                              "layers": []}
        # projector: Linear -> GELU(erf) -> Linear (vl_pythia.py:226-234,270)
        f2 = feats.reshape(B * P, cfg.vision_hidden_size)
        if f2.dtype not in (torch.float32, torch.bfloat16):
            f2 = f2.float()
        fc = f2.contiguous() if f2.dtype == cd else ops.cast(f2.contiguous(), cd)
        u0 = torch.empty((B * P, h), dtype=cd, device=fc.device) if train else None
        a0 = ops.gemm(fc, w("vision_embed_tokens.0.weight"), False, True, bias=self._p("vision_embed_tokens.0.bias"), epilogue=EPI_GELU, aux=u0)
        img = ops.gemm(a0, w("vision_embed_tokens.2.weight"), False, True, bias=self._p("vision_embed_tokens.2.bias"))
        x = ops.embed_concat_fwd(img, self._p("gpt_neox.embed_in.weight"), input_ids, B, P, T)  # fp32 residual stream
        if train:
            sv["proj"] = (fc, u0, a0)
        hidden = [x.view(B, S, h)]
        hook = self.hidden_ready_hook if train else None
        if hook is not None:
            hook(0, x)
        n_layers = L if n_hidden is None else max(0, min(L, n_hidden - 1))
        for i in range(n_layers):
            pre = f"gpt_neox.layers.{i}."
            if pe is not None:
                main_st.wait_event(pe[("layer", i)])
            ln1, ln2, mean, rstd = ops.layernorm_fwd(x, self._p(pre + "input_layernorm.weight"), self._p(pre + "input_layernorm.bias"),
                                                     self._p(pre + "post_attention_layernorm.weight"), self._p(pre + "post_attention_layernorm.bias"),
                                                     cfg.layer_norm_eps, cd, save_stats=train)
            qkv = ops.gemm(ln1, w(pre + "attention.query_key_value.weight"), False, True, bias=self._p(pre + "attention.query_key_value.bias"),
                           out=qkv_out[i] if qkv_out is not None else None)  # (a captured decode graph reads its K/V cache at fixed addresses)
            ao, lse = ops.attn_fwd(qkv, B, S, H, D, rot, cos, sin, attention_mask)
            # the attention branch output is a bf16 tensor under the reference's autocast too (it meets the fp32 residual in the add)
            attn = ops.gemm(ao, w(pre + "attention.dense.weight"), False, True, bias=self._p(pre + "attention.dense.bias"), out_dtype=cd)
            u = torch.empty((rows, cfg.intermediate_size), dtype=cd, device=x.device) if train else None
            a = ops.gemm(ln2, w(pre + "mlp.dense_h_to_4h.weight"), False, True, bias=self._p(pre + "mlp.dense_h_to_4h.bias"), epilogue=EPI_GELU, aux=u)
            # h + attn(LN1(h)) + mlp(LN2(h)) in the last GEMM's epilogue (tf:271-274)
            xn = ops.gemm(a, w(pre + "mlp.dense_4h_to_h.weight"), False, True, bias=self._p(pre + "mlp.dense_4h_to_h.bias"),
                          out_dtype=torch.float32, res1=attn, res2=x)
            if train:
                sv["layers"].append({"x": x, "mean": mean, "rstd": rstd, "ln1": ln1, "ln2": ln2, "qkv": qkv, "ao": ao, "lse": lse, "u": u, "a": a})
            elif keep_qkv:
                sv["layers"].append({"qkv": qkv})  # the prefill's K/V cache: exactly what the fused QKV GEMM wrote
            x = xn
            if i < L - 1:
                hidden.append(x.view(B, S, h))
                if hook is not None:
                    hook(i + 1, x)
        sv["hidden"] = hidden
        sv["loss"] = None
        sv["logits"] = None
        if pe is not None:  # from here on the caller's stream is ordered behind every chunk of the pipelined update
            for k in range(n_layers, L):
                main_st.wait_event(pe[("layer", k)])
            main_st.wait_event(pe["head"])
            self._param_events = None
        if n_hidden is not None:
            return sv
        if last_only:
            # a decode prefill only needs the last position's logits: final LN + head on B rows instead of B * T (-> logits [B, 1, V])
            xl = x.view(B, S, h)[:, -1, :].contiguous()
            lnl, _, _, _ = ops.layernorm_fwd(xl, self._p("gpt_neox.final_layer_norm.weight"), self._p("gpt_neox.final_layer_norm.bias"),
                                             None, None, cfg.layer_norm_eps, cd, save_stats=False)
            sv["logits"] = ops.gemm(lnl, w("embed_out.weight"), False, True).view(B, 1, cfg.vocab_size)
            return sv
        # final LN (fp32 hidden state L only when asked for) + LM head on the T text positions (vl_pythia.py:89,310)
        xt = x.view(B, S, h)[:, P:, :].reshape(B * T, h)
        lnf, _, fmean, frstd = ops.layernorm_fwd(xt, self._p("gpt_neox.final_layer_norm.weight"), self._p("gpt_neox.final_layer_norm.bias"),
                                                 None, None, cfg.layer_norm_eps, cd, save_stats=train)
        if want_hidden:
            full, _, _, _ = ops.layernorm_fwd(x, self._p("gpt_neox.final_layer_norm.weight"), self._p("gpt_neox.final_layer_norm.bias"),
                                              None, None, cfg.layer_norm_eps, torch.float32, save_stats=False)
            hidden.append(full.view(B, S, h))
        # Row-sparse head (training, with the caller's bound on labelled positions per sample): only rows whose shifted label is a token
        # enter the head GEMM, the CE and -- in the backward -- the head's two gradient GEMMs: 4 answer tokens of 32 text positions in
        # the VQA batches, i.e. 256 of 1024 rows at B = 32 (Rc = slots per sample incl. the unlabelled last one, B * Rc a tile multiple)
        Rc = None
        if train and labels is not None and label_rows_hint is not None:
            need = max(2, int(label_rows_hint) + 1)   # (slots per sample incl. the unlabelled last one; a hint of 0 still gets two)
            Rc = need if cd == torch.float32 else next((r for r in range(need, T + 1) if (B * r) % 128 == 0), None)  # (the MFMA tiles want whole 128-row tiles)
            if Rc is not None and Rc * 2 > T:
                Rc = None   # not worth it
        if Rc is not None:
            ros, sor, labels_c, ov = ops.label_rows(labels, Rc)
            # device flag: 1 if a sample had more labelled positions than the hint promised -- rows were dropped; the CE below then
            # returns NaN (no host synchronisation: the step fails loudly instead of training on a wrong loss)
            self.last_label_overflow = ov
            lnf_c = ops.gather_rows(lnf, ros)
            logits = ops.gemm(lnf_c, w("embed_out.weight"), False, True).view(B, Rc, cfg.vocab_size)
            sv["logits"] = logits
            loss, lse_ce = ops.ce_fwd(logits, labels_c, poison=ov)
            sv["loss"], sv["ce_lse"] = loss, lse_ce
            sv["sparse_head"] = (sor, labels_c)
            sv["final"] = (xt, lnf_c, fmean, frstd)
            sv["x_last"] = x
            return sv
        logits = ops.gemm(lnf, w("embed_out.weight"), False, True).view(B, T, cfg.vocab_size)
        sv["logits"] = logits
        if labels is not None:
            loss, lse_ce = ops.ce_fwd(logits, labels)
            sv["loss"] = loss
            if train:
                sv["ce_lse"] = lse_ce
        if train:
            sv["final"] = (xt, lnf, fmean, frstd)
            sv["x_last"] = x
        return sv

    def hidden_grad_taps(self, batch: Dict[str, torch.Tensor], layers: Sequence[int]) -> Dict[int, torch.Tensor]:
        """dL_CE / d hidden_states[l] for every l in ``layers`` from ONE backward sweep (the adaptive-weights pass of
        mafed/methods/distillation_loss_weights.py:91-146 asks autograd for them one layer at a time)."""
        feats = batch["patch_embeddings"] if "patch_embeddings" in batch else self.get_patch_embeddings(batch["pixel_values"])
        dev = self.flat_params.device
        sv = self._engine_forward(feats.to(dev).contiguous(), batch["input_ids"].to(dev, torch.int64).contiguous(),
                                  batch["attention_mask"].to(dev, torch.int64).contiguous(),
                                  batch["labels"].to(dev, torch.int64).contiguous(), False, train=True)
        taps: Dict[int, torch.Tensor] = {int(l): None for l in layers}
        self._engine_backward(sv, torch.ones(1, device=dev), [], taps=taps)
        S = sv["S"]
        return {l: t.view(sv["B"], S, -1) for l, t in taps.items()}

    def _dw_group_fuses_squares(self, rows: int) -> bool:
        """Will a grouped weight-gradient launch of this model (``dw_group_layers`` layers x four matrices, K = rows) emit the squares of
        its outputs from the epilogue?  Asked of the library once per (rows, group size)."""
        cache = self.__dict__.setdefault("_dw_fuse_cache", {})
        key = (int(rows), int(getattr(self, "dw_group_layers", 2) or 0))
        if key not in cache:
            cfg = self.config
            h, f = cfg.hidden_size, cfg.intermediate_size
            per_layer = [(3 * h, h, rows), (h, h, rows), (f, h, rows), (h, f, rows)]
            n_layers = max(1, min(4, key[1]))
            cache[key] = bool(key[1]) and ops.gemm_grouped_fuses_sumsq(per_layer * n_layers, True, False)
        return cache[key]

    def _engine_backward(self, sv, dloss: Optional[torch.Tensor], dhidden: Sequence[Optional[torch.Tensor]], taps=None):
        # Data parallel, last micro-batch of a window: RCCL's all-reduce kernels hold a workgroup per channel for milliseconds while this
        # backward runs.  The persistent GEMMs assume all 256 of their blocks are resident at once -- with 8 CUs taken the late blocks run
        # a second wave and a launch takes 1.7x as long (tools/contention_bench.py: qkv 61.8 -> 105 us, grouped dW 440 -> 785), where the
        # 128 x 128 kernels' many small blocks lose 1.1 - 1.45x.  So this backward runs on those (Trainer sets the flag).
        # (per call: every GEMM this thread issues inside the block carries MAFED_EPI_NO_PERSISTENT; no process-wide switch is touched, a
        #  forced tuning variant or another thread's / model's launches are unaffected)
        cb = getattr(self, "contended_backward", False)
        if cb and self.flat_params.is_cuda:
            # "ticketed": the persistent kernels stay, their blocks draw tiles from per-XCD queues (MAFED_EPI_TICKETED) -- a launch then
            # tolerates the CUs the collective holds (1.2 - 1.3x instead of 1.7 - 1.9x with 8 - 32 CUs taken, tools/contention_bench.py);
            # True / "128x128": every GEMM of this backward on the 128 x 128 kernels (round 3's choice)
            ctx = ops.ticketed_gemm() if cb == "ticketed" else ops.no_persistent_gemm()
            with ctx:
                return self._engine_backward_impl(sv, dloss, dhidden, taps)
        return self._engine_backward_impl(sv, dloss, dhidden, taps)

    def _engine_backward_impl(self, sv, dloss: Optional[torch.Tensor], dhidden: Sequence[Optional[torch.Tensor]], taps=None):
        self._bw_serial = getattr(self, "_bw_serial", 0) + 1   # lets a gradient hook tell which backward sweep reported a range
        cfg, cd = self.config, self.compute_dtype
        B, T, P, S = sv["B"], sv["T"], sv["P"], sv["S"]
        h, H, D, L = cfg.hidden_size, cfg.num_attention_heads, cfg.head_dim, cfg.num_hidden_layers
        rows = B * S
        rot = cfg.rotary_ndims
        cos, sin = self.rotary_tables(S)
        w, g = self._w, self._g
        am = sv["attention_mask"]
        dev = self.flat_params.device
        if len(dhidden) > L and dhidden[L] is not None:
            raise NotImplementedError("gradient w.r.t. the post-final-LayerNorm hidden state is not on the MAFED path")
        inject = sv.get("inject")  # {layer: (teacher hidden state, device [4] = d loss / d {sum_lang, sum_vision, ., .})}
        inj_cos = bool(sv.get("inject_cosine", False))   # the injected loss is the cosine distance, not the MSE
        main = torch.cuda.current_stream()
        sides = self.side_streams() if self.overlap_param_grads else None
        keep: List[torch.Tensor] = []  # temporaries read by the side streams: kept alive until the join at the end
        rr = [0]

        mark = [None]  # event of the main stream's current position; dropped (main_moved) whenever more work is queued on it

        def main_moved():
            mark[0] = None

        def on_side(fn, *tensors, k=None):
            """Run parameter-gradient work after everything queued on the main stream so far, off the dX chain.  Consecutive
            hand-offs with no main-stream work in between share one event: each record is a marker packet the dX chain's next
            kernel waits behind (~4 us apiece in the step's timeline)."""
            if sides is None:
                fn()
                return
            if k is None:
                k = rr[0] % len(sides)
                rr[0] += 1
            if mark[0] is None:
                mark[0] = main.record_event()
            ev = mark[0]
            with torch.cuda.stream(sides[k]):
                sides[k].wait_event(ev)
                fn()
            keep.extend(tensors)

        def wgrad(dY, X, wname, bname=None):
            def run():
                ops.gemm(dY, X, True, False, out=g(wname), beta=1.0)
                if bname is not None:
                    ops.colsum_(dY, g(bname))
            on_side(run, dY, X)

        # layer weight gradients, grouped: (dY, X, gradient) records wait here (the list keeps dY / X alive) until `flush_dw`
        # (beside collectives the weight gradients go back to one 128 x 128-kernel launch per product on the side streams, as in round 2:
        #  a grouped call would fall back to eight serial launches on the dX chain's stream)
        group_dw = (cd == torch.bfloat16 and int(getattr(self, "dw_group_layers", 0)) > 0
                    and getattr(self, "contended_backward", False) in (False, None, "ticketed"))
        # First micro-batch of an accumulation window (Trainer sets ``grad_overwrite``): the grouped weight-gradient GEMMs WRITE the layers'
        # matrix gradients (beta = 0) instead of adding to a zeroed buffer -- the optimiser pass then does not zero-write those 1.2 GB
        # (FlatAdamW: ``skip_matrix_zero``) and the GEMM epilogues do not read them back.  ``_dw_stale`` = the last optimiser pass left the
        # matrices un-zeroed: a sweep that accumulates anyway (another caller, another kernel path) zeroes them first.
        overwrite = group_dw and bool(getattr(self, "grad_overwrite", False)) and taps is None
        if getattr(self, "_dw_stale", False) and not overwrite:
            self._zero_layer_matrices(range(L))
        self._dw_stale = False
        dw_beta = 0.0 if overwrite else 1.0
        # squares of the final matrix gradients from the weight-gradient epilogues (FlatAdamW.begin_incremental_norm): only a sweep whose
        # products all go through the grouped call can promise them -- it records its serial, the norm hook checks it
        dw_sq = getattr(self, "dw_sumsq", None) if (group_dw and taps is None) else None
        if dw_sq is not None and not self._dw_group_fuses_squares(sv["B"] * sv["S"]):
            dw_sq = None   # (h = 768 / 2048: the 256 x 256-tile kernel has no fused squares -- the norm hook's range pass is cheaper than a pass per matrix)
        self._dw_sumsq_used = self._bw_serial if dw_sq is not None else None
        DW_SLOT = {"attention.query_key_value.weight": 0, "attention.dense.weight": 1, "mlp.dense_h_to_4h.weight": 2, "mlp.dense_4h_to_h.weight": 3}
        pending_dw: List[dict] = []
        pending_layers: List[int] = []

        def wgrad_layer(dY, X, wname, bname=None):
            if not group_dw:
                wgrad(dY, X, wname, bname)
                return
            q = dict(A=dY, B=X, out=g(wname), beta=dw_beta)
            if dw_sq is not None:
                li_, kind = wname[len("gpt_neox.layers."):].split(".", 1)
                q["sumsq"] = dw_sq[int(li_), DW_SLOT[kind]]
            pending_dw.append(q)
            if bname is not None:
                on_side(lambda: ops.colsum_(dY, g(bname)), dY)

        def flush_dw():
            # at most PP_MAXP = 16 products per grouped launch (mafed_gemm_grouped launches larger lists one product at a time, serially on
            # this stream -- worse than both forms): `dw_group_layers` >= 5 is cut into several launches
            for c0 in range(0, len(pending_dw), 16):
                ops.gemm_grouped(pending_dw[c0:c0 + 16], True, False)
            if pending_dw:
                pending_dw.clear()
                main_moved()
            for li in pending_layers:
                ready(li)
            pending_layers.clear()

        def ready(i):
            """Bucket hook: fires on side stream 0 once every side stream has finished the gradients queued so far."""
            if self.grad_ready_hook is None:
                return
            if sides is None:
                self.grad_ready_hook(i)
                return
            evs = [st.record_event() for st in sides[1:]]
            def run():
                for e in evs:
                    sides[0].wait_event(e)
                self.grad_ready_hook(i)
            on_side(run, k=0)

        # deferred LayerNorm parameter reduction: only with side streams and when no external hidden-state gradient adds into the
        # same bias gradients from the main stream (generic autograd path of the cosine / CLS losses)
        defer_ln = (sides is not None and self.defer_ln_param_reduce and taps is None
                    and not any(d is not None for d in dhidden))
        dx = None  # gradient w.r.t. the residual stream leaving the current layer, fp32 [rows, h]
        if dloss is not None and sv["loss"] is not None:
            xt, lnf, fmean, frstd = sv["final"]
            logits = sv["logits"]
            gl = dloss.reshape(1).to(torch.float32).contiguous()
            sp = sv.get("sparse_head")   # (slot of every text row, compact labels): the head ran on the labelled rows only
            n_head = logits.shape[0] * logits.shape[1]
            dlog = ops.ce_bwd(logits, sp[1] if sp is not None else sv["labels"], sv["ce_lse"], gl).view(n_head, cfg.vocab_size)
            wgrad(dlog, lnf, "embed_out.weight")
            if cd == torch.bfloat16:
                # [rows, V] . [V, h]: few output tiles with K = 50304 -- accumulate-only fp32 output so that the GEMM splits K
                dlnf = torch.zeros((n_head, h), dtype=torch.float32, device=dev)
                ops.gemm(dlog, w("embed_out.weight"), False, False, out=dlnf, beta=1.0)
            else:
                dlnf = ops.gemm(dlog, w("embed_out.weight"), False, False)
            if sp is not None:
                dlnf = ops.gather_rows(dlnf if dlnf.dtype == torch.float32 else dlnf.float(), sp[0])   # back to the [B*T, h] text rows (zeros elsewhere)
            if defer_ln:
                dxt, _, fws = ops.layernorm_bwd_rows(dlnf, None, xt, fmean, frstd, self._p("gpt_neox.final_layer_norm.weight"), None, None)
                main_moved()
                on_side(lambda ws=fws: ops.layernorm_bwd_params(ws, B * T, h, g("gpt_neox.final_layer_norm.weight"),
                                                                g("gpt_neox.final_layer_norm.bias")), fws)
            else:
                dxt, _ = ops.layernorm_bwd(dlnf, None, xt, fmean, frstd, self._p("gpt_neox.final_layer_norm.weight"), None, None,
                                           g("gpt_neox.final_layer_norm.weight"), g("gpt_neox.final_layer_norm.bias"))
            dx, dy0 = ops.pad_text_rows(dxt, B, S, P, cd if cd != torch.float32 else None)
            main_moved()
            ready(L)
        else:
            dy0 = None
        dy = dy0  # dx in compute dtype (GEMM operand)
        dy_bias_done = False  # colsum(dy) already accumulated into this layer's two residual-branch bias gradients
        for i in range(L - 1, -1, -1):
            ext = dhidden[i + 1] if (i + 1) < min(len(dhidden), L) else None  # grad of hidden_states[i+1] = output of layer i
            if ext is not None:
                ext = ext.reshape(rows, h)
                if dx is not None and dy_bias_done:
                    # the LayerNorm backward above already added colsum(dx) to this layer's bias gradients: add the rest
                    ops.colsum_(ext.to(torch.float32).contiguous(), g(f"gpt_neox.layers.{i}.mlp.dense_4h_to_h.bias"))
                    ops.colsum_(ext.to(torch.float32).contiguous(), g(f"gpt_neox.layers.{i}.attention.dense.bias"))
                dx = ext.to(torch.float32) if dx is None else dx.add_(ext)
                dy = None
                main_moved()
            inj = inject.get(i) if inject else None
            if dx is None:
                # nothing flows into this layer's output (distillation of shallower layers only): its own backward is skipped,
                # but a distilled hidden_states[i] (this layer's input) still starts the gradient for the layers below
                if inj is not None:
                    dx = ops.distill_bwd(sv["layers"][i]["x"].view(B, S, h), inj[0], am, P, inj[1], cosine=inj_cos).view(rows, h)
                    main_moved()
                if overwrite:
                    self._zero_layer_matrices([i])   # (no weight-gradient GEMM will write this layer's matrices in this sweep)
                    main_moved()
                continue
            if dy is None:
                dy = dx if cd == torch.float32 else ops.cast(dx, cd)
                main_moved()
            pre = f"gpt_neox.layers.{i}."
            s = sv["layers"][i]
            # parameter gradients that only need dy: MLP down-projection and attention output projection
            wgrad_layer(dy, s["a"], pre + "mlp.dense_4h_to_h.weight", None if dy_bias_done else pre + "mlp.dense_4h_to_h.bias")
            wgrad_layer(dy, s["ao"], pre + "attention.dense.weight", None if dy_bias_done else pre + "attention.dense.bias")
            # MLP branch
            # (the bias gradients of the two up-projections are column sums of du / dqkv: folded into the producing kernels)
            du = ops.gemm(dy, w(pre + "mlp.dense_4h_to_h.weight"), False, False, epilogue=EPI_GELU_BWD, aux=s["u"],
                          colsum=g(pre + "mlp.dense_h_to_4h.bias"))
            main_moved()
            wgrad_layer(du, s["ln2"], pre + "mlp.dense_h_to_4h.weight")
            dln2 = ops.gemm(du, w(pre + "mlp.dense_h_to_4h.weight"), False, False)
            # attention branch
            dao = ops.gemm(dy, w(pre + "attention.dense.weight"), False, False)
            dqkv = ops.attn_bwd(s["qkv"], s["ao"], dao, s["lse"], B, S, H, D, rot, cos, sin, am,
                                colsum=g(pre + "attention.query_key_value.bias"))
            main_moved()
            wgrad_layer(dqkv, s["ln1"], pre + "attention.query_key_value.weight")
            dln1 = ops.gemm(dqkv, w(pre + "attention.query_key_value.weight"), False, False)
            # both LayerNorms + the residual path, one pass; also emits the compute-dtype copy the next layer's GEMMs read
            ln_kw = dict(want_lp=(cd != torch.float32), teacher=inj[0].view(rows, h) if inj is not None else None,
                         attention_mask=am if inj is not None else None, S=S, P=P, inj_scale=inj[1] if inj is not None else None,
                         inj_mul=-1.0 if inj_cos else 2.0 / h)   # (a negative factor selects the cosine-distance gradient, mafed_hip.h)
            dxa = g(f"gpt_neox.layers.{i - 1}.mlp.dense_4h_to_h.bias") if i > 0 else None
            dxb = g(f"gpt_neox.layers.{i - 1}.attention.dense.bias") if i > 0 else None
            if defer_ln:
                # row kernel on the dX chain; the slab reduction into the LayerNorm / bias gradients goes to a side stream (it feeds
                # parameter gradients only, and on the main stream the whole chip waited for it once per layer)
                dx, dy, ln_ws = ops.layernorm_bwd_rows(dln1, dln2, s["x"], s["mean"], s["rstd"], self._p(pre + "input_layernorm.weight"),
                                                       self._p(pre + "post_attention_layernorm.weight"), dx, want_dxsum=i > 0, **ln_kw)
                main_moved()
                on_side(lambda ws=ln_ws, pre=pre, dxa=dxa, dxb=dxb: ops.layernorm_bwd_params(
                    ws, rows, h, g(pre + "input_layernorm.weight"), g(pre + "input_layernorm.bias"),
                    g(pre + "post_attention_layernorm.weight"), g(pre + "post_attention_layernorm.bias"), dxa, dxb), ln_ws)
            else:
                dx, dy = ops.layernorm_bwd(dln1, dln2, s["x"], s["mean"], s["rstd"], self._p(pre + "input_layernorm.weight"),
                                           self._p(pre + "post_attention_layernorm.weight"), dx,
                                           g(pre + "input_layernorm.weight"), g(pre + "input_layernorm.bias"),
                                           g(pre + "post_attention_layernorm.weight"), g(pre + "post_attention_layernorm.bias"),
                                           dxsum_a=dxa, dxsum_b=dxb, **ln_kw)
                main_moved()
            dy_bias_done = i > 0
            if cd == torch.float32:
                dy = dx
            if taps is not None and i in taps:
                taps[i] = dx  # = dL/d hidden_states[i] (fresh buffer, never written again on this path)
            if group_dw:
                pending_layers.append(i)
                if len(pending_layers) >= int(self.dw_group_layers):
                    flush_dw()
            else:
                ready(i)
        flush_dw()
        # every layer's LayerNorm / distillation kernel -- the last readers of the teacher's hidden states -- is queued: a consumer
        # that only has to stay behind THOSE (the next step's teacher forward re-uses that memory) can wait for this event instead of
        # for the whole backward, whose side streams still carry ~0.3 ms of parameter-gradient tail
        self.dx_chain_event = main.record_event()
        ext0 = dhidden[0] if len(dhidden) > 0 else None
        if ext0 is not None:
            ext0 = ext0.reshape(rows, h)
            dx = ext0.to(torch.float32).contiguous() if dx is None else dx.add_(ext0)
            main_moved()
        if dx is not None:
            fc, u0, a0 = sv["proj"]
            dimg = ops.embed_concat_bwd(dx, sv["input_ids"], B, P, T, h, cfg.vocab_size, g("gpt_neox.embed_in.weight"), cd)
            main_moved()
            wgrad(dimg, a0, "vision_embed_tokens.2.weight", "vision_embed_tokens.2.bias")
            du0 = ops.gemm(dimg, w("vision_embed_tokens.2.weight"), False, False, epilogue=EPI_GELU_BWD, aux=u0,
                           colsum=g("vision_embed_tokens.0.bias"))
            main_moved()
            wgrad(du0, fc, "vision_embed_tokens.0.weight")
        ready(-1)
        if sides is not None:
            for st in sides:
                main.wait_stream(st)  # gradients complete (and `keep` safe to release) from the main stream's point of view
        keep.clear()


class _DecodeCache:
    """K/V cache of a greedy decode: per layer the prefill's [B*S0, 3*H*D] fused-QKV output (kept as written -- no split, no
    transpose, k un-rotated) and a [B, cap, 3*H*D] tensor that receives one row per generated token."""

    def __init__(self, model, prefix, B: int, S0: int, cap: int, attention_mask: torch.Tensor, prerotate: bool = True, fused: bool = True,
                 prefix_storage: Optional[torch.Tensor] = None):
        self.prefix, self.B, self.S0, self.cap, self.attention_mask = prefix, B, S0, max(1, cap), attention_mask
        self.prefix_storage = prefix_storage   # [L, B*S0, 3h] holding every entry of `prefix` (then one rotation launch serves all layers)
        cfg = model.config
        n = 3 * cfg.num_attention_heads * cfg.head_dim
        self.new = [torch.zeros((B, self.cap, n), dtype=prefix[0].dtype, device=prefix[0].device) for _ in prefix]
        # Pre-rotated cache (round 4): once the prefill's attention has read the un-rotated keys, rotate them in place -- every decode step
        # then loads k and v only (mafed_attn_decode_prerot).  Needs rot % 16 == 0 and an MFMA head size (every VLPythia preset).
        self.prerot = bool(prerotate) and cfg.rotary_ndims % 16 == 0 and cfg.head_dim in (64, 128, 256)
        self._model = model
        # fused decode layer (csrc/decode.hip): bf16 mode over the pre-rotated cache, shapes per mafed_decode_supported
        self.fused = bool(fused) and self.prerot and prefix[0].dtype == torch.bfloat16 and ops.decode_supported(B, cfg.hidden_size, cfg.intermediate_size)
        self.workspace = ops.decode_out_workspace(B, cfg.hidden_size, prefix[0].device) if self.fused else None
        # opt-in (`model.flow_decode = True`): the whole step as ONE launch whose work items hand over through arrival counters
        # (csrc/decode_flow.hip).  Correct and deterministic, but measured SLOWER than the three-launch layers (1.46 vs 0.88 ms per step at
        # 410M / B = 32): a hand-over through the memory side costs more than a kernel boundary in a graph
        self.flow = None
        if self.fused and getattr(model, "flow_decode", False) and ops.decode_flow_supported(
                B, cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.head_dim, cfg.vocab_size, S0 + self.cap):
            w, p = model._w, model._p
            recs = []
            for i in range(cfg.num_hidden_layers):
                pre = f"gpt_neox.layers.{i}."
                recs.append([p(pre + "input_layernorm.weight"), p(pre + "input_layernorm.bias"), p(pre + "post_attention_layernorm.weight"),
                             p(pre + "post_attention_layernorm.bias"), w(pre + "attention.query_key_value.weight"),
                             p(pre + "attention.query_key_value.bias"), w(pre + "mlp.dense_h_to_4h.weight"), p(pre + "mlp.dense_h_to_4h.bias"),
                             w(pre + "attention.dense.weight"), p(pre + "attention.dense.bias"), w(pre + "mlp.dense_4h_to_h.weight"),
                             p(pre + "mlp.dense_4h_to_h.bias"), prefix[i], self.new[i]])
            recs.append([p("gpt_neox.final_layer_norm.weight"), p("gpt_neox.final_layer_norm.bias"), None, None, w("embed_out.weight")] + [None] * 9)
            self.flow = ops.DecodeFlow(recs, B, cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.head_dim, cfg.vocab_size,
                                       prefix[0].device)
        # opt-in (`model.pair_decode = True`): two launches per layer -- the strips, then attention + [dense|fc2] as one grid with one
        # hand-over.  Also correct and also slower than the three-launch layers (1.13 vs 0.88 ms per step)
        self.pair = None
        if self.fused and self.flow is None and getattr(model, "pair_decode", False) and ops.decode_flow_supported(
                B, cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.head_dim, 16, S0 + self.cap):
            w, p = model._w, model._p
            recs = []
            for i in range(cfg.num_hidden_layers):
                pre = f"gpt_neox.layers.{i}."
                recs.append([None] * 8 + [w(pre + "attention.dense.weight"), p(pre + "attention.dense.bias"), w(pre + "mlp.dense_4h_to_h.weight"),
                                          p(pre + "mlp.dense_4h_to_h.bias"), prefix[i], self.new[i]])
            self.pair = ops.DecodeAttnOut(recs, B, cfg.hidden_size, cfg.intermediate_size, cfg.num_attention_heads, cfg.head_dim, self.cap,
                                          prefix[0].device)
        if self.prerot:
            self.rotate_prefix()

    def rotate_prefix(self) -> None:
        """Rotate the prefix keys in place (call once per prefill: the prefix must hold what the QKV GEMMs wrote)."""
        cfg = self._model.config
        cos, sin = self._model.rotary_tables(self.S0 + self.cap)
        whole = self.prefix_storage   # every layer's prefix in ONE tensor: one launch for all
        if whole is not None:
            ops.rotate_k_rows_(whole, whole.shape[0] * self.B, self.S0, cfg.num_attention_heads, cfg.head_dim, cfg.rotary_ndims, cos, sin)
            return
        for p in self.prefix:
            ops.rotate_k_rows_(p, self.B, self.S0, cfg.num_attention_heads, cfg.head_dim, cfg.rotary_ndims, cos, sin)


class _GraphedDecode:
    """Greedy decode steps 1 .. max_new-1 for one (B, T, max_new) shape as a single hipGraph.  Static buffers: the per-layer
    K/V cache (prefix written by the prefill's QKV GEMMs through ``qkv_out``, plus the per-token rows), the prompt mask, the
    prefill's last-position logits, the ``unfinished`` flags and the generated tokens."""

    def __init__(self, model, B: int, T: int, max_new: int, eos_token_id, pad_token_id):
        cfg = model.config
        dev, cd = model.flat_params.device, model.compute_dtype
        S0 = cfg.num_vision_tokens + T
        n = 3 * cfg.num_attention_heads * cfg.head_dim
        self.model, self.B, self.T, self.S0, self.max_new = model, B, T, S0, max_new
        self.prefix_storage = torch.empty((cfg.num_hidden_layers, B * S0, n), dtype=cd, device=dev)   # [L, B*S0, 3h]: rotated by one launch
        self.prefix = list(self.prefix_storage.unbind(0))
        self.am = torch.ones((B, T), dtype=torch.int64, device=dev)
        self.first_logits = torch.zeros((B, cfg.vocab_size), dtype=cd if cd != torch.float32 else torch.float32, device=dev)
        self.tokens = torch.zeros((B, max_new), dtype=torch.int64, device=dev)
        model.rotary_tables(S0 + max(1, max_new))  # built (host -> device copy) before the capture, not inside it
        self.cache = _DecodeCache(model, self.prefix, B, S0, max_new, self.am, fused=getattr(model, "fused_decode", True),
                                  prefix_storage=self.prefix_storage)   # (rotates the still-empty prefix once: harmless)
        eos, pad = eos_token_id, pad_token_id

        def body():
            unfinished = torch.ones(B, dtype=torch.int64, device=dev)
            logits = self.first_logits
            for t in range(max_new):
                nxt = logits.float().argmax(dim=-1)
                if eos is not None:
                    nxt = nxt * unfinished + pad * (1 - unfinished)
                    unfinished = unfinished * (nxt != eos).to(torch.int64)
                self.tokens[:, t] = nxt
                if t + 1 < max_new:
                    logits = model._engine_decode_step(nxt, t, self.cache)

        # one eager pass on a side stream (lazy initialisations must not happen inside the capture), then the capture
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            body()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            body()

    def run(self, feats, ids, am) -> torch.Tensor:
        m = self.model
        st = m._engine_forward(feats, ids, am, None, False, train=False, qkv_out=self.prefix, last_only=True)
        if self.cache.prerot:
            self.cache.rotate_prefix()   # this prefill's keys, rotated in place for the captured steps
        self.am.copy_(am)
        self.first_logits.copy_(st["logits"][:, -1, :])
        self.graph.replay()
        return self.tokens.clone()


class _ModelFn(torch.autograd.Function):
    """The whole model as one autograd node: outputs (loss, logits, *hidden_states)."""

    @staticmethod
    def forward(ctx, anchor, model: VLPythiaForCausalLM, feats, input_ids, attention_mask, labels, want_hidden, ctx_box, label_rows_hint=None):
        sv = model._engine_forward(feats, input_ids, attention_mask, labels, want_hidden, train=True, label_rows_hint=label_rows_hint)
        ctx.model, ctx.sv = model, sv
        ctx_box.append(sv)
        ctx.set_materialize_grads(False)  # outputs nobody differentiated arrive as None, not as zero tensors
        # (a 0-dim view of the CE kernel's own output: a clone here was a device copy on the chain between forward and backward)
        loss = sv["loss"].reshape(()) if sv["loss"] is not None else torch.zeros((), device=anchor.device)
        # outs[2] is a 0-dim "hook": the fused distillation node takes it as an input so that this node's backward runs
        # (after the distillation node has left its per-layer coefficients in sv["inject"]) even without a CE gradient; nobody reads
        # its value, so it is not filled
        pub = sv["logits"] if sv.get("sparse_head") is None else torch.empty(0, device=anchor.device)   # compact logits are internal
        # (one cached zero per device: an uninitialised scalar may hold NaN / Inf, which trips anomaly detection and would propagate if
        #  autograd ever accumulated the hook's gradient with another path; re-using the tensor costs no fill kernel per step)
        outs = [loss, pub.detach(), model._hook_zero().view(())]
        ctx.mark_non_differentiable(outs[1])
        if want_hidden:
            outs += [x.detach() for x in sv["hidden"]]  # aliases: no reference cycle through ctx
        return tuple(outs)

    @staticmethod
    def backward(ctx, dloss, dlogits, dhook, *dhidden):
        sv = ctx.sv
        ctx.sv = None
        if sv is None:
            raise RuntimeError("mafed_amd: backward through the model twice (activations already released)")
        if sv["loss"] is None:
            dloss = None
        ctx.model._engine_backward(sv, dloss, list(dhidden))
        sv.pop("inject", None)
        return (None,) * 9


model_architecture = {"vlpythia": VLPythiaForCausalLM}
