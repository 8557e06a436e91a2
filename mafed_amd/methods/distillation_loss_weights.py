"""Layer coefficients and modality weights of MAFED (reference: mafed/methods/distillation_loss_weights.py).

Per-step getters return DEVICE tensors / Python floats without host synchronisation; the whole layer loop is
vectorised by ``FeatureDistillation`` (one [n_layers] coefficient vector instead of n_layers scalar look-ups).
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch


class DistillationWeights:
    def __init__(self, distillation_modality_weighing_strategy: str = "equal",
                 distillation_layer_weighing_strategy: str = "single", gamma: float = 0.9, num_hidden_layers: int = 11,
                 distillation_layer: Optional[int] = -1, num_vision_tokens: int = 256) -> None:
        self.gamma = gamma
        self._dev_cache = {}
        self.num_vision_tokens = num_vision_tokens
        self._hidden_state_layer = distillation_layer
        self._modality_weighing_strategy = distillation_modality_weighing_strategy
        if distillation_modality_weighing_strategy == "balanced":
            self.lang_coeff = 0.5
        strategy = distillation_layer_weighing_strategy
        # same assertion texts as the reference (distillation_loss_weights.py:33-36)
        if distillation_layer is None and strategy == "single":
            raise AssertionError("Invalid layer weighting strategy 'single'. Use 'equal' or 'discounted' instead!")
        if distillation_layer is None and strategy == "cumulative":
            raise AssertionError("Invalid layer weighting strategy 'cumulative'. Please pass the distillation layer!")
        self.num_hidden_layers = distillation_layer if strategy == "cumulative" else num_hidden_layers
        if distillation_layer is not None and strategy != "cumulative":
            strategy = "single"  # a concrete layer wins over equal / discounted
        self._layer_weighing_strategy = strategy
        self.prepare_layer_coeffs()

    # ---- layers ---------------------------------------------------------------------------------------------------
    def prepare_layer_coeffs(self) -> None:
        """single: no vector (weight 1); equal: 1/n; discounted / cumulative: gamma^(n-l) normalised (deepest weighs most)."""
        n = self.num_hidden_layers
        if self._layer_weighing_strategy == "single":
            self.layer_coeffs = None
        elif self._layer_weighing_strategy == "equal":
            self.layer_coeffs = torch.ones(n) / n
        else:
            c = torch.tensor([self.gamma ** d for d in torch.arange(n, 0, -1)])
            self.layer_coeffs = c / c.sum()

    def get_distillation_layers(self) -> List[int]:
        if self._layer_weighing_strategy == "single":
            return [self._hidden_state_layer]
        return list(range(self.num_hidden_layers))

    def get_layer_loss_weight(self, layer: int):
        if self.layer_coeffs is None or self._layer_weighing_strategy == "single":
            return 1.0
        return self.layer_coeffs[layer]

    def layer_coeff_vector(self, device) -> torch.Tensor:
        """Coefficients of get_distillation_layers(), as one fp32 device vector (uploaded once: the per-step path must
        not copy from the host, it is replayed from a hipGraph)."""
        key = ("coeff", str(device))
        v = self._dev_cache.get(key)
        if v is None:
            layers = self.get_distillation_layers()
            if self.layer_coeffs is None or self._layer_weighing_strategy == "single":
                v = torch.ones(len(layers), dtype=torch.float32, device=device)
            else:
                v = self.layer_coeffs[layers].to(device=device, dtype=torch.float32)
            self._dev_cache[key] = v
        return v

    # ---- modalities -------------------------------------------------------------------------------------------------
    def get_modality_loss_weights(self, batch, layer: int):
        s = self._modality_weighing_strategy
        if s == "equal":
            nt, nv = batch["lang_masks"].sum(), batch["image_masks"].sum()
            return nt / (nt + nv), nv / (nt + nv)
        if s == "balanced":
            return self.lang_coeff, 1 - self.lang_coeff
        if s == "adaptive":
            lc = self.lang_coeff
            lw = lc[0] if lc.shape[0] == 1 else lc[layer]
            return lw, 1 - lw
        raise NotImplementedError

    def modality_weight_vectors(self, n_lang: torch.Tensor, n_vision: torch.Tensor, layers: Sequence[int], device):
        """(lang_w[n_layers], vision_w[n_layers]) as device tensors; counts come from the distillation kernel."""
        s = self._modality_weighing_strategy
        nl = len(layers)
        if s == "equal":
            lw = (n_lang / (n_lang + n_vision)).expand(nl)
        elif s == "balanced":
            key = ("balanced", str(device), nl, float(self.lang_coeff))
            lw = self._dev_cache.get(key)
            if lw is None:
                lw = self._dev_cache[key] = torch.full((nl,), float(self.lang_coeff), dtype=torch.float32, device=device)
        elif s == "adaptive":
            key = ("adaptive", str(device), tuple(layers), id(self.lang_coeff))
            lw = self._dev_cache.get(key)
            if lw is None:
                lc = torch.as_tensor(self.lang_coeff, dtype=torch.float32).reshape(-1).to(device)
                lw = lc.expand(nl) if lc.shape[0] == 1 else lc[torch.as_tensor(list(layers), device=device)]
                self._dev_cache[key] = lw = lw.contiguous()
        else:
            raise NotImplementedError
        return lw, 1.0 - lw

    def modality_mode(self, layers: Sequence[int], device):
        """(mode, lang_weight, lang_weight_vec) for ``mafed_distill_combine``: 0 "equal" (from the token counts), 1 "balanced"
        (constant), 2 "adaptive" (per-layer device vector) -- the same weights as ``modality_weight_vectors``."""
        s = self._modality_weighing_strategy
        if s == "equal":
            return 0, 0.0, None
        if s == "balanced":
            return 1, float(self.lang_coeff), None
        if s == "adaptive":
            lw, _ = self.modality_weight_vectors(None, None, layers, device)
            return 2, 0.0, lw
        raise NotImplementedError

    def update_weights(self, model, dataloader, task_id) -> None:
        """Between tasks; only the adaptive strategy has state (running mean over tasks)."""
        if self._modality_weighing_strategy != "adaptive":
            return
        imp = self.compute_adaptive_weights(model, dataloader)
        if task_id < 1:
            self.lang_coeff = imp
        else:
            self.lang_coeff = (imp + task_id * self.lang_coeff) / (task_id + 1)

    def compute_adaptive_weights(self, model, dataloader) -> torch.Tensor:
        """Gradient-norm importance of each modality per distilled layer (distillation_loss_weights.py:91-146).

        The reference runs one ``autograd.grad`` sweep per layer with ``retain_graph``; here ONE backward sweep taps the
        residual-stream gradient at every layer boundary (``model.hidden_grad_taps``)."""
        layers = self.get_distillation_layers()
        was_training = model.training
        model.eval()
        dev = model.flat_params.device
        lang = torch.zeros(len(layers), dtype=torch.float32, device=dev)
        img = torch.zeros(len(layers), dtype=torch.float32, device=dev)
        n_lang = torch.zeros((), dtype=torch.float32, device=dev)
        n_img = torch.zeros((), dtype=torch.float32, device=dev)
        P = self.num_vision_tokens
        for batch in dataloader:
            model.zero_grad()
            am = batch["attention_mask"].to(dev)
            B, T = am.shape
            taps = model.hidden_grad_taps(batch, layers)  # {layer: dL/dhidden[layer]  [B,S,h]}
            lm = torch.cat([torch.zeros(B, P, device=dev), am.float()], dim=1)
            im = torch.cat([torch.ones(B, P, device=dev), torch.zeros(B, T, device=dev)], dim=1)
            for k, l in enumerate(layers):
                gn = torch.linalg.norm(taps[l].float(), dim=-1)
                lang[k] += (gn * lm).sum()
                img[k] += (gn * im).sum()
            n_lang += lm.sum()
            n_img += im.sum()
        lang /= n_lang
        img /= n_img
        lang /= lang + img
        model.zero_grad()
        model.train(was_training)
        return lang
