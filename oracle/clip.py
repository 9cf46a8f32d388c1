"""CPU oracle (test infrastructure) for the text encoder call of the walk:

    self.text_encoder(text_input.input_ids.to(self.device))[0]
        /root/reference/stable_diffusion_videos/stable_diffusion_pipeline.py:819 (embed_text), :306, :348

``text_encoder`` is ``transformers.CLIPTextModel`` (third-party; ``pyproject.toml`` lists ``transformers``
unpinned).  Unlike diffusers, transformers IS installed in the build image (5.15.0), so this restatement is
**pinned**: ``tests/golden/make_golden_clip.py`` runs the real ``CLIPTextModel`` on seeded weights / ids and commits
its ``last_hidden_state`` (``tests/golden/clip_*.npz``); ``tests/test_oracle.py`` checks this file against those
vectors and - when transformers can be imported - against the live model.

Algorithm (CLIP text transformer, Radford et al. 2021; pre-LN):
    x = token_embedding[ids] + position_embedding[0..L)
    per layer:  h = LN1(x); q,k,v = Linear(h); a = softmax(q k^T / sqrt(dh) + causal_mask) v; x = x + Linear(a)
                h = LN2(x); x = x + fc2(act(fc1(h)))            act = quick_gelu (ViT-L/14) or gelu (OpenCLIP-H)
    last_hidden_state = final_layer_norm(x)          (eps 1e-5 everywhere)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F


def strip_prefix(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """transformers < 5 prefixes every key with ``text_model.``; 5.x does not."""
    return {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in sd.items()}


def _act(name: str, x: torch.Tensor) -> torch.Tensor:
    if name == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    if name == "gelu":
        return F.gelu(x)
    raise ValueError(f"unsupported hidden_act {name}")


@torch.no_grad()
def clip_text_forward(sd: Dict[str, torch.Tensor], ids: torch.Tensor, num_heads: int, hidden_act: str = "quick_gelu",
                      eps: float = 1e-5) -> torch.Tensor:
    """``CLIPTextModel(ids)[0]`` in fp32 from a state dict.  ids: int64 [B, L] -> [B, L, D]."""
    sd = {k: v.float() for k, v in strip_prefix(sd).items() if v.is_floating_point()}
    B, L = ids.shape
    tok, pos = sd["embeddings.token_embedding.weight"], sd["embeddings.position_embedding.weight"]
    D = tok.shape[1]
    dh = D // num_heads
    x = tok[ids] + pos[:L][None]
    mask = torch.full((L, L), float("-inf")).triu(1)
    n_layers = 1 + max(int(k.split(".")[2]) for k in sd if k.startswith("encoder.layers."))
    for i in range(n_layers):
        p = f"encoder.layers.{i}."
        h = F.layer_norm(x, (D,), sd[p + "layer_norm1.weight"], sd[p + "layer_norm1.bias"], eps)
        q = F.linear(h, sd[p + "self_attn.q_proj.weight"], sd[p + "self_attn.q_proj.bias"])
        k = F.linear(h, sd[p + "self_attn.k_proj.weight"], sd[p + "self_attn.k_proj.bias"])
        v = F.linear(h, sd[p + "self_attn.v_proj.weight"], sd[p + "self_attn.v_proj.bias"])
        q, k, v = (t.view(B, L, num_heads, dh).transpose(1, 2) for t in (q, k, v))
        a = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(dh) + mask, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, L, D)
        x = x + F.linear(a, sd[p + "self_attn.out_proj.weight"], sd[p + "self_attn.out_proj.bias"])
        h = F.layer_norm(x, (D,), sd[p + "layer_norm2.weight"], sd[p + "layer_norm2.bias"], eps)
        h = _act(hidden_act, F.linear(h, sd[p + "mlp.fc1.weight"], sd[p + "mlp.fc1.bias"]))
        x = x + F.linear(h, sd[p + "mlp.fc2.weight"], sd[p + "mlp.fc2.bias"])
    return F.layer_norm(x, (D,), sd["final_layer_norm.weight"], sd["final_layer_norm.bias"], eps)
