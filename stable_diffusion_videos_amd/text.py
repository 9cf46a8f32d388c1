"""Text side of the walk path: ``tokenizer(...)`` and ``text_encoder(ids)[0]`` as the reference calls them
(stable_diffusion_pipeline.py:291-306, :340-348, :811-819).

``CLIPTextEngine`` is the CLIP text transformer (``transformers.CLIPTextModel`` in the reference) on the HIP kernels:
token + position embedding gather, pre-LN blocks with the fused [Wq;Wk] projection, V computed transposed, the flash
attention kernel with its causal mask, quick_gelu / gelu fused into the fc1 epilogue, residual adds fused into the
out-proj / fc2 epilogues, final LayerNorm.  It is on the path but not hot (2 forwards per clip + 1 per walk,
~13 GFLOP each).  Same call shape as the reference's module: ``text_encoder(ids)[0]`` is ``last_hidden_state``.

Offline there are neither CLIP weights nor ``vocab.json`` / ``merges.txt``.  When a model directory with
``text_encoder/`` / ``tokenizer/`` sub-directories is given the real weights / ``CLIPTokenizer`` are used; otherwise
seeded synthetic weights and ``HashTokenizer`` (deterministic, clearly synthetic ids: BOS + one id per whitespace
word + EOS padding), which is all the throughput and parity harness needs.
"""
from __future__ import annotations

import hashlib
from pathlib import Path
from types import SimpleNamespace
from typing import List, Optional, Union

import torch

from .config import TextConfig


class HashTokenizer:
    """Stand-in for CLIPTokenizer with the same call shape (``padding="max_length"``, ``truncation=True``,
    ``return_tensors="pt"`` -> object with ``.input_ids``) and ``model_max_length``."""

    def __init__(self, cfg: TextConfig):
        self.model_max_length = cfg.max_position_embeddings
        self.bos, self.eos = cfg.bos_token_id, cfg.eos_token_id
        self.n_words = min(cfg.bos_token_id, cfg.eos_token_id)  # ids below the special tokens
        self.is_synthetic = True

    def _ids(self, text: str, max_length: int, truncation: bool) -> List[int]:
        words = text.lower().split()
        ids = [int.from_bytes(hashlib.sha256(w.encode()).digest()[:4], "little") % (self.n_words - 1) + 1 for w in words]
        ids = [self.bos] + ids + [self.eos]
        if truncation and len(ids) > max_length:
            ids = ids[: max_length - 1] + [self.eos]
        return ids + [self.eos] * (max_length - len(ids))

    def __call__(self, text: Union[str, List[str]], padding="max_length", max_length: Optional[int] = None,
                 truncation: bool = False, return_tensors="pt"):
        texts = [text] if isinstance(text, str) else list(text)
        max_length = max_length or self.model_max_length
        rows = [self._ids(t, max_length, truncation) for t in texts]
        width = max(len(r) for r in rows)
        rows = [r + [self.eos] * (width - len(r)) for r in rows]
        return SimpleNamespace(input_ids=torch.tensor(rows, dtype=torch.long))

    def batch_decode(self, ids):
        return ["<synthetic ids>" for _ in ids]


def load_tokenizer(model_dir: Optional[Path], cfg: TextConfig):
    if model_dir is not None and (Path(model_dir) / "tokenizer" / "vocab.json").exists():
        from transformers import CLIPTokenizer
        return CLIPTokenizer.from_pretrained(str(Path(model_dir) / "tokenizer"))
    return HashTokenizer(cfg)


class CLIPTextEngine:
    """``text_encoder(ids)[0]`` of the reference on the HIP kernels.  Weights are re-laid out at ``.to(device)``."""

    def __init__(self, cfg: TextConfig, state_dict):
        if cfg.hidden_size % cfg.num_attention_heads or cfg.hidden_size // cfg.num_attention_heads not in (40, 64, 80, 160):
            raise ValueError("CLIPTextEngine: head dim must be one of 40 / 64 / 80 / 160 (both SD text encoders use 64)")
        if cfg.hidden_act not in ("quick_gelu", "gelu"):
            raise ValueError(f"CLIPTextEngine: unsupported hidden_act {cfg.hidden_act}")
        self.config = cfg
        self.state_dict_ = state_dict
        self.device = torch.device("cpu")
        self._w = None

    def state_dict(self):
        return self.state_dict_

    def to(self, device):
        device = torch.device(device)
        if device.type == "cuda" and (self._w is None or device != self.device):
            from . import hip
            from .weights import lin_w, vec
            hip.load()
            sd, c = self.state_dict_, self.config
            w = {"tok": sd["embeddings.token_embedding.weight"].to(device, torch.float32).contiguous(),
                 "pos": sd["embeddings.position_embedding.weight"].to(device, torch.float32).contiguous(),
                 "fin": (vec(sd["final_layer_norm.weight"], device), vec(sd["final_layer_norm.bias"], device)), "layers": []}
            qs = hip.q_prescale(c.hidden_size // c.num_attention_heads)
            for i in range(c.num_hidden_layers):
                p = f"encoder.layers.{i}."
                a = p + "self_attn."
                w["layers"].append(dict(
                    ln1=(vec(sd[p + "layer_norm1.weight"], device), vec(sd[p + "layer_norm1.bias"], device)),
                    ln2=(vec(sd[p + "layer_norm2.weight"], device), vec(sd[p + "layer_norm2.bias"], device)),
                    # q / k / v as one projection [3D, D]; the attention kernel reads V row-major out of its output
                    wqkv=lin_w(torch.cat([sd[a + "q_proj.weight"], sd[a + "k_proj.weight"], sd[a + "v_proj.weight"]], 0), device),
                    # the Q third of the bias carries the softmax scale * log2(e) that the projection's alpha puts on Q
                    bqkv=vec(torch.cat([sd[a + "q_proj.bias"].float() * qs, sd[a + "k_proj.bias"].float(),
                                        sd[a + "v_proj.bias"].float()], 0), device),
                    wo=lin_w(sd[a + "out_proj.weight"], device), bo=vec(sd[a + "out_proj.bias"], device),
                    w1=lin_w(sd[p + "mlp.fc1.weight"], device), b1=vec(sd[p + "mlp.fc1.bias"], device),
                    w2=lin_w(sd[p + "mlp.fc2.weight"], device), b2=vec(sd[p + "mlp.fc2.bias"], device)))
            self._w = w
        self.device = device
        return self

    @torch.no_grad()
    def __call__(self, input_ids: torch.Tensor, *args, **kwargs):
        """input_ids int64 [B, L] -> (last_hidden_state fp32 [B, L, D],) - index 0 as the reference takes it (:819)."""
        from . import hip
        if self._w is None or not input_ids.is_cuda:
            raise hip.SdvHipError("CLIPTextEngine runs on the MI355X HIP path only (no CPU fallback): call .to('cuda') and "
                                  "pass GPU-resident ids")
        c, w = self.config, self._w
        ids = input_ids.to(torch.int64)
        B, L = ids.shape
        if L > c.max_position_embeddings:
            raise ValueError(f"sequence length {L} exceeds max_position_embeddings {c.max_position_embeddings}")
        if int(ids.min()) < 0 or int(ids.max()) >= c.vocab_size:
            raise IndexError(f"token id outside [0, {c.vocab_size})")
        D, H = c.hidden_size, c.num_attention_heads
        dh = D // H
        epi = 4 if c.hidden_act == "quick_gelu" else 5
        x = hip.embed_tokens(ids, w["tok"], w["pos"])                                          # [B*L, D]
        o = torch.empty_like(x)
        for lw in w["layers"]:
            h = hip.layernorm(x, *lw["ln1"], eps=1e-5)
            qkv = hip.linear(h, lw["wqkv"], lw["bqkv"], alpha=hip.q_prescale(dh), alpha_cols=D)   # [M, 3D] = [Q * qs | K | V]
            hip.attention(qkv, qkv, qkv, o, B=B, H=H, Lq=L, Lk=L, dh=dh, ldq=3 * D, ldk=3 * D, ldv=3 * D, ldo=D,
                          scale=dh ** -0.5, k_off=D, v_off=2 * D, causal=True, q_prescaled=True, v_rowmajor=True)
            x = hip.linear(o, lw["wo"], lw["bo"], residual=x)
            h = hip.layernorm(x, *lw["ln2"], eps=1e-5)
            f = hip.linear(h, lw["w1"], lw["b1"], epi=epi)
            x = hip.linear(f, lw["w2"], lw["b2"], residual=x)
        out = hip.layernorm(x, *w["fin"], eps=1e-5)
        return (out.float().view(B, L, D),)


def build_text_encoder(cfg: TextConfig, model_dir: Optional[Path] = None, seed: int = 0) -> CLIPTextEngine:
    """Real weights from ``<model_dir>/text_encoder`` when present (its ``config.json`` overrides ``cfg``), otherwise
    the same architecture with seeded synthetic weights."""
    from .weights import clip_text_shapes, load_clip_text, synthetic_state_dict
    if model_dir is not None and (Path(model_dir) / "text_encoder" / "config.json").exists():
        import json
        data = json.loads((Path(model_dir) / "text_encoder" / "config.json").read_text())
        data = data.get("text_config", data)
        cfg = TextConfig(**{k: data[k] for k in TextConfig.__dataclass_fields__ if k in data})
        return CLIPTextEngine(cfg, load_clip_text(model_dir, clip_text_shapes(cfg)))
    return CLIPTextEngine(cfg, synthetic_state_dict(clip_text_shapes(cfg), seed=seed))
