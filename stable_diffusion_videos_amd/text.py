"""Text side of the walk path: ``tokenizer(...)`` and ``text_encoder(ids)[0]`` as the reference calls them
(stable_diffusion_pipeline.py:291-306, :340-348, :811-819).

The text encoder is on the path but not hot (2 forwards per clip + 1 per walk, ~13 GFLOP each), so it stays
the installed ``transformers.CLIPTextModel`` run through PyTorch-ROCm in fp32;
a native kernel for it is a section-8(f) follow-up.

Offline there are neither CLIP weights nor ``vocab.json`` / ``merges.txt``.  When a model directory with a
``tokenizer/`` sub-directory is given the real ``CLIPTokenizer`` is used; otherwise ``HashTokenizer``
produces deterministic, clearly synthetic ids (BOS + one id per whitespace word + EOS padding), which is all
the throughput and parity harness needs.
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


def build_text_encoder(cfg: TextConfig, model_dir: Optional[Path] = None, seed: int = 0):
    """``transformers.CLIPTextModel`` - real weights from ``<model_dir>/text_encoder`` when present,
    otherwise the same architecture with seeded random weights."""
    from transformers import CLIPTextConfig, CLIPTextModel
    if model_dir is not None and (Path(model_dir) / "text_encoder" / "config.json").exists():
        return CLIPTextModel.from_pretrained(str(Path(model_dir) / "text_encoder"), torch_dtype=torch.float32).eval()
    tc = CLIPTextConfig(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, intermediate_size=cfg.intermediate_size,
                        num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                        max_position_embeddings=cfg.max_position_embeddings, hidden_act=cfg.hidden_act,
                        bos_token_id=cfg.bos_token_id, eos_token_id=cfg.eos_token_id, pad_token_id=cfg.eos_token_id,
                        projection_dim=cfg.hidden_size)
    with torch.random.fork_rng(devices=[]):
        torch.manual_seed(seed)
        model = CLIPTextModel(tc)
    return model.float().eval()
