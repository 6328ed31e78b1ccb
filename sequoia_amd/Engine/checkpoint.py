"""Checkpoint sources for LlamaWeights: a HF-style state dict in memory, or a HF checkpoint DIRECTORY streamed from disk.

The reference's engines start from `from_pretrained` (Engine/Engine.py:18,74,81; Engine/offload_engine.py:268-300), which
materialises the whole model per process.  Here every rank reads exactly the slices it owns -- `safe_open(...).get_slice(name)
[rows, cols]` per tensor -- so a tensor-parallel job never holds more than one fused layer tensor of its own shard on the
host (Llama-2-70B at TP = 8: 17 GB of shard per rank instead of 8 x 138 GB of state dicts), and a single-GPU load streams
tensor by tensor into device memory.

Layout accepted: `config.json` + `*.safetensors` (one file, or shards with or without `model.safetensors.index.json`);
`lm_head.weight` may be absent (tied to `model.embed_tokens.weight`).  Parameter names are the reference's
(LlamaForCausalLM_FI / _TG, Engine/Llama_model.py:136-300 == HF Llama).
"""
from __future__ import annotations

import glob
import json
import os

import torch


class StateDictSource:
    """Tensors already in memory (HF parameter names)."""

    def __init__(self, sd):
        self.sd = sd

    def has(self, name) -> bool:
        return name in self.sd

    def shape(self, name):
        return tuple(self.sd[name].shape)

    def full(self, name):
        return torch.as_tensor(self.sd[name])

    def rows(self, name, r0, r1):
        return torch.as_tensor(self.sd[name])[r0:r1]

    def cols(self, name, c0, c1):
        return torch.as_tensor(self.sd[name])[:, c0:c1]

    def close(self):
        pass


class CheckpointDirSource:
    """A HF checkpoint directory read slice by slice (safetensors: the header is parsed once per file, tensor bytes are read
    on demand).  `bytes_read` counts what this process actually pulled from disk (tests assert a rank reads its shard only)."""

    def __init__(self, path: str):
        self.path = path
        files = sorted(glob.glob(os.path.join(path, "*.safetensors")))
        if not files:
            raise FileNotFoundError(f"no *.safetensors under {path}")
        try:
            from safetensors import safe_open
        except ImportError as e:                                   # pragma: no cover
            raise RuntimeError("loading a checkpoint directory needs the `safetensors` package") from e
        self._safe_open = safe_open
        self._handles: dict = {}
        self.where: dict = {}
        index = os.path.join(path, "model.safetensors.index.json")
        if os.path.exists(index):
            with open(index) as f:
                for name, fname in json.load(f)["weight_map"].items():
                    self.where[name] = os.path.join(path, fname)
        else:
            for fpath in files:                                     # header only: no tensor data is touched
                for name in self._handle(fpath).keys():
                    self.where[name] = fpath
        self.bytes_read = 0

    def _handle(self, fpath):
        h = self._handles.get(fpath)
        if h is None:
            h = self._safe_open(fpath, framework="pt", device="cpu")
            h.__enter__()
            self._handles[fpath] = h
        return h

    def _slice(self, name):
        if name not in self.where:
            raise KeyError(f"{name} is not in the checkpoint {self.path}")
        return self._handle(self.where[name]).get_slice(name)

    def has(self, name) -> bool:
        return name in self.where

    def shape(self, name):
        return tuple(self._slice(name).get_shape())

    def _count(self, t):
        self.bytes_read += t.numel() * t.element_size()
        return t

    def full(self, name):
        return self._count(self._slice(name)[:])

    def rows(self, name, r0, r1):
        return self._count(self._slice(name)[r0:r1])

    def cols(self, name, c0, c1):
        return self._count(self._slice(name)[:, c0:c1])

    def close(self):
        for h in self._handles.values():
            h.__exit__(None, None, None)
        self._handles.clear()


def read_config(path: str) -> dict:
    with open(os.path.join(path, "config.json")) as f:
        return json.load(f)


def save_checkpoint_dir(state_dict, config: dict, path: str, n_shards: int = 2, tie_lm_head: bool = False, index: bool = True):
    """Write a HF-style directory (tests, and exporting the synthetic pairs): config.json + n_shards safetensors files
    (+ the index json).  tie_lm_head: leave `lm_head.weight` out (it must equal the embedding)."""
    from safetensors.torch import save_file
    os.makedirs(path, exist_ok=True)
    sd = {k: torch.as_tensor(v).contiguous() for k, v in state_dict.items() if "rotary" not in k and "inv_freq" not in k}
    if tie_lm_head:
        assert torch.equal(sd["lm_head.weight"], sd["model.embed_tokens.weight"])
        del sd["lm_head.weight"]
    with open(os.path.join(path, "config.json"), "w") as f:
        json.dump(dict(config, tie_word_embeddings=bool(tie_lm_head)), f)
    names = sorted(sd)
    weight_map = {}
    for i in range(n_shards):
        part = {k: sd[k] for k in names[i::n_shards]}
        fname = f"model-{i + 1:05d}-of-{n_shards:05d}.safetensors" if n_shards > 1 else "model.safetensors"
        save_file(part, os.path.join(path, fname))
        weight_map.update({k: fname for k in part})
    if index and n_shards > 1:
        with open(os.path.join(path, "model.safetensors.index.json"), "w") as f:
            json.dump(dict(metadata={}, weight_map=weight_map), f)
    return path
