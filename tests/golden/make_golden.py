#!/usr/bin/env python3
"""Generate tests/golden/*.npz — the vectors that pin oracle/vit_oracle.c.

The reference (LimpBunion22/VIT-FPGA) holds no ViT, no tests and no golden vectors
(SURVEY.md §8c), so "parity" cannot be pinned by it.  This script pins the oracle with an
INDEPENDENT implementation instead: `transformers.ViTForImageClassification`, constructed
from a local config (no hub access, HF_HUB_OFFLINE=1), loaded with this repo's seeded
synthetic weights, evaluated in float64 and float32 on seeded synthetic images.

It runs only in the build container (needs torch + transformers); the GPU box and the test
suite use only the small .npz files it writes.  Re-run:  python tests/golden/make_golden.py
"""
import os
import sys

os.environ.setdefault("HF_HUB_OFFLINE", "1")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np
import torch
from transformers import ViTConfig, ViTForImageClassification

import vh_synth as S

CASES = [  # (config name, weight seed, image seed, batch)
    ("vit_micro", 11, 12, 3),
    ("vit_mini", 21, 22, 2),
    ("vit_tiny", 0, 1, 2),
    ("vit_base", 0, 1, 1),
]


def hf_model(cfg, tensors, dtype):
    hc = ViTConfig(hidden_size=cfg["dim"], num_hidden_layers=cfg["layers"],
                   num_attention_heads=cfg["heads"], intermediate_size=cfg["mlp_dim"],
                   image_size=cfg["image_size"], patch_size=cfg["patch_size"],
                   num_channels=cfg["channels"], layer_norm_eps=1e-6, hidden_act="gelu",
                   qkv_bias=True, num_labels=cfg["classes"], hidden_dropout_prob=0.0,
                   attention_probs_dropout_prob=0.0)
    hc._attn_implementation = "eager"
    m = ViTForImageClassification(hc).eval()
    sd = {}
    t = {k: torch.from_numpy(v.copy()) for k, v in tensors.items()}
    D, T = cfg["dim"], S.tokens(cfg)
    sd["vit.embeddings.cls_token"] = t["cls"].reshape(1, 1, D)
    sd["vit.embeddings.position_embeddings"] = t["pos"].reshape(1, T, D)
    sd["vit.embeddings.patch_embeddings.projection.weight"] = t["patch.weight"]
    sd["vit.embeddings.patch_embeddings.projection.bias"] = t["patch.bias"]
    for l in range(cfg["layers"]):
        p = f"vit.layers.{l}."
        for ours, theirs in (("q", "attention.q_proj"), ("k", "attention.k_proj"),
                             ("v", "attention.v_proj"), ("o", "attention.o_proj"),
                             ("ln1", "layernorm_before"), ("ln2", "layernorm_after"),
                             ("fc1", "mlp.fc1"), ("fc2", "mlp.fc2")):
            sd[p + theirs + ".weight"] = t[f"l{l}.{ours}.weight"]
            sd[p + theirs + ".bias"] = t[f"l{l}.{ours}.bias"]
    sd["vit.layernorm.weight"] = t["lnf.weight"]
    sd["vit.layernorm.bias"] = t["lnf.bias"]
    sd["classifier.weight"] = t["head.weight"]
    sd["classifier.bias"] = t["head.bias"]
    missing, unexpected = m.load_state_dict(sd, strict=True), None
    return m.to(dtype)


def main():
    torch.set_num_threads(8)
    for name, wseed, iseed, batch in CASES:
        cfg = S.CONFIGS[name]
        tensors = S.make_tensors(cfg, wseed)
        images = S.make_images(cfg, iseed, batch)            # NHWC fp32
        nchw = torch.from_numpy(images.transpose(0, 3, 1, 2).copy())
        out = {}
        for tag, dt in (("f64", torch.float64), ("f32", torch.float32)):
            m = hf_model(cfg, tensors, dt)
            with torch.no_grad():
                r = m(pixel_values=nchw.to(dt), output_hidden_states=True)
            out[f"logits_{tag}"] = r.logits.to(torch.float64).numpy()
            hs = r.hidden_states
            # hidden_states[i] = residual stream entering layer i; [-1] = after last layer
            out[f"hidden_last_{tag}"] = hs[-1].to(torch.float64).numpy().reshape(-1, cfg["dim"])[:64]
            out[f"hidden_l1_{tag}"] = hs[1].to(torch.float64).numpy().reshape(-1, cfg["dim"])[:64]
            out[f"embed_{tag}"] = hs[0].to(torch.float64).numpy().reshape(-1, cfg["dim"])[:64]
        # checksums that let the test detect a drifted generator without storing the weights
        out["weights_checksum"] = np.array(
            [float(np.float64(v.astype(np.float64).sum())) for v in tensors.values()][:8])
        out["images_checksum"] = np.array([float(images.astype(np.float64).sum())])
        meta = np.array([wseed, iseed, batch], dtype=np.int64)
        path = os.path.join(HERE, f"{name}_s{wseed}_i{iseed}_b{batch}.npz")
        np.savez_compressed(path, meta=meta, **{k: (v.astype(np.float32) if k.endswith("f32") else v)
                                                for k, v in out.items()})
        print(name, "logits f64[0,:4] =", out["logits_f64"][0, :4], "->", os.path.basename(path),
              os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
