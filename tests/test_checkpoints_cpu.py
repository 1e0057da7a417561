"""CPU: checkpoint conventions of the reference loaders (FX.py:1190-1332, VAE.py:1059-1079) on the
drop-in classes: config.json + safetensors (single file and shards), yaml dict_mapping, zero-pad /
crop of patch_embedding in-channels, size-mismatched keys skipped; VAE .pth keys get 'model.'."""
import json
import os

import torch
from safetensors.torch import save_file

from oracle import cases as C
from oracle import dit as O


def _write_dit(tmp, cfg, sd, shards=1):
    conf = {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items() if k not in ("eps", "in_dim", "dim")}
    conf["in_channels_ckpt"], conf["hidden_size_ckpt"] = cfg["in_dim"], cfg["dim"]
    json.dump(conf, open(os.path.join(tmp, "config.json"), "w"))
    keys = sorted(sd)
    if shards == 1:
        save_file({k: sd[k].contiguous() for k in keys}, os.path.join(tmp, "diffusion_pytorch_model.safetensors"))
    else:
        for i in range(shards):
            save_file({k: sd[k].contiguous() for k in keys[i::shards]}, os.path.join(tmp, f"model-{i:05d}-of-{shards:05d}.safetensors"))


def test_dit_from_pretrained_roundtrip_and_mapping(tmp_path):
    from flexam_amd import Wan2_2Transformer3DModel_FlexAM
    cfg = dict(O.DIT_TINY)
    sd = C.dit_weights(cfg, 3)
    for shards in (1, 3):
        d = tmp_path / f"ckpt{shards}"
        d.mkdir()
        _write_dit(str(d), cfg, sd, shards)
        m = Wan2_2Transformer3DModel_FlexAM.from_pretrained(
            str(tmp_path), subfolder=f"ckpt{shards}", torch_dtype=torch.float32,
            transformer_additional_kwargs={"dict_mapping": {"in_channels_ckpt": "in_dim", "hidden_size_ckpt": "dim"},
                                           "add_ref_conv": True, "add_cnn_block": True, "in_dim_cnn_block": 288, "out_dim_cnn_block": 48})
        got = m.state_dict()
        assert set(got) == set(sd)
        for k in sd:
            torch.testing.assert_close(got[k], sd[k], rtol=0, atol=0)
    m16 = Wan2_2Transformer3DModel_FlexAM.from_pretrained(str(tmp_path / "ckpt1"), transformer_additional_kwargs={
        "dict_mapping": {"in_channels_ckpt": "in_dim", "hidden_size_ckpt": "dim"}})
    assert m16.dtype == torch.bfloat16                               # reference default torch_dtype (FX.py:1193,1331)


def test_dit_patch_embedding_pad_and_mismatch_skip(tmp_path):
    from flexam_amd import Wan2_2Transformer3DModel_FlexAM
    cfg = dict(O.DIT_TINY)
    sd = C.dit_weights(cfg, 4)
    base = dict(sd)
    base["patch_embedding.weight"] = sd["patch_embedding.weight"][:, :100].clone()      # checkpoint with fewer in-channels
    base["head.head.bias"] = torch.zeros(7)                                             # wrong size: must be skipped
    _write_dit(str(tmp_path), cfg, base)
    m = Wan2_2Transformer3DModel_FlexAM.from_pretrained(str(tmp_path), torch_dtype=torch.float32, transformer_additional_kwargs={
        "dict_mapping": {"in_channels_ckpt": "in_dim", "hidden_size_ckpt": "dim"}})
    w = m.state_dict()["patch_embedding.weight"]
    torch.testing.assert_close(w[:, :100], sd["patch_embedding.weight"][:, :100], rtol=0, atol=0)
    assert float(w[:, 100:].abs().max()) == 0                                           # FX.py:1307-1310 zero pad
    assert m.state_dict()["head.head.bias"].shape == (192,)


def test_vae_from_pretrained_prefixes_model(tmp_path):
    from flexam_amd import AutoencoderKLWan3_8
    raw = C.vae_weights(C.VAE_SMALL, seed=5, prefix="")                                 # Wan2.2_VAE.pth style keys
    raw.update(C.vae_enc_weights(C.VAE_ENC_SMALL, seed=6, prefix=""))                   # encoder + conv1 keys of the same file
    path = str(tmp_path / "Wan2.2_VAE.pth")
    torch.save(raw, path)
    vae = AutoencoderKLWan3_8.from_pretrained(path, additional_kwargs=dict(
        latent_channels=48, c_dim=16, dec_dim=16, temporal_compression_ratio=4, spatial_compression_ratio=16, vae_type="AutoencoderKLWan3_8"))
    got = vae.state_dict()
    assert set(got) == {"model." + k for k in raw}                                       # the full reference inventory, nothing else
    for k, v in raw.items():
        torch.testing.assert_close(got["model." + k], v, rtol=0, atol=0)
    assert vae.config.latent_channels == 48 and vae.spatial_compression_ratio == 16 and vae.temporal_compression_ratio == 4
