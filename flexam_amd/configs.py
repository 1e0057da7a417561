"""Model hyper-parameters of the FlexAM checkpoints (not shipped in the reference repo: they live in
the checkpoint's config.json, read at wan_transformer3d_FlexAM.py:1199-1211; derived in SURVEY F6)."""

WAN22_FUN_5B_FLEXAM = dict(
    model_type="ti2v", patch_size=(1, 2, 2), text_len=512, in_dim=148, dim=3072, ffn_dim=14336, freq_dim=256, text_dim=4096,
    out_dim=48, num_heads=24, num_layers=30, eps=1e-6, add_ref_conv=True, in_dim_ref_conv=48, add_cnn_block=True,
    in_dim_cnn_block=288, out_dim_cnn_block=48)

WAN22_VAE = dict(latent_channels=48, c_dim=160, dec_dim=256, dim_mult=(1, 2, 4, 4), temperal_downsample=(False, True, True),
                 temporal_compression_ratio=4, spatial_compression_ratio=16)
