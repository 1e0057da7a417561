"""CPU: the synthetic workloads bench.py measures -- the mask modes of BASELINE configs[3] (foreground_edit) behave like the
reference's mask handling says they should (PIPE.py:675-690, 891-898; demo.py:87-124) -- and the host-side launch plans follow the
CU budget."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _timestep_rows(mask):
    sub = mask[0, 0, :, ::2, ::2].reshape(-1).to(torch.bfloat16).float()        # the reference's mask lives in weight_dtype
    return int(torch.unique(sub).numel())


def test_foreground_edit_masks_of_the_bench():
    import bench
    from flexam_amd.pipeline_wan2_2_fun_control_FlexAM import prepare_masks
    frames, h, w = 17, 128, 224                                                  # small clip, same geometry rules
    lat = (1, 48, 5, h // 16, w // 16)
    # demo.py's form: frame 0 untouched -> pinned: frame 0 known, every later frame fully regenerated, two timestep rows
    mp = bench.blob_mask_pixels(frames, h, w, "blob")
    assert float(mp[:, :, 0].abs().max()) == 0 and 0 < float(mp[:, :, 1:].mean()) < 1
    ml, mask, pinned = prepare_masks(mp.clone(), lat)
    assert pinned and float(mask[:, :, 0].max()) == 0 and float(mask[:, :, 1:].min()) == 1 and _timestep_rows(mask) == 2
    assert 0 < float(ml.mean()) < 1                                              # fractional mask latents drive the 4 mask channels
    # the disc also covers frame 0 -> not pinned, soft edges, more than two rows
    mp = bench.blob_mask_pixels(frames, h, w, "blob-open")
    _, mask, pinned = prepare_masks(mp.clone(), lat)
    assert not pinned and _timestep_rows(mask) >= 2 and float(mask[:, :, 0].max()) > 0      # (14 distinct rows at 97x512x896)
    # random soft mask: many rows (the stress case)
    mp = bench.blob_mask_pixels(frames, h, w, "soft")
    _, mask, pinned = prepare_masks(mp.clone(), lat)
    assert not pinned and _timestep_rows(mask) > 50
    inp = bench.synthetic_inputs(frames, h, w, 64, "blob")
    assert inp["mask"] is None and inp["mask_pixels"] is not None and inp["masked"].shape == (1, 48, 5, h // 16, w // 16)
    base = bench.synthetic_inputs(frames, h, w, 64, "motion")
    assert base["mask_pixels"] is None and float(base["mask"][:, :, 0].max()) == 0 and float(base["mask_latents"][:, :, 0].min()) == 1
    assert torch.equal(inp["latents"], base["latents"])                          # same noise whatever the mask mode


def test_block_flops_match_the_survey_figures():
    import bench
    tf = bench.block_flops(11648, 3072, 14336, 512) / 1e12
    assert abs(tf - 5.131) < 2e-3                                                # 5.131 TFLOP per block and sample


def test_attention_split_plan_follows_the_cu_count():
    from flexam_amd import hip
    # DiT self-attention, CFG pair: 2208 units = 8 full rounds of 256 + 160 -> only the last round is split
    s, start = hip.attn_split_plan(48, 11648, 11648, 256)
    assert start == 2048 and s >= 2
    # one row planned for 128 CUs: 1104 units = 8 rounds of 128 + 80
    s, start = hip.attn_split_plan(24, 11648, 11648, 128)
    assert start == 1024 and s >= 2
    # a whole number of rounds: nothing to split
    assert hip.attn_split_plan(16, 4096, 4096, 256) == (1, 256)
