"""Model hyper-parameters for the SAM 2.1 checkpoints this backend serves.

A plain-dict restatement of the values in the reference's Hydra YAML
(/root/reference/sam2/sam2/configs/sam2.1/sam2.1_hiera_l.yaml:1-120 and
sam2.1_hiera_t.yaml) plus the overrides the reference's builders inject
(sam2/build_sam.py:81-88 for the image model, :110-131 for the video
predictor).  Hydra/OmegaConf are not needed to run this backend.
"""
from __future__ import annotations

import copy

_COMMON = dict(
    image_size=1024,
    backbone_stride=16,
    d_model=256,            # FpnNeck.d_model == SAM2Base.hidden_dim
    mem_dim=64,             # MemoryEncoder.out_dim
    num_maskmem=7,
    max_obj_ptrs_in_encoder=16,
    sigmoid_scale_for_mem_enc=20.0,
    sigmoid_bias_for_mem_enc=-10.0,
    # hiera
    q_stride=2,
    window_pos_embed_bkg_spatial_size=(7, 7),
    fpn_top_down_levels=(2, 3),
    scalp=1,
    # memory attention
    memattn_layers=4,
    memattn_ffn=2048,
    rope_theta=10000.0,
    rope_feat_size=64,
    # sam heads
    dec_depth=2,
    dec_heads=8,
    dec_mlp=2048,
    num_multimask_outputs=3,
    multimask_min_pt_num=0,
    multimask_max_pt_num=1,
    dynamic_multimask_stability_delta=0.05,
    dynamic_multimask_stability_thresh=0.98,
    # predictor-level overrides (build_sam.py:110-131)
    binarize_mask_from_pts_for_mem_enc=True,
    fill_hole_area=8,
    img_mean=(0.485, 0.456, 0.406),
    img_std=(0.229, 0.224, 0.225),
)

MODEL_CONFIGS = {
    # sam2.1_hiera_l.yaml:11-16
    "large": dict(_COMMON, name="sam2.1_hiera_large", embed_dim=144, num_heads=2,
                  stages=(2, 6, 36, 4), global_att_blocks=(23, 33, 43),
                  window_spec=(8, 4, 16, 8)),
    # sam2.1_hiera_t.yaml:11-15 (window_spec is the Hiera default, hieradet.py:185-190)
    "tiny": dict(_COMMON, name="sam2.1_hiera_tiny", embed_dim=96, num_heads=1,
                 stages=(1, 2, 7, 2), global_att_blocks=(5, 7, 9),
                 window_spec=(8, 4, 14, 7)),
    # sam2.1_hiera_s.yaml:11-16
    "small": dict(_COMMON, name="sam2.1_hiera_small", embed_dim=96, num_heads=1,
                  stages=(1, 2, 11, 2), global_att_blocks=(7, 10, 13),
                  window_spec=(8, 4, 14, 7)),
    # sam2.1_hiera_b+.yaml:11-13: everything else is the Hiera default (hieradet.py:175-200)
    "base_plus": dict(_COMMON, name="sam2.1_hiera_base_plus", embed_dim=112, num_heads=2,
                      stages=(2, 3, 16, 3), global_att_blocks=(12, 16, 20),
                      window_spec=(8, 4, 14, 7), window_pos_embed_bkg_spatial_size=(14, 14)),
}


def get_config(name: str = "large") -> dict:
    if name not in MODEL_CONFIGS:
        raise KeyError(f"unknown SAM2 model {name!r}; have {sorted(MODEL_CONFIGS)}")
    return copy.deepcopy(MODEL_CONFIGS[name])


def hiera_block_specs(cfg: dict) -> list:
    """Per-block (dim_in, dim_out, heads, window, q_pool) exactly as
    Hiera.__init__ derives them (hieradet.py:243-268): the window size lags one
    block behind the stage change, global blocks get window 0."""
    stages = cfg["stages"]
    depth = sum(stages)
    stage_ends = [sum(stages[:i]) - 1 for i in range(1, len(stages) + 1)]
    q_pool_blocks = [x + 1 for x in stage_ends[:-1]]
    dim, heads, cur_stage = cfg["embed_dim"], cfg["num_heads"], 1
    specs = []
    for i in range(depth):
        dim_out = dim
        window = cfg["window_spec"][cur_stage - 1]
        if i in cfg["global_att_blocks"]:
            window = 0
        if i - 1 in stage_ends:
            dim_out = dim * 2
            heads = heads * 2
            cur_stage += 1
        specs.append(dict(idx=i, dim=dim, dim_out=dim_out, heads=heads, window=window,
                          q_pool=i in q_pool_blocks, stage_end=i in stage_ends))
        dim = dim_out
    return specs
