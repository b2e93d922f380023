"""Container-only harness: import the reference's own PyTorch path (backend="torch")
from /root/reference and build its models WITHOUT hydra, from this repo's plain-dict
config.  TEST INFRASTRUCTURE - used only by oracle/gen_golden.py and by tests that
cross-check oracle/sam2_ref.py against the real reference when /root/reference is
mounted.  Nothing here travels to the GPU box and the product never imports it.

The reference cannot be imported as shipped: `sam2/__init__.py:7-11` needs hydra,
`sam2_image_predictor.py:19` & co need the un-vendored `ytools` submodule
(.gitmodules:1-6), `hieradet.py:14` needs iopath, `utils/transforms.py:12` needs
torchvision.  None of those is used by the torch backend's arithmetic, so they are
pre-seeded in sys.modules with inert stand-ins.
"""
from __future__ import annotations

import os
import sys
import types

REF_ROOT = "/root/reference/sam2"


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "sam2", "modeling"))


def _install_shims():
    if "hydra" not in sys.modules:
        hydra = types.ModuleType("hydra")
        hydra.initialize_config_module = lambda *a, **k: None
        hydra.compose = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("hydra shim: compose unused"))
        core = types.ModuleType("hydra.core")
        gh = types.ModuleType("hydra.core.global_hydra")

        class GlobalHydra:
            @staticmethod
            def instance():
                return GlobalHydra()

            def is_initialized(self):
                return True

        gh.GlobalHydra = GlobalHydra
        utils = types.ModuleType("hydra.utils")
        utils.instantiate = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("hydra shim: instantiate unused"))
        hydra.core, hydra.utils, core.global_hydra = core, utils, gh
        sys.modules.update({"hydra": hydra, "hydra.core": core,
                            "hydra.core.global_hydra": gh, "hydra.utils": utils})
    if "omegaconf" not in sys.modules:
        oc = types.ModuleType("omegaconf")
        oc.OmegaConf = type("OmegaConf", (), {"resolve": staticmethod(lambda cfg: None)})
        sys.modules["omegaconf"] = oc
    if "ytools" not in sys.modules:
        yt = types.ModuleType("ytools")
        ex = types.ModuleType("ytools.executor")
        ex.ModelExectuor = object          # type annotation only on the torch backend
        yt.executor = ex
        sys.modules.update({"ytools": yt, "ytools.executor": ex})
    if "iopath" not in sys.modules:
        io = types.ModuleType("iopath")
        com = types.ModuleType("iopath.common")
        fio = types.ModuleType("iopath.common.file_io")
        fio.g_pathmgr = None               # only touched when weights_path is given
        io.common, com.file_io = com, fio
        sys.modules.update({"iopath": io, "iopath.common": com, "iopath.common.file_io": fio})
    try:
        import torchvision  # noqa: F401
    except Exception:
        shim_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_shims")
        if shim_dir not in sys.path:
            sys.path.insert(0, shim_dir)


def import_reference():
    """Returns the imported reference `sam2` package."""
    if not reference_available():
        raise RuntimeError("reference not mounted at /root/reference (container-only harness)")
    _install_shims()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    import sam2  # noqa: F401
    return sam2


def build_reference_model(cfg: dict, kind: str = "video", state_dict=None, fill_hole_area: int = 0, **predictor_kw):
    """Instantiate the reference model tree by hand (what hydra `instantiate` would do
    for configs/sam2.1/sam2.1_hiera_*.yaml plus the overrides of build_sam.py:81-88 /
    :110-131).  kind: "video" -> SAM2VideoPredictor, "base" -> SAM2Base."""
    import_reference()
    from sam2.modeling.backbones.hieradet import Hiera
    from sam2.modeling.backbones.image_encoder import FpnNeck, ImageEncoder
    from sam2.modeling.memory_attention import MemoryAttention, MemoryAttentionLayer
    from sam2.modeling.memory_encoder import CXBlock, Fuser, MaskDownSampler, MemoryEncoder
    from sam2.modeling.position_encoding import PositionEmbeddingSine
    from sam2.modeling.sam.transformer import RoPEAttention
    from sam2.modeling.sam2_base import SAM2Base
    from sam2.sam2_video_predictor import SAM2VideoPredictor

    C = cfg["d_model"]
    trunk = Hiera(embed_dim=cfg["embed_dim"], num_heads=cfg["num_heads"], stages=tuple(cfg["stages"]),
                  global_att_blocks=tuple(cfg["global_att_blocks"]),
                  window_pos_embed_bkg_spatial_size=tuple(cfg["window_pos_embed_bkg_spatial_size"]),
                  window_spec=tuple(cfg["window_spec"]))
    neck = FpnNeck(position_encoding=PositionEmbeddingSine(num_pos_feats=C, normalize=True, scale=None,
                                                           temperature=10000),
                   d_model=C, backbone_channel_list=list(trunk.channel_list),
                   fpn_top_down_levels=list(cfg["fpn_top_down_levels"]), fpn_interp_model="nearest")
    image_encoder = ImageEncoder(trunk=trunk, neck=neck, scalp=cfg["scalp"])

    def rope(**kw):
        return RoPEAttention(rope_theta=cfg["rope_theta"], feat_sizes=[cfg["rope_feat_size"]] * 2,
                             embedding_dim=C, num_heads=1, downsample_rate=1, dropout=0.1, **kw)

    layer = MemoryAttentionLayer(activation="relu", dim_feedforward=cfg["memattn_ffn"], dropout=0.1,
                                 pos_enc_at_attn=False, self_attention=rope(), d_model=C,
                                 pos_enc_at_cross_attn_keys=True, pos_enc_at_cross_attn_queries=False,
                                 cross_attention=rope(rope_k_repeat=True, kv_in_dim=cfg["mem_dim"]))
    memory_attention = MemoryAttention(d_model=C, pos_enc_at_input=True, layer=layer,
                                       num_layers=cfg["memattn_layers"])
    memory_encoder = MemoryEncoder(
        out_dim=cfg["mem_dim"],
        position_encoding=PositionEmbeddingSine(num_pos_feats=cfg["mem_dim"], normalize=True, scale=None,
                                                temperature=10000),
        mask_downsampler=MaskDownSampler(kernel_size=3, stride=2, padding=1),
        fuser=Fuser(layer=CXBlock(dim=C, kernel_size=7, padding=3, layer_scale_init_value=1e-6,
                                  use_dwconv=True), num_layers=2))
    kwargs = dict(
        image_encoder=image_encoder, memory_attention=memory_attention, memory_encoder=memory_encoder,
        num_maskmem=cfg["num_maskmem"], image_size=cfg["image_size"],
        sigmoid_scale_for_mem_enc=cfg["sigmoid_scale_for_mem_enc"],
        sigmoid_bias_for_mem_enc=cfg["sigmoid_bias_for_mem_enc"],
        use_mask_input_as_output_without_sam=True, directly_add_no_mem_embed=True,
        no_obj_embed_spatial=True, use_high_res_features_in_sam=True, multimask_output_in_sam=True,
        iou_prediction_use_sigmoid=True, use_obj_ptrs_in_encoder=True, add_tpos_enc_to_obj_ptrs=True,
        proj_tpos_enc_in_obj_ptrs=True, use_signed_tpos_enc_to_obj_ptrs=True,
        only_obj_ptrs_in_the_past_for_eval=True, pred_obj_scores=True, pred_obj_scores_mlp=True,
        fixed_no_obj_ptr=True, multimask_output_for_tracking=True, use_multimask_token_for_obj_ptr=True,
        multimask_min_pt_num=cfg["multimask_min_pt_num"], multimask_max_pt_num=cfg["multimask_max_pt_num"],
        use_mlp_for_obj_ptr_proj=True, compile_image_encoder=False,
        sam_mask_decoder_extra_args=dict(
            dynamic_multimask_via_stability=True,
            dynamic_multimask_stability_delta=cfg["dynamic_multimask_stability_delta"],
            dynamic_multimask_stability_thresh=cfg["dynamic_multimask_stability_thresh"]),
    )
    if kind == "video":
        model = SAM2VideoPredictor(fill_hole_area=fill_hole_area,
                                   binarize_mask_from_pts_for_mem_enc=cfg["binarize_mask_from_pts_for_mem_enc"],
                                   **predictor_kw, **kwargs)       # predictor_kw: non_overlap_masks, clear_non_cond_mem_around_input, ...
    elif kind == "base":
        model = SAM2Base(**kwargs)
    else:
        raise ValueError(kind)
    if state_dict is not None:
        missing, unexpected = model.load_state_dict(state_dict, strict=True)
        assert not missing and not unexpected
    model.eval()
    return model
