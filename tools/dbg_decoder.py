"""Stage-by-stage comparison of the HIP two-way transformer with the oracle (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from oracle import sam2_ref as R
from oracle.gen_golden import plug_inputs
from sam2_opt_amd.config import get_config
from sam2_opt_amd.weights import synthetic_state_dict
from sam2_opt_amd.native import Engine

cfg = get_config("large"); sd = synthetic_state_dict(cfg, 0)
src, tokens, pos_src, hr0, hr1 = plug_inputs(cfg)["maskdec_N1T8"]
# oracle stages
t = "sam_mask_decoder.transformer."; H = 8
keys = src.flatten(2).permute(0, 2, 1); kpe = pos_src.flatten(2).permute(0, 2, 1)
q, qpe = tokens, tokens
stages = []
with torch.inference_mode():
    for l in range(2):
        p = f"{t}layers.{l}."
        if l == 0:
            q = R._attn(q, q, q, sd, p + "self_attn", H)
        else:
            qq = q + qpe
            q = q + R._attn(qq, qq, q, sd, p + "self_attn", H)
        q = R._ln(q, sd, p + "norm1", 1e-5); stages.append((q.clone(), keys.clone()))
        q = R._ln(q + R._attn(q + qpe, keys + kpe, keys, sd, p + "cross_attn_token_to_image", H), sd, p + "norm2", 1e-5); stages.append((q.clone(), keys.clone()))
        q = R._ln(q + R._mlp(q, sd, p + "mlp", 2), sd, p + "norm3", 1e-5); stages.append((q.clone(), keys.clone()))
        keys = R._ln(keys + R._attn(keys + kpe, q + qpe, q, sd, p + "cross_attn_image_to_token", H), sd, p + "norm4", 1e-5); stages.append((q.clone(), keys.clone()))
eng = Engine("large", state_dict=sd)
ins = [x.cuda() for x in (src, tokens, pos_src, hr0, hr1)]
for i, (rq, rk) in enumerate(stages):
    os.environ["SAM2MI_DEC_STOP"] = str(i)
    eng.mask_decoder(*ins)
    gq = eng.debug_read("d_tok", 8, 256).cpu(); gk = eng.debug_read("d_keys", 4096, 256).cpu()
    eq = ((gq - rq[0]).abs().max() / rq.abs().max()).item(); ek = ((gk - rk[0]).abs().max() / rk.abs().max()).item()
    per_tok = ((gq - rq[0]).abs().amax(1) / rq.abs().max()).numpy().round(4)
    print(f"stage {i}: q max_rel {eq:.3e} keys max_rel {ek:.3e} per-token {per_tok}", flush=True)
