"""CPU: the look-ahead image plug of the drop-in adapter (sam2_opt_amd/plugin.py::LookaheadImagePlug) with a stub engine.
The reference hands the plug `inference_state["images"][t].unsqueeze(0)` - a view of frame t of the clip tensor
(sam2_video_predictor_official.py:810-841); the plug must serve exactly the frame it is asked for, whatever it encoded ahead."""
import torch

from sam2_opt_amd.plugin import LookaheadImagePlug

S = 8          # tiny frames: the plug only looks at shapes, strides and storage offsets


class StubEngine:
    def __init__(self, max_batch=8):
        self.device, self.max_batch, self.batches = torch.device("cpu"), max_batch, []

    def image_encoder(self, img):
        self.batches.append(img.shape[0])
        tag = img.mean(dim=(1, 2, 3)).view(-1, 1, 1, 1)            # identifies the frame
        return tuple(tag.expand(-1, c, 2, 2).clone() for c in (1, 2, 3, 4, 5, 6, 7))


def _clip(T):
    return torch.arange(T, dtype=torch.float32).view(T, 1, 1, 1).expand(T, 3, S, S).contiguous()


def _ask(plug, clip, t):
    outs = plug(clip[t].unsqueeze(0))                               # what _get_image_feature passes
    assert len(outs) == 7 and all(o.shape[0] == 1 for o in outs)
    assert all(float(o.flatten()[0]) == float(t) for o in outs), (t, [float(o.flatten()[0]) for o in outs])


def test_forward_pass_encodes_in_batches_and_serves_the_right_frames():
    eng, clip = StubEngine(), _clip(21)
    plug = LookaheadImagePlug(eng, depth=8)
    for t in range(21):
        _ask(plug, clip, t)
    assert eng.batches == [8, 8, 5], eng.batches
    assert plug.stats["hits"] == 18 and plug.stats["frames_encoded"] == 21


def test_reverse_pass_looks_behind():
    eng, clip = StubEngine(), _clip(12)
    plug = LookaheadImagePlug(eng, depth=4)
    _ask(plug, clip, 11)                      # direction unknown yet: a forward batch of what is left (1 frame)
    for t in range(10, -1, -1):
        _ask(plug, clip, t)
    assert eng.batches == [1, 4, 4, 3], eng.batches       # the second call already shows the direction: nothing is encoded twice


def test_revisit_and_jump_are_served_correctly():
    eng, clip = StubEngine(), _clip(30)
    plug = LookaheadImagePlug(eng, depth=8)
    for t in (0, 1, 2, 2, 1, 17, 18, 3, 29, 0):
        _ask(plug, clip, t)


def test_other_clip_drops_the_cache_and_copies_take_the_plain_path():
    eng = StubEngine()
    plug = LookaheadImagePlug(eng, depth=8)
    a, b = _clip(10), _clip(10) + 100.0
    _ask(plug, a, 0)
    assert len(plug.cache) == 7
    outs = plug(b[0].unsqueeze(0))
    assert float(outs[0].flatten()[0]) == 100.0 and all(float(v[0][0].flatten()[0]) >= 100.0 for v in plug.cache.values())
    n = len(eng.batches)
    one = a[3].clone().unsqueeze(0)            # a copy (e.g. frames offloaded to the host): its storage holds one frame
    outs = plug(one)
    assert float(outs[0].flatten()[0]) == 3.0 and eng.batches[n:] == [1]


def test_depth_is_bounded_by_the_engine_batch():
    eng, clip = StubEngine(max_batch=2), _clip(5)
    plug = LookaheadImagePlug(eng, depth=8)
    for t in range(5):
        _ask(plug, clip, t)
    assert eng.batches == [2, 2, 1]
