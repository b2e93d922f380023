"""Frame ingest: the reference's two resizers.
CPU: the oracle restatements (oracle/ingest.py) bit-exact against Pillow / within 2e-6 of torch's antialiased bilinear.
GPU: the device kernels (csrc/resize.hip) bit-exact against the oracle (and Pillow), the JPEG-folder path of init_state against
a fixture recorded from the REFERENCE's loader (tests/golden/ingest_jpeg.npz, oracle/gen_golden.py::gen_ingest)."""
import io
import os

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIZES = [(180, 320), (540, 960), (1536, 2048), (1024, 700), (37, 53), (1024, 1024), (2000, 1024)]


def _image(h, w, seed):
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([(np.sin(yy / 17.0 + seed) * 0.5 + 0.5) * 255, (xx * 255.0 / max(w - 1, 1)), ((yy + xx) % 64) * 4.0], -1)
    noise = rs.randint(-40, 40, (h, w, 3))
    img = np.clip(base + noise, 0, 255).astype(np.uint8)
    img[h // 3: h // 3 + 5, :, :] = 255                       # hard edges: overshoot of the bicubic kernel must clamp
    img[:, w // 2: w // 2 + 3, :] = 0
    return img


@pytest.mark.parametrize("hw", SIZES[:5])
def test_oracle_pil_bicubic_is_bit_exact_vs_pillow(hw):
    Image = pytest.importorskip("PIL.Image")
    from oracle.ingest import pil_bicubic_u8
    S = 256 if max(hw) > 1200 else 128                          # small targets keep the CPU test short; same code path
    img = _image(*hw, seed=hw[0])
    want = np.array(Image.fromarray(img).resize((S, S)))
    assert np.array_equal(pil_bicubic_u8(img, S), want)


@pytest.mark.parametrize("hw", SIZES[:5])
def test_oracle_aa_bilinear_vs_torch(hw):
    from oracle.ingest import aa_bilinear_f32
    S = 192
    img = _image(*hw, seed=hw[1])
    t = torch.from_numpy(img).permute(2, 0, 1)[None].float().div(255)
    want = torch.nn.functional.interpolate(t, size=(S, S), mode="bilinear", align_corners=False, antialias=True)[0].numpy()
    got = aa_bilinear_f32(img, S)
    assert got.shape == want.shape and np.abs(got - want).max() <= 2e-6


@pytest.mark.gpu
def test_device_resizers_match_oracle_and_libraries():
    from oracle.ingest import aa_bilinear_f32, pil_bicubic_u8
    from sam2_opt_amd.native import Engine
    eng = Engine("large", state_dict=None)
    try:
        for i, hw in enumerate(SIZES):
            img = _image(*hw, seed=30 + i)
            d = torch.from_numpy(img).cuda().contiguous()
            got = eng.resize_u8_pil_bicubic(d, 1024).cpu().numpy()
            try:
                from PIL import Image
                want = np.array(Image.fromarray(img).resize((1024, 1024)))
            except ImportError:
                want = pil_bicubic_u8(img, 1024)
            assert np.array_equal(got, want), (hw, int((got != want).sum()))
            gf = eng.resize_image_aa_bilinear(d, 1024).cpu()
            t = torch.from_numpy(img).permute(2, 0, 1)[None].float().div(255)
            wf = torch.nn.functional.interpolate(t, size=(1024, 1024), mode="bilinear", align_corners=False, antialias=True)[0]
            assert float((gf - wf).abs().max()) <= 2e-6, (hw, float((gf - wf).abs().max()))
        small = _image(90, 160, seed=1)
        assert np.array_equal(eng.resize_u8_pil_bicubic(torch.from_numpy(small).cuda(), 256).cpu().numpy(), pil_bicubic_u8(small, 256))
        assert np.abs(eng.resize_image_aa_bilinear(torch.from_numpy(small).cuda(), 256).cpu().numpy() - aa_bilinear_f32(small, 256)).max() <= 2e-6
    finally:
        eng.close()


@pytest.mark.gpu
def test_jpeg_folder_matches_reference_loader(sd_large, cfg_large, tmp_path):
    """init_state(video_path=<folder of JPEGs>): PIL decode on the host, PIL-exact bicubic resize + normalisation on the device,
    against load_video_frames_from_jpg_images of the REFERENCE run on the same JPEG bytes (tests/golden/ingest_jpeg.npz)."""
    pytest.importorskip("PIL.Image")
    from sam2_opt_amd.synthetic import normalize_frames
    from sam2_opt_amd.video_predictor import SAM2VideoPredictor
    g = np.load(os.path.join(ROOT, "tests", "golden", "ingest_jpeg.npz"))
    n = int(g["num_frames"][0])
    for i in range(n):
        (tmp_path / f"{i:05d}.jpg").write_bytes(g[f"jpeg{i}"].tobytes())
    pred = SAM2VideoPredictor("large", state_dict=sd_large, encode_batch=2)
    try:
        st = pred.init_state(video_path=str(tmp_path))
        assert (st["video_height"], st["video_width"]) == (int(g["video_hw"][0]), int(g["video_hw"][1])) and st["num_frames"] == n
        u8 = st["images"]
        assert u8.dtype == torch.uint8 and tuple(u8.shape) == (n, 1024, 1024, 3)
        got = normalize_frames(u8.cpu().numpy(), cfg_large).numpy().reshape(-1)             # /255, mean / std like misc.py:270-276
        stride, size = (int(v) for v in g["images/meta"])
        assert got.size == size
        assert np.abs(got[::stride] - g["images/sample"]).max() <= 1e-6
        # and the clip is usable: a click on frame 0 at video coordinates, two frames tracked
        pred.add_new_points_or_box(st, 0, 1, points=np.array([[160.0, 90.0]], np.float32), labels=np.array([1], np.int32))
        out = [vm for _, _, vm in pred.propagate_in_video(st)]
        assert len(out) == n and tuple(out[0].shape[-2:]) == (st["video_height"], st["video_width"])
    finally:
        pred.release()


@pytest.mark.gpu
def test_init_state_reference_options(sd_large, tmp_path):
    """init_state(video_path, offload_video_to_cpu, offload_state_to_cpu, async_loading_frames) - the reference's signature
    (sam2_video_predictor_official.py:148-154): every combination tracks the JPEG clip to the SAME masks as the plain call
    (bit-equal), async loading decodes in a background thread (utils/misc.py:104-169), offloaded per-frame outputs live on the host."""
    pytest.importorskip("PIL.Image")
    from sam2_opt_amd.video_predictor import SAM2VideoPredictor
    g = np.load(os.path.join(ROOT, "tests", "golden", "ingest_jpeg.npz"))
    n = int(g["num_frames"][0])
    for i in range(n):
        (tmp_path / f"{i:05d}.jpg").write_bytes(g[f"jpeg{i}"].tobytes())
    pred = SAM2VideoPredictor("large", state_dict=sd_large, encode_batch=2)
    try:
        outs = {}
        for name, args in (("plain", ()), ("positional", (True, True, True)), ("async", (False, False, True)), ("offload", (False, True, False))):
            st = pred.init_state(str(tmp_path), *args)
            assert st["num_frames"] == n and (st["video_height"], st["video_width"]) == (int(g["video_hw"][0]), int(g["video_hw"][1]))
            pred.add_new_points_or_box(st, 0, 1, points=np.array([[160.0, 90.0]], np.float32), labels=np.array([1], np.int32))
            outs[name] = [vm.clone() for _, _, vm in pred.propagate_in_video(st)]
            assert len(outs[name]) == n
            stored = st["output_dict_per_obj"][0]["non_cond_frame_outputs"][n - 1]["pred_masks"]
            assert stored.device.type == ("cpu" if args[1:2] == (True,) else "cuda"), name
            if args[2:3] == (True,):
                st["images"].thread.join(timeout=60)
                assert not st["images"].thread.is_alive() and st["images"].exception is None
            # a correction click reads the stored (possibly host-resident) logits of the frame
            pred.add_new_points_or_box(st, n - 1, 1, points=np.array([[100.0, 60.0]], np.float32), labels=np.array([0], np.int32))
            pred.release_state(st)
        for name in ("positional", "async", "offload"):
            for a, b in zip(outs["plain"], outs[name]):
                assert torch.equal(a, b), name
        with pytest.raises(ValueError, match="async_loading_frames"):
            pred.init_state(frames_u8=np.zeros((2, 1024, 1024, 3), np.uint8), async_loading_frames=True)
        with pytest.raises(TypeError):
            pred.init_state(str(tmp_path), no_such_option=True)          # unknown options are refused, not swallowed
    finally:
        pred.release()
