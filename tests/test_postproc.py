"""Hole filling.  CPU: the oracle against hand-made known answers.  GPU: the HIP kernel (bounded
flood fill, sam2_opt_amd/csrc/postproc.hip) bit-exact against the oracle on planted and random masks."""
import numpy as np
import pytest


def _known_case():
    m = np.ones((12, 12), np.float32)
    m[1, 1] = -1.0                                   # area 1 hole
    m[3, 3] = m[4, 4] = m[5, 5] = 0.0                # diagonal chain: ONE component under 8-connectivity (area 3)
    m[8:10, 2:6] = -2.0                              # 2 x 4 = 8
    m[7:10, 8:11] = -3.0                             # 3 x 3 = 9
    m[0, 6:12] = -0.5                                # border strip, area 6
    return m


def test_oracle_known_answers():
    from oracle.postproc import fill_holes_in_mask_scores
    m = _known_case()
    out = fill_holes_in_mask_scores(m, 8)
    assert out[1, 1] == np.float32(0.1)
    assert out[3, 3] == out[4, 4] == out[5, 5] == np.float32(0.1)
    assert np.all(out[8:10, 2:6] == np.float32(0.1))             # area == max_area is filled (areas <= max_area)
    assert np.all(out[7:10, 8:11] == -3.0)                       # area 9 > 8 stays
    assert np.all(out[0, 6:12] == np.float32(0.1))               # touching the border does not matter
    assert np.all(out[m > 0] == m[m > 0])                        # foreground untouched
    out2 = fill_holes_in_mask_scores(m, 2)
    assert out2[1, 1] == np.float32(0.1) and out2[3, 3] == 0.0   # the diagonal chain (area 3) is not 3 singletons


def test_oracle_batch_and_idempotence():
    from oracle.postproc import fill_holes_in_mask_scores
    rs = np.random.RandomState(0)
    m = rs.standard_normal((3, 1, 64, 64)).astype(np.float32)
    a = fill_holes_in_mask_scores(m, 8)
    assert a.shape == m.shape and np.all(a[m > 0] == m[m > 0])
    assert np.array_equal(fill_holes_in_mask_scores(a, 8), a)    # filled pixels become foreground: nothing left to fill


def _same_partition(a, b):
    """Two labelings describe the same components iff the label pairs are in 1-1 correspondence."""
    pairs = np.unique(np.stack([a.ravel(), b.ravel()], 1), axis=0)
    return len(np.unique(pairs[:, 0])) == len(pairs) == len(np.unique(pairs[:, 1]))


def _adversarial_masks():
    rs = np.random.RandomState(123)
    out = []
    for dens in (0.05, 0.3, 0.5, 0.6, 0.75, 0.95):                  # around the 8-connectivity percolation threshold too
        out.append(rs.rand(64, 96) < dens)
    yy, xx = np.mgrid[0:64, 0:64]
    out.append((yy + xx) % 2 == 0)                                   # checkerboard: ONE component under 8-connectivity
    out.append((yy % 2 == 0) & (xx % 2 == 0))                        # isolated pixels at block origins
    out.append((yy % 2 == 1) & (xx % 2 == 1))                        # isolated pixels at the block corner the kernel never tests
    out.append((yy % 4 == 1) & (xx % 4 == 2) | (yy % 4 == 2) & (xx % 4 == 1))     # anti-diagonal pairs across block borders
    out.append(np.abs(yy - xx) <= 0)                                 # one long diagonal (corner contacts only)
    out.append(np.abs(yy + xx - 63) <= 0)                            # anti-diagonal
    sp = np.zeros((64, 64), bool)                                    # a spiral: one snake component with long union chains
    y = x = 0
    for step in range(62, 0, -4):
        sp[y, x:x + step] = True; x += step
        sp[y:y + step, x] = True; y += step
        sp[y, x - step + 2:x + 1] = True; x -= step - 2
        sp[y - step + 2:y + 1, x] = True; y -= step - 2
    out.append(sp)
    out.append(np.ones((2, 2), bool))
    out.append(np.zeros((4, 6), bool))
    out.append(np.ones((30, 2), bool))
    return out


def test_blockuf_restatement_equals_scipy_oracle():
    """The pin for SURVEY 8 f-2: the CPU restatement of csrc/connected_components.cu (oracle/cc_blockuf.py) and the
    scipy-based oracle the HIP kernel is held to (oracle/postproc.py) give the same partition and the same areas."""
    from oracle import cc_blockuf, postproc
    for m in _adversarial_masks():
        la, ca = cc_blockuf.get_connected_components(m)
        lb, cb = postproc.connected_components(m)
        assert np.array_equal(la > 0, m) and np.array_equal(lb > 0, m)
        assert np.array_equal(ca, cb), m.shape
        assert _same_partition(la, lb), m.shape
        roots = np.unique(la[la > 0]) - 1                            # a label is (block-origin index of the root) + 1
        W = m.shape[1]
        assert np.all((roots // W) % 2 == 0) and np.all((roots % W) % 2 == 0)


def test_blockuf_fill_equals_scipy_fill_and_known_answers():
    from oracle import cc_blockuf, postproc
    m = _known_case()
    for area in (1, 2, 8, 9, 63):
        assert np.array_equal(cc_blockuf.fill_holes_in_mask_scores(m, area), postproc.fill_holes_in_mask_scores(m, area))
    rs = np.random.RandomState(7)
    for dens in (0.55, 0.8, 0.97):
        s = np.where(rs.rand(2, 1, 128, 128) < dens, 1.0, -1.0).astype(np.float32) * (0.5 + rs.rand(2, 1, 128, 128).astype(np.float32))
        s[0, 0, 5, 5] = 0.0                                          # score exactly 0 is background
        for area in (1, 8, 40):
            assert np.array_equal(cc_blockuf.fill_holes_in_mask_scores(s, area), postproc.fill_holes_in_mask_scores(s, area))


def test_blockuf_rejects_odd_sizes_like_the_cuda_kernel():
    """connected_components.cu:226-227 asserts even H and W; utils/misc.py:325-336 then skips the filling.  The scipy oracle
    and the HIP kernel fill odd-sized masks too (a superset of the reference's behaviour; pred_masks are always 256x256)."""
    from oracle import cc_blockuf, postproc
    odd = -np.ones((37, 53), np.float32)
    odd[3:30, 4:40] = 1.0
    odd[10, 10] = -1.0
    with pytest.raises(ValueError):
        cc_blockuf.get_connected_components(odd <= 0)
    assert np.array_equal(cc_blockuf.fill_holes_in_mask_scores(odd, 8), odd)          # the reference's failure path: unchanged
    assert postproc.fill_holes_in_mask_scores(odd, 8)[10, 10] == np.float32(0.1)


@pytest.mark.gpu
@pytest.mark.parametrize("max_area", [1, 8, 63])
def test_fill_holes_gpu_matches_oracle(max_area):
    import torch
    from oracle.postproc import fill_holes_in_mask_scores
    from sam2_opt_amd.native import Engine
    eng = Engine("large", state_dict=None)
    rs = np.random.RandomState(max_area)
    cases = [np.tile(_known_case(), (1, 1))]
    for dens in (0.55, 0.7, 0.9, 0.98):                          # random masks: many holes of every size and shape
        cases.append(np.where(rs.rand(256, 256) < dens, 1.0, -1.0).astype(np.float32) * (0.5 + rs.rand(256, 256).astype(np.float32)))
    big = np.ones((256, 256), np.float32)
    big[10:10 + max_area, 5] = -1.0                              # exactly max_area (filled)
    big[10:11 + max_area, 9] = -1.0                              # max_area + 1 (kept)
    big[100:140, 100:140] = -1.0                                 # large region
    big[255, 255] = 0.0                                          # corner pixel, score exactly 0 counts as background
    cases.append(big)
    cases.append(-np.ones((37, 53), np.float32))                 # ragged size, all background
    cases.append(np.ones((1, 1), np.float32))
    for m in cases:
        want = fill_holes_in_mask_scores(m, max_area)
        got = eng.fill_holes(torch.from_numpy(m).cuda().contiguous(), max_area).cpu().numpy()
        assert np.array_equal(got, want), (m.shape, max_area, int((got != want).sum()))
    stack = np.stack(cases[1:5])[:, None]                        # (N, 1, H, W) batch like pred_masks
    got = eng.fill_holes(torch.from_numpy(stack).cuda().contiguous(), max_area).cpu().numpy()
    assert np.array_equal(got, fill_holes_in_mask_scores(stack, max_area))
    eng.close()


@pytest.mark.gpu
def test_video_with_fill_hole_area(sd_large, cfg_large):
    """fill_hole_area=8 in the fused video path: every yielded mask equals the oracle's fill of the unfilled run's
    mask (frame 0, whose filled mask feeds its memory, can change later frames - so compare frame 0 exactly and
    check on the others that the fill is a fixed point: no small holes remain)."""
    import torch
    from oracle.postproc import fill_holes_in_mask_scores
    from sam2_opt_amd.synthetic import synthetic_frames_u8
    from sam2_opt_amd.video_predictor import SAM2VideoPredictor
    u8 = synthetic_frames_u8(seed=3, num_frames=4)
    res = {}
    for area in (0, 8):
        p = SAM2VideoPredictor("large", state_dict=sd_large, encode_batch=4, fill_hole_area=area)
        st = p.init_state(frames_u8=u8, video_height=256, video_width=256)     # video res == low res: masks come out as stored
        p.add_new_points_or_box(st, 0, 1, points=np.array([[64.0, 64.0]], np.float32), labels=np.array([1], np.int32))
        res[area] = [vm.float().cpu().numpy() for _, _, vm in p.propagate_in_video(st)]
        p.release()
    assert np.array_equal(res[8][0], fill_holes_in_mask_scores(res[0][0], 8))
    for m in res[8]:
        assert np.array_equal(fill_holes_in_mask_scores(m, 8), m)
