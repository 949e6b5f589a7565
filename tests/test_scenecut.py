"""Scene-cut chunker: the numpy restatement on CPU, the HIP kernel + host rule
against it on the GPU, and scene-based chunking through the run_av1an drop-in."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import scenecut  # noqa: E402


def clip(oracle, w, h, bd, n, scene_len, seed=31):
    return [oracle.synthclip_frame(w, h, bd, seed=seed, t=t, scene_len=scene_len) for t in range(n)]


def raw_of(planes, bd):
    dt = np.uint8 if bd == 8 else np.dtype("<u2")
    return b"".join(np.ascontiguousarray(p.astype(dt)).tobytes() for p in planes)


def test_rule_finds_the_synthetic_cuts(oracle):
    """synthclip v1 re-seeds its rectangles and noise every `scene_len` frames: exactly those
    frames must start scenes; a too-short scene is merged; streaming window by window changes nothing."""
    for bd in (8, 10):
        fr = clip(oracle, 200, 120, bd, 40, 15)
        luma = [f[0] for f in fr]
        sads, cuts, _ = scenecut.scene_cuts(luma, bd, min_scene_len=12)
        assert [t for t, c in enumerate(cuts) if c] == [0, 15, 30]
        assert sads[0] == 0 and sads[15] > 2 * sads[14]
        _, cuts20, _ = scenecut.scene_cuts(luma, bd, min_scene_len=20)
        assert [t for t, c in enumerate(cuts20) if c] == [0, 30]
        s1, c1, st = scenecut.scene_cuts(luma[:17], bd, min_scene_len=12)
        s2, c2, _ = scenecut.scene_cuts(luma[17:], bd, min_scene_len=12, prev=luma[16], state=st)
        assert s1 + s2 == sads and c1 + c2 == cuts


def test_static_and_flat_clips_have_one_scene(oracle):
    y = [np.full((64, 64), 100, np.uint16)] * 30
    _, cuts, _ = scenecut.scene_cuts(y, 8)
    assert sum(cuts) == 1 and cuts[0] == 1


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,bd", [(200, 120, 8), (200, 120, 10), (648, 360, 10)])
def test_hip_scene_cuts_equal_the_restatement(av1mi, ctx, oracle, w, h, bd):
    import torch
    n = 36
    fr = clip(oracle, w, h, bd, n, 14)
    luma = [f[0] for f in fr]
    ref_sad, ref_cut, _ = scenecut.scene_cuts(luma, bd, min_scene_len=12)
    p = av1mi.default_params(w, h, bd)
    raw = [raw_of(f, bd) for f in fr]
    sad, cut, _ = ctx.scene_cuts(p, b"".join(raw), n)
    assert sad == ref_sad and cut == ref_cut and sum(cut) == 3
    # streamed in two windows with carried state == one shot
    s1, c1, st = ctx.scene_cuts(p, b"".join(raw[:20]), 20)
    s2, c2, _ = ctx.scene_cuts(p, b"".join(raw[20:]), n - 20, prev_frame=raw[19], state=st)
    assert s1 + s2 == ref_sad and c1 + c2 == ref_cut
    # HBM-resident frames
    d = torch.frombuffer(bytearray(b"".join(raw)), dtype=torch.uint8).to("cuda:0")
    torch.cuda.synchronize()
    s3, c3, _ = ctx.scene_cuts(p, d.data_ptr(), n, on_device=True)
    assert s3 == ref_sad and c3 == ref_cut


@pytest.mark.gpu
def test_encode_file_chunks_at_scene_cuts(av1mi, oracle, tmp_path):
    """chunk_frames = 0: chunks end at the detected cuts.  All frames are key frames, so the stream is the
    same as with fixed-length chunks; the report says how the clip was split."""
    w, h, n = 136, 72, 40
    fr = clip(oracle, w, h, 8, n, 13)
    y4m = tmp_path / "clip.y4m"
    with open(y4m, "wb") as f:
        f.write(b"YUV4MPEG2 W%d H%d F30:1 Ip A1:1 C420jpeg\n" % (w, h))
        for x in fr:
            f.write(b"FRAME\n" + raw_of(x, 8))
    a, b = tmp_path / "scene.ivf", tmp_path / "fixed.ivf"
    rep_a = av1mi.run_mi355x(av1mi.EncodeParams(y4m, a, tmp_path, av1mi.derive_plan(8), chunk_frames=0))
    rep_b = av1mi.run_mi355x(av1mi.EncodeParams(y4m, b, tmp_path, av1mi.derive_plan(8), chunk_frames=16))
    assert rep_a.chunks == 4 and rep_b.chunks == 3 and rep_a.frames == rep_b.frames == n   # cuts at 0, 13, 26, 39
    assert a.read_bytes() == b.read_bytes()
