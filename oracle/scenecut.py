"""scenecut.py - CPU restatement (numpy) of the scene-cut chunker.  TEST INFRASTRUCTURE ONLY.

What it restates: the scene detection av1an runs before handing chunks to its `--workers N` encoders
(/root/reference/crates/daemon/src/encode/av1an.rs:100-104).  The reference holds no
code or fixtures for it (av1an is an absent, unpinned external tool): PARITY UNPINNED; the rule is the one this
build defines (include/av1mi.h: av1mi_scene_cuts) and this file is the checker for the HIP kernel + host rule.
"""
import numpy as np


def luma_sad(cur, prev):
    """SAD of two luma planes (any integer dtype)."""
    return int(np.abs(cur.astype(np.int64) - prev.astype(np.int64)).sum())


class SceneState:
    def __init__(self):
        self.frames_since_cut = 0
        self.hist = []


def rule_step(st, has_prev, sad, luma_samples, bit_depth, min_len):
    """One frame of the integer cut rule; returns 1 if the frame starts a new scene."""
    if not has_prev:
        st.hist = []
        st.frames_since_cut = 1
        return 1
    d = ((sad >> (bit_depth - 8)) << 8) // luma_samples
    if st.hist:
        strong = 2 * d * len(st.hist) >= 5 * sum(st.hist) and d >= 8 * 256
    else:
        strong = d >= 24 * 256
    if strong and st.frames_since_cut >= min_len:
        st.hist = []
        st.frames_since_cut = 1
        return 1
    if not strong:
        st.hist = (st.hist + [d])[-8:]
    st.frames_since_cut += 1
    return 0


def scene_cuts(luma_planes, bit_depth, min_scene_len=12, prev=None, state=None):
    """luma_planes: list of 2-D arrays.  Returns (sads, cuts, state)."""
    st = state if state is not None else SceneState()
    sads, cuts = [], []
    n = luma_planes[0].size
    for t, y in enumerate(luma_planes):
        p = luma_planes[t - 1] if t > 0 else prev
        sad = luma_sad(y, p) if p is not None else 0
        sads.append(sad)
        cuts.append(rule_step(st, p is not None, sad, n, bit_depth, max(1, min_scene_len)))
    return sads, cuts, st
