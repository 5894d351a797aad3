"""Pins the oracle's tile pipeline against the known-answer vectors derived from the reference's own formulas
(the reference ships no tests or fixtures: SURVEY.md section 4)."""
import json
import os

import numpy as np
import pytest

from oracle import pipeline as P

HERE = os.path.dirname(os.path.abspath(__file__))
KATS = json.load(open(os.path.join(HERE, "golden", "tile_kats.json")))


@pytest.mark.parametrize("case", KATS["cases"], ids=lambda c: c["name"])
def test_calculate_tiles_kat(case):
    c = case
    n, ins, outs = P.calculate_tiles(c["W"], c["H"], c["W"] * c["s"], c["H"] * c["s"], (c["T"], c["T"]),
                                     (c["Tout"], c["Tout"]), c["s"], (c["ov"], c["ov"]))
    assert n == c["nx"] * c["ny"]
    ny = c["ny"]
    assert list(ins[0].astuple()) == c["in0"]
    assert list(ins[1].astuple()) == c["in1"]
    assert list(ins[ny].astuple()) == c["in_ny"]
    assert list(ins[-1].astuple()) == c["in_last"]
    assert list(outs[1].astuple()) == c["out1"]
    assert list(outs[-1].astuple()) == c["out_last"]
    assert ins[0].x == -c["border"] and ins[ny].x - ins[0].x == c["sIn"] - c["inOv"]
    assert outs[ny].x == c["Tout"] - c["outOv"]
    # canvas is exactly covered: last tile ends at the canvas edge
    assert outs[-1].x + outs[-1].w == c["W"] * c["s"] and outs[-1].y + outs[-1].h == c["H"] * c["s"]


def test_q6_overlap_rounding_mismatch():
    q = KATS["q6_rounding"]
    n, ins, outs = P.calculate_tiles(1920, 1080, 7680, 4320, (q["T"],) * 2, (q["Tout"],) * 2, q["s"], (q["ov"],) * 2)
    ny = round(n / len({r.x for r in ins}))
    assert ins[ny].x == q["in_x1"] and ins[ny].x - ins[0].x == 371
    assert outs[ny].x == q["out_x1"]
    assert P.c_lround(q["T"] * q["ov"]) == q["inOv"] and P.c_lround(q["T"] * q["s"] * q["ov"]) == q["outOv"]


def test_lround_half_away_from_zero():
    assert [P.c_lround(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 12.5, 12.49)] == [1, 2, 3, -1, -2, 13, 12]


def test_ramp_weights_kat():
    # ov=64 -> top mask row r = fl32((r+1)/65.0) for r<64, 1.0 after; bottom/right are flips
    top, right, bottom, left = P.create_tile_weights((64, 64), (960, 960))
    for r in (0, 1, 31, 63):
        assert top[r, 0] == np.float32((r + 1) / 65.0) and left[5, r] == np.float32((r + 1) / 65.0)
    assert top[64, 0] == 1.0 and left[0, 64] == 1.0
    assert np.array_equal(bottom, top[::-1]) and np.array_equal(right, left[:, ::-1])
    # partition of unity across an interior seam (left tile weight (64-k)/65 + right tile (k+1)/65 = 1 in exact arithmetic)
    k = np.arange(64)
    s = right[0, 960 - 64 + k].astype(np.float64) + left[0, k].astype(np.float64)
    assert np.allclose(s, 1.0, atol=1e-7)


def test_apply_weights_order_and_conditions():
    rng = np.random.default_rng(0)
    t = rng.random((16, 16, 3), dtype=np.float32)
    w = P.create_tile_weights((4, 4), (16, 16))
    r = P.Rect(10, 0, 16, 16)   # interior in x on both sides, top edge of the canvas, not bottom edge
    out = P.apply_weights(t, r, 100, 100, w)
    exp = t.copy(); exp *= w[3][..., None]; exp *= w[1][..., None]; exp *= w[2][..., None]
    assert np.array_equal(out, exp)
    r = P.Rect(0, 0, 16, 16)
    out = P.apply_weights(t, r, 16, 16, w)
    assert np.array_equal(out, t)


def test_d4_closure_and_inverse():
    rng = np.random.default_rng(1)
    x = rng.integers(0, 255, (7, 7, 3), dtype=np.uint8)
    seen = set()
    for k in range(8):
        a = np.ascontiguousarray(P.apply_augmentation(x, k))
        assert np.array_equal(P.reverse_augmentation(a, k), x)
        seen.add(a.tobytes())
    assert len(seen) == 8      # the 8 augmentations are exactly the dihedral group, each once
    assert np.array_equal(P.apply_augmentation(x, 3), np.rot90(x, 1))
    assert np.array_equal(P.apply_augmentation(x, 1), x[::-1])


def test_u8_conversions():
    v = np.array([0.0, 0.5 / 255, 1.5 / 255, 2.5 / 255, 1.0, 1.2, -0.3, 254.5 / 255], np.float32)
    exp = np.clip(np.rint(v * np.float32(255.0)), 0, 255).astype(np.uint8)
    assert np.array_equal(P.to_u8(v), exp)
    assert P.to_u8(np.array([2.5 / 255], np.float32))[0] in (2, 3)   # half-to-even on the fp32 product
    blob = P.blob_from_tiles([np.full((2, 2, 3), 255, np.uint8)])
    assert blob.shape == (1, 3, 2, 2) and blob[0, 0, 0, 0] == np.float32(255) * np.float32(1.0 / 255.0)


def test_pad_roi_replicates_border():
    img = np.arange(5 * 6 * 3, dtype=np.uint8).reshape(5, 6, 3)
    t = P.pad_roi(img, P.Rect(-2, -1, 10, 8))
    assert t.shape == (8, 10, 3)
    assert np.array_equal(t[0, 0], img[0, 0]) and np.array_equal(t[7, 9], img[4, 5]) and np.array_equal(t[1, 2], img[0, 0])
    assert np.array_equal(t[3, 4], img[2, 2])


def test_render_identity_network_no_blend():
    """A pass-through 'network' (nearest upscale of the tile centre) must reproduce the nearest-upscaled frame exactly."""
    s, T, border = 2, 16, 2
    To = s * (T - 2 * border)

    def net(x):
        c = x[:, :, border:T - border, border:T - border]
        return np.repeat(np.repeat(c, s, axis=2), s, axis=3)

    rng = np.random.default_rng(2)
    frame = rng.integers(0, 256, (29, 41, 3), dtype=np.uint8)
    out = P.render(frame, net, batch=3, tile=T, scaling=s, overlap=(0.0, 0.0))
    assert np.array_equal(out, np.repeat(np.repeat(frame, s, axis=0), s, axis=1))
    # blending a constant image is a no-op up to 1 LSB (weights sum to 1 in exact arithmetic)
    const = np.full((29, 41, 3), 200, np.uint8)
    outb = P.render(const, net, batch=2, tile=T, scaling=s, overlap=(0.125, 0.125))
    assert np.abs(outb.astype(int) - 200).max() <= 1
    # TTA of a pass-through network is still the identity; the bug-compat mode too (last de-augmented output)
    outt = P.render(frame, net, batch=4, tile=T, scaling=s, overlap=(0.0, 0.0), tta=True)
    assert np.array_equal(outt, np.repeat(np.repeat(frame, s, axis=0), s, axis=1))
