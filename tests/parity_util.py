"""Shared measurement helpers of the GPU parity tests: every comparison with the oracle is printed, appended to
gpurun_out/parity/parity.jsonl (copied to profiles/<round>/parity.jsonl for the record) and asserted against a bound that
sits just above what was measured, so that a regression in accuracy fails instead of hiding under a loose tolerance.

Units.  The network maps [0,1] images to [0,1] images and runs in fp16 with fp32 accumulation, like the reference's kFP16
engine (img2img_build.cpp:128).  Its errors are absolute (sums of O(1) terms), so they are quoted in units of the fp16 ULP of
the output range's top binade [0.5, 1): ULP16 = 2^-11 = 4.88e-4 (north_star: "within 1 ULP fp16 per pixel"), next to the ULP
of each reference value for reference values >= 0.5.  Frames are quoted in u8 LSB and PSNR."""
import json
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ULP16 = 2.0 ** -11
_OUT = os.path.join(ROOT, "gpurun_out", "parity")


def smooth_frame(rows, cols, seed):
    """Seeded test frame (u8 BGR): low-frequency sinusoids + a few LSB of noise (SURVEY 8d's smooth variant)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:rows, 0:cols]
    img = 120 + 70 * np.sin(xx / 11.0 + seed) * np.cos(yy / 9.0) + 30 * np.sin((xx + yy) / 23.0)
    return np.clip(img[..., None] + rng.integers(-6, 7, (rows, cols, 3)), 0, 255).astype(np.uint8)


def psnr(a, b):
    mse = float(np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2))
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def _record(rec):
    try:
        os.makedirs(_OUT, exist_ok=True)
        with open(os.path.join(_OUT, "parity.jsonl"), "a") as f:
            f.write(json.dumps(rec) + "\n")
    except OSError:
        pass
    print("PARITY " + json.dumps(rec), flush=True)


def network_report(name, y, ref16, ref32=None):
    """y: engine output [B,3,T',T'] f32; ref16: oracle in fp16-boundary mode; ref32: oracle in fp32 (optional)."""
    d = np.abs(y.astype(np.float64) - ref16.astype(np.float64))
    top = ref16 >= 0.5
    ulp_ref = np.where(top, np.exp2(np.floor(np.log2(np.maximum(np.abs(ref16), 2.0 ** -14))) - 10), np.inf)
    rec = {"test": name, "kind": "network", "max_abs": float(d.max()), "mean_abs": float(d.mean()),
           "max_ulp16": float(d.max() / ULP16), "p999_ulp16": float(np.quantile(d, 0.999) / ULP16),
           "max_ulp_of_ref_top_binade": float((d / ulp_ref).max()) if top.any() else 0.0,
           # how far from north_star's "within 1 ULP fp16 per pixel" the outputs are, as fractions of the outputs (the distance itself cannot be
           # measured against TensorRT here; these are against the oracle's model of an fp16 engine, and below against fp32)
           "frac_within_1_ulp16": float((d <= ULP16 * (1 + 1e-9)).mean()), "frac_within_2_ulp16": float((d <= 2 * ULP16 * (1 + 1e-9)).mean())}
    if ref32 is not None:
        e32 = y.astype(np.float64) - ref32.astype(np.float64)
        o32 = ref16.astype(np.float64) - ref32.astype(np.float64)
        d32, o = np.abs(e32), np.abs(o32)
        # the engine and the fp16-boundary oracle are two fp16 evaluations of the same fp32 function: each sits within ~2 ULP16 of the fp32
        # value, their mutual distance is largest where they fall on opposite sides of it.  `straddle_at_worst` = the two signed errors
        # (ULP16) at the element where |engine - fp16 oracle| is largest.
        k = np.unravel_index(int(np.argmax(d)), d.shape)
        rec.update({"max_ulp16_vs_fp32_oracle": float(d32.max() / ULP16), "mean_abs_vs_fp32_oracle": float(d32.mean()),
                    "p999_ulp16_vs_fp32_oracle": float(np.quantile(d32, 0.999) / ULP16),
                    "rms_ulp16_vs_fp32_oracle": float(np.sqrt(np.mean(e32 ** 2)) / ULP16),
                    "frac_within_1_ulp16_vs_fp32_oracle": float((d32 <= ULP16).mean()), "frac_within_2_ulp16_vs_fp32_oracle": float((d32 <= 2 * ULP16).mean()),
                    "fp16_oracle_frac_within_1_ulp16_of_fp32_oracle": float((o <= ULP16).mean()),
                    "fp16_oracle_vs_fp32_oracle_max_ulp16": float(o.max() / ULP16),
                    "fp16_oracle_vs_fp32_oracle_p999_ulp16": float(np.quantile(o, 0.999) / ULP16),
                    "fp16_oracle_vs_fp32_oracle_rms_ulp16": float(np.sqrt(np.mean(o32 ** 2)) / ULP16),
                    "straddle_at_worst": [round(float(e32[k] / ULP16), 3), round(float(o32[k] / ULP16), 3)]})
    _record(rec)
    return rec


def frame_report(name, out, ref):
    d = np.abs(out.astype(np.int32) - ref.astype(np.int32))
    rec = {"test": name, "kind": "frame", "max_lsb": int(d.max()), "psnr_db": round(float(psnr(out, ref)), 2),
           "frac_pixels_off_by_1": float((d == 1).mean()), "frac_pixels_off_by_more": float((d > 1).mean())}
    _record(rec)
    return rec


# The engine measured against an IDEAL fp16 engine (asserted next to the absolute bounds): against the fp32 oracle the engine must not be
# further off than the fp16-boundary oracle is - the oracle's per-operator-fp16 model of an engine with fp32 accumulation - beyond a small
# allowance for where the two round differently.  Measured (profiles/r3_late/parity.jsonl, every graph family): engine max 1.04-2.33 vs
# oracle max 1.04-2.40 ULP16, differences -0.13..+0.14.
REL_MAX_ALLOW = 0.5      # ULP16 on the maxima (a maximum over 1e5-1e7 elements of a quantised quantity moves in steps; measured <= +0.14)
REL_RMS_FACTOR = 1.10    # on the rms errors
REL_P999_ULP16 = 2.0     # p99.9 of |engine - fp32 oracle| (measured <= 2.0 on every graph)


def assert_as_accurate_as_ideal_fp16(r):
    assert r["max_ulp16_vs_fp32_oracle"] <= r["fp16_oracle_vs_fp32_oracle_max_ulp16"] + REL_MAX_ALLOW, r
    assert r["rms_ulp16_vs_fp32_oracle"] <= REL_RMS_FACTOR * r["fp16_oracle_vs_fp32_oracle_rms_ulp16"], r
    assert r["p999_ulp16_vs_fp32_oracle"] <= REL_P999_ULP16, r


BLOCK_MEAN_TOL = 0.12    # u8 LSB: engine frames differ from the oracle's by <= 1 LSB on a few per cent of the pixels (measured block-mean differences <= 0.06)


def check_config_fixture(name, out):
    """Compare an engine frame with the committed oracle output of tests/golden/make_config_fixtures.py: the stored windows of the
    expected frame at <= 1 LSB (reported like any frame comparison), the 32 x 32 block means of the whole frame within BLOCK_MEAN_TOL."""
    z = np.load(os.path.join(ROOT, "tests", "golden", f"cfg_{name}.npz"))
    wins, crops, bm = z["windows"], z["crops"], z["block_means"]
    got = np.stack([out[y:y + h, x:x + w] for y, x, h, w in wins])
    rec = frame_report(f"config fixture {name}: {len(wins)} windows of {crops.shape[1]}x{crops.shape[2]}", got, crops)
    hb, wb = bm.shape[:2]
    B = 32
    mine = out[:hb * B, :wb * B].reshape(hb, B, wb, B, 3).astype(np.float64).mean(axis=(1, 3))
    rec["max_block_mean_diff"] = float(np.abs(mine - bm).max())
    _record({"test": f"config fixture {name}: 32x32 block means of the whole frame", "kind": "block_means", "max_block_mean_diff": rec["max_block_mean_diff"]})
    return rec
