"""Kernel-level A/B on the GPU box: the LDS-staged fused MLP of round 1 (tools/ab/k_mlp_staged.hip, no longer part of the
library) against the shipped wave-private one (csrc/k_mlp2.hip) on the same random rows and weights, at small ragged sizes and
at the headline row counts (where the shipped kernel is also run twice and must reproduce itself bit for bit).  Builds
tools/ab/mlp_ab.hip with hipcc; no oracle involved."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_fused_mlp_kernels_agree_and_are_deterministic(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    csrc = os.path.join(ROOT, "waifu2x-tensorrt_amd", "csrc")
    exe = str(tmp_path / "mlp_ab")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", csrc, os.path.join(ROOT, "tools", "ab", "mlp_ab.hip"),
                    os.path.join(ROOT, "tools", "ab", "k_mlp_staged.hip"), os.path.join(csrc, "k_mlp2.hip"), os.path.join(csrc, "k_mlp96q.hip"), "-o", exe], check=True, timeout=900)
    out = subprocess.run([exe], check=True, capture_output=True, text=True, timeout=600).stdout
    cases = re.findall(r"C=(\d+) M=(\d+) stats=(\d): max\|dy\|=([0-9.]+) .*max rel stats diff=([0-9.e+-]+)", out)
    assert len(cases) >= 18, out
    for C, M, stats, dy, ds in cases:
        assert float(dy) <= 8e-3, (C, M, stats, dy)          # both round to fp16 once; sums differ in order only
        assert float(ds) <= 5e-2, (C, M, stats, ds)
    twice = re.findall(r"run twice: (\d+) elements differ", out)
    assert len(twice) == 2 and all(int(n) == 0 for n in twice), out


@pytest.mark.gpu
def test_fused_attention96_schedules_agree_with_each_other_and_with_fp32(tmp_path):
    """tools/ab/attn_ab.hip: the round-1 schedule of the C = 96 fused attention (2 windows per workgroup, a wave = a window's 3 heads)
    against the shipped one (2 windows per workgroup, three (window, head) units per wave of which two share a head, the left-over
    queries of a wave's units in one tile) on random token maps,
    shifted and unshifted windows, several mask classes, window counts that do not fill the last workgroup - and both against a
    plain fp32 host evaluation of y = x + proj(W-MSA(LN(x)))."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    csrc = os.path.join(ROOT, "waifu2x-tensorrt_amd", "csrc")
    exe = str(tmp_path / "attn_ab")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", csrc, os.path.join(ROOT, "tools", "ab", "attn_ab.hip"),
                    os.path.join(ROOT, "tools", "ab", "k_swinattn96_g2.hip"), os.path.join(csrc, "k_swinattn96.hip"), "-o", exe], check=True, timeout=900)
    out = subprocess.run([exe, "timing"], check=True, capture_output=True, text=True, timeout=600).stdout   # + the headline size, timed, run twice
    print(out)
    ab = re.findall(r"g2 vs shipped max\|dy\|=([0-9.]+) .* nan=(\d+)", out)
    host = re.findall(r"fp32 host evaluation: g2 max\|d\|=([0-9.]+) shipped max\|d\|=([0-9.]+)", out)
    assert len(ab) == 5 and len(host) >= 3, out
    assert re.search(r"shipped kernel run twice: 0 elements differ", out), out
    for dy, nan in ab:
        assert float(dy) <= 8e-3 and int(nan) == 0, out      # |y| is O(3): one fp16 ULP there is 2e-3; the two differ in summation order only
    for a, b in host:
        assert float(b) <= 1.2e-2 and float(b) <= float(a) + 4e-3, out


@pytest.mark.gpu
def test_fused_mlp96_tile_shapes_agree_with_each_other_and_with_fp32(tmp_path):
    """tools/ab/mlp96_ab.hip: the C = 96 MLP that ships (csrc/k_mlp96q.hip: v_mfma_f32_32x32x16_f16, rows through buffer resources,
    biases through LDS) against the round-2 kernel on 16x16x32 tiles (tools/ab/k_mlp96p.hip) - ragged row counts, with and without
    the LayerNorm-statistics output, both against a plain fp32 host evaluation with the exact erf GELU; the shipped kernel must
    reproduce itself bit for bit at the headline row count."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    csrc = os.path.join(ROOT, "waifu2x-tensorrt_amd", "csrc")
    exe = str(tmp_path / "mlp96_ab")
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", csrc, "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-fno-honor-nans",
                    os.path.join(ROOT, "tools", "ab", "mlp96_ab.hip"), os.path.join(ROOT, "tools", "ab", "k_mlp96p.hip"), os.path.join(csrc, "k_mlp96q.hip"), "-o", exe],
                   check=True, timeout=900)
    out = subprocess.run([exe, "timing"], check=True, capture_output=True, text=True, timeout=600).stdout
    print(out)
    ab = re.findall(r"16x16 vs 32x32 max\|dy\|=([0-9.]+) .*?nan=(\d+), max rel stats diff=([0-9.e+-]+)", out)
    host = re.findall(r"fp32 host evaluation: 16x16 max\|d\|=([0-9.]+) 32x32 max\|d\|=([0-9.]+)", out)
    assert len(ab) >= 8 and len(host) >= 6, out
    for dy, nan, ds in ab:
        assert float(dy) <= 8e-3 and int(nan) == 0 and float(ds) <= 5e-2, out   # |y| is O(4): one fp16 ULP there is 4e-3; summation order only
    for a, b in host:
        assert float(b) <= 6e-3 and float(b) <= float(a) + 2e-3, out
    assert re.search(r"32x32 kernel run twice: 0 elements differ", out), out
