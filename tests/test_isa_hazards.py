"""The device ISA of every kernel file (waifu2x-tensorrt_amd/build/<file>.isa.s, left by the Makefile) is checked for the one code-generation
gap this library has hit: hipcc puts no wait states between two DEPENDENT matrix instructions of DIFFERENT shapes (round 5: a
v_mfma_f32_16x16x16_f16 accumulating onto the result of a v_mfma_f32_16x16x32_f16 read a half-written accumulator - wrong and run-to-run
different outputs in three of four builds; profiles/r5_kernels/a96_mixed_chain.txt).  No GPU needed: the listings are read as text."""
import glob
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("isa_mfma_chain", os.path.join(ROOT, "tools", "isa_mfma_chain.py"))
chain = importlib.util.module_from_spec(spec)
spec.loader.exec_module(chain)


def test_the_scanner_sees_a_mixed_chain_and_ignores_same_shape_and_rewritten_registers(tmp_path):
    bad = tmp_path / "bad.s"
    bad.write_text("\n".join([
        "kernel:",
        "\tv_mfma_f32_16x16x32_f16 v[0:3], v[8:11], v[12:15], 0",
        "\tv_add_f32_e32 v20, v21, v22",
        "\tv_mfma_f32_16x16x16_f16 v[0:3], v[16:17], v[18:19], v[0:3]",       # different shape, 1 wait state: the hazard
        "\ts_endpgm"]))
    assert len(chain.scan(str(bad), 10)) == 1
    ok = tmp_path / "ok.s"
    ok.write_text("\n".join([
        "\tv_mfma_f32_16x16x32_f16 v[0:3], v[8:11], v[12:15], 0",
        "\tv_mfma_f32_16x16x32_f16 v[0:3], v[8:11], v[12:15], v[0:3]",        # same shape back to back: forwarded by the hardware
        "\tv_mfma_f32_16x16x32_f16 v[4:7], v[8:11], v[12:15], 0",
        "\ts_nop 7", "\ts_nop 1",
        "\tv_mfma_f32_16x16x16_f16 v[4:7], v[16:17], v[18:19], v[4:7]",       # different shape behind 10 wait states
        "\tv_mfma_f32_16x16x32_f16 v[24:27], v[8:11], v[12:15], 0",
        "\tv_cvt_pk_f16_f32 v24, v30, v31", "\tv_cvt_pk_f16_f32 v25, v30, v31", "\tv_mov_b32_e32 v26, 0", "\tv_mov_b32_e32 v27, 0",
        "\tv_mfma_f32_16x16x16_f16 v[24:27], v[16:17], v[18:19], v[24:27]",   # the registers were rewritten in between: no dependency
        "\ts_endpgm"]))
    assert chain.scan(str(ok), 10) == []


def test_no_kernel_file_chains_matrix_instructions_of_different_shapes_through_one_accumulator():
    pkg = os.path.join(ROOT, "waifu2x-tensorrt_amd")
    sources = sorted(glob.glob(os.path.join(pkg, "csrc", "k_*.hip")))
    assert sources
    total = 0
    for src in sources:
        lst = os.path.join(pkg, "build", os.path.basename(src)[:-4] + ".isa.s")
        assert os.path.exists(lst), f"{lst} is missing - `make -C waifu2x-tensorrt_amd` (__graft_entry__.build()) writes one listing per kernel file"
        assert os.path.getmtime(lst) >= os.path.getmtime(src) - 1, f"{lst} is older than {src}: rebuild"
        total += sum(1 for l in open(lst) if l.strip().startswith("v_mfma"))
        assert chain.scan(lst, 10) == [], f"{lst}: dependent matrix instructions of different shapes without the wait states the hardware needs"
    assert total > 1000       # (the listings are the real ones: ~6 000 matrix instructions in the library)
