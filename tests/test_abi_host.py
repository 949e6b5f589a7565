"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol
include/av1mi.h declares, header bytes match the oracle's, the reference-interface mirror behaves
like the reference (error taxonomy, ConcurrencyPlan), and there is no CPU fallback."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def av1mi():
    lib = os.path.join(ROOT, "av1-base_amd", "libav1mi.so")
    if not os.path.exists(lib):
        import importlib.util
        spec = importlib.util.spec_from_file_location("av1mi_build", os.path.join(ROOT, "av1-base_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()
    import av1mi as m
    return m


def test_exports_every_declared_symbol(av1mi):
    hdr = open(os.path.join(ROOT, "include", "av1mi.h")).read()
    declared = set(re.findall(r"\b(av1mi_[a-z0-9_]+)\s*\(", hdr)) - {"av1mi_progress_cb"}
    assert declared == set(av1mi.ABI_SYMBOLS), (declared ^ set(av1mi.ABI_SYMBOLS))
    lib = C.CDLL(os.path.abspath(av1mi.LIB_PATH))
    for sym in declared:
        assert hasattr(lib, sym), sym
    hdr_version = int(re.search(r"#define AV1MI_ABI_VERSION (\d+)", hdr).group(1))
    assert av1mi._lib.av1mi_abi_version() == hdr_version == av1mi.ABI_VERSION


def test_structure_layout_is_pinned(av1mi):
    """av1mi_struct_sizes: the library's own sizes / offsets of every ABI structure equal the ctypes mirror's (the mirror refuses to
    import otherwise) and the constants written into the Rust shim (integration/mi355x.rs `const _: () = assert!(...)`), so a field
    added to av1mi_params and forgotten in a binding fails here instead of corrupting memory."""
    lib_sizes = av1mi.struct_sizes()
    assert len(lib_sizes) == 12 and lib_sizes == av1mi.mirror_sizes()
    rs = open(os.path.join(ROOT, "integration", "mi355x.rs")).read()
    # the Rust mirror declares one u32 per av1mi_params field
    body = rs[rs.index("pub struct Av1miParams"):]
    body = body[:body.index("}")]
    assert len(re.findall(r"pub \w+: u32", body)) * 4 == lib_sizes[0] == C.sizeof(av1mi.Params)
    m = re.search(r"size_of::<Av1miParams>\(\) == (\d+) \* 4", rs)
    assert m and int(m.group(1)) * 4 == lib_sizes[0]
    m = re.search(r"size_of::<Av1miReport>\(\) == (\d+)", rs)
    assert m and int(m.group(1)) == lib_sizes[2]
    assert "pub const AV1MI_ABI_VERSION: u32 = %d;" % av1mi.ABI_VERSION in rs


def test_free_takes_null_and_plain_heap_blocks(av1mi):
    """av1mi_free: NULL is a no-op; a block that is not one of the pool's page-locked ones goes to free()."""
    av1mi._lib.av1mi_free(None)
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    av1mi._lib.av1mi_free(libc.malloc(4096))


def test_cq_mapping_matches_aom_table(av1mi):
    assert av1mi.cq_to_qindex(30) == 120     # CQ 30 <-> base_q_idx 120
    assert av1mi.cq_to_qindex(0) == 0 and av1mi.cq_to_qindex(63) == 255
    assert [av1mi.cq_to_qindex(i) for i in range(1, 5)] == [4, 8, 12, 16]


@pytest.mark.parametrize("w,h,bd,cdf", [(64, 64, 8, 1), (200, 120, 8, 1), (1920, 1080, 10, 1), (3840, 2160, 10, 0), (200, 120, 8, 0)])
def test_headers_match_oracle(av1mi, oracle, w, h, bd, cdf):
    """The product's C++ header writer must emit exactly the oracle's sequence + frame header."""
    p = av1mi.default_params(w, h, bd, cdf_update=cdf)
    seq, fh, bits = av1mi.write_headers(p)
    cfg = oracle.default_config(w, h, bd, min_bs_log2=5, max_bs_log2=5, disable_cdf_update=0 if cdf else 1)
    if w * h <= 200 * 120:
        src = oracle.synthclip_frame(w, h, bd, seed=1, t=0)
        tu, _, _ = oracle.encode_frame(cfg, src)
        assert tu[2:2 + len(seq)] == seq
        # OBU_FRAME payload starts right after its header + leb128 size
        off = 2 + len(seq) + 1
        while tu[off] & 0x80:
            off += 1
        off += 1
        assert tu[off:off + len(fh)] == fh
    else:
        buf = C.create_string_buffer(64)
        n = oracle.lib().av1o_write_sequence_header(C.byref(cfg), buf, 64)
        assert buf.raw[:n] == seq


def test_color_range_bit_of_the_sequence_header(av1mi, oracle):
    """color_config.color_range: studio (0) unless asked otherwise - Y4M input and the reference's ffmpeg -> SVT-AV1 pipeline
    carry limited-range video; the bit must follow av1mi_params.color_range exactly as the oracle writes it."""
    seqs = {}
    for cr in (0, 1):
        p = av1mi.default_params(200, 120, 10, color_range=cr)
        seq, _, _ = av1mi.write_headers(p)
        cfg = oracle.default_config(200, 120, 10, color_range=cr)
        buf = C.create_string_buffer(64)
        n = oracle.lib().av1o_write_sequence_header(C.byref(cfg), buf, 64)
        assert buf.raw[:n] == seq
        seqs[cr] = seq
    assert seqs[0] != seqs[1] and av1mi.default_params(64, 64, 8).color_range == 0
    with pytest.raises(av1mi.EncodeFailed):
        av1mi.write_headers(av1mi.default_params(64, 64, 8, color_range=2))


def test_colour_description_of_the_sequence_header(av1mi, oracle, golden_cases):
    """color_config's colour description (BASELINE config 5 "8K 10-bit HDR"; the reference's ffmpeg -> SVT-AV1 pipeline passes colour
    metadata through, av1an.rs:90): av1mi_params.{color_primaries, transfer_characteristics, matrix_coefficients} = 9 / 16 / 9 writes the
    same sequence header as the oracle - whose stream dav1d decoded and libavif read the code points back from (tests/golden/
    k200x120_hdr_bt2020_pq_10b, tools/make_golden.py) - and none (0 / 0 / 0) leaves the header as it was."""
    m = [g for g in golden_cases if g["name"] == "k200x120_hdr_bt2020_pq_10b"][0]
    p = av1mi.default_params(200, 120, 10, color_primaries=9, transfer_characteristics=16, matrix_coefficients=9)
    seq, _, _ = av1mi.write_headers(p)
    assert m["obu"][2:2 + len(seq)] == seq   # temporal delimiter, then the sequence header OBU
    # color_config: high_bitdepth 1, mono 0, description present 1, then the three code points
    bits = "".join("{:08b}".format(b) for b in seq[2:])
    i = bits.find("101" + "{:08b}{:08b}{:08b}".format(9, 16, 9))
    assert i > 0
    plain, _, _ = av1mi.write_headers(av1mi.default_params(200, 120, 10))
    assert len(seq) == len(plain) + 3
    cfg = oracle.default_config(200, 120, 10, color_primaries=1, transfer_characteristics=1, matrix_coefficients=1, color_range=1)
    buf = C.create_string_buffer(64)
    n = oracle.lib().av1o_write_sequence_header(C.byref(cfg), buf, 64)
    assert buf.raw[:n] == av1mi.write_headers(av1mi.default_params(200, 120, 10, color_primaries=1, transfer_characteristics=1, matrix_coefficients=1, color_range=1))[0]
    for bad in (dict(color_primaries=256), dict(color_primaries=1, transfer_characteristics=13, matrix_coefficients=0)):   # sRGB + identity implies 4:4:4
        with pytest.raises(av1mi.EncodeFailed):
            av1mi.write_headers(av1mi.default_params(64, 64, 8, **bad))


def test_y4m_with_frame_parameters_is_read_sequentially(av1mi, tmp_path):
    """"FRAME <params>\n" markers are legal Y4M: such a clip cannot be cut into frames by offset, so its length is unknown to the probe
    (0) instead of wrong - av1mi_encode_file reads it frame by frame (GPU test: test_encode_file_ragged_and_empty_inputs)."""
    w, h = 72, 56
    f = tmp_path / "p.y4m"
    f.write_bytes(b"YUV4MPEG2 W%d H%d F25:1 C420jpeg\n" % (w, h) + (b"FRAME Ip\n" + bytes(w * h * 3 // 2)) * 3)
    ci = av1mi.probe_y4m(f)
    assert (ci.width, ci.height, ci.frames) == (w, h, 0)


def test_probe_y4m_reports_rate_range_and_length(av1mi, tmp_path):
    """av1mi_probe_y4m: what JobMetrics.total_frames / bitrate_kbps need (frame count from the file size, frame rate),
    plus the XCOLORRANGE tag; malformed input is AV1MI_E_FORMAT, a missing file -errno."""
    w, h, n = 72, 56, 5
    f = tmp_path / "a.y4m"
    f.write_bytes(b"YUV4MPEG2 W%d H%d F30000:1001 Ip A1:1 C420p10 XYSCSS=420P10 XCOLORRANGE=FULL\n" % (w, h) + (b"FRAME\n" + bytes(w * h * 3)) * n)
    ci = av1mi.probe_y4m(f)
    assert (ci.width, ci.height, ci.bit_depth, ci.fps_num, ci.fps_den, ci.color_range, ci.frames) == (w, h, 10, 30000, 1001, 1, n)
    g = tmp_path / "b.y4m"
    g.write_bytes(b"YUV4MPEG2 W%d H%d F25:1 C420jpeg\n" % (w, h) + (b"FRAME\n" + bytes(w * h * 3 // 2)) * 2)
    ci = av1mi.probe_y4m(g)
    assert (ci.bit_depth, ci.fps_num, ci.fps_den, ci.color_range, ci.frames) == (8, 25, 1, 0, 2)
    bad = tmp_path / "c.y4m"
    bad.write_bytes(b"RIFF....")
    with pytest.raises(av1mi.EncodeFailed) as ei:
        av1mi.probe_y4m(bad)
    assert ei.value.code == av1mi.E_FORMAT
    with pytest.raises(av1mi.EncodeIo):
        av1mi.probe_y4m(tmp_path / "missing.y4m")


def test_invalid_parameters_map_to_failed(av1mi):
    p = av1mi.default_params(61, 64, 8)   # odd width (4:2:0 needs even sizes; multiples of 8 are not required)
    with pytest.raises(av1mi.EncodeFailed) as ei:
        av1mi.write_headers(p)
    assert ei.value.code == av1mi.E_INVALID_ARG
    p = av1mi.default_params(64, 64, 12)
    with pytest.raises(av1mi.EncodeFailed):
        av1mi.write_headers(p)
    p = av1mi.default_params(64, 64, 8, keyint=240, me_range=12)   # motion search range: 8 or 16 only
    with pytest.raises(av1mi.EncodeFailed) as ei:
        av1mi.write_headers(p)
    assert ei.value.code == av1mi.E_INVALID_ARG
    av1mi.write_headers(av1mi.default_params(64, 64, 8, keyint=240))   # the reference's production keyint (av1an.rs:14)
    av1mi.write_headers(av1mi.default_params(7680, 4320, 10))   # 120 x 68 superblocks: tiles of 2 x 2 superblocks
    p = av1mi.default_params(64 * 129, 64, 8)   # more than 128 superblock columns: would need tiles wider than two superblocks
    with pytest.raises(av1mi.EncodeFailed) as ei:
        av1mi.write_headers(p)
    assert ei.value.code == av1mi.E_UNSUPPORTED


def test_error_taxonomy_mirrors_reference(av1mi):
    """av1an.rs:17-30: non-zero exit -> Av1anFailed(code); spawn failure -> Io."""
    with pytest.raises(av1mi.EncodeFailed) as ei:
        av1mi._raise_for(3, "x")
    assert ei.value.code == 3 and isinstance(ei.value, av1mi.EncodeError)
    with pytest.raises(av1mi.EncodeIo) as ei:
        av1mi._raise_for(-2)
    assert ei.value.errno == 2
    av1mi._raise_for(0)


def test_no_device_fails_loudly_not_silently(av1mi, tmp_path):
    """Without a HIP device the product must refuse (no CPU fallback): ctx creation and
    run_mi355x both report failure, and no output file is left behind."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(av1mi.EncodeFailed) as ei:
        av1mi.Context(0)
    assert ei.value.code == av1mi.E_NO_DEVICE
    y4m = tmp_path / "in.y4m"
    y4m.write_bytes(b"YUV4MPEG2 W64 H64 F30:1 Ip C420jpeg\nFRAME\n" + bytes(64 * 64 * 3 // 2))
    out = tmp_path / "out.ivf"
    plan = av1mi.derive_plan(8)
    with pytest.raises(av1mi.EncodeFailed):
        av1mi.run_mi355x(av1mi.EncodeParams(y4m, out, tmp_path, plan))
    assert not out.exists() and not list(tmp_path.glob("out.ivf.tmp*"))


def test_missing_input_maps_to_io_error(av1mi, tmp_path):
    plan = av1mi.derive_plan(32)
    with pytest.raises(av1mi.EncodeIo) as ei:
        av1mi.run_mi355x(av1mi.EncodeParams(tmp_path / "nope.y4m", tmp_path / "o.ivf", tmp_path, plan))
    assert ei.value.errno == 2   # ENOENT, like io::Error from a failed spawn/open


def test_bad_y4m_maps_to_format_error(av1mi, tmp_path):
    bad = tmp_path / "bad.y4m"
    bad.write_bytes(b"RIFF....not a y4m")
    with pytest.raises(av1mi.EncodeFailed) as ei:
        av1mi.run_mi355x(av1mi.EncodeParams(bad, tmp_path / "o.ivf", tmp_path, av1mi.derive_plan(4)))
    assert ei.value.code == av1mi.E_FORMAT


def test_concurrency_plan_rules(av1mi):
    """concurrency.rs:67-89: 8 workers if >= 32 cores else 4; 1 job if >= 24 cores else 2;
    utilisation clamped to [0.5, 1.0]."""
    assert av1mi.derive_plan(32).av1an_workers == 8 and av1mi.derive_plan(31).av1an_workers == 4
    assert av1mi.derive_plan(24).max_concurrent_jobs == 1 and av1mi.derive_plan(23).max_concurrent_jobs == 2
    assert av1mi.derive_plan(10, 0.1).target_threads == 5 and av1mi.derive_plan(10, 2.0).target_threads == 10
    assert av1mi.derive_plan(64, workers_override=3, max_jobs_override=5).av1an_workers == 3


def test_product_never_references_the_oracle():
    """The product tree must not import, include or link anything under oracle/."""
    bad = []
    for dp, _, fns in os.walk(os.path.join(ROOT, "av1-base_amd")):
        if "build" in dp.split(os.sep)[-1:]:
            continue
        for fn in fns:
            if fn.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, fn), errors="ignore").read()
                if re.search(r'#include\s*"[^"]*oracle|import\s+av1o|from\s+av1o|libav1o|oracle/_', txt):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_job_execute_failure_paths_without_gpu(av1mi, tmp_path):
    """The caller's encode segment (job_executor.rs:266-317, 413-436): the stage sequence and failure strings of the
    reference, and its temp-dir lifecycle, on the paths that need no GPU."""
    # missing input: the encoder reports an I/O error -> state failed, "IO error: ...", chunks dir removed
    rc, stages, m, err = av1mi.job_execute("j1", tmp_path / "nope.y4m", tmp_path / "o.ivf", tmp_path / "tmp")
    assert rc == -2 and stages == ["encoding", "failed"] and err.startswith("IO error: ")
    assert not (tmp_path / "tmp" / "chunks_j1").exists() and not (tmp_path / "o.ivf").exists()
    # not a Y4M: encoder failure code -> "MI355X encoder failed with exit code: N"
    bad = tmp_path / "bad.y4m"
    bad.write_bytes(b"RIFF....")
    rc, stages, m, err = av1mi.job_execute("j2", bad, tmp_path / "o.ivf", tmp_path / "tmp")
    assert rc == av1mi.E_FORMAT and stages == ["encoding", "failed"] and err == "MI355X encoder failed with exit code: %d" % av1mi.E_FORMAT
    assert m.stage == b"failed" and not (tmp_path / "tmp" / "chunks_j2").exists()


def test_integration_patches_apply_to_the_reference(tmp_path):
    """integration/*.patch (the backend selector, the startup gate, the call site) apply cleanly to the reference's
    files.  Runs where the reference tree is present (this container); it is not part of what travels to the GPU box."""
    import shutil
    import subprocess
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "crates")) or not shutil.which("patch"):
        pytest.skip("no reference tree / no patch(1) here")
    work = tmp_path / "ref"
    shutil.copytree(os.path.join(ref, "crates"), work / "crates")
    patches = sorted(f for f in os.listdir(os.path.join(ROOT, "integration")) if f.endswith(".patch"))
    assert len(patches) == 6
    for f in patches:
        r = subprocess.run(["patch", "-p1", "--forward", "-i", os.path.join(ROOT, "integration", f)], cwd=work, capture_output=True, text=True)
        assert r.returncode == 0, (f, r.stdout, r.stderr)
    shutil.copy(os.path.join(ROOT, "integration", "mi355x.rs"), work / "crates" / "daemon" / "src" / "encode" / "mi355x.rs")
    ex = (work / "crates" / "daemon" / "src" / "job_executor.rs").read_text()
    assert "EncoderBackend::Mi355x => run_mi355x(&params, cq_level)" in ex and "EncoderBackend::Av1an => run_av1an(&params)" in ex
    assert 'pub mod mi355x;' in (work / "crates" / "daemon" / "src" / "encode" / "mod.rs").read_text()
