"""The profile-reduction helpers bench.py depends on: rocprofv3 kernel names -> the class names the bench line prints
(bench.py matches `roofline.kernel` against profiles/r04_bench_kernel_stats.csv and profiles/r04_pmc_traffic.json through them)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.pmc_sq import short                      # noqa: E402
from tools.pmc_traffic import bench_name            # noqa: E402


def test_kernel_class_names_of_both_gemm_families():
    cases = {
        "void etp::mm32::group_kernel<float, true, true, 128, 128, 2>(etp::GemmGroup)": "mm32_group<bf16,f32,TN,128x128,s2>",
        "void etp::mm32::group_kernel<float, true, true, 256, 128, 3>(etp::GemmGroup)": "mm32_group<bf16,f32,TN,256x128,s3>",
        "void etp::mm32::kernel<unsigned short, false, false, 128, 64, 3>(etp::GemmArgs)": "mm32<bf16,bf16,NT,128x64,s3>",
        "void etp::mm32::kernel<float, false, true, 128, 128, 2>(etp::GemmArgs)": "mm32<bf16,f32,NN,128x128,s2>",
        "void etp::gemm_dma_kernel<unsigned short, float, false, true, 32, 64, 4>(etp::GemmArgs)": "gemm_dma<bf16,f32,NN,32x64,s4>",
        "void etp::gemm_group_kernel<float, float, true, true, 128, 128, 2>(etp::GemmGroup)": "gemm_group<f32,f32,TN,128x128,s2>",
        "void etp::gemm_kernel<unsigned short, unsigned short, false, false, 64, 64>(etp::GemmArgs)": "gemm<bf16,bf16,NT,64x64>",
    }
    for raw, want in cases.items():
        assert short(raw) == want, (raw, short(raw))


def test_traffic_rows_drop_the_ring_depth_and_other_kernels_keep_their_name():
    assert bench_name("void etp::mm32::group_kernel<float, true, true, 128, 128, 2>(etp::GemmGroup)") == "mm32_group<bf16,f32,TN,128x128>"
    assert bench_name("void etp::gemm_dma_kernel<unsigned short, float, false, false, 64, 64, 4>(etp::GemmArgs)") == "gemm_dma<bf16,f32,NT,64x64>"
    assert bench_name("void etp::ln_bwd_s_kernel<unsigned short, 3>(float const*, float const*)") == "ln_bwd_s_kernel"
    assert bench_name("etp::cast_f32_bf16_kernel(float const*, unsigned short*, long)") == "cast_f32_bf16_kernel"
    # the key bench.py derives from its own kernel label must hit the same row
    label = "mm32_group<bf16,f32,TN,128x128,s2>"
    assert label.split(",s")[0] + ">" == "mm32_group<bf16,f32,TN,128x128>"


def test_kernel_resource_policy_holds_for_the_built_objects():
    """tools/kernel_resources.py: no kernel outside the matrix-core families holds AGPRs, no kernel uses scratch --
    read from the code-object metadata of the objects etpnav_amd.build produced (the build itself fails on a violation; this test
    keeps the parser honest and pins the production row kernels' register budgets)."""
    import pytest
    from etpnav_amd import build as b
    from tools import kernel_resources as kr
    objs = [os.path.join(b.HERE, "build", s.replace(".hip", ".o")) for s in b.SOURCES]
    if not all(os.path.exists(o) for o in objs):
        pytest.skip("objects not built here")
    rows = kr.audit()
    assert len(rows) > 250 and not [r for r in rows if r["violation"]], [r["full"] for r in rows if r["violation"]]
    by = {r["name"]: r for r in rows}
    for k in ("pano_embed_bwd_kernel<unsigned short, 3, 0>", "pano_embed_bwd_kernel<unsigned short, 3, 1>", "pano_embed_bwd_kernel<unsigned short, 3, 2>",
              "pano_embed_fwd_kernel<unsigned short, 3>", "gmap_embed_bwd_kernel<float, 3, 7>"):
        assert int(by[k]["agpr_count"]) == 0 and int(by[k]["vgpr_count"]) <= 256 and int(by[k]["private_segment_fixed_size"]) == 0, by[k]
    assert any(int(r["agpr_count"]) > 0 and r["mfma"] for r in rows)      # the parser does see AGPRs where they are allowed
    # the parser on a literal note
    note = """
  - .agpr_count:     8
    .args:
      - .offset:         0
        .size:           8
    .name:           _ZN3etp3fooEv
    .private_segment_fixed_size: 16
    .sgpr_count:     10
    .vgpr_count:     20
    .vgpr_spill_count: 1
"""
    (k,) = kr.parse(note)
    assert k["agpr_count"] == "8" and k["private_segment_fixed_size"] == "16" and k["vgpr_spill_count"] == "1" and k["name"] == "_ZN3etp3fooEv"


def test_lds_dma_m0_discipline_scanner():
    """tools/kernel_resources.py::m0_scan: the LDS-DMA statements write M0 from inline asm the compiler cannot be told about;
    the build checks the ISA around every global_load_lds.  Here: the scanner accepts the two statement forms of the product (glds of
    gemm_mm32.hip, glds16 of gemm_tiles.h with its save / restore) and rejects each way the assumption can break; on the built objects
    the audit finds the DMA instructions and no violation."""
    import pytest
    from etpnav_amd import build as b
    from tools import kernel_resources as kr
    good = """
0000000000001000 <_ZN3etp4mm326kernelEv>:
	s_add_i32 s5, s60, 0x1000                                  // 000000001000: 8105FF3C
	s_mov_b32 m0, s5                                           // 000000001008: BEFC0005
	s_nop 0                                                    // 00000000100C: BF800000
	global_load_lds_dwordx4 v68, s[0:1]                        // 000000001010: DDF48000
	s_mov_b32 s1, m0                                           // 000000001018: BE81007C
	s_mov_b32 m0, s0                                           // 00000000101C: BEFC0000
	s_nop 0                                                    // 000000001020: BF800000
	global_load_lds_dwordx4 v2, off                            // 000000001024: DDF48000
	s_mov_b32 m0, s1                                           // 00000000102C: BEFC0001
	s_endpgm                                                   // 000000001030: BF810000

0000000000002000 <_ZN3etp9no_dma_m0Ev>:
	s_mov_b32 m0, s3                                           // 000000002000: BEFC0003
	s_movrels_b32 s4, s8                                       // 000000002004: BE840008
	s_endpgm                                                   // 000000002008: BF810000
"""
    stats, bad = kr.m0_scan(good)
    assert stats == {"_ZN3etp4mm326kernelEv": 2} and bad == []              # kernels without LDS-DMA are not this check's business
    head = "0000000000001000 <k>:\n"
    dma = "\ts_mov_b32 m0, s5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 v68, s[0:1]\n"
    for broken, what in [
            (head + "\ts_mov_b32 m0, s5\n\ts_add_i32 s5, s5, 4\n\tglobal_load_lds_dwordx4 v68, s[0:1]\n", "not directly behind"),
            (head + dma + "\ts_mov_b32 m0, s9\n\tv_add_f32 v0, v1, v2\n", "neither consumed"),
            (head + dma + "\ts_movrels_b32 s4, s8\n", "implicit M0 use"),
            (head + dma + "\ts_set_gpr_idx_on s3, gpr_idx(SRC0)\n", "implicit M0 use"),
            (head + dma + "\tv_readlane_b32 s4, v3, m0\n", "compiler-generated M0 use"),
            (head + dma + "\ts_mov_b32 s7, m0\n\tv_add_f32 v0, v1, v2\n", "M0 read outside"),
            (head + dma + "\tbuffer_load_dword v1, s[4:7], 0 offen lds\n", "implicit M0 use")]:
        _, bad = kr.m0_scan(broken)
        assert any(what in t for _, _, t in bad), (what, bad)
    objs = [os.path.join(b.HERE, "build", s.replace(".hip", ".o")) for s in b.SOURCES]
    if not all(os.path.exists(o) for o in objs):
        pytest.skip("objects not built here")
    nk, ni, m0bad = kr.m0_audit()
    assert nk > 50 and ni > 1000 and not m0bad, m0bad[:5]


def test_forcing_a_gemm_tile_class_switches_the_mm32_family_off():
    """_lib.force_gemm_tile: mm32_class is consulted before gemm.hip's tile choice, so a forced gemm.hip class must come
    with MM32=0 or eligible bf16 products keep running the mm32 kernel; "auto" hands both choices back to the library.  Round 6: the
    switches live in the library's own table (csrc/options.h, `etp_option_set`), not in os.environ."""
    from etpnav_amd import _lib
    old = {k: _lib.get_option(k) for k in ("GEMM_TILE", "MM32")}
    try:
        _lib.set_option("GEMM_TILE", None); _lib.set_option("MM32", None)
        _lib.force_gemm_tile("64s3")
        assert _lib.get_option("GEMM_TILE") == "64s3" and _lib.get_option("MM32") == "0"
        assert _lib.options()["GEMM_TILE"] == "64s3"
        _lib.force_gemm_tile("auto")
        assert _lib.get_option("GEMM_TILE") is None and _lib.get_option("MM32") is None
    finally:
        for k, v in old.items():
            _lib.set_option(k, v)


def test_library_switch_table_rejects_unknown_names_and_restores():
    """etp_option_set / etp_option_get (include/etpnav_hip.h): unknown names are an error, the ETP_ prefix is optional, values are
    truncated to 31 characters, `with _lib.option(...)` restores the previous value."""
    import pytest
    from etpnav_amd import _lib
    with pytest.raises(_lib.EtpError):
        _lib.set_option("NO_SUCH_SWITCH", "1")
    with pytest.raises(_lib.EtpError):
        _lib.get_option("NO_SUCH_SWITCH")
    before = _lib.get_option("MM32_GROUP")
    with _lib.option("ETP_MM32_GROUP", 256):
        assert _lib.get_option("MM32_GROUP") == "256"
        with _lib.option("MM32_GROUP", "x" * 100):
            assert _lib.get_option("MM32_GROUP") == "x" * 31
        assert _lib.get_option("MM32_GROUP") == "256"
    assert _lib.get_option("MM32_GROUP") == before


def test_bench_cpu_baseline_leg_measures_the_full_batch_when_it_fits():
    """bench.py's `cpu_baseline` (the oracle timed beside the GPU path): the quoted mode runs the whole batch when
    1 warm-up + 5 timed steps fit the leg's budget and says so; a budget that cannot hold the batch falls back to a scaled share and
    says that instead."""
    import bench
    w = dict(task="r2r", B=4, L=8, V=6, G=5, image_feat_size=768)
    r = bench.cpu_baseline(w, dict(image_feat_size=768), budget_s=60.0, train=True)
    assert r["full_batch"] and r["mode"] == "train" and r["value"] == r["train_value"] > 0 and r["eval_value"] > 0
    assert "the full batch of 4 episodes (measured, not scaled)" in r["sample"] and r["kind"] == "port" and r["cores"] >= 1
    r = bench.cpu_baseline(w, dict(image_feat_size=768), budget_s=1e-3, train=False)
    assert not r["full_batch"] and r["mode"] == "eval" and "1 of the 4 episodes per step" in r["sample"] and r["value"] == r["eval_value"] > 0


def test_bench_line_watchdog_prints_the_fallback_once_and_leaves_with_status_zero():
    """bench.py LineWatchdog (N > 1): the metric is measured when the timed region ends; a collective of the reported legs behind it that
    never returns on some rank must not cost the job its one JSON line.  (a) main thread stuck: the fallback line appears after the
    timeout, exit status 0; (b) the real line went out first: the watchdog prints nothing more and only ends the process; (c) a rank that
    does not own the line prints nothing."""
    import subprocess
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pre = "import sys, time; sys.path.insert(0, %r); import bench; " % root
    cases = [
        ("w = bench.LineWatchdog(0.5, lambda: '{\"fallback\": 1}'); w.start(); time.sleep(60)", ['{"fallback": 1}']),
        ("w = bench.LineWatchdog(0.5, lambda: '{\"fallback\": 1}'); w.start(); w.print_line('{\"real\": 1}'); time.sleep(60)", ['{"real": 1}']),
        ("w = bench.LineWatchdog(0.5, None); w.start(); time.sleep(60)", []),
    ]
    for code, want in cases:
        t0 = time.time()
        r = subprocess.run([sys.executable, "-c", pre + code], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-1000:]
        assert [l for l in r.stdout.splitlines() if l.startswith("{")] == want, (code, r.stdout)
        assert time.time() - t0 < 45, "the watchdog did not end the process"


def test_two_gpu_rccl_cases_are_the_last_file_of_the_suite():
    """The driver runs `pytest -x`; the four cases of tests/test_zz_two_gpu.py would be RCCL's first contact with a second rank, so they
    must come after every other GPU test (pytest collects files in name order) and nowhere else may a test skip on device_count() < 2."""
    import glob
    root = os.path.dirname(os.path.abspath(__file__))
    files = sorted(os.path.basename(f) for f in glob.glob(os.path.join(root, "test_*.py")))
    assert files[-1] == "test_zz_two_gpu.py", files
    for f in files[:-1]:
        if f == os.path.basename(__file__):
            continue
        assert "device_count() < 2" not in open(os.path.join(root, f)).read(), f
