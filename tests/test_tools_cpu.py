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
