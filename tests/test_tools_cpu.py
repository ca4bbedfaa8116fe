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
