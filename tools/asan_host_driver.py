"""Driver of tools/asan_host_check.sh: exercises the host-side code paths of the (ASan-instrumented) library without a GPU."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
from etpnav_amd import _lib
_lib.LIB_PATH = os.environ["ETP_ASAN_LIB"]
import torch  # noqa: E402
L = _lib.lib()
from etpnav_amd.planner import make_c_config, default_config  # noqa: E402
n_ok = 0
for kw in (dict(), dict(task_type="rxr"), dict(use_lang2visn_attn=True), dict(num_l_layers=1, num_pano_layers=0, num_x_layers=0),
           dict(use_depth_embedding=False, graph_sprels=False)):
    task = kw.pop("task_type", "r2r")
    for dt in (torch.float32, torch.bfloat16):
        c = make_c_config(default_config(task, **kw), dt)
        h = L.etp_planner_create(ctypes.byref(c))
        assert h, L.etp_last_error()
        n = L.etp_planner_param_count(h)
        info = _lib.ParamInfo()
        end = 0
        for i in range(n):
            assert L.etp_planner_param_info(h, i, ctypes.byref(info)) == 0
            end = max(end, info.offset)
        assert L.etp_planner_param_info(h, n, ctypes.byref(info)) != 0          # out-of-range index is refused, not read
        assert L.etp_planner_arena_elems(h) > end
        for B, Lt, V, G in ((1, 1, 1, 1), (2, 20, 17, 9), (32, 80, 36, 16), (16, 512, 36, 64)):
            for fn, args in (("etp_txt_stash_bytes", (B, Lt)), ("etp_txt_ws_bytes", (B, Lt)), ("etp_pano_stash_bytes", (B, V)),
                             ("etp_pano_ws_bytes", (B, V)), ("etp_nav_stash_bytes", (B, Lt, G)), ("etp_nav_ws_bytes", (B, Lt, G)),
                             ("etp_nav_kv_bytes", (B, Lt))):
                assert getattr(L, fn)(h, *args) > 0
            if c.use_lang2visn:
                assert L.etp_mlm_stash_bytes(h, B, Lt, G, 7) > 0 and L.etp_mlm_ws_bytes(h, B, Lt, G, 7) > 0
        assert L.etp_planner_bind(h, None, None, None) != 0                      # null arenas refused
        assert L.etp_txt_fwd(h, None, None, 2, 3, None, None, None) != 0         # unbound / null arguments refused
        assert L.etp_planner_set_dropout(h, 0.1, 0.1, 0.1, 0.4, 123) == 0 and L.etp_planner_set_dropout(h, 1.5, 0, 0, 0, 0) != 0
        # round 4 entry points: the K/V cache under the batched rollout call and the stamp ring refuse null / empty arguments
        assert L.etp_nav_kv_repeat(h, None, 2, 3, 4, None, None) != 0 and L.etp_nav_kv_sum_steps(h, None, 2, 3, 4, None, None) != 0
        assert L.etp_nav_kv_bytes(h, 4 * 2, 3) >= L.etp_nav_kv_bytes(h, 2, 3)
        assert L.etp_nav_kv_grad_elems(h, 2, 3) == c.n_x * 2 * 3 * 2 * c.hidden
        L.etp_planner_destroy(h)
        n_ok += 1
out = (ctypes.c_float * 4096)()
assert L.etp_dropout_multipliers(0.1, 77, 1, 3, 2, 4096, out) == 0
kept = sum(1 for v in out if v > 0)
assert 3500 < kept < 3900
assert L.etp_ln_bwd_part_bytes(2560, 768) == 640 * 2 * 768 * 4           # one row per wavefront: 640 slabs of [2][H] fp32
assert L.etp_stamp_count() == 0                                           # no sink installed: marks are no-ops
bad = _lib.Config()
assert not L.etp_planner_create(ctypes.byref(bad)) and b"unsupported" in L.etp_last_error()
# explicit-graph recorder bookkeeping without a device: begin/abort must not leak or double free
print(f"ASan host check: {n_ok} planner layouts built and destroyed, argument checks and dropout generator exercised: clean")
