"""round 5, last GPU call: build the experiment library for tools/experiments/r05_pano_bwd_isolation.py WITHOUT touching the product
source: a patched temporary copy of csrc/embed.hip whose pano_embed_bwd launcher reads, at every call,
    ETP_PANO_BWD_LDS    bytes of dynamic LDS requested per launch (product: 160 KB = CU-exclusive; 12288 = what the kernels use)
    ETP_PANO_BWD_DRAIN  1: hipDeviceSynchronize() before the three launches; 2: before AND after them (the kernels run alone)
-> etpnav_amd/build/libetp_panoexpt.so (git-ignored; travels with the gpurun snapshot)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from etpnav_amd import build as b  # noqa: E402

src = open(os.path.join(b.HERE, "csrc", "embed.hip")).read()
old_smem = "  const size_t smem = 160 * 1024;\n"
assert src.count(old_smem) == 1
src = src.replace(old_smem, '''  const char* e_lds = getenv("ETP_PANO_BWD_LDS");
  const size_t smem = e_lds ? (size_t)atol(e_lds) : 160 * 1024;
  const char* e_drain = getenv("ETP_PANO_BWD_DRAIN");
  const int drain = e_drain ? atoi(e_drain) : 0;
  if (drain >= 1) ETP_CHECK_HIP(hipDeviceSynchronize());
''')
tail = '''  ETP_CHECK_LAUNCH("pano_embed_bwd");
  return ETP_OK;
}

int gmap_embed_fwd('''
assert src.count(tail) == 1
src = src.replace(tail, '''  ETP_CHECK_LAUNCH("pano_embed_bwd");
  if (drain >= 2) ETP_CHECK_HIP(hipDeviceSynchronize());
  return ETP_OK;
}

int gmap_embed_fwd(''')
# the attribute is a maximum: set it to the largest value any mode asks for
old_attr = "hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));"
assert src.count(old_attr) == 1
src = src.replace(old_attr, "hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));")
tmp = os.path.join(b.HERE, "csrc", "embed_expt_tmp.hip")
obj = "/tmp/embed_expt.o"
out = os.path.join(b.HERE, "build", "libetp_panoexpt.so")
try:
    open(tmp, "w").write(src)
    subprocess.check_call([b.HIPCC, *b.FLAGS, "-c", tmp, "-o", obj])
finally:
    os.remove(tmp)
objs = [obj if s == "embed.hip" else os.path.join(b.HERE, "build", s.replace(".hip", ".o")) for s in b.SOURCES]
subprocess.check_call([b.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs])
print("built", out)
