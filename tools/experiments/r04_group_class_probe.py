"""Grouped weight-gradient launch (one text / panorama / navigation layer's four products) on the 128x128 and the 256x128 class:
back-to-back isolated launches over rotating operand sets, per token count.  ETP_MM32_GROUP is read per call."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT)
from tools.mm32_probe import group_time

out = {}
for Mt in (2560, 1152, 512, 8192):
    for cls in ("128", "256"):
        _lib.set_option("MM32_GROUP", cls)
        out[f"tokens{Mt}:{cls}"] = round(group_time(Mt=Mt, iters=24), 2)
print(json.dumps(out, indent=1))
