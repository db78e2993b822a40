"""Measured ceiling of the hash-grid request shape on this GPU (immoco_probe_gather): random aligned
8/16-byte loads, 1 024 000 lanes x 64 loads (= motion encode forward's 16 levels x 4 corner pairs)."""
import ctypes as C, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from miccai24_immoco_amd import _lib as L

torch.cuda.init()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
out = []
for nbytes in (16, 8):
    for fp_mb in (1, 2, 4, 8, 16, 64, 256):
        ms = C.c_float()
        L.check(L.lib().immoco_probe_gather(fp_mb << 20, nbytes, 1024000, 64, 5, st, C.byref(ms)), "probe")
        req = 1024000 * 64
        out.append({"bytes_per_load": nbytes, "footprint_MB": fp_mb, "ms": round(ms.value, 4),
                    "G_requests_per_s": round(req / ms.value / 1e6, 1)})
        print(out[-1], flush=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "probe_gather.json"), "w"), indent=1)
