"""immoco_probe_gather over request size, footprint and wave lifetime (loads per lane): what the chip sustains for
divergent line-granular loads.  GPU box.   python tools/probe_matrix.py"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from miccai24_immoco_amd import _lib
torch.zeros(1, device="cuda")
L = _lib.lib()
TOTAL = 1 << 26          # loads per launch
print("bytes footprint loads/lane    lanes      ms   Gloads/s")
for nbytes in (8, 16):
    for fp in (1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 21, 1 << 22, 1 << 23):
        for lpl in (4, 8, 16, 64):
            lanes = TOTAL // lpl
            ms = ctypes.c_float(0)
            rc = L.immoco_probe_gather(fp, nbytes, lanes, lpl, 5, None, ctypes.byref(ms))
            assert rc == 0, rc
            print(f"{nbytes:5d} {fp >> 10:6d}KB {lpl:9d} {lanes:9d} {ms.value:7.4f} {TOTAL / ms.value / 1e6:9.1f}")
