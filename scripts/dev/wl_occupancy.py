"""What the HIP runtime says about the LDS-staged warp kernel's residency, and a census: how many of its workgroups does a CU
really hold at once?  (rocprofv3 counters gave SQ_WAVE_CYCLES / (GRBM cycles x 256 CUs) = one workgroup per CU fewer than the
register / LDS budgets allow.)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from wild_deep_mvs_amd import _lib as L
lib = C.CDLL(L.LIB_PATH)
b, l, t = C.c_int(0), C.c_int(0), C.c_int(0)
torch.cuda.init(); torch.zeros(1, device="cuda")
rc = lib.pscv_debug_wl_occupancy(C.byref(b), C.byref(l), C.byref(t))
print(f"hipOccupancyMaxActiveBlocksPerMultiprocessor rc={rc}: {b.value} blocks/CU of {t.value} threads with {l.value} B dynamic LDS")
p = torch.cuda.get_device_properties(0)
print("CUs", p.multi_processor_count, "LDS per block max", getattr(p, "shared_memory_per_block", None), "per CU", getattr(p, "shared_memory_per_multiprocessor", None))
