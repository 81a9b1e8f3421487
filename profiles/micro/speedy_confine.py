"""Does the SPEEDY window (104 small dependent kernels) run faster on a CU-masked stream, where every kernel finds what the previous
one wrote in the same L2s?  (Confining the LU's panel chain that way took a panel update beside the trailing GEMM from 29 to 9 us.)
Times HybridRank.speedy_leg alone on the default stream and on masks of the first N compute units."""
import ctypes as C, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package; load_package()
from speedy_ml_amd import hybrid, synth
from speedy_ml_amd._lib import check, lib
sea = synth.land_mask(); classes = hybrid.region_classes(sea)
m = hybrid.HybridRank(list(range(1152)), classes, sea_mask=sea, mode="hybrid", n_override=1)
ncu = torch.cuda.get_device_properties(0).multi_processor_count
words = (ncu + 31) // 32
def masked(n):
    mk = (C.c_uint32 * words)()
    for cu in range(n): mk[cu // 32] |= 1 << (cu % 32)
    h = C.c_void_p(); check(lib().sml_stream_create_cu_mask(mk, words, C.byref(h)))
    return torch.cuda.ExternalStream(h.value), h
def run(stream, reps=30):
    with torch.cuda.stream(stream):
        for _ in range(3): m.speedy_leg(stream)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): m.speedy_leg(stream)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3
print("all %d CUs (default stream): %.3f ms per window" % (ncu, run(torch.cuda.current_stream())))
keep = []
for n in (32, 64, 96, 128, 192):
    s, h = masked(n); keep.append((s, h))
    print("first %3d CUs: %.3f ms per window" % (n, run(s)))
