"""debug: fgx_record_boundaries_device on a large simulated stream (FGX_BOUND_DEBUG=1 prints the first repair rounds)."""
import ctypes as C, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from fgumi_amd import VanillaUmiConsensusCaller, VanillaUmiConsensusOptions, lib
fam = int(sys.argv[1]) if len(sys.argv) > 1 else 150000
c = VanillaUmiConsensusCaller("", "A", VanillaUmiConsensusOptions(min_reads=1), overlapping_consensus=True)
dg = c.simulate_on_device(fam, family_size=8)
cut = int(sys.argv[2]) if len(sys.argv) > 2 else 150
n, used = C.c_uint64(), C.c_uint64()
off = torch.empty(dg.n_rec + 8, dtype=torch.int64, device=dg.blob.device)
ln = torch.empty(dg.n_rec + 8, dtype=torch.int32, device=dg.blob.device)
torch.cuda.synchronize()
for rep in range(2):
    t = time.perf_counter()
    rc = lib.fgx_record_boundaries_device(c._h, dg.blob.data_ptr(), dg.blob_len - cut, 0, off.data_ptr(), ln.data_ptr(), dg.n_rec + 8, C.byref(n), C.byref(used))
    dt = time.perf_counter() - t
    print("rc", rc, "n_rec", n.value, "of", dg.n_rec, "consumed", used.value, "len", dg.blob_len - cut, "ms %.2f" % (dt * 1e3))
ok = bool((off[:n.value] == dg.rec_off[:n.value]).all()) and bool((ln[:n.value] == dg.rec_len[:n.value]).all())
print("equal to the simulator's table:", ok)
