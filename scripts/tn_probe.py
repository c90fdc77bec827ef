"""One wgrad-shaped launch (C = A^T B over P rows) through the test entry point, for
rocprofv3 --pmc runs: python scripts/tn_probe.py [P] [Mo] [Ni] [reps]."""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from pointnet_refine_amd import _lib as L

P = int(sys.argv[1]) if len(sys.argv) > 1 else 4096 * 1024
MO = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
NI = int(sys.argv[3]) if len(sys.argv) > 3 else 1984
REPS = int(sys.argv[4]) if len(sys.argv) > 4 else 2
lib = L.lib()
a = torch.randn(P, MO, device="cuda")
b = torch.randn(P, NI, device="cuda")
c = torch.empty(MO, NI, device="cuda")
nb = lib.prh_test_gemm_tn_workspace_bytes(P, MO, NI)
ws = torch.empty(nb, dtype=torch.uint8, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
p = lambda t: C.c_void_p(t.data_ptr())
ev = [torch.cuda.Event(enable_timing=True) for _ in range(REPS + 1)]
ev[0].record()
for i in range(REPS):
    rc = lib.prh_test_gemm_tn(p(a), p(b), p(c), None, P, MO, NI, p(ws), nb, 0, st)
    assert rc == 0, lib.prh_last_error()
    ev[i + 1].record()
torch.cuda.synchronize()
print("ms per call (incl. absmax passes):", [round(ev[i].elapsed_time(ev[i + 1]), 2) for i in range(REPS)])
ref = (a[:65536].double().t() @ b[:65536].double())
print("finite:", bool(torch.isfinite(c).all()), "sample ref norm", float(ref.norm()))

# where the dispatcher puts the workgroups of such a launch (the kernels assume XCD = id % 8)
blocks, lds = 768, 73728
m = torch.zeros(blocks, 2, dtype=torch.int32, device="cuda")
rc = lib.prh_test_xcc_map(blocks, lds, p(m), 0, st)
assert rc == 0, lib.prh_last_error()
torch.cuda.synchronize()
x = (m[:, 0] & 15).cpu()
ids = torch.arange(blocks)
print("xcc of workgroups 0..31:", x[:32].tolist())
print("share with xcc == (id + c) % 8 for the best c:", max(float(((ids + c) % 8 == x).float().mean()) for c in range(8)))
print("workgroups per xcc:", torch.bincount(x, minlength=8).tolist())
print("first 256 workgroups per xcc:", torch.bincount(x[:256], minlength=8).tolist())
