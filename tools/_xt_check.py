import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gpzoo_amd import _lib
lib = _lib.load()
L, M = 1, 256
g = torch.Generator().manual_seed(0)
B = torch.randn(L, M, M, generator=g, dtype=torch.float64)
A = (B @ B.transpose(-1, -2) / M + torch.eye(M, dtype=torch.float64)).cuda()
nblk = M // 128
lib.gpz_debug_factor_sync_words.restype = C.c_size_t
lib.gpz_debug_factor_sync_words.argtypes = [C.c_int64, C.c_int64]
lib.gpz_debug_factor_invert.restype = C.c_int
lib.gpz_debug_factor_invert.argtypes = [C.c_void_p, C.c_int64, C.c_int64] + [C.c_void_p] * 6
Dinv = torch.empty(L * nblk * 128 * 128, dtype=torch.float64, device="cuda")
Linv = torch.full((L, M, M), float("nan"), dtype=torch.float64, device="cuda")
T = torch.full((L, M, M), float("nan"), dtype=torch.float64, device="cuda")
sync = torch.empty(lib.gpz_debug_factor_sync_words(M, L), dtype=torch.int32, device="cuda")
info = torch.empty(L, dtype=torch.int32, device="cuda")
W = A.clone()
rc = lib.gpz_debug_factor_invert(W.data_ptr(), M, L, Dinv.data_ptr(), Linv.data_ptr(), T.data_ptr(), sync.data_ptr(), info.data_ptr(), None)
torch.cuda.synchronize()
print("rc", rc, "info", info.tolist())
Lc = torch.tril(W[0])
ref = torch.linalg.inv(Lc)
for b in range(nblk):
    blk = T[0, 128*b:128*b+128, 128*b:128*b+128]
    print("XT diag block", b, "strict lower max", float(torch.tril(blk, -1).abs().max()), "nan count", int(torch.isnan(blk).sum()),
          "vs ref^T", float((blk - ref[128*b:128*b+128, 128*b:128*b+128].T).abs().max()))
print("Linv err", float((torch.tril(Linv[0]) - ref).abs().max()))
print("XT(0,1) vs Linv(1,0)^T", float((T[0, :128, 128:] - ref[128:, :128].T).abs().max()))
d = (torch.tril(Linv[0]) - ref)[128:, :128].abs()
print("rows with error", (d.max(1).values > 1e-9).nonzero().flatten().tolist()[:20], "cols", (d.max(0).values > 1e-9).nonzero().flatten().tolist()[:40])
