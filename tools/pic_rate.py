import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import sionna_amd.phy as phy
from sionna_amd import _ffi
_ffi.device()
phy.config.seed = 11
for (B, K, M, nb) in ((2048, 4, 4, 4), (2048, 2, 4, 2), (1024, 4, 8, 6)):
    nsym = 300
    h = phy.utils.complex_normal([B, nsym, M, K], 1.0).as_subclass(torch.Tensor)
    x = phy.utils.complex_normal([B, nsym, K], 1.0).as_subclass(torch.Tensor)
    no = 0.2
    y = (h @ x.unsqueeze(-1)).squeeze(-1) + phy.utils.complex_normal([B, nsym, M], no).as_subclass(torch.Tensor)
    s = (no * torch.eye(M, dtype=torch.complex64, device=y.device)).expand(B, nsym, M, M).contiguous()
    prior = torch.randn((B, nsym, K, nb), dtype=torch.float32, device=y.device)
    lin = phy.mimo.LinearDetector("lmmse", "bit", "maxlog", constellation_type="qam", num_bits_per_symbol=nb)
    for name, f in (("MMSEPICDetector(maxlog, 1 iteration)", lambda: phy.mimo.MMSEPICDetector("bit", "maxlog", num_iter=1, constellation_type="qam", num_bits_per_symbol=nb)),
                    ("MMSEPICDetector(app, 1 iteration)", lambda: phy.mimo.MMSEPICDetector("bit", "app", num_iter=1, constellation_type="qam", num_bits_per_symbol=nb))):
        pic = f()
        pic(y, h, s, prior); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): pic(y, h, s, prior)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / 5
        print(f"{M}x{K} {1<<nb}-QAM {B*nsym} REs: {name} {t*1e3:.3f} ms = {B*nsym/t/1e6:.1f} M RE/s", flush=True)
    lin(y, h, s); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): lin(y, h, s)
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 5
    print(f"{M}x{K} {1<<nb}-QAM {B*nsym} REs: LinearDetector(lmmse, maxlog) {t*1e3:.3f} ms = {B*nsym/t/1e6:.1f} M RE/s", flush=True)
