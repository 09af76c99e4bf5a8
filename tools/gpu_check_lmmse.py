#!/usr/bin/env python3
"""Runs the library-algebra rows (LMMSEInterpolator, EPDetector step methods) on the MI355X against the reference-executed
fixtures - the device twin of tests/test_lmmse_interpolator.py / tests/test_ep_steps_ref_exec.py (which use host tensors)."""
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sionna_amd import _ffi  # noqa: E402
from sionna_amd.phy.ofdm import LMMSEInterpolator  # noqa: E402
from sionna_amd.phy.mimo import EPDetector  # noqa: E402

dev = _ffi.device()
G = np.load(os.path.join(ROOT, "tests", "golden", "lmmse_interp_ref_golden.npz"))
worst = 0.0
for gi in (0, 1):
    g = {k.split("/", 1)[1]: G[k] for k in G.files if k.startswith(f"g{gi}/")}
    pp = types.SimpleNamespace(mask=g["mask"], pilots=g["pilots"])
    for order in [str(o) for o in G["orders"]]:
        h, e = LMMSEInterpolator(pp, g["cov_time"], g["cov_freq"], g["cov_space"], order=order)(g["h"], g["err_var"])
        assert h.is_cuda and h.dtype == torch.complex64 and e.dtype == torch.float32
        dh = np.abs(h.cpu().numpy() - g[f"h_{order}"]).max() / np.abs(g[f"h_{order}"]).max()
        de = np.abs(e.cpu().numpy() - g[f"e_{order}"]).max() / max(np.abs(g[f"e_{order}"]).max(), 1.0)
        worst = max(worst, dh, de)
        assert dh <= 2e-4 and de <= 2e-4, (gi, order, dh, de)
print(f"LMMSEInterpolator on {torch.cuda.get_device_name(0)}: 10 cases within 2e-4 of the reference-executed values (worst {worst:.2e})")
E = np.load(os.path.join(ROOT, "tests", "golden", "ep_steps_ref_golden.npz"))
for nb in (2, 4, 6):
    g = {k.split("/", 1)[1]: torch.from_numpy(E[k]).float().to(dev) for k in E.files if k.startswith(f"nb{nb}/")}
    det = EPDetector("bit", nb, l=2, beta=0.7)
    sigma, mu = det.compute_sigma_mu(g["hth"], g["hty"], g["no"], g["lam_init"], g["gam_init"])
    v_obs, x_obs = det.compute_v_x_obs(sigma, mu, g["lam_init"], g["gam_init"])
    v, x, logits = det.compute_v_x(v_obs, x_obs)
    lam, gam = det.update_lam_gam(v, v_obs, x, x_obs, g["lam_init"], g["gam_init"])
    for a, b in ((sigma, "sigma0"), (mu, "mu0"), (v, "v0"), (x, "x0"), (logits, "logits0"), (lam, "lam0"), (gam, "gam0")):
        assert float((a - g[b]).abs().max()) <= 1e-3 * max(float(g[b].abs().max()), 1.0), (nb, b)
print("EPDetector step methods on the device: agree with the reference-executed values")
print("GPU_CHECK_OK")
