"""Device twin of tests/test_lmmse_interpolator.py / tests/test_ep_steps_ref_exec.py: ``LMMSEInterpolator`` (through its
public call, inputs moved to the MI355X) and the ``EPDetector`` step methods on device tensors against the reference's
executed classes (tests/golden/lmmse_interp_ref_golden.npz, ep_steps_ref_golden.npz).  The same checks as
tools/gpu_check_lmmse.py, which passed on the MI355X at the end of round 4 (profiles/r04last_gpu_check_lmmse.txt)."""
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "lmmse_interp_ref_golden.npz"))
EP = np.load(os.path.join(os.path.dirname(__file__), "golden", "ep_steps_ref_golden.npz"))


@pytest.mark.parametrize("gi", [0, 1])
def test_lmmse_interpolator_on_device(gi):
    import torch
    from sionna_amd import _ffi
    from sionna_amd.phy.ofdm import LMMSEInterpolator
    _ffi.device()
    g = {k.split("/", 1)[1]: GOLD[k] for k in GOLD.files if k.startswith(f"g{gi}/")}
    pp = types.SimpleNamespace(mask=g["mask"], pilots=g["pilots"])
    for order in [str(o) for o in GOLD["orders"]]:
        h, e = LMMSEInterpolator(pp, g["cov_time"], g["cov_freq"], g["cov_space"], order=order)(g["h"], g["err_var"])
        assert h.is_cuda and h.dtype == torch.complex64 and e.dtype == torch.float32
        dh = np.abs(h.cpu().numpy() - g[f"h_{order}"]).max() / np.abs(g[f"h_{order}"]).max()
        de = np.abs(e.cpu().numpy() - g[f"e_{order}"]).max() / max(np.abs(g[f"e_{order}"]).max(), 1.0)
        assert dh <= 2e-4 and de <= 2e-4, (order, dh, de)          # (measured worst 1.5e-4)


@pytest.mark.parametrize("nb", [2, 4, 6])
def test_ep_step_methods_on_device(nb):
    import torch
    from sionna_amd import _ffi
    from sionna_amd.phy.mimo import EPDetector
    dev = _ffi.device()
    g = {k.split("/", 1)[1]: torch.from_numpy(EP[k]).float().to(dev) for k in EP.files if k.startswith(f"nb{nb}/")}
    det = EPDetector("bit", nb, l=2, beta=0.7)
    sigma, mu = det.compute_sigma_mu(g["hth"], g["hty"], g["no"], g["lam_init"], g["gam_init"])
    v_obs, x_obs = det.compute_v_x_obs(sigma, mu, g["lam_init"], g["gam_init"])
    v, x, logits = det.compute_v_x(v_obs, x_obs)
    lam, gam = det.update_lam_gam(v, v_obs, x, x_obs, g["lam_init"], g["gam_init"])
    for a, b in ((sigma, "sigma0"), (mu, "mu0"), (v, "v0"), (x, "x0"), (logits, "logits0"), (lam, "lam0"), (gam, "gam0")):
        assert float((a - g[b]).abs().max()) <= 1e-3 * max(float(g[b].abs().max()), 1.0), b
