#!/usr/bin/env python3
"""Diagnostic for the IDD notebook curves (Introduction_to_Iterative_Detection_and_Decoding.ipynb, perfect-CSI Rayleigh 16 x 4):
which stage of detection -> soft decoding (state) -> MMSE-PIC with priors -> final decoding carries the gain."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import notebook_curves as nc


def main():
    import sionna_amd.phy as phy
    from sionna_amd import _ffi
    ref = nc.load_tables()
    for e in (-8.0, -7.0):
        i = int(round(e + 10))
        print(f"== {e} dB   reference BLER: LMMSE {ref['Introduction_to_Iterative_Detection_and_Decoding/c15/t0']['rows'][i]['bler']:.4f}  "
              f"IDD2 {ref['Introduction_to_Iterative_Detection_and_Decoding/c15/t3']['rows'][i]['bler']:.4f}  "
              f"IDD3 {ref['Introduction_to_Iterative_Detection_and_Decoding/c15/t4']['rows'][i]['bler']:.4f}")
        phy.config.seed = 3
        m = nc._IddRayleigh("idd2")
        B = 2048
        no = torch.full([B], float(phy.utils.ebnodb2no(e, num_bits_per_symbol=4, coderate=0.5)), dtype=torch.float32, device=_ffi.device())
        b = m.source([B, 4, 1, m.K])
        c = m.encoder(b)
        y, h = m.channel(m.rg_mapper(m.mapper(c)), no.reshape(-1, 1, 1, 1, 1))
        hh = m.remove_nulled(h)
        ev = torch.zeros(tuple(hh.shape), dtype=torch.float32, device=hh.device)
        bler = lambda bh: float((bh != b).reshape(-1, m.K).any(-1).float().mean())
        ld = phy.fec.ldpc
        llr0 = m.detector(y, hh, ev, no)
        mk = lambda it, **kw: ld.LDPC5GDecoder(m.encoder, num_iter=it, cn_update="minsum", **kw)
        print("  one-shot LMMSE + 12 it:", bler(mk(12)(llr0)), " + 24 it:", bler(mk(24)(llr0)))
        llr_dec, msg = m.siso_decoder(llr0, msg_v2c=None)
        cw_ber = float(((llr_dec > 0).float() != c).float().mean())
        print("  after soft decoder (12 it): codeword BER", cw_ber, " |llr_dec| mean", float(llr_dec.abs().mean()))
        llr1 = m.siso_detector(y, hh, llr_dec, ev, no)
        print("  MMSE-PIC out: mean |llr|", float(llr1.abs().mean()), " LMMSE out mean |llr|", float(llr0.abs().mean()),
              " raw BER pic", float(((llr1 > 0).float() != c).float().mean()), " raw BER lmmse", float(((llr0 > 0).float() != c).float().mean()))
        bh, _ = m.decoder(llr1, msg_v2c=msg)
        print("  IDD2 as in the notebook (final decoder continues from the state):", bler(bh))
        print("  IDD2, final decoder restarted (no state):", bler(mk(12)(llr1)))
        llr1z = m.siso_detector(y, hh, torch.zeros_like(llr_dec), ev, no)
        bhz, _ = m.decoder(llr1z, msg_v2c=msg)
        print("  MMSE-PIC with zero prior, state kept:", bler(bhz), " max |pic0 - lmmse|", float((llr1z - llr0).abs().max()))
        ext = llr_dec - torch.clamp(llr0, -20., 20.)
        llr1e = m.siso_detector(y, hh, ext, ev, no)
        bhe, _ = m.decoder(llr1e, msg_v2c=msg)
        print("  prior = EXTRINSIC decoder LLRs, state kept:", bler(bhe))
        print("  decoder input = MMSE-PIC out + prior (a-posteriori), state kept:", bler(m.decoder(llr1 + llr_dec, msg_v2c=msg)[0]))


if __name__ == "__main__":
    main()
