"""``plot_ber`` / ``PlotBER`` - mirror of ``sionna.phy.utils.plotting`` (reference src/sionna/phy/utils/plotting.py:11-520):
the container BER notebooks use to run ``sim_ber`` per curve and draw the results.  Host-side glue only; matplotlib
is imported when a figure is actually drawn."""
from itertools import compress

import numpy as np

from .misc import sim_ber


def _as_np(x):
    return x.detach().cpu().numpy() if hasattr(x, "detach") else np.asarray(x)


def plot_ber(snr_db, ber, legend="", ylabel="BER", title="Bit Error Rate", ebno=True, is_bler=None, xlim=None,
             ylim=None, save_fig=False, path=""):
    """Semilog plot of one curve or of a list of curves (BLER curves dashed); returns ``(fig, ax)``
    (plotting.py:11-135)."""
    import matplotlib.pyplot as plt
    assert isinstance(legend, (str, list)), "legend must be str or list of str."
    assert isinstance(title, str), "title must be str."
    many = isinstance(ber, list)
    if many:
        assert all(isinstance(l, str) for l in legend), "legend must be str or list of str."
        if not isinstance(snr_db, list):
            snr_db = [snr_db] * len(ber)
    if is_bler is None:
        is_bler = [False] * len(ber) if many else False
    elif isinstance(is_bler, list):
        assert len(is_bler) == len(ber), "is_bler has invalid size."
    else:
        assert isinstance(is_bler, bool), "is_bler must be bool or list of bool."
        is_bler = [is_bler]
    fig, ax = plt.subplots(figsize=(16, 10))
    plt.xticks(fontsize=18)
    plt.yticks(fontsize=18)
    if xlim is not None:
        plt.xlim(xlim)
    if ylim is not None:
        plt.ylim(ylim)
    plt.title(title, fontsize=25)
    curves = zip(snr_db, ber, is_bler) if many else [(snr_db, ber, is_bler[0] if isinstance(is_bler, list) else is_bler)]
    for s, b, dashed in curves:
        plt.semilogy(_as_np(s), _as_np(b), "--" if dashed else "", linewidth=2)
    plt.grid(which="both")
    plt.xlabel(r"$E_b/N_0$ (dB)" if ebno else r"$E_s/N_0$ (dB)", fontsize=25)
    plt.ylabel(ylabel, fontsize=25)
    plt.legend(legend, fontsize=20)
    if save_fig:
        plt.savefig(path)
        plt.close(fig)
    return fig, ax


class PlotBER:
    """``PlotBER(title)``: stores (snr, ber, legend, is_bler) curves; ``simulate(...)`` runs ``sim_ber`` and appends
    its result, calling the object draws everything stored (plotting.py:138-520)."""

    def __init__(self, title="Bit/Block Error Rate"):
        assert isinstance(title, str), "title must be str."
        self._title = title
        self.reset()

    # pylint: disable=dangerous-default-value
    def __call__(self, snr_db=[], ber=[], legend=[], is_bler=[], show_ber=True, show_bler=True, xlim=None, ylim=None,
                 save_fig=False, path=""):
        assert isinstance(path, str), "path must be str"
        assert isinstance(save_fig, bool), "save_fig must be bool"
        if isinstance(ber, list) and not isinstance(snr_db, list):
            snr_db = [snr_db] * len(ber)
        lst = lambda v: v if isinstance(v, list) else [v]
        snrs, bers = self._snrs + lst(snr_db), self._bers + lst(ber)
        legends, flags = self._legends + lst(legend), self._is_bler + lst(is_bler)
        if flags:
            keep = [(f and show_bler) or (not f and show_ber) for f in flags]
            snrs, bers, legends, flags = (list(compress(v, keep)) for v in (snrs, bers, legends, flags))
        ylabel = "BLER" if flags and all(flags) else ("BER" if not any(flags) else "BER / BLER")
        return plot_ber(snr_db=snrs, ber=bers, legend=legends, is_bler=flags, title=self._title, ylabel=ylabel,
                        xlim=xlim, ylim=ylim, save_fig=save_fig, path=path)

    @property
    def title(self):
        return self._title

    @title.setter
    def title(self, title):
        assert isinstance(title, str), "title must be string"
        self._title = title

    ber = property(lambda self: self._bers)
    snr = property(lambda self: self._snrs)
    legend = property(lambda self: self._legends)
    is_bler = property(lambda self: self._is_bler)

    def simulate(self, mc_fun, ebno_dbs, batch_size, max_mc_iter, legend="", add_ber=True, add_bler=False,
                 soft_estimates=False, num_target_bit_errors=None, num_target_block_errors=None, target_ber=None,
                 target_bler=None, early_stop=True, graph_mode=None, distribute=None, add_results=True,
                 forward_keyboard_interrupt=True, show_fig=True, verbose=True):
        """Runs ``sim_ber`` with the given stopping rules, stores the curve(s) and returns ``(ber, bler)``."""
        assert isinstance(legend, str), "legend must be str."
        for name, v in (("add_ber", add_ber), ("add_bler", add_bler), ("add_results", add_results),
                        ("show_fig", show_fig), ("verbose", verbose)):
            assert isinstance(v, bool), f"{name} must be bool."
        ber, bler = sim_ber(mc_fun, ebno_dbs, batch_size, soft_estimates=soft_estimates, max_mc_iter=max_mc_iter,
                            num_target_bit_errors=num_target_bit_errors,
                            num_target_block_errors=num_target_block_errors, target_ber=target_ber,
                            target_bler=target_bler, early_stop=early_stop, graph_mode=graph_mode,
                            distribute=distribute, verbose=verbose,
                            forward_keyboard_interrupt=forward_keyboard_interrupt)
        added = 0
        if add_ber:
            self.add(ebno_dbs, ber, False, legend)
            added += 1
        if add_bler:
            self.add(ebno_dbs, bler, True, legend + " (BLER)")
            added += 1
        if show_fig:
            self()
        if add_results is False:
            for _ in range(added):
                self.remove(-1)
        return ber, bler

    def add(self, ebno_db, ber, is_bler=False, legend=""):
        """Adds a static curve."""
        assert len(ebno_db) == len(ber), "ebno_db and ber must have same number of elements."
        assert isinstance(legend, str), "legend must be str."
        assert isinstance(is_bler, bool), "is_bler must be bool."
        self._bers.append(ber)
        self._snrs.append(ebno_db)
        self._legends.append(legend)
        self._is_bler.append(is_bler)

    def reset(self):
        """Removes all stored curves."""
        self._bers, self._snrs, self._legends, self._is_bler = [], [], [], []

    def remove(self, idx=-1):
        """Removes the curve with index ``idx`` (negative indices count from the end)."""
        assert isinstance(idx, int), "id must be int."
        for v in (self._bers, self._snrs, self._legends, self._is_bler):
            del v[idx]
