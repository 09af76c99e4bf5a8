"""sionna_amd - MI355X-native (gfx950) implementation of Sionna PHY's link-level
Monte-Carlo hot path: hand-written HIP kernels behind a C-ABI (``libsionna_amd.so``,
``include/sionna_amd.h``) and a host-side mirror of the ``sionna.phy`` Block API
(``sionna_amd.phy``).  See DESIGN.md / INTEGRATION.md.
"""
__version__ = "0.1.0"


def install_as_sionna(tf_shim=False):
    """Alias this package as ``sionna`` so that ``from sionna.phy... import ...`` in an
    existing BER notebook resolves to the MI355X implementation.

    ``tf_shim=True`` additionally registers ``sionna_amd.tf_shim`` as ``tensorflow`` IF the real package cannot be
    imported, for notebooks that decorate their model with ``@tf.function`` and pass ``tf.constant`` scalars."""
    import sys
    import importlib
    phy = importlib.import_module("sionna_amd.phy")
    sys.modules.setdefault("sionna", sys.modules[__name__])
    for name, mod in list(sys.modules.items()):
        if name.startswith("sionna_amd.phy"):
            sys.modules.setdefault("sionna" + name[len("sionna_amd"):], mod)
    if tf_shim and "tensorflow" not in sys.modules:
        try:
            importlib.import_module("tensorflow")
        except ImportError:
            sys.modules["tensorflow"] = importlib.import_module("sionna_amd.tf_shim")
    return phy
