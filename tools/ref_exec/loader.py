"""Import reference modules UNMODIFIED from /root/reference under stand-ins for ``tensorflow`` (tools/ref_exec/tf_numpy.py)
and for the ``sionna.phy`` package scaffolding (Block/Object/config/dtypes), so that the reference's own algorithms run
here.  Only the files named in ``load()`` calls are executed; the package ``__init__`` files of the reference (which import
everything, including Keras-dependent code) are replaced by empty namespace packages.

    from tools.ref_exec.loader import reference
    ref = reference()                       # installs the stand-ins into sys.modules (idempotent)
    dec = ref.load("sionna.phy.fec.ldpc.decoding")
    dec.cn_update_minsum(...)
"""
import contextlib
import importlib.util
import os
import pathlib
import sys
import types

import numpy as np

from . import tf_numpy

REF_SRC = "/root/reference/src"


class _Block:
    """Stand-in for sionna.phy.Block / Object (reference src/sionna/phy/block.py:13-155): precision bookkeeping and the
    build-once-then-call protocol; inputs are converted to float32/complex64 tensors of the stand-in."""

    def __init__(self, *args, precision=None, **kwargs):
        self._precision = precision or "single"
        self._built = False

    precision = property(lambda self: self._precision)
    rdtype = property(lambda self: tf_numpy.DType("float32" if self._precision == "single" else "float64"))
    cdtype = property(lambda self: tf_numpy.DType("complex64" if self._precision == "single" else "complex128"))
    built = property(lambda self: self._built)

    def _conv(self, v):
        if isinstance(v, tf_numpy.RaggedTensor) or v is None or isinstance(v, (str, bool)):
            return v
        a = np.array(v)                                            # copy: the reference mutates its inputs in place
        if a.dtype.kind == "f":
            a = a.astype(self.rdtype)
        elif a.dtype.kind == "c":
            a = a.astype(self.cdtype)
        return a.view(tf_numpy.Tensor)

    def _cast_or_check_precision(self, v):
        """block.py:54-81: tensors are cast to the object's real / complex dtype (variables would be checked)."""
        a = np.asarray(v)
        return a.astype(self.cdtype if a.dtype.kind == "c" else self.rdtype).view(tf_numpy.Tensor)

    def build(self, *a, **k):
        pass

    def __call__(self, *args, **kwargs):
        args = [self._conv(a) for a in args]
        kwargs = {k: self._conv(v) for k, v in kwargs.items()}
        if not self._built:
            shapes = [tuple(a.shape) if hasattr(a, "shape") else () for a in args]
            self.build(*shapes, **{k: (tuple(v.shape) if hasattr(v, "shape") else ()) for k, v in kwargs.items()})
            self._built = True
        return self.call(*args, **kwargs)


class Reference:
    def __init__(self):
        self.tf = tf_numpy.make_tf()
        self._installed = {}

    def install(self):
        if "tensorflow" in sys.modules and getattr(sys.modules["tensorflow"], "__doc__", "") != self.tf.__doc__:
            raise RuntimeError("a real tensorflow is imported; the stand-in must not shadow it")
        sys.modules["tensorflow"] = self.tf
        ir = types.ModuleType("importlib_resources")
        ir.files = lambda pkg: pathlib.Path(list(pkg.__path__)[0])
        ir.as_file = lambda p: contextlib.nullcontext(p)
        sys.modules["importlib_resources"] = ir
        for name in ("sionna", "sionna.phy", "sionna.phy.fec", "sionna.phy.fec.ldpc", "sionna.phy.fec.polar", "sionna.phy.mimo",
                     "sionna.phy.utils", "sionna.phy.ofdm", "sionna.phy.channel", "sionna.phy.signal"):
            if name not in sys.modules:
                m = types.ModuleType(name)
                m.__path__ = []                                    # namespace stub: nothing is importable implicitly
                m.__package__ = name
                sys.modules[name] = m
                parent, _, child = name.rpartition(".")
                if parent:
                    setattr(sys.modules[parent], child, m)
        phy = sys.modules["sionna.phy"]
        phy.Block, phy.Object = _Block, _Block
        phy.PI = np.pi
        phy.SPEED_OF_LIGHT = 299792458.0
        D = tf_numpy.DType
        dt = {"single": {"tf": {"rdtype": D("float32"), "cdtype": D("complex64")},
                         "np": {"rdtype": np.float32, "cdtype": np.complex64}},
              "double": {"tf": {"rdtype": D("float64"), "cdtype": D("complex128")},
                         "np": {"rdtype": np.float64, "cdtype": np.complex128}}}
        phy.dtypes = dt
        phy.config = types.SimpleNamespace(precision="single", tf_rdtype=D("float32"), tf_cdtype=D("complex64"),
                                           np_rdtype=np.float32, np_cdtype=np.complex64,
                                           tf_rng=self.tf.random.Generator(12345), np_rng=np.random.default_rng(12345))
        blk = types.ModuleType("sionna.phy.block")
        blk.Block, blk.Object = _Block, _Block
        sys.modules["sionna.phy.block"] = blk
        tnp = types.ModuleType("tensorflow.experimental.numpy")
        tnp.log10 = lambda x: tf_numpy._t(np.log10(np.asarray(x)))
        tnp.log2 = lambda x: tf_numpy._t(np.log2(np.asarray(x)))
        tnp.swapaxes = self.tf.experimental.numpy.swapaxes
        tnp.sinc = self.tf.experimental.numpy.sinc
        sys.modules["tensorflow.experimental"] = types.ModuleType("tensorflow.experimental")
        sig = types.ModuleType("tensorflow.signal")
        for _n in ("fft", "ifft", "fftshift", "ifftshift"):
            setattr(sig, _n, getattr(self.tf.signal, _n))
        sys.modules["tensorflow.signal"] = sig
        sys.modules["tensorflow.experimental.numpy"] = tnp
        cfg = types.ModuleType("sionna.phy.config")
        cfg.config, cfg.dtypes = phy.config, dt
        sys.modules["sionna.phy.config"] = cfg
        return self

    def load(self, modname, package_dir=False):
        """Execute the reference file for ``modname`` (e.g. 'sionna.phy.fec.ldpc.decoding') and register it."""
        if modname in self._installed:
            return self._installed[modname]
        rel = modname.replace(".", os.sep)
        path = os.path.join(REF_SRC, rel, "__init__.py") if package_dir else os.path.join(REF_SRC, rel + ".py")
        spec = importlib.util.spec_from_file_location(
            modname, path, submodule_search_locations=[os.path.dirname(path)] if package_dir else None)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[modname] = mod
        parent, _, child = modname.rpartition(".")
        if parent in sys.modules:
            setattr(sys.modules[parent], child, mod)
        spec.loader.exec_module(mod)
        self._installed[modname] = mod
        return mod


    def load_utils(self):
        """The reference's ``sionna.phy.utils`` star-imports its sub-modules (utils/__init__.py:5-10); load the ones the hot
        path uses and merge their public names into the stub package the same way."""
        pkg = sys.modules["sionna.phy.utils"]
        for sub in ("tensors", "metrics", "linalg", "misc"):
            m = self.load(f"sionna.phy.utils.{sub}")
            for k, v in vars(m).items():
                if not k.startswith("_"):
                    setattr(pkg, k, v)
        return pkg


    def load_signal(self):
        """sionna.phy.signal: only utils.py (fft / ifft / convolve), merged like the package's __init__ does"""
        pkg = sys.modules["sionna.phy.signal"]
        m = self.load("sionna.phy.signal.utils")
        for k, v in vars(m).items():
            if not k.startswith("_"):
                setattr(pkg, k, v)
        return pkg


_REF = None


def reference():
    global _REF
    if _REF is None:
        if not os.path.isdir(REF_SRC):
            raise RuntimeError(f"{REF_SRC} is not present: fixtures are generated in the build container only")
        _REF = Reference().install()
    return _REF
