"""Which receiver detects which transmit streams - mirror of
``sionna.phy.mimo.StreamManagement`` (reference src/sionna/phy/mimo/stream_management.py:9-246).
Pure host-side index bookkeeping (NumPy), vectorised."""
import numpy as np

from ..block import Object


class StreamManagement(Object):
    def __init__(self, rx_tx_association, num_streams_per_tx):
        super().__init__()
        self._num_streams_per_tx = int(num_streams_per_tx)
        self.rx_tx_association = rx_tx_association

    rx_tx_association = property(lambda self: self._rx_tx_association)
    num_rx = property(lambda self: self._num_rx)
    num_tx = property(lambda self: self._num_tx)
    num_streams_per_tx = property(lambda self: self._num_streams_per_tx)
    num_tx_per_rx = property(lambda self: self._num_tx_per_rx)
    num_rx_per_tx = property(lambda self: self._num_rx_per_tx)
    precoding_ind = property(lambda self: self._precoding_ind)
    stream_association = property(lambda self: self._stream_association)
    detection_desired_ind = property(lambda self: self._detection_desired_ind)
    detection_undesired_ind = property(lambda self: self._detection_undesired_ind)
    tx_stream_ids = property(lambda self: self._tx_stream_ids)
    rx_stream_ids = property(lambda self: self._rx_stream_ids)
    stream_ind = property(lambda self: self._stream_ind)

    @property
    def num_streams_per_rx(self):
        return int(self.num_tx * self.num_streams_per_tx / self.num_rx)

    @property
    def num_interfering_streams_per_rx(self):
        return int(self.num_tx * self.num_streams_per_tx - self.num_streams_per_rx)

    @rx_tx_association.setter
    def rx_tx_association(self, rx_tx_association):
        a = np.array(rx_tx_association, np.int32)
        assert np.isin(a, (0, 1)).all(), "All elements of `stream_association` must be 0 or 1"
        self._num_rx, self._num_tx = a.shape
        per_rx, per_tx = a.sum(1), a.sum(0)
        assert per_rx.min() == per_rx.max(), \
            "Each receiver needs to be associated with the same number of transmitters."
        assert per_tx.min() == per_tx.max(), \
            "Each transmitter needs to be associated with the same number of receivers."
        self._num_tx_per_rx, self._num_rx_per_tx = int(per_rx[0]), int(per_tx[0])
        self._rx_tx_association = a
        # receivers served by each transmitter
        self._precoding_ind = np.stack([np.flatnonzero(a[:, j]) for j in range(self._num_tx)]).astype(np.int32)
        # stream_association[i, j, k] = 1: receiver i gets stream k of transmitter j.  The
        # receivers of a transmitter take consecutive blocks of num_streams_per_rx streams.
        ns, nsr = self._num_streams_per_tx, self.num_streams_per_rx
        sa = np.zeros((self._num_rx, self._num_tx, ns), np.int32)
        order = np.cumsum(a, axis=0) - 1                     # rank of receiver i among tx j's receivers
        for i, j in zip(*np.nonzero(a)):
            lo = order[i, j] * nsr
            sa[i, j, lo:lo + min(nsr, ns)] = 1               # slice clips at num_streams_per_tx
        self._stream_association = sa
        flat = sa.reshape(-1)
        self._detection_desired_ind = np.flatnonzero(flat == 1)
        self._detection_undesired_ind = np.flatnonzero(flat == 0)
        self._tx_stream_ids = np.arange(self._num_tx * ns).reshape(self._num_tx, ns)
        # global ids (tx * num_streams_per_tx + stream) of the streams of each receiver, ascending
        self._rx_stream_ids = np.stack([np.flatnonzero(sa[i].reshape(-1)) for i in range(self._num_rx)]).astype(np.int32)
        self._stream_ind = np.argsort(self._rx_stream_ids.reshape(-1))
