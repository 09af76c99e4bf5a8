"""Utilities of the hot path (mirror of ``sionna.phy.utils``)."""
from .metrics import count_errors, count_block_errors, compute_ber, compute_bler
from .misc import (ebnodb2no, hard_decisions, complex_normal, sim_ber, get_throughput, spawn_sim_ber,
                   init_distributed)
from .tensors import expand_to_rank, insert_dims, flatten_dims, flatten_last_dims, split_dim, log2, log10, db
from .plotting import plot_ber, PlotBER
from .linalg import inv_cholesky, matrix_pinv
