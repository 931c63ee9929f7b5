"""bliss_gnn_amd -- MI355X (gfx950) implementation of the BLISS-GNN hot path.

Layer-wise bandit / LADIES block sampling, per-block weighted SpMM message passing and the EXP3
reward update, as hand-written HIP kernels behind a C ABI (include/bliss_gnn.h), surfaced through
classes that keep the reference's sampler / block / model interface.  Importing this package loads
libbliss_gnn.so and raises if it is missing -- there is no CPU or PyTorch fallback.
"""
from . import _lib  # noqa: F401  (fail loudly when the HIP library is absent)
from .graph import Block, Graph, NID, EID  # noqa: F401
from .bandit_sampler import BanditLadiesSampler, PoissonBanditLadiesSampler, normalized_edata  # noqa: F401
from .ladies_sampler import LadiesSampler, PoissonLadiesSampler  # noqa: F401
