"""bliss_gnn_amd -- MI355X (gfx950) implementation of the BLISS-GNN hot path.

Layer-wise bandit / LADIES block sampling, per-block weighted SpMM message passing and the EXP3
reward update, as hand-written HIP kernels behind a C ABI (include/bliss_gnn.h), surfaced through
classes that keep the reference's sampler / block / model interface.  Importing this package loads
libbliss_gnn.so and raises if it is missing -- there is no CPU or PyTorch fallback.
"""
import os as _os

# The pipelined train loop keeps four streams busy at once (critical chain, backward pass, early-layer blocks, random-number
# generator), RCCL adds its own, and HIP multiplexes streams onto 4 hardware queues by default: two of those streams on one
# queue serialise (measured: 0.85 -> 1.40 ms/step when the generator shared the backward pass's queue).  Ask for 8 queues;
# read by the HIP runtime when it initialises, so this only helps if the package is imported before the first GPU call
# (bench.py and tests/conftest.py also set it first thing).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import _lib  # noqa: F401,E402  (fail loudly when the HIP library is absent)
from .graph import Block, Graph, NID, EID  # noqa: F401,E402
from .bandit_sampler import BanditLadiesSampler, PoissonBanditLadiesSampler, normalized_edata  # noqa: F401,E402
from .ladies_sampler import LadiesSampler, PoissonLadiesSampler  # noqa: F401,E402
