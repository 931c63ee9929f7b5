"""CPU suite: the MT19937 jump-ahead polynomials the parallel generator uses (csrc/mt_jump.hip, host code -- no GPU):
convolving torch's own word sequence with t^(n-1) mod phi must give torch's state n words later, word 0 included."""
import ctypes as C

import numpy as np
import pytest
import torch

from bliss_gnn_amd import _lib

N, M = 624, 397


def _next_block(od):
    """at::mt19937::next_state (ATen/core/MT19937RNGEngine.h), out of place."""
    nw = np.zeros(N, dtype=np.uint32)
    for k in range(N):
        u, v = int(od[k]), int(od[k + 1]) if k < N - 1 else int(nw[0])
        m = int(od[k + M]) if k < N - M else int(nw[k - (N - M)])
        y = (u & 0x80000000) | (v & 0x7FFFFFFF)
        nw[k] = m ^ (y >> 1) ^ (0x9908B0DF if v & 1 else 0)
    return nw


@pytest.mark.parametrize("blocks_ahead", [33, 41, 120])
def test_jump_polynomial_reproduces_torch_state(blocks_ahead):
    torch.manual_seed(2024)
    st = torch.get_rng_state().numpy()
    state = st[24:24 + N * 8].view(np.uint64).astype(np.uint32)
    blocks = [state]
    for _ in range(blocks_ahead):
        blocks.append(_next_block(blocks[-1]))
    # (sanity of the restatement above: torch's next outputs are the tempered words of the following block)
    x = np.concatenate(blocks)
    poly = np.zeros(N, dtype=np.uint32)
    _lib.check(_lib.lib.bliss_mt_jump_poly(blocks_ahead * N - 1, poly.ctypes.data), "bliss_mt_jump_poly")
    coef = np.nonzero(np.unpackbits(poly.view(np.uint8), bitorder="little")[:19937])[0]
    assert coef.size > 100                                       # (sparse this close to the degree; ~half the coefficients far out)
    new = np.array([np.bitwise_xor.reduce(x[1 + k + coef]) for k in range(N)], dtype=np.uint32)
    assert np.array_equal(new, blocks[blocks_ahead])


def test_restated_recurrence_is_torchs():
    torch.manual_seed(7)
    st = torch.get_rng_state().numpy()
    state = st[24:24 + N * 8].view(np.uint64).astype(np.uint32)
    left = int(st[8:12].view(np.int32)[0])
    assert left == 1                                             # freshly seeded: the first draw regenerates the block
    y = _next_block(state).astype(np.uint64)
    y ^= y >> 11
    y ^= (y << 7) & 0x9D2C5680
    y ^= (y << 15) & 0xEFC60000
    y ^= y >> 18
    want = ((y & 0xFFFFFF).astype(np.float32) * np.float32(1.0 / 16777216.0))
    assert np.array_equal(torch.rand(N).numpy(), want)
