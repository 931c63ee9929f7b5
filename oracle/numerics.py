"""Arithmetic contract shared by the oracle and (restated in HIP) by the kernels.

TEST INFRASTRUCTURE ONLY -- nothing under ``bliss_gnn_amd/`` may import this.

The reference keeps every value tensor in bfloat16 (``load_graph.py:7``,
``bandit_sampler.py:24,343``) and performs its reductions inside DGL
(``dgl.ops.copy_e_sum`` at ``bandit_sampler.py:67,73,129,150,151,316``).  DGL is
absent from this image, so the order in which DGL adds the terms of such a sum
cannot be observed ("parity unpinned").  The contract used
by BOTH sides of every parity test is therefore the order-independent ideal:

  * an element-wise op reads bf16, computes in fp32 (IEEE, correctly rounded
    ``+ - * / sqrt``) and rounds ONCE to bf16 with round-to-nearest-even.  That
    is exactly what torch's CPU kernels do for bf16 tensors, so the oracle
    simply calls torch for those;
  * a segment reduction adds its bf16 terms EXACTLY (as integers, in a fixed
    point format given per call site) and rounds the exact sum once to bf16.
    Integer addition commutes, so any schedule on any number of GPUs gives the
    same bits.

Fixed-point formats (value = integer * 2**-frac):
  FRAC_DST = 40   per-destination sums of weights / probabilities
  FRAC_SRC = 44   per-source sum of squared normalised probabilities
  FRAC_BLK = 36   per-destination sum of Hajek weights inside a block
A bf16 term whose least significant mantissa bit lies below 2**-frac is
truncated toward zero (never happens for in-degrees < ~2**18).
"""
import numpy as np
import torch

FRAC_DST = 40
FRAC_SRC = 44
FRAC_BLK = 36


def bf16_bits(x: torch.Tensor) -> torch.Tensor:
    """bf16 tensor -> int64 tensor holding the 16 raw bits (0..65535)."""
    assert x.dtype == torch.bfloat16
    return x.contiguous().view(torch.int16).to(torch.int64) & 0xFFFF


def bits_to_bf16(bits: torch.Tensor) -> torch.Tensor:
    b = (bits & 0xFFFF).to(torch.int64)
    b = torch.where(b >= 32768, b - 65536, b).to(torch.int16)
    return b.view(torch.bfloat16)


def bf16_to_fixed(x: torch.Tensor, frac) -> torch.Tensor:
    """Exact (truncating below 2**-frac) bf16 -> signed int64 fixed point (``frac``: int, or an int64 tensor per element).

    Non-finite inputs are not representable: they raise, and the HIP side sets
    its error flag for them (tests cover both)."""
    bits = bf16_bits(x)
    sign = bits >> 15
    exp = (bits >> 7) & 0xFF
    man = bits & 0x7F
    if bool((exp == 255).any()):
        raise FloatingPointError("non-finite bf16 term in an exact reduction")
    m = torch.where(exp == 0, man, man | 0x80)
    e = torch.where(exp == 0, torch.ones_like(exp), exp)  # subnormals share exponent 1
    shift = e - 134 + frac                                # value = m * 2**(e-134)
    if bool((shift > 55).any()):
        raise OverflowError("bf16 term too large for the fixed-point format")
    left = m << shift.clamp(min=0)
    right = m >> (-shift).clamp(min=0, max=63)
    mag = torch.where(shift >= 0, left, right)
    return torch.where(sign == 1, -mag, mag)


def fixed_to_bf16(n: torch.Tensor, frac) -> torch.Tensor:
    """Signed int64 fixed point -> bf16, round-to-nearest-even, exact (``frac``: int, or an int64 tensor per element)."""
    n = n.to(torch.int64)
    sign = (n < 0).to(torch.int64)
    mag = n.abs()
    mag_np = mag.numpy().astype(np.uint64)
    # position of the most significant set bit (float64 guess, then corrected)
    guess = np.zeros(mag_np.shape, dtype=np.int64)
    nz = mag_np != 0
    guess[nz] = np.frexp(mag_np[nz].astype(np.float64))[1] - 1
    too_high = nz & ((np.uint64(1) << guess.astype(np.uint64)) > mag_np)
    guess[too_high] -= 1
    msb = torch.from_numpy(guess)
    shift = (msb - 7).clamp(min=0)
    q = mag >> shift
    rem = mag & ((torch.ones_like(mag) << shift) - 1)
    half = torch.where(shift > 0, torch.ones_like(mag) << (shift - 1).clamp(min=0), torch.zeros_like(mag))
    up = (shift > 0) & ((rem > half) | ((rem == half) & ((q & 1) == 1)))
    q = q + up.to(torch.int64)
    # left-justify values with fewer than 8 significant bits
    lshift = (7 - msb).clamp(min=0)
    q = q << lshift
    carry = q >= 256
    q = torch.where(carry, q >> 1, q)
    e = msb - frac + carry.to(torch.int64) + 127
    if bool((e >= 255).any()):
        raise OverflowError("fixed-point sum above the bf16 range")
    bits = (sign << 15) | (e << 7) | (q & 0x7F)
    # below the normal range (block-floating sums only: frac > 126): round at the subnormal spacing 2**-133 like IEEE --
    # torch's CPU kernels do not flush, and the reference runs on into this regime before a bandit row dies
    sub = (msb - frac + 127 <= 0) & (mag != 0)
    if bool(sub.any()):
        fr = frac if isinstance(frac, torch.Tensor) else torch.full_like(mag, int(frac))
        sh = (fr - 133)
        shr = sh.clamp(min=1, max=62)
        qs = torch.where(sh > 62, torch.zeros_like(mag), mag >> shr)
        rem = mag & ((torch.ones_like(mag) << shr) - 1)
        half_s = torch.ones_like(mag) << (shr - 1)
        qs = qs + ((sh <= 62) & ((rem > half_s) | ((rem == half_s) & ((qs & 1) == 1)))).to(torch.int64)
        qs = torch.where(sh <= 0, mag << (-sh).clamp(min=0, max=62), qs)
        bits = torch.where(sub, (sign << 15) | qs, bits)
    bits = torch.where(mag == 0, torch.zeros_like(bits), bits)
    return bits_to_bf16(bits)


def exact_segment_sum(values: torch.Tensor, seg: torch.Tensor, nseg: int, frac: int) -> torch.Tensor:
    """``dgl.ops.copy_e_sum`` under the contract: exact sum per segment -> bf16.

    Returns (bf16 sums [nseg], int64 fixed-point sums [nseg])."""
    fx = bf16_to_fixed(values, frac)
    acc = torch.zeros(nseg, dtype=torch.int64)
    acc.index_add_(0, seg.to(torch.int64), fx)
    return fixed_to_bf16(acc, frac), acc


def exact_segment_sum_rel(values: torch.Tensor, seg: torch.Tensor, nseg: int, frac: int) -> torch.Tensor:
    """exact_segment_sum in block-floating form, for sums whose terms may sit anywhere in bf16's range (the EXP3 weights
    of a seed column, bandit_sampler.py:129, once the bandit has concentrated a row): every segment's terms are scaled
    by 2**s, s = max(0, 126 - largest biased exponent in the segment), before the exact integer addition.  Bits a term
    loses below the accumulator's unit go to a second accumulator 40 bits finer, and what falls below even that sets a
    sticky flag that breaks rounding ties upward (all terms >= 0) -- so the rounded sum is the rounding of the EXACT sum
    (found by the long reference run tests/golden/collapse0_*: tiny weights beside two large ones turned a "just above the
    tie" into a tie).  Subnormal results are produced like IEEE.  (csrc/common.cuh: rel_frac, bf_to_fixed_wide,
    fixed_wide_to_bf; csrc/sampler.hip: k_col_sums.)  Returns (bf16 sums, the int64 main accumulators)."""
    bits = bf16_bits(values)
    exp = (bits >> 7) & 0xFF
    man = bits & 0x7F
    if bool((exp == 255).any()):
        raise FloatingPointError("non-finite bf16 term in an exact reduction")
    if bool(((bits >> 15) == 1).any() & (values != 0).any() & (values < 0).any()):
        raise FloatingPointError("negative weight in a block-floating sum")
    m = torch.where(exp == 0, man, man | 0x80)
    e = torch.where(exp == 0, torch.ones_like(exp), exp)
    seg = seg.to(torch.int64)
    emax = torch.ones(nseg, dtype=torch.int64)
    emax.scatter_reduce_(0, seg, e, reduce="amax", include_self=True)
    fr = frac + (126 - emax).clamp(min=0)
    shift = e - 134 + fr[seg]
    if bool((shift > 55).any()):
        raise OverflowError("bf16 term too large for the fixed-point format")
    rs = (-shift).clamp(min=0)
    top = torch.where(shift >= 0, m << shift.clamp(min=0), torch.where(rs >= 8, torch.zeros_like(m), m >> rs.clamp(max=62)))
    r = torch.where(shift >= 0, torch.zeros_like(m), torch.where(rs >= 8, m, m & ((torch.ones_like(m) << rs.clamp(max=62)) - 1)))
    d = (rs - 40).clamp(min=0)
    lo = torch.where(rs <= 40, r << (40 - rs).clamp(min=0), torch.where(d >= 8, torch.zeros_like(r), r >> d.clamp(max=62)))
    lost = (rs > 40) & ((d >= 8) & (r != 0) | ((r & ((torch.ones_like(r) << d.clamp(max=62)) - 1)) != 0))
    acc = torch.zeros(nseg, dtype=torch.int64).index_add_(0, seg, top)
    acc_lo = torch.zeros(nseg, dtype=torch.int64).index_add_(0, seg, lo)
    sticky = torch.zeros(nseg, dtype=torch.int64).index_add_(0, seg, lost.to(torch.int64)) > 0
    out = fixed_to_bf16(acc, fr)
    for k in torch.nonzero((acc_lo != 0) | sticky).flatten().tolist():       # rare: Python ints, exact
        out[k] = _int_to_bf16_sticky((int(acc[k]) << 40) + int(acc_lo[k]), int(fr[k]) + 40, bool(sticky[k]))
    return out, acc


def _int_to_bf16_sticky(total: int, frac: int, sticky: bool) -> torch.Tensor:
    """exact non-negative Python int * 2**-frac (+ a positive remainder below the last bit if ``sticky``) -> bf16, RNE."""
    if total == 0:
        return torch.zeros((), dtype=torch.bfloat16)
    msb = total.bit_length() - 1

    def rshift(sh):
        if sh <= 0:
            return total << (-sh)
        q, rem, half = total >> sh, total & ((1 << sh) - 1), 1 << (sh - 1)
        return q + 1 if (rem > half or (rem == half and (sticky or (q & 1)))) else q

    if msb - frac + 127 <= 0:
        return bits_to_bf16(torch.tensor(rshift(frac - 133), dtype=torch.int64))
    q, e = rshift(msb - 7), msb - frac + 127
    if q >= 256:
        q >>= 1
        e += 1
    if e >= 255:
        raise OverflowError("fixed-point sum above the bf16 range")
    return bits_to_bf16(torch.tensor((e << 7) | (q & 0x7F), dtype=torch.int64))


# ----------------------------------------------------------------------------
# Three-limb accumulator for the whole-row L1 norm of the EXP3 weights
# (bandit_sampler.py:249).  value = (l0 + l1*2**32 + l2*2**64) * 2**-ROW_FRAC
# ----------------------------------------------------------------------------
ROW_FRAC = 64


def row_exact_sum(values: torch.Tensor):
    """Exact sum of a non-negative bf16 row as a Python int scaled by 2**ROW_FRAC."""
    bits = bf16_bits(values)
    exp = (bits >> 7) & 0xFF
    man = bits & 0x7F
    if bool((exp == 255).any()) or bool((values < 0).any()):
        raise FloatingPointError("row sum needs finite non-negative weights")
    m = torch.where(exp == 0, man, man | 0x80)
    e = torch.where(exp == 0, torch.ones_like(exp), exp)
    # one pass: sum of mantissas per exponent (exact in fp64: < 2^8 * 2^40), then shift in Python ints
    msum = torch.bincount(e, weights=m.double(), minlength=256)
    total = 0
    for ee in torch.nonzero(msum).flatten().tolist():
        s = ee - 134 + ROW_FRAC
        if s >= 0:
            total += int(msum[ee]) << s
        else:
            # below 2^-ROW_FRAC truncation happens per term
            total += int((m[e == ee] >> min(-s, 63)).sum())
    return total


def int_to_bf16(total: int, frac: int) -> torch.Tensor:
    """Exact Python-int fixed point -> bf16 scalar tensor (RNE)."""
    if total == 0:
        return torch.zeros((), dtype=torch.bfloat16)
    msb = total.bit_length() - 1
    if msb > 7:
        shift = msb - 7
        q = total >> shift
        rem = total & ((1 << shift) - 1)
        half = 1 << (shift - 1)
        if rem > half or (rem == half and (q & 1)):
            q += 1
    else:
        q = total << (7 - msb)
    e = msb - frac + 127
    if q >= 256:
        q >>= 1
        e += 1
    assert 0 < e < 255
    return bits_to_bf16(torch.tensor((e << 7) | (q & 0x7F), dtype=torch.int64))
