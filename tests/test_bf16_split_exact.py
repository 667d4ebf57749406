"""The arithmetic claim behind the f32 heads on the bf16 matrix pipe (csrc/ppo_step_bf16.hip: split3 / mfma3):

    every f32 number w is EXACTLY hi + mid + lo with hi = bf16(w), mid = bf16(w - hi), lo = bf16(w - hi - mid)
    (round-to-nearest-even conversions, the two differences exact in f32), and a product of two bf16 numbers is exact in f32

so that  x . w  with x in bf16 is three bf16 MFMAs with f32 accumulation -- the same real-number sum as the f32 dot product.
Checked here on the CPU with torch's own bf16 conversions, exhaustively over all 2^23 significands of a binade (the split is
scale-invariant for normal numbers), both signs, and over the exponent range the heads' weights and gradients live in."""

import torch


def _split3(w: torch.Tensor):
    hi = w.to(torch.bfloat16)
    r1 = w - hi.float()          # exact: |r1| <= half a bf16 ulp of w, representable in f32
    mid = r1.to(torch.bfloat16)
    r2 = r1 - mid.float()        # exact
    lo = r2.to(torch.bfloat16)
    return hi, mid, lo


def _exact(w: torch.Tensor) -> bool:
    hi, mid, lo = _split3(w)
    return bool(torch.equal(hi.double() + mid.double() + lo.double(), w.double()))


def test_every_significand_splits_exactly():
    bits = torch.arange(0, 1 << 23, dtype=torch.int32)
    for sign in (0, -(1 << 31)):
        for exponent in (127, 100, 140):  # [1, 2), ~1e-8, ~1e4
            w = (bits | (exponent << 23) | sign).view(torch.float32)
            assert _exact(w), (sign, exponent)


def test_random_weights_and_gradients_split_exactly():
    g = torch.Generator().manual_seed(0)
    for scale in (1.0, 0.05, 1e-4, 3e-7, 50.0):
        assert _exact(torch.randn(1_000_000, generator=g) * scale), scale


def test_where_the_split_stops_being_exact():
    """Honest boundary: once w - hi - mid falls below bf16's smallest subnormal (2^-133) the third term is lost -- for |w| below
    ~2^-110 (1e-33).  Head weights and loss gradients are nowhere near; the kernel makes no claim there."""
    w = torch.tensor([1.2345678e-36], dtype=torch.float32)
    hi, mid, lo = _split3(w)
    rec = hi.double() + mid.double() + lo.double()
    assert abs(float(rec - w.double())) <= 2.0 ** -133  # inexact at most by the dropped subnormal tail, never by more


def test_bf16_products_are_exact_in_f32():
    g = torch.Generator().manual_seed(1)
    a = torch.randn(1_000_000, generator=g).to(torch.bfloat16)
    b = (torch.randn(1_000_000, generator=g) * 0.1).to(torch.bfloat16)
    assert torch.equal((a.float() * b.float()).double(), a.double() * b.double())  # 8 x 8 significand bits fit the 24 of f32


def test_three_plane_dot_equals_the_f32_dot_up_to_summation_order():
    """x in bf16, w in f32: sum_k x_k (hi_k + mid_k + lo_k) evaluated in f64 is the f64 value of sum_k x_k w_k -- the planes
    change nothing but the order in which f32 partial sums are rounded."""
    g = torch.Generator().manual_seed(2)
    x = torch.tanh(torch.randn(64, 256, generator=g)).to(torch.bfloat16)
    w = torch.randn(8, 256, generator=g) * 0.06
    hi, mid, lo = _split3(w)
    planes = (x.double() @ hi.double().T) + (x.double() @ mid.double().T) + (x.double() @ lo.double().T)
    assert torch.allclose(planes, x.double() @ w.double().T, rtol=0, atol=1e-13)
