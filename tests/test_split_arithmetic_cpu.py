"""The opt-in split-bf16 arithmetic of the Winograd GEMMs (csrc/wino_split.hip.inc), on the CPU: (1) the host side of the switch
(tai_conv3x3_wino_set_arithmetic: process-wide mode, buffer sizes, refusal of bad arguments -- nothing is launched); (2) the
numerical claim the kernel rests on, emulated in PyTorch: a product of fp32 operands taken as three bf16 terms each and SIX bf16
products accumulated in fp32 is at or below the error of the plain fp32 product against float64, the two-term / three-product form is
not (a bf16 x bf16 product is exact in fp32, so an fp32 matmul of bf16-rounded operands is the bf16 MFMA's arithmetic up to summation
order).  Layers: nn.Conv2d(C, K, 3, padding=1) of src/models/mcnet/mcnet.py:79-224 as Winograd F(2x2, 3x3) GEMMs."""
import pytest
import torch

from video_frame_inpainting_amd import _native


def test_arithmetic_switch_host_side():
    L = _native.lib()
    assert L.tai_conv3x3_wino_get_arithmetic() == 0                         # the default is the fp32 MFMA
    assert L.tai_conv3x3_wino_weight_floats(64, 64) == 16 * 64 * 64
    assert L.tai_conv3x3_wino_set_arithmetic(1) == 0                        # returns the previous mode
    try:
        assert L.tai_conv3x3_wino_get_arithmetic() == 1
        assert L.tai_conv3x3_wino_weight_floats(64, 64) == 40 * 64 * 64     # fp32 image + three bf16 terms per transformed weight
        assert L.tai_conv3x3_wino_weight_floats(51, 65) == 40 * 64 * 72
        assert L.tai_conv3x3_wino_set_arithmetic(2) < 0 and b'set_arithmetic' in L.tai_sepconv_last_error()
        assert L.tai_conv3x3_wino_get_arithmetic() == 1                     # a refused call changes nothing
    finally:
        assert L.tai_conv3x3_wino_set_arithmetic(0) == 1
    assert L.tai_conv3x3_wino_get_arithmetic() == 0


def _split(x, terms):
    out, rest = [], x
    for _ in range(terms):
        t = rest.to(torch.bfloat16).to(x.dtype)        # round to nearest even, as v_cvt_pk_bf16_f32
        out.append(t)
        rest = rest - t                                # exact in fp32
    return out, rest


def test_three_bf16_terms_hold_an_fp32_value_exactly():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1 << 16, generator=g) * torch.exp(torch.randn(1 << 16, generator=g) * 4)
    terms, rest = _split(x, 3)
    assert torch.equal(terms[0] + terms[1] + terms[2], x) and float(rest.abs().max()) == 0.0
    _, rest2 = _split(x, 2)
    assert float((rest2.abs() / x.abs()).max()) > 2 ** -18                  # two terms do not


@pytest.mark.parametrize('C', [64, 512])
def test_six_bf16_products_are_at_or_below_the_fp32_products_error(C):
    # one transform position of a Winograd GEMM: [K, C] x [C, tiles], operands with the spread of transformed weights / patches
    g = torch.Generator().manual_seed(C)
    U = torch.randn(64, C, generator=g) * (2.0 / (9 * C)) ** 0.5
    V = torch.randn(C, 4096, generator=g) * 2.0
    ref = U.double() @ V.double()
    scale = (U.double().abs() @ V.double().abs()).max()
    e_f32 = float(((U @ V).double() - ref).abs().max() / scale)
    (uh, um, ul), _ = _split(U, 3)
    (vh, vm, vl), _ = _split(V, 3)
    six = ((uh @ vl + ul @ vh + um @ vm) + (uh @ vm + um @ vh)) + uh @ vh
    e_six = float((six.double() - ref).abs().max() / scale)
    three = (uh @ vm + um @ vh) + uh @ vh
    e_three = float((three.double() - ref).abs().max() / scale)
    assert e_six <= 1.25 * e_f32, (e_six, e_f32)
    assert e_three > 5 * e_f32, (e_three, e_f32)                             # two terms, three products: an order of magnitude worse
