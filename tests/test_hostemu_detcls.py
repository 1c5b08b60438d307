"""Fused Detect class branch (csrc/detcls.hip: DWConv3x3 -> Conv1x1 -> DWConv3x3 -> Conv1x1 -> Conv2d 1x1, one kernel) on the CPU lane
emulator against the five-convolution composition it replaces (torch fp32 arithmetic on bf16 operands, every stage rounded to bf16).
Map sizes exercise the borders; nc values the padded / partial last fragment.  Shared with the GPU test."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

CASES = [(1, 8, 16, 128, 80), (1, 5, 7, 128, 3), (2, 11, 21, 256, 80), (1, 9, 33, 128, 20)]   # B, H, W, Cin, nc


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def operands(case):
    from yolo_master_amd import ops

    B, H, W, Cin, nc = case
    g = torch.Generator().manual_seed(H * 100 + W + nc)
    bf = torch.bfloat16
    x = torch.randn(B, H, W, Cin, generator=g).to(bf)
    ncpad = (nc + 3) // 4 * 4
    w = {"d1": torch.randn(Cin, 1, 3, 3, generator=g) * 0.4, "p1": torch.randn(128, Cin, 1, 1, generator=g) * Cin ** -0.5,
         "d2": torch.randn(128, 1, 3, 3, generator=g) * 0.4, "p2": torch.randn(128, 128, 1, 1, generator=g) * 128 ** -0.5,
         "w3": torch.cat([torch.randn(nc, 128, 1, 1, generator=g) * 128 ** -0.5, torch.zeros(ncpad - nc, 128, 1, 1)])}
    b = {k: torch.randn(v.shape[0], generator=g) * 0.3 for k, v in w.items()}
    b["w3"][nc:] = 0
    packed = {k: (ops.pack_dw_weight(v, bf) if k[0] == "d" else ops.pack_conv_weight(v, bf)) for k, v in w.items()}
    return x, w, b, packed


def reference(x, w, b):
    bf = torch.bfloat16
    t = x.float().permute(0, 3, 1, 2)
    for k in ("d1", "p1", "d2", "p2"):
        wk = w[k].to(bf).float()
        t = F.conv2d(t, wk, b[k], 1, 1, 1, t.shape[1]) if k[0] == "d" else F.conv2d(t, wk, b[k])
        t = F.silu(t).to(bf).float()
    return F.conv2d(t, w["w3"].to(bf).float(), b["w3"]).permute(0, 2, 3, 1)


def run_case(lib, case, dev="cpu", stream=None):
    B, H, W, Cin, nc = case
    x, w, b, packed = operands(case)
    ref = reference(x, w, b)
    ncpad = (nc + 3) // 4 * 4
    xd = x.to(dev)
    yb = torch.full((B, H, W, ncpad + 4), 7.0, dtype=torch.float32, device=dev)
    pk = {k: v.to(dev) for k, v in packed.items()}
    bd = {k: v.to(dev) for k, v in b.items()}
    assert lib.ymk_detect_cls_fused_supported(1, Cin, 128, nc) and not lib.ymk_detect_cls_fused_supported(1, 64, 128, nc)
    rc = lib.ymk_detect_cls_fused(_p(xd), xd.stride(2), B, H, W, Cin, _p(pk["d1"]), _p(bd["d1"]), _p(pk["p1"]), pk["p1"].shape[1], _p(bd["p1"]),
                                  _p(pk["d2"]), _p(bd["d2"]), _p(pk["p2"]), pk["p2"].shape[1], _p(bd["p2"]), _p(pk["w3"]), pk["w3"].shape[1],
                                  _p(bd["w3"]), ncpad, _p(yb), yb.stride(2), None, 0, 0, 0, None, None, stream)
    assert rc == 0
    got = yb[..., :ncpad].cpu()
    err = (got - ref).abs()
    scale = max(1.0, float(ref.abs().max()))
    assert float(err.max()) <= 4e-2 * scale, f"{case}: max err {float(err.max()):.3e}"
    assert float(err.mean()) <= 3e-3 * scale, f"{case}: mean err {float(err.mean()):.3e}"
    assert bool((yb[..., ncpad:].cpu() == 7.0).all()), "bytes between pixels were touched"
    # round 4: the decode's class half in the epilogue — sigmoid into the class rows of y and the per-anchor best class, with and
    # without the logits: bit-identical to ymk_detect_decode on the logits above
    A, a_off = H * W + 9, 5
    box = torch.zeros((B, H, W, 64), dtype=torch.float32, device=dev)
    y_ref = torch.full((B, 4 + nc, A), 3.0, dtype=torch.float32, device=dev)
    bc_ref, bi_ref = torch.full((B, A), -2.0, device=dev), torch.full((B, A), -2, dtype=torch.int32, device=dev)
    cls_l = yb[..., :ncpad].contiguous()
    assert lib.ymk_detect_decode(_p(box), _p(cls_l), _p(y_ref), B, H, W, 16, nc, ncpad, 8.0, a_off, A, _p(bc_ref), _p(bi_ref), stream) == 0
    for keep in (True, False):
        y2 = torch.full((B, 4 + nc, A), 3.0, dtype=torch.float32, device=dev)
        bc, bi = torch.full((B, A), -2.0, device=dev), torch.full((B, A), -2, dtype=torch.int32, device=dev)
        raw2 = torch.full((B, H, W, ncpad), 7.0, dtype=torch.float32, device=dev)
        rc = lib.ymk_detect_cls_fused(_p(xd), xd.stride(2), B, H, W, Cin, _p(pk["d1"]), _p(bd["d1"]), _p(pk["p1"]), pk["p1"].shape[1], _p(bd["p1"]),
                                      _p(pk["d2"]), _p(bd["d2"]), _p(pk["p2"]), pk["p2"].shape[1], _p(bd["p2"]), _p(pk["w3"]), pk["w3"].shape[1],
                                      _p(bd["w3"]), ncpad, _p(raw2) if keep else None, ncpad, _p(y2), nc, a_off, A, _p(bc), _p(bi), stream)
        assert rc == 0
        # (the epilogue's sigmoid is the hardware exp / rcp pair: within a few ulp of detect_decode's libm expression)
        d = float((y2[:, 4:].cpu() - y_ref[:, 4:].cpu()).abs().max())
        assert d <= 5e-7, f"{case} keep={keep}: class rows differ from detect_decode's by {d:.2e}"
        assert bool((y2[:, :4].cpu() == 3.0).all()), "box rows were touched"
        lv = slice(a_off, a_off + H * W)
        conf, j = y2[:, 4:, lv].cpu().max(1)                          # first maximum in class order, of the values stored in y
        assert torch.equal(bc[:, lv].cpu(), conf) and torch.equal(bi[:, lv].cpu().long(), j), f"{case}: best class differs"
        assert bool((bc[:, :a_off].cpu() == -2.0).all()) and bool((bc[:, a_off + H * W:].cpu() == -2.0).all())
        assert float((bc.cpu() - bc_ref.cpu()).abs().max()) <= 5e-7
        if keep:
            assert torch.equal(raw2.cpu(), yb[..., :ncpad].cpu())
    return got


@pytest.mark.parametrize("case", CASES)
def test_detect_cls_fused_on_emulator(case, hostlib):
    run_case(hostlib, case)


BOX_CASES = [(2, 8, 16, 8.0), (1, 5, 7, 16.0), (3, 9, 21, 32.0)]   # B, H, W, stride


def run_box_case(ops, case, dev="cpu"):
    """ymk_detect_box_tail (1x1 + bias -> DFL -> dist2bbox -> rows 0..3 of y) against the 1x1 convolution core + ymk_detect_decode."""
    B, H, W, stride = case
    nc, reg_max = 5, 16
    g = torch.Generator().manual_seed(H * 31 + W)
    bf = torch.bfloat16
    x = (torch.randn(B, H, W, 64, generator=g) * 1.5).to(bf).to(dev)
    w = ops.pack_conv_weight(torch.randn(64, 64, 1, 1, generator=g) * 0.35, bf).to(dev)
    bias = (torch.randn(64, generator=g) * 0.5 + 1.0).to(dev)
    A, a_off = H * W + 11, 4
    logits = ops.conv2d(x, w, bias, 1, 1, False, out_dtype=torch.float32)
    y_ref = torch.full((B, 4 + nc, A), 3.0, dtype=torch.float32, device=dev)
    ops.detect_decode(logits, torch.zeros((B, H, W, nc), dtype=torch.float32, device=dev), y_ref, stride, a_off, reg_max)
    for keep in (True, False):
        y = torch.full((B, 4 + nc, A), 3.0, dtype=torch.float32, device=dev)
        raw = ops.detect_box_tail(x, w, bias, y, stride, a_off, reg_max, raw=keep)
        d = float((y[:, :4, a_off: a_off + H * W] - y_ref[:, :4, a_off: a_off + H * W]).abs().max())
        assert d <= 1e-4 * stride, f"{case}: boxes differ by {d:.3e} px"
        assert bool((y[:, 4:].cpu() == 3.0).all()) and bool((y[:, :4, :a_off].cpu() == 3.0).all()) and \
            bool((y[:, :4, a_off + H * W:].cpu() == 3.0).all()), "wrote outside the level's box rows"
        if keep:
            assert float((raw - logits).abs().max()) <= 1e-5 * max(1.0, float(logits.abs().max()))
        else:
            assert raw is None
    assert float((y_ref[:, 2:4, a_off: a_off + H * W]).min()) > 0


@pytest.mark.parametrize("case", BOX_CASES)
def test_detect_box_tail_on_emulator(case, hostlib, monkeypatch):
    from yolo_master_amd import ops

    monkeypatch.setattr(ops, "lib", hostlib)
    monkeypatch.setattr(ops, "_stream", lambda: None)
    monkeypatch.setattr(ops, "require_gpu", lambda t, what="": None)
    run_box_case(ops, case)
