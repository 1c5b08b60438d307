"""The one-kernel ES-MoE expert body (csrc/esfused.hip: wave-specialised depthwise stencil -> pointwise grouped GEMM, LDS-DMA halo staged
once for both retained experts of an image) on the CPU lane emulator (tests/hostemu: the unmodified kernel source compiled for the
host, lanes as fibers, LDS-DMA / MFMA / barriers emulated), through the product's own wrappers:

  * bit-identical to the two-kernel form it replaces (ymk_esmoe_dw + ymk_esmoe_pw, themselves held to the oracle / the real
    reference's vectors elsewhere) — same operand rounding, same accumulation order;
  * and against the torch restatement of the contract (tests/emu_ops.py) within the 16-bit tolerance.

Shapes exercise what the kernel's bookkeeping can get wrong: maps that are not multiples of the 8 x 16 / 8 x 8 tile (tiles hanging over
the right / bottom edge), images with two, one and NO retained expert, every stencil size, runs of items that start and end inside an
image (the emulator launches 3 workgroups), both channel widths."""
import pytest
import torch

from tests import emu_ops


@pytest.fixture
def host_ops(hostlib, monkeypatch):
    from yolo_master_amd import ops

    monkeypatch.setattr(ops, "lib", hostlib)
    monkeypatch.setattr(ops, "_stream", lambda: None)
    monkeypatch.setattr(ops, "require_gpu", lambda t, what="": None)
    return ops


def _rnd(*shape, seed, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def _csr(sel, E):
    off, pairs = [0], []
    flat = sel.reshape(-1).tolist()
    for e in range(E):
        pairs += [i for i, v in enumerate(flat) if v == e]
        off.append(len(pairs))
    return torch.tensor(off, dtype=torch.int32), torch.tensor(pairs + [0] * (len(flat) - len(pairs)), dtype=torch.int32)


CASES = [
    # C, B, H, W, sel [B][2]
    (128, 4, 11, 21, [[0, 3], [2, -1], [-1, -1], [3, 1]]),        # 8 x 16 tiles over a 11 x 21 map: ragged right and bottom tiles; 2 / 1 / 0 / 2 experts
    (128, 2, 16, 32, [[1, -1], [3, 2]]),                          # whole tiles; k = 5 alone, then 9 + 7
    (256, 3, 9, 12, [[3, 0], [1, -1], [2, 3]]),                   # 8 x 8 tiles, four channel chunks (the router always retains its rank-0 slot: dropped slots trail)
    (256, 1, 8, 8, [[0, 1]]),                                     # one tile, one item, one workgroup
]


def run_fused_case(ops_, case, dev="cpu", dtype=torch.bfloat16, check_emu=True):
    C, B, H, W, sel = case
    E, top_k, ks = 4, 2, [3, 5, 7, 9]
    sel = torch.tensor(sel, dtype=torch.int32)
    if sel.shape[0] != B:                                              # a pattern to repeat over a larger batch
        sel = sel.repeat((B + sel.shape[0] - 1) // sel.shape[0], 1)[:B].contiguous()
    x = torch.zeros((B, H, W, C + 16), dtype=dtype, device=dev)[..., :C]          # a channel-slice view: ldx != C
    x.copy_(_rnd(B, H, W, C, seed=1).to(dtype))
    parts, offs, off = [], [], 0
    for e, k in enumerate(ks):
        w = (_rnd(k * k, C, seed=10 + e) / k).to(dtype)
        parts.append(w.reshape(-1)); offs.append(off); off += w.numel()
    dw_w, dw_off, ksz = torch.cat(parts).contiguous().to(dev), torch.tensor(offs, dtype=torch.int32, device=dev), torch.tensor(ks, dtype=torch.int32, device=dev)
    pw_w = (_rnd(E, C, C, seed=2) * C ** -0.5).to(dtype).contiguous().to(dev)
    pw_b, ns, nt = _rnd(E, C, seed=3, scale=0.3).to(dev), (1.0 + _rnd(C, seed=4, scale=0.1)).to(dev), _rnd(C, seed=5, scale=0.2).to(dev)
    gate = torch.zeros(B, E)
    for b in range(B):
        for k in range(top_k):
            if sel[b, k] >= 0:
                gate[b, sel[b, k]] = 0.3 + 0.1 * ((b % 5) + 2 * k)
    assert ops_.esmoe_fused_supported(dtype, C, C, H, W, 9, E, top_k)
    csr_off, csr_pair = _csr(sel, E)
    sel_d, gate_d = sel.to(dev), gate.to(dev)
    dw = ops_.esmoe_dw(x, dw_w, dw_off, ksz, 9, top_k, sel_d, csr_off.to(dev), csr_pair.to(dev))
    want = ops_.esmoe_pw(dw, B, H, W, pw_w, pw_b, ns, nt, top_k, sel_d, gate_d)
    out = torch.full((B, H, W, C + 8), 7.0, dtype=dtype, device=dev)[..., :C]     # a view with a row pitch: ldy != Cout
    got = ops_.esmoe_fused(x, dw_w, dw_off, ksz, 9, pw_w, pw_b, ns, nt, top_k, sel_d, gate_d, out=out)
    gi, wi = got.cpu().view(torch.int16), want.cpu().view(torch.int16)
    assert torch.equal(gi, wi), f"fused != depthwise + pointwise: {int((gi != wi).sum())} elements differ, max |d| {float((got.float() - want.float()).abs().max()):.3e}"
    assert torch.all(out.as_strided((B, H, W, 8), out.stride(), C) == 7.0), "wrote past the channel range"
    if check_emu:
        ref = emu_ops.esmoe_fused(x.cpu(), dw_w.cpu(), dw_off.cpu(), ksz.cpu(), 9, pw_w.cpu(), pw_b.cpu(), ns.cpu(), nt.cpu(), top_k, sel, gate)
        assert torch.allclose(got.cpu().float(), ref.float(), atol=2e-2, rtol=2e-2), float((got.cpu().float() - ref.float()).abs().max())


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"C{c[0]}-{c[2]}x{c[3]}-B{c[1]}")
def test_fused_expert_body_equals_the_two_kernel_form(case, host_ops):
    run_fused_case(host_ops, case)


def test_fused_expert_body_is_refused_outside_its_shapes(host_ops):
    f = host_ops.esmoe_fused_supported
    assert not f(torch.float32, 128, 128, 16, 16, 9, 4, 2) and not f(torch.bfloat16, 512, 512, 20, 20, 9, 4, 2)
    assert not f(torch.bfloat16, 128, 128, 16, 16, 11, 4, 2) and not f(torch.bfloat16, 128, 128, 16, 16, 9, 4, 3)
    assert not f(torch.bfloat16, 128, 256, 16, 16, 9, 4, 2) and f(torch.bfloat16, 256, 256, 80, 80, 9, 4, 2)
