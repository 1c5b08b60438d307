"""Golden vectors for ES_MOE's dispatch modes and eval-time state, produced by the REAL reference module
(ultralytics/nn/modules/moe/modules.py:410-779) on CPU.  Run in the build container (needs /root/reference):

    python tests/golden/make_golden_esmoe.py

Cases (esmoe_<name>.npz): x, y, the reference's `expert_usage_counts` and `load_balancing_loss` buffers after the eval
forward (modules.py:706-741), the router's weights, and the full state_dict:
  sparse   default ctor (top_k=2, sparse dispatch, threshold 0.4)
  dense    use_sparse_inference=False: hard top-k routing weights, every expert summed (modules.py:648-656)
  disabled enable_sparse_inference(False) after construction (same forward as `dense`, other entry point)
  all      top_k=None: plain softmax over all experts, dense
  k3of4    top_k=3, sparse, threshold 0.2
The oracle restatement (oracle/model_ref.es_moe) is checked bit-exact against the reference here."""
import sys
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
ROOT = HERE.parent.parent
sys.path.insert(0, str(ROOT))
from oracle import model_ref, refboot  # noqa: E402

refboot.boot()
from ultralytics.nn.modules.moe.modules import ES_MOE  # noqa: E402

from yolo_master_amd.weights import synth_state_dict  # noqa: E402


def case(name, C, B, H, W, seed, post=None, **kw):
    torch.manual_seed(seed)
    m = ES_MOE(C, C, **kw)
    sd = synth_state_dict(m.state_dict(), seed=seed, calib=None)
    m.load_state_dict(sd)
    for mod in m.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.eps = 1e-3
    if post:
        post(m)
    m.eval()
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(B, C, H, W, generator=g) + 1.5 * torch.randn(B, C, 1, 1, generator=g)
    routes = {}
    m.routing.register_forward_hook(lambda mod, i, o: routes.__setitem__("w", o[:, :, 0, 0].clone()))
    with torch.inference_mode():
        y = m(x)
        info = {}
        top_k = kw.get("top_k", 2)
        oy = model_ref.es_moe({f"m.{k}": v for k, v in sd.items()}, "m", x, top_k=m.num_experts if top_k is None else top_k,
                              thr=kw.get("dynamic_threshold", 0.4), info=info, sparse=m._eager_sparse_enabled(),
                              hard_top_k=top_k is not None)
    r = info["m"]
    exact = torch.equal(y, oy) and torch.equal(routes["w"], r["route_w"])
    print(f"[esmoe_{name}] oracle bit-exact vs reference: {exact} (max|dy| {(y - oy).abs().max().item():.2e}); usage "
          f"{[round(float(v), 4) for v in m.expert_usage_counts]} (oracle {[round(float(v), 4) for v in r['usage']]}) "
          f"loss {float(m.load_balancing_loss):.6f} (oracle {float(r['lb_loss']):.6f}); retained/expert {r['retained'].sum(0).tolist()}")
    assert exact and torch.equal(m.expert_usage_counts, r["usage"]) and torch.equal(m.load_balancing_loss, r["lb_loss"])
    rec = {"x": x.numpy(), "y": y.numpy(), "route_w": routes["w"].numpy(), "usage": m.expert_usage_counts.numpy(),
           "lb_loss": m.load_balancing_loss.numpy(), "retained": r["retained"].numpy(), "gate_w": r["gate_w"].numpy(),
           "keys": np.array(list(sd.keys()))}
    rec.update({f"sd::{k}": v.numpy() for k, v in sd.items()})
    np.savez_compressed(HERE / f"esmoe_{name}.npz", **rec)


if __name__ == "__main__":
    torch.set_num_threads(4)
    case("sparse", 64, 6, 14, 18, seed=21)
    case("dense", 64, 6, 14, 18, seed=22, use_sparse_inference=False)
    case("disabled", 64, 5, 12, 12, seed=23, post=lambda m: m.enable_sparse_inference(False))
    case("all", 64, 5, 12, 12, seed=24, top_k=None)
    case("k3of4", 64, 6, 10, 14, seed=25, top_k=3, dynamic_threshold=0.2)
