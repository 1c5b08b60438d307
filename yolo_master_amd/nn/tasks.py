"""Model-YAML parser and detection model of the ymk path.

``parse_model`` follows the reference's YAML contract (ultralytics/nn/tasks.py:2022-2270:
``[from, repeats, module, args]`` rows, ``scales``, width/depth/max_channels rules, and the
mixture-module argument adaptation of ultralytics/nn/mixture_registry.py:84-156) for the module
set of the YOLO-Master detection graphs.  ``DetectionModel`` keeps the reference's attribute
surface (``model``, ``save``, ``stride``, ``yaml``, ``names``, ``fuse()``, ``forward/predict``;
tasks.py:123-218, 502-560) and walks the graph on NHWC buffers through libymk.
"""
from __future__ import annotations

import ast
import contextlib
import math
import re
from copy import deepcopy
from pathlib import Path

import torch
import torch.nn as nn
import yaml

from .. import ops
from .mixture import MIXTURE_BOUNDARY_MODULES, MIXTURE_BOUNDARY_REPEAT
from .modules import (A2C2f, C2f, C3, C3k, C3k2, ES_MOE, Bottleneck, Concat, Conv, Detect, DetectPreds, DWConv, LazyUpsample, Segment,
                      VirtualCat, YmkModule, _is_silu, set_compute_dtype)

CFG_DIR = Path(__file__).resolve().parent.parent / "cfg"

BASE_MODULES = {"Conv": Conv, "DWConv": DWConv, "Bottleneck": Bottleneck, "C2f": C2f, "C3": C3, "C3k2": C3k2,
                "A2C2f": A2C2f}
REPEAT_MODULES = {C2f, C3, C3k2, A2C2f}
# the plugin registry the reference resolves YAML names through (mixture_registry.py:39-81)
# (the config-5 modules are registered as drop-in boundaries: they build and load checkpoints, their kernels are next)
MIXTURE_MODULES = {"ES_MOE": ES_MOE, **MIXTURE_BOUNDARY_MODULES}
MIXTURE_REPEAT_MODULES = set(MIXTURE_BOUNDARY_REPEAT)
HEAD_MODULES = {"Detect": Detect, "Segment": Segment}


def make_divisible(x, divisor):
    """ultralytics/utils/ops.py:161-173."""
    return math.ceil(x / divisor) * divisor


def guess_model_scale(path) -> str:
    m = re.search(r"-([nslmx])(-[a-z0-9]+)?$", Path(path).stem)
    return m.group(1) if m else ""


def yaml_model_load(path):
    """Load a model YAML; ``yolo-master-s.yaml`` resolves to ``yolo-master.yaml`` + scale 's'."""
    p = Path(path)
    cands = [p, CFG_DIR / p.name]
    scale = guess_model_scale(p)
    if scale:
        unified = re.sub(r"-([nslmx])$", "", p.stem) + p.suffix
        cands += [p.with_name(unified), CFG_DIR / unified]
    for c in cands:
        if c.exists():
            d = yaml.safe_load(c.read_text())
            d["yaml_file"] = str(path)
            if scale and "scale" not in d:
                d["scale"] = scale
            return d
    raise FileNotFoundError(f"model yaml {path} not found (searched {[str(c) for c in cands]})")


def parse_model(d, ch, verbose=False):
    """YAML dict -> (nn.Sequential, save list); mirrors tasks.py:2022-2270 for the supported modules."""
    legacy = True
    max_channels = float("inf")
    nc, act, scales, end2end = (d.get(x) for x in ("nc", "activation", "scales", "end2end"))
    reg_max = d.get("reg_max", 16)
    depth, width = d.get("depth_multiple", 1.0), d.get("width_multiple", 1.0)
    scale = d.get("scale")
    if scales:
        if not scale:
            scale = next(iter(scales.keys()))
        depth, width, max_channels = scales[scale]
    if act:
        raise NotImplementedError("ymk parse_model: custom default activations are not supported (SiLU only)")
    ch = [ch]
    layers, save, c2 = [], [], ch[-1]
    from .mixture import SharedExpertMoE
    SharedExpertMoE.reset_shared_pools()      # expert pools are per model (nn/tasks.py:2037)
    for i, (f, n, m, args) in enumerate(d["backbone"] + d["head"]):
        name = m
        if m.startswith("nn."):
            mod = getattr(nn, m[3:])
        elif m in BASE_MODULES:
            mod = BASE_MODULES[m]
        elif m == "Concat":
            mod = Concat
        elif m in HEAD_MODULES:
            mod = HEAD_MODULES[m]
        elif m in MIXTURE_MODULES:
            mod = MIXTURE_MODULES[m]
        else:
            raise KeyError(f"unknown model module {m!r} (ymk builds the YOLO-Master detection module set)")
        args = list(args)
        for j, a in enumerate(args):
            if isinstance(a, str):
                with contextlib.suppress(ValueError):
                    args[j] = {"nc": nc}[a] if a == "nc" else ast.literal_eval(a)
        n = n_ = max(round(n * depth), 1) if n > 1 else n
        if mod in BASE_MODULES.values():
            c1, c2 = ch[f], args[0]
            if c2 != nc:
                c2 = make_divisible(min(c2, max_channels) * width, 8)
            args = [c1, c2, *args[1:]]
            if mod in REPEAT_MODULES:
                args.insert(2, n)
                n = 1
            if mod is C3k2:
                legacy = False
                if scale in "mlx":
                    args[3] = True
            if mod is A2C2f:
                legacy = False
                if scale in "lx":
                    args.extend((True, 1.2))
        elif mod in MIXTURE_MODULES.values():  # adapt_mixture_args (mixture_registry.py:143-156)
            c1, c2 = ch[f], args[0]
            if c2 != nc:
                c2 = make_divisible(min(c2, max_channels) * width, 8)
            args = [c1, c2, *args[1:]]
            if mod in MIXTURE_REPEAT_MODULES:   # the depth-scaled repeat becomes the module's internal n
                args.insert(2, n)
                n = 1
        elif mod is Concat:
            c2 = sum(ch[x] for x in f)
        elif mod in (Detect, Segment):   # tasks.py:2189-2236
            args.extend([reg_max, end2end, [ch[x] for x in f]])
            if mod is Segment:
                args[2] = make_divisible(min(args[2], max_channels) * width, 8)
            Detect.legacy = legacy
        else:
            c2 = ch[f]
        m_ = nn.Sequential(*(mod(*args) for _ in range(n))) if n > 1 else mod(*args)
        m_.np = sum(x.numel() for x in m_.parameters())
        m_.i, m_.f, m_.type = i, f, name
        if verbose:
            print(f"{i:>3}{f!s:>20}{n_:>3}{m_.np:10.0f}  {name:<20}{args!s:<30}")
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)
        layers.append(m_)
        if i == 0:
            ch = []
        ch.append(c2)
    SharedExpertMoE.reset_shared_pools()      # (nn/tasks.py:2272)
    mark_pooled_producers(layers)
    return nn.Sequential(*layers), sorted(save)


def mark_pooled_producers(layers):
    """An ES-MoE layer starts with a global average pool of its input (moe/routers.py:458-527).  Where that input is the previous YAML row's
    output and the row ends in a 1x1 convolution (C2f / C3k2 / A2C2f `cv2`, a plain Conv), the convolution is asked to leave the per-tile channel
    sums of what it stores (Conv.pool_out -> ops.conv2d(pool=True) -> ymk_conv1x1_pooled where the shape is taken): the router then reads
    B x chunks x C floats instead of the whole map."""
    for i, m in enumerate(layers):
        if isinstance(m, ES_MOE) and m.f == -1 and i > 0:
            prev = layers[i - 1]
            # the row's TAIL 1x1: cv3 for C3 / C3k (their cv2 is the parallel branch that fills half of the concat), cv2 for C2f / C3k2 /
            # A2C2f / SPPF / C2PSA; an A2C2f with a residual scale (gamma) multiplies its output into a new tensor afterwards, so the
            # sums of what cv2 stored would describe a tensor nobody reads — skipped
            if isinstance(prev, Conv):
                tail = prev
            elif type(prev).__name__ in ("C3", "C3k"):
                tail = getattr(prev, "cv3", None)
            elif type(prev).__name__ == "A2C2f" and getattr(prev, "gamma", None) is not None:
                tail = None
            else:
                tail = getattr(prev, "cv2", None)
            if isinstance(tail, Conv) and tail.conv.kernel_size == (1, 1) and tail.conv.groups == 1:
                tail.pool_out = True


class DetectionModel(nn.Module):
    """YOLO-Master detection model on the ymk path (reference surface: tasks.py:502-560)."""

    def __init__(self, cfg="yolo-master-s.yaml", ch=3, nc=None, verbose=False):
        super().__init__()
        self.yaml = cfg if isinstance(cfg, dict) else yaml_model_load(cfg)
        self.yaml = deepcopy(self.yaml)
        self.yaml["channels"] = ch
        if nc and nc != self.yaml["nc"]:
            self.yaml["nc"] = nc
        self.model, self.save = parse_model(deepcopy(self.yaml), ch=ch, verbose=verbose)
        self.names = {i: f"{i}" for i in range(self.yaml["nc"])}
        self.inplace = self.yaml.get("inplace", True)
        self.end2end = False
        m = self.model[-1]
        if isinstance(m, Detect):
            m.stride = torch.tensor(self._head_strides(), dtype=torch.float32)
            self.stride = m.stride
            m.bias_init()
        else:
            self.stride = torch.Tensor([32])
        # initialize_weights (ultralytics/utils/torch_utils.py:552-562): BN eps/momentum
        for mod in self.modules():
            if type(mod) is nn.BatchNorm2d:
                mod.eps = 1e-3
                mod.momentum = 0.03
        self.ymk_dtype = torch.float32
        self._flags = None
        self.fuse_concat = True  # Upsample+Concat feeding a C2f/C3k2 is read in place by its 1x1 cv1 (no concat buffer)

    # the reference finds strides with a 256x256 dry run (tasks.py:547-549); the graph is static, so
    # walk it symbolically instead (no forward pass needed, works without a GPU)
    def _head_strides(self):
        scale_of = []  # cumulative downsampling factor of every layer's output
        for m in self.model:
            f = m.f
            if isinstance(m, Detect):
                return [scale_of[j] for j in f]
            if isinstance(f, int):
                base = (scale_of[-1] if scale_of else 1.0) if f == -1 else scale_of[f]
            else:
                base = scale_of[-1] if f[0] == -1 else scale_of[f[0]]
            if isinstance(m, Conv):
                base = base * m.conv.stride[0]
            elif isinstance(m, nn.Upsample):
                base = base / float(m.scale_factor)
            scale_of.append(base)
        raise RuntimeError("no Detect head")

    # -- configuration -------------------------------------------------------------------
    def set_compute_dtype(self, dtype: torch.dtype):
        set_compute_dtype(self, dtype)
        self.ymk_dtype = dtype
        return self

    def fuse(self, verbose=False):
        """BN folding happens when weights are packed for libymk (same algebra as
        fuse_conv_and_bn, torch_utils.py:315-349); kept for API compatibility."""
        return self

    def is_fused(self, thresh=10):
        return True

    def repack(self):
        """Drop packed weights (call after loading / editing parameters)."""
        for m in self.modules():
            if isinstance(m, YmkModule):
                m.clear_pack()
        return self

    def load_state_dict(self, state_dict, strict=True, assign=False):
        r = super().load_state_dict(state_dict, strict=strict, assign=assign)
        self.repack()
        return r

    # -- forward ---------------------------------------------------------------------------
    def forward(self, x, *args, **kwargs):
        if isinstance(x, dict):
            raise RuntimeError("ymk DetectionModel implements inference only (loss/training: use the reference)")
        return self.predict(x, *args, **kwargs)

    def predict(self, x, profile=False, visualize=False, augment=False, embed=None):
        if augment or visualize or embed:
            raise NotImplementedError("ymk DetectionModel.predict: augment/visualize/embed are not on the hot path")
        return self._predict_once(x)

    def _stem_pair_ok(self):
        """Rows 0 and 1 are `Conv(3, C0, 3, 2)` -> `Conv(C0, C1, 3, 2)`, SiLU both, nothing else reads row 0, and libymk has the
        fused kernel for these widths in the compute dtype."""
        if len(self.model) < 3:
            return False
        m0, m1 = self.model[0], self.model[1]
        if not (type(m0) is Conv and type(m1) is Conv and m1.f == -1 and 0 not in self.save):
            return False
        c0, c1 = m0.conv, m1.conv
        if not (c0.in_channels <= 4 and _is_silu(m0.act) and _is_silu(m1.act) and c1.groups == 1 and c0.kernel_size[0] == c0.kernel_size[1]
                and c1.kernel_size[0] == c1.kernel_size[1]):
            return False
        return ops.stem_pair_supported(m1.ymk_dtype, c0.in_channels, c0.out_channels, c1.out_channels, c0.kernel_size[0], c0.stride[0],
                                       c1.kernel_size[0], c1.stride[0])

    def _predict_once(self, x, taps=None):
        """Graph walk on NHWC buffers (reference loop: tasks.py:182-218).  x: NCHW fp32 [B,3,H,W] on GPU.
        Returns (y [B,4+nc,A] fp32, preds dict) like Detect in eval mode.  ``taps``: optional dict that
        receives every layer's NHWC output (parity tests)."""
        if self.training:
            raise RuntimeError("ymk DetectionModel implements eval-mode inference only; call .eval()")
        ops.require_gpu(x, "yolo_master_amd models")
        if self._flags is None or self._flags.device != x.device:
            self._flags = torch.zeros((1,), dtype=torch.int32, device=x.device)
        ys = []
        cur = x
        raw = det_in = None
        skip = -1
        # Detect levels start as soon as their input map exists (Detect.start_level: a side HIP stream per level, joined at the head)
        det = self.model[-1]
        early = None
        if type(det) is Detect and det.early_levels and taps is None and x.is_cuda and isinstance(det.f, (list, tuple)) and len(det.f) > 1:
            sizes = []
            for s_ in det.stride.tolist():
                h, w = int(x.shape[2]), int(x.shape[3])
                for _ in range(int(round(__import__("math").log2(s_)))):
                    h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1      # a 3x3 / stride-2 / pad-1 convolution per octave
                sizes.append((h, w))
            early = {"state": det.begin(x.shape[0], sizes, x.device), "levels": {int(j) % len(self.model): i for i, j in enumerate(det.f)}}
        for m in self.model:
            timed = ops.TIMER.on and m.i != skip
            if ops.TIMER.on:     # per-layer attribution of the op timer (bench.py `roofline_layers`)
                ops.TIMER.layer = m.i
            if m.i == skip:      # produced together with the previous layer (fused stem pair)
                ys.append(cur if m.i in self.save else None)
                continue
            if m.f != -1:
                cur = ys[m.f] if isinstance(m.f, int) else [cur if j == -1 else ys[j] for j in m.f]
            if timed:
                n_in = _logical_elems(cur)
            if m.i == 0 and taps is None and self._stem_pair_ok():
                # rows 0 + 1 as one kernel: the stem map (the largest tensor of the network) never leaves the CU (csrc/stem2.hip)
                m1 = self.model[1]
                p0, p1 = m._packed(x.device), m1._packed(x.device)
                cur = ops.stem_pair(cur, p0["wt"], p0["b"], p1["w"], p1["b"])
                if timed:   # SURVEY 8(d) counts per YAML layer: row 0 writes and row 1 reads the stem map although the fused kernel keeps it on chip
                    stem = cur.shape[0] * (2 * cur.shape[1]) * (2 * cur.shape[2]) * m.conv.out_channels
                    ops.TIMER.io[0] = (n_in, stem)
                    ops.TIMER.io[1] = (stem, _logical_elems(cur))
                ys.append(None)
                skip = 1
                continue
            if isinstance(m, Conv):
                if isinstance(cur, LazyUpsample):
                    cur = cur.materialise()
                cur = m._run_stem(cur) if m.i == 0 and m.conv.in_channels <= 4 else m._run(cur)
            elif isinstance(m, nn.Upsample):
                if m.mode != "nearest" or float(m.scale_factor) != 2.0:
                    raise NotImplementedError("ymk: nn.Upsample must be nearest x2")
                cur = LazyUpsample(cur)
            elif isinstance(m, Concat):
                nxt = self.model[m.i + 1] if m.i + 1 < len(self.model) else None
                fusable = (self.fuse_concat and len(cur) == 2 and m.i not in self.save and isinstance(nxt, C2f)
                           and nxt.f == -1 and nxt.cv1.conv.kernel_size == (1, 1) and m.d == 1
                           and torch.is_tensor(cur[1])
                           and (cur[0].src if isinstance(cur[0], LazyUpsample) else cur[0]).shape[-1] % 8 == 0)
                cur = VirtualCat(cur) if fusable else m._run(cur)
            elif isinstance(m, ES_MOE):
                m.bind_flags(self._flags)
                cur = m._run(cur)
            elif isinstance(m, Detect):
                det_in = cur
                cur, raw = m.finish(early["state"], cur) if early is not None else m._run(cur)
            elif isinstance(m, nn.Sequential):
                for mm in m:
                    cur = mm._run(cur)
            else:
                if isinstance(cur, LazyUpsample):
                    cur = cur.materialise()
                cur = m._run(cur)
            if timed:
                ops.TIMER.io[m.i] = (n_in, _logical_elems(cur))
            ys.append(cur if m.i in self.save else None)
            if early is not None and m.i in early["levels"] and m.i != len(self.model) - 2 and torch.is_tensor(cur):
                det.start_level(early["state"], early["levels"][m.i], cur)   # (the last level's map is the head's direct input: it runs on the main stream)
            if taps is not None:
                taps[m.i] = cur.materialise() if isinstance(cur, VirtualCat) else cur
        ops.TIMER.layer = -1
        det = self.model[-1]
        preds = DetectPreds(raw, det_in, det.reg_max, det.nc, raw_fn=lambda: det.raw_logits(det_in))
        if isinstance(det, Segment):   # mask coefficients fp32 [B, nm, A] and prototypes NHWC [B, 2H0, 2W0, nm]
            preds["mask_coefficient"], preds["proto"] = det.last_mc, det.last_proto
        return cur, preds

    def check_flags(self):
        """One host sync per batch: raise MoERouterError if any routed layer saw NaN/Inf."""
        for m in self.model:
            if isinstance(m, ES_MOE):
                m.check_flags()
                break


def _logical_elems(v) -> int:
    """Elements of a layer input / output as the reference's graph sees it (SURVEY 8(d) "layer-fused ideal": each YAML layer reads its
    inputs once and writes its output once): lazy upsamples count at their upsampled size, virtual concatenations as the sum of their parts."""
    if torch.is_tensor(v):
        return int(v.numel())
    if isinstance(v, LazyUpsample):
        return 4 * _logical_elems(v.src)
    if isinstance(v, VirtualCat):
        return sum(_logical_elems(p) for p in v.parts)
    if isinstance(v, (list, tuple)):
        return sum(_logical_elems(p) for p in v)
    return 0


class SegmentationModel(DetectionModel):
    """YOLO-Master segmentation model on the ymk path (reference surface: nn/tasks.py:696-727).  `predict` returns the
    reference's eval structure ((y with the mask-coefficient rows, prototypes NCHW fp32), preds); the device path for a
    predictor is `_predict_once` + yolo_master_amd.postprocess."""

    def __init__(self, cfg="yolo-master-seg-s.yaml", ch=3, nc=None, verbose=False):
        super().__init__(cfg, ch, nc, verbose)
        if not isinstance(self.model[-1], Segment):
            raise ValueError("SegmentationModel needs a model YAML that ends in a Segment head")

    def predict(self, x, profile=False, visualize=False, augment=False, embed=None):
        y, preds = super().predict(x, profile, visualize, augment, embed)
        proto = ops.nhwc_to_nchw_f32(preds["proto"])
        return (torch.cat([y, preds["mask_coefficient"]], 1), proto), preds   # concatenation: API compatibility only
