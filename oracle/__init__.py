"""ORACLE — test infrastructure only.

CPU restatements of the reference's algorithms for the detection hot path (model_ref: torch
fp32 functional forward; nms_ref: numpy NMS / CW-NMS; moa_ref / mot_ref / gated_ref: the Mixture-of-Attention,
Mixture-of-Transformer and gated-MoE blocks of config 5, the next rows of the path (model_ref.forward walks that
YAML too); post_ref: scale_boxes / clip_boxes, the step after the path; refboot: import shim for the real
reference, usable only where /root/reference exists).  Nothing under yolo_master_amd/ may
import this package; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
Parity status: pinned against golden vectors generated from the real reference
(tests/golden/make_golden.py, make_golden_moa.py, make_golden_mot.py, make_golden_gated.py, make_golden_cfg5.py, make_golden_post.py) for the forward pass, routing
decisions, greedy NMS and the MoA/MoT blocks;
CW-NMS has no Python implementation in the reference: its C++ edge demo is compiled IN PLACE
(oracle/cwref/build.py -> oracle/_ref/libcwref.so, against a small OpenCV geometry shim) and pins nms_ref.cw_refine
(tests/golden/make_golden_cw.py, tests/test_oracle_cw.py).
"""
