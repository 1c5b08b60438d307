#!/bin/bash
# Build container only: put the reference's Python package beside the tree for ONE gpurun call, so that the GPU box (which has no
# /root/reference) can run the REAL reference — hooked (tests/test_gpu_dropin_reference.py) and un-hooked (its eager PyTorch-ROCm time,
# its CPU time on the GPU box's host: tools/gpu_reference_timing.py).
#
#   tools/stage_reference.sh            copy /root/reference/ultralytics -> .refstage/ultralytics (git-ignored, travels with gpurun)
#   tools/stage_reference.sh clean      remove it again (ALWAYS do this after the call: reference sources do not live in this repo)
#
# .refstage/ is listed in .gitignore and must never be committed; nothing under yolo_master_amd/ reads it.
set -e
R="$(cd "$(dirname "$0")/.." && pwd)"
if [ "$1" = "clean" ]; then
  rm -rf "$R/.refstage"
  echo "[stage_reference] removed $R/.refstage"
  exit 0
fi
SRC="${YMK_REFERENCE_SRC:-/root/reference}"
[ -d "$SRC/ultralytics" ] || { echo "[stage_reference] no reference checkout at $SRC" >&2; exit 1; }
mkdir -p "$R/.refstage"
rsync -a --delete --exclude '__pycache__' --exclude '*.pyc' --exclude 'assets' "$SRC/ultralytics" "$R/.refstage/" 2>/dev/null || {
  rm -rf "$R/.refstage/ultralytics"; cp -r "$SRC/ultralytics" "$R/.refstage/ultralytics"; find "$R/.refstage" -name __pycache__ -type d -prune -exec rm -rf {} +; }
du -sh "$R/.refstage" | sed 's/^/[stage_reference] staged: /'
