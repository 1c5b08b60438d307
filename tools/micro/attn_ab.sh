#!/bin/bash
# (round-6 record; the YMK_ATTN_NQ=2 lines need the parked two-query-tile form: tools/micro/parked/attn_nq2.hip.txt)
# GPU box: resident area attention — two query tiles per pass (YMK_ATTN_NQ=2) and the stage ablation of the kernel (tools/micro/_dwab/libymk_at*.so,
# built by `bash tools/micro/attn_ab.sh build` on the build host: AT_ABLATE bits 1 K reads, 2 V reads, 4 exp2, 8 score MFMAs, 16 P V MFMAs)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
if [ "$1" == "build" ]; then
  for a in 1 2 3 4 8 16 24 31; do bash tools/micro/lib_variant.sh at$a attn.hip -DAT_ABLATE=$a > /dev/null; done
  bash tools/micro/lib_variant.sh atnp4 attn.hip -DAT_NQ2_NP=4 > /dev/null
  bash tools/micro/lib_variant.sh atnp2 attn.hip -DAT_NQ2_NP=2 > /dev/null
  ls tools/micro/_dwab/ | grep libymk_at; exit 0
fi
for r in 1 2; do
  python tools/micro/attn_ab.py yolo_master_amd/libymk.so "tree NQ=1"
  YMK_ATTN_NQ=2 python tools/micro/attn_ab.py yolo_master_amd/libymk.so "tree NQ=2 (np 3)"
  YMK_ATTN_NQ=2 python tools/micro/attn_ab.py tools/micro/_dwab/libymk_atnp4.so "NQ=2 np 4"
  YMK_ATTN_NQ=2 python tools/micro/attn_ab.py tools/micro/_dwab/libymk_atnp2.so "NQ=2 np 2"
  YMK_ATTN_WAVES=6 python tools/micro/attn_ab.py yolo_master_amd/libymk.so "six waves NQ=1"
done
for a in 1 2 3 4 8 16 24 31; do
  python tools/micro/attn_ab.py tools/micro/_dwab/libymk_at$a.so "ablate $a NQ=1"
  YMK_ATTN_NQ=2 python tools/micro/attn_ab.py tools/micro/_dwab/libymk_at$a.so "ablate $a NQ=2"
done
