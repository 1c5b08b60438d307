#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/micro/calls_ab.sh "moe_dw|moe_pw" 3 rev=tools/micro/_dwab/libymk_dwrev.so
