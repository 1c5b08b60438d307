#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
bash tools/micro/calls_ab.sh "moe_pw" 2 ahead2=tools/micro/_dwab/libymk_pwl2.so
