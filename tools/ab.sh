#!/bin/bash
# A/B of two builds of the library on the GPU box: tools/ab.sh "<command>" runs it with the in-tree library and with
# relpose-gnn_amd/lib/ab/libold.so (RPG_HIP_LIB override of relpose_gnn_amd/_lib.py).
R=${GRAFT_REPO_ROOT:-$(pwd)}
echo "== new"; eval "$1"
echo "== old"; RPG_HIP_LIB=$R/relpose-gnn_amd/lib/ab/libold.so eval "$1"
