#!/bin/bash
# split heuristic of the weight-gradient GEMMs: work-model bias per tile (SNSDE_WGRAD_BIAS) x total workgroups (SNSDE_WGRAD_WGS)
for b in 2 4 8; do for w in 256 512 768; do echo -n "bias=$b wgs=$w: "; SNSDE_WGRAD_BIAS=$b SNSDE_WGRAD_WGS=$w python $GRAFT_REPO_ROOT/tools/time_train.py 2>/dev/null | grep -E "native snsde_param" ; done; done
