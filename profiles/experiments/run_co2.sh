#!/bin/bash
cd $GRAFT_REPO_ROOT
for V in "JAF_WGRAD_PARTIALS=1" "JAF_WGRAD_PARTIALS=0" "JAF_WGRAD_PARTIALS=1" "JAF_WGRAD_PARTIALS=0" "JAF_WGRAD_PARTIALS=1" "JAF_WGRAD_PARTIALS=0"; do bash profiles/experiments/ab_w.sh part_${V##*=} $V 2>&1 | head -3; done
