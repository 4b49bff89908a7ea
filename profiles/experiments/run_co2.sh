#!/bin/bash
cd $GRAFT_REPO_ROOT
for V in "JAF_WGRAD_WC_CAP=2" "JAF_WGRAD_WC_CAP=4" "JAF_WGRAD_WC_CAP=2" "JAF_WGRAD_WC_CAP=4"; do bash profiles/experiments/ab_w.sh wc_$V $V 2>&1 | head -8; done
