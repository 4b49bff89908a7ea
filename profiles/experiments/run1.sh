#!/bin/bash
# experiment batch 1: main chain without weight gradients; which phase bounds the part-network conv kernels
cd $GRAFT_REPO_ROOT
bash scratch/ab.sh x1_base
JAFPRO_HIP_LIB=$PWD/scratch/x/lib_nowg.so JAF_X_SKIP_WGRAD=1 bash scratch/ab.sh x1_nowg
JAF_WGRAD_STREAM=0 bash scratch/ab.sh x1_wg_main
for xf in 0 1 2 4 8 3 12; do
  echo "=== JAF_CD_X=$xf"
  JAFPRO_HIP_LIB=$PWD/scratch/x/lib_dmax.so JAF_CD_X=$xf python scratch/layer_table.py 2>/dev/null > gpurun_out/x1_layers_$xf.txt
  grep -E "total timed| G24 " gpurun_out/x1_layers_$xf.txt | grep -E "total|conv_dma" | head -24
done
