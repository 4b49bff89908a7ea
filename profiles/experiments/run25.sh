#!/bin/bash
cd $GRAFT_REPO_ROOT
for xf in 0 4096 384 512 1024 0 4096; do
  echo "=== JAF_CD_X=$xf"
  JAFPRO_HIP_LIB=$PWD/scratch/x/lib_dmax.so JAF_CD_X=$xf python scratch/layer_table.py 2>/dev/null > gpurun_out/x25_layers_$xf.txt
  grep -E " G24 " gpurun_out/x25_layers_$xf.txt | grep -E "conv_dma_kernel<., 4, false, true" | head -6
done
