#!/bin/bash
# experiment batch 2: which part of the epilogue bounds the ConvLSTM forward and the fused-dz data-gradient kernels
cd $GRAFT_REPO_ROOT
for xf in 0 16 32 64 48 112 128 256 384 512 1024 1920; do
  echo "=== JAF_CD_X=$xf"
  JAFPRO_HIP_LIB=$PWD/scratch/x/lib_dmax.so JAF_CD_X=$xf python scratch/layer_table.py 2>/dev/null > gpurun_out/x2_layers_$xf.txt
  grep -E " G24 " gpurun_out/x2_layers_$xf.txt | grep -E "conv_dma_kernel<3, 4, true|conv_dma_kernel<., 4, false, true" | head -12
done
