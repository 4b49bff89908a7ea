#!/bin/bash
# experimental library: scratch/xbuild.sh TAG file.hip[,file2.hip] "sed-expr" [extra flags...]
# compiles patched copies of the named sources (sed applied) and links them with the product objects of all others
set -e
TAG=$1; FILES=$2; SED=$3; shift 3
cd /root/repo
OBJS=""
for f in conv conv_bf16 conv_dma wgrad wgrad_bf16 wgrad_dma elementwise norm resample gather raster raster_bwd raster_texture linear ubench input_pipeline metrics; do
  if [[ ",$FILES," == *",$f.hip,"* ]]; then
    if [[ -f "$SED" ]]; then python "$SED" < jafpro_amd/csrc/$f.hip; else sed -e "$SED" jafpro_amd/csrc/$f.hip; fi > jafpro_amd/csrc/_x_${TAG}_$f.hip
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wno-unused-result -Xclang -target-feature -Xclang -packed-fp32-ops "$@" -c jafpro_amd/csrc/_x_${TAG}_$f.hip -o scratch/x/${TAG}_$f.o 2>&1 | grep -v "packed-fp32-ops" || true
    rm -f jafpro_amd/csrc/_x_${TAG}_$f.hip
    OBJS="$OBJS scratch/x/${TAG}_$f.o"
  else
    OBJS="$OBJS jafpro_amd/csrc/_obj/$f.o"
  fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scratch/x/lib_${TAG}.so $OBJS
ls -la scratch/x/lib_${TAG}.so
