#!/bin/bash
# Ablation / timeline builds of the conv GEMM kernels: build_abl/a<N>/libpsg_hip.so with -DPSG_ABL=<N> (conv_gemm_kernel.h:
# bit 0 one K step only, bit 1 no epilogue, bit 3 per-workgroup phase timestamps).  Run a tool against one with
#   PSG_LIB_PATH=$PWD/build_abl/a8/libpsg_hip.so python tools/conv_timeline.py
# Needs the regular build first (the other objects are linked from pokemon_sprite_generator_amd/csrc/*.o).
#   tools/build_abl.sh 8        tools/build_abl.sh 1 2 3
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/pokemon_sprite_generator_amd/csrc
FL="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off"
for A in "$@"; do
  D=$ROOT/build_abl/a$A; mkdir -p $D
  for t in 128_128 128_64 64_64 128_160 64_160; do
    bm=${t%_*}; bn=${t#*_}
    /opt/rocm/bin/hipcc $FL -DPSG_ABL=$A -DPSG_TILE_BF16=1 -DPSG_TILE_BM=$bm -DPSG_TILE_BN=$bn -c conv_tile.hip -o $D/conv_tile_bf16_$t.o &
  done
  /opt/rocm/bin/hipcc $FL -DPSG_ABL=$A -c conv_gemm.hip -o $D/conv_gemm.o &
  wait
  OBJS=""
  for o in *.o; do if [ -f $D/$o ]; then OBJS="$OBJS $D/$o"; else OBJS="$OBJS $o"; fi; done
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libpsg_hip.so $OBJS
  rm -f $D/*.o
  ls -la $D/libpsg_hip.so
done
