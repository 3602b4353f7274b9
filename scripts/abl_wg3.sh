#!/bin/bash
# Time decomposition of conv_ps_wgrad3_kernel by COMPILE-TIME ablation (BD_WG3_ABL bits, conv_ps.hip): one libbd_hip.so per variant, linked from the
# objects of the normal build + a re-compiled conv_ps.o, each timed on the three main shapes of the CIFAR step (scripts/abl_wg3.py).
#   here (no GPU):  scripts/abl_wg3.sh build      -> .abl/libs/libbd_abl<N>.so
#   on the box:     scripts/abl_wg3.sh run        -> one line per variant
set -e
cd "$(dirname "$0")/.."
VARIANTS="0 1 2 4 16 32 64 96 6 7 23"
if [ "$1" = build ]; then
  python -m baddiffusion_amd.build > /dev/null
  mkdir -p .abl/libs /tmp/abl_wg3
  TL=$(python -c "import torch,os;print(os.path.join(os.path.dirname(torch.__file__),'lib'))")
  for a in $VARIANTS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DBD_WG3_ABL=$a -c baddiffusion_amd/csrc/conv_ps.hip -o /tmp/abl_wg3/conv_ps_$a.o 2> /dev/null &
  done; wait
  for a in $VARIANTS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o .abl/libs/libbd_abl$a.so $(ls baddiffusion_amd/csrc/obj/*.o | grep -v /conv_ps.o) /tmp/abl_wg3/conv_ps_$a.o -L$TL -lamdhip64
  done
  ls .abl/libs
else
  keep=$(mktemp); cp baddiffusion_amd/libbd_hip.so $keep
  for a in $VARIANTS; do
    cp .abl/libs/libbd_abl$a.so baddiffusion_amd/libbd_hip.so
    BD_WG3_ABL=$a PYTHONPATH=$PWD timeout 100 python scripts/abl_wg3.py 2>&1 | grep -v amdgpu.ids
  done
  cp $keep baddiffusion_amd/libbd_hip.so
fi
