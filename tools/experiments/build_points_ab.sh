#!/bin/bash
# Ablation build: libmmfn_hip with the F(4x4,3x3) Winograd transforms over Lavin & Gray's points 0, +-1, +-2 (rounds 1-3) instead
# of 0, +-3/4, +-3/2 -> mmfn_amd/lib/exp/libmmfn_hip_lavin.so; select it with MMFN_HIP_LIB=<path> (tools/grad_cosine.py, bench.py).
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/mmfn_amd/lib/exp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DMMFN_WINO_A=1.0f -DMMFN_WINO_B=2.0f \
  -c $R/mmfn_amd/csrc/winograd.hip -o $R/mmfn_amd/lib/exp/winograd_lavin.o
OBJS=$(ls $R/mmfn_amd/lib/*.o | grep -v winograd.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/mmfn_amd/lib/exp/libmmfn_hip_lavin.so $OBJS $R/mmfn_amd/lib/exp/winograd_lavin.o
ls -la $R/mmfn_amd/lib/exp/
