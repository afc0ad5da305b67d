#!/bin/bash
# experiment build of attention_wg.hip with phase stamps, then tools/experiments/attn_phases.py (run through gpurun)
R=$GRAFT_REPO_ROOT; B=$R/tools/experiments/build; mkdir -p $B
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DMMFN_ATTN_STAMPS -c $R/mmfn_amd/csrc/attention_wg.hip -o $B/attention_wg_st.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $B/libmmfn_hip_st.so $B/attention_wg_st.o $(ls $R/mmfn_amd/lib/*.o | grep -v attention_wg.o)
MMFN_HIP_LIB=$B/libmmfn_hip_st.so python $R/tools/experiments/attn_phases.py
