#!/bin/bash
# Dev experiment: phase time stamps (s_memtime, 100 MHz) of gpt_mlp_fwd_kernel, workgroup 0.  Builds an instrumented copy of the
# library beside the product one (mmfn_amd/lib/libmmfn_hip_stamps.so) - run the build part here, the python part on the GPU box.
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/../.. && pwd)}
L=$R/mmfn_amd/lib
if [ "$1" = "build" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DMMFN_GPT_STAMPS -c $R/mmfn_amd/csrc/gpt_block.hip -o $L/gpt_block_stamps.o.x || exit 1
  objs=$(ls $L/*.o | grep -v gpt_block.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $L/libmmfn_hip_stamps.so $objs $L/gpt_block_stamps.o.x
  exit $?
fi
MMFN_HIP_LIB=$L/libmmfn_hip_stamps.so python - <<'PY'
import ctypes, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
import torch
from mmfn_amd import ops, _lib
import test_gpt_block_gpu as tg
dev = torch.device("cuda:0")
B, T, NH = 32, 192, 4
rng = torch.tensor([11, 3], dtype=torch.int64, device=dev)
names = ["start", "o staged", "proj mfma", "proj epi", "barrier", "ln2", "barrier", "fc1 mfma", "fc1 epi", "barrier", "fc2 mfma", "fc2 epi"]
for C in (64, 128):
    g = torch.Generator().manual_seed(C)
    p = tg._params(C, g, dev)
    x = torch.randn(B * T, C, generator=g).to(dev)
    out = tg._bufs(B, T, C, NH, dev)
    d = ops.gpt_block_desc(B, T, C, NH, attn_pdrop=0.1, resid_pdrop=0.1, rng_state=rng, rng_stream=40, x=x, **p, **out)
    ops.gpt_block_attn_fwd(d)
    for _ in range(5):
        ops.gpt_block_mlp_fwd(d)
    torch.cuda.synchronize()
    buf = (ctypes.c_int64 * 64)()
    assert _lib.lib().mmfn_gpt_debug_read(buf) == 0
    for wv, off in ((0, 0), (7, 32)):
        t = [buf[off + i] for i in range(12)]
        print("C=%d wave %d: " % (C, wv) + "  ".join("%s +%.2f" % (names[i], (t[i] - t[i - 1]) / 100.0) for i in range(1, 12)) + "  | total %.2f us" % ((t[11] - t[0]) / 100.0))
PY
