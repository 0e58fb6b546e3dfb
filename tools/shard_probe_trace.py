"""One middle rank of 8 (separator mode, loop-back transport) at C4: a few LM trials, for a kernel trace (tools/shard_probe_trace.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cube_slam_wu_amd import capi, synth_ba
R = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pr = synth_ba.make_problem(n_cams=1000, n_points=200000, n_cuboids=500, seed=42)
P = capi.ba_from_dict(pr)
P.stage_timing(True)
if R > 1:
    P.set_shard(R // 2, R)
    si = P.shard_info()
    msg = 3 * si["w_max"] ** 2 + 2 * si["w_max"]
    def loopback(ptr, n, on_device, op):
        if on_device and n == msg * R:
            t = torch.as_tensor(capi.DeviceDoubles(ptr, n), device="cuda").view(R, msg)
            t.copy_(t[R // 2].clone().expand(R, msg)); torch.cuda.synchronize()
        return 0
    P.optimize_sharded(1, loopback); P.optimize_sharded(3, loopback)
else:
    P.optimize(1); P.optimize(3)
print(P.timing())
