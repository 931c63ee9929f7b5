"""Time the streaming MT19937 generator alone (no sampler beside it)."""
import sys, torch
sys.path.insert(0, '.')
from bliss_gnn_amd import _lib
dev = torch.device('cuda:0')
cap = int(sys.argv[1]) if len(sys.argv) > 1 else 507000
state = torch.zeros(626, dtype=torch.int32, device=dev)
st = torch.get_rng_state().numpy()
import numpy as np
state[:624] = torch.from_numpy(st[24:24 + 624 * 8].view(np.uint64).astype(np.uint32).view(np.int32)).to(dev)
state[624] = 1; state[625] = 624
ctl = torch.zeros(16, dtype=torch.int32, device=dev)
out = torch.empty(cap + 2 * 624, dtype=torch.float32, device=dev)
raw = torch.empty(624 * (cap // 624 + 3), dtype=torch.int32, device=dev)
err = torch.zeros(1, dtype=torch.int32, device=dev)
s = torch.cuda.current_stream().cuda_stream
for rep in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    _lib.check(_lib.lib.bliss_rng_stream_begin(state.data_ptr(), ctl.data_ptr(), out.data_ptr(), raw.data_ptr(), cap, s), "begin")
    _lib.check(_lib.lib.bliss_rng_stream_end(state.clone().data_ptr(), ctl.data_ptr(), raw.data_ptr(), cap, err.data_ptr(), s), "end")
    b.record()
    torch.cuda.synchronize()
    print("cap %d: %.1f us  (%.3f us / 624-block)  progress %d" % (cap, 1e3 * a.elapsed_time(b), 1e3 * a.elapsed_time(b) / (cap / 624), int(ctl[0])))
torch.manual_seed(0); 
