"""Times the bf16x3 GEMM (l3ac_gemm_split_f32, bias epilogue) on the C = 512 stage's shapes and prints a digest of every output, so that two
builds / two settings of an environment switch can be compared on one box: `L3AC_GEMM_W256=0 python tools/hc_time.py` against `=1`."""
import hashlib
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
from l3ac_amd import _capi
lib=_capi.load_library()
s=torch.cuda.current_stream().cuda_stream
shapes=((24480,2048,512),(24480,512,2048),(21600,2048,512),(21600,512,2048),(46080,256,512),(23040,2048,512),(23040,512,2048),(46080,2048,512),(46080,512,2048))
if len(sys.argv) > 1:
    shapes=tuple(tuple(int(v) for v in a.split('x')) for a in sys.argv[1:])
for m,n,k in shapes:
    g=torch.Generator(device="cuda").manual_seed(m+n+k)
    a=torch.randn(m,k,device="cuda",generator=g); w=torch.randn(n,k,device="cuda",generator=g); bias=torch.randn(n,device="cuda",generator=g); c=torch.empty(m,n,device="cuda")
    img=torch.empty(lib.l3ac_gemm_split_image_bytes(n,k),dtype=torch.uint8,device="cuda")
    _capi.check(lib.l3ac_gemm_split_image(w.data_ptr(),n,k,img.data_ptr(),s))
    f=lambda: _capi.check(lib.l3ac_gemm_split_f32(a.data_ptr(),k,img.data_ptr(),bias.data_ptr(),c.data_ptr(),n,m,n,k,s))
    for _ in range(10): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    res=[]
    for r in range(3):
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1)/20)
    digest=hashlib.sha256(c.cpu().numpy().tobytes()).hexdigest()[:12]
    print(m,n,k,' '.join('%.4f'%x for x in res),'ms  %.1f TFLOP/s'%(2.0*m*n*k/min(res)/1e9),'digest',digest, flush=True)
