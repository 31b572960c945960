import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "shim")]
from gaussiangrasper_amd import _lib
_lib.LIB_PATH = os.path.abspath(sys.argv[1]); lib = _lib.load(build_if_missing=False)
from gaussiangrasper_amd.mlp import MLP
dev="cuda:0"; torch.manual_seed(0)
for cin,(h,w) in ((32,(1200,1600)),(128,(1080,1920))):
    m=MLP(cin,512,[128]).to(dev); img=torch.randn(h,w,cin,device=dev)
    with torch.no_grad():
        for ph in ("warm","timed"):
            lib.gg_prof_reset(); lib.gg_prof_enable(1)
            for _ in range(5): y=m(img)
            torch.cuda.synchronize()
        lib.gg_prof_enable(0)
    nn, ms = ctypes.c_int(0), ctypes.c_double(0.0); lib.gg_prof_get(8, ctypes.byref(nn), ctypes.byref(ms))
    print(sys.argv[1], cin, round(ms.value/nn.value,4), "ms")
    del y, img
# a plain fill of the output's size
y=torch.empty(1200*1600,512,device=dev); torch.cuda.synchronize()
e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
y.zero_(); e0.record()
for _ in range(5): y.zero_()
e1.record(); torch.cuda.synchronize(); print("fill 3.93 GB", round(e0.elapsed_time(e1)/5,4), "ms")
