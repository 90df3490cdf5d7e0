import sys, gc
sys.path.insert(0, ".")
import numpy as np, torch
import claxon_amd as cx
def use():
    a = torch.from_numpy(np.zeros(1 << 20, dtype=np.uint8)).to("cuda:0")
    torch.cuda.synchronize()
    return int(a.sum().item())
for second_wait in (0, 120):
    c1 = cx.Context(0, wait_s=120)
    print("ctx1 ok", use())
    del c1; gc.collect()
    print("after destroy", end=" ")
    try:
        print(use())
    except Exception as e:
        print("FAIL", str(e).splitlines()[0])
    c2 = cx.Context(0, wait_s=second_wait)
    try:
        print("ctx2 wait=%d ok" % second_wait, use())
    except Exception as e:
        print("ctx2 wait=%d FAIL" % second_wait, str(e).splitlines()[0])
    del c2; gc.collect()
