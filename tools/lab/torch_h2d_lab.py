#!/usr/bin/env python3
"""lab: does torch.as_tensor(numpy view, device="cuda") in a worker thread hand back a tensor whose data is on the device?
Reader: a clone on a side stream after torch.cuda.synchronize() (not ordered with the default stream otherwise)."""
import sys, threading
import numpy as np, torch
T = int(sys.argv[1]) if len(sys.argv) > 1 else 3
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
junk = [torch.randint(0, 300, ((512 << 20) // 4,), dtype=torch.int32, device="cuda")]
torch.cuda.synchronize(); del junk; torch.cuda.empty_cache()
bad = [0] * T
def work(r):
    torch.cuda.set_device(0)
    side = torch.cuda.Stream()
    rng = np.random.default_rng(r)
    for it in range(iters):
        xyz = rng.random((229, 3)) * 20 + 1.0
        ts = [torch.as_tensor(np.asarray(xyz[:, k], np.float64), dtype=torch.float64, device="cuda") for k in range(3)]
        z = [torch.zeros(229, dtype=torch.float64, device="cuda") for _ in range(4)]
        torch.cuda.synchronize()
        with torch.cuda.stream(side):
            got = [t.clone() for t in ts]
        side.synchronize()
        for k in range(3):
            if not np.array_equal(got[k].cpu().numpy(), xyz[:, k]):
                bad[r] += 1
                if bad[r] < 3:
                    print("thread %d iter %d: coordinate %d read on the side stream: %r..., uploaded %r..." % (r, it, k, got[k][:3].tolist(), xyz[:3, k].tolist()), flush=True)
        del ts, z, got
        if it % 50 == 0:
            torch.cuda.empty_cache()
th = [threading.Thread(target=work, args=(r,)) for r in range(T)]
[t.start() for t in th]; [t.join() for t in th]
print("%d threads x %d uploads x 3: %d read back wrong on a side stream after a device synchronise" % (T, iters, sum(bad)))
