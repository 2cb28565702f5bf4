"""Times the direct kernel variants (env-selected) on one GPU.  Dev tool, not part of the product."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import os as _os
_os.environ.setdefault("NBODY_HIP_LIBRARY", "lab")  # tools switch kernel variants: the laboratory build (csrc/env.h)
import nbody_simulation_amd as nb  # noqa: E402

C = nb._capi


def run(n, n_tgt, cfgs, reps=3, uniform=1.0):
    import torch
    dev = torch.device("cuda:0")
    pos, vel, w = nb.scenes.plummer(n, seed=1)
    tp = torch.from_numpy(pos).to(dev)
    tm = torch.from_numpy(w.astype(np.float32)).to(dev)
    tv = torch.from_numpy(vel[:n_tgt].copy()).to(dev)
    out = torch.empty((n_tgt, 2), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    for cfg in cfgs:
        for k, v in cfg.items():
            os.environ[k] = str(v)
        ws_bytes = C.direct_workspace_bytes(n, n_tgt)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        t = C.Timer()
        for r in range(reps + 1):
            C.direct_step_dev(stream, n, tp.data_ptr(), tm.data_ptr(), 0, n_tgt, tv.data_ptr(), out.data_ptr(), None,
                              0.1, 0.001, C.ARITH_FAST, ws.data_ptr(), ws_bytes, t, uniform_mass=uniform)
            torch.cuda.synchronize()
            if r == 0:
                t.read()
        ms, cnt = t.read()
        pairs = float(n) * n_tgt
        print(f"n={n} tgt={n_tgt} uni={uniform} {' '.join(f'{k[13:]}={v}' for k, v in cfg.items())}: {ms:9.3f} ms  {pairs / ms / 1e9:8.3f} Gpairs/s  {14 * pairs / ms / 1e9:8.2f} TFLOP/s"
              f"  ({14 * pairs / ms / 1e9 / 157.3 * 100:5.1f}% of 157.3)", flush=True)
        for k in cfg:
            os.environ.pop(k, None)


if __name__ == "__main__":
    n = 1 << 20
    cfgs = [{"NBODY_DIRECT_ASM": 1}, {"NBODY_DIRECT_ASM": 0}]
    run(n, n, cfgs, reps=2, uniform=0.0)
    run(n, n, cfgs[:1], reps=2, uniform=1.0)
