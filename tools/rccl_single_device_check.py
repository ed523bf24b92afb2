"""Child-process checks of the collectives on ONE device (run by tests/test_gls_gpu.py).

    PDC_FORCE_RCCL=1 python tools/rccl_single_device_check.py         # pdc_gls_scan_multi + plan, RCCL forced
    PDC_FORCE_RCCL=1 PDC_FORCE_RCCL_FAIL=1 python tools/rccl_single_device_check.py fail
                                                                       # ncclCommInitAll "fails": the plan must fall back, loudly
    python tools/rccl_single_device_check.py torch                    # torchrun_sharded.sharded_gls, 1-rank nccl group
"""
import os
import sys

sys.path.insert(0, os.getcwd())
import numpy as np

from periodicity_amd import _cabi

TORCH = len(sys.argv) > 1 and sys.argv[1] == "torch"
FAIL = len(sys.argv) > 1 and sys.argv[1] == "fail"
if TORCH:   # torch brings its own HIP runtime: it has to initialise before the library's first call
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29541", RANK="0", WORLD_SIZE="1")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))

rng = np.random.default_rng(0)
t = np.sort(rng.uniform(0, 500, 500))
dy = rng.uniform(.05, .2, 500)
y = np.sin(t / 3) + dy * rng.standard_normal(500)
f0, delta, nf = 0.001, 0.0007, 3001
a = _cabi.gls_scan(t, y, dy, f0, delta, nf)
if FAIL:
    # the communicator "cannot be built": no error, a warning on stderr, the reason kept in the plan, the copy exchange
    one = _cabi.GlsPlan([0], 600, 4000)                       # (one real device: RCCL forced, then failed)
    info1 = one.info()
    one.upload(t, y, dy)
    one.scan(f0, delta, nf)
    b = one.download()
    one.close()
    many = _cabi.GlsPlan([0], 600, 4000, loopback_slots=3)    # (three slots: the injected failure + the copies)
    info3 = many.info()
    many.upload(t, y, dy)
    many.scan(f0, delta, nf)
    c = many.download(2)
    many.close()
    print("fallback info:", info1, info3)
    print("fallback after a failed ncclCommInitAll equal:",
          np.array_equal(a, b) and np.allclose(a, c, rtol=1e-11, atol=1e-13)   # (three slabs: other tile shapes, other summation order)
          and info1["rccl_ranks"] == 0 and bool(info1["init_error"])
          and info3["exchange"] == "copy" and bool(info3["init_error"]))
elif TORCH:
    from tools.torchrun_sharded import sharded_gls
    b = sharded_gls(t, y, dy, f0, delta, nf)
    print("sharded_gls equal:", np.array_equal(a, b))
    dist.destroy_process_group()
else:
    b = _cabi.gls_scan_multi(t, y, dy, f0, delta, nf, devices=(0,))
    c = _cabi.gls_scan_multi(t, y, dy, f0, delta, nf, devices=(0,))      # second call: the cached plan
    plan = _cabi.GlsPlan([0], 600, 4000)
    plan.upload(t, y, dy)
    for _ in range(3):
        plan.scan(f0, delta, nf)
    d = plan.download()
    plan.close()
    print("forced RCCL single-device all-gather equal:",
          np.array_equal(a, b) and np.array_equal(a, c) and np.array_equal(a, d))
