"""N > 1 path: world_size-2 gloo groups.  The CPU test covers the host sharding logic; the GPU test runs
two real engine ranks on one GPU with a gloo-backed all-reduce callback and compares with one rank."""
import json
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch(world, mode, shard, timeout=600):
    port = free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "multi_rank_worker.py"), mode, shard],
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env))
    outputs = []
    for proc in procs:
        try:
            out, err = proc.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for p in procs:
                p.kill()
            raise
        assert proc.returncode == 0, err[-2000:]
        outputs.append(out)
    return json.loads(outputs[0].strip().splitlines()[-1])


def test_position_windows_and_gloo_all_reduce_cpu(built):
    out = launch(2, "cpu", "position")
    for name, length in (("main", 997), ("S", 211)):
        windows = out[name]["windows"]
        assert windows[0][0] == 0 and windows[-1][1] == length
        assert all(a[1] == b[0] for a, b in zip(windows[:-1], windows[1:]))  # disjoint cover
        assert out[name]["equal"]


@pytest.mark.gpu
@pytest.mark.parametrize("shard", ["position", "sequence"])
def test_two_ranks_match_one_rank_gpu(built, shard):
    single = launch(1, "gpu", shard)
    double = launch(2, "gpu", shard)
    assert [s for s, _ in single] == [200, 200, 200, 200]
    assert single == double


@pytest.mark.gpu
def test_torch_can_alias_engine_device_memory(built):
    """bench.py's RCCL leg wraps the engine's count buffer in a torch tensor through
    __cuda_array_interface__ (int32 view): check the aliasing and that kernels still run with torch's
    HIP runtime loaded in the same process."""
    import ctypes

    import numpy as np
    import torch

    from silo_amd import binding

    assert torch.cuda.is_available()
    torch.zeros(1, device="cuda")  # initialise torch's runtime first
    ref = np.ones(5, dtype=np.uint8)
    with binding.GpuStore(1000, [dict(name="m", alphabet="nuc", reference=ref)]) as store:
        buf = store.malloc(4 * 16)
        store.memset(buf, 0, 64)

        class View:
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 2}

        tensor = torch.as_tensor(View(buf.value, 16), device="cuda")
        tensor += torch.arange(16, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        assert np.array_equal(store.read(buf, np.uint32, 16), np.arange(16, dtype=np.uint32))
        store.append_sequences(0, 0, ["ACGTN"] * 1000)
        store.finalize()
        counts = store.mutations_scan(0)
        assert counts[:, 1:].diagonal().tolist() == [1000, 1000, 1000, 1000] and counts[4].sum() == 0
