"""CPU: host-side logic of the product (no GPU compute), the C ABI surface, and the sharded runner."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_builds_loads_and_exports_every_declared_symbol():
    """include/otvm_hip.h is the contract: every function it declares must be exported by the .so (and nothing
    the binding uses may be missing from the header)."""
    import __graft_entry__ as g
    path = g.build()
    header = open(os.path.join(ROOT, "include", "otvm_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    declared = set(re.findall(r"\b(otvm_[a-z0-9_]+)\s*\(", header))
    declared -= {"otvm_conv_params", "otvm_preprocess_params"}
    assert len(declared) >= 20
    lib = ctypes.CDLL(path)
    for sym in sorted(declared):
        assert hasattr(lib, sym), "declared in the header but not exported: " + sym
    from otvm_amd import lib as L
    assert set(L.EXPORTED) == declared, (set(L.EXPORTED) ^ declared)
    lib.otvm_abi_version.restype = ctypes.c_int
    assert lib.otvm_abi_version() == 1


def test_ctypes_struct_matches_c_layout():
    """sizeof(otvm_conv_params) / otvm_preprocess_params as compiled by gcc == the ctypes mirrors."""
    src = '#include <stdio.h>\n#include "otvm_hip.h"\nint main(){printf("%zu %zu\\n", sizeof(otvm_conv_params), sizeof(otvm_preprocess_params));return 0;}\n'
    exe = os.path.join(ROOT, "otvm_amd", "csrc", "build", "abi_sizes")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(ROOT, "include"), "-o", exe], input=src.encode(), check=True)
    a, b = (int(v) for v in subprocess.check_output([exe]).split())
    from otvm_amd import lib as L
    assert ctypes.sizeof(L.ConvParams) == a and ctypes.sizeof(L.PreprocessParams) == b


def test_bank_policy_engine_equals_oracle():
    from otvm_amd.engine import bank_update as eng_update
    from oracle.otvm_oracle import bank_update as orc_update
    rng = np.random.Generator(np.random.PCG64(0))
    for max_num in (0, 1, 2, 3, 5):
        for skip in (2, 3, 5):
            eb, ob = [], []
            for t in range(40):
                mem = (t % skip == 0) if skip > 2 else False
                slot = dict(frame=t)
                eb, released = eng_update(eb, slot, t == 0, mem, max_num)
                ob = orc_update(ob, (None, None, t), t == 0, mem, max_num)
                assert [s["frame"] for s in eb] == [b[2] for b in ob]
                assert all(not any(r is k for k in eb) for r in released)
                assert len(eb) <= max(1, max_num)


def test_pad_amounts_and_schedule():
    from otvm_amd.engine import pad_amounts
    from otvm_amd.video import memory_schedule
    assert pad_amounts(1080, 1920, 32) == (0, 0, 4, 4)          # 1080p -> 1088, pad (0,0,4,4)
    assert pad_amounts(100, 150, 32) == (5, 5, 14, 14)
    assert pad_amounts(33, 65, 32) == (15, 16, 15, 16)          # odd remainder: extra pixel bottom/right
    # eval.py:184-189: large input doubles the skip and halves the bank
    assert memory_schedule(10, 1080, 1920) == (True, 5, False)
    assert memory_schedule(10, 2160, 3840) == (False, 2, True)
    assert memory_schedule(20, 2160, 3840) == (True, 2, True)
    assert memory_schedule(4, 480, 832, skip=2) == (False, 5, False)


def test_trimap_file_to_onehot():
    from otvm_amd.video import trimap_file_to_onehot
    g = np.array([[0, 128, 254], [254, 0, 128]], np.uint8)      # demo/dove levels {0,128,254}
    oh = trimap_file_to_onehot(g)
    assert oh.shape == (3, 2, 3)
    np.testing.assert_array_equal(oh[0], [[1, 0, 0], [0, 1, 0]])
    np.testing.assert_array_equal(oh[1], [[0, 1, 0], [0, 0, 1]])
    np.testing.assert_array_equal(oh[2], [[0, 0, 1], [1, 0, 0]])
    two = np.array([[0, 255]], np.uint8)                        # two-level quirk (dataset.py:890-893)
    oh = trimap_file_to_onehot(two)
    np.testing.assert_array_equal(oh[1], [[1, 1]])


def test_shard_sequences_partitions():
    from otvm_amd.dist import shard_sequences
    for world in (1, 2, 4, 8):
        for n in (1, 5, 48):
            lengths = [(7 * i) % 13 + 1 for i in range(n)]
            seen = []
            for r in range(world):
                seen += shard_sequences(n, r, world, lengths)
            assert sorted(seen) == list(range(n))
            assert sorted(sum((shard_sequences(n, r, world) for r in range(world)), [])) == list(range(n))
    loads = [sum([10, 9, 8, 1, 1, 1][i] for i in shard_sequences(6, r, 2, [10, 9, 8, 1, 1, 1])) for r in range(2)]
    assert abs(loads[0] - loads[1]) <= 4 and sum(loads) == 30      # LPT greedy: 13 vs 17


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from otvm_amd.dist import run_sharded
    seqs = [dict(frames=list(range(3 + i)), id=i) for i in range(5)]

    def matte(seq):                         # stand-in for the GPU path: the sharding/reduction logic is what is tested
        return dict(alpha=torch.full((len(seq["frames"]), 4, 4), float(seq["id"])))

    def ref(seq):
        return torch.full((len(seq["frames"]), 4, 4), float(seq["id"]) + 0.5)
    out = run_sharded(seqs, matte, rank=rank, world=world, reference_fn=ref)
    q.put((rank, out["frames"], out["sad"], out["max_abs"], out["sequences"]))
    dist.destroy_process_group()


def test_sharded_runner_two_ranks_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
    total_frames = sum(3 + i for i in range(5))
    for rank, frames, sad, max_abs, mine in res:
        assert frames == total_frames                               # SUM all-reduce
        assert abs(sad - 0.5 * 16 * total_frames / 1000.0) < 1e-9
        assert abs(max_abs - 0.5) < 1e-12                           # MAX all-reduce
    assert sorted(res[0][4] + res[1][4]) == [0, 1, 2, 3, 4]


def test_product_does_not_import_oracle():
    """The oracle is test infrastructure: nothing under otvm_amd/ may import it (or the reference)."""
    for dp, _, files in os.walk(os.path.join(ROOT, "otvm_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", src, flags=re.S).replace("# oracle", ""), f
                assert "/root/reference" not in src, f
