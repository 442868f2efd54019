"""mmsr.data.pil_bicubic against the installed Pillow (the library the reference's dataset calls,
mmsr/data/ref_cufed_dataset.py:118-130): bit-exact on uint8 images, CPU here and GPU under -m gpu."""
import numpy as np
import pytest
import torch


def _images():
    rng = np.random.default_rng(5)
    smooth = np.kron(rng.random((3, 20, 24)) * 255, np.ones((1, 8, 8))) + rng.random((3, 160, 192)) * 30
    return [np.clip(smooth, 0, 255).astype(np.uint8), (rng.random((3, 164, 100)) * 255).astype(np.uint8)]


def _check(device):
    Image = pytest.importorskip("PIL.Image")
    from mmsr.data import make_lq_and_up, pil_bicubic_resize
    for a in _images():
        C, H, W = a.shape
        pil = Image.fromarray(a.transpose(1, 2, 0))
        for (oh, ow) in ((H // 4, W // 4), (H * 2, W * 2), (H // 4, W), (37, 53)):
            want = np.array(pil.resize((ow, oh), Image.BICUBIC)).transpose(2, 0, 1)
            got = pil_bicubic_resize(torch.from_numpy(a).to(device), oh, ow).cpu().numpy()
            assert np.array_equal(got, want), (a.shape, oh, ow, int(np.abs(got.astype(int) - want.astype(int)).max()))
        lq_pil = pil.resize((W // 4, H // 4), Image.BICUBIC)
        up_pil = lq_pil.resize((W, H), Image.BICUBIC)
        lq, up = make_lq_and_up(torch.from_numpy(a)[None].to(device), 4)
        assert np.array_equal(lq[0].cpu().numpy(), np.array(lq_pil).transpose(2, 0, 1))
        assert np.array_equal(up[0].cpu().numpy(), np.array(up_pil).transpose(2, 0, 1))


def test_pil_bicubic_bit_exact_cpu():
    _check(torch.device("cpu"))


@pytest.mark.gpu
def test_pil_bicubic_bit_exact_gpu(dev):
    _check(dev)


# ---------------------------------------------------------------------------------------------------------------------
# rank partition of the training set: DistIterSampler + the batch rule
# ---------------------------------------------------------------------------------------------------------------------
def test_dist_iter_sampler_matches_the_reference(golden_dir):
    """Index lists of every rank equal those of the reference's class (fixture: tests/golden/make_golden.py sampler)."""
    from make_golden import SAMPLER_CASES
    from mmsr.data import DistIterSampler
    gold = np.load(f"{golden_dir}/sampler_golden.npz")
    for (n, ratio, world) in SAMPLER_CASES:
        for epoch in (0, 5):
            lists = []
            for rank in range(world):
                s = DistIterSampler(list(range(n)), num_replicas=world, rank=rank, ratio=ratio)
                s.set_epoch(epoch)
                got = np.array(list(iter(s)), np.int64)
                assert len(s) == len(got)
                assert np.array_equal(got, gold[f"n{n}_r{ratio}_w{world}_e{epoch}_rank{rank}"]), (n, ratio, world, epoch, rank)
                lists.append(got)
            # the ranks' lists interleave into one permutation of the enlarged epoch, folded onto the dataset
            whole = np.stack(lists, 1).ravel()
            assert np.array_equal(whole, s.epoch_slots().numpy())
            counts = np.bincount(whole, minlength=n)
            assert counts.max() - counts.min() <= 1 + (len(whole) % n != 0)


def test_per_rank_batch_rule():
    from mmsr.data import per_rank_batch_size
    assert per_rank_batch_size(32, 8) == 4 and per_rank_batch_size(32, 1) == 32
    with pytest.raises(AssertionError):
        per_rank_batch_size(9, 8)      # stage 3's YAML batch of 9 cannot be sharded over 8 GPUs (data/__init__.py:72)


def _sampler_worker(rank, world, port, q):
    import os
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "c2-matching_amd"))
    import torch.distributed as dist
    from mmsr.data import DistIterSampler, per_rank_batch_size
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        data = list(range(37))
        s = DistIterSampler(data, ratio=2)          # world size and rank from the process group, as the reference
        s.set_epoch(3)
        mine = torch.tensor(list(iter(s)), dtype=torch.int64)
        bs = per_rank_batch_size(32, world)
        # what the ranks exchange to check the partition is test scaffolding: the product's data path has no collective
        everyone = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(everyone, mine)
        q.put((rank, [e.tolist() for e in everyone], s.epoch_slots().tolist(), bs))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_dist_iter_sampler_partitions_the_epoch_across_gloo_ranks(world):
    import os
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() + 7 * world) % 2000
    procs = [ctx.Process(target=_sampler_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict((r, (lists, slots, bs)) for r, lists, slots, bs in (q.get(timeout=120) for _ in range(world)))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    lists, slots, bs = got[0]
    assert bs == 32 // world
    assert all(got[r][1] == slots for r in range(world))                    # same epoch permutation on every rank
    assert all(got[r][0] == lists for r in range(world))
    assert len({len(x) for x in lists}) == 1                                # equally long shares
    inter = [v for t in zip(*lists) for v in t]
    assert inter == slots                                                   # disjoint slices whose union is the epoch
    assert sorted(set(inter)) == list(range(37))                            # every sample is drawn


def test_bench_gpus2_launch_path_reaches_the_process_group():
    """`python bench.py --gpus 2` as the driver's plain command: bench.py re-executes itself under torch.distributed.run with
    two ranks on 127.0.0.1, every rank reads RANK / LOCAL_RANK / WORLD_SIZE, joins the process group, passes the barrier and
    the MAX all-reduce that the timing uses, rank 0 prints one JSON line (the first multi-GPU run must not
    die on plumbing).  CPU host: the gloo backend stands in for RCCL ($C2M_BENCH_BACKEND), the device calls are skipped."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, C2M_BENCH_BACKEND="gloo")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--plumbing-only"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1, r.stdout                  # rank 0 only
    d = lines[0]
    assert d["plumbing"] == "ok" and d["n_gpus"] == 2 and d["world_size"] == 2 and d["process_group_world"] == 2
    assert d["max_over_ranks"] == 2.0 and d["master"] == "127.0.0.1"
    # a mismatched launch is refused, not silently run on the wrong world size
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--plumbing-only"],
                       env=dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)
