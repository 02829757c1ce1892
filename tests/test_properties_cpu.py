"""Property-based tier (hypothesis): invariants of the pure-Python planning code over random inputs —
the distributed sampler against torch's for every (length, world, shuffle, drop_last, seed, epoch),
and the gradient bucket planner (every element of every parameter in exactly one bucket, reverse
order, caps, large-parameter splitting, tail bucket) — the parts of the multi-GPU path whose
mistakes would only show on hardware this project cannot reach."""
import math

import pytest

hypothesis = pytest.importorskip("hypothesis")
from hypothesis import given, settings, strategies as st  # noqa: E402

from distributed_training_pytorch_amd.parallel.reducer import plan_buckets  # noqa: E402
from distributed_training_pytorch_amd.parallel.sampler import ShardedSampler  # noqa: E402


@settings(max_examples=120, deadline=None)
@given(n=st.integers(1, 300), world=st.integers(1, 9), shuffle=st.booleans(), drop_last=st.booleans(),
       seed=st.integers(0, 10 ** 6), epoch=st.integers(0, 50))
def test_sampler_is_torch_distributed_sampler(n, world, shuffle, drop_last, seed, epoch):
    from torch.utils.data.distributed import DistributedSampler
    ds = list(range(n))
    total = []
    for r in range(world):
        ref = DistributedSampler(ds, num_replicas=world, rank=r, shuffle=shuffle, seed=seed, drop_last=drop_last)
        ref.set_epoch(epoch)
        mine = ShardedSampler(n, r, world, shuffle=shuffle, seed=seed, drop_last=drop_last)
        mine.set_epoch(epoch)
        assert list(mine) == list(ref) and len(mine) == len(ref)
        total += list(mine)
    if not drop_last:
        assert set(total) == set(range(n))


@settings(max_examples=120, deadline=None)
@given(n=st.integers(0, 200), world=st.integers(1, 9), shuffle=st.booleans(), seed=st.integers(0, 1000))
def test_unpadded_shards_partition_the_dataset(n, world, shuffle, seed):
    """pad=False (validation): disjoint shards, no duplicates, sizes differ by at most one — also when
    the dataset is smaller than the world (empty shards)."""
    parts = [list(ShardedSampler(n, r, world, shuffle=shuffle, seed=seed, pad=False)) for r in range(world)]
    flat = sum(parts, [])
    assert sorted(flat) == list(range(n))
    sizes = [len(p) for p in parts]
    assert max(sizes) - min(sizes) <= 1
    assert all(len(ShardedSampler(n, r, world, pad=False)) == sizes[r] for r in range(world))


@settings(max_examples=200, deadline=None)
@given(numels=st.lists(st.integers(1, 5000), min_size=1, max_size=24), elem=st.sampled_from([2, 4]),
       first=st.integers(1, 4000), cap=st.integers(1, 20000),
       split=st.one_of(st.none(), st.integers(64, 12000)), tail=st.one_of(st.none(), st.integers(1, 6000)))
def test_bucket_plan_invariants(numels, elem, first, cap, split, tail):
    plan = plan_buckets(numels, elem, first, cap, split, tail)
    assert all(len(b) > 0 for b in plan)
    cover = {}
    for b in plan:
        for i, s, e in b:
            assert 0 <= s < e <= numels[i]
            cover.setdefault(i, []).append((s, e))
    assert sorted(cover) == list(range(len(numels)))           # every parameter appears ...
    for i, n in enumerate(numels):                             # ... and every element exactly once
        segs = sorted(cover[i])
        assert segs[0][0] == 0 and segs[-1][1] == n
        assert all(segs[k][1] == segs[k + 1][0] for k in range(len(segs) - 1))
    order = [i for b in plan for (i, _, _) in b]
    assert order == sorted(order, reverse=True)                # reverse registration order throughout
    if split:                                                  # no range larger than the split size
        assert all((e - s) * elem <= max(split, elem) for b in plan for (_, s, e) in b)
    base = plan_buckets(numels, elem, first, cap, split, None)
    if tail is None:
        assert plan == base
    else:                                                      # the tail only ever cuts the LAST bucket in two
        assert plan == base or (plan[:-2] == base[:-1] and plan[-2] + plan[-1] == base[-1])
        if plan != base:
            assert sum((e - s) * elem for _, s, e in plan[-1]) <= tail


def test_sampler_lengths_match_the_formula():
    for n, world in [(10, 3), (11, 4), (3, 4), (64, 8)]:
        assert len(ShardedSampler(n, 0, world)) == math.ceil(n / world)
