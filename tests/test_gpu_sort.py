"""The engine's own radix sort of (u32 key, u32 value) pairs (alga_amd/csrc/radix_sort.hip; what orders the nodes of the index build by
minimizer key in place of the reference's per-length re-bucketing, src/GraphCreators/GraphCreatorPrefSuf.cpp:317-332) against torch's
stable sort: every size class (one partial tile, exact tiles, many chunks, both sides of 2^22 where the library path changes its bits),
every pass plan (1 .. 32 key bits: one to four passes, digits of 1 .. 10 bits), skewed keys (all equal, two values, sorted, reversed),
bit for bit including the order of equal keys (stability).  The rocPRIM path (option own_sort = 0) is held to the same on full keys."""
import numpy as np
import pytest

import alga_amd
from alga_amd.engine import device_view

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    e = alga_amd.Engine(0)
    yield e
    e.close()


def _check(eng, keys_u32, begin_bit, own=True):
    import torch
    n = len(keys_u32)
    k = torch.from_numpy(keys_u32.view(np.int32)).cuda()
    v = torch.arange(n, dtype=torch.int32, device="cuda")
    kp, vp, _ = eng.sort_u32_pairs_device(k, v, begin_bit, own)
    torch.cuda.synchronize()
    if n == 0:
        return
    gk = device_view(kp, (n,), k.device).clone()
    gv = device_view(vp, (n,), k.device).clone()
    # the reference order: stable on the looked-at bits
    sk = (k.to(torch.int64) & 0xFFFFFFFF) >> begin_bit
    _, perm = torch.sort(sk, stable=True)
    assert torch.equal(gv.to(torch.int64), perm), "values (= stable order) differ, begin_bit %d n %d" % (begin_bit, n)
    assert torch.equal(gk, k[perm])
    assert torch.equal(k, torch.from_numpy(keys_u32.view(np.int32)).cuda())          # the input is left untouched


@pytest.mark.parametrize("n", [0, 1, 63, 64, 65, 8191, 8192, 8193, 100_000, 3 * 8192 * 512 + 17, (1 << 22) - 5, (1 << 22) + 12345])
def test_random_keys_every_size_class(eng, n):
    rng = np.random.default_rng(n + 7)
    keys = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
    _check(eng, keys, 3)                                   # 29 bits: 10 + 10 + 9, the north-star plan
    _check(eng, keys, 0)                                   # 32 bits: four passes of 8


@pytest.mark.parametrize("begin_bit", list(range(0, 32)))
def test_every_pass_plan(eng, begin_bit):
    rng = np.random.default_rng(100 + begin_bit)
    keys = rng.integers(0, 1 << 32, size=200_003, dtype=np.uint64).astype(np.uint32)
    _check(eng, keys, begin_bit)


@pytest.mark.parametrize("kind", ["equal", "two", "sorted", "reversed", "few_high_bits", "one_digit_hot"])
def test_skewed_keys(eng, kind):
    n = 300_000
    rng = np.random.default_rng(5)
    if kind == "equal":
        keys = np.full(n, 0xDEADBEE8, dtype=np.uint32)
    elif kind == "two":
        keys = np.where(rng.random(n) < 0.5, 0x00000008, 0xFFFFFFF8).astype(np.uint32)
    elif kind == "sorted":
        keys = np.sort(rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32))
    elif kind == "reversed":
        keys = np.sort(rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32))[::-1].copy()
    elif kind == "few_high_bits":
        keys = (rng.integers(0, 4, size=n, dtype=np.uint64) << 30).astype(np.uint32) | rng.integers(0, 8, size=n, dtype=np.uint64).astype(np.uint32)
    else:
        keys = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
        keys[rng.random(n) < 0.9] = 0x12345678                  # nine items in ten share every digit: one wave-match group of 64 lanes per step
    _check(eng, keys, 3)
    _check(eng, keys, 0)


def test_library_path_on_full_keys(eng):
    rng = np.random.default_rng(9)
    for n in (100_000, (1 << 22) + 999):
        keys = rng.integers(0, 1 << 32, size=n, dtype=np.uint64).astype(np.uint32)
        _check(eng, keys, 0, own=False)


def test_index_build_same_graph_with_either_sort(eng):
    """the whole build with the engine's sort and with the library's: same edges (the order of equal keys is the same for both -- stable)."""
    import gen_reads
    from alga_amd import workload
    codes, _ = gen_reads.sample_reads(20_000, 150, 60_000, 77)
    words, lens, _ = workload.make_nodes(codes)
    lo, rs = alga_amd.derive_params(144.0)
    a = eng.prefsuf_host(words, lens, lo, rs)
    eng.set_option("own_sort", 0)
    try:
        b = eng.prefsuf_host(words, lens, lo, rs)
    finally:
        eng.set_option("own_sort", 1)
    assert a.shape == b.shape and (a == b).all()


# ---- the 16-byte records of the supplement (radix_sort.hip: rsort_u64_pairs) ---------------------------------------------------------------

def _check64(eng, keys_u64, bits, own=True):
    import torch
    n = len(keys_u64)
    k = torch.from_numpy(keys_u64.view(np.int64)).cuda()
    v = (torch.arange(n, dtype=torch.int64, device="cuda") << 33) | 5          # values wider than 32 bits
    kp, vp, ms = eng.sort_u64_pairs_device(k, v, bits, own)
    torch.cuda.synchronize()
    if n == 0:
        return ms
    gk = device_view(kp, (n,), k.device, "<i8").clone()
    gv = device_view(vp, (n,), k.device, "<i8").clone()
    low = k & ((1 << bits) - 1) if bits < 63 else k
    _, perm = torch.sort(low, stable=True)
    assert torch.equal(gv, v[perm]), "values (= stable order) differ, bits %d n %d" % (bits, n)
    assert torch.equal(gk, k[perm])
    assert torch.equal(k, torch.from_numpy(keys_u64.view(np.int64)).cuda())
    return ms


@pytest.mark.parametrize("n", [0, 1, 63, 65, 4095, 4096, 4097, 100_000, 3 * 4096 * 512 + 17, (1 << 22) + 12345])
def test_u64_records_every_size_class(eng, n):
    rng = np.random.default_rng(n + 11)
    keys = rng.integers(0, 1 << 63, size=n, dtype=np.uint64) * 2 + rng.integers(0, 2, size=n, dtype=np.uint64)
    _check64(eng, keys, 30)                                # the supplement's plan at 10 M reads: 10 + 10 + 10
    _check64(eng, keys, 40)


@pytest.mark.parametrize("bits", [1, 7, 8, 9, 10, 11, 19, 20, 21, 29, 31, 32, 33, 41, 50])
def test_u64_records_every_pass_plan(eng, bits):
    rng = np.random.default_rng(300 + bits)
    keys = rng.integers(0, 1 << 63, size=150_001, dtype=np.uint64)
    _check64(eng, keys, bits)


@pytest.mark.parametrize("kind", ["equal", "two", "sorted", "reversed", "one_digit_hot"])
def test_u64_records_skewed_keys(eng, kind):
    n = 200_000
    rng = np.random.default_rng(6)
    if kind == "equal":
        keys = np.full(n, 0x123456789ABCDEF, dtype=np.uint64)
    elif kind == "two":
        keys = np.where(rng.random(n) < 0.5, np.uint64(0x3FFFFFFF), np.uint64(1 << 40)).astype(np.uint64)
    elif kind == "sorted":
        keys = np.sort(rng.integers(0, 1 << 62, size=n, dtype=np.uint64))
    elif kind == "reversed":
        keys = np.sort(rng.integers(0, 1 << 62, size=n, dtype=np.uint64))[::-1].copy()
    else:
        keys = rng.integers(0, 1 << 62, size=n, dtype=np.uint64)
        keys[rng.random(n) < 0.9] = np.uint64(0x2AAAAAAA)
    _check64(eng, keys, 30)


def test_u64_records_library_path_agrees(eng):
    rng = np.random.default_rng(8)
    keys = rng.integers(0, 1 << 62, size=500_000, dtype=np.uint64)
    _check64(eng, keys, 32, own=False)
    _check64(eng, keys, 40, own=False)
