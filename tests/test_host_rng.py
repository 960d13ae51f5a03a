"""scipnp_host_legacy_normal: NumPy's legacy Gaussian stream restated in libscipnp (host code, GIL-free) must reproduce
np.random.normal bit for bit -- values AND final generator state -- because the reference's FastDVDnet finetune draws its
input noise from the global legacy generator (utils/utils_image.py:183-192)."""
import threading
import time

import numpy as np

from adaptivepnp_sci_amd.finetune import legacy_normal


def test_stream_and_state_equal_numpy():
    for seed, sizes in ((0, [(7,), (1000,), (3, 5, 11)]), (42, [(8, 3, 16, 16), (1,), (2,)]), (123456789, [(625,), (1249,), (1,)])):
        np.random.seed(seed)
        ref = [np.random.normal(0.25, 5 / 255, s) for s in sizes]
        ref_next = np.random.random(4)
        ref_int = np.random.randint(0, 1000, 5)
        np.random.seed(seed)
        got = [legacy_normal(0.25, 5 / 255, s) for s in sizes]
        assert all(np.array_equal(a, b) for a, b in zip(got, ref))
        # the generator continues exactly where NumPy's would (including the cached second deviate of an odd draw)
        assert np.array_equal(np.random.random(4), ref_next) and np.array_equal(np.random.randint(0, 1000, 5), ref_int)
    np.random.seed(5)
    np.random.normal(size=3)                       # leaves a cached deviate behind
    st = np.random.get_state()
    a = np.random.normal(0, 1, 10)
    np.random.set_state(st)
    assert np.array_equal(legacy_normal(0, 1, 10), a)


def test_reference_seeding_gives_the_golden_noise():
    """worker_init_fn(0) of the reference = np.random.seed(42) (utilspy.py:22-25); the first draw of the finetune is the
    noise captured in the golden file from the reference run"""
    from conftest import load_gold
    g = load_gold('fastdvd_finetune_64x64x8')
    np.random.seed(42)
    assert np.array_equal(legacy_normal(0, 5 / 255, (8, 3, 64, 64)), g['noise'])


def test_draw_does_not_hold_the_gil():
    """NumPy's legacy normal keeps the GIL for the whole call; this one must let another Python thread run"""
    def busy(n=1_500_000):
        s = 0
        for i in range(n):
            s += i
        return s
    ok = False
    for _attempt in range(4):                       # timing on a shared host: accept the first clean measurement
        t0 = time.perf_counter(); busy(); t_busy = time.perf_counter() - t0
        t0 = time.perf_counter(); legacy_normal(0, 1, (8, 3, 512, 512)); t_rng = time.perf_counter() - t0
        th = threading.Thread(target=legacy_normal, args=(0, 1, (8, 3, 512, 512)))
        t0 = time.perf_counter(); th.start(); busy(); t_both = time.perf_counter() - t0; th.join()
        if t_both < 0.8 * (t_busy + t_rng):
            ok = True
            break
    assert ok, (t_busy, t_rng, t_both)


def test_prefetch_close_hands_unused_draws_back():
    """NoisePrefetch.close(): the global generator ends up where a synchronous caller who drew only the draws actually
    used would have left it (a failed solve must not leave the stream advanced)"""
    from adaptivepnp_sci_amd.finetune import NoisePrefetch
    shape = (2, 3, 8, 8)
    np.random.seed(7)
    first = legacy_normal(0, 5 / 255, shape)
    want_next = np.random.normal(size=5)
    np.random.seed(7)
    p = NoisePrefetch(shape, 3)
    got = p.get()
    time.sleep(0.2)                                   # let the worker draw ahead
    p.close()
    assert np.array_equal(got, first)
    assert np.array_equal(np.random.normal(size=5), want_next)
    # nothing consumed at all: the state is untouched
    np.random.seed(9)
    st = np.random.get_state()
    p = NoisePrefetch(shape, 2)
    time.sleep(0.2)
    p.close()
    assert np.array_equal(np.random.get_state()[1], st[1]) and np.random.get_state()[2] == st[2]
