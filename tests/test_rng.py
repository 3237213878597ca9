"""The counter-based draw function: scalar, vectorised and list helpers agree, and a few
known answers pin the hash (the HIP kernels and the C checker evaluate the same function;
their agreement with these values is what the parity tests exercise end to end)."""
import numpy

from raytracer_amd import rng


def test_known_answers():
    assert rng.mix64(0) == 0xE220A8397B1DCDAF            # splitmix64's first output for state 0
    assert rng.draw64(1, rng.PHASE_TRACE, 0, 0, 0) == rng.draw64(1, rng.PHASE_TRACE, 0, 0, 0)
    # distinct in every argument
    base = rng.draw64(1, rng.PHASE_TRACE, 5, 3, 2)
    assert len({base, rng.draw64(2, rng.PHASE_TRACE, 5, 3, 2), rng.draw64(1, rng.PHASE_PIXEL, 5, 3, 2),
                rng.draw64(1, rng.PHASE_TRACE, 6, 3, 2), rng.draw64(1, rng.PHASE_TRACE, 5, 4, 2),
                rng.draw64(1, rng.PHASE_TRACE, 5, 3, 3)}) == 6


def test_array_form_matches_scalar():
    pix = numpy.arange(0, 5000, 7)
    for counter in (0, 1):
        h = rng.draw64_array(9, rng.PHASE_PIXEL, pix, 0, counter)
        want = [rng.draw64(9, rng.PHASE_PIXEL, int(p), 0, counter) for p in pix]
        assert [int(x) for x in h] == want
        assert list(rng.index_array(h, 83)) == [rng.index_from(x, 83) for x in want]


def test_index_and_unit_ranges():
    hs = [rng.draw64(3, rng.PHASE_TRACE, 1, 2, c) for c in range(2000)]
    for n in (1, 2, 16, 83, 8300):
        idx = [rng.index_from(h, n) for h in hs]
        assert min(idx) >= 0 and max(idx) < n
    assert len(set(rng.index_from(h, 83) for h in hs)) == 83
    u = [rng.unit_from(h) for h in hs]
    assert 0.0 <= min(u) and max(u) < 1.0 and 0.45 < sum(u) / len(u) < 0.55


def test_context_streams():
    ctx = rng.DrawContext(seed=4)
    a = [ctx.next64() for _ in range(3)]
    ctx.point(rng.PHASE_TRACE, 10, 2)
    t = ctx.next64()
    ctx.point(rng.PHASE_TABLE)                # the table stream resumes where it stopped
    b = ctx.next64()
    ctx.reseed(4)
    assert [ctx.next64() for _ in range(4)] == a + [b]
    ctx.point(rng.PHASE_TRACE, 10, 2)
    assert ctx.next64() == t


def test_shuffle_is_a_permutation():
    rng.context.reseed(11)
    items = list(range(64))
    rng.shuffle(items)
    assert sorted(items) == list(range(64)) and items != list(range(64))
    assert rng.choice([7]) == 7
    assert 2.0 <= rng.uniform(2.0, 3.0) < 3.0
