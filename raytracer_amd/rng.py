"""Counter-based random draws shared by the host scene layer and the device kernels.

The reference draws from Python's global Mersenne stream in execution order
(`random.choice/shuffle/uniform` at /root/reference/sampler.py:26,27,31,79,82,85,
geometry.py:379, camera.py:84,118,197,252) and never seeds it, so a render cannot be
replayed on a parallel machine.  Every draw here is instead a pure function of

    (seed, phase, a, b, counter)

where `phase` says what is being drawn and (a, b) identify the pixel and the sample:

    PHASE_TABLE  sampler table construction        a = 0,           b = 0
    PHASE_PIXEL  per-pixel pattern choice          a = pixel index, b = 0
    PHASE_FRAME  per-frame sample-order shuffle    a = 0,           b = 0
    PHASE_TRACE  draws inside one traced sample    a = pixel index, b = sample index

`counter` numbers the draws made under one (phase, a, b); it restarts at zero whenever
the context is re-pointed.  The HIP kernels evaluate the same hash (see
`rtamd_draw` in csrc/rt_device.h), so a sample's colour does not depend on which lane,
wave or GPU traced it, nor on the order samples are traced in.
"""

MASK64 = (1 << 64) - 1

PHASE_TABLE = 1
PHASE_PIXEL = 2
PHASE_FRAME = 3
PHASE_TRACE = 4

_PHASE_MUL = 0xD6E8FEB86659FD93


def mix64(z):
    """splitmix64 finaliser (Steele, Lea, Flood 2014) on a 64-bit integer."""
    z = (z + 0x9E3779B97F4A7C15) & MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


def draw64(seed, phase, a, b, counter):
    """The 64-bit draw for one (seed, phase, a, b, counter) tuple."""
    h = mix64((seed ^ (phase * _PHASE_MUL)) & MASK64)
    h = mix64(h ^ (a & MASK64))
    h = mix64(h ^ (((b & 0xFFFFFFFF) << 32) | (counter & 0xFFFFFFFF)))
    return h


def index_from(h, n):
    """Map a 64-bit draw to an index in [0, n): high 32 bits, multiply-shift."""
    return ((h >> 32) * n) >> 32


def unit_from(h):
    """Map a 64-bit draw to a double in [0, 1) with 53 random bits."""
    return (h >> 11) * (1.0 / 9007199254740992.0)


class DrawContext(object):
    """Mutable cursor: which (seed, phase, a, b) the next draws belong to."""

    def __init__(self, seed=1):
        self.seed = seed
        self.table_counter = 0
        self.point(PHASE_TABLE, 0, 0)

    def reseed(self, seed):
        self.seed = seed
        self.phase = None
        self.table_counter = 0
        self.point(PHASE_TABLE, 0, 0)

    def point(self, phase, a=0, b=0, counter=0):
        # table draws form one stream for the whole scene build: keep its position
        if getattr(self, "phase", None) == PHASE_TABLE:
            self.table_counter = self.counter
        self.phase = phase
        self.a = a
        self.b = b
        self.counter = self.table_counter if phase == PHASE_TABLE else counter

    def next64(self):
        h = draw64(self.seed, self.phase, self.a, self.b, self.counter)
        self.counter += 1
        return h

    def index(self, n):
        return index_from(self.next64(), n)

    def unit(self):
        return unit_from(self.next64())


#: the context the host samplers (and a patched `random` module) draw from
context = DrawContext()


def choice(seq):
    """Stand-in for `random.choice`."""
    return seq[context.index(len(seq))]


def shuffle(lst):
    """Stand-in for `random.shuffle`: Fisher-Yates from the back, one draw per swap."""
    for i in range(len(lst) - 1, 0, -1):
        j = context.index(i + 1)
        lst[i], lst[j] = lst[j], lst[i]


def uniform(lo, hi):
    """Stand-in for `random.uniform`."""
    return lo + (hi - lo) * context.unit()


# ---------------------------------------------------------------------------------------
# Vectorised forms (numpy uint64 arithmetic wraps modulo 2**64, as the scalar code masks).

def _np():
    import numpy
    return numpy


def mix64_array(z):
    numpy = _np()
    with numpy.errstate(over="ignore"):
        z = z + numpy.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> numpy.uint64(30))) * numpy.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> numpy.uint64(27))) * numpy.uint64(0x94D049BB133111EB)
        return z ^ (z >> numpy.uint64(31))


def draw64_array(seed, phase, a, b=0, counter=0):
    """draw64 for arrays of `a` (and optionally `b`)."""
    numpy = _np()
    h0 = numpy.uint64(mix64((seed ^ (phase * _PHASE_MUL)) & MASK64))
    a = numpy.asarray(a).astype(numpy.uint64)
    h = mix64_array(h0 ^ a)
    low = (numpy.asarray(b).astype(numpy.uint64) << numpy.uint64(32)) | numpy.uint64(counter & 0xFFFFFFFF)
    return mix64_array(h ^ low)


def index_array(h, n):
    numpy = _np()
    return (((h >> numpy.uint64(32)) * numpy.uint64(n)) >> numpy.uint64(32)).astype(numpy.int64)
