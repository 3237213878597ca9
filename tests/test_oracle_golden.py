"""The CPU restatement (oracle/rt_oracle.c) against the reference's own outputs.

tests/golden/*.npz were produced by oracle/make_golden.py running /root/reference
(converted to Python 3) on seeded samples; every scene below is traced again by the C
oracle from the flattened host tables, and primary-hit identities must match exactly."""
import pytest

from tests import common


@pytest.mark.parametrize("scene", common.GOLDEN_SCENES)
def test_oracle_matches_reference(scene):
    from raytracer_amd import flatten
    from oracle import oracle
    world = common.world_for(scene)
    osc = oracle.OracleScene(common.flat_for(scene))
    pix, smp, plane, lens = common.golden_inputs(scene)
    out = osc.render_pixels(flatten.camera_record(world), world.seed, pix, smp, plane, lens, threads=8)
    worst = common.compare_with_golden(scene, out)
    print("%s: worst relative radiance error vs reference %.3g" % (scene, worst))
