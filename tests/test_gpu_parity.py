"""GPU parity tests proper: the HIP path (through the C ABI) against the oracle on the same seeded inputs.
Bit-exact: identical path sets in identical order, exact integer fields, bit-identical FP64 weight / abd / reads;
conf (exp of summed libm logs, reported only) within 1e-9 relative."""
import numpy as np
import pytest

import aletsch_amd as A
import common

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(common.PARITY_CONFIGS))
def test_hip_matches_oracle(name):
    pg = A.synth(**common.PARITY_CONFIGS[name])
    want = common.oracle_run(pg)[0]
    got = A.decompose(pg, device=0)
    bad = common.compare_results(want, got, pg.n, conf_tol=1e-9)
    assert not bad, f"{name}: {len(bad)} mismatches, first {bad[:3]}"
    assert int((got.status != 0).sum()) == 0


def test_iteration_counts_match_oracle():
    pg = A.synth(**common.PARITY_CONFIGS["cfg2_64v256e"])
    _, st, _, _ = common.oracle_run(pg)
    with A.DecompBatch(0) as b:
        b.add(pg); b.upload(); b.run(); b.download()
        it = b.iterations()
    assert np.array_equal(it, st[:, 3])
