"""A seeded 40-pedigree slice of the GPU soak in the suite (tools/soak_gpu.py runs the long ones):
random pedigrees of 3-10 members, half of them with marriage loops (0-3 conditioned members),
batch sizes around the chunk boundaries, mu in {1e-7, 1e-4, 0}, every engine and the fused call
path against the oracle.  build() pre-compiles the generated kernels of these very pedigrees
(__graft_entry__.prebuild_generated_kernels), so the box does not spend the test on hipcc."""
import pytest

from _soak import run_seed

from famseq_amd.prebuild_sets import SOAK_SEEDS

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", SOAK_SEEDS)
def test_soak_seed(seed):
    line, bad = run_seed(seed)
    assert not bad, line
