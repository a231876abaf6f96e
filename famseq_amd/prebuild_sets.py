"""Seeded pedigree recipes shared by build() (which pre-compiles their generated kernels into the in-tree cache) and
by the tests that run them on the GPU box — one definition, so the two cannot drift, and build() needs neither pytest
nor the oracle (the recipes only grow pedigrees; nothing here computes a posterior)."""
import numpy as np

from .synth import grow_pedigree

SOAK_SEEDS = range(40)          # tests/test_gpu_soak.py
RANDOM_SEEDS = range(10)        # tests/test_gpu_random_pedigrees.py, tests/test_generated_host.py
WIDE_SIZES = (24, 32, 48)       # tests/golden/wide_peds.npz (oracle/gen_golden_wide.py)
WIDE_RANDOM_SEEDS = range(6)    # tests/test_gpu_wide.py
WIDE_EXTRA_SIZES = (128,)       # beyond what LDS rows could stage: tests/test_wide.py, tests/test_gpu_wide.py (against the numpy oracle)


def soak_pedigree(seed, max_n=10):
    """-> (rng positioned after the pedigree draw, pedigree, mutation rate)."""
    rng = np.random.RandomState(9000 + seed)
    n = int(rng.randint(3, max_n + 1))
    ped = grow_pedigree(rng, n, allow_loops=seed % 2 == 0)
    ped.relations()
    return rng, ped, [1e-7, 1e-4, 0.0][seed % 3]


def random_pedigree(seed):
    """-> (rng positioned after the pedigree draw, pedigree): 3-9 members, loops on every third seed."""
    rng = np.random.RandomState(1000 + seed)
    ped = grow_pedigree(rng, int(rng.randint(3, 10)), allow_loops=seed % 3 == 0)
    return rng, ped


def wide_pedigree(n):
    """The fixture pedigree of n > 20 members: loop-free, about a quarter unsequenced (`NA`)."""
    return grow_pedigree(np.random.RandomState(7000 + n), n, allow_loops=False)


def wide_random_pedigree(seed):
    """-> (rng positioned after the pedigree draw, pedigree, mutation rate): 21-60 members, loop-free."""
    rng = np.random.RandomState(4200 + seed)
    ped = grow_pedigree(rng, int(rng.randint(21, 61)), allow_loops=False)
    return rng, ped, [1e-7, 1e-4, 0.0][seed % 3]
