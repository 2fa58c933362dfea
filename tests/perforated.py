"""A box of hexes with a fraction of its elements knocked out (largest connected component kept): the
irregular-mesh class of tests/fuzz.py at a chosen size -- now part of the workload generator (stan_amd/cube.py)."""
from stan_amd.cube import perforated_mesh  # noqa: F401
from stan_amd.problem import perforated_job  # noqa: F401
