"""``gpim.gprutils`` -> gpim_amd.gprutils (reference: gpim/gprutils.py:23-210)."""
from gpim_amd.gprutils import *                   # noqa: F401,F403
from gpim_amd.gprutils import (get_full_grid, get_sparse_grid, prepare_test_data,      # noqa: F401
                               prepare_training_data)
