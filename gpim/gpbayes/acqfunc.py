"""``gpim.gpbayes.acqfunc`` -> gpim_amd.acqfunc (reference: gpim/gpbayes/acqfunc.py:11-92)."""
from gpim_amd.acqfunc import (confidence_bound, expected_improvement,                  # noqa: F401
                              probability_of_improvement)
