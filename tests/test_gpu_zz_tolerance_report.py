"""Runs last (file name): reports how many pedestal comparisons of this session needed the
conditioning allowance of tests/test_gpu_parity.py::assert_spectrum, and bounds them.

The allowance exists for one property of the reference's recurrence (spectra.c:66-78): with line
centres exactly on window edges it amplifies last-bit differences of the profile values by many
orders of magnitude.  It is measured, per case, by re-running the oracle on
inputs perturbed by one unit in the last place.  Only the seeded fuzz builds such tables; none of
the BASELINE-size workloads and none of the golden vectors may need it."""
import pytest

from tests.test_gpu_parity import CONDITIONING_MARGIN, TOLERANCE_LOG

pytestmark = pytest.mark.gpu


def test_conditioning_allowance_is_rare():
    conditioned = TOLERANCE_LOG["conditioned"]
    compared = TOLERANCE_LOG["compared"]
    print(f"\npedestal comparisons: {compared}; needed the conditioning allowance "
          f"({CONDITIONING_MARGIN:g} x the reference's own movement under a one-ulp perturbation): "
          f"{len(conditioned)}")
    for record in conditioned:
        print(f"  {record['label']}: {record['worst_plain']:.3g} x the plain tolerance, "
              f"{record['worst_conditioned']:.3g} x with the allowance; the reference itself "
              f"moves by {record['reference_moves_by']:.3g} of its maximum")
    for label, points, worst, unit in TOLERANCE_LOG["baseline"]:
        print(f"  {label}: {points} points, worst {worst:.3g} ({unit})")
    assert not any("baseline" in record["label"] for record in conditioned)
    # Soaks (PYLBL_FUZZ_CASES=20000) meet a handful; the regular run must not need more than
    # one case in a hundred.
    assert len(conditioned) <= max(2, compared//100)
