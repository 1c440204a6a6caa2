"""Pins the CPU restatement (oracle/lbl_oracle.c) to the reference's own compiled C via
the committed golden vectors -- CPU only, no GPU, no /root/reference at run time."""
import numpy as np
import pytest

from tests import golden_io


def test_voigt_profile_matches_reference(oracle):
    """voigt() alone (pyLBL/c_lib/voigt.c:4-191): every region boundary +- ulps, all y."""
    for grid, centre, alpha, gamma, strength, k_ref in golden_io.load_voigt():
        k = oracle.voigt_port(grid, 0, grid.size - 1, centre, alpha, gamma, strength)
        # Same libm, same operation order, no contraction: equal to the last bit.
        assert np.array_equal(k, k_ref)


@pytest.mark.parametrize("group", golden_io.ABSORPTION_GROUPS)
def test_absorption_matches_reference(oracle, group):
    """absorption() (pyLBL/c_lib/absorption.c:19-99) on synthetic databases."""
    table, cases = golden_io.load_group(group)
    for case in cases:
        assert case.rc == 0
        k, _ = oracle.absorption_port(table, case.temperature, case.pressure, case.vmr,
                                      case.v0, case.vn, case.n_per_v, cut_off=case.cut_off,
                                      remove_pedestal=case.remove_pedestal)
        assert k.shape == case.k.shape
        assert np.array_equal(k, case.k), \
            f"{group}[{case.index}] max rel {np.max(np.abs(k - case.k)/np.abs(case.k).max())}"


def test_break_quirk_is_in_the_vectors():
    """absorption.c:80-83: first row below v0-26 zeroes the spectrum; a row out of range in
    the middle drops every later row."""
    _, cases = golden_io.load_group("break_first")
    assert not cases[0].k.any()
    _, cases = golden_io.load_group("break_middle")
    assert cases[0].k.any()


def test_pedestal_depends_on_row_order():
    """spectra.c:66-78: the pedestal is taken from the accumulated spectrum."""
    _, a = golden_io.load_group("h2o40")
    _, b = golden_io.load_group("h2o40_swapped")
    same_args = [c for c in a if c.n_per_v == 10 and c.pressure == 98388.]
    no_ped = [c for c in same_args if not c.remove_pedestal][0]
    ped = [c for c in same_args if c.remove_pedestal][0]
    assert np.allclose(no_ped.k, b[0].k, rtol=1e-12, atol=0.)
    assert np.max(np.abs(ped.k - b[1].k)) > 1e-7*np.max(ped.k)


def test_no_tips_and_unknown_alias_return_codes():
    """absorption.c:53-59 (rc 0, zeros) and spectral_database.c:152-156 (rc 1)."""
    _, cases = golden_io.load_group("no_tips")
    assert cases[0].rc == 0 and not cases[0].k.any()
    _, cases = golden_io.load_group("unknown_alias")
    assert cases[0].rc == 1


def test_tips_interpolation_known_form(oracle):
    """spectral_database.c:97-104 against the Python twin (pyLBL/tips.py:26-39)."""
    from pylbl_amd.database import TotalPartitionFunction
    table, _ = golden_io.load_group("iso_ten")
    tips = TotalPartitionFunction("O3", table.tips_temperature, table.tips_data)
    for temperature in (203.37, 288.99, 296.5, 150.25):
        for iso in (1, 5, 10):
            a = oracle.tips_port(table.tips_temperature, table.tips_data, temperature, iso - 1)
            b = tips.total_partition_function(temperature, iso)
            assert a == pytest.approx(b, rel=1e-14)
