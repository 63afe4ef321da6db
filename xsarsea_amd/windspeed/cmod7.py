"""CMOD7 table model (reference: src/xsarsea/windspeed/cmod7.py:10-106).

File `gmf_cmod7_vv.dat_little_endian`: float32 little-endian words, one record marker word at each
end, then a Fortran-ordered (250 wspd x 73 phi x 51 incidence) table in linear units on the
low-resolution grid (0.2..50 m/s step 0.2, 0..180 deg step 2.5, 16..66 deg step 1).
"""
import os

import numpy as np

from .lut import Lut
from .models import ArrayLutModel

N_WSPD, N_PHI, N_INC = 250, 73, 51


def read_cmod7_table(path):
    raw = np.fromfile(path, dtype="<f4")
    if raw.size != N_WSPD * N_PHI * N_INC + 2:
        raise ValueError(f"{path}: expected {N_WSPD * N_PHI * N_INC + 2} float32 words, found {raw.size}")
    table = raw[1:-1].reshape((N_WSPD, N_PHI, N_INC), order="F")
    wspd = np.arange(0.2, 50.0 + 0.2, 0.2)[:N_WSPD]
    phi = np.arange(0, 180 + 2.5, 2.5)[:N_PHI]
    inc = np.arange(16, 66 + 1, 1).astype(np.float64)[:N_INC]
    return Lut(np.ascontiguousarray(np.transpose(table, (2, 0, 1)), dtype=np.float64), inc, wspd, phi,
               units="linear", resolution="low")


def write_cmod7_table(path, table_wpi):
    """Inverse of `read_cmod7_table` for a (250, 73, 51) array (used to synthesise test tables)."""
    t = np.asarray(table_wpi, dtype="<f4")
    assert t.shape == (N_WSPD, N_PHI, N_INC)
    words = np.concatenate([[np.float32(0)], t.reshape(-1, order="F"), [np.float32(0)]]).astype("<f4")
    words.tofile(path)


def register_cmod7(topdir):
    """Register `gmf_cmod7` (pol VV) from the directory holding `gmf_cmod7_vv.dat_little_endian`."""
    if not os.path.isdir(topdir):
        raise FileNotFoundError(topdir)
    path = os.path.join(topdir, "gmf_cmod7_vv.dat_little_endian")
    if not os.path.isfile(path):
        raise FileNotFoundError(path)
    lut = read_cmod7_table(path)
    return ArrayLutModel("gmf_cmod7", lut, pol="VV", inc_range=[16, 66], wspd_range=[0.2, 50.0], phi_range=[0, 180],
                         wspd_step_lr=0.2, inc_step_lr=1, phi_step_lr=2.5)
