"""sarwing LUT directories (reference: src/xsarsea/windspeed/pickle_luts.py:11-133).

A LUT is a directory `GMF_<name>/` holding `sigma.npy` (dB, stored (incidence, phi, wspd) resp.
(incidence, wspd), i.e. transposed with respect to the axes pickles), `incidence_angle.pkl` and either
`wind_speed_and_direction.pkl` (-> (phi, wspd), co-pol) or `wind_speed.pkl` (cross-pol).  The tables are at
"high" resolution already; models are registered as `sarwing_lut__<name>`.
"""
import os
import pickle

import numpy as np

from .lut import Lut
from .models import LutModel


def _load_pickle(path):
    with open(path, "rb") as f:
        return np.asarray(pickle.load(f, encoding="iso-8859-1"), dtype=np.float64)


def _step(axis):
    return float(np.round(np.unique(np.diff(axis)), decimals=2)[0])


def _range(axis):
    return [float(np.round(np.min(axis), decimals=2)), float(np.round(np.max(axis), decimals=2))]


class PickleLutModel(LutModel):
    _name_prefix = "sarwing_lut__"
    _priority = 10

    def __init__(self, name, path, **kwargs):
        super().__init__(name, **kwargs)
        self.path = path

    def _raw_lut(self, **kwargs):
        if not os.path.isdir(self.path):
            raise FileNotFoundError(self.path)
        sigma_db = np.ascontiguousarray(np.transpose(np.load(os.path.join(self.path, "sigma.npy"))), dtype=np.float64)
        inc = _load_pickle(os.path.join(self.path, "incidence_angle.pkl"))
        both = os.path.join(self.path, "wind_speed_and_direction.pkl")
        if os.path.exists(both):
            with open(both, "rb") as f:
                phi, wspd = (np.asarray(a, dtype=np.float64) for a in pickle.load(f, encoding="iso-8859-1"))
        else:
            phi, wspd = None, _load_pickle(os.path.join(self.path, "wind_speed.pkl"))
        self.wspd_step, self.inc_step = _step(wspd), _step(inc)
        self.inc_range, self.wspd_range = _range(inc), _range(wspd)
        if phi is not None:  # stored (wspd, phi, incidence) after the transpose
            values = np.transpose(sigma_db, (2, 0, 1))
            self.phi_step, self.phi_range = _step(phi), _range(phi)
            self.inc_step_lr, self.wspd_step_lr, self.phi_step_lr = 1.0, 0.4, 2.5
        else:  # (wspd, incidence)
            values = np.transpose(sigma_db, (1, 0))
            self.inc_step_lr, self.wspd_step_lr, self.phi_step_lr = 1.0, 0.1, 1
        return Lut(np.ascontiguousarray(values), inc, wspd, phi, units="dB", resolution="high", model=self.name)


def register_pickle_luts(path):
    """Register one `GMF_*` LUT directory, or every `GMF_*` directory found directly under `path`."""
    def register_one(lut_dir):
        name = os.path.basename(lut_dir).replace("GMF_", PickleLutModel._name_prefix)
        if os.path.exists(os.path.join(lut_dir, "wind_speed_and_direction.pkl")):
            pol = "VV"
        elif os.path.exists(os.path.join(lut_dir, "wind_speed.pkl")):
            pol = "VH"
        else:
            pol = None
        PickleLutModel(name, lut_dir, pol=pol)

    if os.path.basename(os.path.normpath(path)).startswith("GMF_"):
        register_one(path)
    elif os.path.isdir(path):
        for entry in os.listdir(path):
            full = os.path.join(path, entry)
            if os.path.isdir(full) and entry.startswith("GMF_"):
                register_one(full)
